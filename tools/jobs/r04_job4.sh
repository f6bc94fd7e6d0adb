# round 4, job 4: the folder driver with worker processes (the __main__ re-import fixed); lowc stagger experiment
set -e -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
D=$O/r04_driver3.jsonl; : > $D
run() { echo "## $*" >> $D; timeout -k 10 300 python tools/driver_bench.py "$@" >> $D 2>> $O/r04_driver3.err; tail -1 $D | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('%-90s %6.1f fps  wall %.3f  %s' % ('$*', j['frames_per_s_end_to_end'], j['wall_s'], j['phase_s_last_run']))"; }
run --size 512 --keys 5 --rate 32
run --size 512 --keys 5 --rate 32 --io-mode thread
run --size 512 --keys 5 --rate 32 --io-threads 24
run --size 512 --keys 5 --rate 32 --io-threads 64
run --size 512 --keys 5 --rate 32 --io-threads 96
run --size 512 --keys 5 --rate 32 --chunk 4
run --size 512 --keys 5 --rate 32 --lanes 1
run --size 512 --keys 5 --rate 32 --compress 1
run --size 512 --keys 3 --rate 32
run --size 512 --keys 5 --rate 32 --dtype bf16
run --height 320 --width 480 --keys 9 --rate 16
run --height 320 --width 480 --keys 9 --rate 16 --io-threads 64
run --height 320 --width 480 --keys 9 --rate 16 --dtype bf16


