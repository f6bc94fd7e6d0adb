# round 4, job 3: pruned library (137 variant entries, 24 shard objects): full GPU suite, frame time, the driver with worker processes,
# and the per-op trace of the batch-4 frame
set -e -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/r04_gpu_tests_c.log 2>&1 || { tail -40 $O/r04_gpu_tests_c.log; exit 1; }
tail -2 $O/r04_gpu_tests_c.log
python bench.py --no-cpu-baseline > $O/r04_bench_pruned.json 2> $O/r04_bench_pruned.err
python -c "
import json; j=json.load(open('$O/r04_bench_pruned.json')); print('pruned lib: %.1f fps %.4f ms frac %.4f' % (j['value'], j['ms_per_step'], j['roofline']['frac']), j['config']['build'][:60])"
D=$O/r04_driver2.jsonl; : > $D
run() { echo "## $*" >> $D; timeout -k 10 300 python tools/driver_bench.py "$@" >> $D 2>> $O/r04_driver2.err; tail -1 $D | python -c "
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
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof_ops_b4 -- python3 $R/tools/prof_ops.py --run --batch 4 > $R/$O/r04_prof_ops_b4.log 2>&1
python3 $R/tools/prof_ops.py --report $R/$O/prof_ops_b4 --batch 4 > $R/$O/r04_prof_ops_512_b4.txt
rm -rf $R/$O/prof_ops_b4
head -8 $R/$O/r04_prof_ops_512_b4.txt
