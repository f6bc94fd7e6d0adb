# round 4, job 2: the folder driver (batched + chunked pipeline) and the batched chain
set -e -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "graph_replay or batched_and_chunked or lanes" > $O/r04_gpu_tests_b.log 2>&1 || { tail -40 $O/r04_gpu_tests_b.log; exit 1; }
tail -2 $O/r04_gpu_tests_b.log
D=$O/r04_driver.jsonl; : > $D
run() { echo "## $*" >> $D; timeout -k 10 300 python tools/driver_bench.py "$@" >> $D 2>> $O/r04_driver.err; tail -1 $D | cut -c1-330; }
run --size 512 --keys 5 --rate 32
run --size 512 --keys 5 --rate 32 --batch 1 --lanes 3 --chunk 0 --io-threads 16
run --size 512 --keys 5 --rate 32 --batch 1 --lanes 3 --chunk 8
run --size 512 --keys 5 --rate 32 --batch 4 --lanes 1 --chunk 8
run --size 512 --keys 5 --rate 32 --batch 4 --lanes 2 --chunk 4
run --size 512 --keys 5 --rate 32 --batch 4 --lanes 2 --chunk 16
run --size 512 --keys 5 --rate 32 --batch 2 --lanes 2 --chunk 8
run --size 512 --keys 5 --rate 32 --io-threads 16
run --size 512 --keys 5 --rate 32 --io-threads 64
run --size 512 --keys 5 --rate 32 --compress 1
run --size 512 --keys 3 --rate 32
run --size 512 --keys 5 --rate 32 --dtype bf16
run --height 320 --width 480 --keys 9 --rate 16
run --height 320 --width 480 --keys 9 --rate 16 --batch 1 --lanes 3 --chunk 0 --io-threads 16
run --height 320 --width 480 --keys 9 --rate 16 --dtype bf16
S=$O/r04_chain_shapes.jsonl; : > $S
for flags in "--mode chain --frames 32" "--mode chain --frames 32 --batch 4" "--mode chain --frames 8 --batch 4" "--mode chain --frames 32 --batch 2" "--mode chain --frames 32 --batch 8" "--mode chain --frames 32 --batch 4 --graph" "--height 320 --width 480 --mode chain --frames 16 --batch 8" "--height 320 --width 480 --mode chain --frames 16" "--mode clips --frames 32" "--mode clips --frames 32 --graph" "--dtype bf16 --mode chain --frames 32 --batch 4" "--dtype bf16 --mode clips --frames 32" "--dtype bf16 --mode clips --frames 32 --graph"; do
  echo "## $flags" >> $S
  python bench.py --no-cpu-baseline --steps 10 --warmup 3 $flags >> $S 2>> $O/r04_chain_shapes.err
  tail -1 $S | python -c "
import json,sys
j=json.loads(sys.stdin.read()); c=j['config']
print('%-70s %8.1f fps  %.3f ms/frame  enqueue %.2f of %.2f ms/step  graph %s' % ('$flags', j['value'], j['ms_per_step']/c['frames_per_step_per_gpu'], c['per_rank_host_enqueue_ms_per_step'][0], c['per_rank_total_ms_per_step'][0], c['graph_replay']))"
done
