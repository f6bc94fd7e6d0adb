set -e -o pipefail
cd $GRAFT_REPO_ROOT
RIB_BENCH_DEVICE=0 RIB_DIST_BACKEND=gloo timeout -k 10 400 python3 tools/multirank_inference_check.py --gpus 5 --keys 9 6 5 2> gpurun_out/j_mr_5.err || { tail -20 gpurun_out/j_mr_5.err; exit 1; }
for flags in "" "--graph"; do
RIB_BENCH_DEVICE=0 RIB_DIST_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus 5 --mode clips --frames 16 --steps 2 --warmup 1 $flags 2>> gpurun_out/j_clips5.err | grep '^{' | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print(json.dumps({'value': d['value'], 'replica_check': c['replica_check'], 'enq': c['per_rank_host_enqueue_ms_per_step'], 'tot': c['per_rank_total_ms_per_step'], 'graph': c['graph_replay']}))"
done
