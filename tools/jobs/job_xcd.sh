set -e -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm_dma_tile" > gpurun_out/xcd_test.log 2>&1 || { tail -30 gpurun_out/xcd_test.log; exit 1; }
tail -2 gpurun_out/xcd_test.log
for g in 0 8 4 16 8 0; do
  echo "## RIB_GEMM_GROUP_M=$g"
  RIB_GEMM_GROUP_M=$g python3 bench.py --no-cpu-baseline 2> gpurun_out/xcd_bench.err | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); c=r['roofline']['classes']
print(r['ms_per_step'], 'igemm', c['igemm']['ms_per_step'], 'spade', c['spade']['ms_per_step'], 'aux', c['conv_aux']['ms_per_step'])"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ops_xcd -- python3 $GRAFT_REPO_ROOT/tools/prof_ops.py --run > $GRAFT_REPO_ROOT/gpurun_out/xcd_prof_ops.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/prof_ops.py --report $GRAFT_REPO_ROOT/gpurun_out/prof_ops_xcd > $GRAFT_REPO_ROOT/gpurun_out/xcd_prof_ops_512.txt
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_ops_xcd
grep -E "gemm" $GRAFT_REPO_ROOT/gpurun_out/xcd_prof_ops_512.txt
