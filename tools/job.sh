set -e -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "warp" > gpurun_out/j_warp_tests.log 2>&1 || { tail -30 gpurun_out/j_warp_tests.log; exit 1; }
tail -2 gpurun_out/j_warp_tests.log
cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out
for th in 32 16; do
RIB_WARP_TH=$th python3 $ROOT/tools/warp_bench.py --time --out $OUT/r05_warp_$th.json > $OUT/r05_warp.log 2>&1 || { tail $OUT/r05_warp.log; exit 1; }
python3 -c "
import json
for r in json.load(open('$OUT/r05_warp_$th.json')): print($th, {k:(round(v,4) if isinstance(v,float) else v) for k,v in r.items() if k in ('H','flow_amplitude_px','us_per_launch_median_of_7','frac_of_8TBps','max_abs_vs_grid_sample')})
"
done
