set -e -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_ops_bf16 -- python3 $R/tools/prof_ops.py --run --dtype bf16 > $R/gpurun_out/bf16_prof_ops.log 2>&1
python3 $R/tools/prof_ops.py --report $R/gpurun_out/prof_ops_bf16 --dtype bf16 --json $R/gpurun_out/bf16_prof_ops_512.json > $R/gpurun_out/bf16_prof_ops_512.txt
rm -rf $R/gpurun_out/prof_ops_bf16
head -12 $R/gpurun_out/bf16_prof_ops_512.txt
