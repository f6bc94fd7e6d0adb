set -e -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in head fm cur; do
  if [ $v = cur ]; then export RIB_LIBRARY=$R/render-in-between_amd/csrc/librib.so; else export RIB_LIBRARY=$R/render-in-between_amd/csrc/ab/librib_$v.so; fi
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_ops_$v -- python3 $R/tools/prof_ops.py --run > $R/gpurun_out/ab_prof_$v.log 2>&1
  python3 $R/tools/prof_ops.py --report $R/gpurun_out/prof_ops_$v --json $R/gpurun_out/ab_ops_$v.json > $R/gpurun_out/ab_ops_$v.txt
  rm -rf $R/gpurun_out/prof_ops_$v
  head -7 $R/gpurun_out/ab_ops_$v.txt
done
