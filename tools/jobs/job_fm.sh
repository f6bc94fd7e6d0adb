set -e -o pipefail
cd $GRAFT_REPO_ROOT
for v in head cur head cur; do
  if [ $v = cur ]; then export RIB_LIBRARY=$PWD/render-in-between_amd/csrc/librib.so; else export RIB_LIBRARY=$PWD/render-in-between_amd/csrc/ab/librib_$v.so; fi
  for only in gammabeta wino; do
    timeout -k 10 300 python3 tools/autotune.py --size 512 --batch 1 --iters 50 --only $only --out gpurun_out/fm_tmp_table.json --report gpurun_out/fm_${v}_$only.json > gpurun_out/fm_${v}_$only.log 2>&1
  done
  echo "== $v"; grep -h "default" gpurun_out/fm_${v}_gammabeta.log gpurun_out/fm_${v}_wino.log | cut -c1-140
done
