set -e -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/epi_gpu_tests.log 2>&1 || { tail -30 gpurun_out/epi_gpu_tests.log; exit 1; }
tail -2 gpurun_out/epi_gpu_tests.log
C=render-in-between_amd/csrc
timeout -k 10 600 python3 tools/ab_lib.py --tuning render-in-between_amd/tuning_gfx950.json $C/ab/librib_prev.so $C/librib.so > gpurun_out/ab_epi.txt 2>&1
grep round gpurun_out/ab_epi.txt
for r in 1 2; do for v in prev cur; do
  if [ $v = cur ]; then export RIB_LIBRARY=$PWD/$C/librib.so; else export RIB_LIBRARY=$PWD/$C/ab/librib_$v.so; fi
  for flags in "--dtype bf16" "--dtype f16" "--batch 4" "--size 1024"; do
    echo -n "$v $flags: "; python3 bench.py --no-cpu-baseline --steps 30 --warmup 5 $flags 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value'],1), round(r['ms_per_step'],3))"
  done
done; done
