cd $GRAFT_REPO_ROOT
for thr in default 0 1024; do
  for dt in bf16 f16; do
    if [ $thr = default ]; then unset RIB_COND_GEMM_MAX_PX; else export RIB_COND_GEMM_MAX_PX=$thr; fi
    echo -n "RIB_COND_GEMM_MAX_PX=$thr $dt: "; python3 bench.py --no-cpu-baseline --dtype $dt --steps 40 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value'],1), round(r['ms_per_step'],4), r['config']['launches_per_step'])"
  done
done
unset RIB_COND_GEMM_MAX_PX
for w in default 0; do
  if [ $w = default ]; then unset RIB_NO_PAIR; else export RIB_NO_PAIR=1; fi
  echo -n "RIB_NO_PAIR=$w bf16: "; python3 bench.py --no-cpu-baseline --dtype bf16 --steps 40 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value'],1), round(r['ms_per_step'],4), r['config']['launches_per_step'])"
done
