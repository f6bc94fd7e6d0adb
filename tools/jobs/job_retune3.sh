set -e -o pipefail
cd $GRAFT_REPO_ROOT
cp render-in-between_amd/tuning_gfx950.json gpurun_out/tuning_before3.json
for i in 1 2; do echo -n "before fp32: "; python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value'],1), round(r['ms_per_step'],4))"; done
timeout -k 10 900 python3 tools/autotune.py --size 512 --batch 1 --iters 100 > gpurun_out/retune3_f32.log 2>&1; tail -1 gpurun_out/retune3_f32.log
cp render-in-between_amd/tuning_gfx950.json gpurun_out/tuning_after3.json
for i in 1 2; do echo -n "after fp32: "; python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value'],1), round(r['ms_per_step'],4))"; done
cp gpurun_out/tuning_before3.json render-in-between_amd/tuning_gfx950.json
echo -n "before again: "; python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value'],1), round(r['ms_per_step'],4))"
