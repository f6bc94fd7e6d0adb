set -e -o pipefail
cd $GRAFT_REPO_ROOT
cp render-in-between_amd/tuning_gfx950.json gpurun_out/tuning_before.json
cp render-in-between_amd/tuning_gfx950_bf16.json gpurun_out/tuning_bf16_before.json
echo -n "before fp32: "; python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value'],1), round(r['ms_per_step'],4))"
echo -n "before bf16: "; python3 bench.py --no-cpu-baseline --dtype bf16 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value'],1), round(r['ms_per_step'],4))"
timeout -k 10 600 python3 tools/autotune.py --size 512 --batch 1 --iters 30 --report gpurun_out/autotune_512_epi.txt > gpurun_out/retune2_f32.log 2>&1; tail -1 gpurun_out/retune2_f32.log
timeout -k 10 600 python3 tools/autotune.py --size 512 --batch 1 --iters 30 --dtype bf16 > gpurun_out/retune2_bf16.log 2>&1; tail -1 gpurun_out/retune2_bf16.log
cp render-in-between_amd/tuning_gfx950.json gpurun_out/tuning_after.json
cp render-in-between_amd/tuning_gfx950_bf16.json gpurun_out/tuning_bf16_after.json
for i in 1 2; do
echo -n "after fp32: "; python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value'],1), round(r['ms_per_step'],4))"
echo -n "after bf16: "; python3 bench.py --no-cpu-baseline --dtype bf16 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value'],1), round(r['ms_per_step'],4))"
echo -n "after f16: "; python3 bench.py --no-cpu-baseline --dtype f16 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value'],1), round(r['ms_per_step'],4))"
done
