set -e -o pipefail
cd $GRAFT_REPO_ROOT
cp render-in-between_amd/tuning_gfx950_bf16.json gpurun_out/tuning_bf16_before.json
for dt in bf16 f16; do echo -n "before $dt: "; python3 bench.py --no-cpu-baseline --dtype $dt --steps 40 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value'],1), round(r['ms_per_step'],4))"; done
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bf16 or half or config3 or config5" > gpurun_out/g16_tests.log 2>&1 || { tail -30 gpurun_out/g16_tests.log; exit 1; }
tail -2 gpurun_out/g16_tests.log
timeout -k 10 300 python3 tools/autotune.py --size 512 --batch 1 --iters 50 --dtype bf16 --only gammabeta > gpurun_out/g16_tune.log 2>&1; grep gammabeta gpurun_out/g16_tune.log | cut -c1-150
cp render-in-between_amd/tuning_gfx950_bf16.json gpurun_out/tuning_bf16_after.json
for dt in bf16 f16; do echo -n "after $dt: "; python3 bench.py --no-cpu-baseline --dtype $dt --steps 40 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value'],1), round(r['ms_per_step'],4))"; done
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bf16 or half or config3 or config5" > gpurun_out/g16_tests2.log 2>&1 || { tail -30 gpurun_out/g16_tests2.log; exit 1; }
tail -2 gpurun_out/g16_tests2.log
