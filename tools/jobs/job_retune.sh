set -e -o pipefail
cd $GRAFT_REPO_ROOT
for spec in "256 1" "512 2" "512 4" "512 8" "1024 1" "1024 4"; do
  set -- $spec
  echo "[retune] fp32 size $1 batch $2"
  timeout -k 10 1000 python3 tools/autotune.py --size $1 --batch $2 --iters 12 > gpurun_out/retune_f32_$1_$2.log 2>&1
  tail -1 gpurun_out/retune_f32_$1_$2.log
done
for spec in "512 1" "1024 4"; do
  set -- $spec
  echo "[retune] bf16 size $1 batch $2"
  timeout -k 10 1000 python3 tools/autotune.py --size $1 --batch $2 --iters 12 --dtype bf16 > gpurun_out/retune_bf16_$1_$2.log 2>&1
  tail -1 gpurun_out/retune_bf16_$1_$2.log
done
cp render-in-between_amd/tuning_gfx950.json gpurun_out/tuning_gfx950_retuned.json
cp render-in-between_amd/tuning_gfx950_bf16.json gpurun_out/tuning_gfx950_bf16_retuned.json
for flags in "--size 256" "--batch 2" "--batch 4" "--batch 8" "--size 1024" "--size 1024 --batch 4" "--dtype bf16" "--size 1024 --batch 4 --dtype bf16" "--dtype f16"; do
  echo "## $flags"; python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 $flags 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value'],1), round(r['ms_per_step'],3))"
done
