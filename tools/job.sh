set -e -o pipefail
cd $GRAFT_REPO_ROOT
cp render-in-between_amd/tuning_gfx950.json gpurun_out/tuning_f32.json
cp render-in-between_amd/tuning_gfx950_bf16.json gpurun_out/tuning_bf16.json
run() { # dtype size width batch tag
  echo "== autotune $1 H=$2 W=$3 B=$4" 
  timeout -k 10 400 python3 tools/autotune.py --dtype $1 --size $2 --width $3 --batch $4 --out gpurun_out/tuning_$1.json > gpurun_out/r05_autotune_$5.txt 2>&1 || { tail -5 gpurun_out/r05_autotune_$5.txt; return 1; }
  tail -1 gpurun_out/r05_autotune_$5.txt
}
run f32 512 512 3 512_b3
run f32 320 480 2 320x480_b2
run f32 320 480 3 320x480_b3
run f32 256 256 8 256_b8
run f32 1024 1024 2 1024_b2
run bf16 1024 1024 2 1024_b2_bf16
run bf16 256 256 8 256_b8_bf16
