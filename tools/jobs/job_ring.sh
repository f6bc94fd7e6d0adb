set -e -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm_dma_tile or dma_staged" > gpurun_out/ring_test.log 2>&1 || { tail -30 gpurun_out/ring_test.log; exit 1; }
tail -3 gpurun_out/ring_test.log
timeout -k 10 900 python3 tools/autotune.py --size 512 --batch 1 --report gpurun_out/autotune_512_ring.txt > gpurun_out/autotune_ring.log 2>&1
tail -5 gpurun_out/autotune_ring.log
cp render-in-between_amd/tuning_gfx950.json gpurun_out/tuning_gfx950_ring.json
python3 bench.py --no-cpu-baseline > gpurun_out/bench_ring.json 2> gpurun_out/bench_ring.err
cat gpurun_out/bench_ring.json
