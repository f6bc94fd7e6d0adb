set -e -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm_dma_tile or dma_staged or single_frame or config2" > gpurun_out/ab_test.log 2>&1 || { tail -30 gpurun_out/ab_test.log; exit 1; }
tail -2 gpurun_out/ab_test.log
C=render-in-between_amd/csrc
timeout -k 10 600 python3 tools/ab_lib.py --tuning render-in-between_amd/tuning_gfx950.json $C/ab/librib_head.so $C/ab/librib_fm.so $C/librib.so > gpurun_out/ab_fm.txt 2>&1
cat gpurun_out/ab_fm.txt
