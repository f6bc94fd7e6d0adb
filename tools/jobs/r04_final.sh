set -e -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/final_gpu_tests.log 2>&1 || { tail -30 gpurun_out/final_gpu_tests.log; exit 1; }
tail -2 gpurun_out/final_gpu_tests.log
bash tools/refresh_profiles.sh r04 > gpurun_out/refresh.log 2>&1 || { tail -20 gpurun_out/refresh.log; exit 1; }
tail -3 gpurun_out/refresh.log
cat gpurun_out/r04_bench.json | cut -c1-300
