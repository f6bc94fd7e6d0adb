# The round's closing job on the GPU box (through gpurun from the repo root): the whole GPU suite, then every measured artefact.
#   gpurun --timeout 1200 -- 'bash tools/r06_final.sh'        copy gpurun_out/r06_* into profiles/ afterwards
set -e -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/final_gpu_tests.log 2>&1 || { tail -30 gpurun_out/final_gpu_tests.log; exit 1; }
tail -2 gpurun_out/final_gpu_tests.log
bash tools/refresh_profiles.sh r06 > gpurun_out/refresh.log 2>&1 || { tail -20 gpurun_out/refresh.log; exit 1; }
tail -3 gpurun_out/refresh.log
cut -c1-300 gpurun_out/r06_bench.json
