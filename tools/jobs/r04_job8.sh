set -e -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_bench.py -x -q -m gpu 2>&1 | tail -15
