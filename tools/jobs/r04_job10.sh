set -e -o pipefail
cd $GRAFT_REPO_ROOT
run() { env "$@" python bench.py --no-cpu-baseline --steps 200 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('%-48s %.4f ms  %d launches' % ('$*', j['ms_per_step'], j['config']['launches_per_step']))"; }
run A=0
run RIB_WINO_MAX_PX=65536
run RIB_WINO_MAX_PX=4096
run RIB_COND_GEMM_MAX_PX=16384
run RIB_COND_GEMM_MAX_PX=1024
run RIB_COND_GEMM_MAX_PX=0
run RIB_WINO_M=2
run RIB_WINO_M=4
run A=1
runb() { env "$@" python bench.py --no-cpu-baseline --steps 50 --batch 4 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('B=4 %-44s %.4f ms/frame' % ('$*', j['ms_per_step']/4))"; }
runb A=0
runb RIB_WINO_MAX_PX=65536
runb RIB_WINO_MAX_PX=4096
runb RIB_COND_GEMM_MAX_PX=16384
runb RIB_COND_GEMM_MAX_PX=1024
