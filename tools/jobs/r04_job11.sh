set -e -o pipefail
cd $GRAFT_REPO_ROOT
run() { env "$@" python tools/driver_bench.py --size 512 --keys 5 --rate 32 --reps 5 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-60s %6.1f fps  runs %s' % ('$*', j['frames_per_s_end_to_end'], j['wall_s_runs']))"; }
run A=1
run RIB_DECODE_AHEAD=1000 RIB_MAX_UNITS_IN_FLIGHT=1000
run A=2
run RIB_DECODE_AHEAD=1000 RIB_MAX_UNITS_IN_FLIGHT=1000
run RIB_DECODE_AHEAD=3
run RIB_DECODE_AHEAD=12
