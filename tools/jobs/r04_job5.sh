set -e -o pipefail
cd $GRAFT_REPO_ROOT
cat /sys/fs/cgroup/cpu.max 2>/dev/null || true; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null || true
for f in "" "--chunk 4" "--chunk 2" "--chunk 4 --io-threads 12" "--chunk 4 --io-threads 20" "--chunk 4 --lanes 1" "--chunk 4 --compress 1" "--chunk 4 --io-mode thread --io-threads 16"; do
  python tools/driver_bench.py --size 512 --keys 5 --rate 32 $f > gpurun_out/tl.json 2> gpurun_out/tl.err || { tail -20 gpurun_out/tl.err; exit 1; }
  F="$f" python - <<'PY'
import json, os
j = json.loads([l for l in open("gpurun_out/tl.json") if l.startswith("{")][-1])
print("%-44s" % os.environ["F"], round(j["frames_per_s_end_to_end"], 1), round(j["wall_s"], 3), "workers", j["io_threads"], "budget", j["cpu_budget"], j["phase_s_last_run"])
tl = sorted(j["unit_timeline_s [decoded, enqueued, on host, written]"])
print("      first unit", tl[0], "last", tl[-1])
PY
done
