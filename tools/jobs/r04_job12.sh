set -e -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/final_gpu_tests2.log 2>&1 || { tail -30 gpurun_out/final_gpu_tests2.log; exit 1; }
tail -2 gpurun_out/final_gpu_tests2.log
python -c "import __graft_entry__ as g; g.smoke()"
python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err
python - <<'PY'
import json
j = json.load(open("gpurun_out/r04_bench_final.json")); r = j["roofline"]
print("%.1f fps %.4f ms frac %.4f (%s) live %.4f rocprof %s traffic %s" % (j["value"], j["ms_per_step"], r["frac"], r["frac_basis"][:30], r["live_frac"], r["rocprof_basis"] and round(r["rocprof_basis"]["frac"], 4), r["traffic_bytes_per_step_all_classes"]))
print(j["config"]["build"][:60], j["cpu_baseline"]["value"], j["parity"])
PY
