# round 4, first GPU job on the stamped build: GPU tests, default bench (live kernel-time roofline), tables for the
# reference's default 320x480 and for the batched chain shapes
set -e -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -c "import os; print('cpus', len(os.sched_getaffinity(0)), os.cpu_count())" > $O/r04_cpus.txt
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/r04_gpu_tests_a.log 2>&1 || { tail -40 $O/r04_gpu_tests_a.log; exit 1; }
tail -2 $O/r04_gpu_tests_a.log
python bench.py > $O/r04_bench_a.json 2> $O/r04_bench_a.err
python bench.py --no-cpu-baseline --height 320 --width 480 --no-tuning > $O/r04_bench_320x480_model.json 2>> $O/r04_bench_a.err
python bench.py --no-cpu-baseline --height 320 --width 480 --dtype bf16 --no-tuning > $O/r04_bench_320x480_bf16_model.json 2>> $O/r04_bench_a.err
cp render-in-between_amd/tuning_gfx950.json $O/tuning_gfx950.json
cp render-in-between_amd/tuning_gfx950_bf16.json $O/tuning_gfx950_bf16.json
python tools/autotune.py --size 320 --width 480 --batch 1 --out $O/tuning_gfx950.json > $O/r04_autotune_320x480.txt 2>&1
python tools/autotune.py --size 320 --width 480 --batch 1 --dtype bf16 --out $O/tuning_gfx950_bf16.json > $O/r04_autotune_320x480_bf16.txt 2>&1
python tools/autotune.py --size 320 --width 480 --batch 8 --out $O/tuning_gfx950.json > $O/r04_autotune_320x480_b8.txt 2>&1
python tools/autotune.py --size 320 --width 480 --batch 8 --dtype bf16 --out $O/tuning_gfx950_bf16.json > $O/r04_autotune_320x480_b8_bf16.txt 2>&1
python tools/autotune.py --size 512 --batch 1 --out $O/tuning_gfx950.json > $O/r04_autotune_512.txt 2>&1
cp $O/tuning_gfx950.json render-in-between_amd/tuning_gfx950.json
cp $O/tuning_gfx950_bf16.json render-in-between_amd/tuning_gfx950_bf16.json
python bench.py --no-cpu-baseline --height 320 --width 480 > $O/r04_bench_320x480_tuned.json 2>> $O/r04_bench_a.err
python bench.py --no-cpu-baseline --height 320 --width 480 --dtype bf16 > $O/r04_bench_320x480_bf16_tuned.json 2>> $O/r04_bench_a.err
python bench.py --no-cpu-baseline > $O/r04_bench_b.json 2>> $O/r04_bench_a.err
for f in $O/r04_bench_a.json $O/r04_bench_b.json $O/r04_bench_320x480_model.json $O/r04_bench_320x480_tuned.json $O/r04_bench_320x480_bf16_model.json $O/r04_bench_320x480_bf16_tuned.json; do python - $f <<'PY'
import json, sys
j = json.load(open(sys.argv[1])); r = j["roofline"]
print(sys.argv[1].split("/")[-1], "%.1f fps %.4f ms  frac %.4f live %.4f  kernel sum %.4f gaps %.4f" % (j["value"], j["ms_per_step"], r["frac"], r.get("live_frac", 0), r["kernel_time_sum_ms_per_step"], r["launch_gaps_ms_per_step"]))
PY
done
