set -e -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
cp render-in-between_amd/tuning_gfx950.json $O/tuning_base.json
python bench.py --no-cpu-baseline --steps 300 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('before: %.4f ms' % j['ms_per_step'])"
cp $O/tuning_base.json $O/tuning_i60.json
python tools/autotune.py --size 512 --batch 1 --iters 60 --out $O/tuning_i60.json > $O/r04_autotune_512_i60.txt 2>&1
grep "^# " $O/r04_autotune_512_i60.txt
cp $O/tuning_i60.json render-in-between_amd/tuning_gfx950.json
python bench.py --no-cpu-baseline --steps 300 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('iters 60 table: %.4f ms' % j['ms_per_step'])"
cp $O/tuning_base.json render-in-between_amd/tuning_gfx950.json
python bench.py --no-cpu-baseline --steps 300 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('before again: %.4f ms' % j['ms_per_step'])"
cp $O/tuning_i60.json render-in-between_amd/tuning_gfx950.json
python bench.py --no-cpu-baseline --steps 300 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('iters 60 again: %.4f ms' % j['ms_per_step'])"
