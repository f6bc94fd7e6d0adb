set -e -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "kernel_time or graph_replay" 2>&1 | tail -3
cp render-in-between_amd/tuning_gfx950.json $O/tuning_gfx950.json
cp render-in-between_amd/tuning_gfx950_bf16.json $O/tuning_gfx950_bf16.json
python tools/autotune.py --size 512 --batch 4 --dtype bf16 --out $O/tuning_gfx950_bf16.json > $O/r04_autotune_512_b4_bf16.txt 2>&1
python tools/autotune.py --size 512 --batch 2 --dtype bf16 --out $O/tuning_gfx950_bf16.json > $O/r04_autotune_512_b2_bf16.txt 2>&1
python tools/autotune.py --size 320 --width 480 --batch 4 --out $O/tuning_gfx950.json > $O/r04_autotune_320x480_b4.txt 2>&1
grep -h "^# " $O/r04_autotune_512_b4_bf16.txt $O/r04_autotune_512_b2_bf16.txt $O/r04_autotune_320x480_b4.txt
cp $O/tuning_gfx950_bf16.json render-in-between_amd/tuning_gfx950_bf16.json
cp $O/tuning_gfx950.json render-in-between_amd/tuning_gfx950.json
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --dtype bf16 --mode chain --frames 32 --batch 4 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('bf16 chain B=4 tuned: %.1f fps' % j['value'], j['config']['kernel_choices'])"
