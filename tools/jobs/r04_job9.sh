# regenerate the lines of profiles/r04_other_shapes.jsonl that depend on the tables added after the main refresh
set -e -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
S=$O/r04_other_shapes_b.jsonl; : > $S
for flags in "--dtype bf16 --mode chain --frames 32 --batch 4" "--dtype bf16 --mode chain --frames 32 --batch 2" "--dtype f16 --mode chain --frames 32 --batch 4" "--height 320 --width 480 --mode chain --frames 16 --batch 4" "--batch 4 --dtype bf16"; do
  echo "## $flags" >> $S
  python bench.py --no-cpu-baseline --steps 20 --warmup 5 $flags >> $S 2>> $O/r04_other_shapes_b.err
  tail -1 $S | python -c "
import json,sys
j=json.loads(sys.stdin.read()); c=j['config']
print('%-70s %8.1f fps  %.3f ms/frame  %s' % ('$flags', j['value'], j['ms_per_step']/c['frames_per_step_per_gpu'], c['kernel_choices']))"
done
D=$O/r04_driver_b.jsonl; : > $D
for flags in "--size 512 --keys 5 --rate 32 --dtype bf16" "--size 512 --keys 9 --rate 32" "--size 512 --keys 9 --rate 32 --compress 1"; do
  echo "## $flags" >> $D
  timeout -k 10 300 python tools/driver_bench.py $flags >> $D 2>> $O/r04_driver_b.err
  tail -1 $D | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('%-70s %6.1f fps wall %.3f' % ('$flags', j['frames_per_s_end_to_end'], j['wall_s']))"
done
