#!/bin/bash
# Regenerate the measured artefacts of a round on the GPU box (run through gpurun from the repo root):
#   bash tools/refresh_profiles.sh r06
# writes gpurun_out/<tag>_*; copy what should be judged into profiles/.
set -e -o pipefail
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# which build made these files: bench.py quotes the committed rocprofv3 / PMC figures only beside the same build's stamp
python3 -c "import sys; sys.path.insert(0, '$ROOT'); from render_in_between_amd import _native; b = _native.build_info(); print(b['stamp'], b['raw'])" > $OUT/${TAG}_build_stamp.txt
echo "[refresh] bench (default command)"; python3 $ROOT/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo "[refresh] rocprofv3 --stats of the bench command"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --no-other-configs > $OUT/${TAG}_bench_prof.log 2>&1
cp $OUT/prof_bench/bench_kernel_stats.csv $OUT/${TAG}_bench_kernel_stats.csv
echo "[refresh] per-op table"
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_ops -- python3 $ROOT/tools/prof_ops.py --run > $OUT/${TAG}_prof_ops.log 2>&1
python3 $ROOT/tools/prof_ops.py --report $OUT/prof_ops --json $OUT/${TAG}_prof_ops_512.json > $OUT/${TAG}_prof_ops_512.txt
echo "[refresh] per-op table at the reference's default working resolution (320x480) and of the batch-4 frame"
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_ops_320 -- python3 $ROOT/tools/prof_ops.py --run --height 320 --width 480 > $OUT/${TAG}_prof_ops_320.log 2>&1
python3 $ROOT/tools/prof_ops.py --report $OUT/prof_ops_320 --height 320 --width 480 > $OUT/${TAG}_prof_ops_320x480.txt
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_ops_b4 -- python3 $ROOT/tools/prof_ops.py --run --batch 4 > $OUT/${TAG}_prof_ops_b4.log 2>&1
python3 $ROOT/tools/prof_ops.py --report $OUT/prof_ops_b4 --batch 4 > $OUT/${TAG}_prof_ops_512_b4.txt
rm -rf $OUT/prof_ops_320 $OUT/prof_ops_b4
echo "[refresh] per-op tables of the bf16 frame and of the opt-in split-product setting (the roof columns price the 16-bit frame against HBM)"
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_ops_bf16 -- python3 $ROOT/tools/prof_ops.py --run --dtype bf16 > $OUT/${TAG}_prof_ops_bf16.log 2>&1
python3 $ROOT/tools/prof_ops.py --report $OUT/prof_ops_bf16 --dtype bf16 > $OUT/${TAG}_prof_ops_512_bf16.txt
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_ops_x3 -- python3 $ROOT/tools/prof_ops.py --run --products bf16x3 > $OUT/${TAG}_prof_ops_x3.log 2>&1
python3 $ROOT/tools/prof_ops.py --report $OUT/prof_ops_x3 > $OUT/${TAG}_prof_ops_512_x3.txt
rm -rf $OUT/prof_ops_bf16 $OUT/prof_ops_x3
echo "[refresh] PMC passes (separate runs, kernel trace only)"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_f -- python3 $ROOT/tools/prof_ops.py --run > $OUT/${TAG}_pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_w -- python3 $ROOT/tools/prof_ops.py --run > $OUT/${TAG}_pmc_w.log 2>&1
python3 $ROOT/tools/pmc_traffic.py $OUT/pmc_f $OUT/pmc_w $OUT/${TAG}_pmc_traffic.json
echo "[refresh] PMC wave counters (matrix-pipe busy, waits, LDS conflicts): three more passes, one counter set each"
echo "# rocprofv3 --pmc passes over tools/prof_ops.py --run (512x512 B=1), joined with the launch plan by tools/pmc_ops.py" > $OUT/${TAG}_pmc_waves.txt
echo "# SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (us * 2400 cycles) = fraction of the time a SIMD's matrix pipe is busy" >> $OUT/${TAG}_pmc_waves.txt
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC"; do
  name=pmc_$(echo $set | cut -c4-12)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$name -- python3 $ROOT/tools/prof_ops.py --run > $OUT/${TAG}_$name.log 2>&1
  echo "" >> $OUT/${TAG}_pmc_waves.txt; echo "## $name" >> $OUT/${TAG}_pmc_waves.txt
  python3 $ROOT/tools/pmc_ops.py $OUT/$name >> $OUT/${TAG}_pmc_waves.txt
  rm -rf $OUT/$name
done
echo "[refresh] other shapes and modes"
rm -f $OUT/${TAG}_other_shapes.jsonl $OUT/${TAG}_other_shapes.err
for flags in "--height 320 --width 480" "--height 320 --width 480 --no-tuning" "--height 320 --width 480 --dtype bf16" "--height 320 --width 480 --dtype bf16 --no-tuning" "--height 320 --width 480 --mode chain --frames 16 --batch 8" "--height 320 --width 480 --dtype bf16 --mode chain --frames 16 --batch 8" "--mode chain --frames 32" "--mode chain --frames 32 --batch 4" "--mode chain --frames 32 --batch 8" "--mode chain --frames 32 --batch 3 --plan-batch 4" "--mode chain --frames 32 --batch 1 --plan-batch 4" "--height 320 --width 480 --mode chain --frames 16 --batch 3 --plan-batch 8" "--mode clips --frames 32" "--mode clips --frames 32 --graph" "--dtype bf16" "--dtype bf16 --mode chain --frames 32" "--dtype bf16 --mode chain --frames 32 --batch 4" "--dtype f16" "--dtype f16 --mode chain --frames 32" "--inflight 3" "--batch 2" "--batch 4" "--batch 8" "--size 1024 --batch 4" "--size 1024 --batch 4 --dtype bf16" "--size 1024" "--size 256"; do
  echo "## $flags" >> $OUT/${TAG}_other_shapes.jsonl
  python3 $ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 5 $flags >> $OUT/${TAG}_other_shapes.jsonl 2>> $OUT/${TAG}_other_shapes.err
done
echo "[refresh] exact fp32 against the opt-in split products: frame rate A/B in this box, error of both against an fp64 oracle"
rm -f $OUT/${TAG}_x3_frame_ab.txt
for i in 1 2; do for m in f32 bf16x3; do
  python3 $ROOT/bench.py --products $m --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('--products $m', round(d['value'], 1), 'frames/s', round(d['ms_per_step'], 4), 'ms')" >> $OUT/${TAG}_x3_frame_ab.txt
done; done
(cd $ROOT && python3 tools/products_error.py --out $OUT/${TAG}_products_error.json > $OUT/${TAG}_products_error.log 2>&1)
echo "[refresh] the day-one checkpoint check on a seed-defined checkpoint (GPU step included)"
python3 - <<PYEOF
import sys, torch
sys.path.insert(0, "$ROOT")
import render_in_between_amd as rib
from render_in_between_amd import synth
torch.save({"state_dict": synth.make_state_dict(rib.GenSpec.from_cfg(rib.hsm_gen_config()), 0)}, "/tmp/synth_netG.pth")
PYEOF
(cd $ROOT && python3 tools/verify_checkpoint.py /tmp/synth_netG.pth --out /tmp/synth_gold --sizes 64 128 256 --report $OUT/${TAG}_verify_checkpoint_synthetic.json > /dev/null 2>&1) || echo "verify_checkpoint failed"
echo "[refresh] bf16 kernel stats (config 3)"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bf16 -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --dtype bf16 --mode chain --frames 32 --steps 10 --warmup 2 > $OUT/${TAG}_bench_bf16_prof.log 2>&1
cp $OUT/prof_bf16/bench_kernel_stats.csv $OUT/${TAG}_bench_bf16_kernel_stats.csv
echo "[refresh] multi-rank rehearsals on one GPU (the ranks share device 0, gloo instead of RCCL; at most 6 processes may use the card): BASELINE config 4 shape"
cd $ROOT
RIB_BENCH_DEVICE=0 RIB_DIST_BACKEND=gloo python3 bench.py --gpus 2 --mode clips --steps 3 --warmup 1 2> $OUT/${TAG}_clips_2ranks_1gpu.err | grep '^{' > $OUT/${TAG}_clips_2ranks_1gpu.json      # (gloo prints its connection banner on stdout)
# five ranks on the one GPU (their parent process counts against the limit of 6), launch by launch and as graph replays: replicas bit-equal, host-enqueue vs device time per rank
# (the known-good reference log for the first real multi-GPU run; the card is time-shared, so the rates say nothing)
rm -f $OUT/${TAG}_clips_5ranks_1gpu.jsonl
for flags in "" "--graph"; do
  echo "## bench.py --gpus 5 --mode clips --frames 16 $flags" >> $OUT/${TAG}_clips_5ranks_1gpu.jsonl
  RIB_BENCH_DEVICE=0 RIB_DIST_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus 5 --mode clips --frames 16 --steps 2 --warmup 1 $flags 2>> $OUT/${TAG}_clips_5ranks_1gpu.err | grep '^{' | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print(json.dumps({'value': d['value'], 'n_gpus': d['n_gpus'], 'replica_check': c['replica_check'], 'blob_checksum_equal_on_all_ranks': c['blob_checksum_equal_on_all_ranks'],
                  'weight_broadcast_ms': c['weight_broadcast_ms'], 'per_rank_host_enqueue_ms_per_step': c['per_rank_host_enqueue_ms_per_step'],
                  'per_rank_total_ms_per_step': c['per_rank_total_ms_per_step'], 'graph_replay': c['graph_replay'], 'per_rank_device': c['per_rank_device']}))" >> $OUT/${TAG}_clips_5ranks_1gpu.jsonl
done
echo "[refresh] the CLI with its default settings, 1 rank against 2 and 5 ranks sharing the GPU: the same bytes"
rm -f $OUT/${TAG}_multirank_inference.jsonl
for g in 2 5; do
  RIB_BENCH_DEVICE=0 RIB_DIST_BACKEND=gloo timeout -k 10 400 python3 tools/multirank_inference_check.py --gpus $g --keys 9 6 5 >> $OUT/${TAG}_multirank_inference.jsonl 2>> $OUT/${TAG}_multirank_inference.err
done
RIB_BENCH_DEVICE=0 RIB_DIST_BACKEND=gloo timeout -k 10 400 python3 tools/multirank_inference_check.py --gpus 2 --keys 9 6 5 --no-reproducible >> $OUT/${TAG}_multirank_inference.jsonl 2>> $OUT/${TAG}_multirank_inference.err || true
cd /tmp
echo "[refresh] k_warp: timing + FETCH / WRITE counters"
python3 $ROOT/tools/warp_bench.py --time --out $OUT/${TAG}_warp.json > $OUT/${TAG}_warp.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/warp_f -- python3 $ROOT/tools/warp_bench.py --run > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/warp_w -- python3 $ROOT/tools/warp_bench.py --run > /dev/null 2>&1
python3 $ROOT/tools/warp_bench.py --report $OUT/warp_f $OUT/warp_w --out $OUT/${TAG}_warp_pmc.json > $OUT/${TAG}_warp_pmc.log 2>&1
rm -rf $OUT/warp_f $OUT/warp_w
echo "[refresh] folder driver end to end (files in, files out)"
rm -f $OUT/${TAG}_driver.jsonl
for flags in "--size 512 --keys 5 --rate 32" "--size 512 --keys 5 --rate 32 --compress 1" "--height 320 --width 480 --keys 9 --rate 16" "--height 320 --width 480 --keys 9 --rate 16 --compress 1"; do
  echo "## $flags" >> $OUT/${TAG}_driver.jsonl
  timeout -k 10 300 python3 $ROOT/tools/driver_bench.py $flags >> $OUT/${TAG}_driver.jsonl 2>> $OUT/${TAG}_driver.err
done
echo "[refresh] motion transformer"
python3 $ROOT/tools/motion_bench.py --out $OUT/${TAG}_motion_bench.json > $OUT/${TAG}_motion_bench.log 2>&1
rm -rf $OUT/prof_bench $OUT/prof_ops $OUT/pmc_f $OUT/pmc_w $OUT/prof_bf16
echo "[refresh] done"
