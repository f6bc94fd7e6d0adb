#!/usr/bin/env python3
"""bench.py — rendered frames/s of the generator forward (+ blend) at 512x512.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode frame|chain|clips]

A "step" is one pass of the hot path over one batch of synthetic input:
  frame  (default) one 512x512 frame (B=1, fp32) through Generator.forward plus the driver's blend,
         device-resident (BASELINE.json configs[1]);
  chain  one autoregressive segment of --frames dependent frames (configs[2] shape);
  clips  every rank renders its OWN --frames-frame clip (seeds 1000*rank + t) per step: BASELINE.json
         configs[3], "8 independent 32-frame clips sharded over 8 GPUs, RCCL weight bcast".

`--gpus N` with N > 1: when this process is not a rank yet (WORLD_SIZE unset) it starts the N ranks itself as child
processes (python -m torch.distributed.run, one per GPU) BEFORE touching the GPU, relays rank 0's JSON line and exits
with their code; the driver's own torch.distributed.run launch takes the other branch.  Rank 0 builds the weights,
they reach the other ranks through ONE RCCL broadcast of the folded blob, and every rank then renders its own
frames with no further communication (weak scaling).

Printed JSON (rank 0): metric/value/unit as the contract asks, plus
  roofline     dominant kernel class (the convolutions): algorithmic FLOPs per forward / device time of EVERY launch
               that takes part in computing them - the matrix-core kernels AND the Winograd transforms and split-K
               slab sums around them - measured with HIP events on the launch stream in a separate profiling
               pass (a start/stop event pair bound to every dispatch: the kernels' own execution times, as
               rocprofv3 --kernel-trace reports them), against the dense fp32 MFMA peak; the executed-FLOP rate
               and the rocprofv3 basis of the same build are reported beside it, and `frac` is the lower of the two
  cpu_baseline the CPU oracle (PyTorch fp32 restatement of the reference) timed on this box's host cores on the
               same workload (rank 0 at N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch                                    # noqa: E402  (importing torch does not initialise the GPU)

PEAK_F32_MFMA_TFLOPS = 157.3                    # MI355X_MICROARCH.md, dense fp32 matrix
PEAK_BF16_MFMA_TFLOPS = 2500.0                  # dense bf16 matrix
PEAK_HBM_GBS = 8000.0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)      # ~1 s of timed device work at 512x512: long enough for an outside GPU-busy sampler to see it
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--size", type=int, default=512, help="square frames: shorthand for --height S --width S")
    ap.add_argument("--height", type=int, default=0, help="frame height (with --width), e.g. the reference's default working "
                                                          "resolution --height 320 --width 480 (PGNR/configs/HSM.yaml:192-193)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--mode", choices=("frame", "chain", "clips"), default="frame",
                    help="frame: one forward+blend per step (BASELINE configs[1], the default); chain: one autoregressive "
                         "segment of --frames dependent frames per step (configs[2] shape); clips: each rank renders its own "
                         "--frames-frame clip per step and the replicas are checked against a 1-GPU run (configs[3])")
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--dtype", choices=("f32", "bf16", "f16"), default="f32",
                    help="f32 (default, the reference's arithmetic), bf16 (BASELINE configs[2]: bf16 storage) or f16 (the same "
                         "16-bit kernels with IEEE half elements); the 16-bit modes are separately reported, never the headline")
    ap.add_argument("--products", choices=("f32", "bf16x3"), default="f32",
                    help="fp32 mode only: f32 (default) = exact-fp32 matrix-core products everywhere, the reference's arithmetic; bf16x3 = "
                         "the plain GEMMs of the frame form each product from six bf16 products of three-way split operands "
                         "(rib_set_products; opt-in, named in config.workload, never the headline)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default run only: skip the short measurements of BASELINE configs[2] (32-frame bf16 chain) and configs[4] "
                         "(1024x1024 batch 4) that are attached to the line as other_configs")
    ap.add_argument("--inflight", type=int, default=1,
                    help="independent frames in flight per GPU: each on its own HIP stream with its own handle and "
                         "workspace (segments between key frames are independent, SURVEY F9); every forward stays batch=B")
    ap.add_argument("--graph", action="store_true", help="chain / clips modes: replay each segment as ONE HIP graph launch (rib_set_graph_replay; "
                                                         "also RIB_GRAPH=1): for hosts where enqueueing ~130 launches per frame per GPU is the limiter")
    ap.add_argument("--plan-batch", type=int, default=0,
                    help="batch-invariant plans (rib_set_plan_batch): every plan follows batch N's kernel choices, as the folder "
                         "driver's default does (N = its group size); 0 = every batch its own table")
    ap.add_argument("--no-tuning", action="store_true", help="ignore the measured tables: every launch takes the analytic cost model's choice")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=8, help="timed CPU-oracle passes of the cpu_baseline leg (>= 5)")
    return ap.parse_args(argv)


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def clip_inputs(spec, synth, r, T, H, W):
    """Clip of rank r (BASELINE configs[3]): frame t is drawn from seed 1000*r + t."""
    frames = [synth.make_inputs(spec, 1, H, W, 1000 * r + t) for t in range(T)]
    labels = torch.stack([f[0] for f in frames])          # [T,1,22,H,W]
    dains = torch.stack([f[1] for f in frames])           # [T,1,3,H,W]
    key = frames[0][2]                                    # the key frame the chain starts from
    return key, labels, dains


def frame_bytes_model(B, H, W, dtype):
    """Fused-minimum HBM bytes of one forward: 2.82 GB per 512x512 fp32 frame, 0.123 GB of it weights."""
    px = B * H * W / (512.0 * 512.0)
    return (2.694e9 * px + 0.123e9) * (0.5 if dtype != "f32" else 1.0)


def measure_other_config(rib, synth, cfg, spec, sd, dev, name, dtype, products, mode, B, H, W, T, steps, warmup):
    """One short measurement of another BASELINE configuration in this process, AFTER the headline's timed region:
    its own Generator (own storage mode, own measured table), `warmup` untimed and `steps` timed steps between device
    synchronisations.  Returns the entry of `other_configs`."""
    G = rib.Generator(cfg, device=dev, compute_dtype=dtype, products=products).eval()
    G.load_state_dict(sd)
    label, fake, prev = [t.to(dev) for t in synth.make_inputs(spec, B, H, W, 0)]
    if mode == "chain":
        labels = label.unsqueeze(0).repeat(T, 1, 1, 1, 1).contiguous()
        dains = fake.unsqueeze(0).repeat(T, 1, 1, 1, 1).contiguous()
        step = lambda: G.chain(prev, labels, dains, want_all=False)[2]
        frames = B * T
    else:
        step = lambda: G.forward_blend(label, None, fake, prev)[2]
        frames = B
    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    flops = sum(G.forward_flops(B, H, W).values()) * (frames / B)
    byts = frame_bytes_model(B, H, W, dtype) * (frames / B)
    t_mfma = flops / ((PEAK_F32_MFMA_TFLOPS if dtype == "f32" else PEAK_BF16_MFMA_TFLOPS) * 1e12)
    t_hbm = byts / (PEAK_HBM_GBS * 1e9)
    entry = {"name": name, "workload": None, "dtype": dtype, "products": products, "value": frames / dt, "unit": "frames/s",
             "ms_per_step": dt * 1e3, "frames_per_step": frames, "steps": steps, "warmup": warmup,
             "bound": "mfma" if t_mfma >= t_hbm else "hbm", "frac": max(t_mfma, t_hbm) / dt,
             "frac_basis": "SURVEY 8(d) roof of the whole step: max(algorithmic FLOPs / dense MFMA peak of the dtype, fused-minimum bytes / 8 TB/s) / measured step time",
             "launches_per_forward": G.num_launches(B, H, W),
             "kernel_choices": "measured table" if G.tuned_ops(B, H, W) else "analytic cost model (no table for this shape)"}
    del G, label, fake, prev
    torch.cuda.empty_cache()
    return entry


def main():
    args = parse_args()
    from render_in_between_amd import distributed as ribdist
    if args.gpus > 1 and not ribdist.is_rank_process():
        # not a rank yet: start one fresh process per GPU (nothing in THIS process has touched the GPU) and hand
        # back their exit code; rank 0 prints the JSON line on the shared stdout
        sys.exit(ribdist.self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus))

    t_start = time.perf_counter()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import render_in_between_amd as rib
    from render_in_between_amd import synth

    # RIB_BENCH_DEVICE / RIB_DIST_BACKEND exist only to rehearse the multi-rank path on a 1-GPU box
    # (all ranks on one device, gloo instead of RCCL); the driver's multi-GPU runs use neither.
    dev_index = ribdist.rank_device_index()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        ribdist.init_process_group(os.environ.get("RIB_DIST_BACKEND"), dev)

    def log(msg):
        if rank == 0:
            print("[bench %.1fs] %s" % (time.perf_counter() - t_start, msg), file=sys.stderr, flush=True)

    cfg = rib.hsm_gen_config()
    spec = rib.GenSpec.from_cfg(cfg)
    H, W = (args.height or args.size), (args.width or args.size)
    B = args.batch
    if args.products != "f32" and args.dtype != "f32":
        raise SystemExit("--products bf16x3 is an option of the fp32 mode")
    G = rib.Generator(cfg, device=dev, compute_dtype=args.dtype, products=args.products, use_tuning=not args.no_tuning).eval()
    G.set_plan_batch(args.plan_batch)
    sd = None
    t_bcast_ms = 0.0
    if rank == 0:
        sd = synth.make_state_dict(spec, 0)
        G.load_state_dict(sd)
    if world > 1:
        t_bcast_ms = ribdist.broadcast_weights(G, src=0)
    blob_sum = ribdist.blob_checksum(G.export_weights())

    graph_on = args.graph or bool(int(os.environ.get("RIB_GRAPH", "0") or 0))
    if graph_on:
        G.set_graph_replay(True)
        torch.cuda.synchronize(dev)
        torch.cuda.set_stream(torch.cuda.Stream(device=dev))      # the NULL stream cannot be captured
    # extra in-flight lanes: clones of the generator (same folded weight blob) on their own streams
    lanes = [(G, torch.cuda.current_stream(dev))]
    if args.inflight > 1:
        blob = G.export_weights()
        torch.cuda.synchronize(dev)
        for _ in range(args.inflight - 1):
            lanes.append((rib.Generator(cfg, device=dev, compute_dtype=args.dtype, products=args.products, use_tuning=not args.no_tuning).eval().set_plan_batch(args.plan_batch).import_weights(blob), torch.cuda.Stream(device=dev)))
        torch.cuda.synchronize(dev)

    frames_per_step = B
    if args.mode == "clips":
        if B != 1:
            raise SystemExit("--mode clips renders batch-1 clips")
        T = args.frames
        frames_per_step = T
        key, labels, dains = [t.to(dev) for t in clip_inputs(spec, synth, rank, T, H, W)]

        def step(i=0):
            g, st = lanes[i % len(lanes)]
            with torch.cuda.stream(st):
                return g.chain(key, labels, dains, want_all=False)[2]
    else:
        # per-rank synthetic inputs (rank r renders its own frames)
        label, fake, prev = [t.to(dev) for t in synth.make_inputs(spec, B, H, W, 1000 * rank)]
        if args.mode == "chain":
            T = args.frames
            frames_per_step = B * T
            labels = label.unsqueeze(0).repeat(T, 1, 1, 1, 1).contiguous()
            dains = fake.unsqueeze(0).repeat(T, 1, 1, 1, 1).contiguous()

            def step(i=0):
                g, st = lanes[i % len(lanes)]
                with torch.cuda.stream(st):
                    return g.chain(prev, labels, dains, want_all=False)[2]
        else:
            def step(i=0):
                g, st = lanes[i % len(lanes)]
                with torch.cuda.stream(st):
                    return g.forward_blend(label, None, fake, prev)[2]     # forward + blend (the mask head writes the fused frame)

    log("weights ready, warming up")
    for i in range(max(args.warmup, len(lanes))):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(i)
    dt_enqueue = time.perf_counter() - t0        # the host is done enqueueing; the device may still be working
    torch.cuda.synchronize()
    dt_rank = time.perf_counter() - t0           # this rank's own time (per-rank frames/s below)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    fps = world * frames_per_step * args.steps / dt

    # ---- control-plane exchange (after the timed region): per-rank rates, blob checksums, replica check ----
    per_rank_fps = ribdist.gather_rows(torch.tensor([frames_per_step * args.steps / dt_rank], dtype=torch.float64, device=dev), rank, world).flatten().tolist()
    # host-bound or device-bound?  per rank: time the launch thread spent enqueueing the timed steps vs the time until the
    # device finished them.  enqueue << total: the host runs ahead (device-bound); enqueue ~ total: the host is the limiter
    # (8 ranks on one host enqueue ~8 x 130 launches per frame: --graph / RIB_GRAPH=1 turns a segment into one launch)
    enq = ribdist.gather_rows(torch.tensor([dt_enqueue / args.steps * 1e3, dt_rank / args.steps * 1e3], dtype=torch.float64, device=dev), rank, world).tolist()
    sums = ribdist.gather_rows(torch.tensor([blob_sum], dtype=torch.int64, device=dev), rank, world).flatten().tolist()
    # which physical GPU each rank ran on (index, name, PCI bus id): 8 ranks must show 8 different devices
    ident = ribdist.device_identity(dev_index)
    idents = [None] * world
    if world > 1:
        torch.distributed.all_gather_object(idents, "rank %d local_rank %s %s" % (rank, os.environ.get("LOCAL_RANK", "0"), ident))
    else:
        idents = ["rank 0 local_rank 0 " + ident]
    replica = None
    if args.mode == "clips":
        last = ribdist.gather_rows(out[-1].contiguous(), rank, world)      # every rank's last fused frame, [world,1,3,H,W]
        if rank == 0:
            # rank 0 (whose weights came from load_state_dict, not from the broadcast) renders every rank's clip
            # itself, as a 1-GPU run would, and compares the last frames bit for bit
            eq = []
            for r in range(world):
                if r == 0:
                    mine = out[-1]
                else:
                    k_r, l_r, d_r = [t.to(dev) for t in clip_inputs(spec, synth, r, args.frames, H, W)]
                    mine = G.chain(k_r, l_r, d_r, want_all=False)[2][-1]
                eq.append(bool(torch.equal(mine, last[r])))
            replica = {"clips_checked": world, "last_frame_bit_equal_to_single_gpu_run": eq, "all_equal": all(eq)}
            log("replica check: %s" % eq)

    if rank != 0:
        if world > 1:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return

    log("timed region done: %d steps, %.3f ms/step" % (args.steps, ms_per_step))
    # ---- roofline of the dominant kernel class: profiling pass (not in the timed region) ----
    # Every launch of the pass carries a (start, stop) HIP event pair bound to the dispatch itself (hipExtLaunchKernelGGL,
    # rib_profile_begin_kernels): the difference is the kernel's own execution time on the launch stream - the duration
    # rocprofv3 --kernel-trace reports for it - with no event packet between two launches.  (Rounds 1-3 put one event in
    # FRONT of every launch and modelled the event cost away; that correction was only valid for one lane and one rank,
    # ADVICE r03.)  The classes therefore do NOT add up to the timed step: the difference is the dependent-launch gaps.
    flops = G.forward_flops(B, H, W)
    G.profile_begin(kernels=True)
    nprof = 20 if args.mode == "frame" else 1      # (5 steps gave the class time a run-to-run spread of +-3 %)
    log("profile pass (separate from the timed region, after it): %d step(s) with an event pair on every dispatch" % nprof)
    for _ in range(nprof):
        step()
    prof = G.profile_collect()
    if args.mode != "frame":
        nprof = nprof * args.frames          # per-forward averages
    # The convolution class = the matrix-core launches (k_igemm incl. the Winograd-domain batched GEMMs, k_conv_lowc,
    # k_conv_head) PLUS the launches that are part of computing those same convolutions: Winograd input / output
    # transforms and split-K slab sums (class "conv_aux").
    n_launch = sum(v["launches"] for v in prof.values()) / nprof
    kernel_sum_ms = sum(v["ms"] for v in prof.values()) / nprof
    step_ms_one = ms_per_step / (frames_per_step / B) if args.mode != "frame" else ms_per_step     # per forward
    mm_ms = prof["igemm"]["ms"] / nprof
    aux_ms = prof["conv_aux"]["ms"] / nprof
    conv_ms = mm_ms + aux_ms
    conv_launches = (prof["igemm"]["launches"] + prof["conv_aux"]["launches"]) / nprof
    conv_tflops = flops["igemm"] / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    spade_ms = prof["spade"]["ms"] / nprof
    classes = {k: {"launches_per_step": v["launches"] / nprof, "kernel_ms_per_step": v["ms"] / nprof} for k, v in prof.items()}
    # executed matrix work: the three mask-network upsample convolutions run as 2x2 phase convolutions, 4/9 of their
    # nine-tap count, the Winograd-domain GEMMs execute 4/9 (F(2x2)) or 1/4 (F(4x4)) of theirs; everything
    # else executes what it is priced at (channel padding not counted)
    executed = 0.0
    for info in G.launch_info(B, H, W):
        if info["class"] == 0:
            tile = info["tile"]
            executed += info["flops"] * (0.25 if "wino4" in tile else 4.0 / 9.0 if ("ups1" in tile or "wino" in tile) else 1.0)
    # rocprofv3 basis of the same class and HBM bytes per launch from the committed passes (tools/prof_ops.py,
    # tools/pmc_traffic.py: FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate rocprofv3 --pmc passes); bench.py itself
    # cannot read counters or kernel traces, so these are pointers to committed measurements of the same workload,
    # NOT measured in this run.  They are only quoted when the committed files were made by THIS build (the stamp
    # rib_build_info() reports is stored beside them by tools/refresh_profiles.sh).
    from render_in_between_amd import _native
    build = _native.build_info()
    traffic = traffic_step = traffic_src = None
    rocprof_basis = None
    default_workload = (B, H, W) == (1, 512, 512) and args.dtype == "f32" and args.mode == "frame"
    prof_stamp = None
    # the newest round's committed passes (profiles/rNN_build_stamp.txt is written by tools/refresh_profiles.sh rNN)
    tags = sorted(f[:3] for f in os.listdir(os.path.join(ROOT, "profiles")) if f[3:] == "_build_stamp.txt" and f[0] == "r")
    tag = tags[-1] if tags else "r00"
    sp = os.path.join(ROOT, "profiles", tag + "_build_stamp.txt")
    if os.path.exists(sp):
        with open(sp) as f:
            prof_stamp = f.read().split()[0]
    same_build = prof_stamp == build["stamp"]
    tp = os.path.join(ROOT, "profiles", tag + "_pmc_traffic.json")
    if os.path.exists(tp) and default_workload:
        with open(tp) as f:
            tj = json.load(f)
        cb = sum(tj["classes"].get(k, {}).get("hbm_bytes_corrected", 0.0) for k in ("igemm", "conv_aux"))
        cl = sum(tj["classes"].get(k, {}).get("launches", 0) for k in ("igemm", "conv_aux"))
        traffic = cb / max(1, cl)
        traffic_step = tj["total_hbm_bytes_per_step"]
        traffic_src = ("profiles/" + tag + "_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/prof_ops.py; not measured in "
                       "this run; made by build %s, this run is build %s)" % (prof_stamp, build["stamp"]))
    rp = os.path.join(ROOT, "profiles", tag + "_prof_ops_512.json")
    if os.path.exists(rp) and default_workload:
        with open(rp) as f:
            rj = json.load(f)
        us = sum(o["us"] for o in rj["ops"] if o["class"] in (0, 6))
        nl = sum(1 for o in rj["ops"] if o["class"] in (0, 6))
        if us > 0:
            rocprof_basis = {"source": "profiles/" + tag + "_prof_ops_512.json (rocprofv3 --kernel-trace of tools/prof_ops.py --run; not measured in this run)",
                             "made_by_build": prof_stamp, "same_build_as_this_run": same_build,
                             "class_us_per_step": us, "launches_per_step": nl, "avg_launch_us": us / nl,
                             "achieved_tflops": flops["igemm"] / (us * 1e-6) / 1e12, "frac": flops["igemm"] / (us * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS}
    peak = {"f32": PEAK_F32_MFMA_TFLOPS, "bf16": PEAK_BF16_MFMA_TFLOPS, "f16": PEAK_BF16_MFMA_TFLOPS}[args.dtype]
    # fused-minimum HBM model of SURVEY 8(d) (every conv reads its input and writes its output once, one extra read per
    # normalised tensor, one cond read per SPADE layer, weights once): 2.82 GB per 512x512 fp32 frame, of which 0.123 GB
    # are weights; activations scale with the pixel count, bf16 storage halves everything
    alg_bytes = frame_bytes_model(B, H, W, args.dtype)
    hbm_gbs = alg_bytes * (frames_per_step / B) / (ms_per_step * 1e-3) / 1e9
    # headline fraction: the live kernel-time figure, unless the committed rocprofv3 trace of this very build says less
    live_frac = conv_tflops / peak
    frac, frac_basis = live_frac, "live: HIP event pairs bound to each dispatch (hipExtLaunchKernelGGL) on the launch stream"
    if rocprof_basis and same_build and rocprof_basis["frac"] < live_frac:
        frac, frac_basis = rocprof_basis["frac"], "rocprofv3 --kernel-trace of the same build (lower than the live figure %.4f)" % live_frac
    roofline = {
        "bound": "mfma",
        # "algorithmic" = nine-tap 2*MAC count of SURVEY 8(d) over the convolutions of the class
        "kernel": "convolution class: k_igemm (%s MFMA implicit GEMM, incl. the Winograd-domain batched GEMMs), k_conv_lowc (first layers), "
                  "k_conv_head (2 heads) = %d matrix-core launches/step, PLUS their %d Winograd-transform / split-K-sum launches/step"
                  % (args.dtype, int(prof["igemm"]["launches"] / nprof), int(prof["conv_aux"]["launches"] / nprof)),
        "achieved": frac * peak, "peak": peak, "unit": "TFLOP/s",
        "frac": frac, "frac_basis": frac_basis,
        "live_achieved": conv_tflops, "live_frac": live_frac,
        "avg_launch_us": conv_ms * 1e3 / max(1.0, conv_launches),
        "class_kernel_ms_per_step": conv_ms, "matrix_core_launches_ms_per_step": mm_ms, "transform_and_splitk_sum_launches_ms_per_step": aux_ms,
        "algorithmic_gflop_per_step": flops["igemm"] / 1e9,
        "executed_gflop_per_step": executed / 1e9,
        "executed_tflops": executed / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0,
        "frac_executed": (executed / (conv_ms * 1e-3) / 1e12 / peak) if conv_ms > 0 else 0.0,
        "rocprof_basis": rocprof_basis,
        "traffic": traffic,
        "traffic_bytes_per_step_all_classes": traffic_step,
        "traffic_source": traffic_src,
        "classes": classes,
        "kernel_time_sum_ms_per_step": kernel_sum_ms, "timed_step_ms": step_ms_one,
        "launch_gaps_ms_per_step": step_ms_one - kernel_sum_ms, "launches_per_step": n_launch,
        "spade_tflops": flops["spade"] / (spade_ms * 1e-3) / 1e12 if spade_ms > 0 else 0.0,
        "whole_step_frac_of_mfma_roof": (sum(flops.values()) * (frames_per_step / B) / (PEAK_F32_MFMA_TFLOPS * 1e12)) / (ms_per_step * 1e-3),
        "whole_step_algorithmic_hbm_gbs": hbm_gbs, "whole_step_frac_of_hbm_roof": hbm_gbs / PEAK_HBM_GBS,
    }
    if args.dtype != "f32":
        # with 16-bit matrix cores (2.5 PFLOP/s) the frame's roof is HBM (0.18 ms at 512x512): report against that;
        # the matrix-core figures of the convolution class stay in the object for reference
        roofline.update({"bound": "hbm", "mfma_achieved_tflops": conv_tflops, "mfma_frac": conv_tflops / peak,
                         "kernel": "whole frame (the bf16 frame is bound by HBM, not by any one kernel): fused-minimum bytes of SURVEY 8(d) / frame time",
                         "achieved": hbm_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": hbm_gbs / PEAK_HBM_GBS,
                         "frac_basis": "fused-minimum bytes / timed step", "algorithmic_bytes_per_frame": alg_bytes})

    log("profile pass done: conv %.3f ms/step" % conv_ms)
    # ---- parity on the bench inputs + CPU baseline (oracle timed on the host cores) ----
    cpu = None
    parity = None
    if not args.no_cpu_baseline and args.mode == "frame" and world == 1:   # rank 0 at N=1 only
        from oracle import generator_ref
        R = generator_ref.RefGenerator(spec, sd)
        # the oracle is timed "with os.cpu_count() threads and again with 8 threads".  The GPU boxes show 256 CPUs
        # but grant a cgroup quota of 16: the first setting is therefore the CPU budget (more threads than the quota only
        # add contention; tools/cpu_threads.py sweep 8/16/32/64/128: 16 is the fastest), the second is 8 threads, the
        # setting BASELINE.md's survey probe used (1.173 s/frame on 8 vCPU).  `value` = the faster of the two.
        from render_in_between_amd.evaluator import cpu_budget
        budget = max(1, min(cpu_budget(), os.cpu_count() or 1))
        lc, fc, pc = label.cpu(), fake.cpu(), prev.cpu()
        reps = max(5, args.cpu_frames)
        legs = []
        for threads in sorted({budget, min(8, budget)}, reverse=True):
            torch.set_num_threads(threads)
            oimg, omask = R(lc, None, fc, pc)                       # warm-up + parity reference
            log("cpu oracle warm-up frame done (%d threads)" % torch.get_num_threads())
            ts = []
            for _ in range(reps):
                t1 = time.perf_counter()
                oi, om = R(lc, None, fc, pc)
                generator_ref.blend(oi, om, fc)
                ts.append(time.perf_counter() - t1)
            med = sorted(ts)[len(ts) // 2]
            legs.append({"threads": torch.get_num_threads(), "value": B / med, "unit": "frames/s", "reps": reps,
                         "seconds_per_frame": {"median": med, "min": min(ts), "max": max(ts)}})
        img, mask = G(label, None, fake, prev)
        parity = {"max_abs_img": float((img.cpu() - oimg).abs().max()),
                  "max_abs_mask": float((mask.cpu() - omask).abs().max()),
                  "tolerance": 1e-3 if args.dtype == "f32" else None}
        best = max(legs, key=lambda l: l["value"])
        cpu = {"value": best["value"], "unit": "frames/s", "cores": best["threads"], "kind": "port",
               "threads": best["threads"], "host_cores": os.cpu_count(), "cpu_budget": budget, "cpu_model": cpu_model_name(), "reps": reps,
               "seconds_per_frame": best["seconds_per_frame"], "settings": legs,
               "sample": "%d forward+blend passes (after 1 warm-up) per thread setting of the same %dx%d B=%d fp32 workload through the CPU oracle "
                         "(PyTorch restatement validated against the imported reference), median; settings = the box's CPU budget "
                         "(cgroup quota) and 8 threads (SURVEY 8d); `value` / `cores` = the faster setting"
                         % (reps, H, W, B)}

    # ---- the other single-GPU BASELINE configurations, measured briefly in the same process: the
    # 32-frame bf16 chain (configs[2]) and one 1024x1024 batch-4 fp32 forward (configs[4]); plus the opt-in split-product
    # setting on the headline workload, named as such.  Only beside the default command's line; never part of `value`.
    others = None
    if default_workload and world == 1 and args.products == "f32" and not args.no_other_configs and not args.no_tuning and args.inflight == 1:
        others = []
        if sd is None:
            sd = synth.make_state_dict(spec, 0)
        for (name, dtype, products, mode, oB, oS, oT, osteps, owarm) in (
                ("BASELINE configs[2]", "bf16", "f32", "chain", 1, 512, 32, 4, 1),
                ("BASELINE configs[4]", "f32", "f32", "frame", 4, 1024, 1, 5, 2),
                ("BASELINE configs[1] with the opt-in split-bf16 products on the plain GEMMs (NOT the reference's arithmetic)", "f32", "bf16x3", "frame", 1, 512, 1, 100, 10)):
            try:
                e = measure_other_config(rib, synth, cfg, spec, sd, dev, name, dtype, products, mode, oB, oS, oS, oT, osteps, owarm)
            except Exception as ex:      # an extra must never take the headline line down with it
                e = {"name": name, "error": "%s: %s" % (type(ex).__name__, ex)}
            else:
                dn = {"f32": "fp32", "bf16": "bf16 storage", "f16": "half storage"}[dtype] + (", split-bf16 (bf16x3) products on the k_gemm_dma launches" if products != "f32" else "")
                e["workload"] = ("%dx%d autoregressive %d-frame segment, batch=%d, %s" % (oS, oS, oT, oB, dn) if mode == "chain"
                                 else "%dx%d single-frame generator fwd + blend, batch=%d, %s" % (oS, oS, oB, dn))
            others.append(e)
            log("other config %s: %s" % (name, {k: e.get(k) for k in ("value", "ms_per_step", "frac", "error") if k in e}))

    dt_name = {"f32": "fp32", "bf16": "bf16 storage", "f16": "half storage"}[args.dtype]
    if args.products != "f32":
        dt_name += ", split-bf16 (bf16x3) products on the k_gemm_dma launches (opt-in: NOT the reference's exact-fp32 arithmetic)"
    if args.mode == "frame":
        workload = "%dx%d single-frame generator fwd + blend, batch=%d, %s" % (H, W, B, dt_name)
    elif args.mode == "chain":
        workload = "%dx%d autoregressive %d-frame segment (prev <- fused frame on device), batch=%d, %s" % (H, W, args.frames, B, dt_name)
    else:
        workload = "%dx%d, %d independent %d-frame clips sharded over %d GPU(s) (one clip per rank and step, seeds 1000*rank+t), %s" % (
            H, W, world, args.frames, world, dt_name)
    line = {
        "metric": "rendered frames/sec at %dx%d (generator forward + blend, device-resident)" % (H, W),
        "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": workload + ", seed-defined random-init HSM.yaml generator (spectral-norm vectors power-iterated)",
                   "frames_per_step_per_gpu": frames_per_step,
                   "parallelism": "%s sharded over %d GPU(s), one RCCL weight broadcast, no per-frame collective" % ("clips" if args.mode == "clips" else "frames", world),
                   "weight_broadcast_ms": t_bcast_ms, "launches_per_step": G.num_launches(B, H, W),
                   "frames_in_flight_per_gpu": len(lanes),
                   "per_rank_frames_per_s": per_rank_fps,
                   "per_rank_host_enqueue_ms_per_step": [e[0] for e in enq], "per_rank_total_ms_per_step": [e[1] for e in enq],
                   "graph_replay": (G.graph_stats() if graph_on else None), "per_rank_device": idents,
                   "distinct_devices": len({i.split(" pci ")[-1] for i in idents}),
                   "blob_checksum": sums[0], "blob_checksum_equal_on_all_ranks": len(set(sums)) == 1,
                   "replica_check": replica,
                   "build": build["raw"], "height": H, "width": W, "plan_batch": args.plan_batch, "products": args.products,
                   "kernel_choices": "analytic cost model" if args.no_tuning else ("measured table" if G.tuned_ops(B, H, W) else "analytic cost model (no table for this shape)")},
        "roofline": roofline, "cpu_baseline": cpu, "parity": parity, "other_configs": others,
    }
    print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
