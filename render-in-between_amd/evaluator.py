"""Sequence driver: the build's restatement of Evaluator.evaluate_from_folder
(PGNR/models/evaluator.py:165-269) on top of the MI355X generator.

Directory contract (PGNR/inference.py:30-35): <input>/inputs/<clip>/*.png are the low-FPS key
frames, <input>/DAIN/<clip>/*.png the interpolated background frames, <input>/Predict_motion/
<clip>/*.json the high-FPS OpenPose joints; frames are written to
<save>/Generated_frames/<clip>/<dain-name>.png.

Differences from the reference, none of which change a pixel:
  * a segment (the frames between two key frames) runs as ONE device-side chain
    (rib_chain): `prev` never leaves HBM and there is no per-frame .cpu() sync
    (evaluator.py:260-262 syncs every frame);
  * the output quantisation runs on the GPU (rib_quantise);
  * the label maps of a whole segment are drawn on the GPU in one call (rib_rasterise) instead of
    per frame with scipy / numpy loops on the host (evaluator.py:221-229);
  * independent segments of equal length are BATCHED: up to `batch` of them run as one chain of batch B
    (rib_chain(T, B, ..): sample b of every step is segment b's frame), which is what fills the 256 CUs on
    the 32x32 / 64x64 maps; a chain is cut into time chunks of `chunk` steps whose first `prev`
    is the previous chunk's last fused frame on the device, so that decode, GPU work and PNG encode of
    different chunks overlap (a unit of the pipeline is `chunk` x B frames);
  * file decode / encode runs on a thread pool, the quantised frames of a chunk come back in
    one pinned device-to-host copy, and the three phases are pipelined over chunks (SURVEY 8 row
    f-1: at hundreds of frames/s the per-frame .cpu() + PNG encode of evaluator.py:260-266 is the wall);
  * multi-GPU: the independent units (the segments between key frames, evaluator.py:240-244, over all clips,
    :169-171) are dealt round-robin to the ranks of the process group (distributed.shard_units); every rank
    decodes, renders and writes only its own frames, with no communication after the weight broadcast.
"""
from __future__ import annotations

import os
import time
from concurrent.futures import ThreadPoolExecutor
from typing import List

import numpy as np
import torch

from . import io_worker, rasterise


def sample_rate_of(num_pose: int, num_key: int) -> int:
    """evaluator.py:190."""
    return 2 ** int(np.log2((num_pose - 1) / (num_key - 1)))


def split_segments(seq_len: int, sample_rate: int):
    """Frame indices: key frames (i % sample_rate == 0) pass through unchanged
    (evaluator.py:240-244); each run of generated frames between them is one independent
    autoregressive segment starting from the preceding key frame."""
    keys = [i for i in range(seq_len) if i % sample_rate == 0]
    segs = []
    for k in keys:
        frames = [i for i in range(k + 1, min(k + sample_rate, seq_len))]
        if frames:
            segs.append((k, frames))
    return keys, segs


_PROC_POOLS = {}


def _process_pool(n):
    """One pool of n worker processes per size for the whole process (Evaluators come and go; 32 interpreters should not).
    forkserver: the workers are forked from a clean server process that never saw the GPU (a plain fork of a process with
    an initialised HIP runtime is not safe), with this package's file-side module preloaded so that a fork is cheap."""
    pool = _PROC_POOLS.get(n)
    if pool is None:
        import atexit
        import multiprocessing as mp
        from concurrent.futures import ProcessPoolExecutor
        ctx = mp.get_context("forkserver")
        ctx.set_forkserver_preload(["render_in_between_amd.io_worker", "PIL.Image", "PIL.PngImagePlugin", "scipy.optimize"])
        pool = _PROC_POOLS[n] = ProcessPoolExecutor(n, mp_context=ctx)
        list(pool.map(_warm, range(n)))          # start them all now, not under the first clip
        atexit.register(pool.shutdown, wait=False, cancel_futures=True)
    return pool


def _warm(_):
    return io_worker.warm()


def _list(d, exts):
    return [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith(exts)]


class Evaluator:
    def __init__(self, cfg, lanes=2, label_fn=None, png_compress_level=None, resize="cv2", batch=None, chunk=8, io_threads=None,
                 io_mode="process"):
        """batch: independent segments of equal length rendered as ONE chain of that batch size (None: by frame size,
        `default_batch`; 1: every segment on its own, the round-1..3 behaviour).  Per-sample arithmetic does not depend
        on the other samples of a batch, but a batch-B plan may pick other tile variants / split-K factors than the
        batch-1 plan: frames agree with batch 1 to ~1e-5, not bit for bit.
        chunk: time steps per rib_chain call (the pipeline's unit is chunk x B frames); 0 = whole segments.
        lanes: independent chains kept in flight on one GPU, each on its own HIP stream with
        its own generator handle (the frames inside a segment stay strictly sequential).
        io_threads: decode / encode workers (None: the CPUs this process may run on, divided by the ranks of the job
        sharing the host, at most 48).
        io_mode: "process" (default) runs the file-side work of the native pipeline - decode, json -> rasteriser tables,
        PNG encode - in a pool of worker processes shared by all Evaluators of this process (io_worker.py), so that it
        cannot hold the interpreter lock of the thread that enqueues the GPU work; "thread": a thread pool (rounds 1-3).
        A model that only speaks the reference's call protocol, or a label_fn, always uses threads.
        label_fn(frames, H, W) -> [T, 22, H, W]: rasteriser override for models that only speak the
        reference's call protocol (the tests pass the CPU oracle); by default the model's GPU
        rasteriser is used and a model without one is an error (no host fallback).
        resize: "cv2" = OpenCV INTER_CUBIC restated (resize.py; what the reference's A.Resize computes), "pil" = PIL BICUBIC."""
        if resize not in ("cv2", "pil"):
            raise ValueError("resize must be 'cv2' or 'pil'")
        if io_mode not in ("process", "thread"):
            raise ValueError("io_mode must be 'process' or 'thread'")
        self.io_mode = io_mode
        self.resize = resize
        self.cfg = cfg
        self.lanes = max(1, int(lanes))
        self.label_fn = label_fn
        # PNG compression: None = PIL's default (zlib level 6), which is what the reference's Image.save(name) writes
        # (PGNR/utils/utils.py:139-142) - byte-identical files; a lower level trades file size for encode time
        self.png_compress_level = png_compress_level
        self.batch = None if batch is None else max(1, int(batch))
        self.chunk = max(0, int(chunk))
        try:
            ncpu = len(os.sched_getaffinity(0))
        except AttributeError:
            ncpu = os.cpu_count() or 1
        # N ranks on one host share its cores: LOCAL_WORLD_SIZE (torchrun) or WORLD_SIZE ranks each take their part, and one
        # core per rank stays with the launch thread (5 k launches per 31-frame segment); more workers than cores only slow it
        ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")) or 1))
        self.io_threads = max(1, int(io_threads)) if io_threads else max(1, min(48, ncpu // ranks - 1))
        self._pool = None                                   # created on first use, kept across calls (thread start-up is ~2 ms each)
        self._finishers = None
        self.timings = {}                                   # seconds per phase of the last evaluate_from_folder
        self.height = cfg.model_height                      # HSM_auto_dataset.py:55-56
        self.width = cfg.model_width
        self.gauss_sigma = getattr(cfg, "gauss_sigma", 5)
        self.skeleton_thres = getattr(cfg, "skeleton_thres", 0.001)
        self.foot_thres = getattr(cfg, "foot_thres", 0.001)

    def default_batch(self):
        """Segments per chain when the caller did not say: enough samples to fill 256 CUs on the deep (1/16-resolution,
        512-channel) layers without growing the working set past the MALL - 8 at the reference's 320x480, 4 at 512x512
        (measured: profiles/r04_driver.jsonl, r04_other_shapes.jsonl)."""
        px = self.height * self.width
        return 8 if px <= 320 * 480 else (4 if px <= 512 * 512 else (2 if px <= 1024 * 1024 else 1))

    @staticmethod
    def group_segments(segs, batch):
        """[(key, frames)] -> [[segment index, ..]]: runs of up to `batch` segments of EQUAL length, in segment order
        (a chain of batch B advances all its samples together, so they must have the same number of steps)."""
        groups, open_by_len = [], {}
        for si, (_, frames) in enumerate(segs):
            g = open_by_len.get(len(frames))
            if g is None:
                g = open_by_len[len(frames)] = []
                groups.append(g)
            g.append(si)
            if len(g) >= batch:
                del open_by_len[len(frames)]
        return groups

    def _lanes(self, model, nsegs):
        """(generator, stream) pairs for concurrent segments; None for single-lane / non-native models."""
        if self.lanes <= 1 or nsegs <= 1 or not hasattr(model, "clone"):
            return None
        ver = getattr(model, "weights_version", 0)
        cache = getattr(self, "_lane_cache", None)
        if cache is None or cache[0] is not model:
            gens = [model] + [model.clone() for _ in range(self.lanes - 1)]
            self._lane_cache = cache = (model, [(g, torch.cuda.Stream(device=model.device)) for g in gens], ver)
        elif cache[2] != ver:
            # the model got new weights (load_state_dict / import_weights) since the lanes were cloned: refresh the
            # clones' blobs, or segments would alternate between old and new weights
            blob = model.export_weights()
            for g, _ in cache[1][1:]:
                g.import_weights(blob)
            torch.cuda.current_stream(model.device).synchronize()
            self._lane_cache = cache = (model, cache[1], ver)
        return cache[1][:max(1, min(self.lanes, nsegs))]

    # ---- per-frame host pre-processing (evaluator.py:205-235) --------------------------------
    def _decode_resized_u8(self, path):
        """PIL decode -> RGB uint8 HWC at the model size.  `resize="cv2"` (default): OpenCV's 8-bit INTER_CUBIC restated
        in resize.py (what the reference's albumentations `A.Resize(interpolation=cv2.INTER_CUBIC)` computes: A = -0.75,
        no low-pass on reduction; unpinned, cv2 is not in this image); `resize="pil"`: PIL's BICUBIC (round 1)."""
        return io_worker.decode_resized_u8(path, self.width, self.height, self.resize)

    def load_image(self, path):
        """PIL open -> resize to the model size (cubic) -> [-1,1] CHW (ToTensor + Normalize(.5,.5));
        PGNR/models/evaluator.py:205-221 with `get_alb_transform` (:18-26)."""
        u8, size0 = self._decode_resized_u8(path)
        a = u8.astype(np.float32) / 255.0
        return torch.from_numpy((a - 0.5) / 0.5).permute(2, 0, 1).contiguous(), size0

    def load_image_u8(self, path):
        """The same decode + resize, left as uint8 HWC: the pipeline uploads a quarter of the bytes and
        applies ToTensor + Normalize(.5,.5) on the GPU (the same two fp32 operations, bit-identical)."""
        u8, size0 = self._decode_resized_u8(path)
        return torch.from_numpy(u8.copy()), size0

    def load_pose(self, json_path, orig_size):
        """json -> (landmarks, conf) in model-size pixels: the keypoints follow the image resize
        (A.Resize keypoint rule, evaluator.py:24-26,219)."""
        return io_worker.scaled_pose(json_path, orig_size, self.width, self.height)

    def make_labels(self, model, frames):
        """[(landmarks, conf)] -> [T, 22, H, W] label maps: 3-ch skeleton image in [-1,1] + 19
        heat-maps in [0,1] (evaluator.py:221-229,250)."""
        if self.label_fn is not None:
            return self.label_fn(frames, self.height, self.width)
        if not hasattr(model, "rasterise"):
            raise RuntimeError("this model has no GPU rasteriser (rib_rasterise); pass Evaluator(label_fn=...)")
        return rasterise.rasterise_labels(model, frames, self.height, self.width, self.gauss_sigma,
                                          self.skeleton_thres, self.foot_thres)

    # ---- the driver ------------------------------------------------------------------------------
    @torch.no_grad()
    def evaluate_from_folder(self, model, train_dir, dain_dir, pose_dir, save_dir, gt_dir=None, gen_vid=False,
                             rank=None, world=None):
        """rank / world: this process's share of the independent units (default: the torch.distributed process
        group when one is initialised, else everything).  Returns the frames THIS rank wrote.
        Pipelined over segments: file decode (thread pool) -> label rasterisation + autoregressive chain +
        quantise on a lane's stream -> one pinned device-to-host copy per segment -> PNG encode (thread pool).
        The main thread only enqueues; decode of later frames and encode of finished segments overlap the
        GPU work (run back to back, the three phases cost about the same: 0.21 / 0.24 / 0.23 s for a
        65-frame 512x512 clip, profiles/r01_raster_driver.json)."""
        from PIL import Image
        if gen_vid:
            # the reference also writes <save_dir>/<clip>.mp4 (evaluator.py:267-269, utils.make_video); not built, and
            # silently ignoring the flag would drop an output the caller asked for
            raise NotImplementedError("evaluate_from_folder: gen_vid is not supported")
        if rank is None or world is None:
            import torch.distributed as dist
            rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
        model.eval()
        written: List[str] = []
        unit = 0                                              # running index of independent units over all clips
        tm = self.timings = {"load": 0.0, "rasterise": 0.0, "generate": 0.0, "save": 0.0, "frames": 0}
        if self._pool is None:
            self._pool = ThreadPoolExecutor(self.io_threads)          # decode + encode workers
            self._finishers = ThreadPoolExecutor(max(4, self.lanes + 2))  # wait for a unit's copy, then fan out its encodes
        pool, finishers = self._pool, self._finishers
        native = hasattr(model, "chain") and hasattr(model, "quantise")
        gpu_labels = native and self.label_fn is None and hasattr(model, "rasterise")
        sizes = {}

        def image_size(path):                                 # header only; the keypoints scale with THIS image (below)
            if path not in sizes:
                with Image.open(path) as im:
                    sizes[path] = im.size
            return sizes[path]

        procs = _process_pool(self.io_threads) if (native and self.io_mode == "process") else None
        level = self.png_compress_level

        def save_q(q, name):                                  # uint8 HWC -> file, here or in a worker process
            if procs is not None:
                return procs.submit(io_worker.save_png, q, name, level).result()
            return io_worker.save_png(q, name, level)

        def save_host(x, name):                               # utils/utils.py:129-142 on the host
            a = np.transpose(x[0].cpu().float().numpy(), (1, 2, 0)) * np.array([0.5] * 3) + np.array([0.5] * 3)
            return save_q((np.clip(a, 0, 1) * 255.0).astype(np.uint8), name)

        t_wall = time.perf_counter()
        up = None
        clip_outputs = []                                     # per clip: (names, {frame index: future})
        for sub in [f for f in sorted(os.listdir(pose_dir)) if os.path.isdir(os.path.join(pose_dir, f))]:
            print("Evaluating {} .....".format(sub))
            frames_dir = os.path.join(save_dir, sub)
            os.makedirs(frames_dir, exist_ok=True)
            image_list = _list(os.path.join(train_dir, sub), ("jpg", "png"))
            dain_list = _list(os.path.join(dain_dir, sub), ("jpg", "png"))
            pose_list = _list(os.path.join(pose_dir, sub), ("json",))
            sample_rate = sample_rate_of(len(pose_list), len(image_list))
            seq_len = (len(image_list) - 1) * sample_rate + 1
            names = [os.path.join(frames_dir, os.path.basename(dain_list[i]))[:-4] + ".png" for i in range(seq_len)]

            gtlist = _list(os.path.join(gt_dir, sub), ("jpg", "png")) if gt_dir is not None else None
            keys, segs = split_segments(seq_len, sample_rate)
            # this rank's share: a segment is one unit and brings the key frame it starts from along (that frame is
            # decoded for the chain anyway); a key frame without a segment (the last one) is a unit of its own
            first_of = {k: si for si, (k, _) in enumerate(segs)}
            my_segs, my_keys = [], []
            for k in keys:
                if unit % world == rank:
                    my_keys.append(k)
                    if k in first_of:
                        my_segs.append(first_of[k])
                unit += 1
            segs = [segs[si] for si in my_segs]
            # native path: segments of equal length are grouped into batches, every (group, time chunk) is one unit of
            # the pipeline with one pinned staging buffer that the decode workers fill in place (no stack on the launch
            # thread, and the upload from pinned memory is asynchronous)
            units, stage, slot = [], {}, {}
            if native:
                B_ = self.batch or self.default_batch()
                for gi, members in enumerate(self.group_segments(segs, B_)):
                    T = len(segs[members[0]][1])
                    step = self.chunk if self.chunk > 0 else T
                    for c0 in range(0, T, step):
                        c1 = min(T, c0 + step)
                        ui = len(units)
                        units.append((gi, members, c0, c1))
                        stage[ui] = torch.empty((c1 - c0, len(members), self.height, self.width, 3), dtype=torch.uint8, pin_memory=True)
                        for b, si in enumerate(members):
                            for t in range(c0, c1):
                                slot[segs[si][1][t]] = (ui, t - c0, b)

            def load(i, dain_list=dain_list, image_list=image_list, pose_list=pose_list, gtlist=gtlist, sample_rate=sample_rate, stage=stage, slot=slot):
                dain, _ = (self.load_image_u8 if native else self.load_image)(dain_list[i])   # pre-load (evaluator.py:205-235)
                if i in slot:
                    ui, t, b = slot[i]
                    stage[ui][t, b].copy_(dain)
                    dain = None
                # evaluator.py:209-212: the "gt" image of frame i is gtlist[i] when a gt_dir is given, else the key frame
                # of its segment; the keypoints go through A.Resize together with THAT image (:219), i.e. they scale
                # by its size, not by the DAIN frame's
                ref_img = gtlist[i] if gtlist is not None else image_list[i // sample_rate]
                gt = self.load_image(ref_img)[0] if i % sample_rate == 0 else None
                pose = self.load_pose(pose_list[i], image_size(ref_img))
                if gpu_labels:                                 # host tables of the GPU rasteriser, built here in the worker
                    pose = rasterise.frame_tables(pose[0], pose[1], self.height, self.width, self.skeleton_thres, self.foot_thres)
                return dain, gt, pose

            def load_in_worker(i, dain_list=dain_list, image_list=image_list, pose_list=pose_list, gtlist=gtlist, sample_rate=sample_rate, stage=stage, slot=slot):
                """The same pre-load in a worker process; its result is unpacked (the DAIN frame copied into its pinned staging
                slot, arrays wrapped as tensors) by the pool's result thread as soon as it arrives."""
                from concurrent.futures import Future
                ref_img = gtlist[i] if gtlist is not None else image_list[i // sample_rate]
                src = procs.submit(io_worker.load_frame, dain_list[i], ref_img, pose_list[i], i % sample_rate == 0, gpu_labels,
                                   self.width, self.height, self.resize, self.skeleton_thres, self.foot_thres)
                out = Future()

                def unpack(f):
                    try:
                        dain, gt, pose = f.result()
                        dain = torch.from_numpy(dain)
                        if i in slot:
                            ui, t, b = slot[i]
                            stage[ui][t, b].copy_(dain)
                            dain = None
                        out.set_result((dain, torch.from_numpy(gt) if gt is not None else None, pose))
                    except BaseException as e:              # noqa: BLE001 (handed to whoever waits for the frame)
                        out.set_exception(e)
                src.add_done_callback(unpack)
                return out
            keys = my_keys
            if native:
                # decode in the order the launch thread will ask for the frames: unit by unit, the unit's key frames first
                order, seen = [], set()
                for gi, members, c0, c1 in units:
                    want = [segs[si][0] for si in members] if c0 == 0 else []
                    want += [segs[si][1][t] for t in range(c0, c1) for si in members]
                    order += [i for i in want if not (i in seen or seen.add(i))]
                order += [k for k in keys if k not in seen]
            else:
                order = sorted(set(my_keys) | {i for _, frames in segs for i in frames})
            loads = {i: (load_in_worker(i) if procs is not None else pool.submit(load, i)) for i in order}      # FIFO
            ngroups = len({u[0] for u in units})
            lanes = self._lanes(model, ngroups) if native else None
            futs = {}
            for k in keys:                                                         # key frames pass through (evaluator.py:240-244)
                futs[k] = finishers.submit(lambda k=k, loads=loads, names=names: save_host(loads[k].result()[1].unsqueeze(0), names[k]))
            prev_of = {}                                                           # group -> last fused frames [B,3,H,W] on its lane
            for ui, (gi, members, c0, c1) in enumerate(units):
                t0 = time.perf_counter()
                Tc, Bc = c1 - c0, len(members)
                got = [loads[segs[si][1][t]].result() for t in range(c0, c1) for si in members]      # (t, b) order
                gt = torch.stack([loads[segs[si][0]].result()[1] for si in members]) if c0 == 0 else None
                t1 = time.perf_counter()
                tm["load"] += t1 - t0
                poses = [g_[2] for g_ in got]
                g, st = lanes[gi % len(lanes)] if lanes else (model, torch.cuda.current_stream(model.device))
                # uploads (pageable host memory: synchronous with respect to their stream) and the label
                # rasterisation go to a stream of their own, which is idle, so that they do not wait behind
                # the previous unit of this lane; the lane joins through an event
                if up is None:
                    up = torch.cuda.Stream(device=model.device)
                with torch.cuda.stream(up):
                    if gpu_labels:
                        lab = rasterise.rasterise_tables(g, poses, self.height, self.width, self.gauss_sigma)
                    else:
                        lab = self.make_labels(g if hasattr(g, "rasterise") else model, poses)
                    lab = lab.to(g.device).reshape(Tc, Bc, *lab.shape[1:])            # [Tc,B,22,H,W]
                    # ToTensor + Normalize(0.5, 0.5) of the uint8 frames on the GPU (HSM_auto_dataset.py:73-75)
                    dn = stage[ui].to(g.device, non_blocking=True).permute(0, 1, 4, 2, 3).to(torch.float32)
                    dn = ((dn / 255.0 - 0.5) / 0.5).contiguous()                      # [Tc,B,3,H,W]
                    gtd = gt.to(g.device) if gt is not None else None
                    ready = torch.cuda.Event()
                    ready.record(up)
                for t_ in (lab, dn, gtd):
                    if t_ is not None:
                        t_.record_stream(st)
                with torch.cuda.stream(st):
                    st.wait_event(ready)
                    t2 = time.perf_counter()
                    # evaluator.py:240-244,252: a segment starts from its key frame; inside it prev <- fused frame
                    fz = g.chain(gtd if c0 == 0 else prev_of[gi], lab, dn, want_all=False)[2]      # [Tc,B,3,H,W]
                    prev_of[gi] = fz[-1]
                    q = g.quantise(fz.reshape(-1, *fz.shape[2:]))                  # [Tc*B,H,W,3] uint8
                    pinned = torch.empty(q.shape, dtype=torch.uint8, pin_memory=True)
                    pinned.copy_(q, non_blocking=True)
                    done = torch.cuda.Event()
                    done.record(st)
                tm["rasterise"] += t2 - t1
                tm["generate"] += time.perf_counter() - t2
                out_frames = [segs[si][1][t] for t in range(c0, c1) for si in members]

                def finish(done=done, pinned=pinned, out_frames=out_frames, names=names, keep=(fz, q, lab, dn, gtd)):
                    done.synchronize()
                    qn = pinned.numpy()
                    if procs is not None:
                        fs = [procs.submit(io_worker.save_png, qn[j], names[out_frames[j]], level) for j in range(len(out_frames))]
                        return [f.result() for f in fs]
                    return list(pool.map(lambda j: save_q(qn[j], names[out_frames[j]]), range(len(out_frames))))
                seg_fut = finishers.submit(finish)
                for j, i in enumerate(out_frames):
                    futs[i] = (seg_fut, j)
            tm["units"] = tm.get("units", 0) + len(units)
            for si, (k, frames) in enumerate([] if native else segs):             # any reference-protocol callable
                t0 = time.perf_counter()
                got = [loads[i].result() for i in frames]
                gt = loads[k].result()[1].unsqueeze(0)
                t1 = time.perf_counter()
                tm["load"] += t1 - t0
                poses = [g[2] for g in got]
                dn = torch.stack([g[0] for g in got]).unsqueeze(1)
                lab = self.make_labels(model, poses).unsqueeze(1)
                t2 = time.perf_counter()
                prev, outs = gt, []
                for t in range(len(frames)):
                    img, mask = model(lab[t], None, dn[t], prev)
                    prev = img * mask.repeat(1, 3, 1, 1) + dn[t].to(img.device) * (1 - mask.repeat(1, 3, 1, 1))
                    outs.append(prev)
                tm["rasterise"] += t2 - t1
                tm["generate"] += time.perf_counter() - t2
                for t, i in enumerate(frames):
                    futs[i] = pool.submit(save_host, outs[t], names[i])
            clip_outputs.append((names, futs))
            tm["frames"] += len(futs)
        t5 = time.perf_counter()
        for names, futs in clip_outputs:                                           # frame order, as the reference writes them
            for i, name in enumerate(names):
                if i not in futs:
                    continue                                                       # another rank's frame
                f = futs[i]
                res = f[0].result()[f[1]] if isinstance(f, tuple) else f.result()
                assert res == name
                written.append(name)
        tm["save"] = time.perf_counter() - t5                                      # tail: encodes still running after the last enqueue
        tm["wall"] = time.perf_counter() - t_wall
        return written
