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
  * file decode / json -> rasteriser tables / PNG encode run in a pool of worker PROCESSES (io_worker.py) that read
    and write the frames in page-locked shared memory: the decoded frames of a chunk go up and its quantised
    frames come back in one asynchronous copy each, and the three phases are pipelined over chunks (SURVEY 8 row
    f-1: at hundreds of frames/s the per-frame .cpu() + PNG encode of evaluator.py:260-266 is the wall - and at PIL's
    default zlib level, the bytes the reference writes, it still is: 54 ms of CPU per 512x512 frame);
  * multi-GPU: the independent units (the segments between key frames, evaluator.py:240-244, over all clips,
    :169-171) are dealt round-robin to the ranks of the process group (distributed.shard_units); every rank
    decodes, renders and writes only its own frames, with no communication after the weight broadcast.
"""
from __future__ import annotations

import collections
import contextlib
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
# A task whose worker process died never completes under multiprocessing.Pool (the pool replaces the worker, not the task):
# every wait on file-side work is bounded, so that a lost task ends the call with an error instead of hanging it.
IO_TIMEOUT_S = 300.0
# Windows of the pipeline, in units (a unit = chunk x B frames): decode runs at most DECODE_AHEAD units ahead of the unit being
# enqueued, and at most MAX_UNITS_IN_FLIGHT units are enqueued but not yet written - host and device memory are bounded by the
# windows, not by the length of the clip.
DECODE_AHEAD = int(os.environ.get("RIB_DECODE_AHEAD", "6"))
MAX_UNITS_IN_FLIGHT = int(os.environ.get("RIB_MAX_UNITS_IN_FLIGHT", "10"))


class _ProcessPool:
    """n worker processes with a concurrent.futures face (submit -> Future).  forkserver: the workers are forked from a clean
    server process that never saw the GPU (a plain fork of a process with an initialised HIP runtime is not safe), with this
    package's file-side module preloaded so that a fork is cheap.  multiprocessing would also re-run the parent's __main__
    script in every worker ("__mp_main__": for a driver script that means importing torch 32 times, which is what made the first
    version of this pool 3x slower than threads); the workers only ever run io_worker functions, so __main__ is hidden
    while they are started - multiprocessing.Pool starts all of them in its constructor."""

    def __init__(self, n):
        import multiprocessing as mp
        import sys
        ctx = mp.get_context("forkserver")
        ctx.set_forkserver_preload(["render_in_between_amd.io_worker", "PIL.Image", "PIL.PngImagePlugin", "scipy.optimize"])
        main = sys.modules.get("__main__")
        saved = {k: getattr(main, k) for k in ("__file__", "__spec__") if hasattr(main, k)}
        # one BLAS / OpenMP thread per worker (io_worker.warm says why); the variables are read when the forkserver imports numpy
        threads_env = {k: os.environ.get(k) for k in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS")}
        try:
            if hasattr(main, "__file__"):
                del main.__file__
            if main is not None:
                main.__spec__ = None
            for k in threads_env:
                os.environ[k] = "1"
            self._pool = ctx.Pool(n, initializer=io_worker.warm)
        finally:
            for k, v in saved.items():
                setattr(main, k, v)
            for k, v in threads_env.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        self.n = n

    def submit(self, fn, *args):
        from concurrent.futures import Future
        fut = Future()
        fut.set_running_or_notify_cancel()
        self._pool.apply_async(fn, args, callback=fut.set_result, error_callback=fut.set_exception)
        return fut

    def shutdown(self):
        self._pool.terminate()


class _ShmBlock:
    """A block of POSIX shared memory that worker processes open by name (io_worker._attach) and that is also page-locked
    for the GPU (hipHostRegister), so that the upload of decoded frames and the download of quantised frames are asynchronous
    copies straight out of / into the memory the workers write / read.  `t`: the block as a flat uint8 tensor."""

    def __init__(self, nbytes):
        from multiprocessing import shared_memory
        self.nbytes = int(nbytes)
        self.shm = shared_memory.SharedMemory(create=True, size=self.nbytes)
        self.name = self.shm.name
        # tmpfs hands out pages on first touch, and a touch beyond what /dev/shm has left is a bus error in whichever process
        # makes it (a decode worker, typically): reserve the pages now, so that exhaustion is an OSError here
        try:
            os.posix_fallocate(self.shm._fd, 0, self.nbytes)
        except (AttributeError, OSError) as e:
            if isinstance(e, OSError):
                self.shm.close(); self.shm.unlink(); self.shm = None
                raise MemoryError("Evaluator: /dev/shm cannot hold a %.1f MB staging block (%s); use io_mode='thread' or a larger /dev/shm"
                                  % (self.nbytes / 1e6, e)) from e
        self.t = torch.from_numpy(np.ndarray((self.nbytes,), np.uint8, buffer=self.shm.buf))
        self.pinned = False
        if torch.cuda.is_available():
            try:
                self.pinned = int(torch.cuda.cudart().cudaHostRegister(self.t.data_ptr(), self.nbytes, 0)) == 0
            except Exception:                                   # noqa: BLE001 (not pinned: the copies are synchronous, still correct)
                self.pinned = False

    def close(self):
        if self.shm is None:
            return
        if self.pinned:
            try:
                torch.cuda.cudart().cudaHostUnregister(self.t.data_ptr())
            except Exception:                                   # noqa: BLE001
                pass
        self.t = None
        try:
            self.shm.close()
            self.shm.unlink()
        except Exception:                                       # noqa: BLE001
            pass
        self.shm = None


_SHM_FREE = {}          # nbytes -> [idle _ShmBlock]: blocks are reused across units and calls (creating + page-locking one costs ms)
_SHM_ALL = []


_SHM_GRAIN = 1 << 20          # block sizes are rounded up to this: ragged groups / tail chunks then share a few size classes
_SHM_KEEP = DECODE_AHEAD + MAX_UNITS_IN_FLIGHT + 4      # idle blocks kept per size class (the pipeline's windows); the rest is released


def _shm_get(nbytes):
    nbytes = (int(nbytes) + _SHM_GRAIN - 1) // _SHM_GRAIN * _SHM_GRAIN
    free = _SHM_FREE.get(nbytes)
    if free:
        return free.pop()
    if not _SHM_ALL:
        import atexit
        atexit.register(_shm_close_all)
    blk = _ShmBlock(nbytes)
    _SHM_ALL.append(blk)
    return blk


def _shm_put(blk):
    """Back to the free list.  Nothing is unmapped here: while a call is running, tensor views of the block may still be
    referenced (a unit's staging dict, the default arguments of a finisher that is still on the stack), and closing the
    mapping under them turns a later access into a segfault instead of an exception.  Surplus blocks are
    released by _shm_trim() at the end of evaluate_from_folder, when no unit is alive."""
    _SHM_FREE.setdefault(blk.nbytes, []).append(blk)


def _shm_trim():
    """Release the idle blocks beyond _SHM_KEEP per size class (unregister, unmap, unlink).  Worker processes drop their own
    attachment of an unlinked block the next time they attach a new one (io_worker._attach)."""
    for free in _SHM_FREE.values():
        while len(free) > _SHM_KEEP:
            blk = free.pop()
            _SHM_ALL.remove(blk)
            blk.close()


def _shm_close_all():
    for blk in _SHM_ALL:
        blk.close()
    _SHM_ALL.clear()
    _SHM_FREE.clear()


def _process_pool(n):
    """One pool of n worker processes per size for the whole process (Evaluators come and go; 32 interpreters should not)."""
    pool = _PROC_POOLS.get(n)
    if pool is None:
        import atexit
        pool = _PROC_POOLS[n] = _ProcessPool(n)
        atexit.register(pool.shutdown)
    return pool


def cpu_budget():
    """CPUs this process can actually use: the affinity mask, capped by the cgroup's CPU quota (a GPU box of the pool shows 256
    CPUs and grants 16 cores' worth of time: more file workers than that only take turns - profiles/r04_encode_probe.txt)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", ):
        try:
            with open(path) as f:
                quota, period = f.read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
    try:                                                        # cgroup v1
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            quota = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            period = int(f.read())
        if quota > 0 and period > 0:
            n = min(n, max(1, quota // period))
    except (OSError, ValueError):
        pass
    return n


def _list(d, exts):
    return [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith(exts)]


class Evaluator:
    def __init__(self, cfg, lanes=2, label_fn=None, png_compress_level=None, resize="cv2", batch=None, chunk=4, io_threads=None,
                 io_mode="process", reproducible=True):
        """batch: independent segments of equal length rendered as ONE chain of that batch size (None: by frame size,
        `default_batch`; 1: every segment on its own, the round-1..3 behaviour).
        reproducible (default True): every chain of the call - full groups, the ragged last group of a clip, a rank's
        smaller share in a multi-GPU run - follows the kernel choices of ONE batch size, the group size
        (Generator.set_plan_batch / rib_set_plan_batch): a frame's bytes do not depend on which segments were rendered
        beside it, so N ranks write what one rank writes ("rank r's frames == single-GPU frames bit-for-bit").
        False: every group size runs its own measured table (a ragged group of 1-3 segments is then a little faster and its
        frames differ from the batched ones by ~1e-5, i.e. at most one uint8 step on ~2e-3 of the pixels).
        chunk: time steps per rib_chain call (the pipeline's unit is chunk x B frames); 0 = whole segments.
        lanes: independent chains kept in flight on one GPU, each on its own HIP stream with
        its own generator handle (the frames inside a segment stay strictly sequential).
        io_threads: decode / encode workers (None: the CPUs this process may run on, divided by the ranks of the job
        sharing the host and by the cgroup's CPU quota (cpu_budget), minus one for the launch thread, at most 48).
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
        self.reproducible = bool(reproducible)
        self.resize = resize
        self.cfg = cfg
        self.lanes = max(1, int(lanes))
        self.label_fn = label_fn
        # PNG compression: None = PIL's default (zlib level 6), which is what the reference's Image.save(name) writes
        # (PGNR/utils/utils.py:139-142) - byte-identical files; a lower level trades file size for encode time
        self.png_compress_level = png_compress_level
        self.batch = None if batch is None else max(1, int(batch))
        self.chunk = max(0, int(chunk))
        ncpu = cpu_budget()
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

    def _shm_fits(self):
        """Is there room in /dev/shm for the pipeline's windows of shared staging blocks?  (A container may mount 64 MB there,
        and a write beyond it is a bus error, not an exception: the file-side work then stays on threads, with a warning.)"""
        unit = max(1, self.chunk) * (self.batch or self.default_batch()) * self.height * self.width * 3
        need = (DECODE_AHEAD + MAX_UNITS_IN_FLIGHT + 4) * unit
        try:
            st = os.statvfs("/dev/shm")
            free = st.f_bavail * st.f_frsize
        except OSError:
            free = 0
        free += sum(b.nbytes for b in _SHM_ALL)             # blocks this process already holds are reused, not allocated again
        if free < need:
            import warnings
            warnings.warn("Evaluator: /dev/shm has %.0f MB free, the shared staging blocks of the worker processes need %.0f MB: "
                          "falling back to io_mode='thread'" % (free / 1e6, need / 1e6))
            return False
        return True

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
        for g, _ in cache[1][1:]:
            if hasattr(g, "set_plan_batch"):
                g.set_plan_batch(model.plan_batch)              # clones made by an earlier call may follow another policy
            if getattr(g, "_graph_replay", False):
                g.set_graph_replay(False)                       # (_plan_policy: no replay in the folder driver)
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

    @contextlib.contextmanager
    def _plan_policy(self, model, native):
        """reproducible: for the length of a call the generator (and, through Generator.clone, its lane clones) follows the
        kernel choices of the group size at every batch; the handle gets its own setting back afterwards.
        Graph replay (RIB_GRAPH=1 / set_graph_replay) is off for the call: every unit of the pipeline brings fresh label / DAIN
        tensors, i.e. other pointers, so each chain would capture and instantiate a graph of its own and none would ever
        replay - the launch-by-launch path is the faster one here; `bench.py --mode clips --graph` is where a
        segment is replayed."""
        if not (native and hasattr(model, "set_plan_batch")):
            yield
            return
        before = model.plan_batch
        graph = bool(getattr(model, "_graph_replay", False))
        model.set_plan_batch((self.batch or self.default_batch()) if self.reproducible else 0)
        if graph:
            model.set_graph_replay(False)
        try:
            yield
        finally:
            model.set_plan_batch(before)
            if graph:
                model.set_graph_replay(True)

    # ---- the driver ------------------------------------------------------------------------------
    @torch.no_grad()
    def evaluate_from_folder(self, model, train_dir, dain_dir, pose_dir, save_dir, gt_dir=None, gen_vid=False,
                             rank=None, world=None):
        """rank / world: this process's share of the independent units (default: the torch.distributed process
        group when one is initialised, else everything).  Returns the frames THIS rank wrote.
        The call is one `_FolderPipeline` (below): plan a clip -> decode -> upload -> render -> sink, pipelined over units
        of `chunk` x B frames; the launch thread only enqueues, decode of later units and encode of finished ones overlap
        the GPU work."""
        if gen_vid:
            # the reference also writes <save_dir>/<clip>.mp4 (evaluator.py:267-269, utils.make_video); not built, and
            # silently ignoring the flag would drop an output the caller asked for
            raise NotImplementedError("evaluate_from_folder: gen_vid is not supported")
        if rank is None or world is None:
            import torch.distributed as dist
            rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
        model.eval()
        if self._pool is None:
            self._pool = ThreadPoolExecutor(self.io_threads)          # decode + encode workers
            self._finishers = ThreadPoolExecutor(max(4, self.lanes + 2))  # wait for a unit's copy, then fan out its encodes
        pipe = _FolderPipeline(self, model, rank, world, gt_dir)
        self.timings = pipe.tm
        try:
            with self._plan_policy(model, pipe.native):
                for sub in [f for f in sorted(os.listdir(pose_dir)) if os.path.isdir(os.path.join(pose_dir, f))]:
                    print("Evaluating {} .....".format(sub))
                    clip = pipe.plan_clip(sub, train_dir, dain_dir, pose_dir, save_dir)
                    if pipe.native:
                        pipe.run_native(clip)
                    else:
                        pipe.run_reference(clip)
                    clip = None
                return pipe.drain()
        finally:
            del pipe                 # the units' tensor views of the shared blocks die with it ...
            _shm_trim()              # ... so the surplus blocks can be unmapped


class _Clip:
    """What this rank renders of one clip, and where every frame's pieces are while the clip is in the pipeline."""

    def __init__(self, names, dain_list, image_list, pose_list, gtlist, sample_rate, keys, segs, units):
        self.names = names                  # output file of every frame index
        self.dain_list, self.image_list, self.pose_list, self.gtlist = dain_list, image_list, pose_list, gtlist
        self.sample_rate = sample_rate
        self.keys = keys                    # this rank's key frames (they pass through, evaluator.py:240-244)
        self.segs = segs                    # this rank's segments [(key, [frame indices])]
        self.units = units                  # native path: [(group, [segment indices], c0, c1)] - a unit is a time chunk of a group
        self.stage = {}                     # unit -> uint8 staging tensor [Tc,B,H,W,3], page-locked; the decoders fill it in place
        self.stage_blk = {}                 # unit -> the shared block behind it (worker processes)
        self.slot = {}                      # frame index -> (unit, t, b): where its DAIN frame goes
        self.loads = {}                     # frame index -> future of (dain | None, key frame | None, pose or rasteriser tables)
        self.futs = {}                      # frame index -> future of its file name, or (unit future, position in the unit)
        self.opened = 0                     # units whose staging block and decode tasks exist
        self.prev_of = {}                   # group -> last fused frames [B,3,H,W] on its lane

    def ref_image(self, i):
        """evaluator.py:209-212: the "gt" image of frame i is gtlist[i] when a gt_dir is given, else the key frame of its
        segment; the keypoints go through A.Resize together with THAT image (:219), i.e. they scale by its size."""
        return self.gtlist[i] if self.gtlist is not None else self.image_list[i // self.sample_rate]

    def unit_frames(self, ui):
        _, members, c0, c1 = self.units[ui]
        return [self.segs[si][1][t] for t in range(c0, c1) for si in members]      # (t, b) order


class _FolderPipeline:
    """One evaluate_from_folder call as explicit stages over units (a unit = `chunk` time steps of a group of up to B segments):

        plan_clip   directory listing -> this rank's key frames, segments, groups and units
        decode      open_units / submit_load: staging block per unit, one decode task per frame (threads or worker processes)
        upload      DAIN frames + label rasterisation + ToTensor/Normalize on the upload stream
        render      the unit's batched chain, the quantiser and ONE device-to-host copy on the lane's stream
        sink        when the copy has landed: PNG encodes, blocks back to the free lists

    Windows: decode runs DECODE_AHEAD units ahead of the unit being enqueued; at most MAX_UNITS_IN_FLIGHT units (of any clip)
    are enqueued but not yet written - a unit holds its label maps and frames on the device and a shared output block on the
    host until its files exist, and the GPU renders faster than PNGs get written.  Every stage can be driven on its own
    (tests/test_driver.py)."""

    def __init__(self, ev, model, rank, world, gt_dir):
        self.ev, self.model, self.rank, self.world, self.gt_dir = ev, model, rank, world, gt_dir
        self.native = hasattr(model, "chain") and hasattr(model, "quantise")
        self.gpu_labels = self.native and ev.label_fn is None and hasattr(model, "rasterise")
        self.procs = _process_pool(ev.io_threads) if (self.native and ev.io_mode == "process" and ev._shm_fits()) else None
        self.pool, self.finishers = ev._pool, ev._finishers
        self.level = ev.png_compress_level
        self.tm = {"load": 0.0, "rasterise": 0.0, "generate": 0.0, "save": 0.0, "frames": 0, "peak_units_in_flight": 0}
        self.t_wall = time.perf_counter()
        self.unit = 0                        # running index of independent units over all clips (the round-robin deal)
        self.inflight = collections.deque()  # sink futures of the units enqueued but not yet written (call-wide window)
        self.clips = []
        self.up = None                       # the upload stream
        self._sizes = {}

    # ---- helpers --------------------------------------------------------------------------------------------------------
    def image_size(self, path):              # header only; the keypoints scale with THIS image
        if path not in self._sizes:
            from PIL import Image
            with Image.open(path) as im:
                self._sizes[path] = im.size
        return self._sizes[path]

    def save_q(self, q, name):               # uint8 HWC -> file, here or in a worker process
        if self.procs is not None:
            return self.procs.submit(io_worker.save_png, q, name, self.level).result(timeout=IO_TIMEOUT_S)
        return io_worker.save_png(q, name, self.level)

    def save_host(self, x, name):            # utils/utils.py:129-142 on the host
        a = np.transpose(x[0].cpu().float().numpy(), (1, 2, 0)) * np.array([0.5] * 3) + np.array([0.5] * 3)
        return self.save_q((np.clip(a, 0, 1) * 255.0).astype(np.uint8), name)

    # ---- stage 0: plan ---------------------------------------------------------------------------------------------------
    def plan_clip(self, sub, train_dir, dain_dir, pose_dir, save_dir):
        ev = self.ev
        frames_dir = os.path.join(save_dir, sub)
        os.makedirs(frames_dir, exist_ok=True)
        image_list = _list(os.path.join(train_dir, sub), ("jpg", "png"))
        dain_list = _list(os.path.join(dain_dir, sub), ("jpg", "png"))
        pose_list = _list(os.path.join(pose_dir, sub), ("json",))
        sample_rate = sample_rate_of(len(pose_list), len(image_list))
        seq_len = (len(image_list) - 1) * sample_rate + 1
        names = [os.path.join(frames_dir, os.path.basename(dain_list[i]))[:-4] + ".png" for i in range(seq_len)]
        gtlist = _list(os.path.join(self.gt_dir, sub), ("jpg", "png")) if self.gt_dir is not None else None
        keys, segs = split_segments(seq_len, sample_rate)
        # this rank's share: a segment is one unit of the deal and brings the key frame it starts from along (that frame is
        # decoded for the chain anyway); a key frame without a segment (the last one) is a unit of its own
        first_of = {k: si for si, (k, _) in enumerate(segs)}
        my_segs, my_keys = [], []
        for k in keys:
            if self.unit % self.world == self.rank:
                my_keys.append(k)
                if k in first_of:
                    my_segs.append(first_of[k])
            self.unit += 1
        segs = [segs[si] for si in my_segs]
        # native path: segments of equal length are grouped into batches, every (group, time chunk) is one unit of the pipeline
        # with one pinned staging buffer that the decode workers fill in place
        units = []
        if self.native:
            B_ = ev.batch or ev.default_batch()
            for gi, members in enumerate(ev.group_segments(segs, B_)):
                T = len(segs[members[0]][1])
                step = ev.chunk if ev.chunk > 0 else T
                for c0 in range(0, T, step):
                    units.append((gi, members, c0, min(T, c0 + step)))
        clip = _Clip(names, dain_list, image_list, pose_list, gtlist, sample_rate, my_keys, segs, units)
        self.clips.append(clip)
        return clip

    # ---- stage 1: decode -------------------------------------------------------------------------------------------------
    def decode_here(self, clip, i):
        """Pre-load of frame i on a pool thread (evaluator.py:205-235)."""
        ev = self.ev
        dain, _ = (ev.load_image_u8 if self.native else ev.load_image)(clip.dain_list[i])
        if i in clip.slot:
            ui, t, b = clip.slot[i]
            clip.stage[ui][t, b].copy_(dain)
            dain = None
        ref_img = clip.ref_image(i)
        gt = ev.load_image(ref_img)[0] if i % clip.sample_rate == 0 else None
        pose = ev.load_pose(clip.pose_list[i], self.image_size(ref_img))
        if self.gpu_labels:                      # host tables of the GPU rasteriser, built here in the worker
            pose = rasterise.frame_tables(pose[0], pose[1], ev.height, ev.width, ev.skeleton_thres, ev.foot_thres)
        return dain, gt, pose

    def decode_in_worker(self, clip, i):
        """The same pre-load in a worker process; its result is unpacked (arrays wrapped as tensors) by the pool's result thread
        as soon as it arrives.  The worker decodes the DAIN frame straight into the unit's shared staging block."""
        from concurrent.futures import Future
        ev = self.ev
        name, off = "", -1
        if i in clip.slot:
            ui, t, b = clip.slot[i]
            name, off = clip.stage_blk[ui].name, (t * clip.stage[ui].shape[1] + b) * ev.height * ev.width * 3
        src = self.procs.submit(io_worker.load_frame_shm, name, off, clip.dain_list[i], clip.ref_image(i), clip.pose_list[i],
                                i % clip.sample_rate == 0, self.gpu_labels, ev.width, ev.height, ev.resize, ev.skeleton_thres, ev.foot_thres)
        out = Future()

        def unpack(f):
            try:
                _, gt, pose = f.result()
                out.set_result((None, torch.from_numpy(io_worker.normalised_chw(gt)) if gt is not None else None, pose))
            except BaseException as e:              # noqa: BLE001 (handed to whoever waits for the frame)
                out.set_exception(e)
        src.add_done_callback(unpack)
        return out

    def submit_load(self, clip, i):
        if i in clip.loads:
            return
        clip.loads[i] = self.decode_in_worker(clip, i) if self.procs is not None else self.pool.submit(self.decode_here, clip, i)
        if i in clip.keys and i not in clip.futs:          # key frames pass through (evaluator.py:240-244)
            clip.futs[i] = self.finishers.submit(
                lambda: self.save_host(clip.loads[i].result(timeout=IO_TIMEOUT_S)[1].unsqueeze(0), clip.names[i]))

    def open_units(self, clip, upto):
        """Staging block, slots and decode tasks of the units up to index `upto`: the launch thread keeps DECODE_AHEAD units open
        beyond the one it is enqueueing, so that host memory (page-locked, shared) does not grow with the clip."""
        ev = self.ev
        while clip.opened <= min(upto, len(clip.units) - 1):
            ui = clip.opened
            _, members, c0, c1 = clip.units[ui]
            shape = (c1 - c0, len(members), ev.height, ev.width, 3)
            if self.procs is not None:      # shared with the decode workers and page-locked (back on the free list once uploaded)
                nb = shape[0] * shape[1] * ev.height * ev.width * 3
                clip.stage_blk[ui] = _shm_get(nb)               # (size classes of 1 MB: the block may be larger than the unit)
                clip.stage[ui] = clip.stage_blk[ui].t[:nb].view(*shape)
            else:
                clip.stage[ui] = torch.empty(shape, dtype=torch.uint8, pin_memory=True)
            for b, si in enumerate(members):
                for t in range(c0, c1):
                    clip.slot[clip.segs[si][1][t]] = (ui, t - c0, b)
            # decode in the order the launch thread will ask for the frames: the unit's key frames first
            for i in ([clip.segs[si][0] for si in members] if c0 == 0 else []) + clip.unit_frames(ui):
                self.submit_load(clip, i)
            clip.opened += 1

    # ---- stage 2: upload -------------------------------------------------------------------------------------------------
    def wait_decoded(self, clip, ui):
        """Blocks until the unit's frames (and, for a first chunk, its key frames) are decoded: (poses or tables, key frames)."""
        _, members, c0, _ = clip.units[ui]
        got = [clip.loads[i].result(timeout=IO_TIMEOUT_S) for i in clip.unit_frames(ui)]
        gt = torch.stack([clip.loads[clip.segs[si][0]].result(timeout=IO_TIMEOUT_S)[1] for si in members]) if c0 == 0 else None
        return [g_[2] for g_ in got], gt

    def upload(self, clip, ui, g, st, poses, gt):
        """Enqueues - on the upload stream, which is idle, so that nothing waits behind the previous unit of the lane - the label
        rasterisation, the upload of the staged DAIN frames with ToTensor + Normalize(0.5, 0.5) on the GPU
        (HSM_auto_dataset.py:73-75) and the key frames; the lane joins through the returned event.  -> (lab, dn, gtd, ready)."""
        ev = self.ev
        _, members, c0, c1 = clip.units[ui]
        Tc, Bc = c1 - c0, len(members)
        if self.up is None:
            self.up = torch.cuda.Stream(device=self.model.device)
        with torch.cuda.stream(self.up):
            if self.gpu_labels:
                lab = rasterise.rasterise_tables(g, poses, ev.height, ev.width, ev.gauss_sigma)
            else:
                lab = ev.make_labels(g if hasattr(g, "rasterise") else self.model, poses)
            lab = lab.to(g.device).reshape(Tc, Bc, *lab.shape[1:])            # [Tc,B,22,H,W]
            dn = clip.stage[ui].to(g.device, non_blocking=True).permute(0, 1, 4, 2, 3).to(torch.float32)
            dn = ((dn / 255.0 - 0.5) / 0.5).contiguous()                      # [Tc,B,3,H,W]
            gtd = gt.to(g.device) if gt is not None else None
            ready = torch.cuda.Event()
            ready.record(self.up)
        for t_ in (lab, dn, gtd):
            if t_ is not None:
                t_.record_stream(st)
        return lab, dn, gtd, ready

    # ---- stage 3: render -------------------------------------------------------------------------------------------------
    def render(self, clip, ui, g, st, lab, dn, gtd, ready):
        """The unit's chain (evaluator.py:240-244,252: a segment starts from its key frame; inside it prev <- fused frame), the
        quantiser and ONE asynchronous copy into page-locked host memory, all on the lane's stream.  Returns what the sink needs."""
        gi, _, c0, _ = clip.units[ui]
        with torch.cuda.stream(st):
            st.wait_event(ready)
            fz = g.chain(gtd if c0 == 0 else clip.prev_of[gi], lab, dn, want_all=False)[2]      # [Tc,B,3,H,W]
            clip.prev_of[gi] = fz[-1]
            q = g.quantise(fz.reshape(-1, *fz.shape[2:]))                  # [Tc*B,H,W,3] uint8
            out_blk = _shm_get(q.numel()) if self.procs is not None else None       # shared with the encode workers, page-locked
            pinned = out_blk.t[:q.numel()].view(q.shape) if out_blk is not None else torch.empty(q.shape, dtype=torch.uint8, pin_memory=True)
            pinned.copy_(q, non_blocking=True)
            done = torch.cuda.Event()
            done.record(st)
        return {"done": done, "pinned": pinned, "out_blk": out_blk, "keep": (fz, q, lab, dn, gtd)}

    # ---- stage 4: sink ---------------------------------------------------------------------------------------------------
    def sink(self, clip, ui, r, t_loaded, t_enq):
        """Runs on a finisher thread: waits for the unit's copy, hands the staging block back, fans the PNG encodes out and
        returns the written names in unit order.  `r["keep"]` keeps the unit's device tensors alive until then."""
        ev = self.ev
        out_frames = clip.unit_frames(ui)
        r["done"].synchronize()                 # the chain, the quantiser and the download are done: so is the upload
        # per unit: inputs decoded, launches enqueued, results on the host, files written (seconds since the call began)
        mark = [round(t_loaded, 4), round(t_enq, 4), round(time.perf_counter() - self.t_wall, 4)]
        self.tm.setdefault("timeline", []).append(mark)
        in_blk = clip.stage_blk.pop(ui, None)
        clip.stage.pop(ui, None)                # (the view of the block goes before the block does)
        if in_blk is not None:
            _shm_put(in_blk)
        out_blk = r["out_blk"]
        if out_blk is not None:
            fsz = ev.height * ev.width * 3
            fs = [self.procs.submit(io_worker.save_png_shm, out_blk.name, j * fsz, ev.height, ev.width, clip.names[out_frames[j]], self.level)
                  for j in range(len(out_frames))]
            res = [f.result(timeout=IO_TIMEOUT_S) for f in fs]
            r["pinned"] = None
            _shm_put(out_blk)
        else:
            qn = r["pinned"].numpy()
            res = list(self.pool.map(lambda j: self.save_q(qn[j], clip.names[out_frames[j]]), range(len(out_frames))))
        r["keep"] = None
        mark.append(round(time.perf_counter() - self.t_wall, 4))
        return res

    # ---- drivers ---------------------------------------------------------------------------------------------------------
    def run_native(self, clip):
        tm = self.tm
        ngroups = len({u[0] for u in clip.units})
        lanes = self.ev._lanes(self.model, ngroups)
        for ui, (gi, members, c0, c1) in enumerate(clip.units):
            self.open_units(clip, ui + DECODE_AHEAD)
            while len(self.inflight) >= MAX_UNITS_IN_FLIGHT:      # back-pressure (call-wide window): wait for the oldest unit's files
                self.inflight.popleft().result(timeout=IO_TIMEOUT_S)
            g, st = lanes[gi % len(lanes)] if lanes else (self.model, torch.cuda.current_stream(self.model.device))
            t0 = time.perf_counter()
            poses, gt = self.wait_decoded(clip, ui)
            t1 = time.perf_counter()
            lab, dn, gtd, ready = self.upload(clip, ui, g, st, poses, gt)
            t2 = time.perf_counter()
            r = self.render(clip, ui, g, st, lab, dn, gtd, ready)
            t3 = time.perf_counter()
            tm["load"] += t1 - t0
            tm["rasterise"] += t2 - t1
            tm["generate"] += t3 - t2
            seg_fut = self.finishers.submit(self.sink, clip, ui, r, t1 - self.t_wall, t3 - self.t_wall)
            self.inflight.append(seg_fut)
            tm["peak_units_in_flight"] = max(tm["peak_units_in_flight"], len(self.inflight))
            for j, i in enumerate(clip.unit_frames(ui)):
                clip.futs[i] = (seg_fut, j)
        for k in clip.keys:
            self.submit_load(clip, k)
        tm["units"] = tm.get("units", 0) + len(clip.units)
        tm["frames"] += len(clip.futs)

    def run_reference(self, clip):
        """Any callable that only speaks the reference's protocol (img, mask = model(label, None, dain, prev)): frame by frame."""
        ev, tm, model = self.ev, self.tm, self.model
        for i in sorted(set(clip.keys) | {i for _, frames in clip.segs for i in frames}):
            self.submit_load(clip, i)
        for k, frames in clip.segs:
            t0 = time.perf_counter()
            got = [clip.loads[i].result() for i in frames]
            gt = clip.loads[k].result()[1].unsqueeze(0)
            t1 = time.perf_counter()
            tm["load"] += t1 - t0
            poses = [g[2] for g in got]
            dn = torch.stack([g[0] for g in got]).unsqueeze(1)
            lab = ev.make_labels(model, poses).unsqueeze(1)
            t2 = time.perf_counter()
            prev, outs = gt, []
            for t in range(len(frames)):
                img, mask = model(lab[t], None, dn[t], prev)
                prev = img * mask.repeat(1, 3, 1, 1) + dn[t].to(img.device) * (1 - mask.repeat(1, 3, 1, 1))
                outs.append(prev)
            tm["rasterise"] += t2 - t1
            tm["generate"] += time.perf_counter() - t2
            for t, i in enumerate(frames):
                clip.futs[i] = self.pool.submit(self.save_host, outs[t], clip.names[i])
        tm["units"] = tm.get("units", 0)
        tm["frames"] += len(clip.futs)

    def drain(self):
        """Waits for every file of the call, in frame order as the reference writes them; returns the names this rank wrote."""
        written: List[str] = []
        t5 = time.perf_counter()
        for clip in self.clips:
            for i, name in enumerate(clip.names):
                if i not in clip.futs:
                    continue                                                       # another rank's frame
                f = clip.futs[i]
                res = f[0].result()[f[1]] if isinstance(f, tuple) else f.result()
                assert res == name
                written.append(name)
        self.tm["save"] = time.perf_counter() - t5                                 # tail: encodes still running after the last enqueue
        self.tm["wall"] = time.perf_counter() - self.t_wall
        return written
