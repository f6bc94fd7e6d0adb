#!/usr/bin/env python3
"""Drop-in for PGNR/inference.py: same flags, same config keys, same directory contract.

    python render-in-between_amd/inference.py --input-dir ../example [--config configs/HSM.yaml]
                                               [--save-dir ../example] [--seed 123]

Unlike the reference it builds only what inference needs: the generator (no discriminator, VGG
perceptual loss, optimisers or h5 dataset; PGNR/models/trainer.py:61-113).

Multi-GPU (new; the reference is single-device): `--gpus N` starts N ranks, one per GPU (or launch it under
torch.distributed.run yourself).  Rank 0 reads and folds the checkpoint, the folded blob reaches the other ranks in
ONE RCCL broadcast, and the independent segments between key frames (PGNR/models/evaluator.py:240-244) of all clips
(:169-171) are dealt round-robin to the ranks; every rank writes its own frames into the same output tree.
"""
import argparse
import os
import random
import sys

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))

import render_in_between_amd as rib                                   # noqa: E402
from render_in_between_amd.evaluator import Evaluator                 # noqa: E402


def load_generator(config, device=None, rank=0, world=1, dtype="f32"):
    """trainer.net_G with its checkpoint (PGNR/models/trainer.py:61,67; utils/utils.py:107-119).  With several
    ranks only rank 0 touches the file; the others receive the folded weights (distributed.broadcast_weights)."""
    net_G = rib.Generator(config.gen, device=device, compute_dtype=dtype)
    path = config.model_pretrain_G
    if not os.path.isfile(path):
        raise ValueError("=> No checkpoint found at '{}'".format(path))
    err = None
    if rank == 0:
        try:
            checkpoint = torch.load(path, map_location="cpu")
            print("=> Loaded checkpoint '{}'".format(path))
            net_G.load_state_dict(checkpoint)
        except Exception as e:                  # noqa: BLE001  (re-raised below, on every rank)
            if world == 1:
                raise
            err = e
    if world > 1:
        from render_in_between_amd import distributed as ribdist
        # only rank 0 touched the file: the other ranks learn here whether the broadcast will happen at all
        try:
            ribdist.agree_or_raise(err is None, "rank 0 could not load '{}': {!r}".format(path, err), net_G.device)
        except RuntimeError:
            if err is not None:
                raise err
            raise
        ms = ribdist.broadcast_weights(net_G, src=0)
        if rank == 0:
            print("=> weights broadcast to {} ranks in {:.1f} ms".format(world, ms))
    return net_G


def summary_line(evaluator, rank=0, world=1):
    """One line per rank at the end of a run: what was rendered and where the wall time went.  The phases are the launch
    thread's waits (load: for decoded inputs; rasterise + generate: enqueueing the GPU work; save: the encode tail after the
    last enqueue); the file-side work itself runs in `workers` processes on `cpu_budget` cores beside them."""
    from render_in_between_amd.evaluator import cpu_budget
    tm = evaluator.timings
    wall = max(tm.get("wall", 0.0), 1e-9)
    return ("[rank %d/%d] %d frames in %.2f s = %.1f frames/s | load %.2f s, rasterise %.2f s, generate %.2f s, save tail %.2f s | "
            "%d units, <= %d in flight | %d file workers (%s), CPU budget %d cores | PNG level %s, batch %d, %s plans"
            % (rank, world, tm.get("frames", 0), wall, tm.get("frames", 0) / wall, tm.get("load", 0.0), tm.get("rasterise", 0.0),
               tm.get("generate", 0.0), tm.get("save", 0.0), tm.get("units", 0), tm.get("peak_units_in_flight", 0),
               evaluator.io_threads, evaluator.io_mode, cpu_budget(),
               "reference (zlib 6)" if evaluator.png_compress_level is None else str(evaluator.png_compress_level),
               evaluator.batch or evaluator.default_batch(), "batch-invariant" if evaluator.reproducible else "per-batch"))


def main(opts):
    from render_in_between_amd import distributed as ribdist
    if opts.gpus > 1 and not ribdist.is_rank_process():
        # not a rank yet: start one process per GPU (before anything here touches the GPU) and hand back their exit code
        sys.exit(ribdist.self_launch(os.path.abspath(__file__), sys.argv[1:], opts.gpus))
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    random.seed(opts.seed)
    np.random.seed(opts.seed)
    torch.manual_seed(opts.seed)
    config = rib.get_config(opts.config)
    config.out_dir = opts.save_dir
    config.eval_dir = opts.save_dir
    device = None
    if world > 1:
        device = torch.device("cuda", ribdist.rank_device_index())
        torch.cuda.set_device(device)
        ribdist.init_process_group(os.environ.get("RIB_DIST_BACKEND"), device)
    net_G = load_generator(config, device, rank, world, opts.dtype)
    evaluator = Evaluator(config, batch=opts.batch or None, reproducible=opts.reproducible,
                          png_compress_level=None if opts.png_level == "reference" else int(opts.png_level))
    train_dir = os.path.join(opts.input_dir, "inputs")
    dain_dir = os.path.join(opts.input_dir, "DAIN")
    pose_dir = os.path.join(opts.input_dir, "Predict_motion")
    save_dir = os.path.join(opts.save_dir, "Generated_frames")
    written = evaluator.evaluate_from_folder(net_G, train_dir, dain_dir, pose_dir, save_dir, gt_dir=None, gen_vid=False)
    print(summary_line(evaluator, rank, world))
    if world > 1:
        print("[rank {}/{}] wrote {} frames".format(rank, world, len(written)))
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="pose-guided neural rendering inference (MI355X)")
    parser.add_argument("--config", type=str, default=os.path.join(_HERE, "configs", "HSM.yaml"), help="Path to the config file.")
    parser.add_argument("--save-dir", type=str, default="../example", help="outputs path")
    parser.add_argument("--input-dir", type=str, required=True, help="input low FPS frames and pose input")
    parser.add_argument("--seed", type=int, default=123)
    parser.add_argument("--dtype", choices=("f32", "bf16", "f16"), default="f32",
                        help="f32: the reference's arithmetic (default); bf16 / f16: 16-bit storage, ~2x the frame rate, ~1e-2 / ~1e-3 mean deviation "
                             "(not in the reference: it is fp32 only)")
    parser.add_argument("--gpus", type=int, default=1, help="ranks to start, one per GPU (not in the reference: it is single-device)")
    parser.add_argument("--batch", type=int, default=0,
                        help="independent segments rendered as one chain of that batch size (0: by frame size - 8 at 320x480, 4 at 512x512; "
                             "1: one chain per segment)")
    parser.add_argument("--png-level", default="reference",
                        help="'reference' (default): PIL's default deflate level 6, the very bytes PGNR/utils/utils.py:139-142 writes - 54 ms of "
                             "CPU per 512x512 frame, i.e. the end-to-end rate is bound by the host cores (~200 frames/s on 16); 0-9: that zlib "
                             "level - same pixels, other file bytes (1: ~2x the end-to-end rate, ~25 %% larger files)")
    parser.add_argument("--reproducible", action=argparse.BooleanOptionalAction, default=True,
                        help="(default) every group size follows the kernel choices of the full group, so a frame's bytes do not depend "
                             "on segment grouping, on --gpus N or on --batch 1 vs N-rank shares; --no-reproducible lets ragged groups "
                             "run their own measured tables (frames then agree to ~1e-5, at most one uint8 step)")
    opts = parser.parse_args()
    if opts.png_level != "reference" and opts.png_level not in [str(i) for i in range(10)]:
        parser.error("--png-level must be 'reference' or a zlib level 0-9")
    main(opts)
