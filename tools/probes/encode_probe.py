#!/usr/bin/env python3
"""How fast can the driver's OUTPUT frames be written?  Renders the 129-frame 512x512 clip of tools/driver_bench.py once,
then re-encodes the frames it wrote (PIL's default zlib level 6, the reference's bytes) single-threaded and through worker
pools of several sizes, from shared memory as the pipeline does.

    python -m tools.probes.encode_probe
"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch                                            # noqa: F401
    from PIL import Image
    import render_in_between_amd as rib
    from render_in_between_amd import evaluator as ev, io_worker, synth
    from tools.driver_bench import write_clip
    H = W = 512
    cfg = rib.AttrDict(gen=rib.hsm_gen_config(), model_height=H, model_width=W, gauss_sigma=5, skeleton_thres=0.001, foot_thres=0.001)
    spec = rib.GenSpec.from_cfg(cfg.gen)
    G = rib.Generator(cfg.gen).eval()
    G.load_state_dict(synth.make_state_dict(spec, 0, power_iters=3))
    with tempfile.TemporaryDirectory() as root:
        write_clip(root, 5, 32, H, W)
        E = ev.Evaluator(cfg)
        out = E.evaluate_from_folder(G, *[os.path.join(root, d) for d in ("inputs", "DAIN", "Predict_motion")], os.path.join(root, "o"))
        frames = [np.asarray(Image.open(f)) for f in out]
        sizes = [os.path.getsize(f) for f in out]
        print("frames", len(frames), "mean png bytes", int(np.mean(sizes)))
        t = time.perf_counter()
        for i, a in enumerate(frames[:16]):
            io_worker.save_png(a, os.path.join(root, "s%d.png" % i), None)
        print("single thread: %.1f ms/frame" % ((time.perf_counter() - t) / 16 * 1e3))
        fsz = H * W * 3
        blk = ev._shm_get(len(frames) * fsz)
        for i, a in enumerate(frames):
            blk.t.numpy()[i * fsz:(i + 1) * fsz] = a.reshape(-1)
        for nproc in (16, 32, 48, 64, 96):
            pool = ev._ProcessPool(nproc)
            for rep in range(2):
                t = time.perf_counter()
                fs = [pool.submit(io_worker.save_png_shm, blk.name, i * fsz, H, W, os.path.join(root, "p%d.png" % i), None) for i in range(len(frames))]
                [f.result() for f in fs]
                dt = time.perf_counter() - t
            print("%3d worker processes: %d frames in %.3f s = %.0f frames/s" % (nproc, len(frames), dt, len(frames) / dt))
            pool.shutdown()


if __name__ == "__main__":
    main()
