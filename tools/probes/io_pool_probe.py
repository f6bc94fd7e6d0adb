#!/usr/bin/env python3
"""Why were worker PROCESSES slower than threads for the driver's file-side work (profiles/r04_driver.jsonl)?
Times the pieces of io_worker.load_frame inside the workers (not the round trip) for pools of 1 / 8 / 32 processes, with the
BLAS / OpenMP thread pools of the workers left at their default (one thread per core of the host, per process) and pinned to 1.

    python -m tools.probes.io_pool_probe
"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def timed(what, *a):
    import numpy as np                                   # noqa: F401
    from render_in_between_amd import io_worker, rasterise
    t0 = time.time()
    if what == "decode":
        io_worker.decode_resized_u8(*a)
    elif what == "pose":
        io_worker.scaled_pose(*a)
    elif what == "tables":
        lm, conf = io_worker.scaled_pose(a[0], a[1], a[2], a[3])
        t0 = time.time()
        rasterise.frame_tables(lm, conf, a[3], a[2], 0.001, 0.001)
    elif what == "encode":
        io_worker.save_png(*a)
    elif what == "nthreads":
        try:
            from threadpoolctl import threadpool_info
            return [(d.get("user_api"), d.get("num_threads")) for d in threadpool_info()], os.getpid(), 0.0
        except Exception as e:                           # noqa: BLE001
            return repr(e), os.getpid(), 0.0
    return time.time() - t0, os.getpid(), t0


def main():
    import numpy as np
    from render_in_between_amd import evaluator as ev
    from tools.driver_bench import write_clip
    from tools.probes import io_pool_probe as me          # the workers unpickle functions by module name: not "__main__"
    timed = me.timed
    with tempfile.TemporaryDirectory() as root:
        n = write_clip(root, 3, 16, 512, 512)
        dains = [os.path.join(root, "DAIN", "clip", f) for f in sorted(os.listdir(os.path.join(root, "DAIN", "clip")))]
        poses = [os.path.join(root, "Predict_motion", "clip", f) for f in sorted(os.listdir(os.path.join(root, "Predict_motion", "clip")))]
        u8 = np.asarray(__import__("PIL.Image").Image.open(dains[0]).convert("RGB"))
        print("cpus", len(os.sched_getaffinity(0)), "frames", n)
        for pin in (True,):        # (round 4: _ProcessPool pins its workers' BLAS / OpenMP pools to one thread; before that: profiles/r04_driver.jsonl)
            for nproc in (1, 8, 32, 64):
                pool = ev._ProcessPool(nproc)
                info = pool.submit(timed, "nthreads").result()[0]
                row = []
                for what, argsets in (("decode", [(d, 512, 512, "cv2") for d in dains]),
                                      ("tables", [(p, (512, 512), 512, 512) for p in poses]),
                                      ("encode", [(u8, os.path.join(root, "o%d.png" % i), None) for i in range(n)])):
                    t0 = time.time()
                    res = [f.result() for f in [pool.submit(timed, what, *a) for a in argsets]]
                    wall = time.time() - t0
                    row.append("%s: %.1f ms in-worker, %.1f ms/frame wall" % (what, 1e3 * float(np.mean([r[0] for r in res])), 1e3 * wall / len(argsets)))
                print("threads pinned to 1: %-5s procs %2d  %s   | pools %s" % (pin, nproc, "   ".join(row), info))
                pool.shutdown()


if __name__ == "__main__":
    main()
