"""TEST INFRASTRUCTURE — not part of the product.

CPU restatements of the reference's algorithms on the scoped path:
  generator_ref   PyTorch fp32 functional ops, NCHW: the pose-guided generator forward and the
                  autoregressive driver step (rows a-1 .. a-14)
  rasterise_ref   numpy / scipy: the label-map rasterisation (row f-2)
  motion_ref      PyTorch fp32 + numpy: stage 1, the motion transformer and its OpenPose json plumbing (row f-4)
  ref_import      loader of the REAL reference generator (build container only; used by the golden scripts)
  precision_model generator_ref with the 16-bit storage modes' roundings applied where the kernels apply them (round 6): what the
                  FORMAT costs, the yardstick the 16-bit modes' error bounds are derived from
Only ``tests/``, ``__graft_entry__.smoke()``, ``bench.py``'s ``cpu_baseline`` leg and the checker / CPU-baseline legs of the
developer scripts under ``tools/`` (measurement scripts, probes, ``products_error.py``, ``verify_checkpoint.py`` - none of them
is imported by the product or shipped as part of it) may import this package, and only as the checker / the CPU side of a
comparison; nothing under ``render-in-between_amd/`` does (tests/test_native_host.py asserts it).

Parity pin: each restatement is validated in the build container against the *imported reference
itself* (tests/golden/make_golden.py, make_golden_raster.py, make_golden_motion.py), whose outputs are
committed as fixtures under tests/golden/.  The reference has no tests or golden vectors of its own
for these paths.
"""
