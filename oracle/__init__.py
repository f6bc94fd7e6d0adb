"""TEST INFRASTRUCTURE — not part of the product.

CPU restatement (PyTorch fp32 functional ops, NCHW) of the reference's
pose-guided generator forward pass and of the autoregressive driver step.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package, and only as the checker.

Parity pin: validated in the build container against the *imported reference
generator itself* (``oracle/ref_import.py``; see tests/golden/make_golden.py),
whose outputs are committed as fixtures under tests/golden/.  The reference
has no tests or golden vectors of its own for this path.
"""
