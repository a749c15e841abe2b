"""CPU oracle for the adversarial-training hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch (CPU, fp32/fp64) restatement of the reference
algorithms on the hot path.  It exists to *check* the HIP
path; it is never the thing measured or shipped.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import anything from here.  The product package
(``depthmodelhardening_amd``) must never import ``oracle`` -- a test enforces it.

Parity pinning: every function here is checked against golden vectors that were
produced by importing and running the reference source itself
(``oracle/make_goldens.py`` -> ``tests/golden/*.npz``).  The one exception is the
torchvision-0.8.2 arithmetic in ``tv082.py`` (``perspective``, ``Resize``,
``Pad``): torchvision is a pinned third-party dependency of the reference
(requirements.txt:93) that is absent from /root/reference and from this image,
and none of the reference's own files pin its results -- **parity unpinned** for
those three ops (restated from the published 0.8.2 algorithm, property-tested).
"""
