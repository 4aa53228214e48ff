"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  **Parity unpinned** (see below).

CPU restatement (pure PyTorch / numpy, fp32 or fp64) of the arithmetic on MatTen's
equivariant message-passing hot path.  The reference (`/root/reference`, wengroup/matten
@ 2025-07-11) is pure Python and delegates every floating point operation on this path to
two un-vendored third-party packages that cannot be installed in the build container:

    e3nn == 0.5.1          (pyproject.toml:29, pretrained/20230627/conda-environment.yaml:210)
    torch_scatter == 2.1.2 (pyproject.toml:28)

so this package restates their *published algorithms* (``oracle/e3nn_lite``) and then restates
the reference's own modules on top of that (``oracle/matten_ref``), anchored on the
reference's call sites (each function cites the reference file:line it follows).

Pinning status: the reference's tests hold **no numeric golden vector** for this path
 -- only property pins (tensor symmetries + rotation equivariance,
tests/model/test_tfn_tensor.py:130-139) and one integer known-answer
(tests/nn/test_embedding.py:7-13); the only stored network output (Si C11/C12/C44,
notebooks/predict_colab.ipynb:324-330) needs the checkpoint that is missing from the
reference tree (.MISSING_LARGE_BLOBS).  The oracle passes all of those pins, but value-level
agreement with a real e3nn install is unverifiable here: **parity unpinned**.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import anything from this package, and only as the checker / the CPU baseline.  The product
(``matten_amd``) never imports it and has no CPU fallback.
"""
