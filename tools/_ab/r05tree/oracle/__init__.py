"""CPU oracle for the CleanUMamba hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  The product (``cleanumamba_amd``) never
imports this package and raises if its HIP library is missing.

Parity pinning (see DESIGN.md "Oracle"):

* encoder / decoder / GLU / padding / normalisation / skip logic: pinned against
  the reference's own ``src/network/CleanUMamba.py`` executed verbatim in the
  build container (``oracle/reference_shim.py`` + ``oracle/make_golden.py``),
  outputs committed under ``tests/golden/``.
* Mamba block arithmetic (selective scan, causal depthwise conv, step): lives in
  third-party ``mamba-ssm==1.2.2`` / ``causal-conv1d==1.1.0``
  (reference ``environment.yml:29-30``), absent from ``/root/reference`` and not
  installable here.  ``oracle/mamba_ref.py`` restates the published algorithm
  and is cross-checked against HF ``transformers``' independent pure-torch
  Mamba; against the upstream CUDA kernels themselves it is PARITY UNPINNED
  (the reference holds no golden vector or known-answer test for this path,
  SURVEY.md section 8c).
"""
