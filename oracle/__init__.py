"""TEST INFRASTRUCTURE ONLY - CPU oracle for the HairFastGAN hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker / CPU baseline, never as the thing measured or
shipped.  The product (``hairfastgan_amd``) never imports this package and
fails loudly when its HIP library is missing.

Parity status: the reference ships no tests or golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself, imported on CPU in the build container by ``oracle/make_golden.py``;
the resulting vectors are committed under ``tests/golden/``.
"""
