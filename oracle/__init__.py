"""CPU oracle for the RegDA self-training (SSL) step.  TEST INFRASTRUCTURE ONLY.

This package is a restatement, in numpy (integer/label arithmetic) and stock
PyTorch CPU fp32 (floating point), of the reference algorithm on the hot path
`tools/train_ssl_reg.py:198-241` of StuLiu/RegDA.  Every function cites the
reference file:line it follows.

Rules (DESIGN.md "Oracle"):
  * only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
    leg may import anything from here, and only as the checker / baseline;
  * the product (`regda_amd/`) never imports it and has no CPU fallback;
  * it is pinned against golden vectors minted from the reference's own Python
    (imported behind stubs in this container only) by
    `tests/golden/make_goldens.py`; the fixtures live in `tests/golden/*.npz`.
    The reference ships no tests / known-answer vectors of its own.
"""
