"""CPU oracle for the gzsl_text hot path.

TEST INFRASTRUCTURE ONLY.  This package is a numpy restatement of the reference's
algorithm (petuum/gzsl_text `modules/*`), written from the reference's behaviour and
pinned against golden vectors captured from the reference itself
(`tests/golden/make_golden.py`, fixtures `tests/golden/*.npz`).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
it.  The product (`gzsl_text_amd/`) never does: it fails loudly without its HIP library.
"""
