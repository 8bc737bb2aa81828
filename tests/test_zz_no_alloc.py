"""Allocation-free entry point (gzsl_colsum_static) against the existing workspace entry point (gzsl_colsum_ws).

Both run the same two kernels over the same row-block partition (the static scratch is never the limiting size at
these shapes), so wherever the partial sums go through the scratch the results must be IDENTICAL bit for bit -- a
stricter bar than the fp32 tolerance the column sums are otherwise held to (inputs of <= 8 row blocks take the
float-atomic path in both entry points and are compared to 1e-5 relative).  Two calls in a row share the library's per-GPU scratch array: the second, smaller one must not
see anything of the first.  Collected last (test_zz_*): the round-end run uses `pytest -x`.
"""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _both(ops, M, N, ld, seed, exact=True):
    from gzsl_text_amd import hip
    warm = getattr(_both, 'warm', False)
    rs = np.random.RandomState(seed)
    x = torch.from_numpy(rs.standard_normal((M, ld)).astype(np.float32)).to(DEV)[:, :N]
    out0 = torch.from_numpy(rs.standard_normal(N).astype(np.float32)).to(DEV)
    want = ops.colsum(x, out=out0.clone())               # gzsl_colsum_ws (M >= 4096) with a torch-allocated workspace
    torch.cuda.synchronize()
    got = out0.clone()
    n_alloc_pre = torch.cuda.memory_stats().get('allocation.all.allocated', 0)
    lib = hip.load()
    free_pre = torch.cuda.mem_get_info()[0]   # device-wide: would also see a hipMalloc made inside the library
    rc = lib.gzsl_colsum_static(x.data_ptr(), got.data_ptr(), M, N, x.stride(0), hip.stream())  # straight through the C ABI
    assert rc == 0, rc
    torch.cuda.synchronize()
    assert torch.cuda.memory_stats().get('allocation.all.allocated', 0) == n_alloc_pre  # torch's allocator untouched
    if warm:  # (the first call on a device may fault in the library's code object; later calls must not move the number)
        assert torch.cuda.mem_get_info()[0] == free_pre, 'device free memory changed across gzsl_colsum_static'
    _both.warm = True
    ref = out0.double().cpu() + x.double().cpu().sum(0)
    assert float((got.double().cpu() - ref).abs().max()) <= 2e-5 * np.sqrt(M) + 1e-5 * float(ref.abs().max())
    if exact:
        assert torch.equal(got, want), float((got - want).abs().max())
    else:  # <= 8 row blocks: both entry points add their few block sums with float atomics, whose order is not fixed
        assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())
    return got


def test_colsum_static_matches_workspace_entry_point_twice_in_a_row():
    from gzsl_text_amd.ops_hip import HipOps
    ops = HipOps()
    _both(ops, 20000, 200, 200, 1)     # 157 row blocks x 200 columns of partial sums in the scratch array
    _both(ops, 5000, 36, 64, 2)        # smaller, strided (16-byte path again): 40 row blocks; stale partials must not leak
    _both(ops, 6000, 37, 67, 4)        # N and ld not multiples of 4: the scalar-load kernel (colsum_kernel<1>) through the scratch
    _both(ops, 300, 100, 100, 3, exact=False)   # few rows: the atomic path, scratch unused


def test_colsum_static_rejects_sizes_beyond_its_maximum():
    from gzsl_text_amd import hip
    lib = hip.load()
    x = torch.zeros(8, 8200, device=DEV)
    out = torch.zeros(8200, device=DEV)
    assert lib.gzsl_colsum_static(x.data_ptr(), out.data_ptr(), 8, 8200, 8200, hip.stream()) == hip.ETOOBIG
    with pytest.raises(RuntimeError, match='maximum'):
        hip.call('gzsl_colsum_static', x.data_ptr(), out.data_ptr(), 8, 8200, 8200)
    torch.cuda.synchronize()
    assert float(out.abs().sum()) == 0.0
    assert isinstance(ctypes.c_int(hip.ETOOBIG).value, int)
