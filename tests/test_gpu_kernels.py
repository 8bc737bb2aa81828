"""GPU parity tests: every HIP kernel (through the C ABI, via gzsl_text_amd.ops_hip.HipOps) against the plain
torch-fp32 statement of the same op (tests/ops_cpu.py) on identical seeded inputs.

Tolerances are fp32 (north star: "within a stated fp32 tolerance"): forward values rtol 1e-5 / atol 1e-5 scaled to the
data, reductions over K terms atol ~ K * 1e-7 * typical magnitude; stated per check.
"""
import numpy as np
import pytest
import torch

from tests.helpers import close
from tests.ops_cpu import CpuOps

pytestmark = pytest.mark.gpu

DEV = 'cuda'


@pytest.fixture(scope='module')
def ops():
    from gzsl_text_amd.ops_hip import HipOps
    return HipOps()


@pytest.fixture(scope='module')
def ref():
    return CpuOps()


def rnd(rs, *shape, scale=1.0):
    return torch.from_numpy((rs.standard_normal(shape) * scale).astype(np.float32))


def g(t):
    return None if t is None else t.to(DEV)


@pytest.mark.parametrize('M,N,K', [(1, 1, 1), (37, 53, 19), (128, 128, 32), (300, 200, 1000), (129, 100, 200),
                                   (1000, 300, 70), (64, 800, 800)])
@pytest.mark.parametrize('ta,tb', [(False, False), (True, False), (False, True), (True, True)])
def test_gemm_all_layouts(ops, ref, M, N, K, ta, tb):
    rs = np.random.RandomState(M * 7 + N * 3 + K + ta * 2 + tb)
    A = rnd(rs, K, M).t() if ta else rnd(rs, M, K)
    B = rnd(rs, N, K).t() if tb else rnd(rs, K, N)
    bias = rnd(rs, N)
    Ad, Bd = (g(A.t().contiguous()).t() if ta else g(A)), (g(B.t().contiguous()).t() if tb else g(B))
    for act in (0, 1, 2, 3):
        out = ops.mm(Ad, Bd, bias=g(bias), act=act)
        close(out, ref.mm(A, B, bias=bias, act=act), 2e-5, 2e-5 * np.sqrt(K), f'mm act={act}')
    C0 = rnd(rs, M, N)
    out = ops.mm(Ad, Bd, out=g(C0).clone(), accumulate=True, alpha=0.5)
    close(out, ref.mm(A, B, out=C0.clone(), accumulate=True, alpha=0.5), 2e-5, 2e-5 * np.sqrt(K), 'mm accumulate')


@pytest.mark.parametrize('M', [1, 8, 16, 17, 32])
@pytest.mark.parametrize('N,K', [(800, 800), (400, 800), (1, 64), (100, 333), (4096, 200), (37, 32)])
def test_gemm_skinny_rows(ops, ref, M, N, K):
    """M <= 32 takes the skinny kernel (gemm.hip): every operand layout the WGAN-GP step feeds it (x @ W^T, x @ W,
    column blocks of W with an offset base, strided activations), bias / activation / accumulate / alpha."""
    rs = np.random.RandomState(M * 131 + N + K)
    x, W, Wt, bias = rnd(rs, M, K), rnd(rs, N, K, scale=0.1), rnd(rs, K, N, scale=0.1), rnd(rs, N)
    for act in (0, 1, 2, 3):
        close(ops.mm(g(x), g(W).t(), bias=g(bias), act=act), ref.mm(x, W.t(), bias=bias, act=act), 2e-5, 2e-5 * np.sqrt(K),
              f'x @ W^T act={act}')
    close(ops.mm(g(x), g(Wt)), x @ Wt, 2e-5, 2e-5 * np.sqrt(K), 'x @ W (row-contiguous B)')
    C0 = rnd(rs, M, N)
    close(ops.mm(g(x), g(W).t(), out=g(C0).clone(), accumulate=True, alpha=0.5), C0 + 0.5 * (x @ W.t()), 2e-5,
          2e-5 * np.sqrt(K), 'accumulate + alpha')
    if K >= 64:  # two K segments of one weight (torch.cat folded into two GEMMs, gan_engine.py) and a strided A
        k1 = (K // 2) // 4 * 4 + (0 if K % 4 == 0 else 1)
        out = ops.mm(g(x)[:, :k1], g(W)[:, :k1].t(), bias=g(bias))
        ops.mm(g(x)[:, k1:], g(W)[:, k1:].t(), out=out, accumulate=True, act=3)
        want = torch.nn.functional.leaky_relu(x @ W.t() + bias, 0.2)
        close(out, want, 2e-5, 2e-5 * np.sqrt(K), 'two K segments with LeakyReLU')


@pytest.mark.parametrize('M,N,K', [(800, 400, 24), (800, 800, 8), (64, 64, 1), (132, 68, 64), (800, 400, 63)])
def test_gemm_short_k_outer_products(ops, ref, M, N, K):
    """K <= 64 with both operands row-contiguous (weight gradients dh^T @ x of the WGAN-GP step at small batches) takes
    the short-K kernel: plain, alpha + accumulate, and a column block of a wider matrix as the output."""
    rs = np.random.RandomState(M + N + K)
    A, B = rnd(rs, K, M), rnd(rs, K, N)
    close(ops.mm(g(A).t(), g(B)), A.t() @ B, 2e-5, 2e-5 * np.sqrt(K), 'A^T B')
    big = rnd(rs, M, N + 100)
    bigd = g(big).clone()
    ops.mm(g(A).t(), g(B), out=bigd[:, 100:], accumulate=True, alpha=0.5)
    big[:, 100:] += 0.5 * (A.t() @ B)
    close(bigd, big, 2e-5, 2e-5 * np.sqrt(K), 'accumulate into a column block')


@pytest.mark.parametrize('M,N,K', [(8, 200, 4096), (16, 400, 8192), (3, 37, 5000)])
def test_gemm_skinny_split_k(ops, ref, M, N, K):
    """Few rows, few columns, long K (keyword head: dlogits [Bg,Vk] @ word_emb [Vk,E]): the skinny kernel with K split
    over blockIdx.y and the ordered slab reduce."""
    rs = np.random.RandomState(M + N + K)
    x, W, bias = rnd(rs, M, K), rnd(rs, K, N, scale=0.1), rnd(rs, N)
    assert ops._split_policy(M, N, K)[0] > 1
    close(ops.mm(g(x), g(W), bias=g(bias), act=3), torch.nn.functional.leaky_relu(x @ W + bias, 0.2), 2e-5,
          2e-5 * np.sqrt(K), 'x @ W split-K')
    C0 = rnd(rs, M, N)
    close(ops.mm(g(x), g(W.t().contiguous()).t(), out=g(C0).clone(), accumulate=True), C0 + x @ W, 2e-5, 2e-5 * np.sqrt(K),
          'x @ W^T split-K accumulate')


def test_gemm_split_k_and_strided_output(ops, ref):
    rs = np.random.RandomState(5)
    M, N, K = 100, 200, 20000
    A, B = rnd(rs, K, M), rnd(rs, K, N)
    out = ops.mm(g(A).t(), g(B))  # weight-gradient shape: small output, huge K -> split-K slabs
    close(out, A.t() @ B, 2e-5, 3e-3, 'split-K TN')
    big = torch.zeros(M, N + 50)
    bigd = g(big).clone()
    ops.mm(g(A).t(), g(B), out=bigd[:, :N], accumulate=True)
    big[:, :N] += A.t() @ B
    close(bigd, big, 2e-5, 3e-3, 'split-K into a column slice (ldc > N)')


@pytest.mark.parametrize('B,T,E,F,k', [(3, 40, 8, 24, 10), (2, 33, 8, 20, 10), (2, 300, 100, 200, 10), (1, 75, 100, 200, 10),
                                       (4, 64, 12, 40, 3)])
def test_conv_fwd_bwd(ops, ref, B, T, E, F, k):
    rs = np.random.RandomState(B + T + E)
    e, W, b = rnd(rs, B, T, E), rnd(rs, F, E, k, scale=0.1), rnd(rs, F, scale=0.1)
    X = ops.conv_relu(g(e), g(W), g(b))
    Xr = ref.conv_relu(e, W, b)
    close(X, Xr, 2e-5, 2e-5, 'conv+relu')
    dY = rnd(rs, B, T - k + 1, F) * (Xr > 0).float()
    de, dW, db = ops.conv_bwd(g(dY), g(e), g(W))
    der, dWr, dbr = ref.conv_bwd(dY, e, W)
    close(de, der, 2e-5, 5e-5, 'conv data grad')
    close(dW, dWr, 2e-5, 1e-4 * np.sqrt(B * T / 100), 'conv weight grad')
    close(db, dbr, 2e-5, 1e-4 * np.sqrt(B * T / 100), 'conv bias grad')


@pytest.mark.parametrize('p', [0.0, 0.5])
def test_embedding_gather_scatter(ops, ref, p):
    rs = np.random.RandomState(3)
    V, E, B, T = 50, 100, 4, 37
    W = rnd(rs, V, E)
    ids = torch.from_numpy(rs.randint(0, V, size=(B, T)).astype(np.int64))
    mask = torch.from_numpy((rs.random_sample((B, T)) < 0.8).astype(np.float32))
    out = ops.emb_gather(g(W), g(ids), g(mask), 77, p, row_offset=11)
    close(out, ref.emb_gather(W, ids, mask, 77, p, row_offset=11), 0, 0, 'emb gather (bit exact)')
    dOut = rnd(rs, B, T, E)
    dW = ops.emb_scatter_add_(torch.zeros(V, E, device=DEV), g(ids), g(mask), g(dOut), 77, p, row_offset=11)
    close(dW, ref.emb_scatter_add_(torch.zeros(V, E), ids, mask, dOut, 77, p, row_offset=11), 1e-5, 1e-5, 'emb scatter')
    assert float(dW[0].abs().sum()) == 0.0, 'padding row must get no gradient'


def test_embedding_scatter_grouped_by_id(ops, ref):
    """>= 4096 rows take the grouped path (histogram / scan / fill / segmented sums); heavy repeats as in word ids."""
    rs = np.random.RandomState(11)
    V, E, B, T = 700, 100, 4, 1500
    ids = np.minimum((rs.zipf(1.3, size=(B, T)) - 1), V - 1).astype(np.int64)  # id 0 (padding) is the most frequent
    ids[0, :50] = V - 1
    ids = torch.from_numpy(ids)
    mask = torch.from_numpy((rs.random_sample((B, T)) < 0.9).astype(np.float32))
    dOut = rnd(rs, B, T, E)
    for p in (0.0, 0.5):
        dW = ops.emb_scatter_add_(torch.zeros(V, E, device=DEV), g(ids), g(mask), g(dOut), 5, p, row_offset=3)
        dWr = ref.emb_scatter_add_(torch.zeros(V, E), ids, mask, dOut, 5, p, row_offset=3)
        close(dW, dWr, 1e-5, 2e-4, f'grouped emb scatter p={p}')  # sums of up to ~2000 terms in a different order
        assert float(dW[0].abs().sum()) == 0.0, 'padding row must get no gradient'


def test_label_avg_bwd_repeated_words(ops, ref):
    """Description words repeat heavily across labels (many adds into the same embedding rows)."""
    rs = np.random.RandomState(12)
    V, E, N, Td = 300, 100, 500, 12
    lens = rs.randint(1, Td + 1, size=N)
    cm = torch.from_numpy((np.arange(Td)[None] < lens[:, None]).astype(np.float32))
    idx = torch.from_numpy(np.minimum(rs.zipf(1.4, size=(N, Td)), V - 1).astype(np.int64)) * cm.long()
    davg = rnd(rs, N, E)
    dW = ops.label_avg_bwd_(torch.zeros(V, E, device=DEV), g(davg), g(idx), g(cm))
    want = ref.label_avg_bwd_(torch.zeros(V, E), davg, idx, cm)
    close(dW, want, 1e-5, 1e-4, 'label avg bwd (repeated words)')
    assert float(dW[0].abs().sum()) == 0.0, 'padding row must get no gradient'
    # the same adjoint as a gather over the inverted description index (what the engine uses), added on top of a
    # pre-loaded table with float atomics
    from gzsl_text_amd.engine import GraphBuffers
    gb = GraphBuffers(g(idx), g(cm), torch.eye(N).to(DEV).to_sparse(), torch.ones(N, device=DEV))
    pre = rnd(rs, V, E)
    dW2 = ops.label_avg_bwd_(g(pre).clone(), g(davg), g(idx), g(cm), inv=gb.desc_inverse(V))
    close(dW2, pre + want, 1e-5, 1e-4, 'label avg bwd through the inverted index')
    assert torch.equal(dW2[0].cpu(), pre[0]), 'padding row must get no gradient'


@pytest.mark.parametrize('M,N,ld', [(20000, 200, 200), (5000, 37, 64), (300, 100, 100), (9000, 100, 300)])
def test_colsum_workspace_and_atomic_paths(ops, M, N, ld):
    rs = np.random.RandomState(M + N)
    x = rnd(rs, M, ld)
    xd = g(x)[:, :N]
    out0 = rnd(rs, N)
    out = ops.colsum(xd, out=g(out0).clone())  # accumulates into out
    close(out, out0 + x[:, :N].double().sum(0).float(), 1e-5, 2e-5 * np.sqrt(M), 'colsum')


def test_label_side_kernels(ops, ref):
    rs = np.random.RandomState(4)
    V, E, N, Td, H = 60, 100, 45, 7, 100
    W = rnd(rs, V, E)
    lens = rs.randint(1, Td + 1, size=N)
    cm = torch.from_numpy((np.arange(Td)[None] < lens[:, None]).astype(np.float32))
    idx = torch.from_numpy(rs.randint(1, V, size=(N, Td)).astype(np.int64)) * cm.long()
    avg = ops.label_avg(g(W), g(idx), g(cm))
    close(avg, ref.label_avg(W, idx, cm), 1e-6, 1e-6, 'label avg')
    davg = rnd(rs, N, E)
    dW = ops.label_avg_bwd_(torch.zeros(V, E, device=DEV), g(davg), g(idx), g(cm))
    close(dW, ref.label_avg_bwd_(torch.zeros(V, E), davg, idx, cm), 1e-5, 1e-5, 'label avg bwd')
    from gzsl_text_amd.engine import build_csr
    from tests.golden import synth
    rows, cols, vals, deg = synth.forest_adjacency(rs, N)
    adj = torch.from_numpy(synth.coo_to_dense(rows, cols, vals, N))
    csr, csr_t = build_csr(adj.to_sparse())
    csr_d, csr_td = build_csr(adj.to(DEV).to_sparse())
    inv = torch.from_numpy(1.0 / (deg + 1e-7))
    x = rnd(rs, N, H)
    close(ops.spmm(csr_d, g(x), post=g(inv)), ref.spmm(csr, x, post=inv), 1e-6, 1e-6, 'spmm mean')
    acc = rnd(rs, N, H)
    close(ops.spmm(csr_td, g(x), pre=g(inv), out=g(acc).clone(), accumulate=True),
          ref.spmm(csr_t, x, pre=inv, out=acc.clone(), accumulate=True), 1e-6, 1e-6, 'spmm adjoint accumulate')
    gi, gh, h, dhn = rnd(rs, N, 3 * H), rnd(rs, N, 3 * H), rnd(rs, N, H), rnd(rs, N, H)
    close(ops.gru_gate_fwd(g(gi), g(gh), g(h)), ref.gru_gate_fwd(gi, gh, h), 1e-5, 1e-6, 'gru gate fwd')
    for a, b, nm in zip(ops.gru_gate_bwd(g(dhn), g(gi), g(gh), g(h)), ref.gru_gate_bwd(dhn, gi, gh, h),
                        ('dgi', 'dgh', 'dh')):
        close(a, b, 1e-5, 1e-6, 'gru gate bwd ' + nm)


ATTN_SHAPES = [
    # B, Tp, E, F, L
    (2, 31, 8, 24, 12),     # golden tiny shape
    (1, 5, 8, 20, 3),       # fewer tokens than one tile, fewer labels than one MFMA tile
    (2, 291, 100, 200, 256),  # golden medium shape: Tp not a multiple of 32
    (3, 100, 100, 200, 77),   # L not a multiple of 64 or 16
    (8, 64, 100, 200, 130),   # B multiple of 8 -> XCD-aware block mapping
    (2, 150, 200, 200, 40),   # real E = 200 class
    (2, 70, 50, 100, 33),     # sizes strictly inside a shape class (zero-padded operands)
]


@pytest.mark.parametrize('B,Tp,E,F,L', ATTN_SHAPES + [(2, 2491, 100, 200, 140), (2, 2491, 200, 200, 70), (1, 3991, 100, 200, 140),
                                                   (1, 3991, 200, 200, 70)])
@pytest.mark.parametrize('p', [0.0, 0.2])
def test_label_attention_fwd_bf16x6_experimental(ref, B, Tp, E, F, L, p):
    """The opt-in forward (GZSL_ATTN_BF16X6=1: fp32 products as six bf16 MFMAs, csrc/attn_bf16.hip) at the SAME
    tolerances as the fp32-MFMA kernel, incl. one full-length note."""
    from gzsl_text_amd.ops_hip import HipOps
    ops = HipOps()
    ops.attn_bf16x6 = True
    rs = np.random.RandomState(B * 1000 + Tp + L)
    PX = torch.tanh(rnd(rs, B, Tp, E))
    X = torch.relu(rnd(rs, B, Tp, F))
    Q = rnd(rs, L, E, scale=0.3)
    l_off, L_tot, seed = 5, L + 9, 4321
    O, lse = ops.attn_fwd(g(PX), g(X), g(Q), l_off, L_tot, seed, p)
    Or, lser = ref.attn_fwd(PX, X, Q, l_off, L_tot, seed, p)
    close(lse, lser, 1e-5, 1e-5, 'attention log-sum-exp (bf16x6)')
    close(O, Or, 2e-5, 2e-5, 'attention output (bf16x6)')


@pytest.mark.parametrize('kind', ['peaky', 'wide_range', 'tiny_and_huge'])
def test_label_attention_fwd_bf16x6_error_against_fp64(kind):
    """The six-product form must be as close to an fp64 evaluation as the exact-fp32 MFMA kernel is, on inputs chosen to
    hurt a reduced-precision product: scores up to +-60 (softmax weight on a handful of tokens), values spanning nine
    decades, operands with very small and very large magnitudes side by side.  Dropout off (the mask is exact)."""
    from gzsl_text_amd.ops_hip import HipOps
    rs = np.random.RandomState({'peaky': 1, 'wide_range': 2, 'tiny_and_huge': 3}[kind])
    B, Tp, E, F, L = 2, 333, 100, 200, 130
    PX = torch.tanh(rnd(rs, B, Tp, E))
    X = torch.relu(rnd(rs, B, Tp, F))
    Q = rnd(rs, L, E, scale=0.3)
    if kind == 'peaky':
        Q = Q * 20.0
    elif kind == 'wide_range':
        X = X * torch.from_numpy(10.0 ** rs.uniform(-6, 3, size=(B, Tp, 1)).astype(np.float32))
    else:
        PX = PX * torch.from_numpy(10.0 ** rs.uniform(-4, 0, size=(1, 1, E)).astype(np.float32))
        Q = Q * torch.from_numpy(10.0 ** rs.uniform(0, 3, size=(1, E)).astype(np.float32)) * 0.05
        X = X * 1e4 + 1e-3
    S = torch.einsum('le,bte->blt', Q.double(), PX.double())
    lse64 = torch.logsumexp(S, dim=2)
    O64 = torch.einsum('blt,btf->blf', torch.softmax(S, dim=2), X.double())
    errs = {}
    for name, flag in (('fp32', False), ('bf16x6', True)):
        ops = HipOps()
        ops.attn_bf16x6 = flag
        O, lse = ops.attn_fwd(g(PX), g(X), g(Q), 0, L, 1, 0.0)
        errs[name] = (float((O.double().cpu() - O64).abs().max() / O64.abs().max()),
                      float((lse.double().cpu() - lse64).abs().max()))
    # same order of error as the fp32 kernel (both accumulate in fp32; what differs is the order and the dropped 2^-24 terms)
    assert errs['bf16x6'][0] <= 2.0 * errs['fp32'][0] + 2e-7, errs
    assert errs['bf16x6'][1] <= 2.0 * errs['fp32'][1] + 2e-6, errs


@pytest.mark.parametrize('B,Tp,E,F,L', ATTN_SHAPES)
@pytest.mark.parametrize('p', [0.0, 0.2])
def test_label_attention_fwd_bwd(ops, ref, B, Tp, E, F, L, p):
    rs = np.random.RandomState(B * 1000 + Tp + L)
    PX = torch.tanh(rnd(rs, B, Tp, E))
    X = torch.relu(rnd(rs, B, Tp, F))
    Q = rnd(rs, L, E, scale=0.3)
    l_off, L_tot, seed = 5, L + 9, 4321
    O, lse = ops.attn_fwd(g(PX), g(X), g(Q), l_off, L_tot, seed, p)
    Or, lser = ref.attn_fwd(PX, X, Q, l_off, L_tot, seed, p)
    close(lse, lser, 1e-5, 1e-5, 'attention log-sum-exp')
    close(O, Or, 2e-5, 2e-5, 'attention output')
    dO = rnd(rs, B, L, F)
    dPX, dX, dQ = torch.zeros(B, Tp, E, device=DEV), torch.zeros(B, Tp, F, device=DEV), torch.zeros(L, E, device=DEV)
    ops.attn_bwd(g(dO), O, g(PX), g(X), g(Q), lse, l_off, L_tot, seed, p, dPX, dX, dQ)
    dPXr, dXr, dQr = torch.zeros(B, Tp, E), torch.zeros(B, Tp, F), torch.zeros(L, E)
    ref.attn_bwd(dO, Or, PX, X, Q, lser, l_off, L_tot, seed, p, dPXr, dXr, dQr)
    close(dX, dXr, 1e-4, 2e-5 * np.sqrt(L), 'attention dX')
    close(dPX, dPXr, 1e-4, 2e-5 * np.sqrt(L), 'attention dPX')
    close(dQ, dQr, 1e-4, 2e-5 * np.sqrt(B * Tp), 'attention dQ')
    # label splits must not change the result beyond summation order
    dPX2, dX2, dQ2 = torch.zeros_like(dPX), torch.zeros_like(dX), torch.zeros_like(dQ)
    ops.attn_bwd(g(dO), O, g(PX), g(X), g(Q), lse, l_off, L_tot, seed, p, dPX2, dX2, dQ2, nsplit=3)
    close(dX2, dX, 1e-5, 1e-5, 'attention dX, nsplit=3')
    close(dQ2, dQ, 1e-5, 1e-5, 'attention dQ, nsplit=3')


def test_label_attention_online_softmax_rescale_branch(ops, ref):
    """Force the lazy-rescale branch (cdna guide rule 26): one key late in the sequence dominates the running max by
    far more than the rescale threshold, for only some labels."""
    rs = np.random.RandomState(9)
    B, Tp, E, F, L = 1, 200, 100, 200, 70
    PX = torch.tanh(rnd(rs, B, Tp, E, scale=0.2))
    X = torch.relu(rnd(rs, B, Tp, F))
    Q = rnd(rs, L, E, scale=0.2)
    PX[0, 150] = torch.sign(Q[3])          # score for label 3 jumps to ||q_3||_1 ~ 16 at token 150
    Q[3] *= 3.0
    PX[0, 170] = torch.sign(Q[40]) * 0.99  # and for label 40 at token 170
    Q[40] *= 5.0
    O, lse = ops.attn_fwd(g(PX), g(X), g(Q), 0, L, 0, 0.0)
    Or, lser = ref.attn_fwd(PX, X, Q, 0, L, 0, 0.0)
    assert float(lser.max()) > 40, 'test input must exceed the rescale threshold'
    close(lse, lser, 1e-5, 1e-5, 'lse with forced rescale')
    close(O, Or, 2e-5, 2e-5, 'output with forced rescale')


def test_label_attention_shard_invariance(ops):
    """Rows [l0, l1) of a sharded call equal the same rows of the unsharded call (dropout keyed on global coordinates).
    Not bit for bit: the lazy-rescale decision is taken per wave, so a label's rounding depends on its 15 neighbours."""
    rs = np.random.RandomState(10)
    B, Tp, E, F, L = 2, 90, 100, 200, 150
    PX, X, Q = g(torch.tanh(rnd(rs, B, Tp, E))), g(torch.relu(rnd(rs, B, Tp, F))), g(rnd(rs, L, E, scale=0.3))
    O, lse = ops.attn_fwd(PX, X, Q, 0, L, 99, 0.2)
    for l0, l1 in ((0, 75), (75, 150), (10, 31)):
        Os, lses = ops.attn_fwd(PX, X, Q[l0:l1].contiguous(), l0, L, 99, 0.2)
        close(Os, O[:, l0:l1], 1e-5, 1e-6, f'shard [{l0},{l1}) output')
        close(lses, lse[:, l0:l1], 1e-6, 1e-6, f'shard [{l0},{l1}) lse')


def test_head_and_loss_kernels(ops, ref):
    rs = np.random.RandomState(6)
    B, L, D = 5, 77, 200
    direct, clf = rnd(rs, B, L, D), rnd(rs, L, D, scale=0.2)
    logits = ops.logits_fwd(g(direct), g(clf))
    lr_ = ref.logits_fwd(direct, clf)
    close(logits, lr_, 1e-5, 2e-5, 'logits')
    y = torch.from_numpy((rs.random_sample((B, L)) < 0.1).astype(np.float32))
    margin = torch.from_numpy(rs.uniform(0.2, 1.0, L).astype(np.float32))
    for m, C in ((margin, 2.0), (None, 0.0)):
        loss, dl, probs = ops.ldam_bce(logits, g(y), g(m), C, 1.0 / B, need_probs=True)
        lossr, dlr, probsr = ref.ldam_bce(lr_, y, m, C, 1.0 / B, need_probs=True)
        close(loss, lossr, 1e-5, 1e-4, 'ldam bce loss')
        close(dl, dlr, 1e-5, 1e-7, 'ldam bce dlogits')
        close(probs, probsr, 1e-5, 1e-6, 'sigmoid scores')
    dd, dclf = ops.logits_bwd(g(direct), g(clf), dl)
    ddr, dclfr = ref.logits_bwd(direct, clf, dlr)
    close(dd, ddr, 1e-5, 1e-7, 'ddirect')
    close(dclf, dclfr, 1e-5, 1e-6, 'dclf')
    Bs = 33
    fc, feats = rnd(rs, L, D, scale=0.2), torch.relu(rnd(rs, Bs, D))
    labs = torch.from_numpy(rs.randint(0, L, size=Bs).astype(np.int64))
    labs[1] = labs[0]
    loss, dfc = ops.forward_final(g(fc), g(feats), g(labs), g(margin), 2.0, 1.0 / Bs)
    lossr, dfcr = ref.forward_final(fc, feats, labs, margin, 2.0, 1.0 / Bs)
    close(loss, lossr, 1e-5, 1e-5, 'forward_final loss')
    close(dfc, dfcr, 1e-5, 1e-6, 'forward_final dfc')
    p, gr, m_, v_ = rnd(rs, L, D), rnd(rs, L, D, scale=0.01), rnd(rs, L, D, scale=0.01), rnd(rs, L, D, scale=0.01).abs()
    pd, md, vd = g(p).clone(), g(m_).clone(), g(v_).clone()
    ops.adamw_step_(pd, g(gr), md, vd, 1e-3, 0.5, 0.999, 1e-6, 5e-4)
    ref.adamw_step_(p, gr, m_, v_, 1e-3, 0.5, 0.999, 1e-6, 5e-4)
    close(pd, p, 1e-6, 1e-7, 'adamw p')
    close(md, m_, 1e-6, 1e-8, 'adamw m')
    close(vd, v_, 1e-6, 1e-9, 'adamw v')


def test_small_helpers(ops, ref):
    rs = np.random.RandomState(8)
    x = rnd(rs, 5000, 300)
    close(ops.colsum(g(x)), x.sum(0), 1e-5, 2e-3, 'colsum')
    close(ops.colsum(g(x)[:, :100]), x[:, :100].sum(0), 1e-5, 2e-3, 'colsum strided')
    for act in (1, 2, 3):
        y, d = (torch.tanh(rnd(rs, 1000, 7)) if act == 2 else rnd(rs, 1000, 7)), rnd(rs, 1000, 7)
        close(ops.act_bwd_(g(d).clone(), g(y), act), ref.act_bwd_(d.clone(), y, act), 1e-6, 1e-7, f'act bwd {act}')


def test_gan_kernels(ops, ref):
    rs = np.random.RandomState(12)
    Bg, D, H, Vk, K = 70, 400, 800, 300, 20
    real, fake, alpha = rnd(rs, Bg, D), rnd(rs, Bg, D), torch.from_numpy(rs.random_sample(Bg).astype(np.float32))
    close(ops.gp_interp(g(real), g(fake), g(alpha)), ref.gp_interp(real, fake, alpha), 1e-6, 1e-6, 'interp')
    h, w2 = rnd(rs, Bg, H), rnd(rs, H)
    for a, b, nm in zip(ops.gp_slope_v(g(h), g(w2)), ref.gp_slope_v(h, w2), 'vs'):
        close(a, b, 0, 0, 'gp ' + nm)
    gg = rnd(rs, Bg, D, scale=0.1)
    ggd = g(gg).clone()
    gp = ops.gp_rownorm_(ggd, 10.0)
    gpr = ref.gp_rownorm_(gg, 10.0)
    close(gp, gpr, 1e-5, 1e-5, 'gp value')
    close(ggd, gg, 1e-5, 1e-7, 'gp dg')
    dv, s = rnd(rs, Bg, H), ref.gp_slope_v(h, w2)[1]
    dw2 = torch.zeros(H, device=DEV)
    ops.gp_dw2_(g(dv), g(s), dw2)
    close(dw2, (dv * s).sum(0), 1e-5, 1e-5, 'gp dw2')
    bias = rnd(rs, 1)
    close(ops.rowdot(g(h), g(w2), g(bias)), ref.rowdot(h, w2, bias), 1e-5, 1e-4, 'rowdot')
    wgt = rnd(rs, Bg)
    close(ops.weighted_sum(g(wgt), g(alpha), 0.3), ref.weighted_sum(wgt, alpha, 0.3), 1e-5, 1e-5, 'weighted sum')
    dout = rnd(rs, Bg)
    dw, db = torch.zeros(H, device=DEV), torch.zeros(1, device=DEV)
    dwr, dbr = torch.zeros(H), torch.zeros(1)
    da = ops.rowdot_bwd(g(dout), g(w2), g(h), dw, db)
    dar = ref.rowdot_bwd(dout, w2, h, dwr, dbr)
    close(da, dar, 1e-6, 1e-7, 'rowdot bwd da')
    close(dw, dwr, 1e-5, 1e-4, 'rowdot bwd dw')
    close(db, dbr, 1e-5, 1e-5, 'rowdot bwd db')
    logits = rnd(rs, Bg, Vk)
    kw_idx = torch.from_numpy(np.stack([rs.choice(Vk, size=K, replace=False) for _ in range(Bg)]).astype(np.int64))
    kw_idx[:, 5] = kw_idx[:, 2]  # duplicates: assignment semantics, last one wins
    kw_mask = torch.from_numpy(rs.uniform(0, 1, (Bg, K)).astype(np.float32))
    w = torch.from_numpy(rs.uniform(0, 1, Bg).astype(np.float32))
    ld = g(logits).clone()
    lb = ops.keyword_ce_(ld, g(kw_idx), g(kw_mask), g(w))
    lr2 = logits.clone()
    lbr = ref.keyword_ce_(lr2, kw_idx, kw_mask, w)
    close(lb, lbr, 1e-5, 1e-5, 'keyword CE loss')
    close(ld, lr2, 1e-5, 1e-6, 'keyword CE dlogits')


@pytest.mark.parametrize('Bg,H,D,Vk', [(8, 800, 400, 4096), (70, 800, 400, 300), (33, 50, 37, 301), (300, 804, 1028, 8196)])
def test_gan_iteration_kernels(ops, ref, Bg, H, D, Vk):
    """The whole-iteration kernels of the WGAN-GP step (critic / generator head, column-owner dw2, register-resident
    row norm and keyword CE incl. their scalar fallbacks: H, D, Vk not multiples of 4, D > 1024, Vk > 8192)."""
    rs = np.random.RandomState(Bg + H)
    hf, lh, w2, b2 = rnd(rs, 2 * Bg, H), rnd(rs, Bg, H), rnd(rs, H), rnd(rs, 1)
    sim = torch.from_numpy(rs.choice([1.0, 0.4, 0.0], size=Bg).astype(np.float32))
    alpha = torch.from_numpy(rs.random_sample(Bg).astype(np.float32))
    for sm in (sim, None):
        dw2, db1, cost = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV), torch.zeros(1, device=DEV)
        dw2r, db1r, costr = torch.zeros(H), torch.zeros(H), torch.zeros(1)
        got = ops.gan_critic_head(g(hf), g(lh), g(w2), g(sm), g(alpha), dw2, db1, cost)
        want = ref.gan_critic_head(hf, lh, w2, sm, alpha, dw2r, db1r, costr)
        for a, b, nm in zip(got, want, ('lst', 'dhs', 'slopes')):
            close(a, b, 1e-6, 1e-7, 'critic head ' + nm)
        close(dw2, dw2r, 1e-5, 2e-6, 'critic head dw2')
        close(db1, db1r, 1e-5, 2e-6, 'critic head db1')
        close(cost, costr, 1e-4, 1e-5 * np.sqrt(H), 'critic head cost')
        cost, costr = torch.zeros(1, device=DEV), torch.zeros(1)
        close(ops.gan_gen_head(g(hf[:Bg]), g(lh), g(w2), g(b2), g(sm), cost), ref.gan_gen_head(hf[:Bg], lh, w2, b2, sm, costr),
              1e-6, 1e-7, 'generator head dh')
        close(cost, costr, 1e-4, 1e-5 * np.sqrt(H), 'generator head cost')
    dv, s = rnd(rs, Bg, H), ref.gp_slope_v(lh, w2)[1]
    dw2 = torch.zeros(H, device=DEV)
    ops.gp_dw2_(g(dv), g(s), dw2)
    close(dw2, (dv * s).sum(0), 1e-5, 1e-5, 'gp dw2')
    gg = rnd(rs, Bg, D, scale=0.1)
    ggd, acc = g(gg).clone(), torch.full((1,), 0.5, device=DEV)
    ops.gp_rownorm_(ggd, 10.0, out=acc)
    close((acc - 0.5)[0], ref.gp_rownorm_(gg, 10.0), 1e-5, 1e-5, 'gp value (accumulated into out)')
    close(ggd, gg, 1e-5, 1e-7, 'gp dg')
    K = 20
    logits = rnd(rs, Bg, Vk)
    kw_idx = torch.from_numpy(np.stack([rs.choice(Vk, size=K, replace=False) for _ in range(Bg)]).astype(np.int64))
    kw_idx[:, 5] = kw_idx[:, 2]
    kw_mask = torch.from_numpy(rs.uniform(0, 1, (Bg, K)).astype(np.float32))
    for sm in (sim, None):
        ld, tot = g(logits).clone(), torch.zeros(1, device=DEV)
        lr2, totr = logits.clone(), torch.zeros(1)
        lb = ops.keyword_ce_sim_(ld, g(kw_idx), g(kw_mask), g(sm), 0.01 / Bg, 1.0 / Bg, tot)
        lbr = ref.keyword_ce_sim_(lr2, kw_idx, kw_mask, sm, 0.01 / Bg, 1.0 / Bg, totr)
        close(lb, lbr, 1e-5, 1e-5, 'keyword CE loss')
        close(ld, lr2, 1e-5, 1e-9, 'keyword CE dlogits')
        close(tot, totr, 1e-5, 1e-6, 'keyword CE weighted loss sum')


def test_adam_flat_matches_torch_adam(ops):
    """gzsl_adam_flat_step against torch.optim.Adam(betas=(0.5, 0.999)) over 7 steps (bias correction from the
    device-side step counter), n not a multiple of 4."""
    rs = np.random.RandomState(3)
    n = 64 * 1000 + 64
    p0 = rnd(rs, n)
    grads = [rnd(rs, n, scale=10.0 ** rs.uniform(-4, 0)) for _ in range(7)]
    pt = torch.nn.Parameter(p0.clone().double())
    opt = torch.optim.Adam([pt], lr=1e-3, betas=(0.5, 0.999))
    p, m, v = g(p0).clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    state = torch.zeros(ops.ADAM_STATE_INTS, dtype=torch.int32, device=DEV)
    for i, gr in enumerate(grads):
        pt.grad = gr.double()
        opt.step()
        ops.adam_flat_step_(p, g(gr), m, v, state, 1e-3, 0.5, 0.999, 1e-8)
        nblk = min(1024, (n // 4 + 255) // 256)
        assert state[:nblk].eq(i + 1).all() and state[nblk:].eq(0).all()
    close(p, pt.detach().float(), 1e-6, 2e-7, 'parameters after 7 Adam steps (fp64 torch reference)')


def test_gemm_random_shapes_and_views(ops, ref):
    """Random shapes, transposes, column slices (ldc > N, unaligned bases) and split-K sizes: every vector / scalar /
    fast / masked path of the GEMM against torch."""
    rs = np.random.RandomState(2024)
    for it in range(40):
        M, N, K = int(rs.randint(1, 400)), int(rs.randint(1, 330)), int(rs.choice([1, 3, 16, 17, 64, 100, 200, 333, 4096]))
        if it % 5 == 0:
            M, N, K = 4 * int(rs.randint(1, 80)), 4 * int(rs.randint(1, 80)), 4 * int(rs.randint(1, 600))
        ta, tb = bool(rs.randint(2)), bool(rs.randint(2))
        pad_a, pad_b = int(rs.choice([0, 0, 4, 5])), int(rs.choice([0, 0, 8, 3]))
        A_store = rnd(rs, *((K, M + pad_a) if ta else (M, K + pad_a)))
        B_store = rnd(rs, *((N, K + pad_b) if tb else (K, N + pad_b)))
        A = A_store[:, pad_a:].t() if ta else A_store[:, pad_a:]
        B = B_store[:, pad_b:].t() if tb else B_store[:, pad_b:]
        Ad = g(A_store)[:, pad_a:].t() if ta else g(A_store)[:, pad_a:]
        Bd = g(B_store)[:, pad_b:].t() if tb else g(B_store)[:, pad_b:]
        bias = rnd(rs, N)
        act = int(rs.randint(4))
        wide, c0 = (N + 8, 4) if it % 10 == 0 else (N + 7, 3)  # aligned slice -> vector epilogue, else scalar
        C0 = rnd(rs, M, wide)
        out = g(C0).clone()
        ops.mm(Ad, Bd, out=out[:, c0:c0 + N], bias=g(bias), act=act, accumulate=bool(it % 2), alpha=0.5 if it % 3 == 0 else 1.0)
        want = C0.clone()
        want[:, c0:c0 + N] = ref.mm(A, B, out=C0[:, c0:c0 + N].clone(), bias=bias, act=act, accumulate=bool(it % 2),
                                  alpha=0.5 if it % 3 == 0 else 1.0)
        close(out, want, 3e-5, 3e-5 * np.sqrt(K), f'mm case {it}: M={M} N={N} K={K} ta={ta} tb={tb} act={act}')


def test_attention_random_shapes(ops, ref):
    """Random (B, T', E, F, L) inside every instantiated shape class, dropout on: forward and backward against torch."""
    rs = np.random.RandomState(77)
    for it in range(14):
        E = int(rs.choice([4, 8, 12, 52, 100, 104, 200]))
        F = int(rs.choice([8, 20, 32, 36, 120, 200, 208]))
        B, Tp, L = int(rs.randint(1, 4)), int(rs.randint(1, 200)), int(rs.randint(1, 150))
        PX, X = torch.tanh(rnd(rs, B, Tp, E)), torch.relu(rnd(rs, B, Tp, F))
        Q = rnd(rs, L, E, scale=0.3)
        l_off, L_tot, seed, p = int(rs.randint(0, 9)), L + 11, int(rs.randint(1, 10000)), float(rs.choice([0.0, 0.2]))
        O, lse = ops.attn_fwd(g(PX), g(X), g(Q), l_off, L_tot, seed, p)
        Or, lser = ref.attn_fwd(PX, X, Q, l_off, L_tot, seed, p)
        what = f'case {it}: B={B} Tp={Tp} E={E} F={F} L={L} p={p}'
        close(lse, lser, 1e-5, 1e-5, 'lse ' + what)
        close(O, Or, 2e-5, 2e-5, 'O ' + what)
        dO = rnd(rs, B, L, F)
        dPX, dX, dQ = torch.zeros(B, Tp, E, device=DEV), torch.zeros(B, Tp, F, device=DEV), torch.zeros(L, E, device=DEV)
        ops.attn_bwd(g(dO), O, g(PX), g(X), g(Q), lse, l_off, L_tot, seed, p, dPX, dX, dQ)
        dPXr, dXr, dQr = torch.zeros(B, Tp, E), torch.zeros(B, Tp, F), torch.zeros(L, E)
        ref.attn_bwd(dO, Or, PX, X, Q, lser, l_off, L_tot, seed, p, dPXr, dXr, dQr)
        close(dX, dXr, 1e-4, 2e-5 * np.sqrt(L), 'dX ' + what)
        close(dPX, dPXr, 1e-4, 2e-5 * np.sqrt(L), 'dPX ' + what)
        close(dQ, dQr, 1e-4, 2e-5 * np.sqrt(B * Tp), 'dQ ' + what)


@pytest.mark.parametrize('B,L,D', [(16, 300, 200), (3, 41, 6), (1, 5, 8), (7, 130, 52)])
def test_logits_bwd_fused_and_fallback_paths(ops, ref, B, L, D):
    """D % 4 == 0 takes the one-pass kernel (both gradients from one read of `direct`), otherwise the two scalar kernels;
    need_ddirect=False (GAN model: only the classifier rows train) takes the dclf kernel alone."""
    rs = np.random.RandomState(B * 1000 + L + D)
    direct, clf, dl = rnd(rs, B, L, D), rnd(rs, L, D, scale=0.2), rnd(rs, B, L, scale=0.1)
    dd, dclf = ops.logits_bwd(g(direct), g(clf), g(dl))
    ddr, dclfr = ref.logits_bwd(direct, clf, dl)
    close(dd, ddr, 1e-6, 1e-7, 'ddirect')
    close(dclf, dclfr, 1e-5, 1e-6, 'dclf')
    none, dclf2 = ops.logits_bwd(g(direct), g(clf), g(dl), need_ddirect=False)
    assert none is None and torch.equal(dclf2, dclf)   # same summation order over the notes in both kernels


@pytest.mark.parametrize('B,L,D,F', [(3, 37, 24, 40), (16, 300, 200, 200), (2, 70, 400, 200), (5, 9, 1024, 16)])
def test_logits_bwd_with_attention_delta(ops, ref, B, L, D, F):
    """gzsl_logits_bwd_delta: ddirect / dclf as gzsl_logits_bwd, and delta = rowsum(dO * O) obtained from `direct` and the
    output_proj bias alone, against the definition (dO = ddirect @ Wo, then gzsl_attn_delta's sum)."""
    rs = np.random.RandomState(B * 1000 + L + D)
    O, Wo, bo = rnd(rs, B, L, F), rnd(rs, D, F, scale=0.2), rnd(rs, D)
    clf, dlogits = rnd(rs, L, D), rnd(rs, B, L)
    direct = O @ Wo.t() + bo
    dd_r, dclf_r = ref.logits_bwd(direct, clf, dlogits)
    delta_r = ((dd_r.double() @ Wo.double()) * O.double()).sum(2).float()
    dd, dclf, delta = ops.logits_bwd_delta(g(direct), g(clf), g(bo), g(dlogits))
    close(dd, dd_r, 1e-6, 1e-7, 'ddirect')
    close(dclf, dclf_r, 1e-5, 2e-6 * np.sqrt(B), 'dclf')
    # fp32: direct - bo recovers O.Wo^T up to one rounding of |direct| per term
    close(delta, delta_r, 1e-4, 3e-6 * np.sqrt(D) * float(direct.abs().max() * dd_r.abs().max()), 'delta')
    dd2, dclf2 = ops.logits_bwd(g(direct), g(clf), g(dlogits))
    assert torch.equal(dd2, dd) and torch.equal(dclf2, dclf)  # same arithmetic, same summation order over notes


def test_ldam_bce_many_blocks(ops, ref):
    rs = np.random.RandomState(12)
    B, L = 16, 9000
    logits, y = rnd(rs, B, L, scale=3.0), torch.from_numpy((rs.random_sample((B, L)) < 0.002).astype(np.float32))
    margin = torch.from_numpy(rs.uniform(0.2, 1.0, L).astype(np.float32))
    loss, dl, _ = ops.ldam_bce(g(logits), g(y), g(margin), 2.0, 1.0 / B)
    lossr, dlr, _ = ref.ldam_bce(logits, y, margin, 2.0, 1.0 / B)
    close(loss, lossr, 2e-6, 1e-3, 'ldam bce loss over 563 blocks')
    close(dl, dlr, 1e-5, 1e-7, 'dlogits')


def test_attention_full_size_properties(ops):
    """BASELINE.json configs[1] sizes (B=16, T'=2491, E=100, F=200, L=9000), where the oracle is too slow: properties
    that do not depend on size.  (1) P is a distribution: with all-ones values and no dropout every output is 1.
    (2) label shards reproduce the unsharded rows.  (3) the backward is linear in dO.  (4) sum_t dS[l,t] = 0 for
    every label, so with the same key at every token the query gradient vanishes."""
    gen = torch.Generator(device=DEV).manual_seed(5)
    B, Tp, E, F, L = 16, 2491, 100, 200, 9000
    PX = torch.tanh(torch.randn(B, Tp, E, device=DEV, generator=gen))
    X = torch.relu(torch.randn(B, Tp, F, device=DEV, generator=gen))
    Q = torch.randn(L, E, device=DEV, generator=gen) * 0.2
    # (1)
    O1, lse1 = ops.attn_fwd(PX, torch.ones_like(X), Q, 0, L, 7, 0.0)
    assert float((O1 - 1).abs().max()) < 2e-5
    # logsumexp bounds: max_t s <= lse <= max_t s + log T'
    smax = torch.stack([(PX[b] @ Q[:512].t()).max(0).values for b in range(B)])
    assert bool((lse1[:, :512] >= smax - 1e-4).all()) and bool((lse1[:, :512] <= smax + np.log(Tp) + 1e-4).all())
    # (2)
    O, lse = ops.attn_fwd(PX, X, Q, 0, L, 7, 0.2)
    for l0, l1 in ((0, 1125), (4500, 9000), (8999, 9000)):
        Os, lses = ops.attn_fwd(PX, X, Q[l0:l1].contiguous(), l0, L, 7, 0.2)
        close(Os, O[:, l0:l1], 1e-5, 2e-6, f'full-size shard [{l0},{l1})')
        close(lses, lse[:, l0:l1], 1e-6, 2e-6, f'full-size shard lse [{l0},{l1})')
    # (3)
    dO = torch.randn(B, L, F, device=DEV, generator=gen) * 0.01
    outs = []
    for scale in (1.0, -2.0):
        dPX, dX, dQ = torch.zeros_like(PX), torch.zeros_like(X), torch.zeros_like(Q)
        ops.attn_bwd((dO * scale).contiguous(), O, PX, X, Q, lse, 0, L, 7, 0.2, dPX, dX, dQ)
        outs.append((dPX, dX, dQ))
    for a, b, what in zip(outs[0], outs[1], ('dPX', 'dX', 'dQ')):
        assert bool(torch.isfinite(a).all())
        close(b, -2.0 * a, 2e-4, 2e-5 * float(a.abs().max()), f'linearity of {what} in dO')
    # (4) identical keys at every token: S[l,t] does not depend on t, P is uniform, dS[l,t] = P (dP[l,t] - delta[l])
    # sums to zero over t, so dQ = sum_t dS[l,t] K[t] = K * 0 = 0 (up to rounding against the size of its terms)
    PXc = PX[:, :1].expand(B, Tp, E).contiguous()
    Oc, lsec = ops.attn_fwd(PXc, X, Q, 0, L, 7, 0.0)
    dPX, dX, dQ = torch.zeros_like(PX), torch.zeros_like(X), torch.zeros_like(Q)
    ops.attn_bwd(dO, Oc, PXc, X, Q, lsec, 0, L, 7, 0.0, dPX, dX, dQ)
    terms = float(dO.abs().mean()) * float(X.abs().mean()) * F   # size of one dP term
    assert float(dQ.abs().max()) < 1e-4 * terms * B, (float(dQ.abs().max()), terms)


def test_conv_full_size_linearity(ops):
    """Implicit-GEMM conv at cfg2 size: conv(e1 + e2) - b = (conv(e1) - b) + (conv(e2) - b) before the ReLU is not
    observable, so use non-negative inputs and weights (ReLU inactive) and check additivity and the bias shift."""
    gen = torch.Generator(device=DEV).manual_seed(6)
    B, T, E, F, k = 16, 2500, 100, 200, 10
    e1, e2 = torch.rand(B, T, E, device=DEV, generator=gen), torch.rand(B, T, E, device=DEV, generator=gen)
    W, b = torch.rand(F, E, k, device=DEV, generator=gen) * 0.01, torch.rand(F, device=DEV, generator=gen)
    y1, y2, y12 = ops.conv_relu(e1, W, b), ops.conv_relu(e2, W, b), ops.conv_relu(e1 + e2, W, b)
    assert y1.shape == (B, T - k + 1, F)
    close(y12 + b, y1 + y2, 1e-5, 1e-4, 'conv additivity (ReLU inactive)')
    # one output against a direct dot product
    want = float((e1[3, 77:77 + k].t() * W[5]).sum() + b[5])
    assert abs(float(y1[3, 77, 5]) - want) < 1e-4 * abs(want)


@pytest.mark.parametrize('B,L,F,D', [(2, 37, 52, 36), (3, 70, 200, 200), (1, 5, 24, 16), (2, 129, 200, 208), (16, 64, 100, 4)])
@pytest.mark.parametrize('need_direct', [True, False])
def test_fused_output_proj_and_label_logits(ops, B, L, F, D, need_direct):
    """gzsl_gemm's fused head epilogue (icd_modules.py:159 + 177-182 in one launch) against the two-launch form and the
    torch-fp32 statement; need_direct=False must not need (or touch) a `direct` buffer."""
    rs = np.random.RandomState(B * 100 + L + D)
    O, Wo, bo, clf = g(rnd(rs, B * L, F)), g(rnd(rs, D, F, scale=0.2)), g(rnd(rs, D, scale=0.1)), g(rnd(rs, L, D))
    out = ops.linear_logits(O, Wo, bo, clf, need_direct=need_direct)
    assert out is not None
    direct, logits = out
    want_direct = O.double() @ Wo.double().t() + bo.double()
    want = (torch.relu(want_direct).view(B, L, D) * clf.double()[None]).sum(2).reshape(-1)
    close(logits, want, 1e-5, 2e-5 * np.sqrt(D), 'fused logits')
    if need_direct:
        close(direct, want_direct, 1e-5, 1e-5 * np.sqrt(F), 'direct')
        two = ops.logits_fwd(ops.linear(O, Wo, bo).view(B, L, D), clf).reshape(-1)
        close(logits, two, 1e-5, 1e-5 * np.sqrt(D), 'fused vs two launches')
    else:
        assert direct is None


def test_fused_head_declines_shapes_without_a_form(ops):
    O, Wo, bo = torch.zeros(6, 8, device=DEV), torch.zeros(210, 8, device=DEV), torch.zeros(210, device=DEV)
    assert ops.linear_logits(O, Wo, bo, torch.zeros(3, 210, device=DEV)) is None        # wider than one tile
    assert ops.linear_logits(O, Wo[:6], bo[:6], torch.zeros(3, 6, device=DEV)) is None  # D % 4
