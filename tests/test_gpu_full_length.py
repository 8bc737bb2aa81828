"""Full token-length parity on the GPU: the HIP attention kernels at T' = 2491 (BASELINE configs[1]) and
T' = 3991 (configs[2]), E in {100, 200}, on label SUB-SHARDS carrying their real global coordinates (l_offset inside
L_total = 9000, as rank r of an 8-way label shard sees them), against the numpy oracle (oracle/base_model.py, pinned to
the reference by tests/test_oracle_golden.py) -- not through size-independent properties.  Plus one whole-model
train-mode step at the per-rank shape of configs[2] (B=2 notes, T=4000, E=H=100, L=1125 = 9000/8 labels).

The oracle finishes these sizes in seconds because the label count is a shard's worth, not 9000.
Tolerances (fp32): forward rtol 1e-5..2e-5; gradients rtol 1e-4 with an absolute floor scaled by sqrt(#terms), exactly
as tests/test_gpu_kernels.py::test_label_attention_fwd_bwd states them.
"""
import numpy as np
import pytest
import torch

from oracle import base_model as om
from tests import helpers
from tests.golden import synth
from tests.helpers import close

pytestmark = pytest.mark.gpu
DEV = 'cuda'
L_TOTAL = 9000


@pytest.fixture(scope='module')
def ops():
    from gzsl_text_amd.ops_hip import HipOps
    return HipOps()


def _g(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize('Tp', [2491, 3991])
@pytest.mark.parametrize('E', [100, 200])
@pytest.mark.parametrize('l_offset,L', [(0, 64), (1125 * 3, 130), (L_TOTAL - 64, 64)])
def test_attention_full_token_length_vs_oracle(ops, Tp, E, l_offset, L):
    B, F, seed, p = 2, 200, 20201, 0.2
    rs = np.random.RandomState(Tp + E + l_offset)
    PX = np.tanh(rs.standard_normal((B, Tp, E))).astype(np.float32)
    X = np.maximum(rs.standard_normal((B, Tp, F)), 0).astype(np.float32)
    Q = (rs.standard_normal((L, E)) * 0.3).astype(np.float32)
    dO = rs.standard_normal((B, L, F)).astype(np.float32)
    Or, lser, cache = om.label_attn_fwd(PX, X, Q, drop_seed=seed, p_attn=p, l_offset=l_offset, L_total=L_TOTAL)
    dPXr, dXr, dQr = om.label_attn_bwd(dO, PX, X, Q, cache)
    PXd, Xd, Qd, dOd = _g(PX), _g(X), _g(Q), _g(dO)
    O, lse = ops.attn_fwd(PXd, Xd, Qd, l_offset, L_TOTAL, seed, p)
    close(lse, lser, 1e-5, 1e-5, 'log-sum-exp')
    close(O, Or, 2e-5, 2e-5, 'attention output')
    dPX, dX, dQ = torch.zeros_like(PXd), torch.zeros_like(Xd), torch.zeros_like(Qd)
    ops.attn_bwd(dOd, O, PXd, Xd, Qd, lse, l_offset, L_TOTAL, seed, p, dPX, dX, dQ)
    close(dX, dXr, 1e-4, 2e-5 * np.sqrt(L), 'dX')
    close(dPX, dPXr, 1e-4, 2e-5 * np.sqrt(L), 'dPX')
    close(dQ, dQr, 1e-4, 2e-5 * np.sqrt(B * Tp), 'dQ')
    # the dropout mask must be the one of the GLOBAL label rows: the same queries at another offset give another output
    if l_offset:
        O0, _ = ops.attn_fwd(PXd, Xd, Qd, 0, L_TOTAL, seed, p)
        assert float((O0 - O).abs().max()) > 1e-3


def test_base_model_step_at_cfg3_rank_shape_vs_oracle():
    """One train-mode step (both dropouts on, LDAM C=2, gate encoder) of the whole base model at B=2, T=4000 (T'=3991),
    E=H=100, F=200, L=1125: logits, loss and every parameter gradient against the oracle."""
    case = synth.synth_base(seed=77, B=2, T=4000, E=100, H=100, L=1125, N=1300, Td=30, V=3000, F=200, labels_per_note=15,
                            emb_scale=0.1, zipf=True)
    d = case['dims']
    margin = om.class_margin_from_count(case['target_count'])
    want = om.base_forward_backward(case['params'], case['bufs'], case['x'], case['y'], case['mask'], d['L'], margin, 2.0,
                                    drop_seed=synth.DROP_SEED, l_chunk=375)
    model = helpers.build_model(case, device=DEV)
    model.train()
    model.set_dropout_seed(synth.DROP_SEED)
    x, y, mask = helpers.batch_tensors(case, DEV)
    model.zero_grad()
    logits, loss = model(x, y, mask)
    loss.mean().backward()
    close(logits, want['logits'], 1e-5, 2e-5, 'logits')
    close(loss, want['loss'], 1e-5, 1e-5, 'loss')
    for k, prm in model.named_parameters():
        ref = want['grads'][k]
        # sums over up to B*T' = 7982 tokens x 1125 labels: absolute floor relative to the gradient's own scale
        close(prm.grad, ref, 1e-4, max(4e-6, 2e-5 * float(np.abs(ref).max())), 'grad ' + k)
