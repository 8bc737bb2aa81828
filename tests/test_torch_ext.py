"""The TORCH_LIBRARY binding (gzsl_text_amd/csrc/torch_ext.cpp -> torch.ops.gzsl.*).

CPU: the extension loads, registers the three operators with the documented schemas and refuses CPU tensors.
GPU: the reference's own forward (modules/icd_modules.py:125-186), written out with ATen for the layers the extension
does not replace and with the three operators where INTEGRATION.md 2b puts them, reproduces the reference's golden
logits / loss / parameter gradients; the attention operator agrees with the ctypes binding of the same kernels, honours
the dropout coordinates (seed, l_offset, L_total) of the oracle, and works under no_grad.
"""
import os

import numpy as np
import pytest
import torch

from tests.helpers import close

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def ext():
    from gzsl_text_amd import torch_ext
    torch_ext.build()  # no-op when lib/libgzsl_torch.so is newer than its sources (g++, ~15 s otherwise)
    torch_ext.load()
    return torch.ops.gzsl


def test_extension_registers_the_documented_operators(ext):
    s = str(ext.label_attention.default._schema)
    assert s == ('gzsl::label_attention(Tensor keys, Tensor values, Tensor queries, float p_drop=0., int seed=0, '
                 'int l_offset=0, int L_total=0) -> (Tensor, Tensor)'), s
    assert str(ext.label_logits.default._schema) == 'gzsl::label_logits(Tensor direct, Tensor label_emb) -> Tensor'
    assert str(ext.ldam_bce.default._schema) == ('gzsl::ldam_bce(Tensor logits, Tensor y, Tensor? class_margin=None, '
                                                 'float C=0.) -> Tensor')


def test_extension_has_no_cpu_path(ext):
    with pytest.raises((RuntimeError, NotImplementedError)):
        ext.label_attention(torch.zeros(1, 4, 8), torch.zeros(1, 4, 8), torch.zeros(2, 8))
    with pytest.raises((RuntimeError, NotImplementedError)):
        ext.label_logits(torch.zeros(1, 2, 8), torch.zeros(2, 8))


def _reference_forward_with_ops(ext, case, g, dev):
    """ConvLabelAttnModel.forward (icd_modules.py:174-186) in eval mode, label embeddings taken from the golden."""
    P = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in case['params'].items()}
    x, y, mask = (torch.from_numpy(case[k]).to(dev) for k in ('x', 'y', 'mask'))
    L = y.shape[1]
    label_emb = torch.from_numpy(g['avg'][:L]).to(dev).requires_grad_(True)
    graph_label_emb = torch.from_numpy(g['graph'][:L]).to(dev).requires_grad_(True)
    e = torch.nn.functional.embedding(x, P['emb.embed.weight']) * mask.unsqueeze(2)          # :153-155
    h = torch.relu(torch.nn.functional.conv1d(e.transpose(1, 2), P['conv_modules.0.weight'],
                                              P['conv_modules.0.bias'])).transpose(1, 2)     # :131-132
    proj_x = torch.tanh(torch.nn.functional.linear(h, P['proj.weight'], P['proj.bias']))      # :134-135
    attn_output, _ = ext.label_attention(proj_x, h, label_emb)                                 # :137-142
    direct = torch.nn.functional.linear(attn_output, P['output_proj.weight'], P['output_proj.bias'])  # :159
    logits = ext.label_logits(direct, graph_label_emb)                                         # :177-182
    loss = ext.ldam_bce(logits, y, torch.from_numpy(g['class_margin']).to(dev), 2.0)           # :162-172
    return P, logits, loss, attn_output, direct


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['base_gate_tiny_eval', 'base_gate_medium_eval', 'base_gate_e200_eval'])
def test_reference_forward_with_the_operators_matches_reference_golden(ext, name):
    from tests.golden import synth
    g = np.load(os.path.join(GOLD, name + '.npz'))
    case = synth.synth_base(**synth.BASE_CASES[name[:-len('_eval')]])  # the seeded inputs the golden was captured on
    P, logits, loss, attn_output, direct = _reference_forward_with_ops(ext, case, g, 'cuda')
    close(attn_output, g['attn'], 1e-5, 1e-5, 'attention output')
    close(direct, g['direct'], 1e-5, 2e-5, 'direct feats')
    close(logits, g['logits'], 1e-5, 2e-5, 'logits')
    close(loss, g['loss'], 1e-5, 1e-4, 'loss')
    loss.backward()
    # the parameters below the label embeddings (taken as constants here) see the whole gradient
    for k in ('conv_modules.0.weight', 'conv_modules.0.bias', 'proj.weight', 'proj.bias', 'output_proj.weight',
              'output_proj.bias'):
        close(P[k].grad, g['grad.' + k], 1e-4, 4e-6, 'grad ' + k)


@pytest.mark.gpu
def test_label_attention_operator_agrees_with_the_ctypes_binding_and_the_oracle(ext):
    """Same kernels through both bindings: forward bit-identical, backward equal up to the order of the float atomics;
    dropout coordinates (seed, l_offset, L_total) as oracle/base_model.py:232 defines them."""
    from gzsl_text_amd.ops_hip import HipOps
    from oracle import base_model as om
    rs = np.random.RandomState(5)
    B, Tp, E, F, L, L_total, l_off, seed, p = 2, 150, 36, 52, 37, 90, 40, 77, 0.2
    PX = torch.tanh(torch.from_numpy(rs.standard_normal((B, Tp, E)).astype(np.float32)))
    X = torch.relu(torch.from_numpy(rs.standard_normal((B, Tp, F)).astype(np.float32)))
    Q = torch.from_numpy((rs.standard_normal((L, E)) * 0.3).astype(np.float32))
    dO = torch.from_numpy(rs.standard_normal((B, L, F)).astype(np.float32))
    a = [t.cuda().requires_grad_(True) for t in (PX, X, Q)]
    O, lse = ext.label_attention(a[0], a[1], a[2], p, seed, l_off, L_total)
    assert not lse.requires_grad
    O.backward(dO.cuda())
    ops = HipOps()
    O2, lse2 = ops.attn_fwd(PX.cuda(), X.cuda(), Q.cuda(), l_off, L_total, seed, p)
    assert torch.equal(O, O2) and torch.equal(lse, lse2)
    g2 = [torch.zeros_like(t).cuda() for t in (PX, X, Q)]
    ops.attn_bwd(dO.cuda(), O2, PX.cuda(), X.cuda(), Q.cuda(), lse2, l_off, L_total, seed, p, *g2)
    for t, r, what in zip(a, g2, ('dkeys', 'dvalues', 'dqueries')):
        close(t.grad, r, 1e-5, 1e-6, what + ' vs ctypes binding')
    Or, lser, cache = om.label_attn_fwd(PX.numpy().astype(np.float64), X.numpy().astype(np.float64),
                                        Q.numpy().astype(np.float64), drop_seed=seed, p_attn=p, l_offset=l_off,
                                        L_total=L_total)
    close(O, Or, 1e-5, 1e-5, 'O vs oracle')
    close(lse, lser, 1e-5, 1e-5, 'lse vs oracle')
    ref = om.label_attn_bwd(dO.numpy().astype(np.float64), PX.numpy().astype(np.float64), X.numpy().astype(np.float64),
                            Q.numpy().astype(np.float64), cache)
    for t, r, what in zip(a, ref, ('dkeys', 'dvalues', 'dqueries')):
        close(t.grad, r, 1e-4, 4e-6, what + ' vs oracle')


@pytest.mark.gpu
def test_operators_under_no_grad_and_with_strided_inputs(ext):
    rs = np.random.RandomState(6)
    PX = torch.from_numpy(rs.standard_normal((2, 70, 16)).astype(np.float32)).cuda()
    X = torch.from_numpy(rs.standard_normal((2, 16, 70)).astype(np.float32)).cuda().transpose(1, 2)  # not contiguous
    Q = torch.from_numpy(rs.standard_normal((9, 16)).astype(np.float32)).cuda()
    with torch.no_grad():
        O, lse = ext.label_attention(PX, X, Q)
    P = torch.softmax(torch.einsum('bte,le->blt', PX, Q), dim=2)
    close(O, P @ X, 1e-5, 1e-5, 'O (eval, strided values)')
    y = (torch.from_numpy(rs.uniform(size=(2, 9))) > 0.5).float().cuda()
    z = O[:, :, :9].sum(2)
    want = torch.nn.functional.binary_cross_entropy_with_logits(z, y, reduction='sum') / 2
    close(ext.ldam_bce(z, y), want, 1e-5, 1e-5, 'plain BCE (no margin)')
