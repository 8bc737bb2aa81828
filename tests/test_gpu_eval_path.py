"""SURVEY 8f-3 / 8f-4 on the GPU, through the C ABI: device multi-hot targets and per-label eval counters against the
reference-generated goldens (tests/golden/eval_path.npz), the oracle, and full-size properties."""
import os

import numpy as np
import pytest
import torch

from oracle import eval_path as oe

pytestmark = pytest.mark.gpu
DEV = 'cuda'
G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'eval_path.npz'))
ORDER = ('acc_macro', 'pre_macro', 'rec_macro', 'f1_macro', 'acc_micro', 'pre_micro', 'rec_micro', 'f1_micro')


def label_lists():
    flat, offs = G['labels_flat'], G['labels_offsets']
    return [flat[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)]


@pytest.mark.parametrize('tag,shuffle', [('ordered', False), ('shuffled', True)])
def test_iterate_minibatch_device_matches_reference(tag, shuffle):
    from gzsl_text_amd.utils.helper import iterate_minibatch
    np.random.seed(5)
    xs, ys, ms = [], [], []
    for x, y, m, yi in iterate_minibatch(G['notes'], label_lists(), G['masks'], int(G['L']), batch_size=int(G['B']),
                                         shuffle=shuffle, device=DEV):
        assert x.is_cuda and y.is_cuda and m.is_cuda and yi.is_cuda and y.dtype == torch.float32
        xs.append(x.cpu()), ys.append(y.cpu()), ms.append(m.cpu())  # consumed while the next batch uploads
    assert [x.shape[0] for x in xs] == G[f'{tag}_sizes'].tolist()
    assert torch.equal(torch.cat(xs), torch.from_numpy(G[f'{tag}_x']))
    assert torch.equal(torch.cat(ys), torch.from_numpy(G[f'{tag}_y']))   # bit-exact
    assert torch.equal(torch.cat(ms), torch.from_numpy(G[f'{tag}_mask']))


def test_multihot_full_size_and_edges():
    from gzsl_text_amd.ops_hip import HipOps
    ops = HipOps()
    rs = np.random.RandomState(3)
    B, L = 16, 9000
    lists = [rs.choice(L, size=rs.randint(0, 40), replace=True) for _ in range(B)]  # duplicates and one-or-more empties
    lists[5] = np.zeros(0, dtype=np.int64)
    W = max(len(c) for c in lists)
    idx = np.full((B, W), -1, dtype=np.int64)
    for r, c in enumerate(lists):
        idx[r, :len(c)] = c
    y = ops.multihot(torch.from_numpy(idx).to(DEV), L)
    assert np.array_equal(y.cpu().numpy(), oe.multihot(lists, L))
    assert float(y.sum()) == sum(len(set(c.tolist())) for c in lists)
    # out-of-range entries are skipped by the kernel (the host mirror raises before it gets there); width-0 lists
    idx[0, 0], idx[1, 0] = L, -7
    y2 = ops.multihot(torch.from_numpy(idx).to(DEV), L).cpu().numpy()
    lists[0], lists[1] = lists[0][1:], lists[1][1:]
    assert np.array_equal(y2, oe.multihot(lists, L))
    assert float(ops.multihot(torch.zeros(3, 0, dtype=torch.int64, device=DEV), 7).abs().sum()) == 0


@pytest.mark.parametrize('name', ['all', 'few', 'zero'])
def test_eval_counters_match_reference(name):
    from gzsl_text_amd.utils.metrics import EvalCounters
    logits, y = torch.from_numpy(G['eval_logits']).to(DEV), torch.from_numpy(G['eval_y']).to(DEV)
    ec = EvalCounters(logits.shape[1], device=DEV)
    scores = [ec.update(logits[i:i + 8], y[i:i + 8], need_probs=True) for i in range(0, logits.shape[0], 8)]
    scores = torch.cat(scores).cpu().numpy()
    assert np.allclose(scores, G['eval_yscore'], rtol=2e-6, atol=1e-7)     # fp32 sigmoid, tolerance as north_star
    assert np.array_equal(np.round(scores), np.round(G['eval_yscore']))
    assert np.array_equal(ec.host_counts(), oe.confusion_counts(G['eval_yscore'], G['eval_y']))  # integers: exact
    got = ec.metrics(G[f'eval_{name}_idx'])
    assert np.allclose([got[k] for k in ORDER], G[f'eval_{name}_metrics'], rtol=1e-9, atol=0, equal_nan=True)


def test_eval_counts_full_size_strided_and_additive():
    """B=32, L=9000 with the dev label prefix (train_base.py:129,133 slices [:, :dev_eval_code_size]); counts of two
    halves add up to the counts of the whole (the counters only ever accumulate)."""
    from gzsl_text_amd.ops_hip import HipOps
    ops = HipOps()
    rs = np.random.RandomState(8)
    B, L, Lc = 32, 9000, 8123
    logits = (rs.standard_normal((B, L)) * 3).astype(np.float32)
    logits[rs.rand(B, L) < 0.02] = 0.0
    y = (rs.rand(B, L) < 0.01).astype(np.float32)
    ld, yd = torch.from_numpy(logits).to(DEV), torch.from_numpy(y).to(DEV)
    whole = torch.zeros(4, Lc, dtype=torch.int32, device=DEV)
    probs = torch.empty(B, Lc, device=DEV)
    ops.eval_counts_(ld, yd, whole, probs)
    want = oe.confusion_counts(oe.sigmoid(logits[:, :Lc]), y[:, :Lc])
    assert np.array_equal(whole.cpu().numpy(), want)
    assert np.allclose(probs.cpu().numpy(), oe.sigmoid(logits[:, :Lc]), rtol=2e-6, atol=1e-7)
    halves = torch.zeros(4, Lc, dtype=torch.int32, device=DEV)
    ops.eval_counts_(ld[:16], yd[:16], halves)
    ops.eval_counts_(ld[16:], yd[16:], halves)
    assert torch.equal(halves, whole)
    c = whole.cpu().numpy().astype(np.int64)   # |A|+|B| = |A&B| + |A|B| per label
    assert np.array_equal(c[1] + c[2], c[0] + c[3])
