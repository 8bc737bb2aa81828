"""SURVEY 8f-3 / 8f-4 on the CPU: the oracle against the reference-generated goldens, and the host mirrors
(`gzsl_text_amd.utils.helper.iterate_minibatch`, `gzsl_text_amd.utils.metrics.EvalCounters`) through the test stand-in."""
import os

import numpy as np
import pytest
import torch

from oracle import eval_path as oe
from tests.ops_cpu import CpuOps

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'eval_path.npz'))
# rtol 1e-9: the reference adds its 1e-10 guards partly in float32 (numpy promotion), the restatements in float64
ORDER = ('acc_macro', 'pre_macro', 'rec_macro', 'f1_macro', 'acc_micro', 'pre_micro', 'rec_micro', 'f1_micro')


def label_lists():
    flat, offs = G['labels_flat'], G['labels_offsets']
    return [flat[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)]


def as_vec(m):
    return np.asarray([m[k] for k in ORDER])


@pytest.mark.parametrize('tag,shuffle', [('ordered', False), ('shuffled', True)])
def test_oracle_minibatches_match_reference(tag, shuffle):
    np.random.seed(5)
    got = list(oe.minibatches(G['notes'], label_lists(), G['masks'], int(G['L']), int(G['B']), shuffle))
    assert [len(b[3]) for b in got] == G[f'{tag}_sizes'].tolist()
    assert np.array_equal(np.concatenate([b[3] for b in got]), G[f'{tag}_pick'])
    assert np.array_equal(np.concatenate([b[0] for b in got]), G[f'{tag}_x'])
    y = np.concatenate([b[1] for b in got])
    assert y.dtype == np.float32 and np.array_equal(y, G[f'{tag}_y'])
    assert np.array_equal(np.concatenate([b[2] for b in got]), G[f'{tag}_mask'])


@pytest.mark.parametrize('name', ['all', 'few', 'zero'])
def test_oracle_metrics_match_reference(name):
    idx = G[f'eval_{name}_idx']
    yscore = oe.sigmoid(G['eval_logits'])
    assert np.allclose(yscore, G['eval_yscore'], rtol=1e-6, atol=1e-7)
    assert np.array_equal(np.round(yscore), np.round(G['eval_yscore']))
    dense = as_vec(oe.thresholded_metrics(yscore[:, idx], G['eval_y'][:, idx]))
    assert np.allclose(dense, G[f'eval_{name}_metrics'], rtol=1e-9, atol=0, equal_nan=True)
    from gzsl_text_amd.utils.metrics import metrics_from_counts
    counts = oe.confusion_counts(yscore, G['eval_y'])[:, idx]
    assert np.allclose(as_vec(metrics_from_counts(counts)), G[f'eval_{name}_metrics'], rtol=1e-9, atol=0, equal_nan=True)


@pytest.mark.parametrize('tag,shuffle', [('ordered', False), ('shuffled', True)])
def test_host_iterate_minibatch_matches_reference(tag, shuffle):
    from gzsl_text_amd.utils.helper import iterate_minibatch
    np.random.seed(5)
    got = list(iterate_minibatch(G['notes'], label_lists(), G['masks'], int(G['L']), batch_size=int(G['B']),
                                 shuffle=shuffle, device='cpu', ops=CpuOps()))
    assert [b[0].shape[0] for b in got] == G[f'{tag}_sizes'].tolist()
    assert torch.equal(torch.cat([b[0] for b in got]), torch.from_numpy(G[f'{tag}_x']))
    y = torch.cat([b[1] for b in got])
    assert y.dtype == torch.float32 and torch.equal(y, torch.from_numpy(G[f'{tag}_y']))
    assert torch.equal(torch.cat([b[2] for b in got]), torch.from_numpy(G[f'{tag}_mask']))
    lists, row = label_lists(), 0
    for b in got:  # y_indices: the reference's -1 padded lists, cut to the widest row of the batch
        for r in b[3]:
            want = lists[G[f'{tag}_pick'][row]]
            assert r[:len(want)].tolist() == want.tolist() and bool((r[len(want):] == -1).all())
            row += 1
    np.random.seed(5)
    picks = [b[3] for b in iterate_minibatch(G['notes'], label_lists(), G['masks'], int(G['L']), batch_size=int(G['B']),
                                             shuffle=shuffle, rtn_indices=True, device='cpu', ops=CpuOps())]
    assert np.array_equal(np.concatenate(picks), G[f'{tag}_pick'])


def test_host_iterate_minibatch_errors_and_edges():
    from gzsl_text_amd.utils.helper import iterate_minibatch
    notes, masks = G['notes'][:4], G['masks'][:4]
    with pytest.raises(IndexError):
        list(iterate_minibatch(notes, [[1], [2], [50], [3]], masks, 50, batch_size=2, shuffle=False, device='cpu', ops=CpuOps()))
    with pytest.raises(AssertionError):
        list(iterate_minibatch(notes, [[1]], masks, 50, device='cpu', ops=CpuOps()))
    with pytest.raises(RuntimeError):  # the product has no CPU path
        list(iterate_minibatch(notes, [[1], [2], [3], [4]], masks, 50, device='cpu'))
    assert list(iterate_minibatch(notes[:0], [], masks[:0], 50, device='cpu', ops=CpuOps())) == []
    (x, y, m, yi), = list(iterate_minibatch(notes, [[], [-1], [0, 0], [49]], masks, 50, batch_size=8, shuffle=False,
                                            device='cpu', ops=CpuOps()))  # empty list, numpy's negative wrap, duplicates
    assert y.sum(1).tolist() == [0, 1, 1, 1] and y[1, 49] == 1 and y[2, 0] == 1
    assert np.array_equal(y.numpy(), oe.multihot([[], [-1], [0, 0], [49]], 50))


@pytest.mark.parametrize('name', ['all', 'few', 'zero'])
def test_host_eval_counters_match_reference(name):
    from gzsl_text_amd.utils.metrics import EvalCounters
    logits, y = torch.from_numpy(G['eval_logits']), torch.from_numpy(G['eval_y'])
    ec = EvalCounters(logits.shape[1], device='cpu', ops=CpuOps())
    scores = [ec.update(logits[i:i + 8], y[i:i + 8], need_probs=True) for i in range(0, logits.shape[0], 8)]
    assert ec.n_rows == logits.shape[0]
    assert np.allclose(torch.cat(scores).numpy(), G['eval_yscore'], rtol=1e-6, atol=1e-7)
    assert np.array_equal(ec.host_counts(), oe.confusion_counts(G['eval_yscore'], G['eval_y']))
    assert np.allclose(as_vec(ec.metrics(G[f'eval_{name}_idx'])), G[f'eval_{name}_metrics'], rtol=1e-9, atol=0,
                       equal_nan=True)
    with pytest.raises(RuntimeError):
        EvalCounters(5, device='cpu')
