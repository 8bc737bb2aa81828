"""Golden vectors for the batching / eval-scoring rows, from the reference's own
`utils/helper.py:iterate_minibatch` and `utils/metrics.py:all_macro/all_micro` (runs ONLY in the build container).

    python tests/golden/make_golden_eval.py        # writes tests/golden/eval_path.npz

Inputs are seeded synthetic data (ragged label lists, random logits with exact zeros and saturated values mixed in);
the fixture holds inputs and the reference's outputs, nothing else.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, '/root/reference')

from utils.helper import iterate_minibatch  # noqa: E402  (reference)
from utils.metrics import all_macro, all_micro  # noqa: E402


def main():
    rs = np.random.RandomState(11)
    n, T, L, B = 23, 17, 50, 5
    notes = rs.randint(0, 99, size=(n, T)).astype(np.int64)
    masks = (rs.rand(n, T) > 0.2).astype(np.float32)
    widths = rs.randint(1, 9, size=n)
    widths[3] = 1
    flat = np.concatenate([rs.choice(L, size=w, replace=False) for w in widths]).astype(np.int64)
    offs = np.concatenate([[0], np.cumsum(widths)]).astype(np.int64)
    labels = [flat[offs[i]:offs[i + 1]] for i in range(n)]
    out = {'notes': notes, 'masks': masks, 'labels_flat': flat, 'labels_offsets': offs, 'L': np.int64(L), 'B': np.int64(B)}

    for tag, shuffle in (('ordered', False), ('shuffled', True)):
        np.random.seed(5)
        xs, ys, ms, picks = [], [], [], []
        for x, y, m, pick in iterate_minibatch(notes, labels, masks, L, batch_size=B, shuffle=shuffle, rtn_indices=True):
            xs.append(x.numpy()), ys.append(y.numpy()), ms.append(m.numpy()), picks.append(pick)
        out[f'{tag}_x'], out[f'{tag}_y'], out[f'{tag}_mask'] = np.concatenate(xs), np.concatenate(ys), np.concatenate(ms)
        out[f'{tag}_pick'] = np.concatenate(picks)
        out[f'{tag}_sizes'] = np.asarray([len(p) for p in picks], dtype=np.int64)
        assert ys[0].dtype == np.float32

    # eval scoring: logits -> sigmoid in fp32 (train_base.py:133) -> metrics; label columns with no positives, no
    # predictions, exact-zero logits (sigmoid == 0.5 rounds to 0) and saturated logits are all present
    import torch
    n_dev, Lc = 37, 41
    logits = (rs.standard_normal((n_dev, Lc)) * 2).astype(np.float32)
    logits[rs.rand(n_dev, Lc) < 0.05] = 0.0
    logits[rs.rand(n_dev, Lc) < 0.03] = 40.0
    logits[rs.rand(n_dev, Lc) < 0.03] = -120.0
    logits[:, 7] = -3.0          # never predicted
    y = (rs.rand(n_dev, Lc) < 0.25).astype(np.float32)
    y[:, 11] = 0                 # no positives
    yscore = torch.sigmoid(torch.from_numpy(logits)).numpy()
    out['eval_logits'], out['eval_y'], out['eval_yscore'] = logits, y, yscore
    subsets = {'all': np.arange(Lc), 'few': np.asarray([1, 7, 11, 12, 30]), 'zero': np.asarray([40, 2])}
    for name, idx in subsets.items():
        yhat = np.round(yscore[:, idx])
        mac = all_macro(yhat, y[:, idx])
        mic = all_micro(yhat.ravel(), y[:, idx].ravel())
        out[f'eval_{name}_idx'] = idx.astype(np.int64)
        out[f'eval_{name}_metrics'] = np.asarray(list(mac) + list(mic), dtype=np.float64)  # acc,pre,rec,f1 macro then micro
    np.savez_compressed(os.path.join(HERE, 'eval_path.npz'), **out)
    print('wrote eval_path.npz', {k: out[k] for k in out if k.endswith('_metrics')})


if __name__ == '__main__':
    main()
