"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the batching and eval-scoring code either side of the train
step.  Imported by tests/ only; the product never touches it.  Pinned by
tests/golden/eval_path.npz, produced by the reference's own `iterate_minibatch` and `all_macro` / `all_micro`
(tests/golden/make_golden_eval.py).
"""
import numpy as np


def multihot(label_lists, eval_code_size):
    """utils/helper.py:117-125: one row per note, 1.0 at each of its label indices, fp32."""
    y = np.zeros((len(label_lists), eval_code_size), dtype=np.float64)
    for row, c_idx in enumerate(label_lists):
        y[row, np.asarray(c_idx, dtype=np.int64)] = 1
    return y.astype(np.float32)


def minibatches(notes, labels, masks, eval_code_size, batch_size, shuffle):
    """utils/helper.py:108-132: the batch order (np.random.shuffle of arange(n)), ceil(n / batch_size) batches."""
    n = len(notes)
    order = np.arange(n)
    if shuffle:
        np.random.shuffle(order)
    for i in range(n // batch_size + (n % batch_size != 0)):
        pick = order[i * batch_size:(i + 1) * batch_size]
        yield notes[pick], multihot([labels[b] for b in pick], eval_code_size), masks[pick], pick


def sigmoid(z):
    """train_base.py:133 torch.sigmoid in fp32."""
    z = np.asarray(z, dtype=np.float32)
    with np.errstate(over='ignore'):  # exp(120) = inf in fp32 -> sigmoid 0, as torch gives
        return (1.0 / (1.0 + np.exp(-z, dtype=np.float32))).astype(np.float32)


def confusion_counts(yscore, y):
    """metrics.py:29 yhat = np.round(yscore) (half-to-even), then the four per-label sums the thresholded metrics use
    (intersect_size / union_size with axis=0, metrics.py:114-121; yhat.sum(0), y.sum(0) at :69,74)."""
    yhat = np.round(yscore)
    return np.stack([np.logical_and(yhat, y).sum(0), (yhat != 0).sum(0), (np.asarray(y) != 0).sum(0),
                     np.logical_or(yhat, y).sum(0)]).astype(np.int64)


def thresholded_metrics(yscore, y):
    """The acc/pre/rec/f1 entries of all_metrics (metrics.py:28-41), dense, as the reference evaluates them."""
    yhat = np.round(yscore)
    inter0 = np.logical_and(yhat, y).sum(0).astype(float)
    union0 = np.logical_or(yhat, y).sum(0).astype(float)
    with np.errstate(divide='ignore', invalid='ignore'):
        mp, mr = np.mean(inter0 / (yhat.sum(0) + 1e-10)), np.mean(inter0 / (y.sum(0) + 1e-10))
        out = {'acc_macro': np.mean(inter0 / (union0 + 1e-10)), 'pre_macro': mp, 'rec_macro': mr,
               'f1_macro': 0. if mp + mr == 0 else 2 * (mp * mr) / (mp + mr)}
        yh, yy = yhat.ravel(), np.asarray(y).ravel()
        inter = np.float64(np.logical_and(yh, yy).sum())
        up, ur = inter / (yh.sum() + 1e-10), inter / yy.sum()
        out.update({'acc_micro': inter / np.float64(np.logical_or(yh, yy).sum()), 'pre_micro': up, 'rec_micro': ur,
                    'f1_micro': 0. if up + ur == 0 else 2 * (up * ur) / (up + ur)})
    return out
