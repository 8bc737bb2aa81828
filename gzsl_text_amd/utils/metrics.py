"""Device-side mirror of the thresholded half of `utils/metrics.py`.

The drivers copy `sigmoid(logits)` and `y` of every dev batch to the host (`train_base.py:129-136`), stack them into
`[n_dev, L]` matrices and hand them to `all_metrics` four times (all / frequent / few-shot / zero-shot label subsets,
`utils/helper.py:190-199`).  Everything in `all_metrics` except the three AUCs depends only on four integers per label:
how many notes have `yhat & y`, `yhat`, `y`, `yhat | y`, with `yhat = np.round(yscore)`.  `EvalCounters` accumulates
them on the device (`gzsl_eval_counts`, one launch per batch, no host sync) and evaluates the reference's formulas in
float64 from one `[4, L]` int32 download per epoch; label subsets index the counters.  The AUCs (`metrics.py:34,42-46`)
rank scores and stay on the host: `update(..., need_probs=True)` returns the scores for callers that want them.
"""
import numpy as np
import torch

from ..ops_hip import HipOps

NAMES = ('acc', 'pre', 'rec', 'f1')


def metrics_from_counts(counts):
    """counts [4, L] (yhat&y, yhat, y, yhat|y per label) -> the eight thresholded entries of `all_metrics`
    (metrics.py:28-41,63-111), computed in float64 exactly as numpy does there (including its 0/0 -> nan)."""
    tp, pp, pos, un = (np.asarray(c, dtype=np.float64) for c in counts)
    with np.errstate(divide='ignore', invalid='ignore'):
        mac_pre, mac_rec = np.mean(tp / (pp + 1e-10)), np.mean(tp / (pos + 1e-10))
        out = {'acc_macro': np.mean(tp / (un + 1e-10)), 'pre_macro': mac_pre, 'rec_macro': mac_rec,
               'f1_macro': 0. if mac_pre + mac_rec == 0 else 2 * (mac_pre * mac_rec) / (mac_pre + mac_rec)}
        TP, PP, POS, UN = tp.sum(), pp.sum(), pos.sum(), un.sum()
        mic_pre, mic_rec = TP / (PP + 1e-10), TP / POS
        out.update({'acc_micro': TP / UN, 'pre_micro': mic_pre, 'rec_micro': mic_rec,
                    'f1_micro': 0. if mic_pre + mic_rec == 0 else 2 * (mic_pre * mic_rec) / (mic_pre + mic_rec)})
    return out


class EvalCounters:
    """Per-label confusion counters over the first `code_size` label columns, accumulated on the device."""

    def __init__(self, code_size, device='cuda', ops=None):
        device = torch.device(device)
        if ops is None:
            if device.type != 'cuda':
                raise RuntimeError('gzsl_text_amd kernels need a HIP device; there is no CPU path')
            ops = HipOps()
        self.ops, self.code_size = ops, int(code_size)
        self.counts = torch.zeros(4, self.code_size, dtype=torch.int32, device=device)
        self.n_rows = 0

    def reset(self):
        self.counts.zero_()
        self.n_rows = 0

    def update(self, logits, y, need_probs=False):
        """logits, y: [B, >= code_size] fp32 on the device.  Returns sigmoid(logits[:, :code_size]) if asked."""
        if logits.shape[0] != y.shape[0]:
            raise ValueError(f'logits {tuple(logits.shape)} and y {tuple(y.shape)} disagree on the batch size')
        probs = None
        if need_probs:
            probs = torch.empty(logits.shape[0], self.code_size, dtype=torch.float32, device=logits.device)
        self.ops.eval_counts_(logits, y, self.counts, probs)
        self.n_rows += logits.shape[0]
        return probs

    def host_counts(self):
        return self.counts.cpu().numpy().astype(np.int64)

    def metrics(self, indices=None):
        """`all_metrics(y_score[:, indices], y_true[:, indices])` without the AUC entries."""
        c = self.host_counts()
        if indices is not None:
            c = c[:, np.asarray(indices)]
        return metrics_from_counts(c)
