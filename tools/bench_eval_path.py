"""Timing of the rows either side of the step at cfg2 sizes: B=16 train batches of T=2500 tokens,
L=9000 labels, 15 labels per note; eval batches of 32.

    python tools/bench_eval_path.py

Prints one JSON line: per-launch time and algorithmic GB/s of `gzsl_multihot` and `gzsl_eval_counts` (HBM-bound; at
these sizes they move < 3 MB and are launch-latency-bound), batches/s of the device iterator against the CPU
restatement of the reference's host batching (oracle, + the H2D copies the drivers do), and the per-epoch cost of the
metrics from counters against the dense host formulas on n_dev=1600 notes.
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gzsl_text_amd.ops_hip import HipOps  # noqa: E402
from gzsl_text_amd.utils.helper import iterate_minibatch  # noqa: E402
from gzsl_text_amd.utils.metrics import EvalCounters  # noqa: E402
from oracle import eval_path as oe  # noqa: E402  (baseline leg only)


def per_launch(fn, n=300):
    for _ in range(20):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3


def main():
    dev = torch.device('cuda', 0)
    ops = HipOps()
    rs = np.random.RandomState(0)
    n, T, L, B, W = 640, 2500, 9000, 16, 15
    notes = rs.randint(1, 50000, size=(n, T)).astype(np.int64)
    masks = np.ones((n, T), dtype=np.float32)
    labels = [rs.choice(L, size=W, replace=False) for _ in range(n)]
    idx = torch.from_numpy(np.stack(labels[:B])).to(dev)
    t_mh = per_launch(lambda: ops.multihot(idx, L))
    Be = 32
    logits, y = torch.randn(Be, L, device=dev), (torch.rand(Be, L, device=dev) < 0.002).float()
    counts, probs = torch.zeros(4, L, dtype=torch.int32, device=dev), torch.empty(Be, L, device=dev)
    t_ec = per_launch(lambda: ops.eval_counts_(logits, y, counts, probs))
    out = {'multihot_us': t_mh * 1e6, 'multihot_GBps': (B * L * 4 + B * W * 8) / t_mh / 1e9,
           'eval_counts_us': t_ec * 1e6, 'eval_counts_GBps': (Be * L * 12 + 4 * L * 8) / t_ec / 1e9}

    def drain_device():
        k = 0
        for x, yy, m, yi in iterate_minibatch(notes, labels, masks, L, batch_size=B, shuffle=True, device=dev):
            k += 1
        torch.cuda.synchronize()
        return k

    def drain_host():
        k = 0
        for x, yy, m, pick in oe.minibatches(notes, labels, masks, L, B, True):
            # what train_base.py:105 adds: y_indices as the reference shapes it, then four H2D copies
            yi = np.zeros((len(pick), L), dtype=int) - 1
            for t in (x, yy, m, yi):
                torch.from_numpy(t).to(dev)
            k += 1
        torch.cuda.synchronize()
        return k

    for name, fn in (('device_iter', drain_device), ('host_iter', drain_host)):
        fn()
        t0 = time.perf_counter()
        k = fn()
        out[name + '_batches_per_s'] = k / (time.perf_counter() - t0)

    n_dev = 1600
    zl, zy = torch.randn(n_dev, L), (torch.rand(n_dev, L) < 0.002).float()
    ec = EvalCounters(L, device=dev)
    zd, yd = zl.to(dev), zy.to(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(0, n_dev, Be):
        ec.update(zd[i:i + Be], yd[i:i + Be])
    m_dev = ec.metrics()
    out['counters_epoch_ms'] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    score = np.vstack([torch.sigmoid(zd[i:i + Be]).cpu().numpy() for i in range(0, n_dev, Be)])
    true = np.vstack([yd[i:i + Be].cpu().numpy() for i in range(0, n_dev, Be)])
    m_host = oe.thresholded_metrics(score, true)
    out['dense_host_epoch_ms'] = (time.perf_counter() - t0) * 1e3
    assert all(abs(m_dev[k] - m_host[k]) <= 1e-9 * abs(m_host[k]) for k in m_host), (m_dev, m_host)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
