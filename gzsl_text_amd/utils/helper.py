"""Device-side mirror of `utils/helper.py:106-132` (`iterate_minibatch`).

The reference builds, on the host and for every step, a dense float64 `[B, eval_code_size]` multi-hot matrix, casts it
to fp32 and ships it over PCIe (576 KB at B=16, L=9000) next to a `[B, eval_code_size]` int64 index matrix (1.1 MB) that
no driver ever reads.  Here the host only gathers the notes into pinned memory and pads the batch's ragged label lists
to the widest row (a few hundred bytes); the multi-hot targets are built on the device by `gzsl_multihot`, all uploads
run on a copy stream one batch ahead of the consumer, and the yielded tensors are already on the device -- the drivers'
`.to(device)` calls (`train_base.py:105,128`) become no-ops.

Same generator contract: same shuffling call (`np.random.shuffle` on `arange(n)`, so a seeded run visits the same
batches), same number of batches, same `(x, y, mask, y_indices | batch_indices)` tuples; `y` is bit-identical to the
reference's.  One documented difference: `y_indices` is `[B, W]` (W = longest label list of the batch, -1 padded)
instead of `[B, eval_code_size]`.
"""
import numpy as np
import torch

from ..ops_hip import HipOps


def _pad_label_lists(train_labels, batch_indices, eval_code_size):
    """The batch's ragged label lists as two [B, W] int64 matrices padded with -1: the lists as given (what the
    reference puts in y_indices) and with numpy's negative-index wrap applied (what the scatter kernel consumes)."""
    lists = [train_labels[b] for b in batch_indices]
    lens = np.fromiter((len(c) for c in lists), dtype=np.int64, count=len(lists))
    width = max(1, int(lens.max())) if len(lists) else 1
    flat = np.concatenate([np.asarray(c, dtype=np.int64).reshape(-1) for c in lists]) if lens.sum() else \
        np.zeros(0, dtype=np.int64)
    bad = (flat >= eval_code_size) | (flat < -eval_code_size)  # what `y[row, c_idx] = 1` raises in numpy
    if bad.any():
        raise IndexError(f'index {flat[bad][0]} is out of bounds for axis 1 with size {eval_code_size}')
    y_indices = np.full((len(lists), width), -1, dtype=np.int64)
    scatter = np.full((len(lists), width), -1, dtype=np.int64)
    fill = np.arange(width)[None, :] < lens[:, None]
    y_indices[fill] = flat
    scatter[fill] = np.where(flat < 0, flat + eval_code_size, flat)
    return y_indices, scatter


class _Stager:
    """Two pinned staging slots + a copy stream: batch k+1 is gathered and uploaded while the consumer runs batch k."""

    def __init__(self, device, ops):
        self.device, self.ops = device, ops
        self.async_ = device.type == 'cuda'
        self.stream = torch.cuda.Stream(device) if self.async_ else None
        self.slots = [{}, {}]
        self.k = 0

    def _pinned(self, slot, name, shape, dtype):
        t = slot.get(name)
        if t is None or t.dtype != dtype or t.numel() < int(np.prod(shape)):
            t = torch.empty(int(np.prod(shape)), dtype=dtype, pin_memory=self.async_)
            slot[name] = t
        return t[:int(np.prod(shape))].view(*shape)

    def issue(self, notes, masks, batch_indices, y_indices, scatter, eval_code_size):
        slot = self.slots[self.k & 1]
        self.k += 1
        if slot.get('event') is not None:
            slot['event'].synchronize()  # the upload that last used these pinned buffers has finished
        host = {}
        for name, src in (('x', notes), ('mask', masks)):
            shape = (len(batch_indices),) + tuple(src.shape[1:])
            buf = self._pinned(slot, name, shape, torch.from_numpy(src[:0]).dtype)
            np.take(src, batch_indices, axis=0, out=buf.numpy())
            host[name] = buf
        for name, arr in (('y_indices', y_indices), ('scatter', scatter)):
            buf = self._pinned(slot, name, arr.shape, torch.int64)
            buf.numpy()[...] = arr
            host[name] = buf
        if not self.async_:
            dev = {k: v.clone() for k, v in host.items()}
            dev['y'] = self.ops.multihot(dev['scatter'], eval_code_size)
            return dev, None
        with torch.cuda.stream(self.stream):
            dev = {k: v.to(self.device, non_blocking=True) for k, v in host.items()}
            dev['y'] = self.ops.multihot(dev['scatter'], eval_code_size)
            event = torch.cuda.Event()
            event.record(self.stream)
        slot['event'] = event
        return dev, event

    def hand_over(self, dev, event):
        if event is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(event)
            for t in dev.values():
                t.record_stream(cur)  # allocated on the copy stream, consumed on the caller's
        return dev


def iterate_minibatch(train_notes, train_labels, train_masks, eval_code_size, batch_size=16, shuffle=True,
                      rtn_indices=False, device='cuda', ops=None):
    """utils/helper.py:106-132 with the targets built on `device`.  Yields device tensors."""
    assert len(train_notes) == len(train_labels)
    device = torch.device(device)
    if ops is None:
        if device.type != 'cuda':
            raise RuntimeError('gzsl_text_amd kernels need a HIP device; there is no CPU path')
        ops = HipOps()
    n = len(train_notes)
    indices = np.arange(n)
    if shuffle:
        np.random.shuffle(indices)
    num_batches = n // batch_size + (n % batch_size != 0)
    stager = _Stager(device, ops)

    def issue(i):
        batch_indices = indices[i * batch_size: (i + 1) * batch_size]
        y_indices, scatter = _pad_label_lists(train_labels, batch_indices, eval_code_size)
        return batch_indices, stager.issue(train_notes, train_masks, batch_indices, y_indices, scatter, eval_code_size)

    ahead = issue(0) if num_batches else None
    for i in range(num_batches):
        batch_indices, (dev, event) = ahead
        ahead = issue(i + 1) if i + 1 < num_batches else None
        dev = stager.hand_over(dev, event)
        if rtn_indices:
            yield dev['x'], dev['y'], dev['mask'], batch_indices
        else:
            yield dev['x'], dev['y'], dev['mask'], dev['y_indices']
