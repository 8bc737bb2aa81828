"""Label-dimension sharding of the base model over one process per GPU (RCCL through torch.distributed).

The reference is single-device; this is what the build adds (8e).  Labels are independent given the
encoder output (softmax is over tokens, not labels), so rank r owns a contiguous range of the attention query rows
(average label embedding), classifier rows (graph label embedding / output_fc), margins and y columns:

  forward   each rank encodes B/P notes ............. all_gather  X||PX      [B,T',F+E]   (mesh over xGMI)
            fused attention + head on its L/P labels
            logits shards + loss partials ........... all_gather  [P, B+1, ceil(L/P)]  (one collective)
  backward  partial dX||dPX from each label shard ... reduce_scatter over the batch
            each rank backprops its B/P notes; label-side gradients are backpropagated from the rank's zero-padded
            d(avg)/d(graph) through its replica (the encoders are linear in the incoming gradient)
            every shared-parameter gradient ......... all_reduce  one flat bucket

When B is not divisible by P the encoder is replicated instead (every rank encodes all notes; dX is all-reduced).
All ranks must hold identical parameters, draw the same batch and use the same dropout seed; the dropout hash is
keyed on global (note, label, token) coordinates, so results do not depend on P.
"""
import os
from collections import defaultdict

import torch
import torch.distributed as dist


def partition(n, world, rank):
    """Balanced contiguous partition of range(n): returns (begin, end) of `rank`."""
    base, rem = divmod(n, world)
    beg = rank * base + min(rank, rem)
    return beg, beg + base + (1 if rank < rem else 0)


class LabelShard:
    def __init__(self, group=None, own_first=None):
        if not dist.is_initialized():
            raise RuntimeError('LabelShard needs torch.distributed to be initialised (one process per GPU)')
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        # own-notes-first ordering (batch-split encoder only): the attention of the rank's OWN notes needs no
        # communication, so it runs while the X / PX all-gather is on the wire (forward), and the other ranks' notes go
        # first in backward so that the reduce-scatter starts before the own block is done.  Costs extra launches
        # (DESIGN.md 6); off by default, GZSL_OWN_FIRST=1 or own_first=True switches it on.
        self.own_first = (os.environ.get('GZSL_OWN_FIRST') == '1') if own_first is None else bool(own_first)
        self.prof = None  # comm profile: tag -> list of (event before, event after, bytes); see start_profile()
        # RCCL ('nccl') moves device tensors directly.  Under gloo (CPU tests; rehearsals with ranks sharing a GPU)
        # device tensors are staged through host memory: gloo's own device path stalls on multi-MB tensors.
        self.stage = dist.get_backend(group) != 'nccl'
        self._cols = {}  # (L, P, device) -> column index that drops the padding of an uneven label partition

    def _run(self, fn, out, inp):
        if self.stage and inp.is_cuda:
            o, i = out.cpu() if out is inp else torch.empty(out.shape, dtype=out.dtype), inp.cpu()
            if out is inp:
                fn(o, o)
            else:
                fn(o, i)
            out.copy_(o)
        else:
            fn(out, inp)
        return out

    # ---- communication profile (bench.py): how long the launching stream sat blocked on each collective -------------
    def start_profile(self):
        self.prof = defaultdict(list)

    def _timed(self, tag, nbytes, fn):
        """Brackets `fn` (a blocking wait on, or a synchronous call of, a collective) with events on the current stream:
        their distance is the time that stream was held up by the collective, i.e. the EXPOSED communication."""
        if self.prof is None or not torch.cuda.is_available():
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        self.prof[tag].append((e0, e1, int(nbytes)))
        return r

    def wait(self, handle, tag, tensor):
        return self._timed(tag, tensor.numel() * tensor.element_size(), handle.wait)

    def profile_summary(self, steps):
        """tag -> {exposed_ms_per_step, calls_per_step, bytes_per_call}; call after a device synchronize."""
        out = {}
        for tag, ev in (self.prof or {}).items():
            ms = sum(a.elapsed_time(b) for a, b, _ in ev)
            out[tag] = {'exposed_ms_per_step': ms / max(steps, 1), 'calls_per_step': len(ev) / max(steps, 1),
                        'bytes_per_call': ev[0][2] if ev else 0}
        return out

    def label_range(self, L):
        return partition(L, self.world, self.rank)

    def batch_split(self, B):
        """(b0, b1) of this rank when the batch divides evenly, else None (replicated encoder)."""
        if B % self.world != 0:
            return None
        return partition(B, self.world, self.rank)

    # ---- collectives (tensors are contiguous) ---------------------------------------------------------------
    def all_gather_batch(self, local):
        """[B/P, ...] on every rank -> [B, ...]."""
        out = local.new_empty((local.shape[0] * self.world,) + tuple(local.shape[1:]))
        return self._run(lambda o, i: dist.all_gather_into_tensor(o, i, group=self.group), out, local.contiguous())

    def reduce_scatter_batch(self, full):
        """[B, ...] partial sums on every rank -> this rank's summed [B/P, ...]."""
        out = full.new_empty((full.shape[0] // self.world,) + tuple(full.shape[1:]))
        return self._run(lambda o, i: dist.reduce_scatter_tensor(o, i, op=dist.ReduceOp.SUM, group=self.group), out,
                         full.contiguous())

    # ---- the same two, started asynchronously (RCCL runs them on its own stream): the caller does independent work
    # (label encoder forward / backward) and then calls .wait() on the handle, which orders the current stream after
    # the collective.  Under gloo they run synchronously and the handle is a no-op.
    class _Done:
        def wait(self):
            return None

    def all_gather_batch_start(self, local):
        out = local.new_empty((local.shape[0] * self.world,) + tuple(local.shape[1:]))
        local = local.contiguous()
        if self.stage:
            self._run(lambda o, i: dist.all_gather_into_tensor(o, i, group=self.group), out, local)
            return out, LabelShard._Done()
        return out, dist.all_gather_into_tensor(out, local, group=self.group, async_op=True)

    def reduce_scatter_batch_start(self, full):
        out = full.new_empty((full.shape[0] // self.world,) + tuple(full.shape[1:]))
        full = full.contiguous()
        if self.stage:
            self._run(lambda o, i: dist.reduce_scatter_tensor(o, i, op=dist.ReduceOp.SUM, group=self.group), out, full)
            return out, LabelShard._Done()
        return out, dist.reduce_scatter_tensor(out, full, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def all_reduce_(self, t, tag='all_reduce'):
        return self._timed(tag, t.numel() * t.element_size(),
                           lambda: self._run(lambda o, i: dist.all_reduce(o, op=dist.ReduceOp.SUM, group=self.group), t, t))

    def all_gather_logits_loss(self, local, loss, L):
        """[B, L_r] logit column shards (uneven allowed) + this rank's partial loss -> ([B, L] logits, global loss) in ONE
        collective: every rank contributes a [B + 1, width] block (row B carries its loss partial), gathered as
        [P, B + 1, width]; the logits come out of one permuted copy (one column gather when L % P != 0) and the loss is
        the sum of P gathered partials in rank order -- bitwise the same on every rank.  (Before: a transposed gather,
        P slices, a cat and a separate 4-byte all_reduce: two latency-bound collectives and ~6 + P launches.)"""
        B, P = local.shape[0], self.world
        width = (L + P - 1) // P
        blk = local.new_zeros((B + 1, width))
        blk[:B, :local.shape[1]] = local
        blk[B, :1] = loss.reshape(1)
        out = local.new_empty((P * (B + 1), width))
        self._timed('logits_loss_all_gather', out.numel() * out.element_size(),
                    lambda: self._run(lambda o, i: dist.all_gather_into_tensor(o, i, group=self.group), out, blk))
        out = out.view(P, B + 1, width)
        logits = out[:, :B, :].permute(1, 0, 2).reshape(B, P * width)
        if P * width != L:
            key = (L, P, str(local.device))
            cols = self._cols.get(key)
            if cols is None:
                idx = []
                for r in range(P):
                    b, e = partition(L, P, r)
                    idx.extend(range(r * width, r * width + (e - b)))
                cols = self._cols[key] = torch.tensor(idx, dtype=torch.int64, device=local.device)
            logits = logits.index_select(1, cols)
        return logits, out[:, B, 0].sum()

    def all_gather_labels(self, local, L):
        """[B, L_r] column shards (uneven allowed) -> [B, L]."""
        B = local.shape[0]
        width = (L + self.world - 1) // self.world
        pad = local.new_zeros((width, B))
        pad[:local.shape[1]] = local.t()
        out = local.new_empty((self.world * width, B))
        self._timed('logits_all_gather', out.numel() * out.element_size(),
                    lambda: self._run(lambda o, i: dist.all_gather_into_tensor(o, i, group=self.group), out, pad))
        cols = []
        for r in range(self.world):
            b, e = partition(L, self.world, r)
            cols.append(out[r * width: r * width + (e - b)])
        return torch.cat(cols, 0).t().contiguous()


class FlatGrads:
    """Gradient tensors carved out of one flat buffer, so the shared-parameter all-reduce is a single collective."""

    def __init__(self, P, names):
        total = sum(P[k].numel() for k in names)
        ref = P[names[0]]
        self.flat = torch.zeros(total, dtype=ref.dtype, device=ref.device)
        self.views = {}
        off = 0
        for k in names:
            n = P[k].numel()
            self.views[k] = self.flat[off:off + n].view_as(P[k])
            off += n

    def __getitem__(self, k):
        return self.views[k]

    def __contains__(self, k):
        return k in self.views
