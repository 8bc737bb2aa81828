"""Mirror of the reference's modules/icd_modules.py: ConvLabelAttnModel / ConvLabelAttnGAN on HIP kernels.

Same class names, constructor signatures, attributes (`name`, `pretrain_name`, `feat_size`, `embed_size`,
`eval_code_size`, `emb.embed.weight`, `output_fc.weight`, `class_margin`, `C`, the four graph buffers) and state-dict
keys as icd_modules.py:36-294, so train_base.py / train_gan.py / finetune.py call them unchanged.  torch.nn layers
(`nn.Conv1d`, `nn.Linear`, `nn.GRUCell`, `nn.Embedding`) are used as PARAMETER HOLDERS only -- they give the
reference's parameter names, shapes and default initialisation; their forwards are never called.  The arithmetic
runs in libgzsl_hip.so through BaseEngine, inside one torch.autograd.Function per model variant whose backward is
hand-written (no autograd graph over the kernels, no B x L x T' tensors).

Extra, not in the reference:
  * `enable_label_sharding(group=None)`: shard the label dimension over the ranks of a torch.distributed group
    (gzsl_text_amd/parallel.py).  Forward then returns the all-gathered [B, L] logits and the global loss.
  * `set_dropout_seed(seed)`: base seed of the counter RNG that replaces torch's dropout stream; the seed of a
    forward is base + number of training forwards so far (all ranks must use the same base).
"""
import math
import os
from collections import OrderedDict

import torch
import torch.nn as nn
from torch.nn.init import xavier_uniform_

from ..engine import BaseEngine, GraphBuffers
from ..parallel import FlatGrads, LabelShard
from ..streams import Beside as _Beside
from .common_module import EmbLayer
from .graph_module import ConvGraphEncoder, GraphGatedEncoder


def log(s):
    import sys
    sys.stderr.write(s if s.endswith('\n') else s + '\n')
    sys.stderr.flush()


def get_graph_encoder(graph_encoder='conv'):
    if graph_encoder == 'conv':
        return ConvGraphEncoder
    elif graph_encoder == 'gate':
        return GraphGatedEncoder
    else:
        return None


def _make_ops():
    from ..ops_hip import HipOps
    return HipOps()


class _BaseModelFn(torch.autograd.Function):
    """ConvLabelAttnModel.forward (icd_modules.py:174-186) and the backward autograd ran for it."""

    @staticmethod
    def forward(ctx, mod, x, y, mask, need_grad, *params):
        names = mod._param_names()
        P = {k: p.detach() for k, p in zip(names, params)}
        eng, gb, sh = mod._engine(), mod._graph_buffers(), mod._shard
        training = mod.training
        seed = mod._next_seed(training)
        B = x.shape[0]
        L = mod.eval_code_size
        x = x.contiguous()
        mask_c = None if mask is None else mask.contiguous().to(torch.float32)
        l0, l1 = (0, L) if sh is None else sh.label_range(L)
        bs = None if sh is None else sh.batch_split(B)
        attn = None
        if bs is not None:
            # this rank encodes its B/P notes; X and PX then travel to every rank
            b0, b1 = bs
            beside = _Beside(x)
            with beside:  # the replicated label encoder runs beside the note encoder AND the gathers
                avg, graph, lcache = eng.label_fwd(P, gb, keep=need_grad)
            Xl, PXl, e = eng.encoder_fwd(P, x[b0:b1].contiguous(), None if mask_c is None else mask_c[b0:b1].contiguous(),
                                         seed, training, row_offset=b0 * x.shape[1])
            X, hx = sh.all_gather_batch_start(Xl)
            PX, hp = sh.all_gather_batch_start(PXl)
            if sh.own_first and sh.world > 1:
                # the attention of this rank's own notes needs nothing from the wire: run it under the gathers
                beside.join()
                Ql = avg[l0:l1]
                O = torch.empty((B, l1 - l0, Xl.shape[2]), dtype=torch.float32, device=Xl.device)
                lse = torch.empty((B, l1 - l0), dtype=torch.float32, device=Xl.device)
                eng.attn_fwd_block(PXl, Xl, Ql, l0, L, seed, training, O, lse, b0)
                sh.wait(hx, 'fwd_all_gather_X', X)
                sh.wait(hp, 'fwd_all_gather_PX', PX)
                for c0, c1 in ((0, b0), (b1, B)):
                    if c1 > c0:
                        eng.attn_fwd_block(PX[c0:c1], X[c0:c1], Ql, l0, L, seed, training, O, lse, c0)
                attn = (O, lse)
            else:
                sh.wait(hx, 'fwd_all_gather_X', X)
                sh.wait(hp, 'fwd_all_gather_PX', PX)
                beside.join()
        else:
            beside = _Beside(x)
            with beside:
                avg, graph, lcache = eng.label_fwd(P, gb, keep=need_grad)
            X, PX, e = eng.encoder_fwd(P, x, mask_c, seed, training)
            beside.join()
        Q = avg[l0:l1]
        clf = graph[l0:l1]
        O, lse, direct, logits_l = eng.head_fwd(P, X, PX, Q, clf, l0, L, seed, training, attn=attn, need_direct=need_grad)
        margin = mod.class_margin
        y_l = y if (l0 == 0 and l1 == y.shape[1]) else y[:, l0:l1]
        y_l = y_l.contiguous().to(torch.float32)
        m_l = None if margin is None else margin[l0:l1]
        loss, dlogits, _ = eng.ops.ldam_bce(logits_l, y_l, m_l, mod.C, 1.0 / B, need_grad=need_grad)
        if sh is not None:
            logits, loss = sh.all_gather_logits_loss(logits_l, loss, L)
        else:
            logits = logits_l
        if need_grad:
            ctx.mod = mod
            ctx.state = (P, x, mask_c, e, X, PX, Q, clf, O, lse, direct, dlogits, lcache, avg.shape, graph.shape,
                         seed, training, l0, l1, bs)
            ctx.needs = [p.requires_grad for p in params]
        ctx.mark_non_differentiable(logits)
        return logits, loss

    @staticmethod
    def backward(ctx, _dlogits_unused, dloss):
        mod = ctx.mod
        (P, x, mask_c, e, X, PX, Q, clf, O, lse, direct, dlogits, lcache, avg_shape, graph_shape, seed, training, l0,
         l1, bs) = ctx.state
        ctx.state = None
        eng, gb, sh = mod._engine(), mod._graph_buffers(), mod._shard
        names = mod._param_names()
        L = mod.eval_code_size
        G = FlatGrads(P, names)
        dlogits = dlogits * dloss  # upstream scalar (train_base.py:112 backpropagates loss.mean(): 1.0)
        # the four accumulation targets of the backward, carved from ONE zero-filled buffer (one fill launch; 256-byte
        # aligned pieces): the kernels below add into them with atomics / accumulate epilogues
        shapes = (tuple(X.shape), tuple(PX.shape), tuple(avg_shape), tuple(graph_shape))
        sizes = [-(-math.prod(sh_) // 64) * 64 for sh_ in shapes]
        zbuf = torch.zeros(sum(sizes), dtype=torch.float32, device=X.device)
        dX, dPX, davg, dgraph = (zbuf[o:o + math.prod(sh_)].view(sh_)
                                 for o, sh_ in zip((0, sizes[0], sizes[0] + sizes[1], sum(sizes[:3])), shapes))
        pending = None
        own = None
        if bs is not None and sh.own_first and sh.world > 1:
            # other ranks' notes first; the reduce-scatter then starts with a zero own block, and the own block's
            # partial (computed while it is on the wire) is added to the scattered result
            b0, b1 = bs
            dclf, dO, pending, delta = eng.head_bwd_pre(dlogits, O, direct, clf, P)
            for c0, c1 in ((0, b0), (b1, X.shape[0])):
                if c1 > c0:
                    eng.attn_bwd_block(dO, O, lse, PX[c0:c1], X[c0:c1], Q, c0, c1, l0, L, seed, training, dPX[c0:c1],
                                       dX[c0:c1], davg[l0:l1], delta=delta)
            own = (dO, torch.zeros_like(dX[b0:b1]), torch.zeros_like(dPX[b0:b1]), delta)
        elif bs is not None:
            dclf, pending = eng.head_bwd(dlogits, O, lse, direct, X, PX, Q, clf, l0, L, seed, training, P, G, dX, dPX,
                                         davg[l0:l1], defer_wgrad=True)
        else:
            dclf = eng.head_bwd(dlogits, O, lse, direct, X, PX, Q, clf, l0, L, seed, training, P, G, dX, dPX,
                                davg[l0:l1])
        dgraph[l0:l1] = dclf
        if bs is not None:
            # the partial dX / dPX of this label shard are summed and scattered over the batch
            b0, b1 = bs
            dXl, hx = sh.reduce_scatter_batch_start(dX)
            dPXl, hp = sh.reduce_scatter_batch_start(dPX)
            if own is not None:
                dO, dX_own, dPX_own, delta = own
                eng.attn_bwd_block(dO, O, lse, PX[b0:b1], X[b0:b1], Q, b0, b1, l0, L, seed, training, dPX_own, dX_own,
                                   davg[l0:l1], delta=delta)
            beside = _Beside(dX)
            with beside:  # beside the reduce-scatters and the note encoder's backward (davg is complete by now)
                eng.label_bwd(davg, dgraph, lcache, P, gb, G)
            eng.head_wgrad(pending, G)  # output_proj weight gradient: compute for the time the reduce-scatter is on the wire
            sh.wait(hx, 'bwd_reduce_scatter_dX', dX)
            sh.wait(hp, 'bwd_reduce_scatter_dPX', dPX)
            if own is not None:
                dXl.add_(dX_own)
                dPXl.add_(dPX_own)
            eng.encoder_bwd(dXl, dPXl, X[b0:b1], PX[b0:b1], e, x[b0:b1].contiguous(),
                            None if mask_c is None else mask_c[b0:b1].contiguous(), seed, training, P, G,
                            row_offset=b0 * x.shape[1])
            beside.join()
        elif sh is not None:
            # replicated encoder: sum the shards' partial dX||dPX, then every rank runs the same encoder backward;
            # its (identical) parameter gradients must not be summed again -> scale by 1/P before the label side adds
            sh.all_reduce_(dX, 'bwd_all_reduce_dX')
            sh.all_reduce_(dPX, 'bwd_all_reduce_dPX')
            eng.encoder_bwd(dX, dPX, X, PX, e, x, mask_c, seed, training, P, G)
            for k in ('emb.embed.weight', 'conv_modules.0.weight', 'conv_modules.0.bias', 'proj.weight', 'proj.bias'):
                G[k].mul_(1.0 / sh.world)
            eng.label_bwd(davg, dgraph, lcache, P, gb, G)
        else:
            # label-side backward beside the encoder's (both add into the word-table gradient: float atomics)
            beside = _Beside(dX)
            with beside:
                eng.label_bwd(davg, dgraph, lcache, P, gb, G)
            eng.encoder_bwd(dX, dPX, X, PX, e, x, mask_c, seed, training, P, G)
            beside.join()
        if sh is not None:
            sh.all_reduce_(G.flat, 'grad_all_reduce')
        return (None, None, None, None, None) + tuple(G[k] if need else None for k, need in zip(names, ctx.needs))


class ConvLabelAttnModel(nn.Module):
    def __init__(self, word_emb, code_idx_matrix, code_idx_mask, adj_matrix, num_neighbors, loss_fn, num_filters=200,
                 kernel_sizes=(10,), eval_code_size=None, label_hidden_size=200, graph_encoder='conv',
                 target_count=None, C=0.):
        super(ConvLabelAttnModel, self).__init__()
        self.name = f"convattn_{graph_encoder}gnn"
        self._graph_encoder_kind = graph_encoder if graph_encoder in ('conv', 'gate') else None

        self.n_nodes = len(code_idx_matrix)
        self.register_buffer('code_idx_matrix', code_idx_matrix)
        self.register_buffer('code_idx_mask', code_idx_mask)
        self.register_buffer('adj_matrix', adj_matrix)
        self.register_buffer('num_neighbors', num_neighbors)

        if not (isinstance(loss_fn, nn.BCEWithLogitsLoss) and loss_fn.reduction == 'sum'):
            raise NotImplementedError('the MI355X hot path implements BCEWithLogitsLoss(reduction="sum") '
                                      '(common_module.py:29-30, the drivers\' default)')
        if len(kernel_sizes) != 1:
            raise NotImplementedError('one conv kernel size (icd_modules.py:130 uses conv_modules[0] only)')
        self.loss_fn = loss_fn
        self.eval_code_size = eval_code_size if eval_code_size is not None else self.n_nodes

        self.emb = EmbLayer(word_emb.shape[1], word_emb.shape[0], W=word_emb, freeze_emb=False)
        self.emb_drop = nn.Dropout(p=0.5)
        self.embed_size = self.emb.embed_size
        self.feat_size = self.embed_size
        self.attn_drop = nn.Dropout(p=0.2)

        graph_encoder_cls = get_graph_encoder(graph_encoder)
        if graph_encoder_cls is not None:
            log(f'Using {graph_encoder_cls.__name__} for encoding ICD hierarchy...')
            self.graph_label_encoder = graph_encoder_cls(self.embed_size, label_hidden_size, self.n_nodes)
            self.feat_size += label_hidden_size
        else:
            self.graph_label_encoder = None
        self.output_proj = nn.Linear(num_filters, self.feat_size)
        xavier_uniform_(self.output_proj.weight)

        self.conv_modules = nn.ModuleList()
        for kernel_size in kernel_sizes:
            self.conv_modules.append(nn.Conv1d(self.embed_size, num_filters, kernel_size=kernel_size))
            xavier_uniform_(self.conv_modules[-1].weight)
        self.proj = nn.Linear(num_filters, self.embed_size)
        xavier_uniform_(self.proj.weight)

        if target_count is not None:
            class_margin = torch.from_numpy(target_count) ** 0.25
            class_margin = class_margin.masked_fill(class_margin == 0, 1)
            self.register_buffer('class_margin', (1.0 / class_margin).to(torch.float32))
            self.C = C
            self.name += f'_cm{int(self.C)}'
        else:
            self.class_margin = None
            self.C = 0
        self.name += '.model'

        # build-side state (not part of the reference's attribute surface)
        self._ops = None
        self._gb = None
        self._gb_key = None
        self._shard = None
        self._drop_seed = 1234
        self._fwd_count = 0

    # ---------------------------------------------------------------------------------------------- plumbing
    _ENCODER_KEYS = ('emb.embed.weight', 'conv_modules.0.weight', 'conv_modules.0.bias', 'proj.weight', 'proj.bias',
                     'output_proj.weight', 'output_proj.bias')

    def _param_names(self):
        names = list(self._ENCODER_KEYS)
        if self.graph_label_encoder is not None:
            names += ['graph_label_encoder.' + k for k, _ in self.graph_label_encoder.named_parameters()]
        return names

    def _param_list(self):
        d = dict(self.named_parameters())
        return [d[k] for k in self._param_names()]

    def _engine(self):
        if self._ops is None:
            self._ops = _make_ops()
        steps = getattr(self.graph_label_encoder, 'steps', 2)
        return BaseEngine(self._ops, self._graph_encoder_kind, steps)

    def use_ops(self, ops):
        """Kernel provider injection point for the test-suite's CPU stand-in (tests/ops_cpu.py)."""
        self._ops = ops
        return self

    def _graph_buffers(self):
        adj = self.adj_matrix
        key = (id(adj), adj.device, self.code_idx_matrix.device)
        if self._gb is None or self._gb_key != key:
            self._gb = GraphBuffers(self.code_idx_matrix, self.code_idx_mask, adj, self.num_neighbors)
            self._gb_key = key
        return self._gb

    def set_dropout_seed(self, seed):
        self._drop_seed = int(seed)
        self._fwd_count = 0

    def _next_seed(self, training):
        if not training:
            return 0
        s = (self._drop_seed + self._fwd_count) & 0xFFFFFFFF
        self._fwd_count += 1
        return s

    def enable_label_sharding(self, group=None, own_first=None):
        self._shard = LabelShard(group, own_first=own_first)
        return self

    # ---------------------------------------------------------------------------------------------- reference API
    def state_dict_to_save(self):
        state_dict = self.state_dict()
        state_dict_save = OrderedDict()
        for k in state_dict:
            if k in ['code_idx_matrix', 'code_idx_mask', 'adj_matrix', 'num_neighbors']:
                continue
            state_dict_save[k] = state_dict[k]
        return state_dict_save

    def _P(self):
        return {k: p.detach() for k, p in zip(self._param_names(), self._param_list())}

    def average_label_embedding(self):
        return self._engine().ops.label_avg(self.emb.embed.weight.detach(), self._graph_buffers().idx,
                                            self._graph_buffers().cmask)

    def get_label_embedding(self):
        """(avg [N,E], graph [N,D]) without gradient tracking (gradients flow inside `forward`'s fused function)."""
        avg, graph, _ = self._engine().label_fwd(self._P(), self._graph_buffers(), keep=False)
        return avg, graph

    def forward_direct_feats(self, x, label_emb, mask=None):
        """icd_modules.py:152-160, inference form: returns (direct [B,L,D], attn_output [B,L,F], None).

        The B x L x T' attention tensor the reference also returns is never formed here."""
        eng = self._engine()
        P = self._P()
        training = self.training
        seed = self._next_seed(training)
        mask_c = None if mask is None else mask.contiguous().to(torch.float32)
        X, PX, _ = eng.encoder_fwd(P, x.contiguous(), mask_c, seed, training)
        Q = label_emb[:self.eval_code_size].contiguous()
        O, lse = eng.ops.attn_fwd(PX, X, Q, 0, Q.shape[0], seed, 0.2 if training else 0.0)
        B, L, F_ = O.shape
        direct = eng.ops.linear(O.view(B * L, F_), P['output_proj.weight'], P['output_proj.bias']).view(B, L, -1)
        return direct, O, None

    def forward_loss(self, logits, y):
        loss, _, _ = self._engine().ops.ldam_bce(logits.contiguous(), y.contiguous().to(torch.float32),
                                                 self.class_margin, self.C, 1.0 / y.size(0), need_grad=False)
        return loss

    def forward(self, x, y, mask=None):
        params = self._param_list()
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        return _BaseModelFn.apply(self, x, y, mask, need_grad, *params)


class _GanModelFn(torch.autograd.Function):
    """ConvLabelAttnGAN.forward (icd_modules.py:251-282): frozen extractor, trainable output_fc rows."""

    @staticmethod
    def forward(ctx, mod, x, y, mask, label_indices, need_grad, fc):
        eng = mod._engine()
        P = mod._P()
        training = mod.training
        seed = mod._next_seed(training)
        L = mod.eval_code_size
        avg, _ = mod.get_froze_label_emb()
        margin = mod.class_margin
        fcw = fc.detach()
        if label_indices is not None:
            idx = label_indices.to(fcw.device)
            Q = avg.index_select(0, idx)
            y = y.index_select(1, idx)
            if margin is not None:
                margin = margin.index_select(0, idx)
            clf = fcw.index_select(0, idx)
        else:
            idx = None
            Q, clf = avg, fcw
        Q = Q[:L].contiguous()
        clf = clf[:L].contiguous()
        if margin is not None:
            margin = margin[:L].contiguous()
        # y is NOT cut to eval_code_size: the reference (icd_modules.py:275) raises a shape error for targets wider than
        # the logits, and so does HipOps.ldam_bce
        B = x.shape[0]
        x = x.contiguous()
        mask_c = None if mask is None else mask.contiguous().to(torch.float32)
        sh = mod._shard
        Lp = Q.shape[0]
        # label sharding (finetune-time scoring over 8 GPUs, finetune.py:224-253): rank r owns rows [l0, l1) of the
        # (possibly gathered) label list; the encoder is batch-split + all-gathered when the batch divides evenly
        l0, l1 = (0, Lp) if sh is None else sh.label_range(Lp)
        bs = None if sh is None else sh.batch_split(B)
        if bs is not None:
            b0, b1 = bs
            Xl, PXl, _ = eng.encoder_fwd(P, x[b0:b1].contiguous(), None if mask_c is None else mask_c[b0:b1].contiguous(),
                                         seed, training, row_offset=b0 * x.shape[1])
            X, PX = sh.all_gather_batch(Xl), sh.all_gather_batch(PXl)
        else:
            X, PX, _ = eng.encoder_fwd(P, x, mask_c, seed, training)
        if sh is not None:
            # same errors as the unsharded path (HipOps.ldam_bce's shape checks) before the columns are cut per rank
            if y.shape[1] != Lp:
                raise RuntimeError(f'ldam_bce: targets {tuple(y.shape)} do not match logits {(B, Lp)}')
            if margin is not None and margin.numel() != Lp:
                raise RuntimeError(f'ldam_bce: {margin.numel()} margins for {Lp} labels')
            Q, clf = Q[l0:l1].contiguous(), clf[l0:l1].contiguous()
            y = y[:, l0:l1]
            if margin is not None:
                margin = margin[l0:l1].contiguous()
        # a label subset keeps its GLOBAL dropout coordinates only when it is the identity; a gathered subset is a
        # different attention problem in the reference too (fresh torch dropout mask), so rows are numbered locally
        O, lse, direct, logits = eng.head_fwd(P, X, PX, Q, clf, l0, Lp, seed, training, need_direct=need_grad)
        loss, dlogits, _ = eng.ops.ldam_bce(logits, y.contiguous().to(torch.float32), margin, mod.C, 1.0 / B,
                                            need_grad=need_grad)
        if sh is not None:
            logits, loss = sh.all_gather_logits_loss(logits, loss, Lp)
        if need_grad:
            ctx.mod = mod
            ctx.state = (direct, clf, dlogits, idx, fcw.shape, l0)
        ctx.mark_non_differentiable(logits)
        return logits, loss

    @staticmethod
    def backward(ctx, _unused, dloss):
        mod = ctx.mod
        direct, clf, dlogits, idx, fc_shape, l0 = ctx.state
        ctx.state = None
        eng = mod._engine()
        _, dclf = eng.ops.logits_bwd(direct, clf, dlogits * dloss, need_ddirect=False)
        dfc = torch.zeros(fc_shape, dtype=torch.float32, device=direct.device)
        n = dclf.shape[0]
        if idx is not None:
            dfc.index_add_(0, idx[l0:l0 + n], dclf)
        else:
            dfc[l0:l0 + n] = dclf
        if mod._shard is not None:  # every rank holds the rows of its label range: sum to the full gradient
            mod._shard.all_reduce_(dfc, 'grad_all_reduce')
        return None, None, None, None, None, None, dfc


class _ForwardFinalFn(torch.autograd.Function):
    """ConvLabelAttnGAN.forward_final (icd_modules.py:233-249)."""

    @staticmethod
    def forward(ctx, mod, feats, labels, need_grad, fc):
        ops = mod._engine().ops
        loss, dfc = ops.forward_final(fc.detach(), feats.contiguous(), labels.contiguous(), mod.class_margin, mod.C, 1.0,
                                      need_grad=need_grad)
        ctx.dfc = dfc
        return loss

    @staticmethod
    def backward(ctx, dloss):
        dfc = ctx.dfc
        ctx.dfc = None
        return None, None, None, None, dfc * dloss


class ConvLabelAttnGAN(ConvLabelAttnModel):
    def __init__(self, word_emb, code_idx_matrix, code_idx_mask, adj_matrix, num_neighbors, loss_fn, num_filters=200,
                 kernel_sizes=(10, ), eval_code_size=None, label_hidden_size=200, graph_encoder='conv',
                 noise_size=256, target_count=None, C=0.):
        super(ConvLabelAttnGAN, self).__init__(word_emb, code_idx_matrix, code_idx_mask, adj_matrix, num_neighbors,
                                               loss_fn, num_filters, kernel_sizes, eval_code_size, label_hidden_size,
                                               graph_encoder, target_count=target_count, C=C)
        self.pretrain_name = self.name
        self.froze_avg_label_emb = None
        self.froze_graph_label_emb = None
        self.noise_size = noise_size
        self.name = 'gan_' + self.name
        self.output_fc = nn.Linear(self.feat_size, self.code_idx_matrix.size(0), bias=False)
        nn.init.xavier_uniform_(self.output_fc.weight)

    def update_class_margin(self, target_count, C):
        assert self.class_margin is not None and self.C > 0
        class_margin = torch.from_numpy(target_count) ** 0.25
        class_margin = 1.0 / class_margin.masked_fill(class_margin == 0, 1)
        self.class_margin = class_margin.to(torch.float32).to(self.class_margin.device)
        self.C = C

    def init_output_fc(self, label_emb):
        with torch.no_grad():
            self.output_fc.weight.data.copy_(label_emb.data)

    def get_froze_label_emb(self):
        if self.froze_avg_label_emb is None or self.froze_graph_label_emb is None:
            avg, graph = self.get_label_embedding()
            if self.froze_avg_label_emb is None:
                self.froze_avg_label_emb = avg
            if self.froze_graph_label_emb is None:
                self.froze_graph_label_emb = graph
        return self.froze_avg_label_emb, self.froze_graph_label_emb

    def forward_real_feats(self, x, mask=None):
        label_emb, _ = self.get_froze_label_emb()
        direct_feats, attn_feats, _ = self.forward_direct_feats(x, label_emb, mask)
        return direct_feats, attn_feats

    def forward_final(self, feats, labels):
        if isinstance(labels, list):
            labels = torch.LongTensor(labels).to(feats.device)
        fc = self.output_fc.weight
        return _ForwardFinalFn.apply(self, feats, labels, torch.is_grad_enabled() and fc.requires_grad, fc)

    def forward(self, x, y, mask=None, label_indices=None):
        fc = self.output_fc.weight
        need_grad = torch.is_grad_enabled() and fc.requires_grad
        if label_indices is not None and not torch.is_tensor(label_indices):
            label_indices = torch.as_tensor(label_indices, dtype=torch.int64)
        return _GanModelFn.apply(self, x, y, mask, label_indices, need_grad, fc)

    def load_pretrained_state_dict(self, pretrained_dict):
        model_dict = self.state_dict()
        for k in pretrained_dict:
            if k in model_dict:
                model_dict[k] = pretrained_dict[k]
        self.load_state_dict(model_dict)
        for name, param in self.named_parameters():
            if name in pretrained_dict:
                param.requires_grad = False
                log(f'Freeze {name} in training...')
