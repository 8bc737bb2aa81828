"""Mirror of the reference's modules/decoder_module.py: keyword head on generated features (decoder_module.py:21-67).

`LinearKeywordsDecoder.forward` returns the per-row soft-target cross entropy over the keyword vocabulary; the
[B, Vk] target matrix the reference scatters is never formed (the fused `gzsl_keyword_ce` kernel reads the <= top_k
target slots per row, last duplicate wins as index assignment does).  `binary=True` is never selected by the drivers
(decoder_module.py:64) and is not built.
"""
import torch
import torch.nn as nn

from ..gan_engine import GanEngine
from .generative_module import _ops_of


def normalize_mask(mask):
    return mask / (mask.sum(1, keepdim=True) + 1e-7)


class _KeywordFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, feats, kw_idx, kw_mask, W, b):
        ops = _ops_of(mod)
        kw = {'proj_fc.weight': W.detach(), 'proj_fc.bias': b.detach(), 'word_emb': mod.word_emb}
        f = feats.detach().contiguous()
        xk = ops.mm(f, kw['proj_fc.weight'].t(), bias=kw['proj_fc.bias'])
        logits = ops.mm(xk, kw['word_emb'].t())
        loss_b = ops.keyword_ce_(logits, kw_idx.contiguous(), kw_mask.contiguous().float(), None)
        ctx.mod, ctx.state = mod, (kw, f, logits)  # logits now holds d loss_b / d logits (unit row weights)
        ctx.need_feat = feats.requires_grad
        return loss_b

    @staticmethod
    def backward(ctx, dloss_b):
        kw, f, dlogits = ctx.state
        ops = _ops_of(ctx.mod)
        dlogits = dlogits * dloss_b[:, None]
        dxk = ops.mm(dlogits, kw['word_emb'])
        dW = ops.mm(dxk.t(), f)
        db = ops.colsum(dxk)
        dfeat = ops.mm(dxk, kw['proj_fc.weight']) if ctx.need_feat else None
        return None, dfeat, None, None, dW, db


class LinearKeywordsDecoder(nn.Module):
    def __init__(self, feat_size, emb_size, word_emb, binary=False):
        super(LinearKeywordsDecoder, self).__init__()
        if binary:
            raise NotImplementedError('binary keyword loss is never selected by the drivers (decoder_module.py:64)')
        self.register_buffer('word_emb', word_emb.contiguous())
        self.proj_fc = nn.Linear(feat_size, emb_size, bias=True)
        self.binary = binary
        self.V = word_emb.size(0)

    def use_ops(self, ops):
        self._ops = ops
        return self

    def predict(self, feats):
        ops = _ops_of(self)
        x = ops.mm(feats.detach().contiguous(), self.proj_fc.weight.detach().t(), bias=self.proj_fc.bias.detach())
        return ops.mm(x, self.word_emb.t())

    def forward(self, feats, keyword_indices, keyword_masks, label_emb=None):
        return _KeywordFn.apply(self, feats, keyword_indices, keyword_masks, self.proj_fc.weight, self.proj_fc.bias)


def load_decoder(decoder, feat_size, embed_size, keyword_emb):
    if decoder == 'linear':
        return LinearKeywordsDecoder(feat_size, embed_size, keyword_emb)
    raise ValueError('Wrong decoder')
