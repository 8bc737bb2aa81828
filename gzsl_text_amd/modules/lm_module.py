"""Mirror of the reference's modules/lm_module.py: GRU encoder of ICD description tokens (lm_module.py:20-88).

Adjacent to the hot path: train_gan.py calls `forward_enc` in every critic / generator
iteration, yet the encoder's parameters receive exactly zero gradient there (labels are detached by `Variable(...)`,
and the penalty depends on them only through LeakyReLU slopes), so with Adam they never leave their initial values.
What the build adds: `LabelEncodingCache`, an exact per-label table that removes the per-iteration recompute (the
dominant CPU cost of the reference's GAN step), filled by `forward_enc_hip`: the packed GRU forward on the HIP kernels
(embedding gather, one GEMM for all input projections, per step one GEMM + `gzsl_gru_seq_step`).  Calls that need
autograd through the GRU (`forward_enc` itself) stay on stock PyTorch-ROCm.
Same constructor signature, attributes and state-dict keys (`encoder.weight`, `rnn.*_l0`, `decoder.weight`).
"""
import torch
import torch.nn as nn
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence


class RNNLabelEncoder(nn.Module):
    def __init__(self, word_emb, nlayers=1, dropout=0., rnn_type='GRU', tie_weights=True, freeze_emb=True):
        super(RNNLabelEncoder, self).__init__()
        ntoken, ninp = word_emb.shape
        rnn_type = rnn_type.upper()
        assert rnn_type in ['LSTM', 'GRU']
        self.drop = nn.Dropout(dropout)
        self.encoder = nn.Embedding(ntoken, ninp, padding_idx=0)
        self.rnn = getattr(nn, rnn_type)(ninp, ninp, nlayers, dropout=dropout, batch_first=True)
        self.decoder = nn.Linear(ninp, ntoken, bias=False)
        self.init_weights(word_emb)
        if tie_weights:
            self.decoder.weight = self.encoder.weight
        self.rnn_type, self.nlayers = rnn_type, nlayers
        if freeze_emb:
            self.encoder.weight.requires_grad = False

    def init_weights(self, word_emb):
        self.encoder.weight.data = torch.from_numpy(word_emb)
        self.decoder.weight.data.uniform_(-0.1, 0.1)

    def init_hidden_states(self, init_hidden=None):
        if init_hidden is None:
            return None
        h = init_hidden.unsqueeze(0)
        return (h, h) if self.rnn_type == 'LSTM' else h

    def forward_emb(self, input, mask):
        emb = self.drop(self.encoder(input) * mask.unsqueeze(2))
        return pack_padded_sequence(emb, mask.sum(1).cpu(), batch_first=True, enforce_sorted=False)

    def forward_enc(self, input, input_mask, init_hidden=None, training=True, pool_mode='last'):
        outputs, hn = self.rnn(self.forward_emb(input, input_mask), self.init_hidden_states(init_hidden))
        if training:
            return hn
        outputs, lengths = pad_packed_sequence(outputs, batch_first=True)
        if pool_mode == 'mean':
            return outputs.sum(1) / lengths.to(outputs.device).unsqueeze(1).float()
        if pool_mode == 'max':
            return outputs.masked_fill(outputs == 0., float('-inf')).max(1)[0]
        if self.rnn_type == 'LSTM':
            hn = hn[0]
        return hn.squeeze()


def forward_enc_hip(enc, input, input_mask, pool_mode='last', ops=None):
    """`enc.forward_enc(input, input_mask, None, training=False, pool_mode)` (lm_module.py:68-88) without autograd, on
    the HIP kernels.  GRU, one layer, zero initial state; descriptions are prefix-masked (mask.sum(1) = length)."""
    if enc.rnn_type != 'GRU' or enc.nlayers != 1:
        raise NotImplementedError('the HIP label encoder implements the drivers\' 1-layer GRU (train_gan.py:150)')
    if ops is None:
        from ..ops_hip import HipOps
        ops = HipOps()
    Bg, T = input.shape
    mask = input_mask.contiguous().to(torch.float32)
    W = enc.encoder.weight.detach()
    rnn = enc.rnn
    Wih, Whh = rnn.weight_ih_l0.detach(), rnn.weight_hh_l0.detach()
    bih, bhh = rnn.bias_ih_l0.detach(), rnn.bias_hh_l0.detach()
    H = Whh.shape[1]
    x = ops.emb_gather(W, input.contiguous(), mask, 0, 0.0)                  # [Bg, T, E] = W[ids] * mask
    gi = ops.linear(x.view(Bg * T, x.shape[2]), Wih, bih).view(Bg, T, 3 * H)  # every step's input projection
    h = torch.zeros((Bg, H), dtype=torch.float32, device=W.device)
    pool = None
    if pool_mode == 'max':
        pool = torch.full((Bg, H), float('-inf'), dtype=torch.float32, device=W.device)
    elif pool_mode == 'mean':
        pool = torch.zeros((Bg, H), dtype=torch.float32, device=W.device)
    for t in range(T):
        gh = ops.linear(h, Whh, bhh)
        ops.gru_seq_step_(gi, t, gh, h, mask, pool, pool_mode if pool is not None else 'last')
    return h if pool is None else pool


class LabelEncodingCache:
    """Exact table of `label_rnn.forward_enc(desc[:, :trunc], mask[:, :trunc], None, training=False, pool_mode)` for
    every label, filled once (in chunks) and indexed afterwards: `cache(label_indices)` equals what
    train_gan.py:179-183 / finetune.py:144-148 recompute per iteration.  Call `invalidate()` if the encoder's
    parameters are ever changed (they are not, by the drivers: SURVEY.md 8a-12)."""

    def __init__(self, label_rnn, code_desc, code_mask, trunc=20, pool_mode='max', chunk=4096, ops=None):
        self.rnn, self.desc, self.mask = label_rnn, code_desc[:, :trunc], code_mask[:, :trunc]
        self.pool_mode, self.chunk, self.table, self.ops = pool_mode, chunk, None, ops

    def invalidate(self):
        self.table = None

    @torch.no_grad()
    def _fill(self):
        was = self.rnn.training
        self.rnn.eval()
        if self.ops is not None or self.desc.is_cuda:  # HIP packed GRU (or the test suite's injected stand-in)
            parts = [forward_enc_hip(self.rnn, self.desc[i:i + self.chunk], self.mask[i:i + self.chunk], self.pool_mode,
                                     self.ops) for i in range(0, self.desc.shape[0], self.chunk)]
        else:
            parts = [self.rnn.forward_enc(self.desc[i:i + self.chunk], self.mask[i:i + self.chunk], None, training=False,
                                          pool_mode=self.pool_mode) for i in range(0, self.desc.shape[0], self.chunk)]
        self.table = torch.cat(parts, 0).contiguous()
        self.rnn.train(was)

    def __call__(self, label_indices):
        if self.table is None:
            self._fill()
        return self.table.index_select(0, label_indices)
