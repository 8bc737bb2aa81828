"""Mirror of the reference's modules/common_module.py (EmbLayer, loss / optimizer / scheduler factories, AdamW).

Same names, signatures and defaults (common_module.py:26-182).  The loss object returned by `get_loss_fn('bce')`
is torch's own `BCEWithLogitsLoss` and serves as the configuration marker the model mirrors check; the arithmetic
of the LDAM-shifted BCE runs in the fused `gzsl_ldam_bce` kernel.  Focal and multi-label-margin losses are never
selected by the drivers' defaults (config.py:26) and are not built: asking for them raises.
"""
import torch
import torch.nn as nn
import torch.optim as optim
from torch.nn import BCEWithLogitsLoss
from torch.optim.optimizer import Optimizer

EPS = 1e-7


def get_loss_fn(loss='FOCAL', reduction='sum'):
    if loss.upper() == 'BCE':
        return BCEWithLogitsLoss(reduction=reduction)
    raise NotImplementedError(f"loss '{loss}' is outside the MI355X hot path (only 'bce', the drivers' default)")


def add_emb_weight_decay(model, weight_decay, emb_weight_decay=1e-2):
    decay, no_decay = [], []
    for name, param in model.named_parameters():
        (decay if name == 'emb.embed.weight' else no_decay).append(param)
    return [{'params': no_decay, 'weight_decay': weight_decay}, {'params': decay, 'weight_decay': emb_weight_decay}]


def get_optimizer(lr, model, freeze_emb=False, opt='ADAM', weight_decay=1e-4):
    """common_module.py:105-119: Adam (or SGD with momentum 0.9) over the trainable parameters; `freeze_emb` pins
    the word table first."""
    named = dict(model.named_parameters())
    if freeze_emb and 'emb.embed.weight' in named:
        named['emb.embed.weight'].requires_grad = False
    trainable = [p for p in model.parameters() if p.requires_grad]
    kind = str(opt).upper() if opt in ('ADAM', 'SGD') else None
    if kind is None:
        raise ValueError(f'optimizer is SGD or ADAM, {opt} not recognized!')
    if kind == 'SGD':
        return optim.SGD(trainable, lr=lr, momentum=0.9, weight_decay=weight_decay)
    # same update rule as the reference's optim.Adam; on the device the whole parameter list is updated by one
    # multi-tensor launch instead of ~8 elementwise passes per tensor
    on_device = bool(trainable) and all(p.is_cuda for p in trainable)
    return optim.Adam(trainable, lr=lr, weight_decay=weight_decay, fused=on_device)


def get_scheduler(optimizer, num_epochs, ratios=(0.6, 0.8)):
    """common_module.py:122-123: step decay at the given fractions of the run."""
    return optim.lr_scheduler.MultiStepLR(optimizer, milestones=[int(r * num_epochs) for r in ratios])


class EmbLayer(nn.Module):
    """Holder of the pretrained word table (common_module.py:70-88; state-dict key `embed.weight`, row `pad_idx` is the
    padding row).  `forward` gathers rows with the HIP kernel."""

    def __init__(self, embed_size, vocab_size, pad_idx=0, use_emb=True, W=None, freeze_emb=True):
        super().__init__()
        if use_emb:
            if W is None:
                raise AssertionError('use_emb=True needs the pretrained table W')
            table = torch.as_tensor(W).detach().clone()
            rows, embed_size = table.shape
            self.embed = nn.Embedding(rows, embed_size, padding_idx=pad_idx)
            with torch.no_grad():
                self.embed.weight.copy_(table)
        else:
            self.embed = nn.Embedding(vocab_size + 1, embed_size, padding_idx=pad_idx)
        self.embed_size = embed_size
        self.embed.weight.requires_grad_(not freeze_emb)

    def forward(self, x):
        if getattr(self, '_ops', None) is None:
            from ..ops_hip import HipOps
            self._ops = HipOps()   # one provider per module, not per call
        return self._ops.emb_gather(self.embed.weight.detach(), x.contiguous(), None, 0, 0.0)


class AdamW(Optimizer):
    """The reference's hand-rolled AdamW (common_module.py:126-182): no bias correction, eps added to sqrt(v),
    weight decay applied to the parameter; per-parameter state keys `step`, `next_m`, `next_v`.  One fused kernel per
    parameter tensor (`gzsl_adamw_nobias_step`)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), e=1e-6, weight_decay=0.0, ops=None):
        for which, beta in enumerate(betas):
            if not 0.0 <= beta < 1.0:
                raise ValueError(f'Invalid beta parameter at index {which}: {beta} - should be in [0.0, 1.0[')
        if e < 0.0 or e != e:
            raise ValueError(f'Invalid epsilon value: {e} - should be >= 0.0')
        super().__init__(params, dict(lr=lr, betas=betas, e=e, weight_decay=weight_decay))
        self._ops = ops

    def _moments(self, p):
        st = self.state[p]
        if not st:
            st.update(step=0, next_m=torch.zeros_like(p.data), next_v=torch.zeros_like(p.data))
        return st

    def step(self, closure=None):
        loss = closure() if closure is not None else None
        if self._ops is None:
            from ..ops_hip import HipOps
            self._ops = HipOps()
        for group in self.param_groups:
            b1, b2 = group['betas']
            for p in group['params']:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError('Adam does not support sparse gradients, please consider SparseAdam instead')
                st = self._moments(p)
                self._ops.adamw_step_(p.data, p.grad.data.contiguous(), st['next_m'], st['next_v'], group['lr'], b1, b2,
                                      group['e'], group['weight_decay'])
                st['step'] += 1
        return loss
