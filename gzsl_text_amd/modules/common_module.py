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
    if freeze_emb:
        for name, param in model.named_parameters():
            if name == 'emb.embed.weight':
                param.requires_grad = False
                break
    params = [p for p in model.parameters() if p.requires_grad]
    if opt == 'ADAM':
        # same update rule as the reference's optim.Adam; on device the whole parameter list is updated by one
        # multi-tensor launch instead of ~8 elementwise passes
        fused = len(params) > 0 and all(p.is_cuda for p in params)
        return optim.Adam(params, weight_decay=weight_decay, lr=lr, fused=fused)
    elif opt == 'SGD':
        return optim.SGD(params, weight_decay=weight_decay, lr=lr, momentum=0.9)
    raise ValueError("optimizer is SGD or ADAM, {} not recognized!".format(opt))


def get_scheduler(optimizer, num_epochs, ratios=(0.6, 0.8)):
    milestones = [int(num_epochs * r) for r in ratios]
    return optim.lr_scheduler.MultiStepLR(optimizer, milestones=milestones)


class EmbLayer(nn.Module):
    """Holder of the pretrained word table (common_module.py:70-88).  `forward` gathers rows with the HIP kernel."""

    def __init__(self, embed_size, vocab_size, pad_idx=0, use_emb=True, W=None, freeze_emb=True):
        super(EmbLayer, self).__init__()
        if use_emb:
            assert W is not None
            W = torch.from_numpy(W)
            self.embed = nn.Embedding(W.size()[0], W.size()[1], padding_idx=pad_idx)
            self.embed.weight.data = W.clone()
            embed_size = W.size()[1]
        else:
            self.embed = nn.Embedding(vocab_size + 1, embed_size, padding_idx=pad_idx)
        self.embed_size = embed_size
        if freeze_emb:
            self.embed.weight.requires_grad = False

    def forward(self, x):
        from ..ops_hip import HipOps
        return HipOps().emb_gather(self.embed.weight.detach(), x.contiguous(), None, 0, 0.0)


class AdamW(Optimizer):
    """The reference's hand-rolled AdamW (common_module.py:126-182): no bias correction, eps added to sqrt(v),
    weight decay applied to the parameter.  One fused kernel per parameter tensor (`gzsl_adamw_nobias_step`)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), e=1e-6, weight_decay=0.0, ops=None):
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError("Invalid beta parameter at index 0: {} - should be in [0.0, 1.0[".format(betas[0]))
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameter at index 1: {} - should be in [0.0, 1.0[".format(betas[1]))
        if not e >= 0.0:
            raise ValueError("Invalid epsilon value: {} - should be >= 0.0".format(e))
        defaults = dict(lr=lr, betas=betas, e=e, weight_decay=weight_decay)
        super(AdamW, self).__init__(params, defaults)
        self._ops = ops

    def step(self, closure=None):
        loss = None
        if closure is not None:
            loss = closure()
        if self._ops is None:
            from ..ops_hip import HipOps
            self._ops = HipOps()
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is None:
                    continue
                grad = p.grad.data
                if grad.is_sparse:
                    raise RuntimeError('Adam does not support sparse gradients, please consider SparseAdam instead')
                state = self.state[p]
                if len(state) == 0:
                    state['step'] = 0
                    state['next_m'] = torch.zeros_like(p.data)
                    state['next_v'] = torch.zeros_like(p.data)
                beta1, beta2 = group['betas']
                self._ops.adamw_step_(p.data, grad.contiguous(), state['next_m'], state['next_v'], group['lr'], beta1,
                                      beta2, group['e'], group['weight_decay'])
                state['step'] += 1
        return loss
