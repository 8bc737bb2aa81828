"""Mirror of the reference's modules/generative_module.py on HIP kernels (generative_module.py:22-68).

`ConditionalGenerator`, `ConditionalDiscriminator` and `calc_gradient_penalty` keep their names, signatures and
state-dict keys (`fc1.weight`, `fc1.bias`, `fc2.weight`, `fc2.bias`), so train_gan.py / finetune.py call them
unchanged.  nn.Linear is a parameter holder; the arithmetic runs through gzsl_text_amd.gan_engine.GanEngine inside
torch.autograd.Functions with hand-written backwards.  The gradient penalty is the closed form (no double backward):
its gradient reaches fc1.weight[:, :D] and fc2.weight only, exactly the entries autograd leaves non-zero.

Inputs `noise` / `label` / `labels` are treated as constants (the driver detaches them: train_gan.py:241,277 wrap them
in `Variable(...)`); the critic's `feat` input does receive a gradient (generator step, train_gan.py:299-306).
"""
import torch
import torch.nn as nn

from ..gan_engine import GanEngine, Layout, zero_grads


def _ops_of(module):
    ops = getattr(module, '_ops', None)
    if ops is None:
        from ..ops_hip import HipOps
        ops = HipOps()
        module._ops = ops
    return ops


def _pd(module):
    return {k: v.detach() for k, v in module.named_parameters()}


def _zero_like_params(params, keys=None):
    """Zeroed gradient tensors for `params` carved out of one buffer (one fill launch, 256-byte aligned tensors)."""
    keys = list(params) if keys is None else keys
    ref = params[keys[0]]
    return zero_grads(Layout([(k, params[k].shape) for k in keys]), ref.device)[1]


class _GeneratorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, noise, label, *params):
        eng = GanEngine(_ops_of(mod))
        g = dict(zip(mod._names, [p.detach() for p in params]))
        noise, label = noise.detach().contiguous(), label.detach().contiguous()
        out, h = eng.generator_fwd(g, noise, label, relu_out=False)
        ctx.mod, ctx.state = mod, (g, noise, label, h)
        return out

    @staticmethod
    def backward(ctx, dout):
        g, noise, label, h = ctx.state
        eng = GanEngine(_ops_of(ctx.mod))
        G = _zero_like_params(g)
        eng.generator_bwd(dout.contiguous().clone(), None, h, g, noise, label, G, relu_out=False)
        return (None, None, None) + tuple(G[k] for k in ctx.mod._names)


class ConditionalGenerator(nn.Module):
    _names = ('fc1.weight', 'fc1.bias', 'fc2.weight', 'fc2.bias')

    def __init__(self, noise_size, label_size, hidden_size, output_size):
        super(ConditionalGenerator, self).__init__()
        self.fc1 = nn.Linear(noise_size + label_size, hidden_size)
        self.fc2 = nn.Linear(hidden_size, output_size)
        self.activation = nn.LeakyReLU(0.2)

    def use_ops(self, ops):
        self._ops = ops
        return self

    def forward(self, noise, label):
        d = dict(self.named_parameters())
        return _GeneratorFn.apply(self, noise, label, *[d[k] for k in self._names])


class _CriticFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, feat, label, *params):
        eng = GanEngine(_ops_of(mod))
        d = dict(zip(mod._names, [p.detach() for p in params]))
        f, label = feat.detach().contiguous(), label.detach().contiguous()
        lab_half = eng.critic_label_half(d, label, f.shape[1])
        logits, h = eng.critic_fwd(d, f, lab_half, consume=True)
        ctx.mod, ctx.state = mod, (d, f, label, h)
        ctx.need_feat = feat.requires_grad
        ctx.need_params = any(p.requires_grad for p in params)
        return logits

    @staticmethod
    def backward(ctx, dout):
        d, f, label, h = ctx.state
        eng = GanEngine(_ops_of(ctx.mod))
        G = _zero_like_params(d) if ctx.need_params else None
        dh = eng.critic_bwd(dout.contiguous(), h, d, G)
        if G is not None:
            eng.critic_fc1_grads(dh, f, label, G)
        dfeat = eng.ops.mm(dh, d['fc1.weight'][:, :f.shape[1]]) if ctx.need_feat else None
        return (None, dfeat, None) + tuple((G[k] if G is not None else None) for k in ctx.mod._names)


class ConditionalDiscriminator(nn.Module):
    _names = ('fc1.weight', 'fc1.bias', 'fc2.weight', 'fc2.bias')

    def __init__(self, feat_size, label_size, hidden_size, output_size=1):
        super(ConditionalDiscriminator, self).__init__()
        if output_size != 1:
            raise NotImplementedError('the WGAN critic has one output (train_gan.py:145)')
        self.fc1 = nn.Linear(feat_size + label_size, hidden_size)
        self.fc2 = nn.Linear(hidden_size, output_size)
        self.activation = nn.LeakyReLU(0.2)

    def use_ops(self, ops):
        self._ops = ops
        return self

    def forward(self, feat, label):
        d = dict(self.named_parameters())
        return _CriticFn.apply(self, feat, label, *[d[k] for k in self._names]).squeeze()


class _GradPenaltyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, real, fake, labels, alpha, lam, *params):
        eng = GanEngine(_ops_of(mod))
        d = dict(zip(mod._names, [p.detach() for p in params]))
        real, fake, labels = real.detach().contiguous(), fake.detach().contiguous(), labels.detach().contiguous()
        D = real.shape[1]
        lab_half = eng.critic_label_half(d, labels, D)
        _, h_real = eng.critic_fwd(d, real, lab_half, need_logits=False)
        _, h_fake = eng.critic_fwd(d, fake, lab_half, consume=True, need_logits=False)
        # the penalty depends on the parameters only through fc1.weight[:, :D] and fc2.weight (the slope mask is
        # piecewise constant): the two biases get no gradient (autograd leaves exact zeros there; None adds nothing)
        G = _zero_like_params(d, ('fc1.weight', 'fc2.weight'))
        gp = eng.gradient_penalty(d, h_real, h_fake, alpha, D, G, lam)
        ctx.mod, ctx.G = mod, G
        return gp

    @staticmethod
    def backward(ctx, dgp):
        G = ctx.G
        return (None,) * 6 + tuple((G[k] * dgp if k in G else None) for k in ctx.mod._names)


def calc_gradient_penalty(discriminator, real_feats, fake_feats, labels, lambda1=10.):
    """generative_module.py:22-38.  alpha is drawn exactly as the reference draws it (torch.rand(b, 1) on the host,
    then moved to the device), so the same torch seed gives the same interpolation points."""
    b = real_feats.size(0)
    alpha = torch.rand(b, 1)  # the reference's draw: host generator, [b, 1]
    if real_feats.is_cuda:
        # a pageable H2D copy waits for everything queued on the stream (0.8 ms per critic iteration at batch 8, five
        # times per step): stage through pinned memory and let the copy queue behind the kernels instead
        alpha = alpha.pin_memory().to(real_feats.device, non_blocking=True)
    else:
        alpha = alpha.to(real_feats.device)
    d = dict(discriminator.named_parameters())
    return _GradPenaltyFn.apply(discriminator, real_feats, fake_feats, labels, alpha, float(lambda1),
                                *[d[k] for k in discriminator._names])
