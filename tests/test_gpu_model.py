"""GPU parity of the whole base model / GAN variant: the module mirrors on cuda:0 (HIP kernels through the C ABI)
against the golden vectors captured from the reference (tests/golden/*.npz) -- logits, loss and every gradient."""
import numpy as np
import pytest
import torch

from gzsl_text_amd.modules.icd_modules import ConvLabelAttnGAN
from tests import helpers
from tests.conftest import load_golden
from tests.helpers import close
from tests.golden import synth

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('name', list(synth.BASE_CASES))
@pytest.mark.parametrize('train', [False, True])
def test_base_model_matches_reference_golden(name, train):
    g = load_golden(f"{name}_{'train' if train else 'eval'}")
    case = synth.synth_base(**synth.BASE_CASES[name])
    model = helpers.build_model(case, device=DEV)
    helpers.check_base_against_golden(model, case, g, train, device=DEV,
                                      grad_atol=4e-6 if case['dims']['T'] > 64 else 2e-6)


def test_gan_variant_matches_reference_golden():
    g = load_golden('gan_model_tiny')
    case = synth.synth_base(**synth.BASE_CASES['base_gate_tiny'])
    model = helpers.build_model(case, cls=ConvLabelAttnGAN, device=DEV)
    model.load_pretrained_state_dict({k: torch.from_numpy(v.copy()).to(DEV) for k, v in case['params'].items()})
    model.eval()
    model.init_output_fc(torch.from_numpy(g['fc0']).to(DEV))
    x, y, mask = helpers.batch_tensors(case, DEV)
    model.zero_grad()
    logits, loss = model(x, y, mask)
    loss.backward()
    helpers.close(logits, g['full_logits'], 1e-5, 2e-5, 'full logits')
    helpers.close(loss, g['full_loss'], 1e-5, 1e-5, 'full loss')
    helpers.close(model.output_fc.weight.grad, g['full_dfc'], 1e-4, 1e-6, 'full dfc')
    model.zero_grad()
    logits, loss = model(x, y, mask, label_indices=torch.from_numpy(g['sub']).to(DEV))
    loss.backward()
    helpers.close(logits, g['sub_logits'], 1e-5, 2e-5, 'subset logits')
    helpers.close(model.output_fc.weight.grad, g['sub_dfc'], 1e-4, 1e-6, 'subset dfc')
    with torch.no_grad():
        direct, attn = model.forward_real_feats(x, mask)
    helpers.close(direct, g['real_direct'], 1e-5, 2e-5, 'forward_real_feats direct')
    helpers.close(attn, g['real_attn'], 1e-5, 2e-5, 'forward_real_feats attn')
    from gzsl_text_amd.modules.common_module import AdamW
    opt = AdamW(model.output_fc.parameters(), lr=1e-3, weight_decay=5e-4, betas=(0.5, 0.999))
    feats, labs = torch.from_numpy(g['ff_feats']).to(DEV), torch.from_numpy(g['ff_labels']).to(DEV)
    for step in (1, 2):
        model.zero_grad()
        loss = model.forward_final(feats, labs) / feats.shape[0]
        loss.backward()
        helpers.close(loss, g[f'ff_loss{step}'], 1e-5, 1e-5, f'forward_final loss {step}')
        helpers.close(model.output_fc.weight.grad, g[f'ff_dfc{step}'], 1e-4, 1e-6, f'forward_final grad {step}')
        opt.step()
        helpers.close(model.output_fc.weight, g[f'ff_fc_after{step}'], 1e-5, 1e-6, f'AdamW params after {step}')


@pytest.mark.parametrize('name', list(synth.GAN_TRAIN_CASES))
def test_gan_variant_train_mode_matches_reference_golden(name):
    """finetune.py:258,271 on the HIP kernels: model.train(), forward(x, y, mask[, label_indices]) with both dropouts."""
    helpers.check_gan_variant_train_mode(name, load_golden('gan_model_train_' + name), device=DEV)


@pytest.mark.parametrize('name', ['gan_step_tiny', 'gan_step_medium'])
def test_wgan_gp_step_matches_reference_golden(name):
    from gzsl_text_amd.ops_hip import HipOps
    from tests.gan_checks import run_gan_step_check
    run_gan_step_check(name, HipOps(), DEV)


def test_gan_module_api_on_hip_follows_train_gan_call_order():
    """The nn.Module mirrors train_gan.py:243-306 calls (generator, critic, gradient penalty, keyword head) on cuda:0
    with HIP kernels underneath and torch autograd around them, against the reference's golden vectors."""
    from gzsl_text_amd.ops_hip import HipOps
    from tests.gan_checks import run_gan_module_api_check
    run_gan_module_api_check(HipOps(), DEV)


def test_label_gru_hip_matches_reference_golden():
    """Packed label GRU (lm_module.py:68-88, training=False) on the HIP kernels and the cache filled through it."""
    from gzsl_text_amd.modules.lm_module import LabelEncodingCache, RNNLabelEncoder, forward_enc_hip
    g = load_golden('label_rnn_tiny')
    enc = RNNLabelEncoder(g['word_emb'].copy())
    for k, p in enc.rnn.named_parameters():
        p.data.copy_(torch.from_numpy(g['rnn.' + k]))
    enc = enc.to(DEV)
    ids, m = torch.from_numpy(g['ids']).to(DEV), torch.from_numpy(g['mask']).to(DEV)
    for mode in ('max', 'mean'):
        close(forward_enc_hip(enc, ids, m, mode), g['pool_' + mode], 1e-5, 2e-6, 'HIP packed GRU ' + mode)
    cache = LabelEncodingCache(enc, ids, m, trunc=20, pool_mode='max', chunk=4)
    sel = torch.tensor([3, 0, 3, 10], device=DEV)
    close(cache(sel), g['pool_max'][sel.cpu().numpy()], 1e-5, 2e-6, 'cache filled by the HIP GRU')
    # a larger, ragged batch against the torch (MIOpen / CPU) GRU of the same module
    rs = np.random.RandomState(3)
    W = (rs.standard_normal((500, 200)) * 0.3).astype(np.float32)
    W[0] = 0
    enc2 = RNNLabelEncoder(W)
    lens = rs.randint(1, 21, size=300)
    mk = (np.arange(20)[None] < lens[:, None]).astype(np.float32)
    idz = rs.randint(1, 500, size=(300, 20)) * mk.astype(np.int64)
    with torch.no_grad():
        want = enc2.forward_enc(torch.from_numpy(idz), torch.from_numpy(mk), None, training=False, pool_mode='max')
    got = forward_enc_hip(enc2.to(DEV), torch.from_numpy(idz).to(DEV), torch.from_numpy(mk).to(DEV), 'max')
    close(got, want, 1e-4, 1e-5, 'HIP packed GRU vs torch GRU (E=200, 300 ragged descriptions)')


@pytest.mark.parametrize('name', ['base_gate_medium', 'base_gate_e200'])
def test_deterministic_mode_is_bit_reproducible(name):
    """HipOps(deterministic=True): attention gradients through slabs + an ordered reduce, the embedding gradient as a
    gather over a sorted index, loss / column sums without atomics, one stream.  Two train steps on the same inputs must
    agree bit for bit in loss, logits and every gradient -- and still match the reference's golden vectors."""
    from gzsl_text_amd import ops_hip
    from gzsl_text_amd.ops_hip import HipOps
    from gzsl_text_amd.streams import Beside
    g = load_golden(f'{name}_train')
    case = synth.synth_base(**synth.BASE_CASES[name])
    try:
        model = helpers.build_model(case, device=DEV, ops=HipOps(deterministic=True))
        helpers.check_base_against_golden(model, case, g, True, device=DEV, grad_atol=4e-6)
        x, y, mask = helpers.batch_tensors(case, DEV)
        runs = []
        for _ in range(2):
            model.train()
            model.set_dropout_seed(synth.DROP_SEED)
            model.zero_grad()
            logits, loss = model(x, y, mask)
            loss.mean().backward()
            runs.append([logits.detach().clone(), loss.detach().clone()] + [p.grad.clone() for p in model.parameters()])
        for a, b in zip(*runs):
            assert torch.equal(a, b), float((a - b).abs().max())
        assert HipOps().deterministic      # the mode is process-wide: a provider built later inherits it ...
        assert not HipOps(deterministic=False).deterministic and not Beside.serial   # ... and an explicit False ends it
    finally:
        ops_hip.set_deterministic(False)


def test_gan_plain_wgan_step_on_hip():
    """No sim weights, no keyword head: the branches of the whole-iteration engine the goldens do not reach."""
    from gzsl_text_amd.ops_hip import HipOps
    from tests.gan_checks import run_gan_plain_wgan_check
    run_gan_plain_wgan_check(HipOps(), 'cuda')


def test_gan_full_step_matches_per_iteration_steps_on_hip():
    from gzsl_text_amd.ops_hip import HipOps
    from tests.gan_checks import run_gan_full_step_check
    run_gan_full_step_check(HipOps(), 'cuda')


def test_gan_step_replayed_as_hip_graph_matches_eager():
    """The whole WGAN-GP step (critic iterations + generator iteration + Adam) records into one HIP graph: nothing in
    the C ABI allocates or synchronises.  One eager step plus three replays must land on the parameters four eager steps reach."""
    from gzsl_text_amd.gan_engine import WganGpStep, graphed_step
    from gzsl_text_amd.ops_hip import HipOps
    from tests.gan_checks import GAN_DIMS
    from tests.golden import synth
    kwargs = dict(GAN_DIMS['gan_step_medium'])
    n_iters = kwargs.pop('n_iters')
    s = synth.synth_gan(n_iters=n_iters, **kwargs)

    def dev(d):
        return {k: torch.from_numpy(v.copy()).to('cuda') for k, v in d.items()}

    def t(a):
        return torch.from_numpy(a).to('cuda')

    its = [{k: t(v) for k, v in I.items()} for I in s['iters']]
    kw_idx, kw_mask = t(s['kw_idx']), t(s['kw_mask'])
    results = []
    for graphed in (False, True):
        step = WganGpStep(HipOps(), dev(s['g']), dev(s['d']), dev(s['kw']), lr=1e-4, critic_iters=n_iters, reg_ratio=0.01,
                          capturable=True)
        start = {k: v.clone() for k, v in list(step.g.items()) + list(step.d.items())}

        def one():
            for i in range(n_iters):
                I = its[i]
                step.critic_step(I['real'], I['lab'], I['noise'], I['alpha'], I['sim'])
            I = its[n_iters]
            return step.generator_step(I['lab'], I['noise'], I['sim'], kw_idx, kw_mask)[0]

        if graphed:
            one = graphed_step(one, warmup=1)   # 1 eager step (creates Adam's state) + 3 replays = 4 steps
            for _ in range(3):
                g_cost = one()
        else:
            for _ in range(4):
                g_cost = one()
        torch.cuda.synchronize()
        assert bool(torch.isfinite(g_cost))
        results.append({k: v.detach().clone() for k, v in list(step.g.items()) + list(step.d.items())})
        assert any(not torch.equal(results[-1][k], start[k]) for k in start)
    for k in results[0]:
        # same kernels, same order, same inputs; float atomics reorder sums, and Adam turns a rounding-noise gradient
        # into a +-lr move per step (tests/gan_checks.py::_close_adam): bound = 4 steps, most elements tight
        a, b = results[0][k].double(), results[1][k].double()
        assert float((a - b).abs().max()) <= 2.02 * 4e-4 + 1e-6, k
        assert float((a - b).abs().mean()) <= 1e-4, k
