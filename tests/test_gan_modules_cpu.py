"""Module-level GAN API (generative_module / decoder_module / lm_module mirrors) on the CPU stand-in: the call
sequence of train_gan.py:228-307 through the nn.Module surface, against the reference's golden vectors."""
import torch

from gzsl_text_amd.modules.lm_module import LabelEncodingCache, RNNLabelEncoder
from tests.conftest import load_golden
from tests.helpers import close
from tests.ops_cpu import CpuOps


def test_module_api_follows_train_gan_call_order():
    from tests.gan_checks import run_gan_module_api_check
    run_gan_module_api_check(CpuOps(), 'cpu')


def test_label_rnn_mirror_and_cache_match_reference():
    g = load_golden('label_rnn_tiny')
    enc = RNNLabelEncoder(g['word_emb'].copy())
    for k, p in enc.rnn.named_parameters():
        p.data.copy_(torch.from_numpy(g['rnn.' + k]))
    assert {'encoder.weight', 'decoder.weight', 'rnn.weight_ih_l0', 'rnn.weight_hh_l0', 'rnn.bias_ih_l0',
            'rnn.bias_hh_l0'} == set(enc.state_dict())
    ids, m = torch.from_numpy(g['ids']), torch.from_numpy(g['mask'])
    with torch.no_grad():
        for mode in ('max', 'mean'):
            close(enc.forward_enc(ids, m, None, training=False, pool_mode=mode), g['pool_' + mode], 1e-5, 1e-6, mode)
    cache = LabelEncodingCache(enc, ids, m, trunc=20, pool_mode='max', chunk=4)
    sel = torch.tensor([3, 0, 3, 10])
    close(cache(sel), g['pool_max'][sel.numpy()], 1e-5, 1e-6, 'cached label encoding')
    # the packed-GRU orchestration the HIP path uses (gather, one input GEMM, per-step GEMM + masked gate/pool step)
    from gzsl_text_amd.modules.lm_module import forward_enc_hip
    from tests.ops_cpu import CpuOps
    for mode in ('max', 'mean'):
        close(forward_enc_hip(enc, ids, m, mode, CpuOps()), g['pool_' + mode], 1e-5, 1e-6, 'packed GRU steps ' + mode)
    cache2 = LabelEncodingCache(enc, ids, m, trunc=20, pool_mode='max', chunk=4, ops=CpuOps())
    close(cache2(sel), g['pool_max'][sel.numpy()], 1e-5, 1e-6, 'cache filled through the step kernels')


def test_plain_wgan_step_without_sim_or_keyword_head():
    from tests.gan_checks import run_gan_plain_wgan_check
    run_gan_plain_wgan_check(CpuOps(), 'cpu')


def test_full_step_matches_per_iteration_steps():
    from tests.gan_checks import run_gan_full_step_check
    run_gan_full_step_check(CpuOps(), 'cpu')
