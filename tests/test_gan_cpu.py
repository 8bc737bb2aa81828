"""WGAN-GP step orchestration (gzsl_text_amd/gan_engine.py) on CPU through the test stand-in kernels, against the
golden vectors captured from the reference's generator / critic / calc_gradient_penalty / keyword head + Adam."""
import pytest

from tests.gan_checks import GAN_DIMS, run_gan_step_check
from tests.ops_cpu import CpuOps


@pytest.mark.parametrize('name', list(GAN_DIMS))
def test_wgan_gp_step_orchestration(name):
    run_gan_step_check(name, CpuOps(), 'cpu')
