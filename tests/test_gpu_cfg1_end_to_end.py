"""BASELINE.json configs[0] shapes end to end on the GPU through the C ABI (see tests/cfg1_flow.py)."""
import pytest

from tests import cfg1_flow

pytestmark = pytest.mark.gpu


def test_train_epoch_then_dev_eval_matches_oracle(tmp_path):
    cfg1_flow.run(tmp_path, 'cuda')
