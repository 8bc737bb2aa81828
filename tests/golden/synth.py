"""The seeded synthetic inputs live in the package (gzsl_text_amd/utils/synth.py: bench.py and smoke() use them too);
this name is kept for the golden generator and the tests."""
from gzsl_text_amd.utils.synth import *  # noqa: F401,F403
from gzsl_text_amd.utils.synth import BASE_CASES, DROP_SEED, GAN_TRAIN_CASES, _uniform  # noqa: F401
