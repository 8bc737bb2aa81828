"""gzsl_text_amd: MI355X-native hot path of petuum/gzsl_text.

Layout:
  csrc/        hand-written HIP kernels (gfx950) + the C ABI (include/gzsl_hip.h) -> lib/libgzsl_hip.so
  hip.py       ctypes binding of that ABI (fails loudly when the library is missing; no CPU fallback)
  ops_hip.py   tensor-level ops, each one or two kernel launches
  engine.py    forward/backward orchestration of the base feature extractor
  gan_engine.py  WGAN-GP critic / generator iterations
  parallel.py  label-dimension sharding over RCCL (one process per GPU)
  modules/     nn.Module mirrors of the reference's modules/ package (same classes, ctor signatures, state-dict keys)
"""
__version__ = '0.1.0'
