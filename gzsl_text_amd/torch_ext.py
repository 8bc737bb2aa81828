"""The TORCH_LIBRARY binding of the hot path's operators (csrc/torch_ext.cpp): `torch.ops.gzsl.label_attention`,
`torch.ops.gzsl.label_logits`, `torch.ops.gzsl.ldam_bce` -- differentiable, callable from the reference's unchanged
modules.  A second binding of libgzsl_hip.so beside the ctypes one (gzsl_text_amd/hip.py).

build() compiles it in-tree with g++ (host code only; needs the torch headers of this image); load() registers the
operators.  The extension is optional for the module mirrors (they bind the C ABI with ctypes) and raises if it was not
built.
"""
import os
import subprocess

_DIR = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_DIR, 'csrc', 'torch_ext.cpp')
LIB_PATH = os.path.join(_DIR, 'lib', 'libgzsl_torch.so')
_loaded = False


def build(force=False):
    import torch
    from . import hip
    deps = [SRC, os.path.join(_DIR, '..', 'include', 'gzsl_hip.h')]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
        return LIB_PATH
    tdir = os.path.dirname(torch.__file__)
    cmd = ['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-D__HIP_PLATFORM_AMD__=1', '-DUSE_ROCM=1',
           f'-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}',
           f'-I{tdir}/include', f'-I{tdir}/include/torch/csrc/api/include', '-I/opt/rocm/include', SRC, '-o', LIB_PATH,
           f'-L{tdir}/lib', '-ltorch', '-ltorch_cpu', '-lc10', '-lc10_hip', '-ltorch_hip',
           f'-L{os.path.dirname(hip.LIB_PATH)}', '-lgzsl_hip', '-Wl,-rpath,$ORIGIN', f'-Wl,-rpath,{tdir}/lib']
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    subprocess.run(cmd, check=True)
    return LIB_PATH


def load():
    """Registers torch.ops.gzsl.* (idempotent).  Raises when the extension has not been built."""
    global _loaded
    if _loaded:
        return
    import torch
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f'{LIB_PATH} is missing: run `python -c "import __graft_entry__ as g; g.build()"`')
    torch.ops.load_library(LIB_PATH)
    _loaded = True
