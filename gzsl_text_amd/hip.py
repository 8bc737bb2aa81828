"""ctypes binding of libgzsl_hip.so (the C ABI declared in include/gzsl_hip.h).

The library is the product's only compute path: importing this module without the built
library, or calling it with tensors that are not on a HIP device, raises -- there is no
CPU fallback (tests that need CPU arithmetic use tests/ops_cpu.py, never this package).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('GZSL_HIP_LIB') or os.path.join(_HERE, 'lib', 'libgzsl_hip.so')  # env: diagnostic builds

ACT_NONE, ACT_RELU, ACT_TANH, ACT_LRELU = 0, 1, 2, 3
EINVAL = 1001
ETOOBIG = 1002

c_void_p, c_int, c_ll, c_float, c_u32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_float, ctypes.c_uint32


class GzslGemm(ctypes.Structure):
    """Mirror of `struct GzslGemm` (include/gzsl_hip.h)."""
    _fields_ = [('A', c_void_p), ('B', c_void_p), ('C', c_void_p), ('bias', c_void_p),
                ('M', c_int), ('N', c_int), ('K', c_int),
                ('a_mb', c_int), ('a_mbs', c_ll), ('a_sm', c_int), ('a_sk', c_int),
                ('a_shift', c_int), ('a_span', ctypes.c_uint), ('k_inner', c_int), ('a_kbs', c_ll),
                ('b_sk', c_int), ('b_sn', c_int), ('b_kbs', c_ll),
                ('ldc', c_int), ('act', c_int), ('accumulate', c_int), ('alpha', c_float),
                ('split_k', c_int), ('k_per_split', c_int), ('slab_stride', c_ll),
                ('dot_w', c_void_p), ('dot_out', c_void_p), ('dot_period', c_int), ('dot_no_store', c_int)]


# name -> argtypes (every entry point include/gzsl_hip.h declares; tests check the export list against the header)
_PROTOS = {
    'gzsl_gemm': [ctypes.POINTER(GzslGemm), c_void_p],
    'gzsl_splitk_reduce': [c_void_p, c_ll, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int,
                           c_void_p],
    'gzsl_label_attn_fwd': [c_void_p] * 5 + [c_int] * 7 + [c_u32, c_u32, c_float, c_void_p],
    'gzsl_label_attn_fwd_bf16x6': [c_void_p] * 5 + [c_int] * 7 + [c_u32, c_u32, c_float, c_void_p, c_ll, c_void_p],
    'gzsl_label_attn_fwd_bf16x6_work_bytes': [c_int] * 4,
    'gzsl_attn_delta': [c_void_p, c_void_p, c_void_p, c_ll, c_int, c_void_p],
    'gzsl_label_attn_bwd': [c_void_p] * 9 + [c_int] * 8 + [c_u32, c_u32, c_float, c_void_p],
    'gzsl_label_attn_bwd_bf16x6': [c_void_p] * 9 + [c_int] * 8 + [c_u32, c_u32, c_float, c_void_p, c_ll, c_int, c_void_p],
    'gzsl_label_attn_bwd_bf16x6_dq_reduce': [c_void_p, c_void_p] + [c_int] * 5 + [c_void_p],
    'gzsl_label_attn_bwd_bf16x6_work_bytes': [c_int] * 5,
    'gzsl_label_attn_bwd_det': [c_void_p] * 9 + [c_int] * 8 + [c_u32, c_u32, c_float, c_void_p, c_ll, c_void_p],
    'gzsl_label_attn_bwd_det_work_floats': [c_int] * 6,
    'gzsl_set_deterministic': [c_int],
    'gzsl_emb_gather_fwd': [c_void_p] * 4 + [c_ll, c_int, c_ll, c_u32, c_u32, c_float, c_void_p],
    'gzsl_emb_scatter_add': [c_void_p] * 4 + [c_ll, c_int, c_ll, c_ll, c_u32, c_u32, c_float, c_void_p],
    'gzsl_emb_scatter_add_grouped': [c_void_p] * 4 + [c_ll, c_int, c_ll, c_ll, c_u32, c_u32, c_float, c_void_p],
    'gzsl_label_avg_fwd': [c_void_p] * 4 + [c_int] * 3 + [c_void_p],
    'gzsl_label_avg_bwd': [c_void_p] * 4 + [c_int] * 3 + [c_ll, c_void_p],
    'gzsl_spmm_csr': [c_void_p] * 7 + [c_int] * 3 + [c_void_p],
    'gzsl_gru_gate_fwd': [c_void_p] * 4 + [c_int, c_int, c_void_p],
    'gzsl_gru_gate_bwd': [c_void_p] * 7 + [c_int, c_int, c_void_p],
    'gzsl_gru_seq_step': [c_void_p, c_ll, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p],
    'gzsl_logits_fwd': [c_void_p] * 3 + [c_int] * 3 + [c_void_p],
    'gzsl_logits_bwd': [c_void_p] * 5 + [c_int] * 4 + [c_void_p],
    'gzsl_logits_bwd_delta': [c_void_p] * 7 + [c_int] * 3 + [c_void_p],
    'gzsl_ldam_bce': [c_void_p] * 3 + [c_float, c_float] + [c_void_p] * 3 + [c_int, c_int, c_void_p],
    'gzsl_forward_final': [c_void_p] * 4 + [c_float, c_float] + [c_void_p] * 3 + [c_int, c_int, c_void_p],
    'gzsl_adamw_nobias_step': [c_void_p] * 4 + [c_ll] + [c_float] * 5 + [c_void_p],
    'gzsl_act_bwd': [c_void_p, c_void_p, c_ll, c_int, c_void_p],
    'gzsl_colsum': [c_void_p, c_void_p, c_ll, c_int, c_int, c_void_p],
    'gzsl_colsum_ws': [c_void_p, c_void_p, c_ll, c_int, c_int, c_void_p, c_ll, c_void_p],
    'gzsl_colsum_static': [c_void_p, c_void_p, c_ll, c_int, c_int, c_void_p],
    'gzsl_multihot': [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p],
    'gzsl_eval_counts': [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    'gzsl_conv_weight_pack': [c_void_p] * 3 + [c_int] * 3 + [c_void_p],
    'gzsl_gp_interp': [c_void_p] * 4 + [c_int, c_int, c_void_p],
    'gzsl_gp_slope_v': [c_void_p] * 4 + [c_int, c_int, c_void_p],
    'gzsl_gp_rownorm': [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p],
    'gzsl_gp_dw2': [c_void_p] * 3 + [c_int, c_int, c_void_p],
    'gzsl_rowdot': [c_void_p] * 4 + [c_int, c_int, c_int, c_void_p],
    'gzsl_weighted_sum': [c_void_p, c_void_p, c_float, c_void_p, c_int, c_void_p],
    'gzsl_rowdot_bwd': [c_void_p] * 6 + [c_int, c_int, c_int, c_void_p],
    'gzsl_keyword_ce': [c_void_p] * 5 + [c_int] * 3 + [c_void_p],
    'gzsl_keyword_ce_sim': [c_void_p] * 4 + [c_float, c_float, c_void_p, c_void_p] + [c_int] * 3 + [c_void_p],
    'gzsl_gan_critic_head': [c_void_p] * 11 + [c_int, c_int, c_void_p],
    'gzsl_gan_gen_head': [c_void_p] * 7 + [c_int, c_int, c_void_p],
    'gzsl_adam_flat_step': [c_void_p] * 4 + [c_ll] + [c_float] * 4 + [c_void_p, c_void_p],
}

_lib = None


class HipLibraryMissing(RuntimeError):
    pass


def load():
    """Load the library (once).  Raises HipLibraryMissing when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            f'(or `make -C gzsl_text_amd/csrc`). gzsl_text_amd has no CPU fallback.')
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in _PROTOS.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = c_ll if name in ('gzsl_label_attn_bwd_det_work_floats', 'gzsl_label_attn_fwd_bf16x6_work_bytes',
                                        'gzsl_label_attn_bwd_bf16x6_work_bytes') else c_int
    lib.gzsl_build_info.restype = ctypes.c_char_p
    lib.gzsl_build_info.argtypes = []
    _lib = lib
    return lib


def exported_names():
    return sorted(_PROTOS) + ['gzsl_build_info']


def _current_device():
    return torch.cuda.current_device()


def dptr(t):
    """Device pointer of a HIP tensor or strided view (None -> NULL).

    Launches go to the CURRENT device's current stream (`stream()`), so a tensor that lives on another device would be
    dereferenced by the wrong GPU: raise instead (a model moved with `.to('cuda:1')` needs `torch.cuda.set_device(1)`
    or a `with torch.cuda.device(1):` block around its calls, as one-process-per-GPU launchers do)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('gzsl_text_amd kernels need tensors on a HIP device (got a CPU tensor); there is no CPU path')
    cur = _current_device()
    if t.device.index != cur:
        raise RuntimeError(f'tensor lives on cuda:{t.device.index} but the current device is cuda:{cur}: kernels launch '
                           f'on the current device\'s stream (call torch.cuda.set_device({t.device.index}) first)')
    return t.data_ptr()


def ptr(t):
    """Device pointer of a contiguous HIP tensor on the current device (None -> NULL)."""
    if t is not None and t.is_cuda and not t.is_contiguous():
        raise RuntimeError('gzsl_text_amd kernels need contiguous tensors')
    return dptr(t)


def stream():
    return torch.cuda.current_stream().cuda_stream


def check(rc, what):
    if rc != 0:
        if rc == EINVAL:
            raise RuntimeError(f'{what}: unsupported arguments (GZSL_EINVAL)')
        if rc == ETOOBIG:
            raise RuntimeError(f'{what}: size beyond the entry point\'s stated maximum (GZSL_ETOOBIG, include/gzsl_hip.h)')
        raise RuntimeError(f'{what}: HIP error {rc}')


def call(name, *args):
    lib = load()
    check(getattr(lib, name)(*args, stream()), name)
