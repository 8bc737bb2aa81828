"""The C-ABI library loads on a CPU-only box and exports every entry point include/gzsl_hip.h declares (no compute
calls here), and the ctypes binding table covers exactly that set."""
import ctypes
import os
import re

import pytest

from gzsl_text_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared():
    src = open(os.path.join(ROOT, 'include', 'gzsl_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(gzsl_[a-z0-9_]+)\s*\(', src)))


def test_header_and_binding_agree():
    assert declared() == sorted(hip.exported_names())


@pytest.mark.skipif(not os.path.exists(hip.LIB_PATH), reason='library not built (run __graft_entry__.build())')
def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(hip.LIB_PATH)
    for name in declared():
        assert hasattr(lib, name), f'{name} missing from {hip.LIB_PATH}'
    assert b'gfx950' in hip.load().gzsl_build_info()


def test_gemm_struct_layout_matches_header():
    # field order of struct GzslGemm in the header == ctypes mirror
    src = open(os.path.join(ROOT, 'include', 'gzsl_hip.h')).read()
    body = src[src.index('typedef struct GzslGemm {'):src.index('} GzslGemm;')]
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    names = []
    for decl in body.split('{', 1)[1].split(';'):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(','):
            names.append(re.findall(r'([A-Za-z_][A-Za-z0-9_]*)\s*$', part.strip())[0])
    assert names == [f[0] for f in hip.GzslGemm._fields_]
