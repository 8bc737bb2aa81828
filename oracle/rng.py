"""Counter-based dropout RNG shared (by definition, not by code) with the HIP kernels.

The reference uses torch's Philox stream for `nn.Dropout` (icd_modules.py:55,59,140,157).
A fused kernel cannot replay that stream, so the build defines its own counter hash,
keyed on GLOBAL coordinates so the mask does not depend on how labels are sharded:

    mix32(x):  x ^= x>>16; x *= 0x7feb352d; x ^= x>>15; x *= 0x846ca68b; x ^= x>>16
    row_key(seed, stream, row) = mix32( mix32(seed + stream*0x632BE5AB) ^ (row*0x9E3779B1) )
    u(row, col)                = mix32( row_key ^ (col*0x85EBCA6B) )
    keep(row, col)             = u >= floor(p * 2^32)

  stream 1 = embedding dropout  (row = b*T + t,        col = e)       p = 0.5
  stream 2 = attention dropout  (row = b*L_total + l,  col = t)       p = 0.2

All arithmetic is uint32 with wrap-around.  Kept values are scaled by 1/(1-p).
`gzsl_text_amd/csrc/rng.h` is the device statement of the same function.
"""
import numpy as np

_M1 = np.uint32(0x7FEB352D)
_M2 = np.uint32(0x846CA68B)
_STREAM = np.uint32(0x632BE5AB)
_ROW = np.uint32(0x9E3779B1)
_COL = np.uint32(0x85EBCA6B)

STREAM_EMB = 1
STREAM_ATTN = 2


def mix32(x):
    x = np.asarray(x, dtype=np.uint32).copy()
    with np.errstate(over="ignore"):
        x ^= x >> np.uint32(16)
        x *= _M1
        x ^= x >> np.uint32(15)
        x *= _M2
        x ^= x >> np.uint32(16)
    return x


def drop_threshold(p):
    return np.uint32(min(int(float(p) * 4294967296.0), 0xFFFFFFFF))


def row_key(seed, stream, rows):
    with np.errstate(over="ignore"):
        s = mix32(np.uint32(seed & 0xFFFFFFFF) + np.uint32(stream) * _STREAM)
        return mix32(s ^ (np.asarray(rows, dtype=np.uint32) * _ROW))


def keep_mask(seed, stream, rows, cols, p):
    """rows: uint32 array [R], cols: uint32 array [C] -> bool [R, C] (True = kept)."""
    rk = row_key(seed, stream, rows)[:, None]
    with np.errstate(over="ignore"):
        u = mix32(rk ^ (np.asarray(cols, dtype=np.uint32)[None, :] * _COL))
    return u >= drop_threshold(p)


def emb_dropout_scale(seed, B, T, E, p):
    """[B, T, E] float32 multiplier (0 or 1/(1-p)) for the embedding dropout."""
    if p <= 0.0:
        return np.ones((B, T, E), np.float32)
    rows = np.arange(B * T, dtype=np.uint32)
    keep = keep_mask(seed, STREAM_EMB, rows, np.arange(E, dtype=np.uint32), p)
    return (keep.astype(np.float32) * np.float32(1.0 / (1.0 - p))).reshape(B, T, E)


def attn_dropout_scale(seed, b_idx, l_global, L_total, t_idx, p):
    """[len(b_idx), len(l_global), len(t_idx)] float32 multiplier for attention dropout."""
    nb, nl, nt = len(b_idx), len(l_global), len(t_idx)
    if p <= 0.0:
        return np.ones((nb, nl, nt), np.float32)
    with np.errstate(over="ignore"):
        rows = (np.asarray(b_idx, np.uint32)[:, None] * np.uint32(L_total)
                + np.asarray(l_global, np.uint32)[None, :]).reshape(-1)
    keep = keep_mask(seed, STREAM_ATTN, rows, np.asarray(t_idx, np.uint32), p)
    return (keep.astype(np.float32) * np.float32(1.0 / (1.0 - p))).reshape(nb, nl, nt)
