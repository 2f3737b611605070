"""Attention-probability dropout mask of the HIP kernels, restated in numpy.

TEST INFRASTRUCTURE ONLY.  The reference applies ``nn.Dropout`` to the softmax output
(src/transformer.py:278) with a mask drawn from torch's generator; the kernels instead derive the
mask from a 64-bit seed and the element's coordinates, so that forward and backward regenerate it
without storing it.  This file restates that function (``ldm_mix32`` / ``ldm_drop_bh`` /
``ldm_drop_row`` / ``ldm_drop_keep`` of the library) so that tests can build the dense mask and run
a plain torch attention with it.  Parity with the reference's own masks is statistical only
("parity unpinned" element by element): same keep probability, same 1/(1-p) scale.
"""
from __future__ import annotations

import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def mix32(h):
    h = np.asarray(h, dtype=np.uint64) & M32
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & M32
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & M32
    h ^= h >> np.uint64(16)
    return h


def threshold(p: float) -> int:
    """Keep threshold for drop probability p: floor(float32(p) * 2^32), capped at 2^32 - 1."""
    t = float(np.float32(p)) * 4294967296.0
    return 4294967295 if t >= 4294967295.0 else int(t)


def scale(p: float) -> float:
    """1 / (1 - p_effective) as the kernels apply it (fp32)."""
    return float(np.float32(4294967296.0 / (4294967296.0 - threshold(p))))


def keep_mask_rows(p: float, seed: int, rows: int, cols: int) -> np.ndarray:
    """bool [rows, cols]: the residual-dropout mask of ldm_dropout_add (batch*head slot 0xffffffff)."""
    return keep_mask(p, seed, 1, rows, cols, bh0=0xFFFFFFFF)[0]


def keep_mask(p: float, seed: int, n_bh: int, seq_q: int, seq_k: int, bh0: int = 0) -> np.ndarray:
    """bool [n_bh, seq_q, seq_k]: True where the probability of (batch*head, query, key) is kept."""
    thr = np.uint64(threshold(p))
    lo, hi = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    bh = (np.arange(n_bh, dtype=np.uint64) + np.uint64(bh0))[:, None, None]
    q = np.arange(seq_q, dtype=np.uint64)[None, :, None]
    k = np.arange(seq_k, dtype=np.uint64)[None, None, :]
    bhkey = mix32((lo + bh * np.uint64(0x9E3779B1)) & M32)
    rowkey = mix32(bhkey ^ ((hi + q * np.uint64(0x85EBCA77)) & M32))
    return keep_step(rowkey, k) >= thr


def keep_step(rowkey, k):
    """ldm_drop_keep's value: one xorshift and one 24 x 24 -> 32 bit multiply of rowkey + k * C."""
    x = (np.asarray(rowkey, dtype=np.uint64) + np.asarray(k, dtype=np.uint64) * np.uint64(0x5ED85E17)) & M32
    a = (x ^ (x >> np.uint64(13))) & np.uint64(0xFFFFFF)
    return (a * np.uint64(0xCB5E35)) & M32
