"""numpy restatement of the in-kernel draw of the masking kernel (TEST INFRASTRUCTURE ONLY).

Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2,
3", SC'11; multipliers 0xD2511F53 / 0xCD9E8D57, Weyl constants 0x9E3779B9 / 0xBB67AE85).
This stream has no counterpart in the reference (which consumes torch's generator,
src/masking_strategy.py:110): parity of the fused-RNG mode is against this
specification only ("parity unpinned" with respect to the reference).

Token i uses word i&3 of block (counter = (i>>2, offset), key = seed); u = (x>>8)*2^-24.
The per-sequence t of sequence b is word 0 of the block with counter
(b, offset ^ bit31-of-low-word), t = fma(u, 1-eps, eps).
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint32).copy() for x in (c0, c1, c2, c3))
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ k0
            n1 = p1.astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ k1
            n3 = p0.astype(np.uint32)
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = np.uint32(k0 + W0)
            k1 = np.uint32(k1 + W1)
    return c0, c1, c2, c3


def _u01(x):
    return (x >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def token_draws(n, seed, offset):
    g = np.arange((n + 3) // 4, dtype=np.uint64)
    o0 = np.uint32(offset & 0x7FFFFFFF)
    o1 = np.uint32((offset >> 31) & 0xFFFFFFFF)
    w = philox4x32_10((g & np.uint64(0xFFFFFFFF)).astype(np.uint32), (g >> np.uint64(32)).astype(np.uint32),
                      np.full(g.shape, o0), np.full(g.shape, o1), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    return _u01(np.stack(w, axis=1).reshape(-1))[:n]


def seq_t(batch, seed, offset, eps):
    b = np.arange(batch, dtype=np.uint64)
    o0 = np.uint32((offset & 0x7FFFFFFF) ^ 0x80000000)
    o1 = np.uint32((offset >> 31) & 0xFFFFFFFF)
    w = philox4x32_10((b & np.uint64(0xFFFFFFFF)).astype(np.uint32), (b >> np.uint64(32)).astype(np.uint32),
                      np.full(b.shape, o0), np.full(b.shape, o1), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u = _u01(w[0]).astype(np.float64)
    # the kernel evaluates fmaf(u, 1-eps, eps) in fp32 with a single rounding
    one_m = np.float64(np.float32(1.0) - np.float32(eps))
    return np.float32(u * one_m + np.float64(np.float32(eps)))


def synth_tokens(batch, seq, id_lo, id_hi, bos, eos, pad, min_len, seed, offset):
    """The synthetic batch of ldm_synth_tokens: (ids int64 [batch, seq], mask int64 [batch, seq])."""
    n = batch * seq
    g = np.arange((n + 3) // 4, dtype=np.uint64)
    o0 = np.uint32(offset & 0x7FFFFFFF)
    o1 = np.uint32((offset >> 31) & 0xFFFFFFFF)
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    w = philox4x32_10((g & np.uint64(0xFFFFFFFF)).astype(np.uint32), (g >> np.uint64(32)).astype(np.uint32),
                      np.full(g.shape, o0), np.full(g.shape, o1), k0, k1)
    words = np.stack(w, axis=1).reshape(-1)[:n].astype(np.uint64)
    ids = (np.int64(id_lo) + ((words * np.uint64(id_hi - id_lo)) >> np.uint64(32)).astype(np.int64)).reshape(batch, seq)
    min_len = min(max(min_len, 2 if seq >= 2 else min_len), seq)
    if min_len >= seq:
        lens = np.full(batch, seq, dtype=np.int64)
    else:
        b = np.arange(batch, dtype=np.uint64)
        wl = philox4x32_10((b & np.uint64(0xFFFFFFFF)).astype(np.uint32), (b >> np.uint64(32)).astype(np.uint32),
                           np.full(b.shape, np.uint32(o0 ^ np.uint32(0x80000000))), np.full(b.shape, o1), k0, k1)[0]
        lens = min_len + ((wl.astype(np.uint64) * np.uint64(seq - min_len + 1)) >> np.uint64(32)).astype(np.int64)
    col = np.arange(seq, dtype=np.int64)[None, :]
    ids[:, 0] = bos
    ids[np.arange(batch), lens - 1] = np.where(lens - 1 == 0, ids[np.arange(batch), lens - 1], eos)
    ids = np.where(col >= lens[:, None], np.int64(pad), ids)
    return ids, (col < lens[:, None]).astype(np.int64)
