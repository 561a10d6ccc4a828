"""Philox4x32-10 counter RNG and the inverted-dropout arithmetic built on it  --  TEST INFRASTRUCTURE ONLY.

The reference's dropout is torch's `nn.Dropout` (timit/models/model_ctc.py:26,34 in BatchRNN, :58,67 in LayerCNN):
y = x * m / (1 - p), m ~ Bernoulli(1 - p), train mode only.  Which generator draws `m` is not part of
the reference's contract (torch's CPU and GPU generators already disagree), so the HIP path uses its own counter-based
stream: Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11), keyed by a
64-bit seed, one 128-bit block per group of four consecutive elements.

This file restates (a) the published generator -- pinned by the known-answer vectors of the Random123 distribution
(`KAT`, checked by tests/test_oracle_golden.py) -- and (b) how `ctcn_dropout(x, y, n, p, seed, offset)` maps a block to a
keep mask and applies it, so that the GPU test can hold the kernel bit-exact:

    block g = element index // 4      counter = (lo32(offset + g), hi32(offset + g), 0, 0)     key = (lo32(seed), hi32(seed))
    u(i)    = (word[i % 4] >> 8) * 2^-24                              (24-bit uniform in [0, 1), exact in float32)
    keep(i) = u(i) >= float32(p)
    y(i)    = keep(i) ? float32(x(i) * scale) : 0,   scale = float32(1) / (float32(1) - float32(p))
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85            # Weyl key increments (golden ratio, sqrt(3) - 1)
MASK = np.uint64(0xFFFFFFFF)

# Random123 kat_vectors, "philox4x32 10": (counter[4], key[2]) -> output[4]
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over numpy arrays of uint32 counters; scalar (python int) keys.  Returns four uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & MASK for c in (c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2                       # 32 x 32 -> 64-bit products
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def dropout_words(n, seed, offset):
    """The uint32 word element i of a ctcn_dropout(n, seed, offset) call consumes, for i in [0, n)."""
    groups = (n + 3) // 4
    ctr = np.uint64(offset) + np.arange(groups, dtype=np.uint64)
    zero = np.zeros(groups, dtype=np.uint64)
    w = philox4x32_10(ctr & MASK, ctr >> np.uint64(32), zero, zero, int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF)
    return np.stack(w, axis=1).reshape(-1)[:n]


def dropout_mask(n, p, seed, offset):
    u = (dropout_words(n, seed, offset) >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return u >= np.float32(p)


def dropout(x, p, seed, offset):
    """y = x * m / (1 - p) exactly as the kernel rounds it (one float32 multiply by the float32 scale)."""
    x = np.asarray(x, dtype=np.float32)
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    keep = dropout_mask(x.size, p, seed, offset).reshape(x.shape)
    return np.where(keep, (x * scale).astype(np.float32), np.float32(0.0))
