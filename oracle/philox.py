"""Philox4x32-10 in numpy: restatement of the generator ar-vae_amd/csrc/rng.h implements (test infrastructure; see
oracle/__init__.py).  Algorithm: Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3" (SC'11); the
known-answer vectors in tests/test_oracle_golden.py are those of the Random123 distribution (kat_vectors, philox4x32 10).

Role in the path: the reference draws eps with torch's global generator (imagevae/mnist_vae.py:79,
measurevae/measure_vae.py:116) and its dropout masks inside nn.Dropout (mnist_vae.py:16-47); parity runs inject explicit
eps / masks, throughput runs draw them on the device with this generator."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter, key):
    """counter (..., 4) uint32, key (2,) uint32 -> (..., 4) uint32."""
    c = np.asarray(counter, dtype=np.uint32).copy()
    k0, k1 = np.uint32(key[0]), np.uint32(key[1])
    with np.errstate(over='ignore'):
        for _ in range(10):
            p0 = M0 * c[..., 0].astype(np.uint64)
            p1 = M1 * c[..., 2].astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & MASK).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & MASK).astype(np.uint32)
            c = np.stack([hi1 ^ c[..., 1] ^ k0, lo1, hi0 ^ c[..., 3] ^ k1, lo0], axis=-1)
            k0 = np.uint32(k0 + W0)
            k1 = np.uint32(k1 + W1)
    return c


def blocks(count, seed, offset=0, step=0):
    """the library's block for element indices 0..count-1: counter = (index lo, index hi, offset, step), key = seed"""
    idx = np.arange(count, dtype=np.uint64)
    ctr = np.stack([(idx & MASK).astype(np.uint32), (idx >> np.uint64(32)).astype(np.uint32),
                    np.full(count, offset, np.uint32), np.full(count, step, np.uint32)], axis=-1)
    return philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))


def unit(x):
    """(x + 1) * 2^-32 in float32, u in (0, 1]"""
    return (x.astype(np.float32) + np.float32(1.0)) * np.float32(2.3283064365386963e-10)


def normal(count, seed, offset=0, step=0):
    """one N(0, 1) draw per element: Box-Muller on the first two words of the element's block (float32)"""
    b = blocks(count, seed, offset, step)
    r = np.sqrt(np.float32(-2.0) * np.log(unit(b[:, 0])))
    return (r * np.cos(np.float32(6.283185307179586) * unit(b[:, 1]))).astype(np.float32)


def keep_mask(count, keep_prob, seed, offset=0, step=0):
    """uint8 keep-mask: byte j of block b (little-endian words) keeps element 16 b + j when the byte < keep_prob * 256"""
    nb = (count + 15) // 16
    by = blocks(nb, seed, offset, step).astype('<u4').view(np.uint8).reshape(-1)[:count]
    return (by.astype(np.int32) < int(keep_prob * 256.0 + 0.5)).astype(np.uint8)
