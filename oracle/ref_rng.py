"""oracle/ref_rng.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy restatement of the per-shape random streams (bdm_amd/csrc/rng_ops.hip, bdm_amd/rng.py): Philox4x32-10 (Salmon et al.,
"Parallel random numbers: as easy as 1, 2, 3", SC'11; known-answer vectors of the Random123 distribution are checked in
tests/test_rng.py), the splitmix64 key derivation, the 24-bit uniform and the Box-Muller transform.  The reference has no
counterpart (it draws from torch's global generators): this pins OUR definition, bit-exact for the integer stages.
"""
import numpy as np

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
MASK64 = (1 << 64) - 1


def philox4x32_10(counter, key):
    """counter (..., 4) uint32, key (..., 2) uint32 -> (..., 4) uint32."""
    c = [np.asarray(counter[..., i], dtype=np.uint64) for i in range(4)]
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    m32 = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = np.uint64(M0) * c[0], np.uint64(M1) * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & m32, p1 >> np.uint64(32), p1 & m32
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
        k0, k1 = (k0 + np.uint64(W0)) & m32, (k1 + np.uint64(W1)) & m32
    return np.stack(c, axis=-1).astype(np.uint32)


def shape_key(seed, shape_index):
    z = (int(seed) * 0x9E3779B97F4A7C15 + int(shape_index) * 0xBF58476D1CE4E5B9 + 0x94D049BB133111EB) & MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def words(key, per_shape, draw, purpose):
    """(ceil(per_shape / 4), 4) uint32 Philox outputs of one shape's draw."""
    nblk = (per_shape + 3) // 4
    ctr = np.zeros((nblk, 4), dtype=np.uint32)
    ctr[:, 0] = np.arange(nblk, dtype=np.uint64) & 0xFFFFFFFF
    ctr[:, 1] = np.arange(nblk, dtype=np.uint64) >> 32
    ctr[:, 2], ctr[:, 3] = draw, purpose
    return philox4x32_10(ctr, (key & 0xFFFFFFFF, key >> 32))


def bits(key, per_shape, draw, purpose):
    return (words(key, per_shape, draw, purpose).reshape(-1)[:per_shape] & 1).astype(np.int64)


def normal(key, per_shape, draw, purpose):
    """float64 evaluation of the transform on the SAME float32 uniforms the kernel forms."""
    w = words(key, per_shape, draw, purpose)
    u = ((w >> 8).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)
    out = np.empty_like(u, dtype=np.float64)
    for h in range(2):
        u1, u2 = u[:, 2 * h].astype(np.float64), u[:, 2 * h + 1].astype(np.float64)
        rad = np.sqrt(-2.0 * np.log(u1))
        ang = np.float64(np.float32(6.283185307179586)) * u2
        out[:, 2 * h], out[:, 2 * h + 1] = rad * np.cos(ang), rad * np.sin(ang)
    return out.reshape(-1)[:per_shape]
