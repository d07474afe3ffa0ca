"""Host replay of libmdm_hip's counter-based normal generator -- TEST INFRASTRUCTURE.

The device kernels (ml-mdm_amd/csrc/diffusion_ops.hip: normal4) draw element i of a tensor as lane ``i & 3`` of
Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) with
counter = (offset + i // 4, stream, 0) and key = seed, followed by Box-Muller on the two uint32 pairs.
The reference draws its noise with ``torch.randn_like`` on the device (samplers.py:241, 340); this generator is what
makes "device RNG replayable from a CPU seed"  testable.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3))
    k0, k1 = np.uint32(k0), np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & MASK).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0, k1 = np.uint32(k0 + W0), np.uint32(k1 + W1)
    return c0, c1, c2, c3


def normals(n, seed, offset=0, stream=0):
    """the n (multiple of 4) standard normals a kernel draws with rng state (seed, offset) on stream id `stream`"""
    assert n % 4 == 0
    blk = np.arange(n // 4, dtype=np.uint64) + np.uint64(offset)
    c = philox4x32_10((blk & MASK).astype(np.uint32), (blk >> np.uint64(32)).astype(np.uint32),
                      np.full(n // 4, stream, np.uint32), np.zeros(n // 4, np.uint32),
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    out = np.empty((n // 4, 4), np.float32)
    for h in range(2):
        u1 = (c[2 * h].astype(np.float32) + np.float32(1.0)) * np.float32(2.3283064365386963e-10)
        u2 = c[2 * h + 1].astype(np.float32) * np.float32(2.3283064365386963e-10)
        r = np.sqrt(np.float32(-2.0) * np.log(u1))
        ang = np.float32(6.283185307179586) * u2
        out[:, 2 * h], out[:, 2 * h + 1] = r * np.cos(ang), r * np.sin(ang)
    return out.reshape(-1)
