"""CPU restatement of the kernel's counter-based noise (include/flowfusion_amd.h, "in-kernel noise"):
Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) and
the Box-Muller pairing.  Test infrastructure only.  The reference draws its noise with torch's
generator (diffusion.py:554); this stream is an extension of the product, so what is pinned here is
the published algorithm (its known-answer vectors) and the header's definition of the mapping."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter, key):
    """counter: uint32 array [..., 4]; key: (k0, k1) python ints.  Returns uint32 [..., 4]."""
    c = [counter[..., i].astype(np.uint64) for i in range(4)]
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return np.stack(c, axis=-1).astype(np.uint32)


def box_muller(a, b):
    """Two uint32 words -> two standard normals, float32 arithmetic as the header defines it."""
    u1 = (a >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24) + np.float32(2.0 ** -25)
    u2 = (b >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    rad = np.sqrt(np.float32(-2.0) * np.log(u1.astype(np.float64))).astype(np.float32)
    ang = 2.0 * np.pi * u2.astype(np.float64)
    return (rad * np.cos(ang)).astype(np.float32), (rad * np.sin(ang)).astype(np.float32)


def normals(seed, sample_offset, batch, dim, noise_indices):
    """[len(noise_indices), batch, dim] float32: the normals the kernel uses for rows
    sample_offset .. sample_offset+batch-1."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    nblk = (dim + 3) // 4
    g = np.arange(batch, dtype=np.uint64) + np.uint64(sample_offset)
    out = np.empty((len(noise_indices), batch, nblk * 4), dtype=np.float32)
    for i, n in enumerate(noise_indices):
        ctr = np.empty((batch, nblk, 4), dtype=np.uint32)
        ctr[..., 0] = (g & MASK).astype(np.uint32)[:, None]
        ctr[..., 1] = (g >> np.uint64(32)).astype(np.uint32)[:, None]
        ctr[..., 2] = np.uint32(n)
        ctr[..., 3] = np.arange(nblk, dtype=np.uint32)[None, :]
        w = philox4x32_10(ctr, (seed & 0xFFFFFFFF, seed >> 32))
        z0, z1 = box_muller(w[..., 0], w[..., 1])
        z2, z3 = box_muller(w[..., 2], w[..., 3])
        out[i] = np.stack([z0, z1, z2, z3], axis=-1).reshape(batch, nblk * 4)
    return out[:, :, :dim]
