"""Oracle of the device noise generator (include/uad_hip.h: uad_rng_fill) -- TEST INFRASTRUCTURE ONLY.

Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; the generator TensorFlow's own random ops are built
on) restated in numpy.  The reference draws eps / dropout masks inside the TF graph, unseeded (models/variational_autoencoder.py:34),
so there is nothing to be bit-compatible WITH; what this pins is the device kernel's own contract: the counter layout
(element quad, global sample index, step, stream id) -> rank-count invariance, the 24-bit uniforms, Box-Muller, nn.dropout's
keep rule u >= rate with 1/(1-rate) scaling."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over uint32 counter arrays; returns four uint32 arrays.  Known-answer (Random123 kat_vectors): counter 0 / key 0 ->
    6627e8d5 e169c58d bc57ac4c 9b00dbd8."""
    c0, c1, c2, c3 = (np.asarray(c, np.uint64) & MASK for c in (c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)
        c0, c1, c2, c3 = n0 & MASK, p1 & MASK, n2 & MASK, p0 & MASK
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def _raw(n, per_sample, seed, step, sample0, stream):
    quads = (per_sample + 3) // 4
    e4 = np.tile(np.arange(quads, dtype=np.uint64), n)
    gs = np.repeat(np.arange(n, dtype=np.uint64) + np.uint64(sample0), quads)
    c3 = ((np.uint64(step >> 32) << np.uint64(8)) ^ ((gs >> np.uint64(32)) << np.uint64(16)) ^ np.uint64(stream)) & MASK
    r = philox4x32_10(e4, gs & MASK, np.full(e4.shape, step & 0xFFFFFFFF, np.uint64), c3, seed & 0xFFFFFFFF, seed >> 32)
    return np.stack(r, axis=1).reshape(n, quads * 4)          # [n, quads*4] uint32, element e of a sample = word e


def uniform24(r):
    return (r >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)


def normal(n, per_sample, seed, step=0, sample0=0, stream=0):
    r = _raw(n, per_sample, seed, step, sample0, stream).reshape(n, -1, 2)
    u1 = ((r[..., 0] >> np.uint32(8)).astype(np.float32) + np.float32(1.0)) * np.float32(2.0 ** -24)
    u2 = uniform24(r[..., 1])
    rad = np.sqrt(np.float32(-2.0) * np.log(u1.astype(np.float64))).astype(np.float32)
    ang = (np.float32(6.283185307179586) * u2).astype(np.float64)
    out = np.stack([rad * np.cos(ang), rad * np.sin(ang)], axis=-1).astype(np.float32)
    return out.reshape(n, -1)[:, :per_sample]


def keep_mask(n, per_sample, rate, seed, step=0, sample0=0, stream=0):
    u = uniform24(_raw(n, per_sample, seed, step, sample0, stream))[:, :per_sample]
    return np.where(u >= np.float32(rate), np.float32(1.0) / (np.float32(1.0) - np.float32(rate)), np.float32(0.0)).astype(np.float32)
