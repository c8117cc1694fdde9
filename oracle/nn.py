"""Oracle primitives: TF-1.15 layer semantics restated in numpy (NHWC).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PINNED against the expectations
published in TensorFlow r1.15's own unit tests (tests/golden/tf_published.json,
tests/test_oracle_tf_published.py); every function cites the reference call site it
restates and the TF semantic it fixes (SURVEY.md §8a notes 1-8).

All functions are dtype-preserving: pass float64 arrays for the "truth" run,
float32 for the timed CPU baseline.
"""
import numpy as np


# --------------------------------------------------------------------------
# padding geometry
# --------------------------------------------------------------------------
def same_pads(size, k, s):
    """TF 'SAME' padding for one spatial dim: out = ceil(size/s); total pad =
    max((out-1)*s + k - size, 0); the extra (odd) element goes AFTER.
    (k5 s2 on an even size -> (1, 2); SURVEY.md §8a note 2)."""
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return out, total // 2, total - total // 2


# --------------------------------------------------------------------------
# Conv2D (tf.layers.Conv2D / keras Conv2D, padding='same')
#   models/customlayers.py:21 (k5 s2), :37 (1x1), variational_autoencoder.py:20-21
#   kernel layout HWIO = [kh, kw, Cin, Cout]
# --------------------------------------------------------------------------
def _geom(size, k, s, padding):
    """(out, pad_before, pad_after) of one spatial dim: 'SAME' as same_pads(); 'VALID' = no padding, out = (size - k)//s + 1
    (only the TF unit-test vectors of tests/golden/tf_published.json use VALID; every layer of the reference is 'same')."""
    if padding == 'SAME':
        return same_pads(size, k, s)
    if padding != 'VALID':
        raise ValueError(padding)
    return (size - k) // s + 1, 0, 0


def _strides(stride):
    return (stride, stride) if np.isscalar(stride) else tuple(stride)


def conv2d_fwd(x, w, b, stride, padding='SAME'):
    n, h, wd, cin = x.shape
    kh, kw, _, cout = w.shape
    sh, sw = _strides(stride)
    oh, pt, pb = _geom(h, kh, sh, padding)
    ow, pl, pr = _geom(wd, kw, sw, padding)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    out = np.zeros((n, oh, ow, cout), dtype=x.dtype)
    for ky in range(kh):
        for kx in range(kw):
            patch = xp[:, ky:ky + sh * oh:sh, kx:kx + sw * ow:sw, :]
            out += (patch.reshape(-1, cin) @ w[ky, kx]).reshape(n, oh, ow, cout)
    if b is not None:
        out += b
    return out


def conv2d_bwd(x, w, g, stride, padding='SAME'):
    """Returns (dx, dw, db) for out = conv2d_fwd(x, w, b, stride), g = dL/dout."""
    n, h, wd, cin = x.shape
    kh, kw, _, cout = w.shape
    sh, sw = _strides(stride)
    oh, pt, pb = _geom(h, kh, sh, padding)
    ow, pl, pr = _geom(wd, kw, sw, padding)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    dxp = np.zeros_like(xp)
    dw = np.zeros_like(w)
    g2 = g.reshape(-1, cout)
    for ky in range(kh):
        for kx in range(kw):
            sl = (slice(None), slice(ky, ky + sh * oh, sh),
                  slice(kx, kx + sw * ow, sw), slice(None))
            dw[ky, kx] = xp[sl].reshape(-1, cin).T @ g2
            dxp[sl] += (g2 @ w[ky, kx].T).reshape(n, oh, ow, cin)
    dx = dxp[:, pt:pt + h, pl:pl + wd, :]
    db = g2.sum(axis=0)
    return dx, dw, db


# --------------------------------------------------------------------------
# Conv2DTranspose (tf.layers.Conv2DTranspose, padding='same')
#   models/customlayers.py:34;  kernel layout [kh, kw, Cout, Cin]
#   == gradient-w.r.t.-input of a SAME conv from the (s*H) image to the H image:
#   y[n, s*i - pt + ky, s*j - pl + kx, co] += x[n,i,j,ci] * w[ky,kx,co,ci]
#   (SURVEY.md §8a note 2: k5 s2 -> pt = pl = 1, output exactly 2x)
# --------------------------------------------------------------------------
def conv2d_transpose_fwd(x, w, b, stride):
    n, h, wd, cin = x.shape
    kh, kw, cout, _ = w.shape
    oh, ow = h * stride, wd * stride
    _, pt, pb = same_pads(oh, kh, stride)
    _, pl, pr = same_pads(ow, kw, stride)
    yp = np.zeros((n, oh + pt + pb, ow + pl + pr, cout), dtype=x.dtype)
    x2 = x.reshape(-1, cin)
    for ky in range(kh):
        for kx in range(kw):
            yp[:, ky:ky + stride * h:stride, kx:kx + stride * wd:stride, :] += \
                (x2 @ w[ky, kx].T).reshape(n, h, wd, cout)
    y = yp[:, pt:pt + oh, pl:pl + ow, :].copy()
    if b is not None:
        y += b
    return y


def conv2d_transpose_bwd(x, w, g, stride):
    """Returns (dx, dw, db) for y = conv2d_transpose_fwd(x, w, b, stride)."""
    n, h, wd, cin = x.shape
    kh, kw, cout, _ = w.shape
    oh, ow = h * stride, wd * stride
    _, pt, pb = same_pads(oh, kh, stride)
    _, pl, pr = same_pads(ow, kw, stride)
    gp = np.pad(g, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    dx = np.zeros_like(x)
    dw = np.zeros_like(w)
    x2 = x.reshape(-1, cin)
    for ky in range(kh):
        for kx in range(kw):
            gs = gp[:, ky:ky + stride * h:stride, kx:kx + stride * wd:stride, :].reshape(-1, cout)
            dx += (gs @ w[ky, kx]).reshape(n, h, wd, cin)
            dw[ky, kx] = gs.T @ x2
    db = g.reshape(-1, cout).sum(axis=0)
    return dx, dw, db


# --------------------------------------------------------------------------
# BatchNormalization (tensorflow.compat.v1.layers.BatchNormalization called
# WITHOUT training= -> inference mode with never-updated moving stats 0/1):
#   y = gamma * x / sqrt(1 + eps) + beta, eps = 1e-3
#   models/customlayers.py:22,30,35 ; SURVEY.md §8a note 1 / Appendix A1
# --------------------------------------------------------------------------
BN_EPS = 1e-3


def bn_frozen_fwd(x, gamma, beta, eps=BN_EPS):
    rstd = x.dtype.type(1.0 / np.sqrt(1.0 + eps))
    return x * (gamma * rstd) + beta


def bn_frozen_bwd(x, gamma, g, eps=BN_EPS):
    rstd = x.dtype.type(1.0 / np.sqrt(1.0 + eps))
    c = x.shape[-1]
    dgamma = (g * x).reshape(-1, c).sum(axis=0) * rstd
    dbeta = g.reshape(-1, c).sum(axis=0)
    dx = g * (gamma * rstd)
    return dx, dgamma, dbeta


# --------------------------------------------------------------------------
# LeakyReLU (keras default alpha = 0.3, customlayers.py:23,36) / ReLU (:31)
# TF LeakyReluGrad: features > 0 ? g : alpha * g
# --------------------------------------------------------------------------
def leaky_relu_fwd(x, alpha):
    return np.where(x > 0, x, x * x.dtype.type(alpha))


# Derivative side of a (Leaky)ReLU as ANOTHER fp32 implementation of the same step took it (tests only).  A pre-activation within round-off of
# 0 lands on either side of the kink depending on summation order; its derivative (alpha or 1) is then a property of that implementation, not
# an error, and one such element moves the small downstream gradient tensors by 1e-3..1e-2 of their max.  The GPU parity tests read the
# pattern the device used (sign of its stored activations), check that every disagreement with this oracle sits within round-off of the
# kink (tests/gpu_util.py: kink_overrides), and differentiate the oracle with the device's pattern: every gradient is then held to the
# 1e-4 bar whether or not flips occur.  The override is keyed by a fingerprint of the site's POST-activation array as this oracle computes
# it (bit-reproducible: the second run of a phase recomputes the same arrays), so no oracle needs per-site plumbing.
_ACT_OVERRIDE = None


def act_fingerprint(post):
    import hashlib
    a = np.ascontiguousarray(post + post.dtype.type(0))          # (+0 folds -0.0 into +0.0: relu as maximum(y, 0) and as where(y > 0, y, 0 * y) agree)
    return (a.shape, str(a.dtype), hashlib.blake2b(a.tobytes(), digest_size=16).digest())


class act_override:
    """with act_override({act_fingerprint(oracle post-activation): bool pattern}): ... -- leaky_relu_bwd uses the given patterns."""

    def __init__(self, table):
        self.table = table

    def __enter__(self):
        global _ACT_OVERRIDE
        self.prev, _ACT_OVERRIDE = _ACT_OVERRIDE, self.table
        self.used = set()
        self.table['_used'] = self.used
        return self

    def __exit__(self, *exc):
        global _ACT_OVERRIDE
        _ACT_OVERRIDE = self.prev
        return False


def leaky_relu_bwd(x, g, alpha):
    pos = x > 0
    if _ACT_OVERRIDE is not None:
        key = act_fingerprint(leaky_relu_fwd(x, alpha))
        o = _ACT_OVERRIDE.get(key)
        if o is not None:
            pos = o.reshape(x.shape)
            _ACT_OVERRIDE['_used'].add(key)
    return np.where(pos, g, g * x.dtype.type(alpha))


# --------------------------------------------------------------------------
# Dense (tf.layers.Dense): y = x @ W[in,out] + b  (variational_autoencoder.py:26-28)
# --------------------------------------------------------------------------
def dense_fwd(x, w, b):
    return x @ w + b


def dense_bwd(x, w, g):
    return g @ w.T, x.T @ g, g.sum(axis=0)


# --------------------------------------------------------------------------
# Dropout (keras Dropout(rate)(x, training)): inverted dropout; the Bernoulli
# keep-mask is an INPUT here (the TF graph RNG is unseeded), already
# scaled by 1/(1-rate):  y = x * mask
# --------------------------------------------------------------------------
def make_dropout_mask(rng, shape, rate, dtype=np.float32):
    keep = rng.random(shape) >= rate
    return (keep / (1.0 - rate)).astype(dtype)


# --------------------------------------------------------------------------
# Adam, TF-1.15 tf.train.AdamOptimizer form (trainers/DLMODEL.py:112-131):
#   lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t)
#   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr_t * m / (sqrt(v) + eps)
# ("epsilon-hat" formulation: eps is added to sqrt(v), not sqrt(v_hat))
# --------------------------------------------------------------------------
def adam_tf_step(p, g, m, v, t, lr, beta1=0.5, beta2=0.999, eps=1e-8):
    dt = p.dtype.type
    lr_t = dt(lr * np.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t))
    m[...] = dt(beta1) * m + dt(1.0 - beta1) * g
    v[...] = dt(beta2) * v + dt(1.0 - beta2) * (g * g)
    p[...] = p - lr_t * m / (np.sqrt(v) + dt(eps))
    return p, m, v


def sgd_tf_step(p, g, lr):
    """tf.train.GradientDescentOptimizer (trainers/DLMODEL.py:116-117), in place."""
    p -= lr * g


def momentum_tf_step(p, g, accum, lr, momentum=0.9):
    """tf.train.MomentumOptimizer (use_nesterov False, :118-119): accum = momentum accum + g ; p -= lr accum.  In place."""
    accum *= momentum
    accum += g
    p -= lr * accum


def rmsprop_tf_step(p, g, ms, mom, lr, momentum=0.9, decay=0.9, eps=1e-10):
    """tf.train.RMSPropOptimizer(learning_rate, momentum=momentum) (:120-121; decay 0.9, epsilon 1e-10, centered False; the `rms` slot
    starts at ONE): ms = decay ms + (1 - decay) g^2 ; mom = momentum mom + lr g / sqrt(ms + eps) ; p -= mom.  In place."""
    ms *= decay
    ms += (1.0 - decay) * g * g
    mom *= momentum
    mom += lr * g / np.sqrt(ms + eps)
    p -= mom


def glorot_uniform(rng, shape, dtype=np.float32):
    """keras glorot_uniform: fan_in/out computed with the receptive field
    (prod(shape[:-2])) for conv kernels; limit = sqrt(6/(fan_in+fan_out))."""
    if len(shape) == 2:
        fan_in, fan_out = shape
    else:
        rf = int(np.prod(shape[:-2]))
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(dtype)
