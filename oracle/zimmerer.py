"""Oracle for the Zimmerer-style VAE: models/variational_autoencoder_Zimmerer.py:7-32 (k4 s2 convolutions 16-64-256-1024 with
tf.nn.leaky_relu (alpha 0.2), no normalisation, no dropout, Dense mu / log-sigma heads on the flattened map, Dense back to
[r, r, 1024], four k4 s2 transposed convolutions 1024-256-64-16, a k4 s1 convolution to one channel) under trainers/VAE.py:36-42
(loss = mean_n(sum |x - x_hat| + KL_n), z_sigma = exp(z_log_sigma)).  numpy forward, hand-written backward.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED (no TensorFlow here, no golden vectors in the reference);
tests/test_oracle_zimmerer.py anchors every gradient on torch autograd in float64.

The model opens no variable scope: convolutions carry their explicit names, the Dense layers are dense (mu), dense_1 (log sigma),
dense_2 (decoder) in first-call order."""
import numpy as np

from . import nn

ALPHA = 0.2                         # tf.nn.leaky_relu default
ENC_F = (16, 64, 256, 1024)
DEC_F = (1024, 256, 64, 16)


def param_spec(height=128, zdim=128, channels=1):
    r = height // 16
    spec, cin = [], channels
    for i, f in enumerate(ENC_F):
        spec += [(f'enc_conv2D_{i + 1}/kernel', (4, 4, cin, f), 'conv_w'), (f'enc_conv2D_{i + 1}/bias', (f,), 'bias')]
        cin = f
    flat = r * r * 1024
    spec += [('dense/kernel', (flat, zdim), 'dense_w'), ('dense/bias', (zdim,), 'bias'),
             ('dense_1/kernel', (flat, zdim), 'dense_w'), ('dense_1/bias', (zdim,), 'bias'),
             ('dense_2/kernel', (zdim, flat), 'dense_w'), ('dense_2/bias', (flat,), 'bias')]
    cin = 1024
    for i, f in enumerate(DEC_F):
        spec += [(f'dec_Conv2DT_{i + 1}/kernel', (4, 4, f, cin), 'conv_w'), (f'dec_Conv2DT_{i + 1}/bias', (f,), 'bias')]
        cin = f
    spec += [('dec_Conv2D_final/kernel', (4, 4, cin, channels), 'conv_w'), ('dec_Conv2D_final/bias', (channels,), 'bias')]
    return spec


def init_params(spec, seed=3, dtype=np.float64, perturb=True):
    rng = np.random.default_rng(seed)
    p = {}
    for name, shape, kind in spec:
        if kind in ('conv_w', 'dense_w'):
            rf = int(np.prod(shape[:-2])) if len(shape) == 4 else 1
            lim = np.sqrt(6.0 / (shape[-2] * rf + shape[-1] * rf))
            p[name] = rng.uniform(-lim, lim, shape).astype(dtype)
        else:
            p[name] = (0.05 * rng.standard_normal(shape) if perturb else np.zeros(shape)).astype(dtype)
    return p


class VAEZimmerer:
    def __init__(self, height=128, zdim=128):
        assert height % 16 == 0
        self.height, self.zdim, self.r = height, zdim, height // 16
        self.spec = param_spec(height, zdim)

    def forward(self, p, x, eps=None):
        n = x.shape[0]
        cache = {'a': [x], 'c': []}
        a = x
        for i in range(4):
            c = nn.conv2d_fwd(a, p[f'enc_conv2D_{i + 1}/kernel'], p[f'enc_conv2D_{i + 1}/bias'], 2)
            a = nn.leaky_relu_fwd(c, ALPHA)
            cache['c'].append(c); cache['a'].append(a)
        flat = a.reshape(n, -1)
        mu = nn.dense_fwd(flat, p['dense/kernel'], p['dense/bias'])
        ls = nn.dense_fwd(flat, p['dense_1/kernel'], p['dense_1/bias'])
        sigma = np.exp(ls)
        eps = np.zeros_like(mu) if eps is None else eps
        z = mu + eps * sigma
        dv = nn.dense_fwd(z, p['dense_2/kernel'], p['dense_2/bias'])
        g = dv.reshape(n, self.r, self.r, 1024)
        cache.update(flat=flat, mu=mu, ls=ls, sigma=sigma, eps=eps, z=z, ga=[g], gc=[])
        for i in range(4):
            c = nn.conv2d_transpose_fwd(g, p[f'dec_Conv2DT_{i + 1}/kernel'], p[f'dec_Conv2DT_{i + 1}/bias'], 2)
            g = nn.leaky_relu_fwd(c, ALPHA)
            cache['gc'].append(c); cache['ga'].append(g)
        xh = nn.conv2d_fwd(g, p['dec_Conv2D_final/kernel'], p['dec_Conv2D_final/bias'], 1)
        return {'x_hat': xh, 'z_mu': mu, 'z_log_sigma': ls, 'z_sigma': sigma, 'z': z}, cache

    def losses(self, x, out):
        """trainers/VAE.py:36-42."""
        l1 = np.abs(out['x_hat'] - x)
        rec = l1.reshape(x.shape[0], -1).sum(axis=1)
        mu, ls, sg = out['z_mu'], out['z_log_sigma'], out['z_sigma']
        kl = 0.5 * (mu * mu + sg * sg - 2.0 * ls - 1.0).sum(axis=1)
        return {'L1': l1, 'reconstructionLoss': rec.mean(), 'kl': kl.mean(), 'loss': (rec + kl).mean()}

    def backward(self, p, x, out, cache):
        n = x.shape[0]
        dt = x.dtype.type
        g = {}
        gx = np.sign(out['x_hat'] - x) * dt(1.0 / n)
        da, g['dec_Conv2D_final/kernel'], g['dec_Conv2D_final/bias'] = nn.conv2d_bwd(cache['ga'][4], p['dec_Conv2D_final/kernel'], gx, 1)
        for i in reversed(range(4)):
            dc = nn.leaky_relu_bwd(cache['gc'][i], da, ALPHA)
            da, g[f'dec_Conv2DT_{i + 1}/kernel'], g[f'dec_Conv2DT_{i + 1}/bias'] = \
                nn.conv2d_transpose_bwd(cache['ga'][i], p[f'dec_Conv2DT_{i + 1}/kernel'], dc, 2)
        dz, g['dense_2/kernel'], g['dense_2/bias'] = nn.dense_bwd(cache['z'], p['dense_2/kernel'], da.reshape(n, -1))
        mu, sg, eps = cache['mu'], cache['sigma'], cache['eps']
        klw = dt(1.0 / n)
        dmu = dz + mu * klw
        dls = dz * eps * sg + (sg * sg - dt(1.0)) * klw
        df1, g['dense/kernel'], g['dense/bias'] = nn.dense_bwd(cache['flat'], p['dense/kernel'], dmu)
        df2, g['dense_1/kernel'], g['dense_1/bias'] = nn.dense_bwd(cache['flat'], p['dense_1/kernel'], dls)
        da = (df1 + df2).reshape(cache['a'][4].shape)
        for i in reversed(range(4)):
            dc = nn.leaky_relu_bwd(cache['c'][i], da, ALPHA)
            da, g[f'enc_conv2D_{i + 1}/kernel'], g[f'enc_conv2D_{i + 1}/bias'] = nn.conv2d_bwd(cache['a'][i], p[f'enc_conv2D_{i + 1}/kernel'], dc, 2)
        g['__dx'] = da
        return g

    def new_opt(self, p):
        return {'t': 0, 'm': {k: np.zeros_like(v) for k, v in p.items()}, 'v': {k: np.zeros_like(v) for k, v in p.items()}}

    def train_step(self, p, opt, x, eps=None, lr=1e-4, beta1=0.5):
        out, cache = self.forward(p, x, eps)
        ls = self.losses(x, out)
        g = self.backward(p, x, out, cache)
        opt['t'] += 1
        for name, _, _ in self.spec:
            nn.adam_tf_step(p[name], g[name], opt['m'][name], opt['v'][name], opt['t'], lr, beta1)
        return out, ls, g
