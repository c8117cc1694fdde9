"""Oracle for the Zimmerer-style VAE and context-encoding VAE: models/variational_autoencoder_Zimmerer.py:7-32 (k4 s2 convolutions 16-64-256-1024 with
tf.nn.leaky_relu (alpha 0.2), no normalisation, no dropout, Dense mu / log-sigma heads on the flattened map, Dense back to
[r, r, 1024], four k4 s2 transposed convolutions 1024-256-64-16, a k4 s1 convolution to one channel) under trainers/VAE.py:36-42
(loss = mean_n(sum |x - x_hat| + KL_n), z_sigma = exp(z_log_sigma)).  numpy forward, hand-written backward.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED (no TensorFlow here, no golden vectors in the reference);
tests/test_oracle_zimmerer.py anchors every gradient on torch autograd in float64.

The VAE model opens no variable scope: convolutions carry their explicit names, the Dense layers are dense (mu), dense_1 (log sigma),
dense_2 (decoder) in first-call order.  models/context_encoder_variational_autoencoder_Zimmerer.py:8-45 is the same stack under the scopes
Encoder / Bottleneck / Decoder with a second, shared-weight branch on x_ce that decodes dec_dense(mu_layer(flatten_ce)) (no sampling);
trainers/ceVAE.py:38-51 scores it: L1_vae against x, L1_ce against x_ce (sic), loss = mean(rec_vae + kl + rec_ce),
anomaly = L1_vae * |d mean(rec_vae + kl) / d x|."""
import numpy as np

from . import nn

ALPHA = 0.2                         # tf.nn.leaky_relu default
ENC_F = (16, 64, 256, 1024)
DEC_F = (1024, 256, 64, 16)


def param_spec(height=128, zdim=128, channels=1, scoped=False):
    """scoped=True: the ceVAE model's names (Encoder/..., Bottleneck/dense*, Decoder/...)."""
    e, b, d = ('Encoder/', 'Bottleneck/', 'Decoder/') if scoped else ('', '', '')
    r = height // 16
    spec, cin = [], channels
    for i, f in enumerate(ENC_F):
        spec += [(f'{e}enc_conv2D_{i + 1}/kernel', (4, 4, cin, f), 'conv_w'), (f'{e}enc_conv2D_{i + 1}/bias', (f,), 'bias')]
        cin = f
    flat = r * r * 1024
    spec += [(b + 'dense/kernel', (flat, zdim), 'dense_w'), (b + 'dense/bias', (zdim,), 'bias'),
             (b + 'dense_1/kernel', (flat, zdim), 'dense_w'), (b + 'dense_1/bias', (zdim,), 'bias'),
             (b + 'dense_2/kernel', (zdim, flat), 'dense_w'), (b + 'dense_2/bias', (flat,), 'bias')]
    cin = 1024
    for i, f in enumerate(DEC_F):
        spec += [(f'{d}dec_Conv2DT_{i + 1}/kernel', (4, 4, f, cin), 'conv_w'), (f'{d}dec_Conv2DT_{i + 1}/bias', (f,), 'bias')]
        cin = f
    spec += [(d + 'dec_Conv2D_final/kernel', (4, 4, cin, channels), 'conv_w'), (d + 'dec_Conv2D_final/bias', (channels,), 'bias')]
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
    def __init__(self, height=128, zdim=128, scoped=False):
        assert height % 16 == 0
        self.height, self.zdim, self.r = height, zdim, height // 16
        self.spec = param_spec(height, zdim, scoped=scoped)
        self.pre = ('Encoder/', 'Bottleneck/', 'Decoder/') if scoped else ('', '', '')

    def _view(self, p):
        """parameters under the un-scoped names the methods below use."""
        if not self.pre[0]:
            return p
        return {k.split('/', 1)[1]: v for k, v in p.items()}

    def _scoped(self, g):
        if not self.pre[0]:
            return g
        e, b, d = self.pre
        return {(k if k.startswith('__') else (e if k.startswith('enc_') else d if k.startswith('dec_') else b) + k): v for k, v in g.items()}

    def forward(self, p, x, eps=None):
        p = self._view(p)
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

    def backward(self, p, x, out, cache, kl=True, sampled=True, inv=None):
        """kl / sampled False: the context branch (z = mu, no KL term: d mu = d z, d log_sigma = 0); inv: weight of a sample (default 1/n)."""
        p = self._view(p)
        n = x.shape[0]
        dt = x.dtype.type
        g = {}
        inv = dt(1.0 / n) if inv is None else dt(inv)
        gx = np.sign(out['x_hat'] - x) * inv
        da, g['dec_Conv2D_final/kernel'], g['dec_Conv2D_final/bias'] = nn.conv2d_bwd(cache['ga'][4], p['dec_Conv2D_final/kernel'], gx, 1)
        for i in reversed(range(4)):
            dc = nn.leaky_relu_bwd(cache['gc'][i], da, ALPHA)
            da, g[f'dec_Conv2DT_{i + 1}/kernel'], g[f'dec_Conv2DT_{i + 1}/bias'] = \
                nn.conv2d_transpose_bwd(cache['ga'][i], p[f'dec_Conv2DT_{i + 1}/kernel'], dc, 2)
        dz, g['dense_2/kernel'], g['dense_2/bias'] = nn.dense_bwd(cache['z'], p['dense_2/kernel'], da.reshape(n, -1))
        mu, sg, eps = cache['mu'], cache['sigma'], cache['eps']
        klw = inv if kl else dt(0.0)
        dmu = dz + mu * klw
        dls = (dz * eps * sg if sampled else 0.0 * dz) + (sg * sg - dt(1.0)) * klw
        df1, g['dense/kernel'], g['dense/bias'] = nn.dense_bwd(cache['flat'], p['dense/kernel'], dmu)
        df2, g['dense_1/kernel'], g['dense_1/bias'] = nn.dense_bwd(cache['flat'], p['dense_1/kernel'], dls)
        da = (df1 + df2).reshape(cache['a'][4].shape)
        for i in reversed(range(4)):
            dc = nn.leaky_relu_bwd(cache['c'][i], da, ALPHA)
            da, g[f'enc_conv2D_{i + 1}/kernel'], g[f'enc_conv2D_{i + 1}/bias'] = nn.conv2d_bwd(cache['a'][i], p[f'enc_conv2D_{i + 1}/kernel'], dc, 2)
        g['__dx'] = da
        return self._scoped(g)

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


class CeVAEZimmerer(VAEZimmerer):
    """models/context_encoder_variational_autoencoder_Zimmerer.py + trainers/ceVAE.py:38-51."""

    def __init__(self, height=128, zdim=128):
        super().__init__(height, zdim, scoped=True)

    def ce_forward(self, p, x, x_ce, eps=None):
        o1, c1 = self.forward(p, x, eps)
        o2, c2 = self.forward(p, x_ce, None)              # eps = 0: z = mu (the context branch decodes dec_dense(mu_layer(.)))
        out = dict(o1)
        out['x_hat_ce'] = o2['x_hat']
        return out, (c1, c2, o2)

    def ce_losses(self, x, x_ce, out):
        n = x.shape[0]
        l1v, l1c = np.abs(x - out['x_hat']), np.abs(x_ce - out['x_hat_ce'])
        rv, rc = l1v.reshape(n, -1).sum(1), l1c.reshape(n, -1).sum(1)
        mu, ls, sg = out['z_mu'], out['z_log_sigma'], out['z_sigma']
        kl = 0.5 * (mu * mu + sg * sg - 2.0 * ls - 1.0).sum(axis=1)
        return {'L1_vae': l1v, 'L1_ce': l1c, 'L1': 0.5 * (l1v + l1c), 'Rec_vae': rv.mean(), 'Rec_ce': rc.mean(),
                'reconstructionLoss': 0.5 * (rv + rc).mean(), 'kl': kl.mean(), 'loss': (rv + kl + rc).mean(), 'loss_vae': (rv + kl).mean()}

    def ce_backward(self, p, x, x_ce, out, caches):
        c1, c2, o2 = caches
        g1 = self.backward(p, x, out, c1)
        g2 = self.backward(p, x_ce, o2, c2, kl=False, sampled=False)
        g = {k: g1[k] + g2[k] for k in g1 if not k.startswith('__')}
        # anomaly (:51): x enters loss_vae through the encoder and, directly, as the L1 target
        n = x.shape[0]
        dx = g1['__dx'] - np.sign(out['x_hat'] - x) / n
        g['__anomaly'] = np.abs(x - out['x_hat']) * np.abs(dx)
        return g

    def ce_train_step(self, p, opt, x, x_ce, eps=None, lr=1e-4, beta1=0.5):
        out, caches = self.ce_forward(p, x, x_ce, eps)
        ls = self.ce_losses(x, x_ce, out)
        g = self.ce_backward(p, x, x_ce, out, caches)
        opt['t'] += 1
        for name, _, _ in self.spec:
            nn.adam_tf_step(p[name], g[name], opt['m'][name], opt['v'][name], opt['t'], lr, beta1)
        return out, ls, g
