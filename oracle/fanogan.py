"""Oracle for the unified f-AnoGAN graph (models/fanogan.py:11-84) and its three
optimisation phases (trainers/fAnoGAN.py:45-77): numpy forward passes, hand-written
first- and second-order backward passes (the WGAN-GP penalty differentiates the
critic's input gradient w.r.t. the critic's weights).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: there is no
TensorFlow in this image and the reference holds no golden vectors for this path;
tests/test_oracle_fanogan.py anchors every gradient on torch autograd (double
backward included) in float64.

Graph restated (all NHWC, fp32 in the reference):
  Encoder       enc blocks (conv k5 s2 + frozen-stats BN + LeakyReLU, customlayers.py:16-25)
                -> 1x1 conv C/8 -> Dense(zDim) -> dropout -> tanh          (fanogan.py:15-30)
  Generator     Dense(flat) -> dropout -> reshape -> 1x1 conv C -> LN+ReLU ->
                ConvT k5 s2 + LN + LeakyReLU blocks -> 1x1 conv -> sigmoid (fanogan.py:33-47,
                customlayers.py:28-38 with use_batchnorm=False)
  Discriminator conv k5 s2 + LN + LeakyReLU blocks -> Dense(1) applied per feature-map
                location (no Flatten, fanogan.py:50-58)
  LN            keras LayerNormalization([1, 2]): statistics over (H, W) per sample and
                channel, eps 1e-3, gamma/beta of shape [H, W]
  losses        trainers/fAnoGAN.py:50-66; Adam(beta1 .5, beta2 .9) per variable group :71-77
"""
import numpy as np

from . import nn

LN_EPS = 1e-3
LRELU = 0.3


# --------------------------------------------------------------------------
# LayerNormalization([1, 2]) and its first / second order backward
# --------------------------------------------------------------------------
def _bc(g):
    return g[None, :, :, None]


def ln_fwd(c, gamma, beta, eps=LN_EPS):
    mu = c.mean(axis=(1, 2), keepdims=True)
    var = ((c - mu) ** 2).mean(axis=(1, 2), keepdims=True)
    r = 1.0 / np.sqrt(var + c.dtype.type(eps))
    xh = (c - mu) * r
    return xh * _bc(gamma) + _bc(beta), (xh, r)


def _E(a):
    return a.mean(axis=(1, 2), keepdims=True)


def ln_bwd(g, gamma, cache):
    """(dc, dgamma, dbeta) for y = ln_fwd(c), g = dL/dy."""
    xh, r = cache
    p = g * _bc(gamma)
    dc = r * (p - _E(p) - xh * _E(p * xh))
    return dc, (g * xh).sum(axis=(0, 3)), g.sum(axis=(0, 3))


def ln_bwd2(q, v, gamma, cache):
    """Adjoint of the map (v, gamma, c) -> dc = ln_bwd(v, gamma, cache)[0]: given q = dP/d(dc)
    returns (dP/dv, dP/dgamma, dP/dc)."""
    xh, r = cache
    p = v * _bc(gamma)
    epx, eqx = _E(p * xh), _E(q * xh)
    dc = r * (p - _E(p) - xh * epx)
    pbar = r * (q - _E(q) - xh * eqx)
    xhbar = -r * (q * epx + p * eqx)
    cbar = r * (xhbar - _E(xhbar) - xh * _E(xhbar * xh)) - r * _E(q * dc) * xh
    return pbar * _bc(gamma), (pbar * v).sum(axis=(0, 3)), cbar


def sigmoid(a):
    return 1.0 / (1.0 + np.exp(-a))


# --------------------------------------------------------------------------
# parameter table in TF variable-creation order (Encoder, Generator, Discriminator)
# --------------------------------------------------------------------------
def _ln_name(scope, idx):
    return scope + ('layer_normalization' if idx == 0 else 'layer_normalization_%d' % idx)


def param_spec(height=128, inter_res=8, zdim=128, channels=1):
    npool = int(round(np.log2(height) - np.log2(inter_res)))
    spec = []
    # Encoder (fanogan.py:15-30): tf.layers names are unique per variable scope, keras names globally
    cin, res = channels, height
    for i in range(npool):
        f = min(128, 32 * 2 ** i)
        spec += [('Encoder/enc_conv2D_%d/kernel' % i, (5, 5, cin, f), 'conv_w'), ('Encoder/enc_conv2D_%d/bias' % i, (f,), 'bias')]
        bn = 'Encoder/batch_normalization' + ('' if i == 0 else '_%d' % i)
        spec += [(bn + '/gamma', (f,), 'gamma'), (bn + '/beta', (f,), 'beta')]
        cin, res = f, res // 2
    cenc, cmid = cin, cin // 8
    flat = inter_res * inter_res * cmid
    spec += [('Encoder/conv2d/kernel', (1, 1, cenc, cmid), 'conv_w'), ('Encoder/conv2d/bias', (cmid,), 'bias'),
             ('Encoder/dense/kernel', (flat, zdim), 'dense_w'), ('Encoder/dense/bias', (zdim,), 'bias')]
    # Generator (fanogan.py:33-47): dec_dense is called first, then the 1x1 conv, then the decoder layers
    spec += [('Generator/dense/kernel', (zdim, flat), 'dense_w'), ('Generator/dense/bias', (flat,), 'bias'),
             ('Generator/conv2d_1/kernel', (1, 1, cmid, cenc), 'conv_w'), ('Generator/conv2d_1/bias', (cenc,), 'bias')]
    ln = 0
    spec += [(_ln_name('Generator/', ln) + '/gamma', (inter_res, inter_res), 'gamma'),
             (_ln_name('Generator/', ln) + '/beta', (inter_res, inter_res), 'beta')]
    ln += 1
    cin, res = cenc, inter_res
    for i in range(npool):
        f = max(32, 128 // 2 ** i)
        spec += [('Generator/dec_Conv2DT_%d/kernel' % i, (5, 5, f, cin), 'conv_w'), ('Generator/dec_Conv2DT_%d/bias' % i, (f,), 'bias')]
        res *= 2
        spec += [(_ln_name('Generator/', ln) + '/gamma', (res, res), 'gamma'), (_ln_name('Generator/', ln) + '/beta', (res, res), 'beta')]
        ln += 1
        cin = f
    spec += [('Generator/dec_Conv2D_final/kernel', (1, 1, cin, channels), 'conv_w'), ('Generator/dec_Conv2D_final/bias', (channels,), 'bias')]
    # Discriminator (fanogan.py:50-58)
    cin, res = channels, height
    for i in range(npool):
        f = min(128, 32 * 2 ** i)
        spec += [('Discriminator/enc_conv2D_%d/kernel' % i, (5, 5, cin, f), 'conv_w'), ('Discriminator/enc_conv2D_%d/bias' % i, (f,), 'bias')]
        res //= 2
        spec += [(_ln_name('Discriminator/', ln) + '/gamma', (res, res), 'gamma'), (_ln_name('Discriminator/', ln) + '/beta', (res, res), 'beta')]
        ln += 1
        cin = f
    spec += [('Discriminator/dense/kernel', (cin, 1), 'dense_w'), ('Discriminator/dense/bias', (1,), 'bias')]
    return spec


def group_of(name):
    return name.split('/')[0]


class FAnoGAN:
    def __init__(self, height=128, inter_res=8, zdim=128, channels=1, scale=10.0, kappa=1.0):
        self.height, self.inter_res, self.zdim, self.channels = height, inter_res, zdim, channels
        self.scale, self.kappa = scale, kappa
        self.npool = int(round(np.log2(height) - np.log2(inter_res)))
        self.spec = param_spec(height, inter_res, zdim, channels)
        self.ln_g = [n[:-len('/gamma')] for n, _, _ in self.spec if n.startswith('Generator/layer_norm') and n.endswith('gamma')]
        self.ln_d = [n[:-len('/gamma')] for n, _, _ in self.spec if n.startswith('Discriminator/layer_norm') and n.endswith('gamma')]
        self.bn_e = [n[:-len('/gamma')] for n, _, _ in self.spec if n.startswith('Encoder/batch_norm') and n.endswith('gamma')]
        self.final_act = 'sigmoid'        # fanogan.py:42,47; AnoVAEGAN's generator output is linear

    # ------------------------------------------------------------------ Encoder
    def enc_forward(self, p, x, mask_z=None):
        cache = {'a': [x], 'c': []}
        a = x
        for i in range(self.npool):
            c = nn.conv2d_fwd(a, p['Encoder/enc_conv2D_%d/kernel' % i], p['Encoder/enc_conv2D_%d/bias' % i], 2)
            bnv = nn.bn_frozen_fwd(c, p[self.bn_e[i] + '/gamma'], p[self.bn_e[i] + '/beta'])
            a = nn.leaky_relu_fwd(bnv, LRELU)
            cache['c'].append(c); cache['a'].append(a)
        t = nn.conv2d_fwd(a, p['Encoder/conv2d/kernel'], p['Encoder/conv2d/bias'], 1)
        flat = t.reshape(t.shape[0], -1)
        zr = nn.dense_fwd(flat, p['Encoder/dense/kernel'], p['Encoder/dense/bias'])
        if mask_z is not None:
            zr = zr * mask_z
        z = np.tanh(zr)
        cache.update(t=t, flat=flat, z=z, mask_z=mask_z)
        return z, cache

    def enc_backward(self, p, cache, dz):
        g = {}
        dzr = dz * (1.0 - cache['z'] ** 2)
        if cache['mask_z'] is not None:
            dzr = dzr * cache['mask_z']
        dflat, g['Encoder/dense/kernel'], g['Encoder/dense/bias'] = nn.dense_bwd(cache['flat'], p['Encoder/dense/kernel'], dzr)
        da, g['Encoder/conv2d/kernel'], g['Encoder/conv2d/bias'] = nn.conv2d_bwd(cache['a'][-1], p['Encoder/conv2d/kernel'],
                                                                               dflat.reshape(cache['t'].shape), 1)
        for i in reversed(range(self.npool)):
            c = cache['c'][i]
            bnv = nn.bn_frozen_fwd(c, p[self.bn_e[i] + '/gamma'], p[self.bn_e[i] + '/beta'])
            dbn = nn.leaky_relu_bwd(bnv, da, LRELU)
            dc, g[self.bn_e[i] + '/gamma'], g[self.bn_e[i] + '/beta'] = nn.bn_frozen_bwd(c, p[self.bn_e[i] + '/gamma'], dbn)
            da, g['Encoder/enc_conv2D_%d/kernel' % i], g['Encoder/enc_conv2D_%d/bias' % i] = \
                nn.conv2d_bwd(cache['a'][i], p['Encoder/enc_conv2D_%d/kernel' % i], dc, 2)
        return g

    # ------------------------------------------------------------------ Generator
    def gen_forward(self, p, z, mask_g=None):
        r = self.inter_res
        dv = nn.dense_fwd(z, p['Generator/dense/kernel'], p['Generator/dense/bias'])
        if mask_g is not None:
            dv = dv * mask_g
        dmap = dv.reshape(z.shape[0], r, r, -1)
        c = nn.conv2d_fwd(dmap, p['Generator/conv2d_1/kernel'], p['Generator/conv2d_1/bias'], 1)
        cache = {'z': z, 'dmap': dmap, 'mask_g': mask_g, 'c': [c], 'ln': [], 'a': []}
        y, lc = ln_fwd(c, p[self.ln_g[0] + '/gamma'], p[self.ln_g[0] + '/beta'])
        a = np.maximum(y, 0)
        cache['ln'].append((y, lc)); cache['a'].append(a)
        for i in range(self.npool):
            c = nn.conv2d_transpose_fwd(a, p['Generator/dec_Conv2DT_%d/kernel' % i], p['Generator/dec_Conv2DT_%d/bias' % i], 2)
            y, lc = ln_fwd(c, p[self.ln_g[i + 1] + '/gamma'], p[self.ln_g[i + 1] + '/beta'])
            a = nn.leaky_relu_fwd(y, LRELU)
            cache['c'].append(c); cache['ln'].append((y, lc)); cache['a'].append(a)
        o = nn.conv2d_fwd(a, p['Generator/dec_Conv2D_final/kernel'], p['Generator/dec_Conv2D_final/bias'], 1)
        xg = sigmoid(o) if self.final_act == 'sigmoid' else o
        cache['x'] = xg
        return xg, cache

    def gen_backward(self, p, cache, dx):
        """dx = dL/d(sigmoid output).  Returns (parameter grads, dL/dz)."""
        g = {}
        xg = cache['x']
        do = dx * xg * (1.0 - xg) if self.final_act == 'sigmoid' else dx
        da, g['Generator/dec_Conv2D_final/kernel'], g['Generator/dec_Conv2D_final/bias'] = \
            nn.conv2d_bwd(cache['a'][-1], p['Generator/dec_Conv2D_final/kernel'], do, 1)
        for i in reversed(range(self.npool)):
            y, lc = cache['ln'][i + 1]
            dy = nn.leaky_relu_bwd(y, da, LRELU)
            dc, g[self.ln_g[i + 1] + '/gamma'], g[self.ln_g[i + 1] + '/beta'] = ln_bwd(dy, p[self.ln_g[i + 1] + '/gamma'], lc)
            da, g['Generator/dec_Conv2DT_%d/kernel' % i], g['Generator/dec_Conv2DT_%d/bias' % i] = \
                nn.conv2d_transpose_bwd(cache['a'][i], p['Generator/dec_Conv2DT_%d/kernel' % i], dc, 2)
        y, lc = cache['ln'][0]
        dy = nn.leaky_relu_bwd(y, da, 0.0)
        dc, g[self.ln_g[0] + '/gamma'], g[self.ln_g[0] + '/beta'] = ln_bwd(dy, p[self.ln_g[0] + '/gamma'], lc)
        dmap, g['Generator/conv2d_1/kernel'], g['Generator/conv2d_1/bias'] = nn.conv2d_bwd(cache['dmap'], p['Generator/conv2d_1/kernel'], dc, 1)
        dv = dmap.reshape(dmap.shape[0], -1)
        if cache['mask_g'] is not None:
            dv = dv * cache['mask_g']
        dz, g['Generator/dense/kernel'], g['Generator/dense/bias'] = nn.dense_bwd(cache['z'], p['Generator/dense/kernel'], dv)
        return g, dz

    # ------------------------------------------------------------------ Discriminator
    def disc_forward(self, p, x):
        cache = {'a': [x], 'ln': []}
        a = x
        for i in range(self.npool):
            c = nn.conv2d_fwd(a, p['Discriminator/enc_conv2D_%d/kernel' % i], p['Discriminator/enc_conv2D_%d/bias' % i], 2)
            y, lc = ln_fwd(c, p[self.ln_d[i] + '/gamma'], p[self.ln_d[i] + '/beta'])
            a = nn.leaky_relu_fwd(y, LRELU)
            cache['ln'].append((y, lc)); cache['a'].append(a)
        d = a @ p['Discriminator/dense/kernel'] + p['Discriminator/dense/bias']     # [n, r, r, 1]
        return a, d, cache

    def disc_backward(self, p, cache, df=None, dd=None, inject=None, want_params=True):
        """Backward of the critic for dL/dfeatures = df and dL/dd = dd; `inject[i]` is an extra
        dL/dc_i (the second-order term of the gradient penalty).  Returns (grads, dL/dx)."""
        g = {}
        feat = cache['a'][-1]
        da = np.zeros_like(feat) if df is None else df.copy()
        if dd is not None:
            da = da + dd * p['Discriminator/dense/kernel'][:, 0]
            g['Discriminator/dense/kernel'] = (feat * dd).reshape(-1, feat.shape[-1]).sum(axis=0)[:, None]
            g['Discriminator/dense/bias'] = dd.sum().reshape(1)
        for i in reversed(range(self.npool)):
            y, lc = cache['ln'][i]
            dy = nn.leaky_relu_bwd(y, da, LRELU)
            dc, g[self.ln_d[i] + '/gamma'], g[self.ln_d[i] + '/beta'] = ln_bwd(dy, p[self.ln_d[i] + '/gamma'], lc)
            if inject is not None:
                dc = dc + inject[i]
            da, g['Discriminator/enc_conv2D_%d/kernel' % i], g['Discriminator/enc_conv2D_%d/bias' % i] = \
                nn.conv2d_bwd(cache['a'][i], p['Discriminator/enc_conv2D_%d/kernel' % i], dc, 2)
        return (g if want_params else None), da

    def disc_input_grad(self, p, cache):
        """ddx = tf.gradients(d_hat, x_hat)[0] (trainers/fAnoGAN.py:55): gradient of sum(d) w.r.t. the input,
        keeping what the second-order pass needs."""
        feat = cache['a'][-1]
        u = np.broadcast_to(p['Discriminator/dense/kernel'][:, 0], feat.shape).astype(feat.dtype)
        tape = []
        for i in reversed(range(self.npool)):
            y, lc = cache['ln'][i]
            v = nn.leaky_relu_bwd(y, u, LRELU)
            dc, _, _ = ln_bwd(v, p[self.ln_d[i] + '/gamma'], lc)
            tape.append((i, v, dc))
            u, _, _ = nn.conv2d_bwd(cache['a'][i], p['Discriminator/enc_conv2D_%d/kernel' % i], dc, 2)
        return u, tape[::-1]

    def gradient_penalty(self, ddx):
        """trainers/fAnoGAN.py:56-57: slopes over axis=1 (sic: the H axis only).  Returns (penalty, d penalty / d ddx)."""
        s = np.sqrt((ddx ** 2).sum(axis=1))                   # [n, W, C]
        pen = self.scale * ((s - 1.0) ** 2).mean()
        ds = self.scale * 2.0 * (s - 1.0) / s.size
        return pen, (ds / s)[:, None, :, :] * ddx

    def disc_penalty_grads(self, p, cache, tape, gbar):
        """Second-order pass: gradient of the penalty w.r.t. the critic's parameters through the
        input-gradient graph.  Returns (direct grads, inject) where inject[i] = d penalty / d c_i to be
        pushed down the ordinary backward pass."""
        g, inject = {}, [None] * self.npool
        ubar = gbar
        for i in range(self.npool):
            _, v, dc = tape[i]
            y, lc = cache['ln'][i]
            w = p['Discriminator/enc_conv2D_%d/kernel' % i]
            # u_i = dgrad(dc_i, W_i): adjoint w.r.t. dc is the forward conv, w.r.t. W the filter-gradient contraction
            q = nn.conv2d_fwd(ubar, w, None, 2)
            _, g['Discriminator/enc_conv2D_%d/kernel' % i], _ = nn.conv2d_bwd(ubar, w, dc, 2)
            vbar, g[self.ln_d[i] + '/gamma'], inject[i] = ln_bwd2(q, v, p[self.ln_d[i] + '/gamma'], lc)
            ubar = nn.leaky_relu_bwd(y, vbar, LRELU)
        g['Discriminator/dense/kernel'] = ubar.reshape(-1, ubar.shape[-1]).sum(axis=0)[:, None]
        return g, inject

    # ------------------------------------------------------------------ phases (trainers/fAnoGAN.py:50-77)
    def gen_phase(self, p, z, mask_g=None, caches=None):
        """`caches` (optional dict) receives the forward caches -- the GPU parity tests compare activation signs with them."""
        xg, gc = self.gen_forward(p, z, mask_g)
        _, d, dcache = self.disc_forward(p, xg)
        if caches is not None:
            caches.update(gen=gc, disc=[dcache])
        gen_loss = -d.mean()
        dd = np.full_like(d, -1.0 / d.size)
        _, dx = self.disc_backward(p, dcache, dd=dd, want_params=False)
        grads, _ = self.gen_backward(p, gc, dx)
        return {'gen_loss': gen_loss, 'generated': xg}, grads

    def disc_phase(self, p, x, z, alpha, mask_g=None, caches=None):
        xg, gcache = self.gen_forward(p, z, mask_g)
        _, d_fake, c_fake = self.disc_forward(p, xg)
        _, d_real, c_real = self.disc_forward(p, x)
        x_hat = x + alpha.reshape(-1, 1, 1, 1).astype(x.dtype) * (xg - x)
        _, _, c_hat = self.disc_forward(p, x_hat)
        if caches is not None:
            caches.update(gen=gcache, disc=[c_fake, c_real, c_hat])
        ddx, tape = self.disc_input_grad(p, c_hat)
        pen, gbar = self.gradient_penalty(ddx)
        losses = {'disc_fake': d_fake.mean(), 'disc_real': d_real.mean(), 'generated': xg}
        losses['disc_loss'] = losses['disc_fake'] - losses['disc_real'] + pen
        losses['penalty'] = pen
        g_f, _ = self.disc_backward(p, c_fake, dd=np.full_like(d_fake, 1.0 / d_fake.size))
        g_r, _ = self.disc_backward(p, c_real, dd=np.full_like(d_real, -1.0 / d_real.size))
        g_2, inject = self.disc_penalty_grads(p, c_hat, tape, gbar)
        g_3, _ = self.disc_backward(p, c_hat, inject=inject)
        grads = {}
        for part in (g_f, g_r, g_2, g_3):
            for k, v in part.items():
                grads[k] = grads.get(k, 0) + v
        return losses, grads

    def enc_phase(self, p, x, mask_z=None, mask_g=None, caches=None):
        z_enc, ec = self.enc_forward(p, x, mask_z)
        x_enc, gc = self.gen_forward(p, z_enc, mask_g)
        f_enc, _, c_enc = self.disc_forward(p, x_enc)
        f_real, _, c_real = self.disc_forward(p, x)
        if caches is not None:
            caches.update(enc=ec, gen=gc, disc=[c_enc, c_real])
        loss_img = ((x - x_enc) ** 2).mean()
        loss_fts = ((f_enc - f_real) ** 2).mean()
        l1 = np.abs(x - x_enc)
        rec = l1.reshape(l1.shape[0], -1).sum(axis=1).mean()
        losses = {'loss_img': loss_img, 'loss_fts': loss_fts, 'enc_loss': loss_img + self.kappa * loss_fts, 'L1': l1,
                  'reconstructionLoss': rec, 'loss': rec, 'z_enc': z_enc, 'reconstruction': x_enc}
        df = self.kappa * 2.0 * (f_enc - f_real) / f_enc.size
        _, dx = self.disc_backward(p, c_enc, df=df, want_params=False)
        dx = dx + 2.0 * (x_enc - x) / x.size
        _, dz = self.gen_backward(p, gc, dx)
        grads = self.enc_backward(p, ec, dz)
        return losses, grads

    def reconstruct(self, p, x):
        """trainers/fAnoGAN.py:220-239 (dropout off)."""
        z_enc, _ = self.enc_forward(p, x)
        return self.gen_forward(p, z_enc)[0]

    # ------------------------------------------------------------------ optimiser (three Adams, beta1 .5, beta2 .9)
    def new_opt(self, p):
        return {'m': {k: np.zeros_like(v) for k, v in p.items()}, 'v': {k: np.zeros_like(v) for k, v in p.items()},
                't': {'Encoder': 0, 'Generator': 0, 'Discriminator': 0}}

    def apply(self, p, opt, grads, group, lr):
        opt['t'][group] += 1
        for k, gk in grads.items():
            if group_of(k) == group:
                nn.adam_tf_step(p[k], np.asarray(gk, p[k].dtype).reshape(p[k].shape), opt['m'][k], opt['v'][k], opt['t'][group],
                                lr, beta1=0.5, beta2=0.9)


# ---------------------------------------------------------------------------------------------------------------
# AnoVAE-GAN (models/anovaegan.py:10-80, trainers/AnoVAEGAN.py:45-86): the same encoder / generator / critic blocks as the
# unified f-AnoGAN, but the encoder ends in mu / log-sigma heads (z = mu + eps * exp(log_sigma)), the generator decodes z_vae and
# has a linear output, and the critic judges the RECONSTRUCTION G(E(x)) against x.  Three Adams (beta1 .5, beta2 .9):
#   optim_vae: enc_loss = mean_n sum|x - out| + kl_weight * mean_n KL      over Encoder + Generator variables
#   optim_gen: gen_loss = -mean D(out)                                        over Generator variables
#   optim_dis: disc_loss = mean D(out) - mean D(x) + scale * penalty          over Discriminator variables
# ---------------------------------------------------------------------------------------------------------------
def param_spec_anovaegan(height=128, inter_res=8, zdim=128, channels=1):
    spec = param_spec(height, inter_res, zdim, channels)
    i = [k for k, (n, _, _) in enumerate(spec) if n == 'Encoder/dense/bias'][0]
    flat = spec[i - 1][1][0]
    return spec[:i + 1] + [('Encoder/dense_1/kernel', (flat, zdim), 'dense_w'), ('Encoder/dense_1/bias', (zdim,), 'bias')] + spec[i + 1:]


class AnoVAEGAN(FAnoGAN):
    def __init__(self, height=128, inter_res=8, zdim=128, channels=1, scale=10.0, kl_weight=1.0):
        super().__init__(height, inter_res, zdim, channels, scale, 1.0)
        self.kl_weight = kl_weight
        self.spec = param_spec_anovaegan(height, inter_res, zdim, channels)
        self.final_act = 'none'

    def vae_enc_forward(self, p, x, eps, mask_mu=None, mask_sg=None):
        cache = {'a': [x], 'c': []}
        a = x
        for i in range(self.npool):
            c = nn.conv2d_fwd(a, p['Encoder/enc_conv2D_%d/kernel' % i], p['Encoder/enc_conv2D_%d/bias' % i], 2)
            a = nn.leaky_relu_fwd(nn.bn_frozen_fwd(c, p[self.bn_e[i] + '/gamma'], p[self.bn_e[i] + '/beta']), LRELU)
            cache['c'].append(c); cache['a'].append(a)
        t = nn.conv2d_fwd(a, p['Encoder/conv2d/kernel'], p['Encoder/conv2d/bias'], 1)
        flat = t.reshape(t.shape[0], -1)
        mu = nn.dense_fwd(flat, p['Encoder/dense/kernel'], p['Encoder/dense/bias'])
        ls = nn.dense_fwd(flat, p['Encoder/dense_1/kernel'], p['Encoder/dense_1/bias'])
        if mask_mu is not None:
            mu = mu * mask_mu
        if mask_sg is not None:
            ls = ls * mask_sg
        sg = np.exp(ls)
        z = mu + eps * sg
        kl = 0.5 * (mu * mu + sg * sg - 2.0 * ls - 1.0).sum(axis=1)           # tf.log(tf.square(sigma)) = 2 log_sigma
        cache.update(t=t, flat=flat, mu=mu, ls=ls, sg=sg, eps=eps, mask_mu=mask_mu, mask_sg=mask_sg)
        return z, kl, cache

    def vae_enc_backward(self, p, cache, dz, klw):
        g = {}
        mu, sg, eps = cache['mu'], cache['sg'], cache['eps']
        dmu = dz + mu * klw
        dls = dz * eps * sg + (sg * sg - 1.0) * klw
        if cache['mask_mu'] is not None:
            dmu = dmu * cache['mask_mu']
        if cache['mask_sg'] is not None:
            dls = dls * cache['mask_sg']
        d1, g['Encoder/dense/kernel'], g['Encoder/dense/bias'] = nn.dense_bwd(cache['flat'], p['Encoder/dense/kernel'], dmu)
        d2, g['Encoder/dense_1/kernel'], g['Encoder/dense_1/bias'] = nn.dense_bwd(cache['flat'], p['Encoder/dense_1/kernel'], dls)
        da, g['Encoder/conv2d/kernel'], g['Encoder/conv2d/bias'] = nn.conv2d_bwd(cache['a'][-1], p['Encoder/conv2d/kernel'],
                                                                               (d1 + d2).reshape(cache['t'].shape), 1)
        for i in reversed(range(self.npool)):
            c = cache['c'][i]
            bnv = nn.bn_frozen_fwd(c, p[self.bn_e[i] + '/gamma'], p[self.bn_e[i] + '/beta'])
            dc, g[self.bn_e[i] + '/gamma'], g[self.bn_e[i] + '/beta'] = nn.bn_frozen_bwd(c, p[self.bn_e[i] + '/gamma'], nn.leaky_relu_bwd(bnv, da, LRELU))
            da, g['Encoder/enc_conv2D_%d/kernel' % i], g['Encoder/enc_conv2D_%d/bias' % i] = \
                nn.conv2d_bwd(cache['a'][i], p['Encoder/enc_conv2D_%d/kernel' % i], dc, 2)
        return g

    def vae_phase(self, p, x, eps, mask_mu=None, mask_sg=None, caches=None):
        n = x.shape[0]
        z, kl, ec = self.vae_enc_forward(p, x, eps, mask_mu, mask_sg)
        out, gc = self.gen_forward(p, z)
        if caches is not None:
            caches.update(enc=ec, gen=gc, disc=[])
        l1 = np.abs(x - out)
        rec = l1.reshape(n, -1).sum(axis=1).mean()
        losses = {'reconstructionLoss': rec, 'loss': rec, 'kl': kl.mean(), 'enc_loss': rec + self.kl_weight * kl.mean(), 'L1': l1,
                  'reconstruction': out, 'z_mu': ec['mu'], 'z_sigma': ec['sg']}
        g_gen, dz = self.gen_backward(p, gc, np.sign(out - x) / n)
        grads = self.vae_enc_backward(p, ec, dz, self.kl_weight / n)
        grads.update(g_gen)
        return losses, grads

    def gen_phase(self, p, x, eps, mask_mu=None, mask_sg=None, caches=None):
        z, _, ec = self.vae_enc_forward(p, x, eps, mask_mu, mask_sg)
        out, gc = self.gen_forward(p, z)
        _, d, dcache = self.disc_forward(p, out)
        if caches is not None:
            caches.update(enc=ec, gen=gc, disc=[dcache])
        _, dx = self.disc_backward(p, dcache, dd=np.full_like(d, -1.0 / d.size), want_params=False)
        grads, _ = self.gen_backward(p, gc, dx)
        return {'gen_loss': -d.mean(), 'reconstruction': out}, grads

    def disc_phase(self, p, x, eps, alpha, mask_mu=None, mask_sg=None, caches=None):
        z, _, ec = self.vae_enc_forward(p, x, eps, mask_mu, mask_sg)
        out, gcache = self.gen_forward(p, z)
        _, d_fake, c_fake = self.disc_forward(p, out)
        _, d_real, c_real = self.disc_forward(p, x)
        x_hat = x + alpha.reshape(-1, 1, 1, 1).astype(x.dtype) * (out - x)
        _, _, c_hat = self.disc_forward(p, x_hat)
        if caches is not None:
            caches.update(enc=ec, gen=gcache, disc=[c_fake, c_real, c_hat])
        ddx, tape = self.disc_input_grad(p, c_hat)
        pen, gbar = self.gradient_penalty(ddx)
        losses = {'disc_fake': d_fake.mean(), 'disc_real': d_real.mean(), 'penalty': pen}
        losses['disc_loss'] = losses['disc_fake'] - losses['disc_real'] + pen
        g_f, _ = self.disc_backward(p, c_fake, dd=np.full_like(d_fake, 1.0 / d_fake.size))
        g_r, _ = self.disc_backward(p, c_real, dd=np.full_like(d_real, -1.0 / d_real.size))
        g_2, inject = self.disc_penalty_grads(p, c_hat, tape, gbar)
        g_3, _ = self.disc_backward(p, c_hat, inject=inject)
        grads = {}
        for part in (g_f, g_r, g_2, g_3):
            for k, v in part.items():
                grads[k] = grads.get(k, 0) + v
        return losses, grads

    def reconstruct(self, p, x, eps=None):
        z, _, _ = self.vae_enc_forward(p, x, np.zeros((x.shape[0], self.zdim), x.dtype) if eps is None else eps)
        return self.gen_forward(p, z)[0]
