"""Oracle for the spatial Gaussian-mixture VAE: train step, losses and the restoration-mode input gradient.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED (no TF): cross-checked against an independent
torch-autograd graph in tests/test_oracle_gmvae.py.

Restates, with hand-written backward passes:
  models/gaussian_mixture_variational_autoencoder_spatial.py:9-65   graph (encoder, 1x1 heads q(w|x) q(z|x), p(z|w,c),
                                                                     decoder applied to the ENCODER FEATURE MAP :52-56, p(c))
  trainers/GMVAE_spatial.py:61-97                                    four-term loss, TV restore term, grads w.r.t. x
  trainers/GMVAE_spatial.py:168-199                                  reconstruct(): restore_steps x (x -= restore_lr * grads)
  models/customlayers.py:16-38, trainers/DLMODEL.py:112-131          trunk + Adam as in oracle/vae.py
Semantics that differ from the VAE path (SURVEY.md §8a note 5): log_sigma here is a log-VARIANCE (sigma = exp(0.5 ls));
z_wc_log_sigma_inv is a log inverse variance with a trainable +0.1 offset (`tf.Variable`, :43).
All noise (e_w, e_z; the unused z_wc sample is not drawn) is an explicit input.
"""
import math

import numpy as np

from . import nn
from .vae import act_bwd

LRELU_ALPHA = 0.3


def param_spec(height, width, channels, inter_res, dim_c, dim_z, dim_w):
    """[(name, shape, kind)] in TF variable-creation order.  The model opens no variable scope, so the unnamed BN layers
    are numbered across encoder and decoder in call order."""
    assert height == width
    n_pool = int(math.log(height, 2) - math.log(float(inter_res), 2))
    spec = []
    cin = channels
    bn = 0

    def bn_name():
        nonlocal bn
        s = 'batch_normalization' if bn == 0 else f'batch_normalization_{bn}'
        bn += 1
        return s
    for i in range(n_pool):
        f = int(min(128, 32 * (2 ** i)))
        b = bn_name()
        spec += [(f'enc_conv2D_{i}/kernel', (5, 5, cin, f), 'conv_w'), (f'enc_conv2D_{i}/bias', (f,), 'bias'),
                 (b + '/gamma', (f,), 'gamma'), (b + '/beta', (f,), 'beta')]
        cin = f
    cenc = cin
    for nm, co in (('q_wz_x/w_mu', dim_w), ('q_wz_x/w_log_sigma', dim_w), ('q_wz_x/z_mu', dim_z),
                   ('q_wz_x/z_log_sigma', dim_z)):
        spec += [(nm + '/kernel', (1, 1, cenc, co), 'conv_w'), (nm + '/bias', (co,), 'bias')]
    spec += [('p_z_wc/1x1convlayer/kernel', (1, 1, dim_w, 64), 'conv_w'), ('p_z_wc/1x1convlayer/bias', (64,), 'bias'),
             ('p_z_wc/z_wc_mu/kernel', (1, 1, 64, dim_z * dim_c), 'conv_w'), ('p_z_wc/z_wc_mu/bias', (dim_z * dim_c,), 'bias'),
             ('p_z_wc/z_wc_log_sigma/kernel', (1, 1, 64, dim_z * dim_c), 'conv_w'),
             ('p_z_wc/z_wc_log_sigma/bias', (dim_z * dim_c,), 'bias'),
             ('Variable', (dim_z * dim_c,), 'const0.1')]
    b = bn_name()
    spec += [(b + '/gamma', (cenc,), 'gamma'), (b + '/beta', (cenc,), 'beta')]
    for i in range(n_pool):
        f = int(max(32, 128 / (2 ** i)))
        b = bn_name()
        spec += [(f'dec_Conv2DT_{i}/kernel', (5, 5, f, cin), 'conv_w'), (f'dec_Conv2DT_{i}/bias', (f,), 'bias'),
                 (b + '/gamma', (f,), 'gamma'), (b + '/beta', (f,), 'beta')]
        cin = f
    spec += [('dec_Conv2D_final/kernel', (1, 1, cin, channels), 'conv_w'), ('dec_Conv2D_final/bias', (channels,), 'bias')]
    return spec


def init_params(spec, seed=3, dtype=np.float32, perturb=False):
    rng = np.random.default_rng(seed)
    p = {}
    for name, shape, kind in spec:
        if kind == 'conv_w':
            p[name] = nn.glorot_uniform(rng, shape, dtype)
        elif kind == 'gamma':
            p[name] = np.ones(shape, dtype) + (rng.uniform(-0.2, 0.2, shape).astype(dtype) if perturb else 0)
        elif kind == 'const0.1':
            p[name] = np.full(shape, 0.1, dtype) + (rng.uniform(-0.05, 0.05, shape).astype(dtype) if perturb else 0)
        else:
            p[name] = np.zeros(shape, dtype) + (rng.uniform(-0.1, 0.1, shape).astype(dtype) if perturb else 0)
    return p


def total_variation(r):
    """tf.image.total_variation per image: sum |r[i+1,j]-r[i,j]| + sum |r[i,j+1]-r[i,j]| over H, W, C."""
    return np.abs(r[:, 1:] - r[:, :-1]).sum(axis=(1, 2, 3)) + np.abs(r[:, :, 1:] - r[:, :, :-1]).sum(axis=(1, 2, 3))


def total_variation_grad(r):
    g = np.zeros_like(r)
    sv = np.sign(r[:, 1:] - r[:, :-1])
    g[:, 1:] += sv
    g[:, :-1] -= sv
    sh = np.sign(r[:, :, 1:] - r[:, :, :-1])
    g[:, :, 1:] += sh
    g[:, :, :-1] -= sh
    return g


class GMVAE:
    def __init__(self, height=256, width=256, channels=1, inter_res=8, dim_c=9, dim_z=1, dim_w=1, c_lambda=1.0):
        self.h, self.w, self.c, self.inter = height, width, channels, inter_res
        self.dim_c, self.dim_z, self.dim_w, self.c_lambda = dim_c, dim_z, dim_w, float(c_lambda)
        self.spec = param_spec(height, width, channels, inter_res, dim_c, dim_z, dim_w)
        self.n_pool = int(math.log(height, 2) - math.log(float(inter_res), 2))
        names = [s[0] for s in self.spec if s[0].endswith('/gamma')]
        self.bn = [n[:-6] for n in names]      # BN scopes in call order: n_pool encoder, 1 decoder-input, n_pool decoder

    # ------------------------------------------------------------------ forward
    def forward(self, p, x, e_w, e_z):
        """e_w [N,h,w,dim_w], e_z [N,h,w,dim_z]: N(0,1) noise of the two reparameterisations (:27,32)."""
        cache = {'x': x}
        a = x
        for i in range(self.n_pool):
            c = nn.conv2d_fwd(a, p[f'enc_conv2D_{i}/kernel'], p[f'enc_conv2D_{i}/bias'], 2)
            bn = nn.bn_frozen_fwd(c, p[self.bn[i] + '/gamma'], p[self.bn[i] + '/beta'])
            cache[f'enc_in{i}'], cache[f'enc_c{i}'], cache[f'enc_bn{i}'] = a, c, bn
            a = nn.leaky_relu_fwd(bn, LRELU_ALPHA)
        h = a
        cache['h'] = h
        lin = lambda t, name: nn.conv2d_fwd(t, p[name + '/kernel'], p[name + '/bias'], 1)
        w_mu, w_ls = lin(h, 'q_wz_x/w_mu'), lin(h, 'q_wz_x/w_log_sigma')
        z_mu, z_ls = lin(h, 'q_wz_x/z_mu'), lin(h, 'q_wz_x/z_log_sigma')
        w_s = w_mu + e_w * np.exp(0.5 * w_ls)
        z_s = z_mu + e_z * np.exp(0.5 * z_ls)
        a7 = lin(w_s, 'p_z_wc/1x1convlayer')
        mid = np.maximum(a7, 0)
        n, hh, ww = x.shape[0], h.shape[1], h.shape[2]
        M = lin(mid, 'p_z_wc/z_wc_mu').reshape(n, hh, ww, self.dim_z, self.dim_c)
        Lq = (lin(mid, 'p_z_wc/z_wc_log_sigma') + p['Variable']).reshape(n, hh, ww, self.dim_z, self.dim_c)
        loglh = -0.5 * ((z_s[..., None] - M) ** 2 * np.exp(Lq)) - Lq + np.log(np.pi)
        logit = loglh.sum(axis=3)
        mx = logit.max(axis=-1, keepdims=True)
        ex = np.exp(logit - mx)
        pc = ex / ex.sum(axis=-1, keepdims=True)
        cache.update(w_mu=w_mu, w_ls=w_ls, z_mu=z_mu, z_ls=z_ls, w_s=w_s, z_s=z_s, a7=a7, mid=mid, M=M, Lq=Lq, pc=pc,
                     e_w=e_w, e_z=e_z)
        # decoder on the encoder feature map
        d = self.n_pool
        bn = nn.bn_frozen_fwd(h, p[self.bn[d] + '/gamma'], p[self.bn[d] + '/beta'])
        cache['dec_bn_in'] = bn
        a = nn.leaky_relu_fwd(bn, 0.0)
        for i in range(self.n_pool):
            c = nn.conv2d_transpose_fwd(a, p[f'dec_Conv2DT_{i}/kernel'], p[f'dec_Conv2DT_{i}/bias'], 2)
            b = nn.bn_frozen_fwd(c, p[self.bn[d + 1 + i] + '/gamma'], p[self.bn[d + 1 + i] + '/beta'])
            cache[f'dec_in{i}'], cache[f'dec_c{i}'], cache[f'dec_bn{i}'] = a, c, b
            a = nn.leaky_relu_fwd(b, LRELU_ALPHA)
        cache['dec_out'] = a
        xh = nn.conv2d_fwd(a, p['dec_Conv2D_final/kernel'], p['dec_Conv2D_final/bias'], 1)
        out = {'xz_mu': xh, 'w_mu': w_mu, 'w_log_sigma': w_ls, 'z_mu': z_mu, 'z_log_sigma': z_ls, 'w_sampled': w_s,
               'z_sampled': z_s, 'z_wc_mus': M, 'z_wc_log_sigma_invs': Lq, 'pc_logit': logit, 'pc': pc}
        return out, cache

    # ------------------------------------------------------------------ losses (trainers/GMVAE_spatial.py:61-97)
    def losses(self, x, out, tv_lambda=0.0):
        n = x.shape[0]
        l1 = np.abs(x - out['xz_mu'])
        z_mu, z_ls, M, Lq, pc = out['z_mu'], out['z_log_sigma'], out['z_wc_mus'], out['z_wc_log_sigma_invs'], out['pc']
        kl = 0.5 * ((np.exp(z_ls)[..., None] + (z_mu[..., None] - M) ** 2) * (np.exp(Lq) + 1e-6) - (Lq + z_ls[..., None]) - 1)
        con = (kl * pc[:, :, :, None, :]).sum(axis=(1, 2, 3, 4))
        w_mu, w_ls = out['w_mu'], out['w_log_sigma']
        wl = 0.5 * (w_mu ** 2 + np.exp(w_ls) - w_ls - 1).sum(axis=(1, 2, 3))
        cl1 = (pc * np.log(pc * self.dim_c + 1e-8)).sum(axis=3)
        cl = np.maximum(cl1, self.c_lambda).sum(axis=(1, 2))
        res = {'L1': l1, 'L1_sum': l1.reshape(n, -1).sum(1), 'L2': (x - out['xz_mu']) ** 2}
        res['L2_sum'] = res['L2'].sum()
        res['reconstructionLoss'] = res['mean_p_loss'] = res['L1_sum'].mean()
        res['conditional_prior_loss'] = con.mean()
        res['w_prior_loss'] = wl.mean()
        res['c_prior_loss'] = cl.mean()
        res['loss'] = res['mean_p_loss'] + res['conditional_prior_loss'] + res['w_prior_loss'] + res['c_prior_loss']
        res['restore'] = tv_lambda * total_variation(x - out['xz_mu'])
        return res

    # ------------------------------------------------------------------ backward
    def backward(self, p, x, out, cache, tv_lambda=None, act=None):
        """tv_lambda None: d loss / d params (the optimizer's objective, :94-97) and g['__dx'] = d loss / d x.
        tv_lambda given: the same backward of  loss + sum_n tv_lambda * TV_n(x - xz_mu)  (the `grads` fetch, :91-92);
        only g['__dx'] is meaningful for the caller then (parameter entries hold the gradient of that other objective).
        act: activation pattern of another fp32 implementation for the trunk's kinks (oracle/vae.py: act_bwd)."""
        n = x.shape[0]
        dt = x.dtype.type
        # `grads` = tf.gradients(loss + restore, x): `loss + restore` has shape [n] (scalar + per-image TV) and tf.gradients differentiates the
        # SUM of its elements = n * loss + sum_n restore_n, i.e. every sample's own loss terms enter with weight 1 (not 1/n)
        inv = dt(1.0 / n) if tv_lambda is None else dt(1.0)
        g = {}
        C = self.dim_c
        # ---- decoder ----
        gx = (act['l1_sign'].astype(x.dtype) if act is not None and 'l1_sign' in act else np.sign(out['xz_mu'] - x)) * inv
        dx_direct = -gx
        if tv_lambda is not None:
            tvg = total_variation_grad(x - out['xz_mu']) * dt(tv_lambda)
            gx = gx - tvg
            dx_direct = dx_direct + tvg
        da, g['dec_Conv2D_final/kernel'], g['dec_Conv2D_final/bias'] = \
            nn.conv2d_bwd(cache['dec_out'], p['dec_Conv2D_final/kernel'], gx, 1)
        d = self.n_pool
        for i in reversed(range(self.n_pool)):
            b = self.bn[d + 1 + i]
            dbn = act_bwd(cache, f'dec_bn{i}', da, LRELU_ALPHA, act)
            dc, g[b + '/gamma'], g[b + '/beta'] = nn.bn_frozen_bwd(cache[f'dec_c{i}'], p[b + '/gamma'], dbn)
            da, g[f'dec_Conv2DT_{i}/kernel'], g[f'dec_Conv2DT_{i}/bias'] = \
                nn.conv2d_transpose_bwd(cache[f'dec_in{i}'], p[f'dec_Conv2DT_{i}/kernel'], dc, 2)
        dbn = act_bwd(cache, 'dec_bn_in', da, 0.0, act)
        dh, g[self.bn[d] + '/gamma'], g[self.bn[d] + '/beta'] = nn.bn_frozen_bwd(cache['h'], p[self.bn[d] + '/gamma'], dbn)
        # ---- latent heads (per location) ----
        c = cache
        pc, M, Lq = c['pc'], c['M'], c['Lq']
        z_mu, z_ls, z_s, w_mu, w_ls = c['z_mu'], c['z_ls'], c['z_s'], c['w_mu'], c['w_ls']
        E = np.exp(Lq)
        E6 = E + 1e-6
        V = np.exp(z_ls)[..., None]
        D2 = z_mu[..., None] - M
        kl = 0.5 * ((V + D2 ** 2) * E6 - (Lq + z_ls[..., None]) - 1)
        dkl = inv * np.broadcast_to(pc[:, :, :, None, :], kl.shape)
        dpc = inv * kl.sum(axis=3)
        cl1 = (pc * np.log(pc * C + 1e-8)).sum(axis=3)
        c_on = (cl1 >= self.c_lambda)[..., None]          # tf.maximum routes the gradient to x where x >= y
        dpc = dpc + inv * c_on * (np.log(pc * C + 1e-8) + pc * C / (pc * C + 1e-8))
        dlogit = pc * (dpc - (dpc * pc).sum(axis=-1, keepdims=True))
        dll = np.broadcast_to(dlogit[:, :, :, None, :], kl.shape)
        D = z_s[..., None] - M
        dz_s = (dll * (-D * E)).sum(axis=-1)
        dM = dll * (D * E) - dkl * D2 * E6
        dLq = dll * (-0.5 * D ** 2 * E - 1) + dkl * 0.5 * ((V + D2 ** 2) * E - 1)
        dz_mu = (dkl * D2 * E6).sum(axis=-1) + dz_s
        dz_ls = (dkl * 0.5 * (V * E6 - 1)).sum(axis=-1) + dz_s * c['e_z'] * 0.5 * np.exp(0.5 * z_ls)
        nb, hh, ww = pc.shape[:3]
        dMf, dLqf = dM.reshape(nb, hh, ww, -1), dLq.reshape(nb, hh, ww, -1)
        g['Variable'] = dLqf.sum(axis=(0, 1, 2))
        dmid1, g['p_z_wc/z_wc_mu/kernel'], g['p_z_wc/z_wc_mu/bias'] = \
            nn.conv2d_bwd(c['mid'], p['p_z_wc/z_wc_mu/kernel'], dMf, 1)
        dmid2, g['p_z_wc/z_wc_log_sigma/kernel'], g['p_z_wc/z_wc_log_sigma/bias'] = \
            nn.conv2d_bwd(c['mid'], p['p_z_wc/z_wc_log_sigma/kernel'], dLqf, 1)
        da7 = (dmid1 + dmid2) * (c['a7'] > 0)
        dw_s, g['p_z_wc/1x1convlayer/kernel'], g['p_z_wc/1x1convlayer/bias'] = \
            nn.conv2d_bwd(c['w_s'], p['p_z_wc/1x1convlayer/kernel'], da7, 1)
        dw_mu = inv * w_mu + dw_s
        dw_ls = inv * 0.5 * (np.exp(w_ls) - 1) + dw_s * c['e_w'] * 0.5 * np.exp(0.5 * w_ls)
        for name, dv in (('q_wz_x/w_mu', dw_mu), ('q_wz_x/w_log_sigma', dw_ls), ('q_wz_x/z_mu', dz_mu),
                         ('q_wz_x/z_log_sigma', dz_ls)):
            dhh, g[name + '/kernel'], g[name + '/bias'] = nn.conv2d_bwd(c['h'], p[name + '/kernel'], dv, 1)
            dh = dh + dhh
        # ---- encoder ----
        da = dh
        for i in reversed(range(self.n_pool)):
            b = self.bn[i]
            dbn = act_bwd(cache, f'enc_bn{i}', da, LRELU_ALPHA, act)
            dc, g[b + '/gamma'], g[b + '/beta'] = nn.bn_frozen_bwd(cache[f'enc_c{i}'], p[b + '/gamma'], dbn)
            da, g[f'enc_conv2D_{i}/kernel'], g[f'enc_conv2D_{i}/bias'] = \
                nn.conv2d_bwd(cache[f'enc_in{i}'], p[f'enc_conv2D_{i}/kernel'], dc, 2)
        g['__dx'] = da + dx_direct
        return g

    # ------------------------------------------------------------------
    def new_opt(self, p):
        return {'t': 0, 'm': {k: np.zeros_like(v) for k, v in p.items()},
                'v': {k: np.zeros_like(v) for k, v in p.items()}}

    def train_step(self, p, opt, x, e_w, e_z, lr=5e-5, beta1=0.5):
        out, cache = self.forward(p, x, e_w, e_z)
        ls = self.losses(x, out)
        g = self.backward(p, x, out, cache)
        opt['t'] += 1
        for name, _, _ in self.spec:
            nn.adam_tf_step(p[name], g[name], opt['m'][name], opt['v'][name], opt['t'], lr, beta1)
        return out, ls, g

    def restore_grads(self, p, x, e_w, e_z, tv_lambda):
        out, cache = self.forward(p, x, e_w, e_z)
        return self.backward(p, x, out, cache, tv_lambda=tv_lambda)['__dx']

    def reconstruct(self, p, x, noise, restore_steps=150, restore_lr=1e-3, tv_lambda=1.8):
        """trainers/GMVAE_spatial.py:168-199.  noise: callable step -> (e_w, e_z) (the graph draws fresh noise on every
        sess.run); restore_steps == 0 returns the plain decoder output."""
        if x.ndim < 4:
            x = x[None]
        if restore_steps == 0:
            e_w, e_z = noise(0)
            rec = self.forward(p, x, e_w, e_z)[0]['xz_mu']
        else:
            rec = x.copy()
            for step in range(restore_steps):
                e_w, e_z = noise(step)
                rec = rec - x.dtype.type(restore_lr) * self.restore_grads(p, rec, e_w, e_z, tv_lambda)
        return {'reconstruction': rec, 'l1err': np.sum(np.abs(x - rec)), 'l2err': np.sum(np.sqrt((x - rec) ** 2))}
