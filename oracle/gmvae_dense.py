"""Oracle for the dense GMVAE: models/gaussian_mixture_variational_autoencoder.py:11-76 + trainers/GMVAE.py:56-101 (losses, the
`grads` fetch of the restoration) and :158-188 (reconstruct / restoration loop).  numpy forward, hand-written backward.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED (no TensorFlow here, no golden vectors in the reference);
tests/test_oracle_gmvae_dense.py anchors every gradient on torch autograd in float64.

Graph: unified encoder -> 1x1 conv C/8 -> flatten -> four Dense heads  w_mu, w_log_sigma, z_mu (dropout on these three, :37-41),
       z_log_sigma (its Dropout is called WITHOUT `training`, :42 -> learning phase 0 -> identity);
       w_s = w_mu + e_w exp(w_ls / 2), z_s = z_mu + e_z exp(z_ls / 2);
       p(z|w,c): Dense(dim_z dim_c) x 2 straight on w_s (no hidden layer, :48-53) + the 0.1 bias Variable, reshaped [dim_z, dim_c];
       decoder: dropout(Dense(flat)(z_s)) -> 1x1 conv C -> unified decoder.
Variable names (layers get their scope name at first CALL): Bottleneck/conv2d, Bottleneck/dense .. dense_3 (the heads, call order),
Bottleneck/dense_4 (dec_dense), Bottleneck/conv2d_1, then un-scoped dense, dense_1, Variable, then Decoder/*."""
import numpy as np

from . import nn
from .aae import AAE
from .gmvae import total_variation, total_variation_grad

HEADS = ('w_mu', 'w_ls', 'z_mu', 'z_ls')


def param_spec(height=128, inter_res=8, dim_c=6, dim_z=1, dim_w=1, channels=1):
    npool = int(round(np.log2(height) - np.log2(inter_res)))
    spec, cin = [], channels
    for i in range(npool):
        f = min(128, 32 * 2 ** i)
        bn = 'Encoder/batch_normalization' + ('' if i == 0 else '_%d' % i)
        spec += [('Encoder/enc_conv2D_%d/kernel' % i, (5, 5, cin, f), 'conv_w'), ('Encoder/enc_conv2D_%d/bias' % i, (f,), 'bias'),
                 (bn + '/gamma', (f,), 'gamma'), (bn + '/beta', (f,), 'beta')]
        cin = f
    cenc, cmid = cin, cin // 8
    flat = inter_res * inter_res * cmid
    spec += [('Bottleneck/conv2d/kernel', (1, 1, cenc, cmid), 'conv_w'), ('Bottleneck/conv2d/bias', (cmid,), 'bias')]
    for k, d in enumerate((dim_w, dim_w, dim_z, dim_z)):
        nm = 'Bottleneck/dense' + ('' if k == 0 else '_%d' % k)
        spec += [(nm + '/kernel', (flat, d), 'dense_w'), (nm + '/bias', (d,), 'bias')]
    spec += [('Bottleneck/dense_4/kernel', (dim_z, flat), 'dense_w'), ('Bottleneck/dense_4/bias', (flat,), 'bias'),
             ('Bottleneck/conv2d_1/kernel', (1, 1, cmid, cenc), 'conv_w'), ('Bottleneck/conv2d_1/bias', (cenc,), 'bias'),
             ('dense/kernel', (dim_w, dim_z * dim_c), 'dense_w'), ('dense/bias', (dim_z * dim_c,), 'bias'),
             ('dense_1/kernel', (dim_w, dim_z * dim_c), 'dense_w'), ('dense_1/bias', (dim_z * dim_c,), 'bias'),
             ('Variable', (dim_z * dim_c,), 'const0.1'),
             ('Decoder/batch_normalization/gamma', (cenc,), 'gamma'), ('Decoder/batch_normalization/beta', (cenc,), 'beta')]
    cin = cenc
    for i in range(npool):
        f = max(32, 128 // 2 ** i)
        spec += [('Decoder/dec_Conv2DT_%d/kernel' % i, (5, 5, f, cin), 'conv_w'), ('Decoder/dec_Conv2DT_%d/bias' % i, (f,), 'bias'),
                 ('Decoder/batch_normalization_%d/gamma' % (i + 1), (f,), 'gamma'), ('Decoder/batch_normalization_%d/beta' % (i + 1), (f,), 'beta')]
        cin = f
    spec += [('Decoder/dec_Conv2D_final/kernel', (1, 1, cin, channels), 'conv_w'), ('Decoder/dec_Conv2D_final/bias', (channels,), 'bias')]
    return spec


def init_params(spec, seed=3, dtype=np.float64, perturb=True):
    """glorot-uniform kernels, zero biases, gamma 1 / beta 0, Variable 0.1 (the TF initial values); perturb=True moves biases / BN
    parameters off their trivial values so that every gradient path is exercised."""
    rng = np.random.default_rng(seed)
    p = {}
    for name, shape, kind in spec:
        if kind in ('conv_w', 'dense_w'):
            rf = int(np.prod(shape[:-2])) if len(shape) == 4 else 1
            fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            p[name] = rng.uniform(-lim, lim, shape)
        elif kind == 'gamma':
            p[name] = np.ones(shape) + (0.1 * rng.standard_normal(shape) if perturb else 0)
        elif kind == 'const0.1':
            p[name] = np.full(shape, 0.1) + (0.05 * rng.standard_normal(shape) if perturb else 0)
        else:
            p[name] = 0.05 * rng.standard_normal(shape) if perturb else np.zeros(shape)
        p[name] = np.asarray(p[name], dtype)
    return p


class GMVAEDense:
    def __init__(self, height=128, inter_res=8, dim_c=6, dim_z=1, dim_w=1, c_lambda=1.0):
        self.height, self.inter_res = height, inter_res
        self.dim_c, self.dim_z, self.dim_w, self.c_lambda = dim_c, dim_z, dim_w, float(c_lambda)
        self.spec = param_spec(height, inter_res, dim_c, dim_z, dim_w)
        # the conv encoder / decoder stacks and the decoder-side bottleneck are those of the dense autoencoders
        self.core = AAE('constrained_ae', height, inter_res, dim_z)
        self.core.nm = {'conv': 'Bottleneck/conv2d', 'z': None, 'dec': 'Bottleneck/dense_4', 'rev': 'Bottleneck/conv2d_1'}
        self.head = {h: 'Bottleneck/dense' + ('' if k == 0 else '_%d' % k) for k, h in enumerate(HEADS)}

    # ------------------------------------------------------------------ forward
    def forward(self, p, x, e_w, e_z, masks=None):
        """e_w [n,dim_w], e_z [n,dim_z] N(0,1); masks: None or dict with 'w_mu', 'w_ls' [n,dim_w], 'z_mu' [n,dim_z], 'dec' [n,flat]
        (keep / (1 - rate))."""
        masks = masks or {}
        core, n = self.core, x.shape[0]
        ec = {'a': [x], 'c': []}
        a = x
        for i in range(core.npool):
            c = nn.conv2d_fwd(a, p['Encoder/enc_conv2D_%d/kernel' % i], p['Encoder/enc_conv2D_%d/bias' % i], 2)
            a = nn.leaky_relu_fwd(nn.bn_frozen_fwd(c, p[core.bn_e[i] + '/gamma'], p[core.bn_e[i] + '/beta']), 0.3)
            ec['c'].append(c); ec['a'].append(a)
        t = nn.conv2d_fwd(a, p['Bottleneck/conv2d/kernel'], p['Bottleneck/conv2d/bias'], 1)
        flat = t.reshape(n, -1)
        hv = {}
        for h in HEADS:
            v = nn.dense_fwd(flat, p[self.head[h] + '/kernel'], p[self.head[h] + '/bias'])
            if masks.get(h) is not None:
                v = v * masks[h]
            hv[h] = v
        w_s = hv['w_mu'] + e_w * np.exp(0.5 * hv['w_ls'])
        z_s = hv['z_mu'] + e_z * np.exp(0.5 * hv['z_ls'])
        M = nn.dense_fwd(w_s, p['dense/kernel'], p['dense/bias']).reshape(n, self.dim_z, self.dim_c)
        Lq = (nn.dense_fwd(w_s, p['dense_1/kernel'], p['dense_1/bias']) + p['Variable']).reshape(n, self.dim_z, self.dim_c)
        loglh = -0.5 * ((z_s[..., None] - M) ** 2 * np.exp(Lq)) - Lq + np.log(np.pi)
        logit = loglh.sum(axis=1)
        ex = np.exp(logit - logit.max(axis=-1, keepdims=True))
        pc = ex / ex.sum(axis=-1, keepdims=True)
        xh, dcache = core.decode(p, z_s, masks.get('dec'))
        out = {'xz_mu': xh, 'w_mu': hv['w_mu'], 'w_log_sigma': hv['w_ls'], 'z_mu': hv['z_mu'], 'z_log_sigma': hv['z_ls'],
               'w_sampled': w_s, 'z_sampled': z_s, 'z_wc_mus': M, 'z_wc_log_sigma_invs': Lq, 'pc_logit': logit, 'pc': pc}
        cache = {'enc': ec, 't': t, 'flat': flat, 'dec': dcache, 'masks': masks, 'e_w': e_w, 'e_z': e_z}
        return out, cache

    # ------------------------------------------------------------------ losses (trainers/GMVAE.py:56-93)
    def losses(self, x, out, tv_lambda=0.0):
        n = x.shape[0]
        l1 = np.abs(x - out['xz_mu'])
        z_mu, z_ls, M, Lq, pc = out['z_mu'], out['z_log_sigma'], out['z_wc_mus'], out['z_wc_log_sigma_invs'], out['pc']
        kl = 0.5 * ((np.exp(z_ls)[..., None] + (z_mu[..., None] - M) ** 2) * (np.exp(Lq) + 1e-6) - (Lq + z_ls[..., None]) - 1)
        con = (kl * pc[:, None, :]).sum(axis=(1, 2))
        wl = 0.5 * (out['w_mu'] ** 2 + np.exp(out['w_log_sigma']) - out['w_log_sigma'] - 1).sum(axis=1)
        cl = np.maximum((pc * np.log(pc * self.dim_c + 1e-8)).sum(axis=1), self.c_lambda)
        res = {'L1': l1, 'L1_sum': l1.reshape(n, -1).sum(1), 'L2': (x - out['xz_mu']) ** 2}
        res['L2_sum'] = res['L2'].sum()
        res['reconstructionLoss'] = res['mean_p_loss'] = res['L1_sum'].mean()
        res['conditional_prior_loss'], res['w_prior_loss'], res['c_prior_loss'] = con.mean(), wl.mean(), cl.mean()
        res['loss'] = res['mean_p_loss'] + res['conditional_prior_loss'] + res['w_prior_loss'] + res['c_prior_loss']
        res['restore'] = tv_lambda * total_variation(x - out['xz_mu'])
        return res

    # ------------------------------------------------------------------ backward
    def backward(self, p, x, out, cache, tv_lambda=None):
        """tv_lambda None: d loss / d params and g['__dx'] = d loss / d x.  tv_lambda given: the backward of
        loss + sum_n tv_lambda TV_n(x - xz_mu) (the `grads` fetch, trainers/GMVAE.py:93-94); only g['__dx'] is meaningful then."""
        core, n = self.core, x.shape[0]
        dt = x.dtype.type
        # `grads` = tf.gradients(loss + restore, x): `loss + restore` has shape [n] (scalar + per-image TV) and tf.gradients differentiates the
        # SUM of its elements = n * loss + sum_n restore_n, i.e. every sample's own loss terms enter with weight 1 (not 1/n)
        inv = dt(1.0 / n) if tv_lambda is None else dt(1.0)
        g = {}
        C = self.dim_c
        gx = np.sign(out['xz_mu'] - x) * inv
        dx_direct = -gx
        if tv_lambda is not None:
            tvg = total_variation_grad(x - out['xz_mu']) * dt(tv_lambda)
            gx = gx - tvg
            dx_direct = dx_direct + tvg
        dz_dec = core.decode_backward(p, cache['dec'], gx, g)
        # ---- latent terms (per sample) ----
        pc, M, Lq = out['pc'], out['z_wc_mus'], out['z_wc_log_sigma_invs']
        z_mu, z_ls, z_s, w_mu, w_ls, w_s = out['z_mu'], out['z_log_sigma'], out['z_sampled'], out['w_mu'], out['w_log_sigma'], out['w_sampled']
        E = np.exp(Lq); E6 = E + 1e-6
        V = np.exp(z_ls)[..., None]
        D2 = z_mu[..., None] - M
        kl = 0.5 * ((V + D2 ** 2) * E6 - (Lq + z_ls[..., None]) - 1)
        dkl = inv * np.broadcast_to(pc[:, None, :], kl.shape)
        dpc = inv * kl.sum(axis=1)
        cl1 = (pc * np.log(pc * C + 1e-8)).sum(axis=1)
        act = (cl1 >= self.c_lambda)[..., None]              # tf.maximum routes the gradient to x where x >= y
        dpc = dpc + inv * act * (np.log(pc * C + 1e-8) + pc * C / (pc * C + 1e-8))
        dlogit = pc * (dpc - (dpc * pc).sum(axis=-1, keepdims=True))
        dll = np.broadcast_to(dlogit[:, None, :], kl.shape)
        D = z_s[..., None] - M
        dz_s = (dll * (-D * E)).sum(axis=-1) + dz_dec
        dM = dll * (D * E) - dkl * D2 * E6
        dLq = dll * (-0.5 * D ** 2 * E - 1) + dkl * 0.5 * ((V + D2 ** 2) * E - 1)
        d = {'z_mu': (dkl * D2 * E6).sum(axis=-1) + dz_s,
             'z_ls': (dkl * 0.5 * (V * E6 - 1)).sum(axis=-1) + dz_s * cache['e_z'] * 0.5 * np.exp(0.5 * z_ls)}
        dMf, dLqf = dM.reshape(n, -1), dLq.reshape(n, -1)
        g['Variable'] = dLqf.sum(axis=0)
        dws1, g['dense/kernel'], g['dense/bias'] = nn.dense_bwd(w_s, p['dense/kernel'], dMf)
        dws2, g['dense_1/kernel'], g['dense_1/bias'] = nn.dense_bwd(w_s, p['dense_1/kernel'], dLqf)
        dw_s = dws1 + dws2
        d['w_mu'] = inv * w_mu + dw_s
        d['w_ls'] = inv * 0.5 * (np.exp(w_ls) - 1) + dw_s * cache['e_w'] * 0.5 * np.exp(0.5 * w_ls)
        dflat = 0
        for h in HEADS:
            dv = d[h] if cache['masks'].get(h) is None else d[h] * cache['masks'][h]
            df, g[self.head[h] + '/kernel'], g[self.head[h] + '/bias'] = nn.dense_bwd(cache['flat'], p[self.head[h] + '/kernel'], dv)
            dflat = dflat + df
        # ---- encoder ----
        ec = cache['enc']
        da, g['Bottleneck/conv2d/kernel'], g['Bottleneck/conv2d/bias'] = \
            nn.conv2d_bwd(ec['a'][-1], p['Bottleneck/conv2d/kernel'], dflat.reshape(cache['t'].shape), 1)
        for i in reversed(range(core.npool)):
            c = ec['c'][i]
            bnv = nn.bn_frozen_fwd(c, p[core.bn_e[i] + '/gamma'], p[core.bn_e[i] + '/beta'])
            dc, g[core.bn_e[i] + '/gamma'], g[core.bn_e[i] + '/beta'] = nn.bn_frozen_bwd(c, p[core.bn_e[i] + '/gamma'], nn.leaky_relu_bwd(bnv, da, 0.3))
            da, g['Encoder/enc_conv2D_%d/kernel' % i], g['Encoder/enc_conv2D_%d/bias' % i] = \
                nn.conv2d_bwd(ec['a'][i], p['Encoder/enc_conv2D_%d/kernel' % i], dc, 2)
        g['__dx'] = da + dx_direct
        return g

    # ------------------------------------------------------------------
    def new_opt(self, p):
        return {'t': 0, 'm': {k: np.zeros_like(v) for k, v in p.items()}, 'v': {k: np.zeros_like(v) for k, v in p.items()}}

    def train_step(self, p, opt, x, e_w, e_z, masks=None, lr=1e-4, beta1=0.5):
        out, cache = self.forward(p, x, e_w, e_z, masks)
        ls = self.losses(x, out)
        g = self.backward(p, x, out, cache)
        opt['t'] += 1
        for name, _, _ in self.spec:
            nn.adam_tf_step(p[name], g[name], opt['m'][name], opt['v'][name], opt['t'], lr, beta1)
        return out, ls, g

    def restore_grads(self, p, x, e_w, e_z, tv_lambda, masks=None):
        out, cache = self.forward(p, x, e_w, e_z, masks)
        return self.backward(p, x, out, cache, tv_lambda=tv_lambda)['__dx']

    def reconstruct(self, p, x, noise, restore_steps=150, restore_lr=1e-3, tv_lambda=1.8):
        """trainers/GMVAE.py:158-188.  noise: callable step -> (e_w, e_z) (the graph draws fresh noise on every sess.run)."""
        if x.ndim < 4:
            x = x[None]
        if restore_steps == 0:
            rec = self.forward(p, x, *noise(0))[0]['xz_mu']
        else:
            rec = x.copy()
            for step in range(restore_steps):
                rec = rec - x.dtype.type(restore_lr) * self.restore_grads(p, rec, *noise(step), tv_lambda)
        return {'reconstruction': rec, 'l1err': np.abs(x - rec).sum(), 'l2err': np.sqrt((x - rec) ** 2).sum()}
