"""Oracle for the dense-bottleneck AE / VAE / ceVAE train step and reconstruct().

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Primitives pinned by TF's published unit-test vectors (oracle/nn.py); the graph wiring is PARITY UNPINNED against TF output (no TF).

Restates, with hand-written backward passes:
  models/customlayers.py:16-38          unified encoder / decoder
  models/autoencoder.py:9-40            AE graph
  models/variational_autoencoder.py:9-47 VAE graph
  models/context_encoder_variational_autoencoder.py:9-59  ceVAE graph
  trainers/AE.py:28-29, VAE.py:36-42, ceVAE.py:38-51      losses
  trainers/CE.py:123-139                host-side context masking
  trainers/DLMODEL.py:112-131           Adam (TF form), beta1 from
                                        utils/default_config_setup.py:257
Parameter order = TF variable-creation order (Encoder, Bottleneck, Decoder).
The reparameterisation noise `eps` and the (pre-scaled) dropout masks are
explicit inputs because the TF graph RNG is unseeded.
"""
import math

import numpy as np

from . import nn

LRELU_ALPHA = 0.3  # keras LeakyReLU() default, customlayers.py:23,36


def dense_names(arch):
    """Variable scopes of the bottleneck Dense layers.  AE/VAE name them (autoencoder.py:26-27,
    variational_autoencoder.py:26-28); the ceVAE graph leaves them unnamed
    (context_encoder_variational_autoencoder.py:30-32), so keras numbers them in construction order."""
    if arch == 'ceVAE':
        return {'mu': 'Bottleneck/dense', 'sigma': 'Bottleneck/dense_1', 'dec': 'Bottleneck/dense_2'}
    return {'mu': 'Bottleneck/dense_mu', 'sigma': 'Bottleneck/dense_sigma', 'z': 'Bottleneck/dense_z',
            'dec': 'Bottleneck/dense_dec'}


def param_spec(arch, height, width, channels, inter_res, zdim):
    """[(name, shape, kind)] in TF variable-creation order."""
    assert arch in ('AE', 'VAE', 'ceVAE')
    assert height == width, 'reference derives num_pooling from input_shape[1] only (customlayers.py:18)'
    n_pool = int(math.log(height, 2) - math.log(float(inter_res), 2))
    spec = []
    cin = channels
    for i in range(n_pool):
        f = int(min(128, 32 * (2 ** i)))
        spec += [(f'Encoder/enc_conv2D_{i}/kernel', (5, 5, cin, f), 'conv_w'),
                 (f'Encoder/enc_conv2D_{i}/bias', (f,), 'bias'),
                 (f'Encoder/batch_normalization_{i}/gamma', (f,), 'gamma'),
                 (f'Encoder/batch_normalization_{i}/beta', (f,), 'beta')]
        cin = f
    cenc = cin
    cmid = cenc // 8
    flat = inter_res * inter_res * cmid
    spec += [('Bottleneck/conv2d/kernel', (1, 1, cenc, cmid), 'conv_w'),
             ('Bottleneck/conv2d/bias', (cmid,), 'bias')]
    nm = dense_names(arch)
    if arch != 'AE':
        spec += [(nm['mu'] + '/kernel', (flat, zdim), 'dense_w'),
                 (nm['mu'] + '/bias', (zdim,), 'bias'),
                 (nm['sigma'] + '/kernel', (flat, zdim), 'dense_w'),
                 (nm['sigma'] + '/bias', (zdim,), 'bias')]
    else:
        spec += [(nm['z'] + '/kernel', (flat, zdim), 'dense_w'),
                 (nm['z'] + '/bias', (zdim,), 'bias')]
    spec += [(nm['dec'] + '/kernel', (zdim, flat), 'dense_w'),
             (nm['dec'] + '/bias', (flat,), 'bias'),
             ('Bottleneck/conv2d_1/kernel', (1, 1, cmid, cenc), 'conv_w'),
             ('Bottleneck/conv2d_1/bias', (cenc,), 'bias'),
             ('Decoder/batch_normalization/gamma', (cenc,), 'gamma'),
             ('Decoder/batch_normalization/beta', (cenc,), 'beta')]
    cin = cenc
    for i in range(n_pool):
        f = int(max(32, 128 / (2 ** i)))
        spec += [(f'Decoder/dec_Conv2DT_{i}/kernel', (5, 5, f, cin), 'conv_w'),
                 (f'Decoder/dec_Conv2DT_{i}/bias', (f,), 'bias'),
                 (f'Decoder/batch_normalization_{i + 1}/gamma', (f,), 'gamma'),
                 (f'Decoder/batch_normalization_{i + 1}/beta', (f,), 'beta')]
        cin = f
    spec += [('Decoder/dec_Conv2D_final/kernel', (1, 1, cin, channels), 'conv_w'),
             ('Decoder/dec_Conv2D_final/bias', (channels,), 'bias')]
    return spec


def init_params(spec, seed=3, dtype=np.float32, perturb=False):
    """glorot_uniform kernels, zero bias, gamma=1, beta=0 (TF/keras defaults).
    perturb=True jitters bias/gamma/beta so parity tests exercise them."""
    rng = np.random.default_rng(seed)
    p = {}
    for name, shape, kind in spec:
        if kind in ('conv_w', 'dense_w'):
            p[name] = nn.glorot_uniform(rng, shape, dtype)
        elif kind == 'gamma':
            p[name] = np.ones(shape, dtype)
            if perturb:
                p[name] += rng.uniform(-0.2, 0.2, shape).astype(dtype)
        else:
            p[name] = np.zeros(shape, dtype)
            if perturb:
                p[name] += rng.uniform(-0.1, 0.1, shape).astype(dtype)
    return p


def flatten_params(spec, p):
    return np.concatenate([p[name].reshape(-1) for name, _, _ in spec])


def unflatten_params(spec, flat):
    p, off = {}, 0
    for name, shape, _ in spec:
        n = int(np.prod(shape))
        p[name] = flat[off:off + n].reshape(shape)
        off += n
    assert off == flat.size
    return p


def act_bwd(cache, key, da, alpha, act=None):
    """(Leaky)ReLU backward at cache[key].  `act` (tests) optionally carries, per key, the boolean "pre-activation counted as positive"
    pattern of ANOTHER fp32 implementation of the same step: a pre-activation within round-off of 0 lands on either side of the kink
    depending on summation order, and its derivative (alpha or 1) is then a legitimate property of that implementation, not an error
    (tests/gpu_util.py: device_activation_pattern)."""
    if act is not None and key in act:
        return np.where(act[key], da, da * da.dtype.type(alpha))
    return nn.leaky_relu_bwd(cache[key], da, alpha)


class Model:
    def __init__(self, arch, height=128, width=128, channels=1, inter_res=8, zdim=128):
        self.arch, self.h, self.w, self.c = arch, height, width, channels
        self.inter, self.zdim = inter_res, zdim
        self.spec = param_spec(arch, height, width, channels, inter_res, zdim)
        self.n_pool = int(math.log(height, 2) - math.log(float(inter_res), 2))
        self.nm = dense_names(arch)
        self.variational = arch != 'AE'

    # ------------------------------------------------------------------
    def forward(self, p, x, eps=None, masks=None):
        """masks: dict with optional pre-scaled keep masks
             VAE: 'mu','sigma' [N,zdim], 'dec' [N,flat]   (variational_autoencoder.py:31,32,35)
             AE : 'z' [N,zdim]                             (autoencoder.py:29; dec_dense dropout never active, :30 / A2)
           eps: [N,zdim] N(0,1) noise (VAE only; variational_autoencoder.py:34)."""
        masks = masks or {}
        cache = {'x': x}
        a = x
        for i in range(self.n_pool):
            pre = f'Encoder/enc_conv2D_{i}'
            bnp = f'Encoder/batch_normalization_{i}'
            c = nn.conv2d_fwd(a, p[pre + '/kernel'], p[pre + '/bias'], 2)
            bn = nn.bn_frozen_fwd(c, p[bnp + '/gamma'], p[bnp + '/beta'])
            cache[f'enc_in{i}'], cache[f'enc_c{i}'], cache[f'enc_bn{i}'] = a, c, bn
            a = nn.leaky_relu_fwd(bn, LRELU_ALPHA)
        cache['enc_out'] = a
        t = nn.conv2d_fwd(a, p['Bottleneck/conv2d/kernel'], p['Bottleneck/conv2d/bias'], 1)
        n = x.shape[0]
        cache['t_shape'] = t.shape
        flat = t.reshape(n, -1)  # H,W,C-major flatten
        cache['flat'] = flat
        out = {}
        if self.variational:
            mu = nn.dense_fwd(flat, p[self.nm['mu'] + '/kernel'], p[self.nm['mu'] + '/bias'])
            ls = nn.dense_fwd(flat, p[self.nm['sigma'] + '/kernel'], p[self.nm['sigma'] + '/bias'])
            if 'mu' in masks:
                mu = mu * masks['mu']
            if 'sigma' in masks:
                ls = ls * masks['sigma']
            sigma = np.exp(ls)
            if eps is None:
                eps = np.zeros_like(mu)
            z = mu + eps * sigma
            out.update(z_mu=mu, z_log_sigma=ls, z_sigma=sigma)
            cache.update(mu=mu, ls=ls, sigma=sigma, eps=eps)
        else:
            z = nn.dense_fwd(flat, p[self.nm['z'] + '/kernel'], p[self.nm['z'] + '/bias'])
            if 'z' in masks:
                z = z * masks['z']
            out['z'] = z
        cache['z'] = z
        d = nn.dense_fwd(z, p[self.nm['dec'] + '/kernel'], p[self.nm['dec'] + '/bias'])
        if self.variational and 'dec' in masks:
            d = d * masks['dec']
        d4 = d.reshape(cache['t_shape'])
        cache['d4'] = d4
        c = nn.conv2d_fwd(d4, p['Bottleneck/conv2d_1/kernel'], p['Bottleneck/conv2d_1/bias'], 1)
        bn = nn.bn_frozen_fwd(c, p['Decoder/batch_normalization/gamma'], p['Decoder/batch_normalization/beta'])
        cache['dec_c_in'], cache['dec_bn_in'] = c, bn
        a = nn.leaky_relu_fwd(bn, 0.0)  # ReLU, customlayers.py:31
        for i in range(self.n_pool):
            pre = f'Decoder/dec_Conv2DT_{i}'
            bnp = f'Decoder/batch_normalization_{i + 1}'
            c = nn.conv2d_transpose_fwd(a, p[pre + '/kernel'], p[pre + '/bias'], 2)
            bn = nn.bn_frozen_fwd(c, p[bnp + '/gamma'], p[bnp + '/beta'])
            cache[f'dec_in{i}'], cache[f'dec_c{i}'], cache[f'dec_bn{i}'] = a, c, bn
            a = nn.leaky_relu_fwd(bn, LRELU_ALPHA)
        cache['dec_out'] = a
        xh = nn.conv2d_fwd(a, p['Decoder/dec_Conv2D_final/kernel'], p['Decoder/dec_Conv2D_final/bias'], 1)
        out['x_hat'] = xh
        return out, cache

    # ------------------------------------------------------------------
    def losses(self, x, out):
        """trainers/AE.py:28-29 ; trainers/VAE.py:36-42.  KL uses the analytic
        2*log_sigma for log(sigma^2)."""
        l1 = np.abs(out['x_hat'] - x)
        rec = l1.reshape(x.shape[0], -1).sum(axis=1)
        res = {'L1': l1, 'reconstructionLoss': rec.mean()}
        if self.variational:
            mu, ls, sg = out['z_mu'], out['z_log_sigma'], out['z_sigma']
            kl = 0.5 * (mu * mu + sg * sg - 2.0 * ls - 1.0).sum(axis=1)
            res['kl'] = kl.mean()
            res['loss'] = (rec + kl).mean()
        else:
            res['loss'] = res['reconstructionLoss']
        return res

    # ------------------------------------------------------------------
    def backward(self, p, x, out, cache, masks=None, kl=True, gx=None, klw=None, act=None):
        """Gradient of losses()['loss'] w.r.t. every parameter (dict by name).  act: see act_bwd() (adds key 'l1_sign' = the other
        implementation's sign(x_hat - x)).  kl=False drops the KL term
        (the ceVAE context branch, whose loss is the reconstruction sum only: trainers/ceVAE.py:43,49).
        gx / klw override d objective / d x_hat and the KL weight (default sign(x_hat - x) / N and 1 / N): restoration objectives."""
        masks = masks or {}
        n = x.shape[0]
        dt = x.dtype.type
        g = {}
        # d loss / d x_hat = sign(x_hat - x) / N   (tf.abs gradient: sign, 0 at 0)
        if gx is None:
            gx = (act['l1_sign'].astype(x.dtype) if act is not None and 'l1_sign' in act else np.sign(out['x_hat'] - x)) * dt(1.0 / n)
        a = cache['dec_out']
        da, g['Decoder/dec_Conv2D_final/kernel'], g['Decoder/dec_Conv2D_final/bias'] = \
            nn.conv2d_bwd(a, p['Decoder/dec_Conv2D_final/kernel'], gx, 1)
        for i in reversed(range(self.n_pool)):
            pre = f'Decoder/dec_Conv2DT_{i}'
            bnp = f'Decoder/batch_normalization_{i + 1}'
            dbn = act_bwd(cache, f'dec_bn{i}', da, LRELU_ALPHA, act)
            dc, g[bnp + '/gamma'], g[bnp + '/beta'] = nn.bn_frozen_bwd(cache[f'dec_c{i}'], p[bnp + '/gamma'], dbn)
            da, g[pre + '/kernel'], g[pre + '/bias'] = \
                nn.conv2d_transpose_bwd(cache[f'dec_in{i}'], p[pre + '/kernel'], dc, 2)
        dbn = act_bwd(cache, 'dec_bn_in', da, 0.0, act)
        dc, g['Decoder/batch_normalization/gamma'], g['Decoder/batch_normalization/beta'] = \
            nn.bn_frozen_bwd(cache['dec_c_in'], p['Decoder/batch_normalization/gamma'], dbn)
        dd4, g['Bottleneck/conv2d_1/kernel'], g['Bottleneck/conv2d_1/bias'] = \
            nn.conv2d_bwd(cache['d4'], p['Bottleneck/conv2d_1/kernel'], dc, 1)
        dd = dd4.reshape(n, -1)
        if self.variational and 'dec' in masks:
            dd = dd * masks['dec']
        dz, g[self.nm['dec'] + '/kernel'], g[self.nm['dec'] + '/bias'] = \
            nn.dense_bwd(cache['z'], p[self.nm['dec'] + '/kernel'], dd)
        if self.variational:
            mu, ls, sg, eps = cache['mu'], cache['ls'], cache['sigma'], cache['eps']
            # z = mu + eps*exp(ls); kl_n = 0.5*sum(mu^2 + exp(2 ls) - 2 ls - 1); loss += mean_n kl_n
            klw = (dt(1.0 / n) if klw is None else dt(klw)) if kl else dt(0.0)
            dmu = dz + mu * klw
            dls = dz * eps * sg + (sg * sg - dt(1.0)) * klw
            if 'mu' in masks:
                dmu = dmu * masks['mu']
            if 'sigma' in masks:
                dls = dls * masks['sigma']
            df1, g[self.nm['mu'] + '/kernel'], g[self.nm['mu'] + '/bias'] = \
                nn.dense_bwd(cache['flat'], p[self.nm['mu'] + '/kernel'], dmu)
            df2, g[self.nm['sigma'] + '/kernel'], g[self.nm['sigma'] + '/bias'] = \
                nn.dense_bwd(cache['flat'], p[self.nm['sigma'] + '/kernel'], dls)
            dflat = df1 + df2
        else:
            if 'z' in masks:
                dz = dz * masks['z']
            dflat, g[self.nm['z'] + '/kernel'], g[self.nm['z'] + '/bias'] = \
                nn.dense_bwd(cache['flat'], p[self.nm['z'] + '/kernel'], dz)
        dt4 = dflat.reshape(cache['t_shape'])
        da, g['Bottleneck/conv2d/kernel'], g['Bottleneck/conv2d/bias'] = \
            nn.conv2d_bwd(cache['enc_out'], p['Bottleneck/conv2d/kernel'], dt4, 1)
        for i in reversed(range(self.n_pool)):
            pre = f'Encoder/enc_conv2D_{i}'
            bnp = f'Encoder/batch_normalization_{i}'
            dbn = act_bwd(cache, f'enc_bn{i}', da, LRELU_ALPHA, act)
            dc, g[bnp + '/gamma'], g[bnp + '/beta'] = nn.bn_frozen_bwd(cache[f'enc_c{i}'], p[bnp + '/gamma'], dbn)
            da, g[pre + '/kernel'], g[pre + '/bias'] = \
                nn.conv2d_bwd(cache[f'enc_in{i}'], p[pre + '/kernel'], dc, 2)
        g['__dx'] = da
        return g

    # ------------------------------------------------------------------
    def train_step(self, p, opt, x, eps=None, masks=None, lr=1e-4, beta1=0.5):
        """One sess.run of trainers/VAE.py:83-96 (fetches + optimizer).
        opt = {'t': int, 'm': {name: arr}, 'v': {name: arr}}; updated in place."""
        out, cache = self.forward(p, x, eps, masks)
        ls = self.losses(x, out)
        g = self.backward(p, x, out, cache, masks)
        opt['t'] += 1
        for name, _, _ in self.spec:
            nn.adam_tf_step(p[name], g[name], opt['m'][name], opt['v'][name], opt['t'], lr, beta1)
        return out, ls, g

    def new_opt(self, p):
        return {'t': 0, 'm': {k: np.zeros_like(v) for k, v in p.items()},
                'v': {k: np.zeros_like(v) for k, v in p.items()}}

    def restore_grads(self, p, x, eps, tv_lambda):
        """trainers/VAE_You.py:52-53: tf.gradients(pixel_loss + restore, x) with pixel_loss = rec_n + kl_n PER SAMPLE (no mean) and
        restore = tv_lambda * total_variation(x - x_hat): x enters through the encoder and, directly, through r = x - x_hat."""
        from .gmvae import total_variation_grad
        out, cache = self.forward(p, x, eps)
        dxhat = np.sign(out['x_hat'] - x) - x.dtype.type(tv_lambda) * total_variation_grad(x - out['x_hat'])
        return self.backward(p, x, out, cache, gx=dxhat, klw=1.0)['__dx'] - dxhat

    def restore(self, p, x, noise, restore_steps=150, restore_lr=1e-3, tv_lambda=1.8):
        """trainers/VAE_You.py:133-144; noise: callable step -> eps (the graph samples z on every sess.run)."""
        rec = x.copy()
        for step in range(restore_steps):
            rec = rec - x.dtype.type(restore_lr) * self.restore_grads(p, rec, noise(step), tv_lambda)
        return rec

    def reconstruct(self, p, x, eps=None, masks=None):
        """trainers/VAE.py:105-123 / AE.py:92-110 (l2err == l1err, sic; A4)."""
        if x.ndim < 4:
            x = x[None]
        out, _ = self.forward(p, x, eps, masks)
        rec = out['x_hat']
        return {'reconstruction': rec, 'l1err': np.sum(np.abs(x - rec)),
                'l2err': np.sum(np.sqrt((x - rec) ** 2))}


class CeVAE(Model):
    """models/context_encoder_variational_autoencoder.py:9-59 + trainers/ceVAE.py:38-51: the same layer objects
    run twice -- x through the full VAE path, the masked x_ce through mu only (z_ce = z_mu_ce, no sampling, :37,43).
    masks: 'mu','sigma','dec' (VAE branch, :36,38,41) and 'mu_ce','dec_ce' (context branch, :37,43)."""

    def __init__(self, height=128, width=128, channels=1, inter_res=8, zdim=128):
        super().__init__('ceVAE', height, width, channels, inter_res, zdim)

    @staticmethod
    def _split(masks):
        masks = masks or {}
        mv = {k: masks[k] for k in ('mu', 'sigma', 'dec') if k in masks}
        mc = {k[:-3]: masks[k] for k in ('mu_ce', 'dec_ce') if k in masks}
        return mv, mc

    def ce_forward(self, p, x, x_ce, eps=None, masks=None):
        mv, mc = self._split(masks)
        out_v, cache_v = self.forward(p, x, eps, mv)
        out_c, cache_c = self.forward(p, x_ce, None, mc)      # eps = 0: z = z_mu_ce
        out = dict(out_v)
        out['x_hat_ce'], out['z_mu_ce'] = out_c['x_hat'], out_c['z_mu']
        return out, (out_v, cache_v, out_c, cache_c)

    def ce_losses(self, x, x_ce, out):
        """trainers/ceVAE.py:38-50"""
        n = x.shape[0]
        l1v = np.abs(out['x_hat'] - x)
        l1c = np.abs(out['x_hat_ce'] - x_ce)
        rv = l1v.reshape(n, -1).sum(axis=1)
        rc = l1c.reshape(n, -1).sum(axis=1)
        mu, ls, sg = out['z_mu'], out['z_log_sigma'], out['z_sigma']
        kl = 0.5 * (mu * mu + sg * sg - 2.0 * ls - 1.0).sum(axis=1)
        return {'L1_vae': l1v, 'L1_ce': l1c, 'L1': 0.5 * (l1v + l1c), 'Rec_ce': rc.mean(), 'Rec_vae': rv.mean(),
                'reconstructionLoss': 0.5 * (rv + rc).mean(), 'kl': kl.mean(), 'loss': (rv + kl + rc).mean(),
                'loss_vae': (rv + kl).mean()}

    def ce_backward(self, p, x, x_ce, out, caches, masks=None, act_v=None, act_c=None):
        """d loss / d params (sum of both branches through the shared variables) and
        anomaly = L1_vae * |d loss_vae / d x| (trainers/ceVAE.py:51; x enters loss_vae through the encoder AND
        directly as the L1 label, whose tf.abs gradient is sign(x - x_hat)/N)."""
        mv, mc = self._split(masks)
        out_v, cache_v, out_c, cache_c = caches
        gv = self.backward(p, x, out_v, cache_v, mv, kl=True, act=act_v)
        gc = self.backward(p, x_ce, out_c, cache_c, mc, kl=False, act=act_c)
        g = {name: gv[name] + gc[name] for name, _, _ in self.spec}
        n = x.shape[0]
        sg = -act_v['l1_sign'].astype(x.dtype) if act_v is not None and 'l1_sign' in act_v else np.sign(x - out['x_hat'])
        dx = gv['__dx'] + sg * x.dtype.type(1.0 / n)
        g['__dx_vae'] = dx
        g['anomaly'] = np.abs(out['x_hat'] - x) * np.abs(dx)
        return g

    def ce_train_step(self, p, opt, x, x_ce, eps=None, masks=None, lr=1e-4, beta1=0.5):
        out, caches = self.ce_forward(p, x, x_ce, eps, masks)
        ls = self.ce_losses(x, x_ce, out)
        g = self.ce_backward(p, x, x_ce, out, caches, masks)
        ls['anomaly'] = g['anomaly']
        opt['t'] += 1
        for name, _, _ in self.spec:
            nn.adam_tf_step(p[name], g[name], opt['m'][name], opt['v'][name], opt['t'], lr, beta1)
        return out, ls, g

    def ce_reconstruct(self, p, x, eps=None, use_gradient_based_restoration=True):
        """trainers/ceVAE.py:119-144: x_ce = x, dropout off; 'reconstruction' = x - c*anomaly when c is truthy."""
        if x.ndim < 4:
            x = x[None]
        out, caches = self.ce_forward(p, x, x, eps, None)
        res = self.ce_losses(x, x, out)
        res['anomaly'] = self.ce_backward(p, x, x, out, caches)['anomaly']
        rec = out['x_hat']
        if use_gradient_based_restoration:
            rec = x - x.dtype.type(use_gradient_based_restoration) * res['anomaly']
        res['reconstruction'] = rec
        res['l1err'] = np.sum(np.abs(x - rec))
        res['l2err'] = np.sum(np.sqrt((x - rec) ** 2))
        return res


def retrieve_masked_batch(batch, brainmasks, rng):
    """trainers/CE.py:123-139, INCLUDING its defect: the loop variable shadows the mask array, so
    `m` after the loop is the LAST sample's [H,W,C] mask and it is broadcast over the whole batch.  `rng` must offer
    randint(a, b) with both ends inclusive (the reference uses the `random` module)."""
    ranges = []
    for bm in brainmasks:
        px = np.argwhere(bm).T
        ranges.append(((px[0].min(), px[0].max()), (px[1].min(), px[1].max())))
    masks = np.ones(batch.shape)
    m = masks
    for m, br in zip(masks, ranges):
        for _ in range(rng.randint(1, 3)):
            sw, sh = 20, 20
            if br[0][0] < br[0][1] - sw and br[1][0] < br[1][1] - sh:
                xx = rng.randint(br[0][0], br[0][1] - sw)
                yy = rng.randint(br[1][0], br[1][1] - sh)
                m[xx:xx + sw, yy:yy + sh] = 0
    return batch * m


# ----------------------------------------------------------------------
# synthetic Brainweb-like batch
# ----------------------------------------------------------------------
def synthetic_slices(n, h=128, w=128, seed=0, dtype=np.float32):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    cy, cx = (h - 1) / 2.0, (w - 1) / 2.0
    out = np.zeros((n, h, w, 1), dtype=dtype)
    for i in range(n):
        ry = h * rng.uniform(0.36, 0.42)
        rx = w * rng.uniform(0.30, 0.36)
        mask = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
        ph = rng.uniform(0, 2 * np.pi, 4)
        f = (0.55 + 0.2 * np.sin(yy / h * 2 * np.pi * 1.5 + ph[0]) * np.cos(xx / w * 2 * np.pi * 1.2 + ph[1])
             + 0.12 * np.sin(xx / w * 2 * np.pi * 3 + ph[2]) * np.sin(yy / h * 2 * np.pi * 2.5 + ph[3]))
        img = np.clip(f + rng.normal(0, 0.03, f.shape), 0.0, 1.0)
        out[i, :, :, 0] = (img * mask).astype(dtype)
    return out


# ---------------------------------------------------------------------------------------------------------------
# Spatial autoencoder (models/autoencoder_spatial.py:7-27): the unified encoder's feature map is the latent code -- dropout on
# it, then straight into the unified decoder (whose first layers are BN + ReLU, customlayers.py:30-31).  Trained by trainers/AE.py.
# ---------------------------------------------------------------------------------------------------------------
def param_spec_spatial(height, width, channels, inter_res):
    assert height == width
    n_pool = int(math.log(height, 2) - math.log(float(inter_res), 2))
    spec, cin = [], channels
    for i in range(n_pool):
        f = int(min(128, 32 * (2 ** i)))
        spec += [(f'Encoder/enc_conv2D_{i}/kernel', (5, 5, cin, f), 'conv_w'), (f'Encoder/enc_conv2D_{i}/bias', (f,), 'bias'),
                 (f'Encoder/batch_normalization_{i}/gamma', (f,), 'gamma'), (f'Encoder/batch_normalization_{i}/beta', (f,), 'beta')]
        cin = f
    spec += [('Decoder/batch_normalization/gamma', (cin,), 'gamma'), ('Decoder/batch_normalization/beta', (cin,), 'beta')]
    for i in range(n_pool):
        f = int(max(32, 128 / (2 ** i)))
        spec += [(f'Decoder/dec_Conv2DT_{i}/kernel', (5, 5, f, cin), 'conv_w'), (f'Decoder/dec_Conv2DT_{i}/bias', (f,), 'bias'),
                 (f'Decoder/batch_normalization_{i + 1}/gamma', (f,), 'gamma'), (f'Decoder/batch_normalization_{i + 1}/beta', (f,), 'beta')]
        cin = f
    spec += [('Decoder/dec_Conv2D_final/kernel', (1, 1, cin, channels), 'conv_w'), ('Decoder/dec_Conv2D_final/bias', (channels,), 'bias')]
    return spec


class SpatialAE:
    def __init__(self, height=128, width=128, channels=1, inter_res=8):
        self.h, self.w, self.c, self.inter = height, width, channels, inter_res
        self.spec = param_spec_spatial(height, width, channels, inter_res)
        self.n_pool = int(math.log(height, 2) - math.log(float(inter_res), 2))

    def forward(self, p, x, masks=None):
        """masks: {'z': pre-scaled keep mask [N, r, r, C]} (autoencoder_spatial.py:16)."""
        masks = masks or {}
        cache = {'x': x}
        a = x
        for i in range(self.n_pool):
            c = nn.conv2d_fwd(a, p[f'Encoder/enc_conv2D_{i}/kernel'], p[f'Encoder/enc_conv2D_{i}/bias'], 2)
            bn = nn.bn_frozen_fwd(c, p[f'Encoder/batch_normalization_{i}/gamma'], p[f'Encoder/batch_normalization_{i}/beta'])
            cache[f'enc_in{i}'], cache[f'enc_c{i}'], cache[f'enc_bn{i}'] = a, c, bn
            a = nn.leaky_relu_fwd(bn, LRELU_ALPHA)
        z = a * masks['z'] if 'z' in masks else a
        bn = nn.bn_frozen_fwd(z, p['Decoder/batch_normalization/gamma'], p['Decoder/batch_normalization/beta'])
        cache['z'], cache['dec_bn_in'] = z, bn
        a = nn.leaky_relu_fwd(bn, 0.0)
        for i in range(self.n_pool):
            c = nn.conv2d_transpose_fwd(a, p[f'Decoder/dec_Conv2DT_{i}/kernel'], p[f'Decoder/dec_Conv2DT_{i}/bias'], 2)
            bn = nn.bn_frozen_fwd(c, p[f'Decoder/batch_normalization_{i + 1}/gamma'], p[f'Decoder/batch_normalization_{i + 1}/beta'])
            cache[f'dec_in{i}'], cache[f'dec_c{i}'], cache[f'dec_bn{i}'] = a, c, bn
            a = nn.leaky_relu_fwd(bn, LRELU_ALPHA)
        cache['dec_out'] = a
        xh = nn.conv2d_fwd(a, p['Decoder/dec_Conv2D_final/kernel'], p['Decoder/dec_Conv2D_final/bias'], 1)
        return {'z': z, 'x_hat': xh}, cache

    def losses(self, x, out):
        l1 = np.abs(out['x_hat'] - x)
        rec = l1.reshape(x.shape[0], -1).sum(axis=1).mean()
        return {'L1': l1, 'reconstructionLoss': rec, 'loss': rec}

    def backward(self, p, x, out, cache, masks=None, act=None):
        masks = masks or {}
        n, g = x.shape[0], {}
        gx = (act['l1_sign'].astype(x.dtype) if act is not None and 'l1_sign' in act else np.sign(out['x_hat'] - x)) * x.dtype.type(1.0 / n)
        da, g['Decoder/dec_Conv2D_final/kernel'], g['Decoder/dec_Conv2D_final/bias'] = \
            nn.conv2d_bwd(cache['dec_out'], p['Decoder/dec_Conv2D_final/kernel'], gx, 1)
        for i in reversed(range(self.n_pool)):
            bnp = f'Decoder/batch_normalization_{i + 1}'
            dbn = act_bwd(cache, f'dec_bn{i}', da, LRELU_ALPHA, act)
            dc, g[bnp + '/gamma'], g[bnp + '/beta'] = nn.bn_frozen_bwd(cache[f'dec_c{i}'], p[bnp + '/gamma'], dbn)
            da, g[f'Decoder/dec_Conv2DT_{i}/kernel'], g[f'Decoder/dec_Conv2DT_{i}/bias'] = \
                nn.conv2d_transpose_bwd(cache[f'dec_in{i}'], p[f'Decoder/dec_Conv2DT_{i}/kernel'], dc, 2)
        dbn = act_bwd(cache, 'dec_bn_in', da, 0.0, act)
        dz, g['Decoder/batch_normalization/gamma'], g['Decoder/batch_normalization/beta'] = \
            nn.bn_frozen_bwd(cache['z'], p['Decoder/batch_normalization/gamma'], dbn)
        da = dz * masks['z'] if 'z' in masks else dz
        for i in reversed(range(self.n_pool)):
            bnp = f'Encoder/batch_normalization_{i}'
            dbn = act_bwd(cache, f'enc_bn{i}', da, LRELU_ALPHA, act)
            dc, g[bnp + '/gamma'], g[bnp + '/beta'] = nn.bn_frozen_bwd(cache[f'enc_c{i}'], p[bnp + '/gamma'], dbn)
            da, g[f'Encoder/enc_conv2D_{i}/kernel'], g[f'Encoder/enc_conv2D_{i}/bias'] = \
                nn.conv2d_bwd(cache[f'enc_in{i}'], p[f'Encoder/enc_conv2D_{i}/kernel'], dc, 2)
        return g
