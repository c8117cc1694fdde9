"""Oracle for the dense-bottleneck autoencoders with a latent constraint and / or a latent critic:
  'constrained_ae'   models/constrained_autoencoder.py:9-48            + trainers/ConstrainedAE.py:36-45
  'aae'              models/adversarial_autoencoder.py:10-72           + trainers/AAE.py:40-67
  'constrained_aae'  models/constrained_adversarial_autoencoder.py:10-79 + trainers/ConstrainedAAE.py:44-70
numpy forward, hand-written backward (incl. the second-order term of the latent WGAN-GP penalty through the MLP critic).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED (no TensorFlow here, no golden vectors in the reference);
tests/test_oracle_aae.py anchors every gradient on torch autograd in float64.

Shared graph: unified encoder (conv k5 s2 + frozen-stats BN + LeakyReLU) -> 1x1 conv C/8 -> Dense(zDim) -> dropout = z_
             -> Dense(flat) -> dropout -> 1x1 conv C -> unified decoder (BN + ReLU, ConvT k5 s2 + BN + LeakyReLU, 1x1 conv) = x_hat
constrained: z_rec = dropout(Dense_z(conv1x1(encoder(x_hat))))  (the SAME layers)        loss = mean_n(L2_n + rho * mean_k (z - z_rec)^2)
critic:      MLP zDim -> h1 -> h2 -> 1 with tf.nn.leaky_relu (alpha 0.2) on z_ (fake), the prior sample z (real) and
             z_hat = z + eps * (z - z_) (sic);  disc_loss = mean d_ - mean d + mean((||d d_hat / d z_hat||_2 - 1)^2 * scale);  gen_loss = -mean d_
Variable scopes differ per model (they decide which variables `optim_gen` touches, `'Encoder' in var.name`), see param_spec()."""
import numpy as np

from . import nn

LRELU = 0.3          # keras LeakyReLU() of the conv blocks
CRITIC_ALPHA = 0.2   # tf.nn.leaky_relu default


def param_spec(kind, height=128, inter_res=8, zdim=128, channels=1):
    """[(name, shape, kind)] in TF variable-creation (= first-call) order + the name map of the bottleneck layers."""
    assert kind in ('constrained_ae', 'aae', 'constrained_aae')
    npool = int(round(np.log2(height) - np.log2(inter_res)))
    spec, cin = [], channels
    for i in range(npool):
        f = min(128, 32 * 2 ** i)
        spec += [('Encoder/enc_conv2D_%d/kernel' % i, (5, 5, cin, f), 'conv_w'), ('Encoder/enc_conv2D_%d/bias' % i, (f,), 'bias'),
                 ('Encoder/batch_normalization' + ('' if i == 0 else '_%d' % i) + '/gamma', (f,), 'gamma'),
                 ('Encoder/batch_normalization' + ('' if i == 0 else '_%d' % i) + '/beta', (f,), 'beta')]
        cin = f
    cenc, cmid = cin, cin // 8
    flat = inter_res * inter_res * cmid
    if kind == 'constrained_aae':
        # intermediate_conv and z_layer are first called inside 'Encoder', dec_dense and the reverse conv inside 'Decoder' (:21-36)
        nm = {'conv': 'Encoder/conv2d', 'z': 'Encoder/dense', 'dec': 'Decoder/dense', 'rev': 'Decoder/conv2d_1'}
        spec += [(nm['conv'] + '/kernel', (1, 1, cenc, cmid), 'conv_w'), (nm['conv'] + '/bias', (cmid,), 'bias'),
                 (nm['z'] + '/kernel', (flat, zdim), 'dense_w'), (nm['z'] + '/bias', (zdim,), 'bias'),
                 (nm['dec'] + '/kernel', (zdim, flat), 'dense_w'), (nm['dec'] + '/bias', (flat,), 'bias'),
                 (nm['rev'] + '/kernel', (1, 1, cmid, cenc), 'conv_w'), (nm['rev'] + '/bias', (cenc,), 'bias')]
    else:
        nm = {'conv': 'Bottleneck/conv2d', 'z': 'Bottleneck/dense', 'dec': 'Bottleneck/dense_1', 'rev': 'Bottleneck/conv2d_1'}
        spec += [(nm['conv'] + '/kernel', (1, 1, cenc, cmid), 'conv_w'), (nm['conv'] + '/bias', (cmid,), 'bias'),
                 (nm['z'] + '/kernel', (flat, zdim), 'dense_w'), (nm['z'] + '/bias', (zdim,), 'bias'),
                 (nm['dec'] + '/kernel', (zdim, flat), 'dense_w'), (nm['dec'] + '/bias', (flat,), 'bias'),
                 (nm['rev'] + '/kernel', (1, 1, cmid, cenc), 'conv_w'), (nm['rev'] + '/bias', (cenc,), 'bias')]
    spec += [('Decoder/batch_normalization/gamma', (cenc,), 'gamma'), ('Decoder/batch_normalization/beta', (cenc,), 'beta')]
    cin = cenc
    for i in range(npool):
        f = max(32, 128 // 2 ** i)
        spec += [('Decoder/dec_Conv2DT_%d/kernel' % i, (5, 5, f, cin), 'conv_w'), ('Decoder/dec_Conv2DT_%d/bias' % i, (f,), 'bias'),
                 ('Decoder/batch_normalization_%d/gamma' % (i + 1), (f,), 'gamma'), ('Decoder/batch_normalization_%d/beta' % (i + 1), (f,), 'beta')]
        cin = f
    spec += [('Decoder/dec_Conv2D_final/kernel', (1, 1, cin, channels), 'conv_w'), ('Decoder/dec_Conv2D_final/bias', (channels,), 'bias')]
    if kind != 'constrained_ae':
        h1, h2 = (50, 50) if kind == 'aae' else (100, 50)
        spec += [('Discriminator/dense/kernel', (zdim, h1), 'dense_w'), ('Discriminator/dense/bias', (h1,), 'bias'),
                 ('Discriminator/dense_1/kernel', (h1, h2), 'dense_w'), ('Discriminator/dense_1/bias', (h2,), 'bias'),
                 ('Discriminator/dense_2/kernel', (h2, 1), 'dense_w'), ('Discriminator/dense_2/bias', (1,), 'bias')]
    return spec, nm


def _lrelu(a, alpha):
    return np.where(a > 0, a, alpha * a)


class AAE:
    def __init__(self, kind, height=128, inter_res=8, zdim=128, rho=1.0, scale=10.0):
        self.kind, self.height, self.inter_res, self.zdim, self.rho, self.scale = kind, height, inter_res, zdim, rho, scale
        self.constrained = kind in ('constrained_ae', 'constrained_aae')
        self.has_critic = kind in ('aae', 'constrained_aae')
        self.npool = int(round(np.log2(height) - np.log2(inter_res)))
        self.spec, self.nm = param_spec(kind, height, inter_res, zdim)
        self.bn_e = ['Encoder/batch_normalization' + ('' if i == 0 else '_%d' % i) for i in range(self.npool)]

    # ------------------------------------------------------------------ autoencoder pieces
    def encode(self, p, x, mask_z=None):
        """x -> z_ (post dropout); cache for encode_backward."""
        cache = {'a': [x], 'c': []}
        a = x
        for i in range(self.npool):
            c = nn.conv2d_fwd(a, p['Encoder/enc_conv2D_%d/kernel' % i], p['Encoder/enc_conv2D_%d/bias' % i], 2)
            a = nn.leaky_relu_fwd(nn.bn_frozen_fwd(c, p[self.bn_e[i] + '/gamma'], p[self.bn_e[i] + '/beta']), LRELU)
            cache['c'].append(c); cache['a'].append(a)
        t = nn.conv2d_fwd(a, p[self.nm['conv'] + '/kernel'], p[self.nm['conv'] + '/bias'], 1)
        flat = t.reshape(t.shape[0], -1)
        z = nn.dense_fwd(flat, p[self.nm['z'] + '/kernel'], p[self.nm['z'] + '/bias'])
        if mask_z is not None:
            z = z * mask_z
        cache.update(t=t, flat=flat, mask=mask_z)
        return z, cache

    def encode_backward(self, p, cache, dz, g):
        """accumulates the encoder-path parameter gradients into g; returns d / d x."""
        def acc(k, v):
            g[k] = g.get(k, 0) + v
        if cache['mask'] is not None:
            dz = dz * cache['mask']
        dflat, dw, db = nn.dense_bwd(cache['flat'], p[self.nm['z'] + '/kernel'], dz)
        acc(self.nm['z'] + '/kernel', dw); acc(self.nm['z'] + '/bias', db)
        da, dw, db = nn.conv2d_bwd(cache['a'][-1], p[self.nm['conv'] + '/kernel'], dflat.reshape(cache['t'].shape), 1)
        acc(self.nm['conv'] + '/kernel', dw); acc(self.nm['conv'] + '/bias', db)
        for i in reversed(range(self.npool)):
            c = cache['c'][i]
            bnv = nn.bn_frozen_fwd(c, p[self.bn_e[i] + '/gamma'], p[self.bn_e[i] + '/beta'])
            dc, dg, dbt = nn.bn_frozen_bwd(c, p[self.bn_e[i] + '/gamma'], nn.leaky_relu_bwd(bnv, da, LRELU))
            acc(self.bn_e[i] + '/gamma', dg); acc(self.bn_e[i] + '/beta', dbt)
            da, dw, db = nn.conv2d_bwd(cache['a'][i], p['Encoder/enc_conv2D_%d/kernel' % i], dc, 2)
            acc('Encoder/enc_conv2D_%d/kernel' % i, dw); acc('Encoder/enc_conv2D_%d/bias' % i, db)
        return da

    def decode(self, p, z, mask_dec=None):
        r = self.inter_res
        dv = nn.dense_fwd(z, p[self.nm['dec'] + '/kernel'], p[self.nm['dec'] + '/bias'])
        if mask_dec is not None:
            dv = dv * mask_dec
        dmap = dv.reshape(z.shape[0], r, r, -1)
        c = nn.conv2d_fwd(dmap, p[self.nm['rev'] + '/kernel'], p[self.nm['rev'] + '/bias'], 1)
        bn = nn.bn_frozen_fwd(c, p['Decoder/batch_normalization/gamma'], p['Decoder/batch_normalization/beta'])
        cache = {'z': z, 'dmap': dmap, 'mask': mask_dec, 'c_in': c, 'bn_in': bn, 'a': [np.maximum(bn, 0)], 'c': [], 'bn': []}
        a = cache['a'][0]
        for i in range(self.npool):
            c = nn.conv2d_transpose_fwd(a, p['Decoder/dec_Conv2DT_%d/kernel' % i], p['Decoder/dec_Conv2DT_%d/bias' % i], 2)
            bn = nn.bn_frozen_fwd(c, p['Decoder/batch_normalization_%d/gamma' % (i + 1)], p['Decoder/batch_normalization_%d/beta' % (i + 1)])
            a = nn.leaky_relu_fwd(bn, LRELU)
            cache['c'].append(c); cache['bn'].append(bn); cache['a'].append(a)
        xh = nn.conv2d_fwd(a, p['Decoder/dec_Conv2D_final/kernel'], p['Decoder/dec_Conv2D_final/bias'], 1)
        return xh, cache

    def decode_backward(self, p, cache, dxh, g):
        da, g['Decoder/dec_Conv2D_final/kernel'], g['Decoder/dec_Conv2D_final/bias'] = \
            nn.conv2d_bwd(cache['a'][-1], p['Decoder/dec_Conv2D_final/kernel'], dxh, 1)
        for i in reversed(range(self.npool)):
            bnp = 'Decoder/batch_normalization_%d' % (i + 1)
            dc, g[bnp + '/gamma'], g[bnp + '/beta'] = nn.bn_frozen_bwd(cache['c'][i], p[bnp + '/gamma'], nn.leaky_relu_bwd(cache['bn'][i], da, LRELU))
            da, g['Decoder/dec_Conv2DT_%d/kernel' % i], g['Decoder/dec_Conv2DT_%d/bias' % i] = \
                nn.conv2d_transpose_bwd(cache['a'][i], p['Decoder/dec_Conv2DT_%d/kernel' % i], dc, 2)
        dc, g['Decoder/batch_normalization/gamma'], g['Decoder/batch_normalization/beta'] = \
            nn.bn_frozen_bwd(cache['c_in'], p['Decoder/batch_normalization/gamma'], nn.leaky_relu_bwd(cache['bn_in'], da, 0.0))
        dmap, g[self.nm['rev'] + '/kernel'], g[self.nm['rev'] + '/bias'] = nn.conv2d_bwd(cache['dmap'], p[self.nm['rev'] + '/kernel'], dc, 1)
        dv = dmap.reshape(dmap.shape[0], -1)
        if cache['mask'] is not None:
            dv = dv * cache['mask']
        dz, g[self.nm['dec'] + '/kernel'], g[self.nm['dec'] + '/bias'] = nn.dense_bwd(cache['z'], p[self.nm['dec'] + '/kernel'], dv)
        return dz

    # ------------------------------------------------------------------ latent critic (MLP, leaky_relu 0.2)
    def critic(self, p, v):
        w1, b1 = p['Discriminator/dense/kernel'], p['Discriminator/dense/bias']
        w2, b2 = p['Discriminator/dense_1/kernel'], p['Discriminator/dense_1/bias']
        w3, b3 = p['Discriminator/dense_2/kernel'], p['Discriminator/dense_2/bias']
        a1 = v @ w1 + b1; h1 = _lrelu(a1, CRITIC_ALPHA)
        a2 = h1 @ w2 + b2; h2 = _lrelu(a2, CRITIC_ALPHA)
        return h2 @ w3 + b3, dict(v=v, a1=a1, h1=h1, a2=a2, h2=h2)

    def critic_backward(self, p, c, dd, g=None):
        """dd [n,1] = dL/dd.  Returns dL/dv; accumulates parameter gradients into g (optional)."""
        w1, w2, w3 = p['Discriminator/dense/kernel'], p['Discriminator/dense_1/kernel'], p['Discriminator/dense_2/kernel']
        da2 = (dd @ w3.T) * np.where(c['a2'] > 0, 1.0, CRITIC_ALPHA)
        da1 = (da2 @ w2.T) * np.where(c['a1'] > 0, 1.0, CRITIC_ALPHA)
        if g is not None:
            for k, val in (('Discriminator/dense_2/kernel', c['h2'].T @ dd), ('Discriminator/dense_2/bias', dd.sum(axis=0)),
                           ('Discriminator/dense_1/kernel', c['h1'].T @ da2), ('Discriminator/dense_1/bias', da2.sum(axis=0)),
                           ('Discriminator/dense/kernel', c['v'].T @ da1), ('Discriminator/dense/bias', da1.sum(axis=0))):
                g[k] = g.get(k, 0) + val
        return da1 @ w1.T

    def critic_penalty(self, p, c, g):
        """scale * mean_n (||d d_hat / d z_hat||_2 - 1)^2 and its gradient w.r.t. the critic's kernels (second order, masks constant a.e.)."""
        w1, w2, w3 = p['Discriminator/dense/kernel'], p['Discriminator/dense_1/kernel'], p['Discriminator/dense_2/kernel']
        m1 = np.where(c['a1'] > 0, 1.0, CRITIC_ALPHA); m2 = np.where(c['a2'] > 0, 1.0, CRITIC_ALPHA)
        n = m1.shape[0]
        u2 = m2 * w3[:, 0]                  # [n,h2]
        t1 = u2 @ w2.T                      # [n,h1]
        u1 = m1 * t1
        gz = u1 @ w1.T                      # [n,zdim] = d d_hat / d z_hat
        s = np.sqrt((gz ** 2).sum(axis=1))
        pen = self.scale * ((s - 1.0) ** 2).mean()
        gbar = (self.scale * 2.0 * (s - 1.0) / s / n)[:, None] * gz
        ub1 = gbar @ w1                     # adjoint of u1
        tb1 = ub1 * m1
        ub2 = tb1 @ w2                      # adjoint of u2
        for k, val in (('Discriminator/dense/kernel', gbar.T @ u1), ('Discriminator/dense_1/kernel', tb1.T @ u2),
                       ('Discriminator/dense_2/kernel', (ub2 * m2).sum(axis=0)[:, None])):
            g[k] = g.get(k, 0) + val
        return pen

    # ------------------------------------------------------------------ phases
    def ae_phase(self, p, x, mask_z=None, mask_dec=None, mask_rec=None):
        """optim_ae: loss = mean_n(L2_n [+ rho * Rec_z_n]); gradients of every autoencoder variable."""
        n = x.shape[0]
        z, ec = self.encode(p, x, mask_z)
        xh, dc = self.decode(p, z, mask_dec)
        l2 = ((x - xh) ** 2).reshape(n, -1).mean(axis=1)
        l1 = np.abs(x - xh)
        losses = {'L1': l1, 'reconstructionLoss': l1.reshape(n, -1).sum(axis=1).mean(), 'L2': l2, 'reconstruction': xh, 'z': z}
        g = {}
        dxh = 2.0 * (xh - x) / (n * xh[0].size)
        dz_direct = 0.0
        if self.constrained:
            z_rec, ec2 = self.encode(p, xh, mask_rec)
            rec_z = ((z - z_rec) ** 2).mean(axis=1)
            losses.update(Rec_z=rec_z, z_rec=z_rec, loss=(l2 + self.rho * rec_z).mean())
            dzr = -2.0 * self.rho * (z - z_rec) / (n * self.zdim)
            dxh = dxh + self.encode_backward(p, ec2, dzr, g)
            dz_direct = -dzr
        else:
            losses['loss'] = l2.mean()
        dz = self.decode_backward(p, dc, dxh, g) + dz_direct
        self.encode_backward(p, ec, dz, g)
        return losses, g

    def _z_hat(self, z_prior, z_, eps):
        return z_prior + eps * (z_prior - z_)            # (sic) adversarial_autoencoder.py:60, constrained_adversarial_autoencoder.py:67

    def disc_phase(self, p, x, z_prior, eps, mask_z=None):
        """optim_dis: disc_loss (+ penalty) w.r.t. the Discriminator variables."""
        z_, _ = self.encode(p, x, mask_z)
        d_fake, cf = self.critic(p, z_)
        d_real, cr = self.critic(p, z_prior)
        z_hat = self._z_hat(z_prior, z_, eps.reshape(-1, 1))
        _, ch = self.critic(p, z_hat)
        g = {}
        n = x.shape[0]
        self.critic_backward(p, cf, np.full_like(d_fake, 1.0 / n), g)
        self.critic_backward(p, cr, np.full_like(d_real, -1.0 / n), g)
        pen = self.critic_penalty(p, ch, g)
        losses = {'disc_fake': d_fake.mean(), 'disc_real': d_real.mean(), 'penalty': pen}
        losses['disc_loss'] = losses['disc_fake'] - losses['disc_real'] + pen
        return losses, g

    def gen_phase(self, p, x, mask_z=None):
        """optim_gen: gen_loss = -mean d_ w.r.t. the variables whose name contains 'Encoder'."""
        z_, ec = self.encode(p, x, mask_z)
        d_fake, cf = self.critic(p, z_)
        dz = self.critic_backward(p, cf, np.full_like(d_fake, -1.0 / x.shape[0]))
        g = {}
        self.encode_backward(p, ec, dz, g)
        return {'gen_loss': -d_fake.mean()}, {k: v for k, v in g.items() if 'Encoder' in k}

    def reconstruct(self, p, x):
        return self.decode(p, self.encode(p, x)[0])[0]
