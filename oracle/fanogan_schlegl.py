"""Oracle for the ResNet f-AnoGAN graph (models/fanogan_schlegl.py:11-161) under the same three optimisation phases
(trainers/fAnoGAN.py:45-77): numpy forward, hand-written first- and second-order backward of the pre-activation residual
blocks (LayerNorm-HW -> ReLU -> k3 conv -> LayerNorm-HW -> ReLU -> k3 conv / ConvT, + shortcut).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED (no TensorFlow here; no golden vectors in the reference);
tests/test_oracle_fanogan.py anchors every gradient on torch autograd in float64 (double backward included).

Graph restated (NHWC):
  Encoder        unified encoder blocks (conv k5 s2 + frozen-stats BN + LeakyReLU) -> Flatten -> Dense(zDim) -> tanh   (:15-22)
  Generator      Dense(r*r*8d) -> [r,r,8d] -> res1 (8d, k3 s1 ConvT, identity shortcut) -> res2..4 (4d, 2d, d: k3 conv, k3 s2
                 ConvT, k1 s2 ConvT shortcut) -> LN -> ReLU -> 1x1 conv -> tanh                            (:25-55, 119-137)
  Discriminator  k3 conv d -> res1..3 (2d, 4d, 8d: k3 conv, k3 s2 conv, shortcut = AvgPool(1x1 conv)) -> res4 (8d, identity
                 shortcut) -> Dense(1) per location                                                         (:64-97, 140-161)
  d = `dim` = 64 in the reference (:13); a parameter here so that the tests can run small.
"""
import numpy as np

from . import nn
from .fanogan import ln_fwd, ln_bwd, ln_bwd2, group_of  # noqa: F401

BN_LRELU = 0.3


def avgpool_fwd(x):
    n, h, w, c = x.shape
    return x.reshape(n, h // 2, 2, w // 2, 2, c).mean(axis=(2, 4))


def avgpool_bwd(g):
    return np.repeat(np.repeat(g, 2, axis=1), 2, axis=2) * g.dtype.type(0.25)


def relu_bwd(y, g):
    return nn.leaky_relu_bwd(y, g, 0.0)


class _Block:
    """One pre-activation residual block.  kind 'g' (generator: conv2 is a ConvT, shortcut a k1 s2 ConvT) or 'd' (critic: conv2 a
    conv, shortcut AvgPool(1x1 conv)); stride 1 => identity shortcut."""

    def __init__(self, kind, names, stride):
        self.kind, self.n, self.stride = kind, names, stride   # names: dict ln1, conv1, ln2, conv2, short (or None)

    def _conv2_fwd(self, p, h):
        w, b = p[self.n['conv2'] + '/kernel'], p[self.n['conv2'] + '/bias']
        return nn.conv2d_transpose_fwd(h, w, b, self.stride) if self.kind == 'g' else nn.conv2d_fwd(h, w, b, self.stride)

    def _conv2_bwd(self, p, h, g):
        w = p[self.n['conv2'] + '/kernel']
        return nn.conv2d_transpose_bwd(h, w, g, self.stride) if self.kind == 'g' else nn.conv2d_bwd(h, w, g, self.stride)

    def _conv2_lin(self, p, h):      # forward without bias
        w = p[self.n['conv2'] + '/kernel']
        return nn.conv2d_transpose_fwd(h, w, None, self.stride) if self.kind == 'g' else nn.conv2d_fwd(h, w, None, self.stride)

    def fwd(self, p, x):
        n = self.n
        y1, l1 = ln_fwd(x, p[n['ln1'] + '/gamma'], p[n['ln1'] + '/beta'])
        h1 = np.maximum(y1, 0)
        c1 = nn.conv2d_fwd(h1, p[n['conv1'] + '/kernel'], p[n['conv1'] + '/bias'], 1)
        y2, l2 = ln_fwd(c1, p[n['ln2'] + '/gamma'], p[n['ln2'] + '/beta'])
        h2 = np.maximum(y2, 0)
        c2 = self._conv2_fwd(p, h2)
        if n['short'] is None:
            out = c2 + x
        elif self.kind == 'g':
            out = c2 + nn.conv2d_transpose_fwd(x, p[n['short'] + '/kernel'], p[n['short'] + '/bias'], 2)
        else:
            out = c2 + avgpool_fwd(nn.conv2d_fwd(x, p[n['short'] + '/kernel'], p[n['short'] + '/bias'], 1))
        return out, dict(x=x, y1=y1, l1=l1, h1=h1, y2=y2, l2=l2, h2=h2)

    def bwd(self, p, c, dout, g=None, inj=None, tape=None):
        """dL/dx for dL/dout; g (dict) receives the parameter gradients; inj = (d/dx, d/dc1) second-order injections;
        tape (dict) receives what the penalty's adjoint pass needs."""
        n = self.n
        dh2, dw2, db2 = self._conv2_bwd(p, c['h2'], dout)
        v2 = relu_bwd(c['y2'], dh2)
        dc1, dg2, dbt2 = ln_bwd(v2, p[n['ln2'] + '/gamma'], c['l2'])
        dc1_lin = dc1
        if inj is not None:
            dc1 = dc1 + inj[1]
        dh1, dw1, db1 = nn.conv2d_bwd(c['h1'], p[n['conv1'] + '/kernel'], dc1, 1)
        v1 = relu_bwd(c['y1'], dh1)
        dx, dg1, dbt1 = ln_bwd(v1, p[n['ln1'] + '/gamma'], c['l1'])
        if inj is not None:
            dx = dx + inj[0]
        dsc = None
        if n['short'] is None:
            dx = dx + dout
        elif self.kind == 'g':
            dxs, dws, dbs = nn.conv2d_transpose_bwd(c['x'], p[n['short'] + '/kernel'], dout, 2)
            dx = dx + dxs
        else:
            dsc = avgpool_bwd(dout)
            dxs, dws, dbs = nn.conv2d_bwd(c['x'], p[n['short'] + '/kernel'], dsc, 1)
            dx = dx + dxs
        if g is not None:
            for k, v in ((n['conv2'] + '/kernel', dw2), (n['conv2'] + '/bias', db2), (n['ln2'] + '/gamma', dg2), (n['ln2'] + '/beta', dbt2),
                         (n['conv1'] + '/kernel', dw1), (n['conv1'] + '/bias', db1), (n['ln1'] + '/gamma', dg1), (n['ln1'] + '/beta', dbt1)):
                g[k] = g.get(k, 0) + v
            if n['short'] is not None:
                g[n['short'] + '/kernel'] = g.get(n['short'] + '/kernel', 0) + dws
                g[n['short'] + '/bias'] = g.get(n['short'] + '/bias', 0) + dbs
        if tape is not None:
            tape.update(dout=dout, v2=v2, dc1=dc1_lin, v1=v1, dsc=dsc)
        return dx

    def adj(self, p, c, tape, ubar_x, g):
        """Adjoint of the data-gradient map of this block (critic only): given ubar_x = dP/d(dx) returns
        (dP/d(dout), (inj_x, inj_c1)) and adds the direct parameter gradients to g."""
        n = self.n
        vb1, dg1, inj_x = ln_bwd2(ubar_x, tape['v1'], p[n['ln1'] + '/gamma'], c['l1'])
        ub_h1 = relu_bwd(c['y1'], vb1)
        w1 = p[n['conv1'] + '/kernel']
        q1 = nn.conv2d_fwd(ub_h1, w1, None, 1)
        _, dw1, _ = nn.conv2d_bwd(ub_h1, w1, tape['dc1'], 1)
        vb2, dg2, inj_c1 = ln_bwd2(q1, tape['v2'], p[n['ln2'] + '/gamma'], c['l2'])
        ub_h2 = relu_bwd(c['y2'], vb2)
        ub_out = self._conv2_lin(p, ub_h2)
        _, dw2, _ = self._conv2_bwd(p, ub_h2, tape['dout'])
        for k, v in ((n['ln1'] + '/gamma', dg1), (n['conv1'] + '/kernel', dw1), (n['ln2'] + '/gamma', dg2), (n['conv2'] + '/kernel', dw2)):
            g[k] = g.get(k, 0) + v
        if n['short'] is None:
            ub_out = ub_out + ubar_x
        else:
            ws = p[n['short'] + '/kernel']
            ub_out = ub_out + avgpool_fwd(nn.conv2d_fwd(ubar_x, ws, None, 1))
            _, dws, _ = nn.conv2d_bwd(ubar_x, ws, tape['dsc'], 1)
            g[n['short'] + '/kernel'] = g.get(n['short'] + '/kernel', 0) + dws
        return ub_out, (inj_x, inj_c1)


def param_spec(height=64, inter_res=8, zdim=128, dim=64, channels=1):
    """TF variable-creation (= first-call) order.  tf.layers names (conv2d, conv2d_transpose, dense, batch_normalization) are
    unique per variable scope in call order; keras LayerNormalization names count globally in construction order."""
    npool = int(round(np.log2(height) - np.log2(inter_res)))
    if npool != 3:
        raise ValueError('the ResNet generator upsamples 8x: height must be 8 * inter_res (fanogan_schlegl.py:28,121-133)')
    spec = []
    cin = channels
    for i in range(npool):
        f = min(128, 32 * 2 ** i)
        spec += [('Encoder/enc_conv2D_%d/kernel' % i, (5, 5, cin, f), 'conv_w'), ('Encoder/enc_conv2D_%d/bias' % i, (f,), 'bias')]
        bn = 'Encoder/batch_normalization' + ('' if i == 0 else '_%d' % i)
        spec += [(bn + '/gamma', (f,), 'gamma'), (bn + '/beta', (f,), 'beta')]
        cin = f
    spec += [('Encoder/dense/kernel', (inter_res * inter_res * cin, zdim), 'dense_w'), ('Encoder/dense/bias', (zdim,), 'bias')]
    ln = [0]

    def ln_pair(scope, res):
        name = scope + ('layer_normalization' if ln[0] == 0 else 'layer_normalization_%d' % ln[0])
        ln[0] += 1
        return name, [(name + '/gamma', (res, res), 'gamma'), (name + '/beta', (res, res), 'beta')]

    cnt = {}

    def tfname(scope, base):
        k = cnt.get((scope, base), 0)
        cnt[(scope, base)] = k + 1
        return scope + base + ('' if k == 0 else '_%d' % k)

    blocks_g, blocks_d = [], []
    spec += [('Generator/dense/kernel', (zdim, inter_res * inter_res * 8 * dim), 'dense_w'),
             ('Generator/dense/bias', (inter_res * inter_res * 8 * dim,), 'bias')]
    cin, res = 8 * dim, inter_res
    for f, stride in ((8 * dim, 1), (4 * dim, 2), (2 * dim, 2), (dim, 2)):
        n1, s1 = ln_pair('Generator/', res)
        c1 = tfname('Generator/', 'conv2d')
        n2, s2 = ln_pair('Generator/', res)
        c2 = tfname('Generator/', 'conv2d_transpose')
        spec += s1 + [(c1 + '/kernel', (3, 3, cin, f), 'conv_w'), (c1 + '/bias', (f,), 'bias')] + s2
        spec += [(c2 + '/kernel', (3, 3, f, f), 'conv_w'), (c2 + '/bias', (f,), 'bias')]
        sh = None
        if stride == 2:
            sh = tfname('Generator/', 'conv2d_transpose')
            spec += [(sh + '/kernel', (1, 1, f, cin), 'conv_w'), (sh + '/bias', (f,), 'bias')]
        blocks_g.append(_Block('g', dict(ln1=n1, conv1=c1, ln2=n2, conv2=c2, short=sh), stride))
        cin, res = f, res * stride
    gl, sg = ln_pair('Generator/', res)
    gf = tfname('Generator/', 'conv2d')
    spec += sg + [(gf + '/kernel', (1, 1, cin, channels), 'conv_w'), (gf + '/bias', (channels,), 'bias')]
    d0 = tfname('Discriminator/', 'conv2d')
    spec += [(d0 + '/kernel', (3, 3, channels, dim), 'conv_w'), (d0 + '/bias', (dim,), 'bias')]
    cin, res = dim, height
    for f, stride in ((2 * dim, 2), (4 * dim, 2), (8 * dim, 2), (8 * dim, 1)):
        n1, s1 = ln_pair('Discriminator/', res)
        c1 = tfname('Discriminator/', 'conv2d')
        n2, s2 = ln_pair('Discriminator/', res)
        c2 = tfname('Discriminator/', 'conv2d')
        spec += s1 + [(c1 + '/kernel', (3, 3, cin, f), 'conv_w'), (c1 + '/bias', (f,), 'bias')] + s2
        spec += [(c2 + '/kernel', (3, 3, f, f), 'conv_w'), (c2 + '/bias', (f,), 'bias')]
        sh = None
        if stride == 2:
            sh = tfname('Discriminator/', 'conv2d')
            spec += [(sh + '/kernel', (1, 1, cin, f), 'conv_w'), (sh + '/bias', (f,), 'bias')]
        blocks_d.append(_Block('d', dict(ln1=n1, conv1=c1, ln2=n2, conv2=c2, short=sh), stride))
        cin, res = f, res // stride
    spec += [('Discriminator/dense/kernel', (cin, 1), 'dense_w'), ('Discriminator/dense/bias', (1,), 'bias')]
    return spec, blocks_g, blocks_d, dict(gen_ln=gl, gen_final=gf, dis_conv=d0)


class FAnoGANSchlegl:
    def __init__(self, height=64, inter_res=8, zdim=128, dim=64, channels=1, scale=10.0, kappa=1.0):
        self.height, self.inter_res, self.zdim, self.dim, self.channels = height, inter_res, zdim, dim, channels
        self.scale, self.kappa = scale, kappa
        self.npool = 3
        self.spec, self.bg, self.bd, self.nm = param_spec(height, inter_res, zdim, dim, channels)
        self.bn_e = [n[:-len('/gamma')] for n, _, _ in self.spec if n.startswith('Encoder/batch_norm') and n.endswith('gamma')]

    # ------------------------------------------------------------------ Encoder (:15-22)
    def enc_forward(self, p, x):
        cache = {'a': [x], 'c': []}
        a = x
        for i in range(self.npool):
            c = nn.conv2d_fwd(a, p['Encoder/enc_conv2D_%d/kernel' % i], p['Encoder/enc_conv2D_%d/bias' % i], 2)
            a = nn.leaky_relu_fwd(nn.bn_frozen_fwd(c, p[self.bn_e[i] + '/gamma'], p[self.bn_e[i] + '/beta']), BN_LRELU)
            cache['c'].append(c); cache['a'].append(a)
        flat = a.reshape(a.shape[0], -1)
        z = np.tanh(nn.dense_fwd(flat, p['Encoder/dense/kernel'], p['Encoder/dense/bias']))
        cache.update(flat=flat, z=z)
        return z, cache

    def enc_backward(self, p, cache, dz):
        g = {}
        dzr = dz * (1.0 - cache['z'] ** 2)
        dflat, g['Encoder/dense/kernel'], g['Encoder/dense/bias'] = nn.dense_bwd(cache['flat'], p['Encoder/dense/kernel'], dzr)
        da = dflat.reshape(cache['a'][-1].shape)
        for i in reversed(range(self.npool)):
            c = cache['c'][i]
            bnv = nn.bn_frozen_fwd(c, p[self.bn_e[i] + '/gamma'], p[self.bn_e[i] + '/beta'])
            dc, g[self.bn_e[i] + '/gamma'], g[self.bn_e[i] + '/beta'] = nn.bn_frozen_bwd(c, p[self.bn_e[i] + '/gamma'], nn.leaky_relu_bwd(bnv, da, BN_LRELU))
            da, g['Encoder/enc_conv2D_%d/kernel' % i], g['Encoder/enc_conv2D_%d/bias' % i] = \
                nn.conv2d_bwd(cache['a'][i], p['Encoder/enc_conv2D_%d/kernel' % i], dc, 2)
        return g

    # ------------------------------------------------------------------ Generator (:119-137)
    def gen_forward(self, p, z):
        r = self.inter_res
        out = nn.dense_fwd(z, p['Generator/dense/kernel'], p['Generator/dense/bias']).reshape(z.shape[0], r, r, -1)
        cache = {'z': z, 'blocks': [], 'in': []}
        for b in self.bg:
            cache['in'].append(out)
            out, bc = b.fwd(p, out)
            cache['blocks'].append(bc)
        y, lc = ln_fwd(out, p[self.nm['gen_ln'] + '/gamma'], p[self.nm['gen_ln'] + '/beta'])
        h = np.maximum(y, 0)
        xg = np.tanh(nn.conv2d_fwd(h, p[self.nm['gen_final'] + '/kernel'], p[self.nm['gen_final'] + '/bias'], 1))
        cache.update(y=y, lc=lc, h=h, x=xg, pre=out)
        return xg, cache

    def gen_backward(self, p, cache, dx):
        g = {}
        do = dx * (1.0 - cache['x'] ** 2)
        dh, g[self.nm['gen_final'] + '/kernel'], g[self.nm['gen_final'] + '/bias'] = nn.conv2d_bwd(cache['h'], p[self.nm['gen_final'] + '/kernel'], do, 1)
        dout, g[self.nm['gen_ln'] + '/gamma'], g[self.nm['gen_ln'] + '/beta'] = ln_bwd(relu_bwd(cache['y'], dh), p[self.nm['gen_ln'] + '/gamma'], cache['lc'])
        for b, bc in zip(reversed(self.bg), reversed(cache['blocks'])):
            dout = b.bwd(p, bc, dout, g)
        dv = dout.reshape(dout.shape[0], -1)
        dz, g['Generator/dense/kernel'], g['Generator/dense/bias'] = nn.dense_bwd(cache['z'], p['Generator/dense/kernel'], dv)
        return g, dz

    # ------------------------------------------------------------------ Discriminator (:140-161)
    def disc_forward(self, p, x):
        out = nn.conv2d_fwd(x, p[self.nm['dis_conv'] + '/kernel'], p[self.nm['dis_conv'] + '/bias'], 1)
        cache = {'x': x, 'blocks': []}
        for b in self.bd:
            out, bc = b.fwd(p, out)
            cache['blocks'].append(bc)
        d = out @ p['Discriminator/dense/kernel'] + p['Discriminator/dense/bias']
        cache['feat'] = out
        return out, d, cache

    def disc_backward(self, p, cache, df=None, dd=None, inject=None, want_params=True, tapes=None):
        g = {} if want_params else None
        feat = cache['feat']
        da = np.zeros_like(feat) if df is None else df.copy()
        if dd is not None:
            da = da + dd * p['Discriminator/dense/kernel'][:, 0]
            if want_params:
                g['Discriminator/dense/kernel'] = (feat * dd).reshape(-1, feat.shape[-1]).sum(axis=0)[:, None]
                g['Discriminator/dense/bias'] = dd.sum().reshape(1)
        for i in reversed(range(len(self.bd))):
            tape = None
            if tapes is not None:
                tape = {}
                tapes[i] = tape
            da = self.bd[i].bwd(p, cache['blocks'][i], da, g, None if inject is None else inject[i], tape)
        dx, dw0, db0 = nn.conv2d_bwd(cache['x'], p[self.nm['dis_conv'] + '/kernel'], da, 1)
        if want_params:
            g[self.nm['dis_conv'] + '/kernel'] = dw0
            g[self.nm['dis_conv'] + '/bias'] = db0
        if tapes is not None:
            tapes['d0'] = da
        return g, dx

    def gradient_penalty(self, ddx):
        s = np.sqrt((ddx ** 2).sum(axis=1))
        pen = self.scale * ((s - 1.0) ** 2).mean()
        return pen, (self.scale * 2.0 * (s - 1.0) / s.size / s)[:, None, :, :] * ddx

    def disc_penalty_grads(self, p, cache, tapes, gbar):
        g = {}
        w0 = p[self.nm['dis_conv'] + '/kernel']
        ub = nn.conv2d_fwd(gbar, w0, None, 1)
        _, g[self.nm['dis_conv'] + '/kernel'], _ = nn.conv2d_bwd(gbar, w0, tapes['d0'], 1)
        inject = [None] * len(self.bd)
        for i, b in enumerate(self.bd):
            ub, inject[i] = b.adj(p, cache['blocks'][i], tapes[i], ub, g)
        g['Discriminator/dense/kernel'] = ub.reshape(-1, ub.shape[-1]).sum(axis=0)[:, None]
        return g, inject

    # ------------------------------------------------------------------ phases
    def gen_phase(self, p, z, caches=None):
        xg, gc = self.gen_forward(p, z)
        _, d, dcache = self.disc_forward(p, xg)
        if caches is not None:
            caches.update(gen=gc, disc=[dcache])
        _, dx = self.disc_backward(p, dcache, dd=np.full_like(d, -1.0 / d.size), want_params=False)
        grads, _ = self.gen_backward(p, gc, dx)
        return {'gen_loss': -d.mean(), 'generated': xg}, grads

    def disc_phase(self, p, x, z, alpha, caches=None):
        xg, gcache = self.gen_forward(p, z)
        _, d_fake, c_fake = self.disc_forward(p, xg)
        _, d_real, c_real = self.disc_forward(p, x)
        x_hat = x + alpha.reshape(-1, 1, 1, 1).astype(x.dtype) * (xg - x)
        _, d_hat, c_hat = self.disc_forward(p, x_hat)
        if caches is not None:
            caches.update(gen=gcache, disc=[c_fake, c_real, c_hat])
        tapes = {}
        u = np.broadcast_to(p['Discriminator/dense/kernel'][:, 0], c_hat['feat'].shape).astype(x.dtype)
        _, ddx = self.disc_backward(p, c_hat, df=u, want_params=False, tapes=tapes)
        pen, gbar = self.gradient_penalty(ddx)
        losses = {'disc_fake': d_fake.mean(), 'disc_real': d_real.mean(), 'generated': xg, 'penalty': pen, 'ddx': ddx}
        losses['disc_loss'] = losses['disc_fake'] - losses['disc_real'] + pen
        g_f, _ = self.disc_backward(p, c_fake, dd=np.full_like(d_fake, 1.0 / d_fake.size))
        g_r, _ = self.disc_backward(p, c_real, dd=np.full_like(d_real, -1.0 / d_real.size))
        g_2, inject = self.disc_penalty_grads(p, c_hat, tapes, gbar)
        g_3, _ = self.disc_backward(p, c_hat, inject=inject)
        grads = {}
        for part in (g_f, g_r, g_2, g_3):
            for k, v in part.items():
                grads[k] = grads.get(k, 0) + v
        return losses, grads

    def enc_phase(self, p, x, caches=None):
        z_enc, ec = self.enc_forward(p, x)
        x_enc, gc = self.gen_forward(p, z_enc)
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
        _, dx = self.disc_backward(p, c_enc, df=self.kappa * 2.0 * (f_enc - f_real) / f_enc.size, want_params=False)
        dx = dx + 2.0 * (x_enc - x) / x.size
        _, dz = self.gen_backward(p, gc, dx)
        return losses, self.enc_backward(p, ec, dz)

    def reconstruct(self, p, x):
        return self.gen_forward(p, self.enc_forward(p, x)[0])[0]
