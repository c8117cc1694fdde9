"""Oracle for the constrained adversarial autoencoder on residual blocks: models/constrained_adversarial_autoencoder_Chen.py:11-162 under
trainers/ConstrainedAAE.py:44-70.  Encoder = k3 conv + four pre-activation residual blocks (LayerNorm-HW -> ReLU -> k3 conv -> LayerNorm-HW
-> ReLU -> k3 conv, shortcut AvgPool(1x1 conv); the last block at stride 1 with the identity) -> Flatten -> Dense(zDim); Decoder = Dense ->
[r, r, 8d] -> four residual blocks (k3 conv, k3 transposed conv, k1 s2 transposed-conv shortcut; the first at stride 1 with the identity)
-> LayerNorm-HW -> ReLU -> 1x1 conv; critic = MLP zDim -> 400 -> 200 -> 1 with tf.nn.leaky_relu on z_, the prior sample z and
z_hat = eps z + (1 - eps) z_ with ONE scalar eps per run (:118-119).  The same block structure as models/fanogan_schlegl.py (whose critic
is this encoder and whose generator is this decoder), so the blocks are oracle/fanogan_schlegl.py's; the phases are oracle/aae.py's.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED (no TensorFlow here, no golden vectors in the reference);
tests/test_oracle_caae_chen.py anchors every gradient on torch autograd in float64."""
import numpy as np

from . import nn
from .aae import AAE
from .fanogan import ln_fwd, ln_bwd
from .fanogan_schlegl import _Block, relu_bwd


def param_spec(height=64, zdim=128, dim=64, channels=1):
    """TF variable-creation (= first-call) order; tf.layers names count per scope in call order, keras LayerNormalization names globally."""
    assert height % 8 == 0
    r = height // 8
    spec, ln, cnt = [], [0], {}

    def ln_pair(scope, res):
        name = scope + ('layer_normalization' if ln[0] == 0 else 'layer_normalization_%d' % ln[0])
        ln[0] += 1
        return name, [(name + '/gamma', (res, res), 'gamma'), (name + '/beta', (res, res), 'beta')]

    def tfname(scope, base):
        k = cnt.get((scope, base), 0)
        cnt[(scope, base)] = k + 1
        return scope + base + ('' if k == 0 else '_%d' % k)

    e0 = tfname('Encoder/', 'conv2d')
    spec += [(e0 + '/kernel', (3, 3, channels, dim), 'conv_w'), (e0 + '/bias', (dim,), 'bias')]
    enc, dec = [], []
    cin, res = dim, height
    for f, stride in ((2 * dim, 2), (4 * dim, 2), (8 * dim, 2), (8 * dim, 1)):
        n1, s1 = ln_pair('Encoder/', res)
        c1 = tfname('Encoder/', 'conv2d')
        n2, s2 = ln_pair('Encoder/', res)
        c2 = tfname('Encoder/', 'conv2d')
        spec += s1 + [(c1 + '/kernel', (3, 3, cin, f), 'conv_w'), (c1 + '/bias', (f,), 'bias')] + s2
        spec += [(c2 + '/kernel', (3, 3, f, f), 'conv_w'), (c2 + '/bias', (f,), 'bias')]
        sh = None
        if stride == 2:
            sh = tfname('Encoder/', 'conv2d')
            spec += [(sh + '/kernel', (1, 1, cin, f), 'conv_w'), (sh + '/bias', (f,), 'bias')]
        enc.append(_Block('d', dict(ln1=n1, conv1=c1, ln2=n2, conv2=c2, short=sh), stride))
        cin, res = f, res // stride
    flat = r * r * 8 * dim
    spec += [('Encoder/dense/kernel', (flat, zdim), 'dense_w'), ('Encoder/dense/bias', (zdim,), 'bias'),
             ('Decoder/dense/kernel', (zdim, flat), 'dense_w'), ('Decoder/dense/bias', (flat,), 'bias')]
    cin, res = 8 * dim, r
    for f, stride in ((8 * dim, 1), (4 * dim, 2), (2 * dim, 2), (dim, 2)):
        n1, s1 = ln_pair('Decoder/', res)
        c1 = tfname('Decoder/', 'conv2d')
        n2, s2 = ln_pair('Decoder/', res)
        c2 = tfname('Decoder/', 'conv2d_transpose')
        spec += s1 + [(c1 + '/kernel', (3, 3, cin, f), 'conv_w'), (c1 + '/bias', (f,), 'bias')] + s2
        spec += [(c2 + '/kernel', (3, 3, f, f), 'conv_w'), (c2 + '/bias', (f,), 'bias')]
        sh = None
        if stride == 2:
            sh = tfname('Decoder/', 'conv2d_transpose')
            spec += [(sh + '/kernel', (1, 1, f, cin), 'conv_w'), (sh + '/bias', (f,), 'bias')]
        dec.append(_Block('g', dict(ln1=n1, conv1=c1, ln2=n2, conv2=c2, short=sh), stride))
        cin, res = f, res * stride
    gl, sg = ln_pair('Decoder/', res)
    gf = tfname('Decoder/', 'conv2d')
    spec += sg + [(gf + '/kernel', (1, 1, cin, channels), 'conv_w'), (gf + '/bias', (channels,), 'bias')]
    spec += [('Discriminator/dense/kernel', (zdim, 400), 'dense_w'), ('Discriminator/dense/bias', (400,), 'bias'),
             ('Discriminator/dense_1/kernel', (400, 200), 'dense_w'), ('Discriminator/dense_1/bias', (200,), 'bias'),
             ('Discriminator/dense_2/kernel', (200, 1), 'dense_w'), ('Discriminator/dense_2/bias', (1,), 'bias')]
    return spec, enc, dec, dict(enc_conv=e0, dec_ln=gl, dec_final=gf)


class CAAEChen(AAE):
    def __init__(self, height=64, zdim=128, dim=64, rho=1.0, scale=10.0):
        self.kind, self.height, self.inter_res, self.zdim, self.rho, self.scale, self.dim = 'caae_chen', height, height // 8, zdim, rho, scale, dim
        self.constrained, self.has_critic = True, True
        self.spec, self.be, self.bd, self.nm = param_spec(height, zdim, dim)

    def _z_hat(self, z_prior, z_, eps):
        return eps * z_prior + (1.0 - eps) * z_              # :119

    def encode(self, p, x, mask_z=None):
        a = nn.conv2d_fwd(x, p[self.nm['enc_conv'] + '/kernel'], p[self.nm['enc_conv'] + '/bias'], 1)
        cache = {'x': x, 'blocks': []}
        for b in self.be:
            a, c = b.fwd(p, a)
            cache['blocks'].append(c)
        cache['feat'] = a
        z = nn.dense_fwd(a.reshape(a.shape[0], -1), p['Encoder/dense/kernel'], p['Encoder/dense/bias'])
        return z, cache

    def encode_backward(self, p, cache, dz, g):
        def acc(k, v):
            g[k] = g.get(k, 0) + v
        feat = cache['feat']
        dflat, dw, db = nn.dense_bwd(feat.reshape(feat.shape[0], -1), p['Encoder/dense/kernel'], dz)
        acc('Encoder/dense/kernel', dw); acc('Encoder/dense/bias', db)
        da = dflat.reshape(feat.shape)
        for b, c in zip(reversed(self.be), reversed(cache['blocks'])):
            da = b.bwd(p, c, da, g)
        dx, dw, db = nn.conv2d_bwd(cache['x'], p[self.nm['enc_conv'] + '/kernel'], da, 1)
        acc(self.nm['enc_conv'] + '/kernel', dw); acc(self.nm['enc_conv'] + '/bias', db)
        return dx

    def decode(self, p, z, mask_dec=None):
        r = self.inter_res
        a = nn.dense_fwd(z, p['Decoder/dense/kernel'], p['Decoder/dense/bias']).reshape(z.shape[0], r, r, -1)
        cache = {'z': z, 'blocks': []}
        for b in self.bd:
            a, c = b.fwd(p, a)
            cache['blocks'].append(c)
        y, l = ln_fwd(a, p[self.nm['dec_ln'] + '/gamma'], p[self.nm['dec_ln'] + '/beta'])
        h = np.maximum(y, 0)
        cache.update(y=y, l=l, h=h)
        return nn.conv2d_fwd(h, p[self.nm['dec_final'] + '/kernel'], p[self.nm['dec_final'] + '/bias'], 1), cache

    def decode_backward(self, p, cache, dxh, g):
        dh, g[self.nm['dec_final'] + '/kernel'], g[self.nm['dec_final'] + '/bias'] = nn.conv2d_bwd(cache['h'], p[self.nm['dec_final'] + '/kernel'], dxh, 1)
        da, g[self.nm['dec_ln'] + '/gamma'], g[self.nm['dec_ln'] + '/beta'] = ln_bwd(relu_bwd(cache['y'], dh), p[self.nm['dec_ln'] + '/gamma'], cache['l'])
        for b, c in zip(reversed(self.bd), reversed(cache['blocks'])):
            da = b.bwd(p, c, da, g)
        dz, g['Decoder/dense/kernel'], g['Decoder/dense/bias'] = nn.dense_bwd(cache['z'], p['Decoder/dense/kernel'], da.reshape(da.shape[0], -1))
        return dz
