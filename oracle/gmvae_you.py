"""Oracle for the original-architecture spatial GMVAE: models/gaussian_mixture_variational_autoencoder_You.py:8-85 under
trainers/GMVAE_spatial.py:61-97 (losses, the `grads` fetch of the restoration).  numpy forward, hand-written backward.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED (no TensorFlow here, no golden vectors in the reference);
tests/test_oracle_gmvae_you.py anchors every gradient on torch autograd in float64.

Graph: six k3 convolutions of 64 filters with ReLU (strides 2,1,1,2,1,1) -> 1x1 heads q(w|x), q(z|x) on the H/4 map -> p(z|w,c)
(1x1 conv 64 + ReLU, two 1x1 heads, the 0.1 Variable) -> p(c|z);  decoder ON z_sampled: k3 conv + ReLU, two k3 s1 transposed convolutions
+ ReLU, nearest-neighbour x2, k3 conv + ReLU, two transposed convolutions + ReLU, nearest-neighbour x2, k3 conv (NO activation, :66),
k3 conv to one channel.  No variable scope: layer names are the explicit `name=` strings; variables appear in first-call order."""
import numpy as np

from . import gmvae as og
from . import nn

ENC = [('q_wz_x/3x3convlayer', 2), ('q_wz_x/3x3convlayer1', 1), ('q_wz_x/3x3convlayer2', 1), ('q_wz_x/3x3convlayer3', 2),
       ('q_wz_x/3x3convlayer4', 1), ('q_wz_x/3x3convlayer5', 1)]
# decoder program: ('conv' | 'convT' | 'up', name, relu)
DEC = [('conv', 'p_x_z/3x3convlayer1', True), ('convT', 'p_x_z/3x3upconvlayer1', True), ('convT', 'p_x_z/3x3upconvlayer2', True), ('up', None, False),
       ('conv', 'p_x_z/3x3convlayer2', True), ('convT', 'p_x_z/3x3upconvlayer3', True), ('convT', 'p_x_z/3x3upconvlayer4', True), ('up', None, False),
       ('conv', 'p_x_z/3x3convlayer3', False), ('conv', 'p_x_z/y_mu', False)]
F = 64


def param_spec(dim_c=6, dim_z=1, dim_w=1, channels=1):
    spec, cin = [], channels
    for name, _ in ENC:
        spec += [(name + '/kernel', (3, 3, cin, F), 'conv_w'), (name + '/bias', (F,), 'bias')]
        cin = F
    for nm, co in (('q_wz_x/w_mu', dim_w), ('q_wz_x/w_log_sigma', dim_w), ('q_wz_x/z_mu', dim_z), ('q_wz_x/z_log_sigma', dim_z)):
        spec += [(nm + '/kernel', (1, 1, F, co), 'conv_w'), (nm + '/bias', (co,), 'bias')]
    q = dim_z * dim_c
    spec += [('p_z_wc/1x1convlayer/kernel', (1, 1, dim_w, 64), 'conv_w'), ('p_z_wc/1x1convlayer/bias', (64,), 'bias'),
             ('p_z_wc/z_wc_mu/kernel', (1, 1, 64, q), 'conv_w'), ('p_z_wc/z_wc_mu/bias', (q,), 'bias'),
             ('p_z_wc/z_wc_log_sigma/kernel', (1, 1, 64, q), 'conv_w'), ('p_z_wc/z_wc_log_sigma/bias', (q,), 'bias'),
             ('Variable', (q,), 'const0.1')]
    cin = dim_z
    for kind, name, _ in DEC:
        if kind == 'up':
            continue
        cout = channels if name.endswith('y_mu') else F
        shape = (3, 3, cin, cout) if kind == 'conv' else (3, 3, cout, cin)
        spec += [(name + '/kernel', shape, 'conv_w'), (name + '/bias', (cout,), 'bias')]
        cin = cout
    return spec


def _up(a):
    return a.repeat(2, axis=1).repeat(2, axis=2)


def _up_bwd(g):
    n, h, w, c = g.shape
    return g.reshape(n, h // 2, 2, w // 2, 2, c).sum(axis=(2, 4))


class GMVAEYou:
    def __init__(self, height=128, dim_c=6, dim_z=1, dim_w=1, c_lambda=1.0):
        assert height % 4 == 0
        self.h, self.dim_c, self.dim_z, self.dim_w, self.c_lambda = height, dim_c, dim_z, dim_w, float(c_lambda)
        self.spec = param_spec(dim_c, dim_z, dim_w)

    def forward(self, p, x, e_w, e_z):
        """e_w [n,H/4,W/4,dim_w], e_z [n,H/4,W/4,dim_z]."""
        cache = {'ea': [x], 'ec': []}
        a = x
        for name, s in ENC:
            c = nn.conv2d_fwd(a, p[name + '/kernel'], p[name + '/bias'], s)
            a = np.maximum(c, 0)
            cache['ec'].append(c); cache['ea'].append(a)
        h = a
        lin = lambda t, name: nn.conv2d_fwd(t, p[name + '/kernel'], p[name + '/bias'], 1)
        w_mu, w_ls = lin(h, 'q_wz_x/w_mu'), lin(h, 'q_wz_x/w_log_sigma')
        z_mu, z_ls = lin(h, 'q_wz_x/z_mu'), lin(h, 'q_wz_x/z_log_sigma')
        w_s = w_mu + e_w * np.exp(0.5 * w_ls)
        z_s = z_mu + e_z * np.exp(0.5 * z_ls)
        a7 = lin(w_s, 'p_z_wc/1x1convlayer')
        mid = np.maximum(a7, 0)
        n, hh, ww = h.shape[:3]
        M = lin(mid, 'p_z_wc/z_wc_mu').reshape(n, hh, ww, self.dim_z, self.dim_c)
        Lq = (lin(mid, 'p_z_wc/z_wc_log_sigma') + p['Variable']).reshape(n, hh, ww, self.dim_z, self.dim_c)
        logit = (-0.5 * ((z_s[..., None] - M) ** 2 * np.exp(Lq)) - Lq + np.log(np.pi)).sum(axis=3)
        ex = np.exp(logit - logit.max(axis=-1, keepdims=True))
        pc = ex / ex.sum(axis=-1, keepdims=True)
        cache.update(h=h, w_mu=w_mu, w_ls=w_ls, z_mu=z_mu, z_ls=z_ls, w_s=w_s, z_s=z_s, a7=a7, mid=mid, M=M, Lq=Lq, pc=pc, e_w=e_w, e_z=e_z,
                     din=[], dc=[])
        a = z_s
        for kind, name, relu in DEC:
            cache['din'].append(a)
            if kind == 'up':
                a = _up(a); cache['dc'].append(None)
                continue
            c = (nn.conv2d_fwd if kind == 'conv' else nn.conv2d_transpose_fwd)(a, p[name + '/kernel'], p[name + '/bias'], 1)
            cache['dc'].append(c)
            a = np.maximum(c, 0) if relu else c
        out = {'xz_mu': a, 'w_mu': w_mu, 'w_log_sigma': w_ls, 'z_mu': z_mu, 'z_log_sigma': z_ls, 'w_sampled': w_s, 'z_sampled': z_s,
               'z_wc_mus': M, 'z_wc_log_sigma_invs': Lq, 'pc_logit': logit, 'pc': pc}
        return out, cache

    def losses(self, x, out, tv_lambda=0.0):
        return og.GMVAE.losses(self, x, out, tv_lambda)          # trainers/GMVAE_spatial.py:61-89: only dim_c / c_lambda are read

    def backward(self, p, x, out, cache, tv_lambda=None):
        """tv_lambda None: d loss / d params, g['__dx'] = d loss / d x; tv_lambda given: the `grads` fetch (n loss + sum_n tv TV_n, see
        oracle/gmvae.py), only g['__dx'] is meaningful then."""
        n = x.shape[0]
        dt = x.dtype.type
        inv = dt(1.0 / n) if tv_lambda is None else dt(1.0)
        g = {}
        C = self.dim_c
        gx = np.sign(out['xz_mu'] - x) * inv
        dx_direct = -gx
        if tv_lambda is not None:
            tvg = og.total_variation_grad(x - out['xz_mu']) * dt(tv_lambda)
            gx = gx - tvg
            dx_direct = dx_direct + tvg
        da = gx
        for (kind, name, relu), a_in, c in zip(reversed(DEC), reversed(cache['din']), reversed(cache['dc'])):
            if kind == 'up':
                da = _up_bwd(da)
                continue
            dc = nn.leaky_relu_bwd(c, da, 0.0) if relu else da
            da, g[name + '/kernel'], g[name + '/bias'] = (nn.conv2d_bwd if kind == 'conv' else nn.conv2d_transpose_bwd)(a_in, p[name + '/kernel'], dc, 1)
        dz_dec = da
        c = cache
        pc, M, Lq = c['pc'], c['M'], c['Lq']
        z_mu, z_ls, z_s, w_mu, w_ls = c['z_mu'], c['z_ls'], c['z_s'], c['w_mu'], c['w_ls']
        E = np.exp(Lq); E6 = E + 1e-6
        V = np.exp(z_ls)[..., None]
        D2 = z_mu[..., None] - M
        kl = 0.5 * ((V + D2 ** 2) * E6 - (Lq + z_ls[..., None]) - 1)
        dkl = inv * np.broadcast_to(pc[:, :, :, None, :], kl.shape)
        dpc = inv * kl.sum(axis=3)
        cl1 = (pc * np.log(pc * C + 1e-8)).sum(axis=3)
        act = (cl1 >= self.c_lambda)[..., None]
        dpc = dpc + inv * act * (np.log(pc * C + 1e-8) + pc * C / (pc * C + 1e-8))
        dlogit = pc * (dpc - (dpc * pc).sum(axis=-1, keepdims=True))
        dll = np.broadcast_to(dlogit[:, :, :, None, :], kl.shape)
        D = z_s[..., None] - M
        dz_s = (dll * (-D * E)).sum(axis=-1) + dz_dec
        dM = dll * (D * E) - dkl * D2 * E6
        dLq = dll * (-0.5 * D ** 2 * E - 1) + dkl * 0.5 * ((V + D2 ** 2) * E - 1)
        dz_mu = (dkl * D2 * E6).sum(axis=-1) + dz_s
        dz_ls = (dkl * 0.5 * (V * E6 - 1)).sum(axis=-1) + dz_s * c['e_z'] * 0.5 * np.exp(0.5 * z_ls)
        nb, hh, ww = pc.shape[:3]
        dMf, dLqf = dM.reshape(nb, hh, ww, -1), dLq.reshape(nb, hh, ww, -1)
        g['Variable'] = dLqf.sum(axis=(0, 1, 2))
        dmid1, g['p_z_wc/z_wc_mu/kernel'], g['p_z_wc/z_wc_mu/bias'] = nn.conv2d_bwd(c['mid'], p['p_z_wc/z_wc_mu/kernel'], dMf, 1)
        dmid2, g['p_z_wc/z_wc_log_sigma/kernel'], g['p_z_wc/z_wc_log_sigma/bias'] = nn.conv2d_bwd(c['mid'], p['p_z_wc/z_wc_log_sigma/kernel'], dLqf, 1)
        da7 = nn.leaky_relu_bwd(c['a7'], dmid1 + dmid2, 0.0)
        dw_s, g['p_z_wc/1x1convlayer/kernel'], g['p_z_wc/1x1convlayer/bias'] = nn.conv2d_bwd(c['w_s'], p['p_z_wc/1x1convlayer/kernel'], da7, 1)
        dw_mu = inv * w_mu + dw_s
        dw_ls = inv * 0.5 * (np.exp(w_ls) - 1) + dw_s * c['e_w'] * 0.5 * np.exp(0.5 * w_ls)
        dh = 0
        for name, dv in (('q_wz_x/w_mu', dw_mu), ('q_wz_x/w_log_sigma', dw_ls), ('q_wz_x/z_mu', dz_mu), ('q_wz_x/z_log_sigma', dz_ls)):
            dhh, g[name + '/kernel'], g[name + '/bias'] = nn.conv2d_bwd(c['h'], p[name + '/kernel'], dv, 1)
            dh = dh + dhh
        da = dh
        for i in reversed(range(len(ENC))):
            name, s = ENC[i]
            dc = nn.leaky_relu_bwd(cache['ec'][i], da, 0.0)
            da, g[name + '/kernel'], g[name + '/bias'] = nn.conv2d_bwd(cache['ea'][i], p[name + '/kernel'], dc, s)
        g['__dx'] = da + dx_direct
        return g

    def new_opt(self, p):
        return {'t': 0, 'm': {k: np.zeros_like(v) for k, v in p.items()}, 'v': {k: np.zeros_like(v) for k, v in p.items()}}

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
