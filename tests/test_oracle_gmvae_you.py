"""CPU: numpy oracle of the original-architecture spatial GMVAE (oracle/gmvae_you.py) vs an autograd graph written like the reference's
(models/gaussian_mixture_variational_autoencoder_You.py + trainers/GMVAE_spatial.py:61-97), fp64."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import gmvae as og
from oracle import gmvae_you as oy
from tests import torch_ref


@pytest.mark.parametrize('dim_c,dim_z,dim_w,c_lambda', [(6, 1, 1, 1.0), (4, 3, 2, 0.01)])
def test_gmvae_you_matches_autograd(dim_c, dim_z, dim_w, c_lambda):
    m = oy.GMVAEYou(16, dim_c, dim_z, dim_w, c_lambda)
    p = og.init_params(m.spec, seed=2, dtype=np.float64, perturb=True)
    rng = np.random.default_rng(1)
    n = 2
    x = rng.random((n, 16, 16, 1))
    e_w, e_z = rng.standard_normal((n, 4, 4, dim_w)), rng.standard_normal((n, 4, 4, dim_z))
    out, cache = m.forward(p, x, e_w, e_z)
    ls = m.losses(x, out, 1.8)
    g = m.backward(p, x, out, cache)
    gr = m.backward(p, x, out, cache, tv_lambda=1.8)['__dx']
    tp = {k: torch.tensor(v, requires_grad=True) for k, v in p.items()}
    xt = torch.tensor(x).permute(0, 3, 1, 2).requires_grad_(True)
    a = xt
    for name, s in oy.ENC:
        a = F.relu(torch_ref._conv_same(a, tp[name + '/kernel'], tp[name + '/bias'], s))
    lin = lambda t, name: torch_ref._conv_same(t, tp[name + '/kernel'], tp[name + '/bias'], 1)
    nchw = lambda v: torch.tensor(v).permute(0, 3, 1, 2)
    w_mu, w_ls, z_mu, z_ls = lin(a, 'q_wz_x/w_mu'), lin(a, 'q_wz_x/w_log_sigma'), lin(a, 'q_wz_x/z_mu'), lin(a, 'q_wz_x/z_log_sigma')
    w_s = w_mu + nchw(e_w) * torch.exp(0.5 * w_ls)
    z_s = z_mu + nchw(e_z) * torch.exp(0.5 * z_ls)
    mid = F.relu(lin(w_s, 'p_z_wc/1x1convlayer'))
    nhwc = lambda t: t.permute(0, 2, 3, 1)
    M = nhwc(lin(mid, 'p_z_wc/z_wc_mu')).reshape(n, 4, 4, dim_z, dim_c)
    Lq = (nhwc(lin(mid, 'p_z_wc/z_wc_log_sigma')) + tp['Variable']).reshape(n, 4, 4, dim_z, dim_c)
    d = z_s
    for kind, name, relu in oy.DEC:
        if kind == 'up':
            d = d.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
            continue
        d = (torch_ref._conv_same if kind == 'conv' else torch_ref._convT_same)(d, tp[name + '/kernel'], tp[name + '/bias'], 1)
        if relu:
            d = F.relu(d)
    xh = d
    zs5 = nhwc(z_s).unsqueeze(-1)
    pc = torch.softmax((-0.5 * ((zs5 - M) ** 2 * torch.exp(Lq)) - Lq + math.log(math.pi)).sum(3), dim=-1)
    zm5, zl5 = nhwc(z_mu).unsqueeze(-1), nhwc(z_ls).unsqueeze(-1)
    kl = 0.5 * ((torch.exp(zl5) + (zm5 - M) ** 2) * (torch.exp(Lq) + 1e-6) - (Lq + zl5) - 1)
    con = (kl * pc.unsqueeze(3)).sum(dim=(1, 2, 3, 4)).mean()
    wl = (0.5 * (w_mu ** 2 + torch.exp(w_ls) - w_ls - 1).sum(dim=(1, 2, 3))).mean()
    cl1 = (pc * torch.log(pc * dim_c + 1e-8)).sum(3)
    cl = torch.maximum(cl1, torch.full_like(cl1, c_lambda)).sum(dim=(1, 2)).mean()
    rec = (xt - xh).abs().sum(dim=(1, 2, 3)).mean()
    loss = rec + con + wl + cl
    np.testing.assert_allclose(out['xz_mu'], nhwc(xh).detach().numpy(), rtol=1e-9, atol=1e-11)
    for k, v in (('mean_p_loss', rec), ('conditional_prior_loss', con), ('w_prior_loss', wl), ('c_prior_loss', cl), ('loss', loss)):
        assert ls[k] == pytest.approx(float(v.detach()), rel=1e-10), k
    names = [s[0] for s in m.spec]
    grads = torch.autograd.grad(loss, [tp[k] for k in names] + [xt], retain_graph=True)
    for k, tg in zip(names, grads[:-1]):
        assert np.abs(g[k] - tg.numpy()).max() <= 1e-9 * max(np.abs(tg.numpy()).max(), 1e-6), k
    np.testing.assert_allclose(g['__dx'], nhwc(grads[-1]).numpy(), rtol=1e-8, atol=1e-13)
    r = xt - xh
    tvn = (r[:, :, 1:, :] - r[:, :, :-1, :]).abs().sum(dim=(1, 2, 3)) + (r[:, :, :, 1:] - r[:, :, :, :-1]).abs().sum(dim=(1, 2, 3))
    ref = torch.autograd.grad((loss + 1.8 * tvn).sum(), xt)[0]
    np.testing.assert_allclose(gr, nhwc(ref).numpy(), rtol=1e-8, atol=1e-12)


def test_gmvae_you_spec():
    names = [s[0] for s in oy.param_spec(6, 1, 1)]
    i = names.index
    assert i('q_wz_x/3x3convlayer5/bias') < i('q_wz_x/w_mu/kernel') < i('q_wz_x/z_log_sigma/bias') < i('p_z_wc/1x1convlayer/kernel') < i('Variable') \
        < i('p_x_z/3x3convlayer1/kernel') < i('p_x_z/3x3upconvlayer2/bias') < i('p_x_z/3x3convlayer2/kernel') < i('p_x_z/y_mu/kernel')
    sh = dict((s[0], s[1]) for s in oy.param_spec(6, 2, 3))
    assert sh['p_x_z/3x3convlayer1/kernel'] == (3, 3, 2, 64) and sh['p_x_z/3x3upconvlayer1/kernel'] == (3, 3, 64, 64) and sh['p_x_z/y_mu/kernel'] == (3, 3, 64, 1)
