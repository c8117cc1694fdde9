"""CPU: numpy oracle of the dense GMVAE (oracle/gmvae_dense.py) vs an autograd graph written like the reference's
(models/gaussian_mixture_variational_autoencoder.py, trainers/GMVAE.py:56-94), fp64: losses, every parameter gradient, and the
`grads` fetch of the restoration (loss + tv * TV(x - rec) w.r.t. x)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import gmvae_dense as ogd
from tests import torch_ref


def _graph(m, tp, xt, e_w, e_z, masks, tv=None):
    n = xt.shape[0]
    core = m.core
    t = torch.tensor

    def bn(v, name):
        return v * (tp[name + '/gamma'] / math.sqrt(1.001)).view(1, -1, 1, 1) + tp[name + '/beta'].view(1, -1, 1, 1)
    a = xt
    for i in range(core.npool):
        a = F.leaky_relu(bn(torch_ref._conv_same(a, tp['Encoder/enc_conv2D_%d/kernel' % i], tp['Encoder/enc_conv2D_%d/bias' % i], 2), core.bn_e[i]), 0.3)
    tt = torch_ref._conv_same(a, tp['Bottleneck/conv2d/kernel'], tp['Bottleneck/conv2d/bias'], 1)
    flat = tt.permute(0, 2, 3, 1).reshape(n, -1)
    hv = {}
    for h in ogd.HEADS:
        v = flat @ tp[m.head[h] + '/kernel'] + tp[m.head[h] + '/bias']
        hv[h] = v * t(masks[h]) if masks.get(h) is not None else v
    w_s = hv['w_mu'] + t(e_w) * torch.exp(0.5 * hv['w_ls'])
    z_s = hv['z_mu'] + t(e_z) * torch.exp(0.5 * hv['z_ls'])
    dv = z_s @ tp['Bottleneck/dense_4/kernel'] + tp['Bottleneck/dense_4/bias']
    if masks.get('dec') is not None:
        dv = dv * t(masks['dec'])
    a = dv.reshape(n, m.inter_res, m.inter_res, -1).permute(0, 3, 1, 2)
    a = F.relu(bn(torch_ref._conv_same(a, tp['Bottleneck/conv2d_1/kernel'], tp['Bottleneck/conv2d_1/bias'], 1), 'Decoder/batch_normalization'))
    for i in range(core.npool):
        a = F.leaky_relu(bn(torch_ref._convT_same(a, tp['Decoder/dec_Conv2DT_%d/kernel' % i], tp['Decoder/dec_Conv2DT_%d/bias' % i], 2),
                            'Decoder/batch_normalization_%d' % (i + 1)), 0.3)
    xh = torch_ref._conv_same(a, tp['Decoder/dec_Conv2D_final/kernel'], tp['Decoder/dec_Conv2D_final/bias'], 1)
    M = (w_s @ tp['dense/kernel'] + tp['dense/bias']).reshape(n, m.dim_z, m.dim_c)
    Lq = (w_s @ tp['dense_1/kernel'] + tp['dense_1/bias'] + tp['Variable']).reshape(n, m.dim_z, m.dim_c)
    zt = z_s.unsqueeze(-1).expand(-1, -1, m.dim_c)
    loglh = -0.5 * ((zt - M) ** 2 * torch.exp(Lq)) - Lq + math.log(math.pi)
    pc = torch.softmax(loglh.sum(1), dim=-1)
    o = {'xz_mu': xh.permute(0, 2, 3, 1), 'pc': pc}
    o['mean_p_loss'] = (xt - xh).abs().sum(dim=(1, 2, 3)).mean()
    zm = hv['z_mu'].unsqueeze(-1).expand(-1, -1, m.dim_c)
    zl = hv['z_ls'].unsqueeze(-1).expand(-1, -1, m.dim_c)
    kl = ((torch.exp(zl) + (zm - M) ** 2) * (torch.exp(Lq) + 1e-6) - (Lq + zl) - 1) * 0.5
    o['conditional_prior_loss'] = torch.matmul(kl, pc.unsqueeze(-1)).squeeze(-1).sum(1).mean()
    o['w_prior_loss'] = (0.5 * (hv['w_mu'] ** 2 + torch.exp(hv['w_ls']) - hv['w_ls'] - 1).sum(1)).mean()
    closs1 = (pc * torch.log(pc * m.dim_c + 1e-8)).sum(1)
    o['c_prior_loss'] = torch.maximum(closs1, torch.full_like(closs1, m.c_lambda)).mean()
    o['loss'] = o['mean_p_loss'] + o['conditional_prior_loss'] + o['w_prior_loss'] + o['c_prior_loss']
    if tv is not None:
        r = (xt - xh)
        tvn = (r[:, :, 1:, :] - r[:, :, :-1, :]).abs().sum(dim=(1, 2, 3)) + (r[:, :, :, 1:] - r[:, :, :, :-1]).abs().sum(dim=(1, 2, 3))
        o['restore'] = tv * tvn
    return o


def _setup(dim_c, dim_z, dim_w, c_lambda, n=3, h=32, seed=0):
    m = ogd.GMVAEDense(h, 8, dim_c, dim_z, dim_w, c_lambda)
    p = ogd.init_params(m.spec, seed=seed)
    rng = np.random.default_rng(seed + 1)
    x = rng.random((n, h, h, 1))
    e_w, e_z = rng.standard_normal((n, dim_w)), rng.standard_normal((n, dim_z))
    flat = 8 * 8 * (min(128, 32 * 2 ** (m.core.npool - 1)) // 8)
    keep = lambda s: (rng.random(s) >= 0.2) / 0.8
    masks = {'w_mu': keep((n, dim_w)), 'w_ls': keep((n, dim_w)), 'z_mu': keep((n, dim_z)), 'dec': keep((n, flat))}
    return m, p, x, e_w, e_z, masks


@pytest.mark.parametrize('dim_c,dim_z,dim_w,c_lambda,use_masks', [(6, 1, 1, 1.0, False), (5, 4, 3, 0.05, True), (9, 8, 2, 0.0, True)])
def test_losses_and_gradients_match_autograd(dim_c, dim_z, dim_w, c_lambda, use_masks):
    m, p, x, e_w, e_z, masks = _setup(dim_c, dim_z, dim_w, c_lambda)
    if not use_masks:
        masks = {}
    out, cache = m.forward(p, x, e_w, e_z, masks)
    ls = m.losses(x, out)
    g = m.backward(p, x, out, cache)
    tp = {k: torch.tensor(v, requires_grad=True) for k, v in p.items()}
    xt = torch.tensor(x).permute(0, 3, 1, 2).requires_grad_(True)
    o = _graph(m, tp, xt, e_w, e_z, masks)
    np.testing.assert_allclose(out['xz_mu'], o['xz_mu'].detach().numpy(), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(out['pc'], o['pc'].detach().numpy(), rtol=1e-9, atol=1e-12)
    for k in ('mean_p_loss', 'conditional_prior_loss', 'w_prior_loss', 'c_prior_loss', 'loss'):
        assert ls[k] == pytest.approx(float(o[k].detach()), rel=1e-10), k
    names = [s[0] for s in m.spec]
    grads = torch.autograd.grad(o['loss'], [tp[k] for k in names] + [xt])
    for k, tg in zip(names, grads[:-1]):
        ref = tg.numpy()
        assert np.abs(g[k] - ref).max() <= 1e-9 * max(np.abs(ref).max(), 1e-6), k
    np.testing.assert_allclose(g['__dx'], grads[-1].permute(0, 2, 3, 1).numpy(), rtol=1e-8, atol=1e-12)
    if c_lambda == 1.0:
        assert ls['c_prior_loss'] == 1.0                      # the clamp is active at the default c_lambda (closs1 <= log dim_c... < 1 here)


def test_restore_gradient_matches_autograd():
    m, p, x, e_w, e_z, _ = _setup(6, 2, 2, 1.0, seed=4)
    dx = m.restore_grads(p, x, e_w, e_z, 1.8)
    tp = {k: torch.tensor(v) for k, v in p.items()}
    xt = torch.tensor(x).permute(0, 3, 1, 2).requires_grad_(True)
    o = _graph(m, tp, xt, e_w, e_z, {}, tv=1.8)
    ref = torch.autograd.grad((o['loss'] + o['restore']).sum(), xt)[0].permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(dx, ref, rtol=1e-8, atol=1e-12)
    r = m.reconstruct(p, x, lambda s: (e_w, e_z), restore_steps=2, restore_lr=1e-3, tv_lambda=1.8)
    x1 = x - 1e-3 * dx
    x2 = x1 - 1e-3 * m.restore_grads(p, x1, e_w, e_z, 1.8)
    np.testing.assert_allclose(r['reconstruction'], x2, rtol=0, atol=1e-15)
    assert m.reconstruct(p, x[0], lambda s: (e_w[:1], e_z[:1]), restore_steps=0)['reconstruction'].shape == (1, 32, 32, 1)


def test_spec_names_and_order():
    names = [s[0] for s in ogd.param_spec(128, 8, 6, 1, 1)]
    i = names.index
    assert i('Bottleneck/conv2d/kernel') < i('Bottleneck/dense/kernel') < i('Bottleneck/dense_3/bias') < i('Bottleneck/dense_4/kernel') \
        < i('Bottleneck/conv2d_1/kernel') < i('dense/kernel') < i('dense_1/bias') < i('Variable') < i('Decoder/batch_normalization/gamma')
    shapes = {s[0]: s[1] for s in ogd.param_spec(128, 8, 6, 2, 3)}
    assert shapes['Bottleneck/dense_1/kernel'] == (1024, 3) and shapes['Bottleneck/dense_2/kernel'] == (1024, 2)
    assert shapes['Bottleneck/dense_4/kernel'] == (2, 1024) and shapes['dense/kernel'] == (3, 12) and shapes['Variable'] == (12,)
