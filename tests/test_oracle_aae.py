"""CPU: numpy oracle of the constrained / adversarial dense autoencoders (oracle/aae.py) vs an autograd graph written like the reference's
(models/constrained_autoencoder.py, adversarial_autoencoder.py, constrained_adversarial_autoencoder.py; trainers/ConstrainedAE.py,
AAE.py, ConstrainedAAE.py), fp64, incl. the second-order gradient of the latent WGAN-GP penalty."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import aae as oaae
from oracle import vae as ovae
from tests import torch_ref


def _graph(m, tp, x, z_prior, eps, mz, md, mr):
    n = x.shape[0]
    t = lambda a: None if a is None else torch.tensor(a)

    def bn(v, name):
        return v * (tp[name + '/gamma'] / math.sqrt(1.001)).view(1, -1, 1, 1) + tp[name + '/beta'].view(1, -1, 1, 1)

    def encode(a, mask):
        for i in range(m.npool):
            a = F.leaky_relu(bn(torch_ref._conv_same(a, tp['Encoder/enc_conv2D_%d/kernel' % i], tp['Encoder/enc_conv2D_%d/bias' % i], 2), m.bn_e[i]), 0.3)
        tt = torch_ref._conv_same(a, tp[m.nm['conv'] + '/kernel'], tp[m.nm['conv'] + '/bias'], 1)
        z = tt.permute(0, 2, 3, 1).reshape(n, -1) @ tp[m.nm['z'] + '/kernel'] + tp[m.nm['z'] + '/bias']
        return z if mask is None else z * t(mask)

    def decode(z):
        dv = z @ tp[m.nm['dec'] + '/kernel'] + tp[m.nm['dec'] + '/bias']
        if md is not None:
            dv = dv * t(md)
        a = dv.reshape(n, m.inter_res, m.inter_res, -1).permute(0, 3, 1, 2)
        a = F.relu(bn(torch_ref._conv_same(a, tp[m.nm['rev'] + '/kernel'], tp[m.nm['rev'] + '/bias'], 1), 'Decoder/batch_normalization'))
        for i in range(m.npool):
            a = F.leaky_relu(bn(torch_ref._convT_same(a, tp['Decoder/dec_Conv2DT_%d/kernel' % i], tp['Decoder/dec_Conv2DT_%d/bias' % i], 2),
                                'Decoder/batch_normalization_%d' % (i + 1)), 0.3)
        return torch_ref._conv_same(a, tp['Decoder/dec_Conv2D_final/kernel'], tp['Decoder/dec_Conv2D_final/bias'], 1)

    def critic(v):
        h = F.leaky_relu(v @ tp['Discriminator/dense/kernel'] + tp['Discriminator/dense/bias'], 0.2)
        h = F.leaky_relu(h @ tp['Discriminator/dense_1/kernel'] + tp['Discriminator/dense_1/bias'], 0.2)
        return h @ tp['Discriminator/dense_2/kernel'] + tp['Discriminator/dense_2/bias']

    xt = torch.tensor(x).permute(0, 3, 1, 2)
    o = {}
    o['z'] = z = encode(xt, mz)
    xh = decode(z)
    o['x_hat'] = xh.permute(0, 2, 3, 1)
    l2 = ((xt - xh) ** 2).mean(dim=(1, 2, 3))
    o['reconstructionLoss'] = (xt - xh).abs().sum(dim=(1, 2, 3)).mean()
    if m.constrained:
        z_rec = encode(xh, mr)
        o['loss'] = (l2 + m.rho * ((z - z_rec) ** 2).mean(dim=1)).mean()
    else:
        o['loss'] = l2.mean()
    if m.has_critic:
        zp = torch.tensor(z_prior)
        d_, d = critic(z), critic(zp)
        z_hat = zp + torch.tensor(eps).view(n, 1) * (zp - z)
        ddz = torch.autograd.grad(critic(z_hat).sum(), z_hat, create_graph=True)[0]
        o['penalty'] = ((torch.sqrt((ddz ** 2).sum(dim=1)) - 1.0) ** 2 * m.scale).mean()
        o['disc_fake'], o['disc_real'] = d_.mean(), d.mean()
        o['disc_loss'] = d_.mean() - d.mean() + o['penalty']
        o['gen_loss'] = -d_.mean()
    return o


@pytest.mark.parametrize('kind,drop', [('constrained_ae', True), ('aae', True), ('constrained_aae', False), ('aae', False)])
def test_aae_family_vs_torch(kind, drop):
    h, inter, zdim, n = 32, 8, 16, 3
    m = oaae.AAE(kind, h, inter, zdim, rho=0.8, scale=10.0)
    p = ovae.init_params(m.spec, seed=12, dtype=np.float64, perturb=True)
    rng = np.random.default_rng(4)
    x = ovae.synthetic_slices(n, h, h, seed=3, dtype=np.float64)
    z_prior = rng.standard_normal((n, zdim)); eps = rng.uniform(0, 1, n)
    flat = inter * inter * 8
    keep = lambda shape: (rng.random(shape) > 0.2) / 0.8
    mz = keep((n, zdim)) if drop else None
    md = keep((n, flat)) if drop else None
    mr = keep((n, zdim)) if (drop and m.constrained) else None
    tp = torch_ref.to_torch(p)
    o = _graph(m, tp, x, z_prior, eps, mz, md, mr)

    def tgrads(loss, names):
        gs = torch.autograd.grad(loss, [tp[k] for k in names], retain_graph=True, allow_unused=True)
        return {k: (np.zeros(p[k].shape) if gg is None else gg.numpy()) for k, gg in zip(names, gs)}

    def check(mine, ref, tag):
        gmax = max(np.abs(v).max() for v in ref.values())
        for k, r in ref.items():
            a = np.asarray(mine.get(k, np.zeros(p[k].shape))).reshape(p[k].shape)
            np.testing.assert_allclose(a, r, rtol=2e-7, atol=1e-9 * gmax + 1e-8 * np.abs(r).max(), err_msg=tag + ':' + k)

    ae_names = [k for k, _, _ in m.spec if not k.startswith('Discriminator')]
    ls, g = m.ae_phase(p, x, mz, md, mr)
    np.testing.assert_allclose(ls['loss'], o['loss'].item(), rtol=1e-10)
    np.testing.assert_allclose(ls['reconstructionLoss'], o['reconstructionLoss'].item(), rtol=1e-10)
    np.testing.assert_allclose(ls['reconstruction'], o['x_hat'].detach().numpy(), rtol=1e-9, atol=1e-12)
    check(g, tgrads(o['loss'], ae_names), 'ae')
    if m.has_critic:
        d_names = [k for k, _, _ in m.spec if k.startswith('Discriminator')]
        ls, g = m.disc_phase(p, x, z_prior, eps, mz)
        for k in ('disc_fake', 'disc_real', 'penalty', 'disc_loss'):
            np.testing.assert_allclose(ls[k], o[k].item(), rtol=1e-10, err_msg=k)
        check(g, tgrads(o['disc_loss'], d_names), 'disc')
        e_names = [k for k, _, _ in m.spec if 'Encoder' in k]
        ls, g = m.gen_phase(p, x, mz)
        np.testing.assert_allclose(ls['gen_loss'], o['gen_loss'].item(), rtol=1e-10)
        check(g, tgrads(o['gen_loss'], e_names), 'gen')
        assert set(g) == set(e_names)


def test_aae_param_tables():
    s, nm = oaae.param_spec('aae', 128, 8, 128)
    names = [k for k, _, _ in s]
    assert names.index('Bottleneck/conv2d/kernel') < names.index('Bottleneck/dense/kernel') < names.index('Bottleneck/dense_1/kernel') < names.index('Bottleneck/conv2d_1/kernel')
    assert dict((k, sh) for k, sh, _ in s)['Discriminator/dense/kernel'] == (128, 50) and names[-1] == 'Discriminator/dense_2/bias'
    s, nm = oaae.param_spec('constrained_aae', 128, 8, 128)
    shp = dict((k, sh) for k, sh, _ in s)
    assert nm == {'conv': 'Encoder/conv2d', 'z': 'Encoder/dense', 'dec': 'Decoder/dense', 'rev': 'Decoder/conv2d_1'}
    assert shp['Discriminator/dense/kernel'] == (128, 100) and shp['Discriminator/dense_1/kernel'] == (100, 50)
    assert sum('Encoder' in k for k, _, _ in s) == 4 * 4 + 4      # encoder blocks + conv2d + dense: what optim_gen trains there
    s, _ = oaae.param_spec('constrained_ae', 128, 8, 128)
    assert not any(k.startswith('Discriminator') for k, _, _ in s)
