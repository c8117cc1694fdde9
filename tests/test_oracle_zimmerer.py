"""CPU: numpy oracle of the Zimmerer VAE (oracle/zimmerer.py) vs an autograd graph written like the reference's
(models/variational_autoencoder_Zimmerer.py + trainers/VAE.py:36-42), fp64: reconstruction, losses, every parameter gradient."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import zimmerer as oz
from tests import torch_ref


def test_zimmerer_vae_matches_autograd():
    m = oz.VAEZimmerer(32, 8)
    p = oz.init_params(m.spec, seed=1)
    rng = np.random.default_rng(0)
    x = rng.random((2, 32, 32, 1))
    eps = rng.standard_normal((2, 8))
    out, cache = m.forward(p, x, eps)
    ls = m.losses(x, out)
    g = m.backward(p, x, out, cache)
    tp = {k: torch.tensor(v, requires_grad=True) for k, v in p.items()}
    a = torch.tensor(x).permute(0, 3, 1, 2)
    xt = a
    for i in range(1, 5):
        a = F.leaky_relu(torch_ref._conv_same(a, tp[f'enc_conv2D_{i}/kernel'], tp[f'enc_conv2D_{i}/bias'], 2), 0.2)
    flat = a.permute(0, 2, 3, 1).reshape(2, -1)
    mu = flat @ tp['dense/kernel'] + tp['dense/bias']
    lsg = flat @ tp['dense_1/kernel'] + tp['dense_1/bias']
    sigma = torch.exp(lsg)
    z = mu + torch.tensor(eps) * sigma
    a = (z @ tp['dense_2/kernel'] + tp['dense_2/bias']).reshape(2, 2, 2, 1024).permute(0, 3, 1, 2)
    for i in range(1, 5):
        a = F.leaky_relu(torch_ref._convT_same(a, tp[f'dec_Conv2DT_{i}/kernel'], tp[f'dec_Conv2DT_{i}/bias'], 2), 0.2)
    xh = torch_ref._conv_same(a, tp['dec_Conv2D_final/kernel'], tp['dec_Conv2D_final/bias'], 1)
    rec = (xt - xh).abs().sum(dim=(1, 2, 3))
    kl = 0.5 * (mu ** 2 + sigma ** 2 - torch.log(sigma ** 2) - 1).sum(dim=1)
    loss = (rec + kl).mean()
    np.testing.assert_allclose(out['x_hat'], xh.permute(0, 2, 3, 1).detach().numpy(), rtol=1e-9, atol=1e-11)
    assert ls['loss'] == pytest.approx(float(loss.detach()), rel=1e-10)
    assert ls['kl'] == pytest.approx(float(kl.mean().detach()), rel=1e-10)
    names = [s[0] for s in m.spec]
    grads = torch.autograd.grad(loss, [tp[k] for k in names])
    for k, tg in zip(names, grads):
        ref = tg.numpy()
        assert np.abs(g[k] - ref).max() <= 1e-9 * max(np.abs(ref).max(), 1e-6), k


def test_zimmerer_spec():
    spec = oz.param_spec(128, 128)
    names = [s[0] for s in spec]
    assert names[:2] == ['enc_conv2D_1/kernel', 'enc_conv2D_1/bias'] and names[8:14] == ['dense/kernel', 'dense/bias', 'dense_1/kernel', 'dense_1/bias',
                                                                                    'dense_2/kernel', 'dense_2/bias']
    shapes = dict((s[0], s[1]) for s in spec)
    assert shapes['enc_conv2D_4/kernel'] == (4, 4, 256, 1024) and shapes['dense/kernel'] == (65536, 128)
    assert shapes['dec_Conv2DT_1/kernel'] == (4, 4, 1024, 1024) and shapes['dec_Conv2D_final/kernel'] == (4, 4, 16, 1)


def test_zimmerer_cevae_matches_autograd():
    """models/context_encoder_variational_autoencoder_Zimmerer.py + trainers/ceVAE.py:38-51: both branches through the shared layers,
    loss = mean(rec_vae + kl + rec_ce), anomaly = L1_vae * |d loss_vae / d x|."""
    m = oz.CeVAEZimmerer(32, 8)
    p = oz.init_params(m.spec, seed=2)
    assert all(k.startswith(('Encoder/', 'Bottleneck/', 'Decoder/')) for k in p)
    rng = np.random.default_rng(5)
    x = rng.random((2, 32, 32, 1))
    x_ce = x.copy(); x_ce[:, 8:20, 10:22] = 0
    eps = rng.standard_normal((2, 8))
    out, caches = m.ce_forward(p, x, x_ce, eps)
    ls = m.ce_losses(x, x_ce, out)
    g = m.ce_backward(p, x, x_ce, out, caches)
    tp = {k: torch.tensor(v, requires_grad=True) for k, v in p.items()}

    def branch(xt, sample):
        a = xt
        for i in range(1, 5):
            a = F.leaky_relu(torch_ref._conv_same(a, tp[f'Encoder/enc_conv2D_{i}/kernel'], tp[f'Encoder/enc_conv2D_{i}/bias'], 2), 0.2)
        flat = a.permute(0, 2, 3, 1).reshape(2, -1)
        mu = flat @ tp['Bottleneck/dense/kernel'] + tp['Bottleneck/dense/bias']
        sigma = torch.exp(flat @ tp['Bottleneck/dense_1/kernel'] + tp['Bottleneck/dense_1/bias'])
        z = mu + torch.tensor(eps) * sigma if sample else mu
        a = (z @ tp['Bottleneck/dense_2/kernel'] + tp['Bottleneck/dense_2/bias']).reshape(2, 2, 2, 1024).permute(0, 3, 1, 2)
        for i in range(1, 5):
            a = F.leaky_relu(torch_ref._convT_same(a, tp[f'Decoder/dec_Conv2DT_{i}/kernel'], tp[f'Decoder/dec_Conv2DT_{i}/bias'], 2), 0.2)
        return torch_ref._conv_same(a, tp['Decoder/dec_Conv2D_final/kernel'], tp['Decoder/dec_Conv2D_final/bias'], 1), mu, sigma
    xt = torch.tensor(x).permute(0, 3, 1, 2).requires_grad_(True)
    xc = torch.tensor(x_ce).permute(0, 3, 1, 2)
    xh, mu, sigma = branch(xt, True)
    xhc, _, _ = branch(xc, False)
    rv, rc = (xt - xh).abs().sum(dim=(1, 2, 3)), (xc - xhc).abs().sum(dim=(1, 2, 3))
    kl = 0.5 * (mu ** 2 + sigma ** 2 - torch.log(sigma ** 2) - 1).sum(dim=1)
    loss, loss_vae = (rv + kl + rc).mean(), (rv + kl).mean()
    for k, v in (('loss', loss), ('loss_vae', loss_vae), ('Rec_ce', rc.mean()), ('reconstructionLoss', 0.5 * (rv + rc).mean())):
        assert ls[k] == pytest.approx(float(v.detach()), rel=1e-10), k
    names = [s[0] for s in m.spec]
    grads = torch.autograd.grad(loss, [tp[k] for k in names], retain_graph=True)
    for k, tg in zip(names, grads):
        assert np.abs(g[k] - tg.numpy()).max() <= 1e-9 * max(np.abs(tg.numpy()).max(), 1e-6), k
    dx = torch.autograd.grad(loss_vae, xt)[0]
    anomaly = ((xt - xh).abs() * dx.abs()).permute(0, 2, 3, 1).detach().numpy()
    np.testing.assert_allclose(g['__anomaly'], anomaly, rtol=1e-8, atol=1e-14)
