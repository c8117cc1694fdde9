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
