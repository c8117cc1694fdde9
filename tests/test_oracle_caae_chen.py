"""CPU: numpy oracle of the residual-block constrained adversarial autoencoder (oracle/caae_chen.py) vs an autograd graph written like the
reference's (models/constrained_adversarial_autoencoder_Chen.py + trainers/ConstrainedAAE.py:44-70), fp64: the autoencoder phase (incl. the
re-encoding constraint through the shared encoder), the critic phase (second-order penalty, scalar eps) and the generator phase."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import caae_chen as oc
from oracle import vae as ovae
from tests import torch_ref


def _graph(m, tp, x, z_prior, eps):
    n = x.shape[0]
    conv = lambda a, name, s: torch_ref._conv_same(a, tp[name + '/kernel'], tp[name + '/bias'], s)
    convT = lambda a, name, s: torch_ref._convT_same(a, tp[name + '/kernel'], tp[name + '/bias'], s)
    ln = lambda a, name: torch_ref._ln_hw(a, tp[name + '/gamma'], tp[name + '/beta'])

    def encode(a):
        out = conv(a, m.nm['enc_conv'], 1)
        for b in m.be:
            t = conv(F.relu(ln(conv(F.relu(ln(out, b.n['ln1'])), b.n['conv1'], 1), b.n['ln2'])), b.n['conv2'], b.stride)
            out = t + (out if b.n['short'] is None else F.avg_pool2d(conv(out, b.n['short'], 1), 2))
        return out.permute(0, 2, 3, 1).reshape(n, -1) @ tp['Encoder/dense/kernel'] + tp['Encoder/dense/bias']

    def decode(z):
        out = (z @ tp['Decoder/dense/kernel'] + tp['Decoder/dense/bias']).reshape(n, m.inter_res, m.inter_res, -1).permute(0, 3, 1, 2)
        for b in m.bd:
            t = convT(F.relu(ln(conv(F.relu(ln(out, b.n['ln1'])), b.n['conv1'], 1), b.n['ln2'])), b.n['conv2'], b.stride)
            out = t + (out if b.n['short'] is None else convT(out, b.n['short'], 2))
        return conv(F.relu(ln(out, m.nm['dec_ln'])), m.nm['dec_final'], 1)

    def critic(v):
        h = F.leaky_relu(v @ tp['Discriminator/dense/kernel'] + tp['Discriminator/dense/bias'], 0.2)
        h = F.leaky_relu(h @ tp['Discriminator/dense_1/kernel'] + tp['Discriminator/dense_1/bias'], 0.2)
        return h @ tp['Discriminator/dense_2/kernel'] + tp['Discriminator/dense_2/bias']
    xt = torch.tensor(x).permute(0, 3, 1, 2)
    o = {}
    o['z'] = z_ = encode(xt)
    xh = decode(z_)
    o['x_hat'] = xh.permute(0, 2, 3, 1)
    z_rec = encode(xh)
    l2 = ((xt - xh) ** 2).mean(dim=(1, 2, 3))
    o['loss'] = (l2 + m.rho * ((z_rec - z_) ** 2).mean(dim=1)).mean()
    zp = torch.tensor(z_prior)
    d_, d = critic(z_), critic(zp)
    z_hat = eps * zp + (1 - eps) * z_
    ddx = torch.autograd.grad(critic(z_hat).sum(), z_hat, create_graph=True)[0]
    o['penalty'] = ((torch.sqrt((ddx ** 2).sum(dim=1)) - 1.0) ** 2 * m.scale).mean()
    o['disc_loss'] = d_.mean() - d.mean() + o['penalty']
    o['gen_loss'] = -d_.mean()
    return o


def test_caae_chen_matches_autograd():
    m = oc.CAAEChen(16, 8, dim=32, rho=0.7)
    from oracle import gmvae as og
    p = og.init_params(m.spec, seed=4, dtype=np.float64, perturb=True)
    rng = np.random.default_rng(2)
    n = 2
    x = ovae.synthetic_slices(n, 16, 16, seed=1).astype(np.float64)
    z_prior = rng.standard_normal((n, 8))
    eps = 0.37
    tp = {k: torch.tensor(v, requires_grad=True) for k, v in p.items()}
    o = _graph(m, tp, x, z_prior, eps)
    names = [s[0] for s in m.spec]

    def check(loss_t, g, only=None):
        sel = [k for k in names if only is None or only(k)]
        grads = torch.autograd.grad(loss_t, [tp[k] for k in sel], retain_graph=True, allow_unused=True)
        for k, tg in zip(sel, grads):
            ref = np.zeros_like(p[k]) if tg is None else tg.numpy()
            got = g.get(k, np.zeros_like(p[k]))
            assert np.abs(got - ref).max() <= 1e-8 * max(np.abs(ref).max(), 1e-6) + 1e-12, k       # conv biases in front of a LayerNorm-HW: exact 0 up to round-off
    ls, g = m.ae_phase(p, x)
    np.testing.assert_allclose(ls['reconstruction'], o['x_hat'].detach().numpy(), rtol=1e-8, atol=1e-11)
    assert ls['loss'] == pytest.approx(float(o['loss'].detach()), rel=1e-10)
    check(o['loss'], g, only=lambda k: not k.startswith('Discriminator'))
    ls, g = m.disc_phase(p, x, z_prior, np.full(n, eps))
    assert ls['penalty'] == pytest.approx(float(o['penalty'].detach()), rel=1e-9)
    assert ls['disc_loss'] == pytest.approx(float(o['disc_loss'].detach()), rel=1e-9)
    check(o['disc_loss'], g, only=lambda k: k.startswith('Discriminator'))
    ls, g = m.gen_phase(p, x)
    assert ls['gen_loss'] == pytest.approx(float(o['gen_loss'].detach()), rel=1e-10)
    check(o['gen_loss'], g, only=lambda k: k.startswith('Encoder'))


def test_caae_chen_spec():
    spec, enc, dec, nm = oc.param_spec(64, 128, 64)
    names = [s[0] for s in spec]
    assert names[0] == 'Encoder/conv2d/kernel' and nm['dec_final'] == 'Decoder/conv2d_4' and nm['dec_ln'] == 'Decoder/layer_normalization_16'
    assert enc[0].n == dict(ln1='Encoder/layer_normalization', conv1='Encoder/conv2d_1', ln2='Encoder/layer_normalization_1', conv2='Encoder/conv2d_2',
                            short='Encoder/conv2d_3')
    assert enc[3].n['short'] is None and enc[3].n['conv2'] == 'Encoder/conv2d_11'
    assert dec[0].n == dict(ln1='Decoder/layer_normalization_8', conv1='Decoder/conv2d', ln2='Decoder/layer_normalization_9',
                            conv2='Decoder/conv2d_transpose', short=None)
    assert dec[1].n['short'] == 'Decoder/conv2d_transpose_2'
    sh = dict((s[0], s[1]) for s in spec)
    assert sh['Encoder/dense/kernel'] == (8 * 8 * 512, 128) and sh['Discriminator/dense_1/kernel'] == (400, 200)
