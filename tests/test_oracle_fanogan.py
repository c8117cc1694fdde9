"""CPU: numpy f-AnoGAN oracle (hand-written backward, incl. the second-order pass of the WGAN-GP penalty through
LayerNorm-HW) vs an autograd graph written like the reference's (fp64 round-off agreement)."""
import numpy as np
import pytest
import torch

from oracle import fanogan as ofa
from oracle import vae as ovae
from tests import torch_ref


def _setup(h, inter, zdim, n, seed=0, dtype=np.float64):
    m = ofa.FAnoGAN(h, inter, zdim, scale=10.0, kappa=1.3)
    p = ovae.init_params(m.spec, seed=11 + seed, dtype=dtype, perturb=True)
    rng = np.random.default_rng(70 + seed)
    x = ovae.synthetic_slices(n, h, h, seed=seed, dtype=dtype)
    z = rng.standard_normal((n, zdim)).astype(dtype)
    alpha = rng.uniform(0, 1, (n, 1)).astype(dtype)
    return m, p, x, z, alpha, rng


def test_ln_second_order_vs_autograd():
    rng = np.random.default_rng(0)
    c = rng.standard_normal((2, 6, 6, 3)); v = rng.standard_normal(c.shape); q = rng.standard_normal(c.shape)
    gamma = 1 + 0.2 * rng.standard_normal((6, 6)); beta = rng.standard_normal((6, 6))
    y, cache = ofa.ln_fwd(c, gamma, beta)
    dc, dg, db = ofa.ln_bwd(v, gamma, cache)
    vbar, gbar, cbar = ofa.ln_bwd2(q, v, gamma, cache)
    tc, tv, tg = (torch.tensor(a, requires_grad=True) for a in (c, v, gamma))
    tb = torch.tensor(beta, requires_grad=True)
    ty = torch_ref._ln_hw(tc.permute(0, 3, 1, 2), tg, tb).permute(0, 2, 3, 1)
    np.testing.assert_allclose(y, ty.detach().numpy(), rtol=1e-12, atol=1e-13)
    tdc, tdg, tdb = torch.autograd.grad(ty, (tc, tg, tb), tv, create_graph=True)
    np.testing.assert_allclose(dc, tdc.detach().numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(dg, tdg.detach().numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(db, tdb.detach().numpy(), rtol=1e-10, atol=1e-12)
    rv, rg, rc = torch.autograd.grad((tdc * torch.tensor(q)).sum(), (tv, tg, tc))
    np.testing.assert_allclose(vbar, rv.numpy(), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(gbar, rg.numpy(), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(cbar, rc.numpy(), rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize('h,inter,zdim,n,drop', [(32, 8, 16, 2, False), (32, 4, 8, 3, True), (64, 8, 16, 1, False)])
def test_fanogan_phases_vs_torch(h, inter, zdim, n, drop):
    m, p, x, z, alpha, rng = _setup(h, inter, zdim, n)
    flat = inter * inter * (min(128, 32 * 2 ** (m.npool - 1)) // 8)
    mz = mg = mge = None
    if drop:
        mz = (rng.random((n, zdim)) > 0.2) / 0.8; mg = (rng.random((n, flat)) > 0.2) / 0.8; mge = (rng.random((n, flat)) > 0.2) / 0.8
    tp = torch_ref.to_torch(p)
    tt = lambda a: None if a is None else torch.tensor(a)
    o = torch_ref.fanogan_graph(tp, torch.tensor(x), torch.tensor(z), torch.tensor(alpha), m.npool, inter, m.scale, m.kappa,
                                tt(mz), tt(mg), tt(mge))
    groups = {g: [k for k, _, _ in m.spec if ofa.group_of(k) == g] for g in ('Encoder', 'Generator', 'Discriminator')}

    def tgrads(loss, group):
        gs = torch.autograd.grad(loss, [tp[k] for k in groups[group]], retain_graph=True, allow_unused=True)
        return {k: (np.zeros(p[k].shape) if g is None else g.numpy()) for k, g in zip(groups[group], gs)}

    def check(mine, ref, group, tag):
        for k in groups[group]:
            a = np.asarray(mine.get(k, np.zeros(p[k].shape))).reshape(p[k].shape)
            np.testing.assert_allclose(a, ref[k], rtol=2e-7, atol=1e-11 + 1e-8 * np.abs(ref[k]).max(), err_msg=tag + ':' + k)

    # generator phase (trainers/fAnoGAN.py:75, 100-113)
    ls, g = m.gen_phase(p, z, mg)
    np.testing.assert_allclose(ls['gen_loss'], o['gen_loss'].item(), rtol=1e-11)
    np.testing.assert_allclose(ls['generated'], o['x_'].detach().numpy(), rtol=1e-10, atol=1e-13)
    check(g, tgrads(o['gen_loss'], 'Generator'), 'Generator', 'gen')
    # critic phase (:50-58, 74, 115-130)
    ls, g = m.disc_phase(p, x, z, alpha, mg)
    for k in ('disc_fake', 'disc_real', 'disc_loss', 'penalty'):
        np.testing.assert_allclose(ls[k], o[k].item(), rtol=1e-10, err_msg=k)
    check(g, tgrads(o['disc_loss'], 'Discriminator'), 'Discriminator', 'disc')
    # encoder phase (:60-66, 76, 146-166)
    ls, g = m.enc_phase(p, x, mz, mge)
    for k in ('loss_img', 'loss_fts', 'enc_loss', 'reconstructionLoss'):
        np.testing.assert_allclose(ls[k], o[k].item(), rtol=1e-10, err_msg=k)
    np.testing.assert_allclose(ls['z_enc'], o['z_enc'].detach().numpy(), rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(ls['reconstruction'], o['x_enc'].detach().numpy(), rtol=1e-10, atol=1e-13)
    check(g, tgrads(o['enc_loss'], 'Encoder'), 'Encoder', 'enc')


def test_fanogan_param_table():
    m = ofa.FAnoGAN(128, 8, 128)
    names = [s[0] for s in m.spec]
    assert m.npool == 4 and len(m.ln_g) == 5 and len(m.ln_d) == 4
    assert names[0] == 'Encoder/enc_conv2D_0/kernel' and 'Encoder/dense/kernel' in names
    assert names.index('Generator/dense/kernel') < names.index('Generator/conv2d_1/kernel') < names.index('Generator/layer_normalization/gamma')
    assert m.ln_d[0] == 'Discriminator/layer_normalization_5' and names[-2:] == ['Discriminator/dense/kernel', 'Discriminator/dense/bias']
    shp = dict((s[0], s[1]) for s in m.spec)
    assert shp['Generator/layer_normalization_4/gamma'] == (128, 128) and shp['Discriminator/layer_normalization_8/gamma'] == (8, 8)
    assert shp['Discriminator/dense/kernel'] == (128, 1)


def test_fanogan_adam_groups():
    m, p, x, z, alpha, _ = _setup(32, 8, 16, 2, dtype=np.float32)
    opt = m.new_opt(p)
    before = {k: v.copy() for k, v in p.items()}
    _, g = m.disc_phase(p, x, z, alpha)
    m.apply(p, opt, g, 'Discriminator', 1e-3)
    for k in p:
        changed = not np.array_equal(before[k], p[k])
        # the critic's dense bias sees +1/size from the fake and -1/size from the real half: exactly zero
        assert changed == (ofa.group_of(k) == 'Discriminator' and k in g and np.any(g[k] != 0)), k
    assert opt['t'] == {'Encoder': 0, 'Generator': 0, 'Discriminator': 1}


# ------------------------------------------------------------------ ResNet graph (models/fanogan_schlegl.py)
@pytest.mark.parametrize('h,dim,zdim,n', [(32, 4, 8, 2), (64, 2, 8, 1)])
def test_fanogan_schlegl_phases_vs_torch(h, dim, zdim, n):
    from oracle import fanogan_schlegl as ofs
    inter = h // 8
    m = ofs.FAnoGANSchlegl(h, inter, zdim, dim, scale=10.0, kappa=0.7)
    p = ovae.init_params(m.spec, seed=5, dtype=np.float64, perturb=True)
    rng = np.random.default_rng(8)
    x = ovae.synthetic_slices(n, h, h, seed=1, dtype=np.float64)
    z = rng.standard_normal((n, zdim)); alpha = rng.uniform(0, 1, (n, 1))
    tp = torch_ref.to_torch(p)
    o = torch_ref.fanogan_schlegl_graph(tp, m.bg, m.bd, m.nm, torch.tensor(x), torch.tensor(z), torch.tensor(alpha), inter, m.scale, m.kappa)
    groups = {g: [k for k, _, _ in m.spec if ofa.group_of(k) == g] for g in ('Encoder', 'Generator', 'Discriminator')}

    def tgrads(loss, group):
        gs = torch.autograd.grad(loss, [tp[k] for k in groups[group]], retain_graph=True, allow_unused=True)
        return {k: (np.zeros(p[k].shape) if g is None else g.numpy()) for k, g in zip(groups[group], gs)}

    def check(mine, ref, group, tag):
        # residual-stream biases shift every critic score by the same constant: their disc_loss gradient is round-off only
        gmax = max(np.abs(ref[k]).max() for k in groups[group])
        for k in groups[group]:
            a = np.asarray(mine.get(k, np.zeros(p[k].shape))).reshape(p[k].shape)
            np.testing.assert_allclose(a, ref[k], rtol=2e-7, atol=1e-9 * gmax + 1e-8 * np.abs(ref[k]).max(), err_msg=tag + ':' + k)

    ls, g = m.gen_phase(p, z)
    np.testing.assert_allclose(ls['gen_loss'], o['gen_loss'].item(), rtol=1e-10)
    np.testing.assert_allclose(ls['generated'], o['x_'].detach().numpy(), rtol=1e-9, atol=1e-12)
    check(g, tgrads(o['gen_loss'], 'Generator'), 'Generator', 'gen')
    ls, g = m.disc_phase(p, x, z, alpha)
    for k in ('disc_fake', 'disc_real', 'disc_loss', 'penalty'):
        np.testing.assert_allclose(ls[k], o[k].item(), rtol=1e-9, err_msg=k)
    check(g, tgrads(o['disc_loss'], 'Discriminator'), 'Discriminator', 'disc')
    ls, g = m.enc_phase(p, x)
    for k in ('loss_img', 'loss_fts', 'enc_loss', 'reconstructionLoss'):
        np.testing.assert_allclose(ls[k], o[k].item(), rtol=1e-9, err_msg=k)
    np.testing.assert_allclose(ls['reconstruction'], o['x_enc'].detach().numpy(), rtol=1e-9, atol=1e-12)
    check(g, tgrads(o['enc_loss'], 'Encoder'), 'Encoder', 'enc')


def test_fanogan_schlegl_param_table():
    from oracle import fanogan_schlegl as ofs
    m = ofs.FAnoGANSchlegl(64, 8, 128, 64)
    names = [s[0] for s in m.spec]
    shp = dict((s[0], s[1]) for s in m.spec)
    assert names.index('Generator/dense/kernel') < names.index('Generator/layer_normalization/gamma') < names.index('Generator/conv2d/kernel')
    assert shp['Generator/dense/kernel'] == (128, 8 * 8 * 512) and shp['Generator/conv2d/kernel'] == (3, 3, 512, 512)
    assert shp['Generator/conv2d_transpose_2/kernel'] == (1, 1, 256, 512)          # res2 shortcut: k1 s2 ConvT 512 -> 256
    assert shp['Generator/conv2d_4/kernel'] == (1, 1, 64, 1) and shp['Generator/layer_normalization_8/gamma'] == (64, 64)
    assert shp['Discriminator/conv2d/kernel'] == (3, 3, 1, 64) and shp['Discriminator/layer_normalization_9/gamma'] == (64, 64)
    assert shp['Discriminator/conv2d_3/kernel'] == (1, 1, 64, 128) and names[-2:] == ['Discriminator/dense/kernel', 'Discriminator/dense/bias']
    n_g = sum(int(np.prod(s)) for k, s, _ in m.spec if k.startswith('Generator'))
    n_d = sum(int(np.prod(s)) for k, s, _ in m.spec if k.startswith('Discriminator'))
    assert 11.0e6 < n_g < 12.0e6 and 9.0e6 < n_d < 10.0e6                            # ~11.4 M / ~9.5 M


# ------------------------------------------------------------------ AnoVAE-GAN (models/anovaegan.py, trainers/AnoVAEGAN.py)
@pytest.mark.parametrize('h,inter,zdim,n,drop', [(32, 8, 16, 2, True), (64, 8, 8, 1, False)])
def test_anovaegan_phases_vs_torch(h, inter, zdim, n, drop):
    import math
    import torch.nn.functional as F
    m = ofa.AnoVAEGAN(h, inter, zdim, scale=10.0, kl_weight=0.7)
    p = ovae.init_params(m.spec, seed=9, dtype=np.float64, perturb=True)
    rng = np.random.default_rng(3)
    x = ovae.synthetic_slices(n, h, h, seed=2, dtype=np.float64)
    eps = rng.standard_normal((n, zdim)); alpha = rng.uniform(0, 1, (n, 1))
    mm = (rng.random((n, zdim)) > 0.2) / 0.8 if drop else None
    ms = (rng.random((n, zdim)) > 0.2) / 0.8 if drop else None
    tp = torch_ref.to_torch(p)
    xt = torch.tensor(x).permute(0, 3, 1, 2)
    ln_g = [k for k in m.ln_g]; ln_d = [k for k in m.ln_d]

    a = xt
    for i in range(m.npool):
        a = torch_ref._conv_same(a, tp['Encoder/enc_conv2D_%d/kernel' % i], tp['Encoder/enc_conv2D_%d/bias' % i], 2)
        a = F.leaky_relu(a * (tp[m.bn_e[i] + '/gamma'] / math.sqrt(1.001)).view(1, -1, 1, 1) + tp[m.bn_e[i] + '/beta'].view(1, -1, 1, 1), 0.3)
    t = torch_ref._conv_same(a, tp['Encoder/conv2d/kernel'], tp['Encoder/conv2d/bias'], 1)
    flat = t.permute(0, 2, 3, 1).reshape(n, -1)
    mu = flat @ tp['Encoder/dense/kernel'] + tp['Encoder/dense/bias']
    ls = flat @ tp['Encoder/dense_1/kernel'] + tp['Encoder/dense_1/bias']
    if drop:
        mu = mu * torch.tensor(mm); ls = ls * torch.tensor(ms)
    sg = torch.exp(ls)
    z = mu + torch.tensor(eps) * sg
    dv = z @ tp['Generator/dense/kernel'] + tp['Generator/dense/bias']
    g = dv.reshape(n, inter, inter, -1).permute(0, 3, 1, 2)
    g = torch_ref._conv_same(g, tp['Generator/conv2d_1/kernel'], tp['Generator/conv2d_1/bias'], 1)
    g = F.relu(torch_ref._ln_hw(g, tp[ln_g[0] + '/gamma'], tp[ln_g[0] + '/beta']))
    for i in range(m.npool):
        g = torch_ref._convT_same(g, tp['Generator/dec_Conv2DT_%d/kernel' % i], tp['Generator/dec_Conv2DT_%d/bias' % i], 2)
        g = F.leaky_relu(torch_ref._ln_hw(g, tp[ln_g[i + 1] + '/gamma'], tp[ln_g[i + 1] + '/beta']), 0.3)
    out = torch_ref._conv_same(g, tp['Generator/dec_Conv2D_final/kernel'], tp['Generator/dec_Conv2D_final/bias'], 1)

    def critic(v):
        for i in range(m.npool):
            v = torch_ref._conv_same(v, tp['Discriminator/enc_conv2D_%d/kernel' % i], tp['Discriminator/enc_conv2D_%d/bias' % i], 2)
            v = F.leaky_relu(torch_ref._ln_hw(v, tp[ln_d[i] + '/gamma'], tp[ln_d[i] + '/beta']), 0.3)
        return v.permute(0, 2, 3, 1) @ tp['Discriminator/dense/kernel'] + tp['Discriminator/dense/bias']

    d_fake, d_real = critic(out), critic(xt)
    x_hat = xt + torch.tensor(alpha).view(n, 1, 1, 1) * (out - xt)
    ddx = torch.autograd.grad(critic(x_hat).sum(), x_hat, create_graph=True)[0].permute(0, 2, 3, 1)
    pen = ((torch.sqrt((ddx ** 2).sum(dim=1)) - 1.0) ** 2).mean() * m.scale
    disc_loss = d_fake.mean() - d_real.mean() + pen
    gen_loss = -d_fake.mean()
    rec = (xt - out).abs().sum(dim=(1, 2, 3)).mean()
    kl = (0.5 * (mu ** 2 + sg ** 2 - torch.log(sg ** 2) - 1).sum(dim=1)).mean()
    enc_loss = rec + m.kl_weight * kl
    groups = {gname: [k for k, _, _ in m.spec if ofa.group_of(k) == gname] for gname in ('Encoder', 'Generator', 'Discriminator')}

    def tgrads(loss, names):
        gs = torch.autograd.grad(loss, [tp[k] for k in names], retain_graph=True, allow_unused=True)
        return {k: (np.zeros(p[k].shape) if gg is None else gg.numpy()) for k, gg in zip(names, gs)}

    def check(mine, ref, tag):
        gmax = max(np.abs(v).max() for v in ref.values())
        for k, r in ref.items():
            a_ = np.asarray(mine.get(k, np.zeros(p[k].shape))).reshape(p[k].shape)
            np.testing.assert_allclose(a_, r, rtol=2e-7, atol=1e-9 * gmax + 1e-8 * np.abs(r).max(), err_msg=tag + ':' + k)

    lsd, gr = m.vae_phase(p, x, eps, mm, ms)
    for k, v in (('reconstructionLoss', rec), ('kl', kl), ('enc_loss', enc_loss)):
        np.testing.assert_allclose(lsd[k], v.item(), rtol=1e-10, err_msg=k)
    np.testing.assert_allclose(lsd['reconstruction'], out.permute(0, 2, 3, 1).detach().numpy(), rtol=1e-9, atol=1e-12)
    check(gr, tgrads(enc_loss, groups['Encoder'] + groups['Generator']), 'vae')
    assert not any(k.startswith('Discriminator') for k in gr)
    lsd, gr = m.gen_phase(p, x, eps, mm, ms)
    np.testing.assert_allclose(lsd['gen_loss'], gen_loss.item(), rtol=1e-10)
    check(gr, tgrads(gen_loss, groups['Generator']), 'gen')
    assert all(k.startswith('Generator') for k in gr)
    lsd, gr = m.disc_phase(p, x, eps, alpha, mm, ms)
    for k, v in (('disc_fake', d_fake.mean()), ('disc_real', d_real.mean()), ('penalty', pen), ('disc_loss', disc_loss)):
        np.testing.assert_allclose(lsd[k], v.item(), rtol=1e-9, err_msg=k)
    check(gr, tgrads(disc_loss, groups['Discriminator']), 'disc')
