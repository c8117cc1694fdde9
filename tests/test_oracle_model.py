"""CPU: the numpy oracle (hand-written backward, TF semantics) vs an independent
torch-autograd formulation.  This is the strongest pin available for the model
path (no TF in the image): two independent restatements must agree to fp64
round-off."""
import numpy as np
import pytest
import torch

from oracle import nn as onn
from oracle import vae as ovae
from tests import torch_ref


def _setup(arch, h, inter, zdim, n, dtype=np.float64, seed=0):
    m = ovae.Model(arch, h, h, 1, inter, zdim)
    p = ovae.init_params(m.spec, seed=3 + seed, dtype=dtype, perturb=True)
    x = ovae.synthetic_slices(n, h, h, seed=seed, dtype=dtype)
    rng = np.random.default_rng(100 + seed)
    eps = rng.standard_normal((n, zdim)).astype(dtype)
    flat = inter * inter * (p['Bottleneck/conv2d/kernel'].shape[-1])
    if arch == 'VAE':
        masks = {'mu': onn.make_dropout_mask(rng, (n, zdim), 0.2, dtype),
                 'sigma': onn.make_dropout_mask(rng, (n, zdim), 0.2, dtype),
                 'dec': onn.make_dropout_mask(rng, (n, flat), 0.2, dtype)}
    else:
        masks = {'z': onn.make_dropout_mask(rng, (n, zdim), 0.2, dtype)}
    return m, p, x, eps, masks


def test_same_pads():
    assert onn.same_pads(128, 5, 2) == (64, 1, 2)
    assert onn.same_pads(8, 1, 1) == (8, 0, 0)
    assert onn.same_pads(64, 3, 2) == (32, 0, 1)
    assert onn.same_pads(64, 4, 2) == (32, 1, 1)
    assert onn.same_pads(7, 5, 2) == (4, 2, 2)


def test_param_count_matches_survey():
    m = ovae.Model('VAE', 128, 128, 1, 8, 128)
    assert sum(int(np.prod(s)) for _, s, _ in m.spec) == 1758449
    m = ovae.Model('AE', 128, 128, 1, 8, 128)
    assert sum(int(np.prod(s)) for _, s, _ in m.spec) == 1627249


@pytest.mark.parametrize('arch,h,inter,zdim,n', [('VAE', 32, 8, 16, 2), ('AE', 32, 8, 16, 2), ('VAE', 64, 8, 32, 1)])
def test_forward_backward_vs_torch(arch, h, inter, zdim, n):
    m, p, x, eps, masks = _setup(arch, h, inter, zdim, n)
    out, cache = m.forward(p, x, eps if arch == 'VAE' else None, masks)
    ls = m.losses(x, out)
    g = m.backward(p, x, out, cache, masks)

    tp = torch_ref.to_torch(p)
    tm = {k: torch.tensor(v) for k, v in masks.items()}
    xt = torch.tensor(x, requires_grad=True)
    tl, xh, extras = torch_ref.forward_loss(arch, m.spec, tp, xt, torch.tensor(eps), tm, inter, m.n_pool)
    tl['loss'].backward()

    np.testing.assert_allclose(out['x_hat'], xh.detach().numpy(), rtol=1e-10, atol=1e-12)
    for k in tl:
        np.testing.assert_allclose(ls[k], tl[k].item(), rtol=1e-11)
    for name, _, _ in m.spec:
        np.testing.assert_allclose(g[name], tp[name].grad.numpy(), rtol=1e-8, atol=1e-12, err_msg=name)
    # torch's x.grad also holds the direct d|x_hat - x|/dx = -sign/N term; __dx is the path through the net
    direct = -np.sign(out['x_hat'] - x) / n
    np.testing.assert_allclose(g['__dx'] + direct, xt.grad.numpy(), rtol=1e-8, atol=1e-12)


def test_conv_transpose_is_adjoint_of_conv():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 16, 16, 3))
    w = rng.standard_normal((5, 5, 3, 4))          # HWIO for the fwd conv 16x16x3 -> 8x8x4
    g = rng.standard_normal((2, 8, 8, 4))
    y = onn.conv2d_fwd(x, w, None, 2)
    # ConvT with kernel [kh,kw,Cout=3,Cin=4] == w maps g (8x8x4) -> 16x16x3
    xt = onn.conv2d_transpose_fwd(g, w, None, 2)
    np.testing.assert_allclose((y * g).sum(), (x * xt).sum(), rtol=1e-12)


def test_adam_tf_form_first_steps():
    p = np.array([1.0, -2.0]); g = np.array([0.5, -0.25])
    m = np.zeros(2); v = np.zeros(2)
    onn.adam_tf_step(p, g, m, v, 1, lr=0.1, beta1=0.5, beta2=0.999, eps=1e-8)
    # step 1: lr_t = lr*sqrt(1-b2)/(1-b1); m=(1-b1)g; v=(1-b2)g^2 -> update ~= lr*sign(g)
    lr_t = 0.1 * np.sqrt(1 - 0.999) / (1 - 0.5)
    exp = np.array([1.0, -2.0]) - lr_t * (0.5 * g) / (np.sqrt(0.001 * g * g) + 1e-8)
    np.testing.assert_allclose(p, exp, rtol=1e-14)


def test_train_steps_decrease_loss_and_match_torch_adam_free_grads():
    m, p, x, eps, masks = _setup('VAE', 32, 8, 16, 4, dtype=np.float64, seed=1)
    opt = m.new_opt(p)
    losses = []
    for _ in range(15):
        _, ls, _ = m.train_step(p, opt, x, eps, masks, lr=1e-3)
        losses.append(float(ls['loss']))
    assert losses[-1] < losses[0]
    assert opt['t'] == 15


def _setup_ce(h, inter, zdim, n, seed=4, dropout=True):
    m = ovae.CeVAE(h, h, 1, inter, zdim)
    p = ovae.init_params(m.spec, seed, np.float64, perturb=True)
    rng = np.random.default_rng(seed + 1)
    x = ovae.synthetic_slices(n, h, h, seed, np.float64)
    x_ce = x.copy()
    x_ce[:, h // 4:h // 4 + 6, h // 3:h // 3 + 6] = 0
    eps = rng.standard_normal((n, zdim))
    flat = inter * inter * (min(128, 32 * 2 ** (m.n_pool - 1)) // 8)
    masks = {}
    if dropout:
        for k, w in (('mu', zdim), ('sigma', zdim), ('dec', flat), ('mu_ce', zdim), ('dec_ce', flat)):
            masks[k] = onn.make_dropout_mask(rng, (n, w), 0.2, np.float64)
    return m, p, x, x_ce, eps, masks


@pytest.mark.parametrize('h,inter,zdim,n', [(32, 8, 16, 2), (64, 8, 32, 1)])
def test_cevae_vs_torch(h, inter, zdim, n):
    """ceVAE restatement (two passes through shared layers, summed gradients, anomaly map) against one autograd
    graph built like the reference's (context_encoder_variational_autoencoder.py, trainers/ceVAE.py:38-51)."""
    m, p, x, x_ce, eps, masks = _setup_ce(h, inter, zdim, n)
    out, caches = m.ce_forward(p, x, x_ce, eps, masks)
    ls = m.ce_losses(x, x_ce, out)
    g = m.ce_backward(p, x, x_ce, out, caches, masks)

    tp = torch_ref.to_torch(p)
    tm = {k: torch.tensor(v) for k, v in masks.items()}
    tl, xh, xh_ce = torch_ref.cevae_losses(tp, torch.tensor(x), torch.tensor(x_ce), torch.tensor(eps), tm, m.n_pool)
    tl['loss'].backward()
    np.testing.assert_allclose(out['x_hat'], xh.detach().numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(out['x_hat_ce'], xh_ce.detach().numpy(), rtol=1e-10, atol=1e-12)
    for k in ('Rec_ce', 'Rec_vae', 'reconstructionLoss', 'kl', 'loss', 'loss_vae'):
        np.testing.assert_allclose(ls[k], tl[k].item(), rtol=1e-11, err_msg=k)
    for k in ('L1_vae', 'L1_ce', 'L1'):
        np.testing.assert_allclose(ls[k], tl[k].detach().numpy(), rtol=1e-10, atol=1e-12, err_msg=k)
    for name, _, _ in m.spec:
        np.testing.assert_allclose(g[name], tp[name].grad.numpy(), rtol=1e-8, atol=1e-12, err_msg=name)
    np.testing.assert_allclose(g['anomaly'], tl['anomaly'].numpy(), rtol=1e-8, atol=1e-14)


def test_cevae_names_and_reconstruct():
    m = ovae.CeVAE(128, 128, 1, 8, 128)
    names = [s[0] for s in m.spec]
    # unnamed Dense layers are numbered in construction order: mu, sigma, dec (context_encoder_..._autoencoder.py:30-32)
    assert names[names.index('Bottleneck/conv2d/bias') + 1] == 'Bottleneck/dense/kernel'
    assert 'Bottleneck/dense_1/kernel' in names and 'Bottleneck/dense_2/bias' in names
    assert sum(int(np.prod(s[1])) for s in m.spec) == 1758449
    m, p, x, _, eps, _ = _setup_ce(32, 8, 16, 2, dropout=False)
    r = m.ce_reconstruct(p, x, eps)
    assert r['reconstruction'].shape == x.shape
    np.testing.assert_allclose(r['reconstruction'], x - r['anomaly'])
    np.testing.assert_allclose(r['l1err'], np.abs(r['anomaly']).sum())
    r0 = m.ce_reconstruct(p, x, eps, use_gradient_based_restoration=False)
    out, _ = m.ce_forward(p, x, x, eps)
    np.testing.assert_allclose(r0['reconstruction'], out['x_hat'])


def test_retrieve_masked_batch_replays_reference_defect():
    """trainers/CE.py:123-139: the returned batch is batch * (LAST sample's mask), broadcast."""
    import random
    n, h = 3, 64
    x = np.ones((n, h, h, 1))
    bm = np.zeros((n, h, h, 1)); bm[:, 8:56, 10:50] = 1
    out = ovae.retrieve_masked_batch(x, bm, random.Random(5))
    assert out.shape == x.shape
    holes = (out == 0)
    assert holes.any()
    for i in range(1, n):
        assert (holes[i] == holes[0]).all()
    # holes are unions of 20x20 squares inside the brain bounding box
    ys, xs = np.nonzero(holes[0, :, :, 0])
    assert ys.min() >= 8 and ys.max() <= 55 and xs.min() >= 10 and xs.max() <= 49


def test_spatial_ae_vs_torch():
    """models/autoencoder_spatial.py: oracle forward / backward vs an autograd graph (fp64)."""
    import torch
    import torch.nn.functional as F
    from tests import torch_ref
    m = ovae.SpatialAE(32, 32, 1, 8)
    p = ovae.init_params(m.spec, seed=2, dtype=np.float64, perturb=True)
    x = ovae.synthetic_slices(3, 32, 32, seed=1, dtype=np.float64)
    rng = np.random.default_rng(0)
    mask = (rng.random((3, 8, 8, 64)) > 0.2) / 0.8
    out, cache = m.forward(p, x, {'z': mask})
    g = m.backward(p, x, out, cache, {'z': mask})
    tp = torch_ref.to_torch(p)
    a = torch.tensor(x).permute(0, 3, 1, 2)

    def bn(t, name):
        return t * (tp[name + '/gamma'] / np.sqrt(1.0 + 1e-3)).view(1, -1, 1, 1) + tp[name + '/beta'].view(1, -1, 1, 1)

    for i in range(2):
        a = F.leaky_relu(bn(torch_ref._conv_same(a, tp[f'Encoder/enc_conv2D_{i}/kernel'], tp[f'Encoder/enc_conv2D_{i}/bias'], 2),
                            f'Encoder/batch_normalization_{i}'), 0.3)
    z = a * torch.tensor(mask).permute(0, 3, 1, 2)
    a = F.relu(bn(z, 'Decoder/batch_normalization'))
    for i in range(2):
        a = F.leaky_relu(bn(torch_ref._convT_same(a, tp[f'Decoder/dec_Conv2DT_{i}/kernel'], tp[f'Decoder/dec_Conv2DT_{i}/bias'], 2),
                            f'Decoder/batch_normalization_{i + 1}'), 0.3)
    xh = torch_ref._conv_same(a, tp['Decoder/dec_Conv2D_final/kernel'], tp['Decoder/dec_Conv2D_final/bias'], 1).permute(0, 2, 3, 1)
    loss = (xh - torch.tensor(x)).abs().sum(dim=(1, 2, 3)).mean()
    loss.backward()
    np.testing.assert_allclose(out['x_hat'], xh.detach().numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(m.losses(x, out)['loss'], loss.item(), rtol=1e-12)
    for name, _, _ in m.spec:
        np.testing.assert_allclose(g[name], tp[name].grad.numpy(), rtol=1e-7, atol=1e-12, err_msg=name)


def test_vae_restore_grads_vs_torch():
    """trainers/VAE_You.py:52-53: d (rec_n + kl_n + tv * TV_n(x - x_hat)) / d x, oracle vs autograd (fp64)."""
    m = ovae.Model('VAE', 32, 32, 1, 8, 16)
    p = ovae.init_params(m.spec, seed=4, dtype=np.float64, perturb=True)
    x = ovae.synthetic_slices(2, 32, 32, seed=3, dtype=np.float64)
    eps = np.random.default_rng(2).standard_normal((2, 16))
    tv = 1.8
    g = m.restore_grads(p, x, eps, tv)
    tp = torch_ref.to_torch(p, requires_grad=False)
    xt = torch.tensor(x, requires_grad=True)
    L, xh, _ = torch_ref.forward_loss('VAE', m.spec, tp, xt, torch.tensor(eps), {}, 8, m.n_pool)
    r = xt - xh
    tvn = (r[:, 1:] - r[:, :-1]).abs().sum() + (r[:, :, 1:] - r[:, :, :-1]).abs().sum()
    obj = x.shape[0] * L['loss'] + tv * tvn          # sum over samples of (rec_n + kl_n) = N * mean
    obj.backward()
    np.testing.assert_allclose(g, xt.grad.numpy(), rtol=1e-7, atol=1e-10)


def test_act_override_differentiates_with_a_foreign_activation_pattern():
    """oracle.nn.act_override (the hook behind tests/gpu_util.py: kink_overrides): with a pattern table keyed by the fingerprint of a site's
    post-activation array, leaky_relu_bwd takes the derivative sides of the table instead of its own; sites without an entry and runs outside
    the context are untouched; the second run of the same forward reproduces the fingerprints bit for bit."""
    from oracle import nn as onn
    rng = np.random.default_rng(3)
    y = rng.standard_normal((2, 4, 4, 8))
    g = rng.standard_normal(y.shape)
    post = onn.leaky_relu_fwd(y, 0.3)
    own = onn.leaky_relu_bwd(y, g, 0.3)
    np.testing.assert_array_equal(own, np.where(y > 0, g, 0.3 * g))
    pat = y > 0
    pat[0, 0, 0, :3] ^= True                                    # three elements taken on the other side
    table = {onn.act_fingerprint(post): pat}
    with onn.act_override(table) as ov:
        got = onn.leaky_relu_bwd(y.copy(), g, 0.3)              # a recomputed (bit-identical) pre-activation finds its entry
        other = onn.leaky_relu_bwd(y + 1.0, g, 0.3)             # another site: no entry
        assert len(ov.used) == 1
    np.testing.assert_array_equal(got, np.where(pat, g, 0.3 * g))
    np.testing.assert_array_equal(other, np.where(y + 1.0 > 0, g, 0.3 * g))
    np.testing.assert_array_equal(onn.leaky_relu_bwd(y, g, 0.3), own)          # outside the context: the oracle's own pattern
    # relu written as maximum(y, 0) and as where(y > 0, y, 0 * y) share a fingerprint (-0.0 folded)
    assert onn.act_fingerprint(np.maximum(y, 0)) == onn.act_fingerprint(onn.leaky_relu_fwd(y, 0.0))
