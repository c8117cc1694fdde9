"""CPU: numpy GMVAE-spatial oracle (hand-written backward incl. the latent-mixture heads, the c-prior max gate and the
TV restore gradient) vs an autograd graph written like the reference's (fp64 round-off agreement)."""
import numpy as np
import pytest
import torch

from oracle import gmvae as og
from oracle import vae as ovae
from tests import torch_ref


def _setup(h, inter, dim_c, dim_z, dim_w, n, seed=0, c_lambda=1.0, dtype=np.float64):
    m = og.GMVAE(h, h, 1, inter, dim_c, dim_z, dim_w, c_lambda)
    p = og.init_params(m.spec, seed=7 + seed, dtype=dtype, perturb=True)
    x = ovae.synthetic_slices(n, h, h, seed=seed, dtype=dtype)
    rng = np.random.default_rng(50 + seed)
    e_w = rng.standard_normal((n, inter, inter, dim_w)).astype(dtype)
    e_z = rng.standard_normal((n, inter, inter, dim_z)).astype(dtype)
    return m, p, x, e_w, e_z


@pytest.mark.parametrize('h,inter,dim_c,dim_z,dim_w,n,c_lambda', [(32, 8, 9, 1, 1, 2, 1.0), (32, 8, 6, 3, 2, 2, 0.001),
                                                                 (64, 8, 9, 1, 1, 1, 0.01)])
def test_gmvae_vs_torch(h, inter, dim_c, dim_z, dim_w, n, c_lambda):
    m, p, x, e_w, e_z = _setup(h, inter, dim_c, dim_z, dim_w, n, c_lambda=c_lambda)
    tv = 1.8
    out, cache = m.forward(p, x, e_w, e_z)
    ls = m.losses(x, out, tv)
    g = m.backward(p, x, out, cache)
    gr = m.backward(p, x, out, cache, tv_lambda=tv)['__dx']

    tp = torch_ref.to_torch(p)
    L, xh, ex = torch_ref.gmvae_losses(tp, m.bn, torch.tensor(x), torch.tensor(e_w), torch.tensor(e_z), m.n_pool, dim_c,
                                       dim_z, c_lambda, tv)
    L['loss'].backward()
    np.testing.assert_allclose(out['xz_mu'], xh.detach().numpy(), rtol=1e-10, atol=1e-12)
    for k in ('pc', 'z_wc_mus', 'z_wc_log_sigma_invs', 'w_sampled', 'z_sampled'):
        np.testing.assert_allclose(out[k], ex[k].detach().numpy(), rtol=1e-9, atol=1e-12, err_msg=k)
    for k in ('mean_p_loss', 'conditional_prior_loss', 'w_prior_loss', 'c_prior_loss', 'loss'):
        np.testing.assert_allclose(ls[k], L[k].item(), rtol=1e-11, err_msg=k)
    np.testing.assert_allclose(ls['restore'], L['restore'].detach().numpy(), rtol=1e-11)
    for name, _, _ in m.spec:
        np.testing.assert_allclose(g[name], tp[name].grad.numpy(), rtol=1e-7, atol=1e-12, err_msg=name)
    np.testing.assert_allclose(g['__dx'], L['dx_loss'].numpy(), rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(gr, L['grads'].numpy(), rtol=1e-7, atol=1e-12)
    # the c-prior gate is exercised on both sides in at least one of the parametrisations
    cl1 = (out['pc'] * np.log(out['pc'] * dim_c + 1e-8)).sum(axis=3)
    assert np.isfinite(cl1).all()


def test_gmvae_param_order_and_counts():
    m = og.GMVAE(256, 256, 1, 8, 9, 1, 1)
    names = [s[0] for s in m.spec]
    assert m.n_pool == 5 and len(m.bn) == 11
    assert names[0] == 'enc_conv2D_0/kernel' and names[2] == 'batch_normalization/gamma'
    assert names.index('q_wz_x/w_mu/kernel') == 20 and names[names.index('Variable') + 1] == 'batch_normalization_5/gamma'
    assert names[-2:] == ['dec_Conv2D_final/kernel', 'dec_Conv2D_final/bias']
    p = og.init_params(m.spec)
    assert np.allclose(p['Variable'], 0.1)


def test_gmvae_restoration_loop_and_train_step():
    m, p, x, e_w, e_z = _setup(32, 8, 9, 1, 1, 2, seed=3)
    noise = lambda step: (e_w, e_z)
    r0 = m.reconstruct(p, x, noise, restore_steps=0)
    out, _ = m.forward(p, x, e_w, e_z)
    np.testing.assert_allclose(r0['reconstruction'], out['xz_mu'])
    r = m.reconstruct(p, x, noise, restore_steps=3, restore_lr=1e-3, tv_lambda=1.8)
    g1 = m.restore_grads(p, x, e_w, e_z, 1.8)
    x1 = x - 1e-3 * g1
    g2 = m.restore_grads(p, x1, e_w, e_z, 1.8)
    x2 = x1 - 1e-3 * g2
    x3 = x2 - 1e-3 * m.restore_grads(p, x2, e_w, e_z, 1.8)
    np.testing.assert_allclose(r['reconstruction'], x3, rtol=1e-12)
    assert r['l1err'] == pytest.approx(np.abs(x - x3).sum())
    opt = m.new_opt(p)
    l0 = m.train_step(p, opt, x, e_w, e_z, lr=1e-3)[1]['loss']
    for _ in range(5):
        l1 = m.train_step(p, opt, x, e_w, e_z, lr=1e-3)[1]['loss']
    assert l1 < l0
