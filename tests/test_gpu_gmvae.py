"""GPU parity of the spatial GMVAE (models/gaussian_mixture_variational_autoencoder_spatial.py,
trainers/GMVAE_spatial.py) through the C-ABI vs the fp64 oracle: reconstruction, latent maps, the four loss terms,
every parameter gradient (trunk + latent heads), Adam, and the restoration-mode input gradient / in-place update.
Tolerance 1e-4 max-norm relative (north_star); 5e-4 on long-reduction bias/BN sums as for the VAE."""
import numpy as np
import pytest
import torch

from oracle import gmvae as og
from oracle import vae as ovae

pytestmark = pytest.mark.gpu

try:
    from unsupervised_anomaly_detection_brain_mri_amd import _lib
    from unsupervised_anomaly_detection_brain_mri_amd.engine import Engine
    from tests.gpu_util import assert_close
except Exception:
    Engine = None


def _f64(d):
    return {k: np.asarray(v, np.float64) for k, v in d.items()}


def _setup(h, inter, dim_c, dim_z, dim_w, n, seed=0, c_lambda=1.0, perturb=True):
    m = og.GMVAE(h, h, 1, inter, dim_c, dim_z, dim_w, c_lambda)
    p32 = og.init_params(m.spec, seed=7 + seed, dtype=np.float32, perturb=perturb)
    x = ovae.synthetic_slices(n, h, h, seed=seed, dtype=np.float32)
    rng = np.random.default_rng(50 + seed)
    e_w = rng.standard_normal((n, inter, inter, dim_w)).astype(np.float32)
    e_z = rng.standard_normal((n, inter, inter, dim_z)).astype(np.float32)
    return m, p32, x, e_w, e_z


def _engine(m, n, math='bf16x3'):
    return Engine('GMVAE_spatial', m.h, m.w, 1, m.inter, max_batch=n, math=math, dim_c=m.dim_c, dim_z=m.dim_z,
                  dim_w=m.dim_w, c_lambda=m.c_lambda)


def test_gmvae_param_table():
    m = og.GMVAE(256, 256, 1, 8, 9, 1, 1)
    eng = _engine(m, 1)
    assert [(n, tuple(s)) for n, s, _ in eng.spec] == [(n, tuple(s)) for n, s, _ in m.spec]
    eng.close()


@pytest.mark.parametrize('math', ['f32', 'bf16x3'])
@pytest.mark.parametrize('h,inter,dim_c,dim_z,dim_w,n,c_lambda', [(32, 8, 9, 1, 1, 2, 1.0), (64, 8, 6, 3, 2, 3, 0.001),
                                                                 (128, 8, 9, 1, 1, 2, 0.01), (256, 8, 9, 1, 1, 1, 1.0),
                                                                 (64, 8, 9, 16, 1, 2, 1.0)])
def test_gmvae_forward_backward_parity(h, inter, dim_c, dim_z, dim_w, n, c_lambda, math):
    m, p32, x, e_w, e_z = _setup(h, inter, dim_c, dim_z, dim_w, n, c_lambda=c_lambda)
    p64 = _f64(p32)
    x64 = x.astype(np.float64)
    out, cache = m.forward(p64, x64, e_w.astype(np.float64), e_z.astype(np.float64))
    ls = m.losses(x64, out)
    g = m.backward(p64, x64, out, cache)

    eng = _engine(m, n, math)
    eng.set_params(p32)
    got = eng.gm_forward(x, e_w, e_z, want_backward=True)
    eng.backward()
    torch.cuda.synchronize()
    assert_close(got['x_hat'].cpu().numpy(), out['xz_mu'], name='xz_mu')
    assert_close(got['L1'].cpu().numpy(), ls['L1'], tol=2e-4, name='L1')
    for k, ok in (('z_mu', 'z_mu'), ('z_log_sigma', 'z_log_sigma'), ('w_mu', 'w_mu'), ('w_log_sigma', 'w_log_sigma'),
                  ('pc', 'pc')):
        assert_close(got[k].cpu().numpy(), out[ok], tol=2e-4, name=k)
    sc = got['scalars'].cpu().numpy()
    for idx, key in ((0, 'mean_p_loss'), (1, 'conditional_prior_loss'), (2, 'loss'), (3, 'w_prior_loss'), (4, 'c_prior_loss')):
        assert abs(sc[idx] - ls[key]) <= 2e-4 * max(abs(ls[key]), 1e-3), (key, sc[idx], ls[key])
    grads = eng.get_grads()
    for name, _, _ in m.spec:
        tol = 1e-4 if name.endswith('kernel') and '/' in name and not name.startswith(('q_wz', 'p_z')) else 5e-4
        assert_close(grads[name], g[name], tol=tol, name=name)
    eng.close()


@pytest.mark.parametrize('h,dim_c,dim_z,n', [(64, 9, 1, 2), (128, 6, 2, 1), (256, 9, 1, 4)])   # the last one takes the bench's kernel path
def test_gmvae_restore_step_matches_oracle(h, dim_c, dim_z, n):
    """`grads` of trainers/GMVAE_spatial.py:91-92 and the in-place update of :189-190, three chained steps."""
    m, p32, x, e_w, e_z = _setup(h, 8, dim_c, dim_z, 1, n, seed=5)
    p64 = _f64(p32)
    eng = _engine(m, n)
    eng.set_params(p32)
    sentinel = np.full(eng.nparams, 3.0, np.float32)
    eng.set_buffer_host(_lib.BUF_GRADS, sentinel)
    xr = torch.from_numpy(x.copy()).cuda()
    ref = x.astype(np.float64)
    # larger step than the reference default so that three steps move x measurably
    lr, tv = 2e-2, 1.8
    for step in range(3):
        gref = m.restore_grads(p64, ref, e_w.astype(np.float64), e_z.astype(np.float64), tv)
        ggot = eng.restore_step(xr, e_w, e_z, tv_lambda=tv, restore_lr=lr, want_grads=True)
        torch.cuda.synchronize()
        if step == 0:
            # identical inputs.  The TV term contributes +-tv_lambda per neighbour through sign(r[p] - r[q]); where two
            # neighbouring residuals agree to ~1e-6 that sign is decided by rounding, so a few pixels may differ by multiples of
            # tv_lambda -- everything else must agree to 3e-4 of the gradient's max
            gg = ggot.cpu().numpy()
            bad = np.abs(gg - gref) > 3e-4 * np.abs(gref).max()
            assert bad.mean() <= 2e-3, f'{bad.mean():.2e} of the pixels differ'
            if bad.any():
                q = np.abs(gg - gref)[bad] / tv
                assert np.abs(q - np.round(q)).max() <= 1e-2, 'differences are not TV sign flips'
        ref = ref - lr * gref
    # after chained steps single pixels may flip a TV sign (|step| = tv_lambda); compare the restored images
    # (a flipped sign moves a pixel by lr * 2 * tv per step and can flip its neighbours' next step: bound the worst pixel by a few
    # such events, and require the images to agree on average)
    assert np.abs(xr.cpu().numpy() - ref).max() <= 8 * lr * tv + 1e-4
    assert np.mean(np.abs(xr.cpu().numpy() - ref)) <= 2e-5
    assert np.array_equal(eng.get_buffer_host(_lib.BUF_GRADS), sentinel)     # no parameter gradient was written
    eng.close()


def test_gmvae_train_trajectory():
    m, p32, x, e_w, e_z = _setup(32, 8, 9, 1, 1, 4, seed=2, perturb=False)
    p64 = _f64(p32)
    opt = m.new_opt(p64)
    eng = _engine(m, 4)
    eng.set_params(p32)
    ref_l, got_l = [], []
    for _ in range(8):
        _, ls, _ = m.train_step(p64, opt, x.astype(np.float64), e_w.astype(np.float64), e_z.astype(np.float64), lr=5e-5)
        ref_l.append(float(ls['loss']))
        out = eng.gm_train_step(x, e_w, e_z, lr=5e-5)
        got_l.append(float(out['scalars'][2].item()))
    np.testing.assert_allclose(got_l, ref_l, rtol=3e-4)
    flat = eng.get_buffer_host(_lib.BUF_PARAMS)
    ref = np.concatenate([p64[nm].reshape(-1) for nm, _, _ in m.spec])
    assert np.abs(flat - ref).max() <= 2e-3 * np.abs(ref).max()
    eng.close()


def test_gmvae_error_paths():
    with pytest.raises(ValueError):
        Engine('GMVAE_spatial', 64, 64, 1, 8, max_batch=1, dim_c=100)       # dim_c > 64
    eng = Engine('AE', 32, 32, 1, 8, 16, max_batch=1)
    with pytest.raises(ValueError):
        eng.restore_step(torch.zeros(1, 32, 32, 1, device='cuda'))          # restoration needs a GMVAE or a VAE handle
    eng.close()
