"""GPU parity of the dense GMVAE (models/gaussian_mixture_variational_autoencoder.py, trainers/GMVAE.py) through the C-ABI
(uad_gan_* with UAD_GAN_AAE / aae_kind 3, uad_gan_restore_step) vs the fp64 oracle: reconstruction, z_sampled, the four loss terms,
every parameter gradient, Adam, and the restoration-mode input gradient / in-place update.
Tolerance 1e-4 max-norm relative (north_star); 5e-4 on long-reduction bias / BN sums and the latent heads, as for the spatial GMVAE."""
import numpy as np
import pytest
import torch

from oracle import gmvae_dense as ogd
from oracle import vae as ovae

pytestmark = pytest.mark.gpu

try:
    from unsupervised_anomaly_detection_brain_mri_amd import _lib
    from unsupervised_anomaly_detection_brain_mri_amd.gan_engine import GanEngine
    from tests.gpu_util import assert_close
except Exception:
    GanEngine = None


def _f64(d):
    return {k: np.asarray(v, np.float64) for k, v in d.items()}


def _setup(h, dim_c, dim_z, dim_w, n, seed=0, c_lambda=1.0, perturb=True, dropout=False):
    m = ogd.GMVAEDense(h, 8, dim_c, dim_z, dim_w, c_lambda)
    p32 = ogd.init_params(m.spec, seed=7 + seed, dtype=np.float32, perturb=perturb)
    x = ovae.synthetic_slices(n, h, h, seed=seed, dtype=np.float32)
    rng = np.random.default_rng(50 + seed)
    e_w = rng.standard_normal((n, dim_w)).astype(np.float32)
    e_z = rng.standard_normal((n, dim_z)).astype(np.float32)
    masks = {}
    if dropout:
        flat = [s for nm, s, _ in m.spec if nm == 'Bottleneck/dense_4/kernel'][0][1]
        keep = lambda s: ((rng.random(s) >= 0.2) / 0.8).astype(np.float32)
        masks = {'w_mu': keep((n, dim_w)), 'w_ls': keep((n, dim_w)), 'z_mu': keep((n, dim_z)), 'dec': keep((n, flat))}
    return m, p32, x, e_w, e_z, masks


def _engine(m, n, math='bf16x3'):
    return GanEngine(m.height, m.height, 1, m.inter_res, zdim=m.dim_z, max_batch=n, variant='aae', aae_kind='gmvae', dim=m.dim_c, dim_w=m.dim_w,
                     c_lambda=m.c_lambda, math=math)


def test_gmvae_dense_param_table():
    m = ogd.GMVAEDense(128, 8, 6, 1, 1)
    eng = _engine(m, 1)
    assert [(n, tuple(s)) for n, s, _ in eng.spec] == [(n, tuple(s)) for n, s, _ in m.spec]
    assert eng.group('AE') == (0, eng.nparams)
    eng.close()
    with pytest.raises(Exception):
        GanEngine(64, 64, 1, 8, zdim=128, max_batch=1, variant='aae', aae_kind='gmvae', dim=64)       # dim_z * dim_c > 4096


@pytest.mark.parametrize('math', ['f32', 'bf16x3'])
@pytest.mark.parametrize('h,dim_c,dim_z,dim_w,n,c_lambda,dropout', [(32, 6, 1, 1, 2, 1.0, False), (64, 5, 3, 2, 3, 0.001, True),
                                                                   (128, 6, 1, 1, 4, 0.01, True), (64, 9, 128, 64, 2, 1.0, False)])
def test_gmvae_dense_forward_backward_parity(h, dim_c, dim_z, dim_w, n, c_lambda, dropout, math):
    m, p32, x, e_w, e_z, masks = _setup(h, dim_c, dim_z, dim_w, n, c_lambda=c_lambda, dropout=dropout)
    p64, x64 = _f64(p32), x.astype(np.float64)
    out, cache = m.forward(p64, x64, e_w.astype(np.float64), e_z.astype(np.float64), _f64(masks))
    ls = m.losses(x64, out)
    g = m.backward(p64, x64, out, cache)
    eng = _engine(m, n, math)
    eng.set_params(p32)
    got = eng.gm_phase(x, e_w, e_z, masks)
    torch.cuda.synchronize()
    assert_close(got['reconstruction'].cpu().numpy(), out['xz_mu'], name='xz_mu')
    assert_close(got['L1'].cpu().numpy(), ls['L1'], tol=2e-4, name='L1')
    assert_close(got['z_sampled'].cpu().numpy(), out['z_sampled'], tol=2e-4, name='z_sampled')
    for dbg, key in (('pc', 'pc'), ('M', 'z_wc_mus'), ('Lq', 'z_wc_log_sigma_invs'), ('w_s', 'w_sampled')):
        ref = out[key].reshape(n, -1)
        assert_close(eng.debug_buffer(dbg).cpu().numpy()[:ref.size].reshape(ref.shape), ref, tol=2e-4, name=dbg)
    for key in ('mean_p_loss', 'conditional_prior_loss', 'w_prior_loss', 'c_prior_loss', 'loss'):
        assert abs(float(got[key]) - ls[key]) <= 2e-4 * max(abs(ls[key]), 1e-3), (key, float(got[key]), ls[key])
    grads = eng.get_grads()
    for name, _, _ in m.spec:
        tol = 1e-4 if name.endswith('kernel') and name.startswith(('Encoder/enc', 'Decoder/dec')) else 5e-4
        assert_close(grads[name], g[name], tol=tol, name=name)
    eng.close()


@pytest.mark.parametrize('h,dim_c,dim_z,dim_w,n', [(64, 6, 1, 1, 2), (128, 6, 2, 3, 4)])
def test_gmvae_dense_restore_step_matches_oracle(h, dim_c, dim_z, dim_w, n):
    """`grads` of trainers/GMVAE.py:93-94 and the in-place update of :183-184, three chained steps."""
    m, p32, x, e_w, e_z, _ = _setup(h, dim_c, dim_z, dim_w, n, seed=5)
    p64 = _f64(p32)
    eng = _engine(m, n)
    eng.set_params(p32)
    sentinel = np.full(eng.nparams, 3.0, np.float32)
    eng.set_buffer_host(_lib.BUF_GRADS, sentinel)
    xr = torch.from_numpy(x.copy()).cuda()
    ref = x.astype(np.float64)
    lr, tv = 2e-2, 1.8
    for step in range(3):
        gref = m.restore_grads(p64, ref, e_w.astype(np.float64), e_z.astype(np.float64), tv)
        ggot = eng.gm_restore_step(xr, e_w, e_z, tv_lambda=tv, restore_lr=lr, want_grads=True)
        torch.cuda.synchronize()
        if step == 0:
            # identical inputs; TV sign flips between neighbouring residuals that agree to ~1e-6 are decided by rounding (see the spatial test)
            gg = ggot.cpu().numpy()
            bad = np.abs(gg - gref) > 3e-4 * np.abs(gref).max()
            assert bad.mean() <= 2e-3, f'{bad.mean():.2e} of the pixels differ'
            if bad.any():
                q = np.abs(gg - gref)[bad] / tv
                assert np.abs(q - np.round(q)).max() <= 1e-2, 'differences are not TV sign flips'
        ref = ref - lr * gref
    assert np.abs(xr.cpu().numpy() - ref).max() <= 8 * lr * tv + 1e-4
    assert np.mean(np.abs(xr.cpu().numpy() - ref)) <= 2e-5
    assert np.array_equal(eng.get_buffer_host(_lib.BUF_GRADS), sentinel)     # no parameter gradient was written
    eng.close()


def test_gmvae_dense_train_trajectory_and_errors():
    m, p32, x, e_w, e_z, _ = _setup(32, 6, 1, 1, 4, seed=2, perturb=False)
    p64 = _f64(p32)
    opt = m.new_opt(p64)
    eng = _engine(m, 4)
    eng.set_params(p32)
    ref_l, got_l = [], []
    for _ in range(8):
        _, ls, _ = m.train_step(p64, opt, x.astype(np.float64), e_w.astype(np.float64), e_z.astype(np.float64), lr=1e-4)
        ref_l.append(float(ls['loss']))
        out = eng.gm_phase(x, e_w, e_z)
        eng.adam('AE', 1e-4, beta1=0.5, beta2=0.999)
        got_l.append(float(out['loss']))
    np.testing.assert_allclose(got_l, ref_l, rtol=3e-4)
    flat = eng.get_buffer_host(_lib.BUF_PARAMS)
    ref = np.concatenate([p64[nm].reshape(-1) for nm, _, _ in m.spec])
    assert np.abs(flat - ref).max() <= 2e-3 * np.abs(ref).max()
    assert eng.step_count('AE') == 8
    with pytest.raises(Exception):
        eng.aae_phase('Discriminator', x)                                     # one phase only
    with pytest.raises(ValueError):
        eng.gm_restore_step(x)                                                # needs a device tensor (updated in place)
    other = GanEngine(32, 32, 1, 8, zdim=16, max_batch=2, variant='aae', aae_kind='aae')
    with pytest.raises(ValueError):
        other.gm_phase(x[:2])
    other.close()
    eng.close()
