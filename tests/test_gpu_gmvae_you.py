"""GPU parity of the original-architecture spatial GMVAE (models/gaussian_mixture_variational_autoencoder_You.py under
trainers/GMVAE_spatial.py) through the C-ABI (uad_gan_* with UAD_GAN_AAE / aae_kind 6, uad_gan_restore_step) vs the fp64 oracle:
reconstruction, z_sampled, the four loss terms, every parameter gradient, the restoration gradient / in-place update, Adam trajectory.
Tolerance 1e-4 max-norm relative (north_star), 5e-4 on bias / head sums; ReLU-kink flips are counted exactly (device vs oracle signs of
every pre-activation) and loosen the gradient bound to 5e-2 in the L2 norm, as in the other materialised-graph tests."""
import numpy as np
import pytest
import torch

from oracle import gmvae as og
from oracle import gmvae_you as oy
from oracle import vae as ovae

from oracle import nn as onn  # noqa: E402
from tests.gpu_util import kink_overrides  # noqa: E402

pytestmark = pytest.mark.gpu

try:
    from unsupervised_anomaly_detection_brain_mri_amd import _lib
    from unsupervised_anomaly_detection_brain_mri_amd.gan_engine import GanEngine
    from tests.gpu_util import assert_close
except Exception:
    GanEngine = None


def _f64(d):
    return {k: np.asarray(v, np.float64) for k, v in d.items()}


def _setup(h, dim_c, dim_z, dim_w, n, seed=0, c_lambda=1.0, perturb=True):
    m = oy.GMVAEYou(h, dim_c, dim_z, dim_w, c_lambda)
    p32 = og.init_params(m.spec, seed=7 + seed, dtype=np.float32, perturb=perturb)
    x = ovae.synthetic_slices(n, h, h, seed=seed, dtype=np.float32)
    rng = np.random.default_rng(50 + seed)
    r = h // 4
    return m, p32, x, rng.standard_normal((n, r, r, dim_w)).astype(np.float32), rng.standard_normal((n, r, r, dim_z)).astype(np.float32)


def _engine(m, n, math='f32'):
    return GanEngine(m.h, m.h, 1, m.h // 4, zdim=m.dim_z, max_batch=n, variant='aae', aae_kind='gmvae_you', dim=m.dim_c, dim_w=m.dim_w,
                     c_lambda=m.c_lambda, math=math)


def _pairs(eng, cache):
    """(device ReLU input, oracle ReLU input, alpha 0) of every ReLU site"""
    def grab(name, ref):
        return eng.debug_buffer(name).cpu().numpy()[:ref.size].reshape(ref.shape), ref, 0.0

    for i, ref in enumerate(cache['ec'][:-1]):
        yield grab(f'yec{i}', ref)
    yield grab('yec5', cache['ec'][-1])
    for i, (op, ref) in enumerate(zip(oy.DEC[:-1], cache['dc'][:-1])):
        if op[0] != 'up' and op[2]:
            yield grab(f'ydc{i}', ref)


@pytest.mark.parametrize('math', ['f32', 'bf16x3_all'])
@pytest.mark.parametrize('h,dim_c,dim_z,dim_w,n,c_lambda', [(32, 6, 1, 1, 2, 1.0), (64, 5, 8, 2, 2, 0.001), (128, 9, 1, 1, 2, 0.01)])
def test_gmvae_you_forward_backward_parity(h, dim_c, dim_z, dim_w, n, c_lambda, math):
    m, p32, x, e_w, e_z = _setup(h, dim_c, dim_z, dim_w, n, c_lambda=c_lambda)
    p64, x64 = _f64(p32), x.astype(np.float64)
    out, cache = m.forward(p64, x64, e_w.astype(np.float64), e_z.astype(np.float64))
    ls = m.losses(x64, out)
    g = m.backward(p64, x64, out, cache)
    eng = _engine(m, n, math)
    assert [(a, tuple(b)) for a, b, _ in eng.spec] == [(a, tuple(b)) for a, b, _ in m.spec]
    eng.set_params(p32)
    got = eng.gm_phase(x, e_w, e_z)
    torch.cuda.synchronize()
    assert_close(got['reconstruction'].cpu().numpy(), out['xz_mu'], name='xz_mu')
    assert_close(got['L1'].cpu().numpy(), ls['L1'], tol=2e-4, name='L1')
    assert_close(got['z_sampled'].cpu().numpy(), out['z_sampled'], tol=2e-4, name='z_sampled')
    for key in ('mean_p_loss', 'conditional_prior_loss', 'w_prior_loss', 'c_prior_loss', 'loss'):
        assert abs(float(got[key]) - ls[key]) <= 2e-4 * max(abs(ls[key]), 1e-3), (key, float(got[key]), ls[key])
    # ReLU-kink flips: the oracle is differentiated with the derivative sides the device took (tests/gpu_util.py: kink_overrides)
    table, flips, worst = kink_overrides(_pairs(eng, cache), math, tag='gmvae_you')
    if flips:
        with onn.act_override(table):
            g = m.backward(p64, x64, out, cache)
        print(f'\n[gmvae_you {h} {math}] {flips} ReLU flips, largest |pre-activation| {worst:.2e} of its site max')
    grads = eng.get_grads()
    for name, _, _ in m.spec:
        assert_close(grads[name].astype(np.float64), g[name], tol=1e-4 if name.endswith('kernel') and '3x3' in name else 5e-4, name=name)
    eng.close()
    with pytest.raises(ValueError):
        GanEngine(h, h, 1, h // 4, zdim=3, max_batch=1, variant='aae', aae_kind='gmvae_you', dim=6)       # dim_z 1 or a multiple of 8


@pytest.mark.parametrize('h,dim_c,dim_z,n', [(32, 6, 1, 2), (64, 9, 1, 3)])
def test_gmvae_you_restore_step_matches_oracle(h, dim_c, dim_z, n):
    m, p32, x, e_w, e_z = _setup(h, dim_c, dim_z, 1, n, seed=5)
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
            gg = ggot.cpu().numpy()
            bad = np.abs(gg - gref) > 3e-4 * np.abs(gref).max()
            assert bad.mean() <= 5e-3, f'{bad.mean():.2e} of the pixels differ'
            if bad.any():      # TV sign flips (multiples of tv_lambda) or a ReLU kink upstream
                q = np.abs(gg - gref)[bad] / tv
                assert np.mean(np.abs(q - np.round(q)) <= 1e-2) >= 0.9
        ref = ref - lr * gref
    assert np.abs(xr.cpu().numpy() - ref).max() <= 8 * lr * tv + 1e-4
    assert np.mean(np.abs(xr.cpu().numpy() - ref)) <= 5e-5
    assert np.array_equal(eng.get_buffer_host(_lib.BUF_GRADS), sentinel)
    eng.close()


def test_gmvae_you_train_trajectory():
    m, p32, x, e_w, e_z = _setup(32, 6, 1, 1, 4, seed=2, perturb=False)
    p64 = _f64(p32)
    opt = m.new_opt(p64)
    eng = _engine(m, 4)
    eng.set_params(p32)
    ref_l, got_l = [], []
    for _ in range(6):
        _, ls, _ = m.train_step(p64, opt, x.astype(np.float64), e_w.astype(np.float64), e_z.astype(np.float64), lr=5e-5)
        ref_l.append(float(ls['loss']))
        out = eng.gm_phase(x, e_w, e_z)
        eng.adam('AE', 5e-5, beta1=0.5, beta2=0.999)
        got_l.append(float(out['loss']))
    np.testing.assert_allclose(got_l, ref_l, rtol=3e-4)
    eng.close()


def test_gmvae_you_trainer(tmp_path):
    """`GMVAE_spatial(sess, config, network=gaussian_mixture_variational_autoencoder_You)` -- the pairing the reference's run.py makes."""
    from unsupervised_anomaly_detection_brain_mri_amd.models import gaussian_mixture_variational_autoencoder_You as net
    from unsupervised_anomaly_detection_brain_mri_amd.trainers import GMVAE_spatial, Phase
    from unsupervised_anomaly_detection_brain_mri_amd.trainers.GMVAE import GMVAE_You
    from unsupervised_anomaly_detection_brain_mri_amd.utils.default_config_setup import get_config, get_options
    from unsupervised_anomaly_detection_brain_mri_amd.utils.synthetic import SyntheticDataset
    h = 32
    opt_ = get_options(batchsize=4, learningrate=2e-4, numEpochs=2, zDim=16, outputWidth=h, outputHeight=h,
                       config={'CHECKPOINTDIR': str(tmp_path / 'ck'), 'SAMPLEDIR': str(tmp_path / 'smp')})
    ds = SyntheticDataset(16, 8, h, h, seed=0)
    cfg = get_config(GMVAE_spatial, opt_, 'ADAM', [8, 8], 0.2, ds)
    cfg.restore_steps = 3
    model = GMVAE_spatial(None, cfg, network=net)
    assert isinstance(model, GMVAE_You) and isinstance(model, GMVAE_spatial)
    assert model.model_dir.startswith('GMVAE_spatial_dSyntheticDataset') and 'gaussian_mixture_variational_autoencoder_You' in model.model_dir
    x = ds.next_batch(4, set='VAL')[0]
    eps = model._draw(4)
    assert eps[0].shape == (4, 8, 8, 1) and eps[1].shape == (4, 8, 8, 1)
    run = model.step(x, Phase.VAL, eps=eps)
    assert set(run) == {'reconstruction', 'L1', 'L2', 'L1_sum', 'L2_sum', 'reconstructionLoss', 'mean_p_loss', 'conditional_prior_loss',
                        'w_prior_loss', 'c_prior_loss', 'loss'}
    m = oy.GMVAEYou(h, 6, 1, 1, 1.0)
    p64 = {k: v.astype(np.float64) for k, v in model.engine.get_params().items()}
    out, _ = m.forward(p64, x.astype(np.float64), eps[0].astype(np.float64), eps[1].astype(np.float64))
    ls = m.losses(x.astype(np.float64), out)
    for k in ('mean_p_loss', 'conditional_prior_loss', 'w_prior_loss', 'c_prior_loss', 'loss'):
        assert run[k] == pytest.approx(ls[k], rel=3e-4, abs=1e-3), k
    model.train(ds)
    assert len(model.curves['TRAIN/loss']) == 2 and model.curves['VAL/loss'][1] < model.curves['VAL/loss'][0]
    # restoration-mode reconstruct (deterministic noise) equals the oracle's loop
    p2 = {k: v.astype(np.float64) for k, v in model.engine.get_params().items()}
    xs = x[:2]
    r = model.reconstruct(xs, eps=0.0)
    ref = xs.astype(np.float64)
    z = (np.zeros((2, 8, 8, 1)), np.zeros((2, 8, 8, 1)))
    for _ in range(3):
        ref = ref - cfg.restore_lr * m.restore_grads(p2, ref, z[0], z[1], cfg.tv_lambda)
    assert np.mean(np.abs(r['reconstruction'] - ref)) <= 5e-5
    model.engine.close()
    with pytest.raises(ValueError):
        cfg.intermediateResolutions = [16, 16]
        GMVAE_spatial(None, cfg, network=net)
