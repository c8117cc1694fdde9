"""GPU: restoration mode on a VAE handle (trainers/VAE_You.py): `grads` = d(rec_n + kl_n + tv * TV_n(x - x_hat))/dx and the in-place
update, chained steps vs the numpy oracle (TV sign ties handled like tests/test_gpu_gmvae.py); trainer surface."""
import numpy as np
import pytest
import torch

from oracle import vae as ovae

pytestmark = pytest.mark.gpu

try:
    from unsupervised_anomaly_detection_brain_mri_amd import _lib
except Exception:
    pass


@pytest.mark.parametrize('h,zdim,n,math', [(64, 32, 2, 'f32'), (128, 128, 4, 'bf16x3')])
def test_vae_restore_step_matches_oracle(h, zdim, n, math):
    from unsupervised_anomaly_detection_brain_mri_amd.engine import Engine
    m = ovae.Model('VAE', h, h, 1, 8, zdim)
    p = ovae.init_params(m.spec, seed=8, dtype=np.float64, perturb=True)
    x = ovae.synthetic_slices(n, h, h, seed=4, dtype=np.float64)
    rng = np.random.default_rng(6)
    eng = Engine('VAE', h, h, 1, 8, zdim, max_batch=n, math=math)
    eng.set_params(p)
    sentinel = np.full(eng.nparams, 3.0, np.float32)
    eng.set_buffer_host(_lib.BUF_GRADS, sentinel)
    xr = torch.from_numpy(x.astype(np.float32)).cuda()
    ref = x.copy()
    lr, tv = 2e-4, 1.8          # per-sample objective: gradients are ~N x larger than the GMVAE's mean-loss ones
    for step in range(3):
        eps = rng.standard_normal((n, zdim))
        gref = m.restore_grads(p, ref, eps, tv)
        ggot = eng.restore_step(xr, None, eps.astype(np.float32), tv_lambda=tv, restore_lr=lr, want_grads=True)
        torch.cuda.synchronize()
        if step == 0:
            gg = ggot.cpu().numpy()
            bad = np.abs(gg - gref) > 3e-4 * np.abs(gref).max()
            assert bad.mean() <= 2e-3, f'{bad.mean():.2e} of the pixels differ'
            if bad.any():      # differences are TV / L1 sign decisions at round-off ties: multiples of tv_lambda (or 2 for the L1 sign)
                q = np.abs(gg - gref)[bad]
                assert (np.minimum(np.abs(q / tv - np.round(q / tv)), np.abs(q - np.round(q))) <= 2e-2).all()
        ref = ref - lr * gref
    assert np.abs(xr.cpu().numpy() - ref).max() <= 8 * lr * (tv + 1.0) + 1e-4
    assert np.mean(np.abs(xr.cpu().numpy() - ref)) <= 2e-5
    assert np.array_equal(eng.get_buffer_host(_lib.BUF_GRADS), sentinel)     # no parameter gradient was written
    eng.close()


def test_vae_you_trainer(tmp_path):
    from unsupervised_anomaly_detection_brain_mri_amd.models import variational_autoencoder
    from unsupervised_anomaly_detection_brain_mri_amd.trainers import VAE_You
    from unsupervised_anomaly_detection_brain_mri_amd.utils.default_config_setup import get_config, get_options
    from unsupervised_anomaly_detection_brain_mri_amd.utils.synthetic import SyntheticDataset
    d = VAE_You.Config()
    assert (d.modelname, d.restore_lr, d.restore_steps, d.tv_lambda) == ('VAE_You', 1e-3, 150, 1.8)
    opt = get_options(batchsize=8, learningrate=2e-4, numEpochs=1, zDim=64, outputWidth=64, outputHeight=64,
                      config={'CHECKPOINTDIR': str(tmp_path / 'ck'), 'SAMPLEDIR': str(tmp_path / 'smp')})
    ds = SyntheticDataset(32, 16, 64, 64, seed=0)
    cfg = get_config(VAE_You, opt, 'ADAM', [8, 8], 0.2, ds)
    cfg.restore_steps = 5
    model = VAE_You(None, cfg, network=variational_autoencoder)
    assert model.model_dir == 'VAE_You_dSyntheticDataset_s64x64_variational_autoencoder_b8_z64_'
    model.train(ds)
    x = ds.next_batch(3, set='VAL')[0]
    # reconstruct() = the oracle's restoration loop with the same (zero) noise
    m = ovae.Model('VAE', 64, 64, 1, 8, 64)
    p = {k: v.astype(np.float64) for k, v in model.engine.get_params().items()}
    ref = m.restore(p, x.astype(np.float64), lambda step: np.zeros((3, 64)), restore_steps=5, restore_lr=1e-3, tv_lambda=1.8)
    r = model.reconstruct(x, eps=0.0)
    assert r['reconstruction'].shape == (3, 64, 64, 1)
    assert np.mean(np.abs(r['reconstruction'] - ref)) <= 5e-5 and np.abs(r['reconstruction'] - ref).max() <= 10 * 1e-3 * 2.8 + 1e-4
    g = model.restore_gradients(x, eps=np.zeros((3, 64), np.float32))
    assert g.shape == x.shape and np.isfinite(g).all()
    model.restore_steps = 2
    ds2 = SyntheticDataset(8, 40, 64, 64, seed=1)
    model.determine_best_lambda(ds2)
    assert 0.0 <= model.tv_lambda_value <= 1.9
    model.engine.close()
