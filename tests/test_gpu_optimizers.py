"""GPU: the non-Adam optimizers of DLMODEL.create_optimizer (SGD, MOMENTUM, RMS; trainers/DLMODEL.py:113-123) on the fused AE-family handle
(uad_optimizer_step) against the oracle's TF-1.15 update rules over a few VAE train steps, and through the trainer (config.optimizer)."""
import numpy as np
import pytest
import torch

from oracle import nn as onn
from oracle import vae as ovae

pytestmark = pytest.mark.gpu

try:
    from unsupervised_anomaly_detection_brain_mri_amd import _lib
    from unsupervised_anomaly_detection_brain_mri_amd.engine import Engine
except Exception:
    Engine = None


@pytest.mark.parametrize('kind', ['SGD', 'MOMENTUM', 'RMS'])
def test_optimizer_trajectory_matches_oracle(kind):
    h, zd, n, lr = 32, 16, 4, 2e-5
    m = ovae.Model('VAE', h, h, 1, 8, zd)
    p32 = ovae.init_params(m.spec, seed=3, perturb=True)
    p = {k: v.astype(np.float64) for k, v in p32.items()}
    x = ovae.synthetic_slices(n, h, h, seed=0)
    eps = np.random.default_rng(1).standard_normal((n, zd)).astype(np.float32)
    s1 = {k: np.zeros_like(v) for k, v in p.items()}
    s2 = {k: np.ones_like(v) for k, v in p.items()}
    eng = Engine('VAE', h, h, 1, 8, zd, max_batch=n, math='f32')
    eng.set_params(p32)
    eng.set_optimizer(kind, momentum=0.9)
    if kind == 'RMS':
        assert np.all(eng.get_buffer_host(_lib.BUF_ADAM_V) == 1.0)          # TF's `rms` slot initialisation
    got_l, ref_l = [], []
    for _ in range(4):
        out, cache = m.forward(p, x.astype(np.float64), eps.astype(np.float64), None)
        ref_l.append(m.losses(x.astype(np.float64), out)['loss'])
        g = m.backward(p, x.astype(np.float64), out, cache, None)
        for name, _, _ in m.spec:
            if kind == 'SGD':
                onn.sgd_tf_step(p[name], g[name], lr)
            elif kind == 'MOMENTUM':
                onn.momentum_tf_step(p[name], g[name], s1[name], lr, 0.9)
            else:
                onn.rmsprop_tf_step(p[name], g[name], s2[name], s1[name], lr, 0.9)
        got_l.append(float(eng.train_step(x, eps, None, lr=lr)['scalars'][2]))
    torch.cuda.synchronize()
    np.testing.assert_allclose(got_l, ref_l, rtol=3e-4)
    assert eng.step_count == 4
    flat = eng.get_buffer_host(_lib.BUF_PARAMS)
    ref = np.concatenate([p[nm].reshape(-1) for nm, _, _ in m.spec])
    start = np.concatenate([p32[nm].reshape(-1) for nm, _, _ in m.spec])
    moved = np.abs(ref - start).max()
    assert moved > 0 and np.abs(flat - ref).max() <= 2e-3 * moved + 1e-7, (np.abs(flat - ref).max(), moved)
    eng.close()


def test_trainer_with_rms(tmp_path):
    from unsupervised_anomaly_detection_brain_mri_amd.models import variational_autoencoder
    from unsupervised_anomaly_detection_brain_mri_amd.trainers import VAE, fAnoGAN
    from unsupervised_anomaly_detection_brain_mri_amd.models import fanogan
    from unsupervised_anomaly_detection_brain_mri_amd.utils.default_config_setup import get_config, get_options
    from unsupervised_anomaly_detection_brain_mri_amd.utils.synthetic import SyntheticDataset
    opt = get_options(batchsize=4, learningrate=1e-5, numEpochs=2, zDim=16, outputWidth=32, outputHeight=32,
                      config={'CHECKPOINTDIR': str(tmp_path / 'ck'), 'SAMPLEDIR': str(tmp_path / 'smp')})
    ds = SyntheticDataset(16, 8, 32, 32, seed=0)
    cfg = get_config(VAE, opt, 'RMS', [8, 8], 0.2, ds)
    model = VAE(None, cfg, network=variational_autoencoder)
    model.train(ds)
    assert model.engine.optimizer == 'RMS' and len(model.curves['TRAIN/loss']) == 2 and np.isfinite(model.curves['VAL/loss']).all()
    assert model.curves['VAL/loss'][1] < model.curves['VAL/loss'][0]
    model.engine.close()
    with pytest.raises(ValueError, match='Invalid optimizer type'):
        VAE(None, get_config(VAE, opt, 'RMSProp', [8, 8], 0.2, ds), network=variational_autoencoder).train(ds)
    cfg2 = get_config(fAnoGAN, opt, 'SGD', [8, 8], 0.2, ds)
    g = fAnoGAN(None, cfg2, network=fanogan)
    with pytest.raises(NotImplementedError):          # the WGAN trainers build their own Adam optimizers; config.optimizer other than ADAM has no meaning there
        g.train(ds)
    g.engine.close()
