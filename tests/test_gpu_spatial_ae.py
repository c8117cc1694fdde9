"""GPU parity of the spatial autoencoder (models/autoencoder_spatial.py, UAD_ARCH_AE_SPATIAL) through the C-ABI vs the numpy oracle:
forward, loss, every gradient, one Adam step, with and without the dropout mask on the latent feature map; trainer surface."""
import numpy as np
import pytest

from oracle import nn as onn
from oracle import vae as ovae

pytestmark = pytest.mark.gpu
TOL = 1e-4      # max-norm relative, fp32 device vs fp64 oracle


def _rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize('h,inter,n,math,drop', [(32, 8, 3, 'f32', True), (64, 8, 2, 'bf16x3', False), (128, 8, 2, 'bf16x3', True)])
def test_spatial_ae_step_matches_oracle(h, inter, n, math, drop):
    from unsupervised_anomaly_detection_brain_mri_amd.engine import Engine
    m = ovae.SpatialAE(h, h, 1, inter)
    p = ovae.init_params(m.spec, seed=6, dtype=np.float64, perturb=True)
    x = ovae.synthetic_slices(n, h, h, seed=2, dtype=np.float64)
    eng = Engine('AE_spatial', h, h, 1, inter, 8, max_batch=n, math=math)
    assert [(k, tuple(s)) for k, s, _ in eng.spec] == [(k, tuple(s)) for k, s, _ in m.spec]
    eng.set_params(p)
    cenc = eng._cenc()
    masks = {'z': onn.make_dropout_mask(np.random.default_rng(1), (n, inter, inter, cenc), 0.2, np.float64)} if drop else None
    out = eng.forward(x, None, masks, want_backward=True)
    eng.backward()
    o_ref, cache = m.forward(p, x, masks)
    ls = m.losses(x, o_ref)
    g_ref = m.backward(p, x, o_ref, cache, masks)
    assert _rel(out['x_hat'].cpu().numpy(), o_ref['x_hat']) < TOL
    assert _rel(out['z'].cpu().numpy(), o_ref['z']) < TOL
    assert _rel(out['L1'].cpu().numpy(), ls['L1']) < TOL
    sc = out['scalars'].cpu().numpy()
    assert abs(sc[0] - ls['reconstructionLoss']) < TOL * ls['reconstructionLoss'] and abs(sc[2] - ls['loss']) < TOL * ls['loss']
    g = eng.get_grads()
    scale = max(np.abs(v).max() for v in g_ref.values())
    for name, shape, _ in m.spec:
        err = np.abs(g[name] - g_ref[name]).max()
        assert err <= TOL * max(np.abs(g_ref[name]).max(), 1e-2 * scale), f'{name}: {err:.3e} vs {np.abs(g_ref[name]).max():.3e}'
    p32 = {k: v.astype(np.float32) for k, v in p.items()}
    before = eng.get_params()
    eng.adam_step(1e-3)
    after = eng.get_params()
    for name, _, _ in m.spec:
        big = np.abs(g_ref[name]) > 1e-2 * np.abs(g_ref[name]).max()
        np.testing.assert_allclose((after[name] - before[name])[big], -1e-3 * np.sign(g_ref[name][big]), rtol=5e-2, atol=1e-7, err_msg=name)
    eng.close()


def test_spatial_ae_trainer(tmp_path):
    from unsupervised_anomaly_detection_brain_mri_amd.models import autoencoder_spatial
    from unsupervised_anomaly_detection_brain_mri_amd.trainers import AE, Phase
    from unsupervised_anomaly_detection_brain_mri_amd.utils.default_config_setup import get_config, get_options
    from unsupervised_anomaly_detection_brain_mri_amd.utils.synthetic import SyntheticDataset
    opt = get_options(batchsize=8, learningrate=2e-4, numEpochs=2, zDim=64, outputWidth=64, outputHeight=64,
                      config={'CHECKPOINTDIR': str(tmp_path / 'ck'), 'SAMPLEDIR': str(tmp_path / 'smp')})
    ds = SyntheticDataset(32, 16, 64, 64, seed=0)
    cfg = get_config(AE, opt, 'ADAM', [8, 8], 0.2, ds)
    model = AE(None, cfg, network=autoencoder_spatial)
    assert model.model_dir == 'AE_dSyntheticDataset_s64x64_autoencoder_spatial_b8_z64_'
    assert not any(n.startswith('Bottleneck') for n, _, _ in model.engine.spec)
    model.train(ds)
    tr = model.curves['TRAIN/loss']
    assert len(tr) == 2 and tr[1] < tr[0]
    run = model.step(ds.next_batch(8, set='VAL')[0], Phase.VAL)
    assert set(run) == {'reconstruction', 'L1', 'reconstructionLoss', 'loss'} and run['loss'] == run['reconstructionLoss']
    r = model.reconstruct(ds.next_batch(1, set='VAL')[0][0])
    assert r['reconstruction'].shape == (1, 64, 64, 1)
    model.engine.close()
