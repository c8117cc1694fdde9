"""GPU: the device noise generator (include/uad_hip.h: uad_rng_fill) against its oracle (oracle/rng.py, Philox4x32-10 known answers in
tests/test_oracle_rng.py); the trainer hot loop that uses it (trainers/AEMODEL.process: device batch gather, device noise, one host
synchronisation per epoch); the ceVAE per-slice reconstruct contract."""
import numpy as np
import pytest
import torch

from oracle import rng as orng

pytestmark = pytest.mark.gpu

try:
    from unsupervised_anomaly_detection_brain_mri_amd.engine import rng_fill
    from unsupervised_anomaly_detection_brain_mri_amd.models import context_encoder_variational_autoencoder, variational_autoencoder
    from unsupervised_anomaly_detection_brain_mri_amd.trainers import VAE, Phase, ceVAE
    from unsupervised_anomaly_detection_brain_mri_amd.utils.default_config_setup import get_config, get_options
    from unsupervised_anomaly_detection_brain_mri_amd.utils.slice_cache import DeviceDataset
    from unsupervised_anomaly_detection_brain_mri_amd.utils.synthetic import SyntheticDataset, synthetic_slices
except Exception:
    pass


@pytest.mark.parametrize('n,per', [(64, 128), (3, 1024), (5, 7), (16, (8, 8, 128))])
def test_rng_fill_matches_the_oracle(n, per):
    seed, step, s0 = 0x1234567890ABCDEF, (1 << 33) + 17, 40
    got = rng_fill([('eps', per, 'normal', 0.0), ('a', per, 'keep', 0.2), ('b', per, 'keep', 0.5)], n, seed, step, s0)
    flat = int(np.prod(per))
    torch.cuda.synchronize()
    e = got['eps'].cpu().numpy().reshape(n, flat)
    ref = orng.normal(n, flat, seed, step, s0, stream=0)
    assert np.abs(e - ref).max() <= 4e-6 * max(1.0, np.abs(ref).max())            # logf / sincosf differ from libm in the last ulps
    assert np.array_equal(got['a'].cpu().numpy().reshape(n, flat), orng.keep_mask(n, flat, 0.2, seed, step, s0, stream=1))      # integer compare: exact
    assert np.array_equal(got['b'].cpu().numpy().reshape(n, flat), orng.keep_mask(n, flat, 0.5, seed, step, s0, stream=2))
    assert tuple(got['eps'].shape) == (n,) + ((per,) if np.isscalar(per) else tuple(per))
    # rank-count invariance: the same global samples drawn in two halves
    if n % 2 == 0:
        h1 = rng_fill([('eps', per, 'normal', 0.0)], n // 2, seed, step, s0)['eps']
        h2 = rng_fill([('eps', per, 'normal', 0.0)], n // 2, seed, step, s0 + n // 2)['eps']
        assert torch.equal(torch.cat([h1, h2]), got['eps'])
    with pytest.raises(ValueError):
        rng_fill([('a', 4, 'keep', 1.0)], 2, 0, 0)


def _cfg(trainer, tmp_path, bs, epochs=1, h=64):
    opt = get_options(batchsize=bs, learningrate=1e-3, numEpochs=epochs, zDim=64, outputWidth=h, outputHeight=h,
                      config={'CHECKPOINTDIR': str(tmp_path / 'ck'), 'SAMPLEDIR': str(tmp_path / 'smp')})
    ds = SyntheticDataset(32, 16, h, h, seed=0)
    return get_config(trainer, opt, 'ADAM', [8, 8], 0.2, ds), opt, ds


def test_process_epoch_on_device_dataset_is_reproducible(tmp_path, capsys):
    """One TRAIN epoch of trainers/VAE.py:76-103 with the batch gathered from an HBM-resident set and the noise drawn on the device:
    two trainers with the same seed end with bit-identical weights and curves; the loop's only host fetch is the epoch's scalar table."""
    imgs = synthetic_slices(48, 64, 64, seed=4)
    sets = np.array([0] * 32 + [1] * 16)
    finals = []
    for _ in range(2):
        cfg, opt, _ = _cfg(VAE, tmp_path, bs=8)
        model = VAE(None, cfg, network=variational_autoencoder, seed=11)
        ds = DeviceDataset(imgs, sets, seed=2)
        out = model.process(ds, 0, Phase.TRAIN)
        val = model.process(ds, 0, Phase.VAL)
        assert set(out) == {'reconstructionLoss', 'kl', 'loss'} and out['loss'] == pytest.approx(out['reconstructionLoss'] + out['kl'], rel=1e-5)
        assert model.noise_step == 4 + 2 and model.engine.step_count == 4
        finals.append((model.engine.get_buffer_host(), out['loss'], val['loss']))
        model.engine.close()
    assert np.array_equal(finals[0][0], finals[1][0]) and finals[0][1:] == finals[1][1:]
    printed = capsys.readouterr().out
    assert printed.count('Epoch (TRAIN): [ 0]') == 8 and 'Epoch (VAL): [ 0] [   1/   2]' in printed
    # host noise stays available (injection tests, device_noise = False)
    cfg, opt, ds = _cfg(VAE, tmp_path, bs=8)
    model = VAE(None, cfg, network=variational_autoencoder, seed=11)
    model.device_noise = False
    assert np.isfinite(model.process(ds, 0, Phase.TRAIN)['loss'])
    model.engine.close()


def test_cevae_batched_reconstruct_equals_slice_by_slice(tmp_path):
    """the anomaly map carries 1/n of the batch mean; Evaluation reconstructs volumes in batches, the reference slice by slice
    (utils/Evaluation.py:246-250).  per_slice=True makes row i of a batched call equal the single-slice call."""
    cfg, opt, ds = _cfg(ceVAE, tmp_path, bs=4)
    model = ceVAE(None, cfg, network=context_encoder_variational_autoencoder)
    model.engine.set_math('f32')
    x = ds.next_batch(4, set='VAL')[0]
    eps = np.random.default_rng(3).standard_normal((4, 64)).astype(np.float32)
    full = model.reconstruct(x, eps=eps, per_slice=True)
    plain = model.reconstruct(x, eps=eps)
    # Row i of the batched call and the single-slice call are the same function of slice i but not the same launches: the planner picks kernels per launch
    # size (round 5: an 8-sample pass runs dec1 on the split spatial kernel, a 2-sample pass on the generic one), and in bf16x3 mode those differ by the
    # mode's round-off.  The property under test -- the 1/n of the batch mean -- is exact arithmetic, so it is tested in the exact-fp32 mode, where only
    # the summation order differs between the two.
    for i in range(4):
        one = model.reconstruct(x[i:i + 1], eps=eps[i:i + 1])
        assert np.abs(full['anomaly'][i] - one['anomaly'][0]).max() <= 2e-5 * np.abs(one['anomaly']).max() + 1e-12
        assert np.abs(full['reconstruction'][i] - one['reconstruction'][0]).max() <= 1e-5
        assert np.abs(plain['anomaly'][i] * 4 - one['anomaly'][0]).max() <= 2e-5 * np.abs(one['anomaly']).max() + 1e-12
    assert model.RECONSTRUCT_PER_SLICE
    model.engine.close()
