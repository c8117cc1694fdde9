"""GPU: the reference's trainer surface (trainers/AE.py, trainers/VAE.py: Config / train / process / reconstruct,
model_dir, checkpoint resume) and the evaluation driver on top of the HIP engine."""
import os

import numpy as np
import pytest

from oracle import vae as ovae
from tests.gpu_util import assert_grads_close, device_activation_pattern

pytestmark = pytest.mark.gpu

try:
    from unsupervised_anomaly_detection_brain_mri_amd import _lib
    from unsupervised_anomaly_detection_brain_mri_amd.models import autoencoder, variational_autoencoder
    from unsupervised_anomaly_detection_brain_mri_amd.trainers import AE, VAE, Phase
    from unsupervised_anomaly_detection_brain_mri_amd.utils import Evaluation
    from unsupervised_anomaly_detection_brain_mri_amd.utils.default_config_setup import get_config, get_options
    from unsupervised_anomaly_detection_brain_mri_amd.utils.synthetic import SyntheticDataset, synthetic_slices
except Exception:
    pass


def _config(trainer, tmp_path, h=64, bs=8, epochs=2):
    opt = get_options(batchsize=bs, learningrate=1e-3, numEpochs=epochs, zDim=64, outputWidth=h, outputHeight=h,
                      config={'CHECKPOINTDIR': str(tmp_path / 'ck'), 'SAMPLEDIR': str(tmp_path / 'smp')})
    ds = SyntheticDataset(32, 16, h, h, seed=0)
    return get_config(trainer, opt, 'ADAM', [8, 8], 0.2, ds), opt, ds


def test_vae_train_process_reconstruct_resume(tmp_path):
    cfg, opt, ds = _config(VAE, tmp_path)
    cfg.learningrate = 3e-4          # (1e-3 makes Adam's first steps on the freshly initialised net spike: not a premise for "the objective falls")
    model = VAE(None, cfg, network=variational_autoencoder)
    assert model.network.__name__ == 'variational_autoencoder'
    assert model.model_dir == 'VAE_dSyntheticDataset_s64x64_variational_autoencoder_b8_z64_'
    xv = ds.next_batch(8, set='VAL')[0]
    before = model.step(xv, Phase.VAL, eps=np.zeros((8, 64), np.float32))['loss']
    model.train(ds)
    tr = model.curves['TRAIN/loss']
    # (the epoch means carry the step's own noise -- eps and dropout are drawn on the device; the objective is judged on a fixed batch with eps = 0)
    assert len(tr) == 2 and model.step(xv, Phase.VAL, eps=np.zeros((8, 64), np.float32))['loss'] < before
    assert set(model.curves) >= {'TRAIN/loss', 'TRAIN/kl', 'TRAIN/reconstructionLoss', 'VAL/loss'}
    ck = os.path.join(model.checkpointDir, model.model_dir)
    assert os.path.isfile(os.path.join(ck, 'VAE.model-2.npz')) and os.path.isfile(os.path.join(ck, 'Config-2.json'))
    # reconstruct(): 3-D input is expanded, keys / shapes as trainers/VAE.py:105-123
    x = ds.next_batch(1, set='VAL')[0][0]
    r = model.reconstruct(x, eps=0.0)
    assert r['reconstruction'].shape == (1, 64, 64, 1) and r['l1err'] == pytest.approx(r['l2err'], rel=1e-6)
    # step() returns the reference fetch keys
    run = model.step(ds.next_batch(8, set='VAL')[0], Phase.VAL)
    assert set(run) == {'reconstruction', 'L1', 'reconstructionLoss', 'kl', 'loss'}
    assert run['loss'] == pytest.approx(run['reconstructionLoss'] + run['kl'], rel=1e-5)
    # resume: a fresh trainer picks up epoch 2 and the same weights
    w = model.engine.get_buffer_host(_lib.BUF_PARAMS)
    model.engine.close()
    cfg2, _, _ = _config(VAE, tmp_path)
    m2 = VAE(None, cfg2, network=variational_autoencoder, seed=123)
    assert m2.load_checkpoint() == 2
    assert np.array_equal(m2.engine.get_buffer_host(_lib.BUF_PARAMS), w) and m2.engine.step_count == 8
    m2.engine.close()


def test_ae_step_matches_oracle_with_injected_masks(tmp_path):
    cfg, opt, ds = _config(AE, tmp_path, h=64, bs=4)
    model = AE(None, cfg, network=autoencoder)
    m = ovae.Model('AE', 64, 64, 1, 8, 64)
    p = {k: v.astype(np.float64) for k, v in model.engine.get_params().items()}
    x = ds.next_batch(4, set='TRAIN')[0]
    mask = {'z': (np.random.default_rng(0).random((4, 64)) >= 0.2).astype(np.float32) / 0.8}
    out, _ = m.forward(p, x.astype(np.float64), None, {'z': mask['z'].astype(np.float64)})
    ls = m.losses(x.astype(np.float64), out)
    run = model.step(x, Phase.TRAIN, dropout_masks=mask)
    assert set(run) == {'reconstruction', 'L1', 'reconstructionLoss', 'loss'}
    assert run['loss'] == pytest.approx(ls['loss'], rel=1e-4)
    assert np.abs(run['reconstruction'] - out['x_hat']).max() <= 1e-4 * np.abs(out['x_hat']).max()
    with pytest.raises(ValueError):
        AE(None, cfg, network=variational_autoencoder)      # trainer / network mismatch
    model.engine.close()


def test_evaluation_driver_runs_and_scores(tmp_path):
    cfg, opt, ds = _config(VAE, tmp_path, h=64, bs=8, epochs=1)
    model = VAE(None, cfg, network=variational_autoencoder)
    model.train(ds)
    vols, labs, masks = [], [], []
    for pth in range(2):
        x, lab, msk = synthetic_slices(12, 64, 64, seed=70 + pth, lesions=True)
        vols.append(x[..., 0].astype(np.float64)); labs.append(lab); masks.append(msk)
    ev = Evaluation.evaluate_arrays(vols, labs, masks, model, opt, eps=0.0)
    assert 0.0 <= ev['diff_AUPRC'] <= 1.0 and 0.0 <= ev['diff_AUC'] <= 1.0 and len(ev['Dice']) == 2
    # residual volume of one patient vs the numpy formula on the same reconstructions
    d, l1 = Evaluation.evaluate_volume(model, vols[0], masks[0], {**opt, 'medianFiltering': False})
    assert d.shape == vols[0].shape and (d >= 0).all() and np.isfinite(l1).all()
    model.engine.close()


def test_cevae_trainer_surface_and_oracle_step(tmp_path):
    """trainers/ceVAE.py: Config, train/process (context masking on the host, x_ce = batch outside TRAIN), the fetch keys
    of one step, reconstruct() with gradient-based restoration -- and one injected-RNG step against the oracle."""
    from unsupervised_anomaly_detection_brain_mri_amd.models import context_encoder_variational_autoencoder as net
    from unsupervised_anomaly_detection_brain_mri_amd.trainers import ceVAE
    from unsupervised_anomaly_detection_brain_mri_amd.trainers.CE import retrieve_masked_batch
    cfg, opt, ds = _config(ceVAE, tmp_path, h=64, bs=4, epochs=2)
    assert cfg.modelname == 'ceVAE' and cfg.use_gradient_based_restoration is True
    cfg.learningrate = 2e-4
    model = ceVAE(None, cfg, network=net)
    assert model.model_dir == 'ceVAE_dSyntheticDataset_s64x64_context_encoder_variational_autoencoder_b4_z64_'

    # one TRAIN step with injected eps / masks / masked batch == oracle ce_train_step on the same weights
    m = ovae.CeVAE(64, 64, 1, 8, 64)
    p64 = {k: np.asarray(v, np.float64) for k, v in model.engine.get_params().items()}
    opt_state = m.new_opt(p64)
    batch, _, bm = ds.next_batch(4, set='TRAIN', return_brainmask=True)
    import random
    xce = retrieve_masked_batch(batch, bm, random.Random(3))
    assert (xce == 0).sum() > (batch == 0).sum()
    eps, masks = model._draw(4, dropout=True)
    assert set(masks) == {'mu', 'mu_ce', 'sigma', 'dec', 'dec_ce'}
    run = model.step(batch, Phase.TRAIN, masked_batch=xce, eps=eps, dropout_masks=masks)
    assert set(run) == {'reconstruction', 'reconstruction_ce', 'L1_vae', 'L1_ce', 'L1', 'Rec_ce', 'Rec_vae',
                        'reconstructionLoss', 'kl', 'loss', 'loss_vae', 'anomaly'}
    _, ls, g = m.ce_train_step(p64, opt_state, batch.astype(np.float64), xce.astype(np.float64), eps.astype(np.float64),
                               {k: v.astype(np.float64) for k, v in masks.items()}, lr=cfg.learningrate, beta1=0.5)
    for k in ('Rec_ce', 'Rec_vae', 'reconstructionLoss', 'kl', 'loss', 'loss_vae'):
        assert run[k] == pytest.approx(ls[k], rel=1e-4), k
    assert np.abs(run['anomaly'] - g['anomaly']).max() <= 3e-4 * np.abs(g['anomaly']).max()
    assert np.abs(run['L1'] - ls['L1']).max() <= 2e-4 * np.abs(ls['L1']).max()
    flat = model.engine.get_buffer_host(_lib.BUF_PARAMS)
    ref = ovae.flatten_params(m.spec, p64)
    # first Adam step = lr * g / (|g| + eps): entries whose gradient is ~0 (|g| ~ 1e-8) move by up to +-lr whatever their
    # rounding, so the bound is 2 * lr relative to max|param| = 1 (the gradients themselves are held to 1e-4 in test_gpu_cevae.py)
    assert np.abs(flat - ref).max() <= 2.5 * cfg.learningrate * np.abs(ref).max()
    assert np.mean(np.abs(flat - ref)) <= 1e-6

    # VAL step: x_ce = batch, no dropout, anomaly still fetched (it sits in self.losses)
    v = model.step(batch, Phase.VAL, masked_batch=xce, eps=eps)
    out, caches = m.ce_forward(p64, batch.astype(np.float64), batch.astype(np.float64), eps.astype(np.float64))
    lv = m.ce_losses(batch.astype(np.float64), batch.astype(np.float64), out)
    assert v['loss'] == pytest.approx(lv['loss'], rel=2e-4) and v['Rec_ce'] == pytest.approx(lv['Rec_ce'], rel=2e-4)

    # train(): epoch loop with host masking, checkpoints, early-stopping bookkeeping
    model.train(ds)
    assert len(model.curves['TRAIN/loss']) == 2 and len(model.curves['VAL/loss_vae']) == 2
    # TRAIN means are noisy (fresh random context holes + dropout every step); VAL runs x_ce = x without dropout
    assert model.curves['VAL/loss'][1] < model.curves['VAL/loss'][0]
    ck = os.path.join(model.checkpointDir, model.model_dir)
    assert os.path.isfile(os.path.join(ck, 'ceVAE.model-2.npz'))

    # reconstruct(): 'reconstruction' = x - anomaly (use_gradient_based_restoration), l1err = sum |anomaly|
    x = ds.next_batch(2, set='VAL')[0]
    r = model.reconstruct(x, eps=0.0)
    assert np.allclose(r['reconstruction'], x - r['anomaly'])
    assert r['l1err'] == pytest.approx(np.abs(r['anomaly']).sum(), rel=1e-5)
    model.config.use_gradient_based_restoration = False
    r0 = model.reconstruct(x, eps=0.0)
    assert not np.allclose(r0['reconstruction'], r['reconstruction'])
    model.engine.close()


def test_gmvae_spatial_trainer_surface(tmp_path):
    """trainers/GMVAE_spatial.py: Config defaults, train/process, step() fetch keys, reconstruct() = restoration loop on
    the device (matches the oracle loop with injected noise), restore_steps == 0 path, determine_best_lambda."""
    from oracle import gmvae as og
    from unsupervised_anomaly_detection_brain_mri_amd.models import gaussian_mixture_variational_autoencoder_spatial as net
    from unsupervised_anomaly_detection_brain_mri_amd.trainers import GMVAE_spatial
    d = GMVAE_spatial.Config()
    assert (d.dim_c, d.dim_z, d.dim_w, d.c_lambda, d.restore_lr, d.restore_steps, d.tv_lambda) == (6, 1, 1, 1, 1e-3, 150, 1.8)
    cfg, opt, ds = _config(GMVAE_spatial, tmp_path, h=64, bs=4, epochs=2)
    cfg.learningrate = 5e-5
    cfg.dim_c, cfg.restore_steps, cfg.restore_lr = 9, 4, 5e-3
    model = GMVAE_spatial(None, cfg, network=net)
    assert model.model_dir == 'GMVAE_spatial_dSyntheticDataset_s64x64_gaussian_mixture_variational_autoencoder_spatial_b4_z64_'
    p = model.engine.get_params()
    assert np.allclose(p['Variable'], 0.1) and np.allclose(p['batch_normalization_3/gamma'], 1.0)

    batch = ds.next_batch(4, set='VAL')[0]
    eps = model._draw(4)
    run = model.step(batch, Phase.VAL, eps=eps)
    assert set(run) == {'reconstruction', 'L1', 'L2', 'L1_sum', 'L2_sum', 'reconstructionLoss', 'mean_p_loss',
                        'conditional_prior_loss', 'w_prior_loss', 'c_prior_loss', 'loss'}
    m = og.GMVAE(64, 64, 1, 8, 9, 1, 1, 1.0)
    p64 = {k: np.asarray(v, np.float64) for k, v in p.items()}
    out, _ = m.forward(p64, batch.astype(np.float64), eps[0].astype(np.float64), eps[1].astype(np.float64))
    ls = m.losses(batch.astype(np.float64), out)
    for k in ('mean_p_loss', 'conditional_prior_loss', 'w_prior_loss', 'c_prior_loss', 'loss'):
        assert run[k] == pytest.approx(ls[k], rel=2e-4, abs=1e-3), k
    assert run['L2_sum'] == pytest.approx(ls['L2_sum'], rel=1e-3)

    # reconstruct(): restore_steps x in-place update on device == oracle loop with the same injected noise
    r = model.reconstruct(batch[:2], eps=(eps[0][:2], eps[1][:2]))
    noise = lambda step: (eps[0][:2].astype(np.float64), eps[1][:2].astype(np.float64))
    ref = m.reconstruct(p64, batch[:2].astype(np.float64), noise, restore_steps=4, restore_lr=5e-3, tv_lambda=1.8)
    assert np.mean(np.abs(r['reconstruction'] - ref['reconstruction'])) <= 1e-5
    assert r['l1err'] == pytest.approx(ref['l1err'], rel=5e-3)
    g = model.restore_gradients(batch[:2], eps=(eps[0][:2], eps[1][:2]))
    gref = m.restore_grads(p64, batch[:2].astype(np.float64), *noise(0), 1.8)
    assert np.abs(g - gref).max() <= 3e-4 * np.abs(gref).max()
    model.restore_steps = 0
    r0 = model.reconstruct(batch[:2], eps=(eps[0][:2], eps[1][:2]))
    assert np.abs(r0['reconstruction'] - out['xz_mu'][:2]).max() <= 1e-4 * np.abs(out['xz_mu']).max()
    model.restore_steps = 4

    model.train(ds)
    assert len(model.curves['TRAIN/loss']) == 2 and model.curves['VAL/loss'][1] < model.curves['VAL/loss'][0]
    assert os.path.isfile(os.path.join(model.checkpointDir, model.model_dir, 'GMVAE_spatial.model-2.npz'))
    # tv_lambda == -1: lambda sweep on 20 % of the VAL batches (GMVAE_spatial.py:201-225)
    model.restore_steps = 2
    ds16 = SyntheticDataset(8, 20, 64, 64, seed=3)
    model.determine_best_lambda(ds16)
    assert 0.0 <= model.tv_lambda_value <= 1.9
    model.engine.close()


def test_fanogan_trainer_surface(tmp_path):
    """trainers/fAnoGAN.py: Config defaults, the two training stages (WGAN epochs: 1 generator + 5 critic steps per batch;
    encoder epochs with validation), the fetch keys of the three sess.runs, reconstruct(), checkpoint resume with the three
    Adam step counters, and the evaluation driver on top."""
    from oracle import fanogan as ofa
    from unsupervised_anomaly_detection_brain_mri_amd.models import fanogan
    from unsupervised_anomaly_detection_brain_mri_amd.trainers import fAnoGAN
    d = fAnoGAN.Config()
    assert (d.modelname, d.scale, d.kappa) == ('fAnoGAN', 10.0, 1.0)
    cfg, opt, ds = _config(fAnoGAN, tmp_path, h=64, bs=4, epochs=1)
    cfg.dropout_rate = 0.0
    model = fAnoGAN(None, cfg, network=fanogan)
    assert model.model_dir == 'fAnoGAN_dSyntheticDataset_s64x64_fanogan_b4_z64_'
    assert [n for n, _, _ in model.engine.spec] == [n for n, _, _ in ofa.param_spec(64, 8, 64)]
    # one critic step against the oracle with injected z / alpha (before any training)
    batch = ds.next_batch(4, set='TRAIN')[0]
    p = {k: v.astype(np.float64) for k, v in model.engine.get_params().items()}
    z = model.sample_z(4)
    alpha = np.linspace(0.1, 0.9, 4).astype(np.float32)
    m = ofa.FAnoGAN(64, 8, 64)
    ls, _ = m.disc_phase(p, batch.astype(np.float64), z.astype(np.float64), alpha.astype(np.float64))
    run = model.discriminator_step(batch, z=z, alpha=alpha)
    assert set(run) == {'generated', 'disc_loss', 'disc_fake', 'disc_real'}
    for k in ('disc_loss', 'disc_fake', 'disc_real'):
        assert run[k] == pytest.approx(ls[k], rel=1e-4, abs=1e-5)
    run = model.generator_step(batch)
    assert set(run) == {'generated', 'gen_loss'} and tuple(run['generated'].shape) == (4, 64, 64, 1)
    model.train(ds)
    nb = ds.num_batches(4, set='TRAIN')
    assert model.engine.step_count('Generator') == nb + 1 and model.engine.step_count('Discriminator') == 5 * nb + 1
    assert model.engine.step_count('Encoder') == nb
    assert {'TRAIN/wgan_gen_loss', 'TRAIN/wgan_disc_loss', 'TRAIN/enc_loss', 'TRAIN/reconstructionLoss', 'VAL/reconstructionLoss'} <= set(model.curves)
    run = model.step(ds.next_batch(4, set='VAL')[0], Phase.VAL)
    assert set(run) == {'loss_img', 'loss_fts', 'enc_loss', 'reconstructionLoss', 'loss', 'reconstruction', 'L1', 'z_enc'}
    assert run['enc_loss'] == pytest.approx(run['loss_img'] + run['loss_fts'], rel=1e-5)
    x = ds.next_batch(1, set='VAL')[0][0]
    r = model.reconstruct(x)
    assert r['reconstruction'].shape == (1, 64, 64, 1) and 0.0 <= r['reconstruction'].min() and r['reconstruction'].max() <= 1.0
    ck = os.path.join(model.checkpointDir, model.model_dir)
    assert os.path.isfile(os.path.join(ck, 'fAnoGAN.model-2.npz'))
    w = model.engine.get_buffer_host(_lib.BUF_PARAMS)
    steps = [model.engine.step_count(g) for g in ('Encoder', 'Generator', 'Discriminator')]
    vols = [synthetic_slices(8, 64, 64, seed=60, lesions=True)]
    ev = Evaluation.evaluate_arrays([v[0][..., 0].astype('float64') for v in vols], [v[1] for v in vols], [v[2] for v in vols], model, opt)
    assert 0.0 <= ev['diff_AUC'] <= 1.0 and 0.0 <= ev['diff_AUPRC'] <= 1.0
    model.engine.close()
    cfg2, _, _ = _config(fAnoGAN, tmp_path, h=64, bs=4, epochs=1)
    m2 = fAnoGAN(None, cfg2, network=fanogan, seed=5)
    assert m2.load_checkpoint() == 2
    assert np.array_equal(m2.engine.get_buffer_host(_lib.BUF_PARAMS), w)
    assert [m2.engine.step_count(g) for g in ('Encoder', 'Generator', 'Discriminator')] == steps
    m2.engine.close()


def test_fanogan_schlegl_trainer(tmp_path):
    """The same trainer on the ResNet graph (network=fanogan_schlegl): parameter table, one critic step against the oracle,
    one epoch of each stage, reconstruct() in [-1, 1] (tanh output)."""
    from oracle import fanogan_schlegl as ofs
    from unsupervised_anomaly_detection_brain_mri_amd.models import fanogan_schlegl
    from unsupervised_anomaly_detection_brain_mri_amd.trainers import fAnoGAN
    cfg, opt, ds = _config(fAnoGAN, tmp_path, h=64, bs=2, epochs=1)
    model = fAnoGAN(None, cfg, network=fanogan_schlegl)
    assert model.model_dir == 'fAnoGAN_dSyntheticDataset_s64x64_fanogan_schlegl_b2_z64_'
    m = ofs.FAnoGANSchlegl(64, 8, 64, 64)
    assert [(n, tuple(s)) for n, s, _ in model.engine.spec] == [(n, tuple(s)) for n, s, _ in m.spec]
    batch = ds.next_batch(2, set='TRAIN')[0]
    run = model.discriminator_step(batch)
    assert set(run) == {'generated', 'disc_loss', 'disc_fake', 'disc_real'} and np.isfinite(run['disc_loss'])
    ds2 = SyntheticDataset(4, 2, 64, 64, seed=1)
    model.train(ds2)
    assert model.engine.step_count('Generator') == 2 and model.engine.step_count('Discriminator') == 11 and model.engine.step_count('Encoder') == 2
    r = model.reconstruct(ds2.next_batch(1, set='VAL')[0][0])
    assert r['reconstruction'].shape == (1, 64, 64, 1) and -1.0 <= r['reconstruction'].min() and r['reconstruction'].max() <= 1.0
    model.engine.close()


def test_anovaegan_trainer(tmp_path):
    """trainers/AnoVAEGAN.py surface: Config defaults, one epoch (VAE + generator + 5 critic steps per batch, then VAL), fetch keys,
    reconstruct(), checkpoint resume."""
    from unsupervised_anomaly_detection_brain_mri_amd.models import anovaegan
    from unsupervised_anomaly_detection_brain_mri_amd.trainers import AnoVAEGAN
    d = AnoVAEGAN.Config()
    assert (d.modelname, d.scale, d.kappa, d.kl_weight) == ('AnoVAEGAN', 10.0, 1.0, 1.0)
    cfg, opt, ds = _config(AnoVAEGAN, tmp_path, h=64, bs=4, epochs=1)
    model = AnoVAEGAN(None, cfg, network=anovaegan)
    assert model.model_dir == 'AnoVAEGAN_dSyntheticDataset_s64x64_anovaegan_b4_z64_'
    model.train(ds)
    nb = ds.num_batches(4, set='TRAIN')
    assert [model.engine.step_count(g) for g in ('Encoder', 'Generator', 'Discriminator')] == [nb, nb, 5 * nb]
    assert {'TRAIN/gen_loss', 'TRAIN/disc_loss', 'TRAIN/reconstructionLoss', 'TRAIN/kl', 'VAL/reconstructionLoss'} <= set(model.curves)
    run = model.step(ds.next_batch(4, set='VAL')[0], Phase.VAL)
    assert set(run) == {'reconstructionLoss', 'kl', 'enc_loss', 'loss', 'reconstruction', 'L1'}
    assert run['enc_loss'] == pytest.approx(run['reconstructionLoss'] + run['kl'], rel=1e-5)
    r = model.reconstruct(ds.next_batch(1, set='VAL')[0][0], eps=0.0)
    assert r['reconstruction'].shape == (1, 64, 64, 1) and np.isfinite(r['l1err'])
    w = model.engine.get_buffer_host(_lib.BUF_PARAMS)
    model.engine.close()
    cfg2, _, _ = _config(AnoVAEGAN, tmp_path, h=64, bs=4, epochs=1)
    m2 = AnoVAEGAN(None, cfg2, network=anovaegan, seed=9)
    assert m2.load_checkpoint() == 1 and np.array_equal(m2.engine.get_buffer_host(_lib.BUF_PARAMS), w)
    assert [m2.engine.step_count(g) for g in ('Encoder', 'Generator', 'Discriminator')] == [nb, nb, 5 * nb]
    m2.engine.close()


@pytest.mark.parametrize('tname,mname', [('ConstrainedAE', 'constrained_autoencoder'), ('AAE', 'adversarial_autoencoder'),
                                         ('ConstrainedAAE', 'constrained_adversarial_autoencoder')])
def test_latent_ae_trainers(tmp_path, tname, mname):
    """trainers/ConstrainedAE.py, AAE.py, ConstrainedAAE.py: Config defaults, one epoch of the reference loop (shortened critic
    iterations), fetch keys, reconstruct(), resume with the three Adam step counters."""
    import importlib
    T = getattr(importlib.import_module('unsupervised_anomaly_detection_brain_mri_amd.trainers'), tname)
    net = getattr(importlib.import_module('unsupervised_anomaly_detection_brain_mri_amd.models'), mname)
    cfg, opt, ds = _config(T, tmp_path, h=64, bs=4, epochs=1)
    assert cfg.modelname == tname and (tname == 'AAE' or cfg.rho == 1) and (tname == 'ConstrainedAE' or cfg.scale == 10.0)
    model = T(None, cfg, network=net)
    model.D_ITERS = 2
    assert model.model_dir == f'{tname}_dSyntheticDataset_s64x64_{mname}_b4_z64_'
    model.train(ds)
    nb = ds.num_batches(4, set='TRAIN')
    steps = [model.engine.step_count(g) for g in ('Encoder', 'AE', 'Discriminator')]
    assert steps == ([0, nb, 0] if tname == 'ConstrainedAE' else [nb, 2 * nb, 2 * nb])
    run = model.step(ds.next_batch(4, set='VAL')[0], Phase.VAL)
    want = {'loss', 'L2', 'reconstructionLoss', 'reconstruction', 'L1'} | (set() if tname == 'AAE' else {'Rec_z'})
    assert set(run) == want
    if tname == 'AAE':
        assert run['loss'] == pytest.approx(run['L2'], rel=1e-6)
    else:
        assert run['loss'] == pytest.approx(run['L2'] + run['Rec_z'], rel=1e-5)
    r = model.reconstruct(ds.next_batch(1, set='VAL')[0][0])
    assert r['reconstruction'].shape == (1, 64, 64, 1) and np.isfinite(r['l1err'])
    w = model.engine.get_buffer_host(_lib.BUF_PARAMS)
    model.engine.close()
    cfg2, _, _ = _config(T, tmp_path, h=64, bs=4, epochs=1)
    m2 = T(None, cfg2, network=net, seed=3)
    assert m2.load_checkpoint() == 1 and np.array_equal(m2.engine.get_buffer_host(_lib.BUF_PARAMS), w)
    assert [m2.engine.step_count(g) for g in ('Encoder', 'AE', 'Discriminator')] == steps
    m2.engine.close()


def test_tf_checkpoint_export_and_resume(tmp_path):
    """save_tf() writes a TF-V2 tensor bundle under the reference's variable names; a fresh trainer's load() reads it back (weights,
    Adam slots and the step count recovered from beta1_power), and the next train step is bit-identical to continuing in place."""
    from unsupervised_anomaly_detection_brain_mri_amd.utils import tf_checkpoint
    cfg, opt, ds = _config(VAE, tmp_path, epochs=1)
    model = VAE(None, cfg, network=variational_autoencoder)
    model.train(ds)
    prefix = model.save_tf(model.checkpointDir, 1)
    names = set(tf_checkpoint.read_index(prefix + '.index')[1])
    assert {'Encoder/enc_conv2D_0/kernel', 'Encoder/enc_conv2D_0/kernel/Adam', 'Encoder/enc_conv2D_0/kernel/Adam_1', 'beta1_power'} <= names
    w, m, v = (model.engine.get_buffer_host(b) for b in (_lib.BUF_PARAMS, _lib.BUF_ADAM_M, _lib.BUF_ADAM_V))
    t = model.engine.step_count
    ck = os.path.join(model.checkpointDir, model.model_dir)
    os.remove(os.path.join(ck, 'VAE.model-1.npz'))                # leave only the TF bundle
    cfg2, _, _ = _config(VAE, tmp_path, epochs=1)
    m2 = VAE(None, cfg2, network=variational_autoencoder, seed=5)
    ok, counter = m2.load(m2.checkpointDir)
    assert ok and counter == 1 and m2.engine.step_count == t
    for b, ref in ((_lib.BUF_PARAMS, w), (_lib.BUF_ADAM_M, m), (_lib.BUF_ADAM_V, v)):
        assert np.array_equal(m2.engine.get_buffer_host(b), ref)
    x = synthetic_slices(8, 64, 64, seed=3)
    eps = np.random.default_rng(0).standard_normal((8, 64)).astype(np.float32)
    for mod in (model, m2):
        mod.engine.train_step(x, eps=eps, masks=None, lr=1e-3)
    assert np.array_equal(model.engine.get_buffer_host(_lib.BUF_PARAMS), m2.engine.get_buffer_host(_lib.BUF_PARAMS))
    model.engine.close(); m2.engine.close()


def test_gmvae_dense_trainer_surface(tmp_path):
    """trainers/GMVAE.py on models/gaussian_mixture_variational_autoencoder.py: Config defaults, fetch keys, an oracle-checked TRAIN
    step with injected noise / dropout masks, training progress, restoration-mode reconstruct(), resume."""
    from oracle import gmvae_dense as ogd
    from unsupervised_anomaly_detection_brain_mri_amd.models import gaussian_mixture_variational_autoencoder as net
    from unsupervised_anomaly_detection_brain_mri_amd.trainers import GMVAE
    assert (GMVAE.Config().dim_c, GMVAE.Config().dim_z, GMVAE.Config().dim_w, GMVAE.Config().c_lambda,
            GMVAE.Config().restore_lr, GMVAE.Config().restore_steps, GMVAE.Config().tv_lambda) == (6, 1, 1, 1, 1e-3, 150, 1.8)
    h, bs = 64, 4
    opt = get_options(batchsize=bs, learningrate=1e-3, numEpochs=2, zDim=64, outputWidth=h, outputHeight=h,
                      config={'CHECKPOINTDIR': str(tmp_path / 'ck'), 'SAMPLEDIR': str(tmp_path / 'smp')})
    ds = SyntheticDataset(16, 8, h, h, seed=0)
    cfg = get_config(GMVAE, opt, 'ADAM', [8, 8], 0.2, ds)
    cfg.dim_c, cfg.dim_z, cfg.dim_w, cfg.restore_steps = 5, 2, 3, 4
    model = GMVAE(None, cfg, network=net)
    assert model.model_dir.startswith('GMVAE_dSyntheticDataset')
    m = ogd.GMVAEDense(h, 8, 5, 2, 3, 1.0)
    assert [n for n, _, _ in model.engine.spec] == [n for n, _, _ in m.spec]
    p = {k: v.astype(np.float64) for k, v in model.engine.get_params().items()}
    x = ds.next_batch(bs, set='TRAIN')[0]
    rng = np.random.default_rng(3)
    e_w, e_z = rng.standard_normal((bs, 3)).astype(np.float32), rng.standard_normal((bs, 2)).astype(np.float32)
    masks = model._masks(bs, True)
    assert set(masks) == {'w_mu', 'w_ls', 'z_mu', 'dec'} and masks['dec'].shape == (bs, model.engine.flat)
    run = model.step(x, Phase.TRAIN, eps=(e_w, e_z), masks=masks)
    assert set(run) == {'reconstruction', 'L1', 'L2', 'L1_sum', 'L2_sum', 'reconstructionLoss', 'mean_p_loss', 'conditional_prior_loss',
                        'w_prior_loss', 'c_prior_loss', 'loss'}
    o = m.new_opt(p)
    _, ls, _ = m.train_step(p, o, x.astype(np.float64), e_w.astype(np.float64), e_z.astype(np.float64), {k: v.astype(np.float64) for k, v in masks.items()},
                            lr=1e-3, beta1=cfg.beta1)
    for k in ('mean_p_loss', 'conditional_prior_loss', 'w_prior_loss', 'c_prior_loss', 'loss'):
        assert run[k] == pytest.approx(ls[k], rel=3e-4), k
    ref = np.concatenate([p[nm].reshape(-1) for nm, _, _ in m.spec])
    got = model.engine.get_buffer_host(_lib.BUF_PARAMS)
    # one Adam step moves every touched weight by ~lr * sign(g): a weight whose gradient is rounding noise may step the other way
    assert np.abs(got - ref).max() <= 2.5e-3 and np.mean(np.abs(got - ref)) <= 2e-5
    model.train(ds)
    assert len(model.curves['TRAIN/loss']) == 2 and model.curves['VAL/loss'][1] < model.curves['VAL/loss'][0]     # TRAIN is noisy under dropout
    # restoration-mode reconstruct: deterministic mode equals the oracle's loop
    xs = ds.next_batch(2, set='VAL')[0]
    r = model.reconstruct(xs, eps=0.0)
    p2 = {k: v.astype(np.float64) for k, v in model.engine.get_params().items()}
    zero = lambda s: (np.zeros((2, 3)), np.zeros((2, 2)))
    rref = m.reconstruct(p2, xs.astype(np.float64), zero, restore_steps=4, restore_lr=cfg.restore_lr, tv_lambda=cfg.tv_lambda)
    assert np.mean(np.abs(r['reconstruction'] - rref['reconstruction'])) <= 2e-5
    assert np.abs(r['reconstruction'] - rref['reconstruction']).max() <= 8 * cfg.restore_lr * cfg.tv_lambda + 1e-4
    assert r['l1err'] == pytest.approx(r['l2err'], rel=1e-6)
    g = model.restore_gradients(xs, eps=(np.zeros((2, 3), np.float32), np.zeros((2, 2), np.float32)))
    assert g.shape == xs.shape and np.isfinite(g).all()
    model.restore_steps = 0
    assert model.reconstruct(xs[0], dropout=True)['reconstruction'].shape == (1, h, h, 1)
    w = model.engine.get_buffer_host(_lib.BUF_PARAMS)
    t = model.engine.step_count('AE')
    model.engine.close()
    m2 = GMVAE(None, cfg, network=net, seed=9)
    assert m2.load_checkpoint() == 2 and m2.engine.step_count('AE') == t
    assert np.array_equal(m2.engine.get_buffer_host(_lib.BUF_PARAMS), w)
    m2.engine.close()


@pytest.mark.parametrize('mname', ['autoencoder', 'autoencoder_spatial'])
def test_context_encoder_trainer(tmp_path, mname):
    """trainers/CE.py: the network reads the masked batch, the L1 term compares with the clean one; TRAIN step vs the oracle (forward on
    x_ce, loss / backward against x), VAL = plain AE forward, reconstruct() feeds x_ce = x."""
    import importlib
    from unsupervised_anomaly_detection_brain_mri_amd.trainers import CE
    from unsupervised_anomaly_detection_brain_mri_amd.trainers.CE import retrieve_masked_batch
    net = getattr(importlib.import_module(f'unsupervised_anomaly_detection_brain_mri_amd.models.{mname}'), mname)
    import random
    cfg, opt, ds = _config(CE, tmp_path, h=64, bs=4, epochs=3)
    model = CE(None, cfg, network=net)
    model.mask_rng = random.Random(0)            # the reference draws the squares from the unseeded `random` module
    assert model.model_dir.startswith('CE_dSyntheticDataset')
    x, _, bm = ds.next_batch(4, set='TRAIN', return_brainmask=True)
    x_ce = retrieve_masked_batch(x, bm, rng=random.Random(1))          # seeded: every input of this test is reproducible
    assert x_ce.shape == x.shape and (x_ce != x).any()
    m = ovae.SpatialAE(64, 64, 1, 8) if mname == 'autoencoder_spatial' else ovae.Model('AE', 64, 64, 1, 8, 64)
    p32 = model.engine.get_params()
    p = {k: v.astype(np.float64) for k, v in p32.items()}
    masks = model._draw(4, True)[1]
    m64 = {k: v.astype(np.float64) for k, v in masks.items()}
    args = (p, x_ce.astype(np.float64), m64) if mname == 'autoencoder_spatial' else (p, x_ce.astype(np.float64), None, m64)
    out, cache = m.forward(*args)
    ls = m.losses(x.astype(np.float64), out)
    bn_names = {'enc': [f'Encoder/batch_normalization_{i}' for i in range(3)], 'dec_in': 'Decoder/batch_normalization',
                'dec': [f'Decoder/batch_normalization_{i + 1}' for i in range(3)]}
    for math in ('f32', 'bf16x3'):
        # round 1's red run of this test (enc0 filter gradient 1.9e-4 off) was activation-kink flips, not arithmetic: with freshly
        # initialised weights the decoder pre-activations are ~1e-2 and hundreds of them lie within 1e-6 of zero
        # (tools/debug/ce_spatial_rootcause.py), and the masked batch was drawn from the unseeded `random` module, so the flip set changed
        # from run to run.  The oracle is therefore differentiated with the device's activation pattern and EVERY tensor is held to 1e-4.
        model.engine.set_math(math)
        got = model.engine.forward(x, None, masks, want_backward=True, x_ce=x_ce)
        act, flips = device_activation_pattern(model.engine, p32, x, got['x_hat'], cache, 3, bn_names)
        model.engine.backward()
        assert float(got['scalars'][0]) == pytest.approx(ls['reconstructionLoss'], rel=1e-4)
        assert np.abs(got['x_hat'].cpu().numpy() - out['x_hat']).max() <= 1e-4 * np.abs(out['x_hat']).max()
        assert np.abs(got['L1'].cpu().numpy() - ls['L1']).max() <= 1e-4 * np.abs(ls['L1']).max()
        g = m.backward(p, x.astype(np.float64), out, cache, m64, act=act)
        assert_grads_close(model.engine.get_grads(), g, [nm for nm, _, _ in model.engine.spec], flips=flips)
        assert sum(flips.values()) <= 64, flips          # a handful of round-off flips, not a different function
    model.engine.set_math('bf16x3')
    run = model.step(x, Phase.TRAIN, x_ce=x_ce, dropout_masks=masks)
    assert set(run) == {'reconstruction', 'L1', 'reconstructionLoss', 'loss'} and run['loss'] == run['reconstructionLoss']
    assert run['loss'] == pytest.approx(ls['loss'], rel=2e-4)
    # VAL: x_ce = x
    v = model.step(x, Phase.VAL)
    ref = model.engine.forward(x, None, None, want_backward=False)
    assert v['loss'] == pytest.approx(float(ref['scalars'][0]), rel=1e-6)
    model.train(ds)
    assert len(model.curves['TRAIN/loss']) == 3 and np.isfinite(model.curves['VAL/loss']).all()
    model.config.learningrate = 2e-4
    fixed = [float(model.step(x, Phase.TRAIN, x_ce=x_ce, dropout_masks=masks, fetch_maps=False)['loss']) for _ in range(16)]      # same batch, holes and masks: the objective falls
    assert np.mean(fixed[-3:]) < fixed[0]
    r = model.reconstruct(x[0])
    assert r['reconstruction'].shape == (1, 64, 64, 1)
    with pytest.raises(ValueError):
        from unsupervised_anomaly_detection_brain_mri_amd.models import variational_autoencoder as vnet
        CE(None, cfg, network=vnet)                                           # the context encoder's loss has no KL term
    model.engine.close()
