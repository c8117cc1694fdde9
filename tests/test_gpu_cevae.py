"""GPU parity of the ceVAE step (models/context_encoder_variational_autoencoder.py, trainers/ceVAE.py) through the
C-ABI vs the fp64 oracle: both reconstructions, every scalar loss, the parameter gradients (sum over the two branches
through the shared variables), the input-gradient anomaly map, Adam, and the data-only backward used by
validation / reconstruct.  Tolerance 1e-4 max-norm relative (north_star)."""
import numpy as np
import pytest
import torch

from oracle import nn as onn
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


def _setup(h, inter, zdim, n, seed=0, dropout=True, perturb=True):
    m = ovae.CeVAE(h, h, 1, inter, zdim)
    p32 = ovae.init_params(m.spec, seed=5 + seed, dtype=np.float32, perturb=perturb)
    x = ovae.synthetic_slices(n, h, h, seed=seed, dtype=np.float32)
    x_ce = x.copy()
    q = h // 4
    x_ce[:, q:q + h // 6, q + 2:q + 2 + h // 6] = 0          # a context hole, same for every sample (CE.py defect A3)
    rng = np.random.default_rng(200 + seed)
    eps = rng.standard_normal((n, zdim)).astype(np.float32)
    flat = inter * inter * p32['Bottleneck/conv2d/kernel'].shape[-1]
    masks = {}
    if dropout:
        for k, w in (('mu', zdim), ('sigma', zdim), ('dec', flat), ('mu_ce', zdim), ('dec_ce', flat)):
            masks[k] = onn.make_dropout_mask(rng, (n, w), 0.2)
    return m, p32, x, x_ce, eps, masks


def test_cevae_param_table():
    eng = Engine('ceVAE', 128, 128, 1, 8, 128, max_batch=2)
    m = ovae.CeVAE(128, 128, 1, 8, 128)
    assert eng.nparams == 1758449
    assert [(n, tuple(s)) for n, s, _ in eng.spec] == [(n, tuple(s)) for n, s, _ in m.spec]
    eng.close()


@pytest.mark.parametrize('math', ['f32', 'bf16x3'])
@pytest.mark.parametrize('h,inter,zdim,n,dropout', [(32, 8, 16, 2, True), (64, 8, 64, 3, True), (128, 8, 128, 2, True),
                                                    (128, 8, 128, 3, False)])
def test_cevae_forward_backward_parity(h, inter, zdim, n, dropout, math):
    m, p32, x, x_ce, eps, masks = _setup(h, inter, zdim, n, dropout=dropout)
    p64, m64 = _f64(p32), _f64(masks)
    x64, xc64, e64 = x.astype(np.float64), x_ce.astype(np.float64), eps.astype(np.float64)
    out, caches = m.ce_forward(p64, x64, xc64, e64, m64)
    ls = m.ce_losses(x64, xc64, out)
    g = m.ce_backward(p64, x64, xc64, out, caches, m64)

    eng = Engine('ceVAE', h, h, 1, inter, zdim, max_batch=n, math=math)
    eng.set_params(p32)
    got = eng.forward(x, eps, masks, want_backward=True, x_ce=x_ce)
    eng.backward()
    torch.cuda.synchronize()
    assert_close(got['x_hat'].cpu().numpy(), out['x_hat'], name='x_hat')
    assert_close(got['x_hat_ce'].cpu().numpy(), out['x_hat_ce'], name='x_hat_ce')
    assert_close(got['L1_vae'].cpu().numpy(), ls['L1_vae'], tol=2e-4, name='L1_vae')
    assert_close(got['L1_ce'].cpu().numpy(), ls['L1_ce'], tol=2e-4, name='L1_ce')
    sc = got['scalars'].cpu().numpy()
    for idx, key in ((0, 'reconstructionLoss'), (1, 'kl'), (2, 'loss'), (4, 'Rec_vae'), (5, 'Rec_ce'), (6, 'loss_vae')):
        assert abs(sc[idx] - ls[key]) <= 1e-4 * abs(ls[key]), (key, sc[idx], ls[key])
    rps = got['rec_per_sample'].cpu().numpy()
    np.testing.assert_allclose(rps[:n], ls['L1_vae'].reshape(n, -1).sum(1), rtol=1e-4)
    np.testing.assert_allclose(rps[n:], ls['L1_ce'].reshape(n, -1).sum(1), rtol=1e-4)
    for k in ('z_mu', 'z_log_sigma', 'z_sigma'):
        assert_close(got[k].cpu().numpy(), out[k], name=k)
    grads = eng.get_grads()
    for name, _, _ in m.spec:
        tol = 1e-4 if name.endswith('kernel') else 5e-4
        assert_close(grads[name], g[name], tol=tol, name=name)
    # anomaly = L1_vae * |d loss_vae / d x|: products of two 1e-4-accurate maps
    assert_close(got['anomaly'].cpu().numpy(), g['anomaly'], tol=3e-4, name='anomaly')
    eng.close()


def test_cevae_data_only_backward_leaves_grads_and_matches():
    """Validation / reconstruct path (trainers/ceVAE.py:105,119-144): x_ce = x, no dropout, anomaly map wanted but no
    parameter gradients -- the data-only backward must give the same map and must not touch UAD_BUF_GRADS."""
    h, inter, zdim, n = 64, 8, 32, 3
    m, p32, x, _, eps, _ = _setup(h, inter, zdim, n, seed=3, dropout=False)
    eng = Engine('ceVAE', h, h, 1, inter, zdim, max_batch=4)
    eng.set_params(p32)
    sentinel = np.full(eng.nparams, 7.0, np.float32)
    eng.set_buffer_host(_lib.BUF_GRADS, sentinel)
    got = eng.forward(x, eps, None, want_backward='data')          # x_ce defaults to x
    eng.backward()
    torch.cuda.synchronize()
    assert np.array_equal(eng.get_buffer_host(_lib.BUF_GRADS), sentinel)
    ref = m.ce_reconstruct(_f64(p32), x.astype(np.float64), eps.astype(np.float64))
    assert_close(got['anomaly'].cpu().numpy(), ref['anomaly'], tol=3e-4, name='anomaly')
    assert_close((torch.as_tensor(x).cuda() - got['anomaly']).cpu().numpy(), ref['reconstruction'], name='restoration')
    # x_ce = x: both branches differ only by the sampling noise of the VAE branch
    full = eng.forward(x, eps, None, want_backward=True)
    eng.backward()
    torch.cuda.synchronize()
    assert_close(full['anomaly'].cpu().numpy(), got['anomaly'].cpu().numpy(), tol=1e-6, name='anomaly(full vs data-only)')
    eng.close()


def test_cevae_train_trajectory_matches_oracle():
    h, inter, zdim, n = 32, 8, 32, 4
    m, p32, x, x_ce, eps, masks = _setup(h, inter, zdim, n, seed=2, perturb=False)
    p64 = _f64(p32)
    opt = m.new_opt(p64)
    eng = Engine('ceVAE', h, h, 1, inter, zdim, max_batch=n)
    eng.set_params(p32)
    ref_losses, got_losses = [], []
    for _ in range(10):
        _, ls, _ = m.ce_train_step(p64, opt, x.astype(np.float64), x_ce.astype(np.float64), eps.astype(np.float64),
                                   _f64(masks), lr=1e-4, beta1=0.5)
        ref_losses.append(float(ls['loss']))
        out = eng.train_step(x, eps, masks, lr=1e-4, beta1=0.5, x_ce=x_ce)
        got_losses.append(float(out['scalars'][2].item()))
    np.testing.assert_allclose(got_losses, ref_losses, rtol=3e-4)
    assert got_losses[-1] < got_losses[0]
    flat = eng.get_buffer_host(_lib.BUF_PARAMS)
    ref = ovae.flatten_params(m.spec, p64)
    assert np.abs(flat - ref).max() <= 2e-3 * np.abs(ref).max()
    eng.close()


def test_cevae_error_paths():
    eng = Engine('ceVAE', 32, 32, 1, 8, 16, max_batch=2)
    x = np.zeros((2, 32, 32, 1), np.float32)
    with pytest.raises(ValueError):
        eng.forward(x, None, {'mu': np.ones((2, 16), np.float32)})    # mask_mu without mask_mu_ce
    eng.close()
    eng = Engine('VAE', 32, 32, 1, 8, 16, max_batch=2)
    with pytest.raises(ValueError):
        eng.forward(x, None, None, x_ce=x)                            # x_ce is ceVAE-only
    eng.close()
