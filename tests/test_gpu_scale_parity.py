"""GPU: whole-model gradient parity AT THE BASELINE.json BATCH SIZES, flip-aware.

C1 AE 128^2 N=8 | C2 VAE 128^2 N=64 | C3 ceVAE 128^2 N=16 and N=64 per GPU | C5 spatial GMVAE 256^2 N=16 -- each in both math modes
('f32' exact-fp32 MFMA, 'bf16x3' split-bf16) against the fp64 oracle (oracle/vae.py, oracle/gmvae.py): reconstruction, loss scalars
and EVERY parameter-gradient tensor at 1e-4 max-norm relative (north_star's tolerance).

Why flip-aware: at these sizes 10^5-10^7 pre-activations per layer are evaluated and a few lie within fp32 round-off of the
(Leaky)ReLU kink; which side they land on depends on summation order, so their derivative (alpha or 1) differs between any two fp32
implementations of the same graph, and ONE such element behind the decoder's input ReLU moves the small dense gradients by 1e-3 of
their max ("Parity at scale").  The test reads the activation pattern the device actually used
(tests/gpu_util.py: device_activation_pattern), counts the disagreements with the oracle per layer, and differentiates the oracle
with the device's pattern -- so the 1e-4 bar is enforced on every tensor whether or not flips occurred, and the flip counts are
printed (run with -s) and bounded.  Restated reference: models/variational_autoencoder.py:9-47, trainers/VAE.py:36-46,
trainers/ceVAE.py:38-51, trainers/GMVAE_spatial.py:61-97."""
import numpy as np
import pytest
import torch

from oracle import gmvae as og
from oracle import nn as onn
from oracle import vae as ovae
from tests.gpu_util import assert_close, assert_grads_close, device_activation_pattern

pytestmark = pytest.mark.gpu

try:
    from unsupervised_anomaly_detection_brain_mri_amd.engine import Engine
except Exception:
    Engine = None

MODES = ('f32', 'bf16x3')
VAE_BN = {'enc': [f'Encoder/batch_normalization_{i}' for i in range(4)], 'dec_in': 'Decoder/batch_normalization',
          'dec': [f'Decoder/batch_normalization_{i + 1}' for i in range(4)]}


def _f64(d):
    return {k: np.asarray(v, np.float64) for k, v in d.items()}


def _masks(rng, n, zdim, flat, keys):
    shapes = {'z': (n, zdim), 'mu': (n, zdim), 'sigma': (n, zdim), 'dec': (n, flat), 'mu_ce': (n, zdim), 'dec_ce': (n, flat)}
    return {k: onn.make_dropout_mask(rng, shapes[k], 0.2) for k in keys}


def _report(tag, math, flips, worst):
    tot = sum(flips.values())
    w = max(worst, key=worst.get)
    mags = getattr(flips, 'mag', None) or {}
    print(f'\n[{tag} {math}] activation flips vs fp64 oracle: {tot} {dict((k, v) for k, v in flips.items() if v)}; largest oracle |pre-activation| among '
          f'them (of the tensor max): {dict((k, float(f"{v:.1e}")) for k, v in mags.items())}; L1-sign disagreements: {getattr(flips, "l1_sign", 0)}; '
          f'worst gradient tensor {w}: {worst[w]:.2e}')
    return tot


@pytest.mark.parametrize('arch,n', [('AE', 8), ('VAE', 64)])
def test_ae_vae_gradients_at_baseline_batch(arch, n):
    """C1 (trainers/AE.py, batch 8) and C2 (trainers/VAE.py, batch 64): the bench's exact step."""
    h, zdim, flat = 128, 128, 8 * 8 * 16
    m = ovae.Model(arch, h, h, 1, 8, zdim)
    p32 = ovae.init_params(m.spec, seed=3, perturb=True)
    x = ovae.synthetic_slices(n, h, h, seed=0)
    rng = np.random.default_rng(1)
    eps = rng.standard_normal((n, zdim)).astype(np.float32) if arch == 'VAE' else None
    masks = _masks(rng, n, zdim, flat, ('mu', 'sigma', 'dec') if arch == 'VAE' else ('z',))
    p64, x64, m64 = _f64(p32), x.astype(np.float64), _f64(masks)
    out, cache = m.forward(p64, x64, None if eps is None else eps.astype(np.float64), m64)
    ls = m.losses(x64, out)
    eng = Engine(arch, h, h, 1, 8, zdim, max_batch=n)
    eng.set_params(p32)
    names = [nm for nm, _, _ in eng.spec]
    for math in MODES:
        eng.set_math(math)
        got = eng.forward(x, eps, masks, want_backward=True)
        act, flips = device_activation_pattern(eng, p32, x, got['x_hat'], cache, 4, VAE_BN, math=math, xhat_oracle=out['x_hat'])
        eng.backward()
        torch.cuda.synchronize()
        assert_close(got['x_hat'].cpu().numpy(), out['x_hat'], name='x_hat')
        sc = got['scalars'].cpu().numpy()
        assert abs(sc[0] - ls['reconstructionLoss']) <= 1e-4 * ls['reconstructionLoss']
        assert abs(sc[2] - ls['loss']) <= 1e-4 * abs(ls['loss'])
        g = m.backward(p64, x64, out, cache, m64, act=act)
        worst = assert_grads_close(eng.get_grads(), g, names, flips=flips)
        tot = _report(f'{arch} N={n}', math, flips, worst)
        assert tot <= 1e-5 * sum(v.size for k, v in cache.items() if 'bn' in k) + 8, flips
    eng.close()


@pytest.mark.parametrize('arch,n', [('AE', 8), ('VAE', 64)])
def test_ae_vae_bf16x6_holds_1e5_at_baseline_batch(arch, n):
    """Round 6: the VAE family's THIRD math mode -- bf16x6 (three bf16 planes per operand, six products per multiply on the bf16 matrix cores, fp32
    accumulate) -- at the bench's exact step, held to 1e-5 against the fp64 oracle where the other two modes are held to 1e-4: reconstruction, loss
    scalars and every gradient tensor; activation flips must be rounding ties of fp32 size (64 ulp of the site's max), as in the exact-fp32 mode.
    The exact-fp32 mode is run beside it on the same inputs, to the same 1e-5."""
    h, zdim, flat = 128, 128, 8 * 8 * 16
    m = ovae.Model(arch, h, h, 1, 8, zdim)
    p32 = ovae.init_params(m.spec, seed=3, perturb=True)
    x = ovae.synthetic_slices(n, h, h, seed=0)
    rng = np.random.default_rng(1)
    eps = rng.standard_normal((n, zdim)).astype(np.float32) if arch == 'VAE' else None
    masks = _masks(rng, n, zdim, flat, ('mu', 'sigma', 'dec') if arch == 'VAE' else ('z',))
    p64, x64, m64 = _f64(p32), x.astype(np.float64), _f64(masks)
    out, cache = m.forward(p64, x64, None if eps is None else eps.astype(np.float64), m64)
    ls = m.losses(x64, out)
    eng = Engine(arch, h, h, 1, 8, zdim, max_batch=n)
    eng.set_params(p32)
    names = [nm for nm, _, _ in eng.spec]
    TOL = 1e-5
    worst_of = {}
    for math in ('f32', 'bf16x6'):
        eng.set_math(math)
        got = eng.forward(x, eps, masks, want_backward=True)
        act, flips = device_activation_pattern(eng, p32, x, got['x_hat'], cache, 4, VAE_BN, math=math, xhat_oracle=out['x_hat'])
        eng.backward()
        torch.cuda.synchronize()
        e_x = assert_close(got['x_hat'].cpu().numpy(), out['x_hat'], tol=TOL, name=f'x_hat ({math})')
        sc = got['scalars'].cpu().numpy()
        assert abs(sc[0] - ls['reconstructionLoss']) <= TOL * ls['reconstructionLoss'], (math, sc[0], ls['reconstructionLoss'])
        assert abs(sc[2] - ls['loss']) <= TOL * abs(ls['loss']), (math, sc[2], ls['loss'])
        g = m.backward(p64, x64, out, cache, m64, act=act)
        worst = assert_grads_close(eng.get_grads(), g, names, tol=TOL, flips=flips)
        worst['x_hat'] = e_x
        worst_of[math] = worst
        _report(f'{arch} N={n}', math, flips, worst)
    # (both are fp32-grade: on the first GPU run the largest bf16x6 error was 2.3e-6 -- the first layer's bias gradient, a sum over 1 M positions --
    # against 5.4e-7 in exact fp32; the bar of this test is the 1e-5 above, the comparison is printed for the record)
    k = max(worst_of['bf16x6'], key=worst_of['bf16x6'].get)
    print(f'[{arch} N={n}] worst tensor in bf16x6: {k} {worst_of["bf16x6"][k]:.2e} (exact fp32: {worst_of["f32"][k]:.2e})')
    eng.close()


def test_vae_small_width_ragged_batch():
    """32 x 32, zDim 64, 80 slices: the narrow graph (two blocks per side; its bottleneck does not split over four workgroups per sample) at a
    batch that is one full 64-sample chunk + a ragged one in the fused bottleneck gradient kernel.  Without the activation pattern the split-bf16
    mode is 1.3e-3 off on Bottleneck/dense_dec/kernel (a handful of the 80 x 64 x 64 ReLU inputs of the decoder sit inside its round-off of the
    kink, tools/debug/n80_bottleneck_grad.py); with the device's pattern injected every tensor has to meet the usual bar."""
    h, zdim, n = 32, 64, 80
    m = ovae.Model('VAE', h, h, 1, 8, zdim)
    p32 = ovae.init_params(m.spec, seed=3, perturb=True)
    x = ovae.synthetic_slices(n, h, h, seed=0)
    rng = np.random.default_rng(1)
    eps = rng.standard_normal((n, zdim)).astype(np.float32)
    flat = 8 * 8 * p32['Bottleneck/conv2d/kernel'].shape[-1]
    masks = _masks(rng, n, zdim, flat, ('mu', 'sigma', 'dec'))
    p64, x64, m64 = _f64(p32), x.astype(np.float64), _f64(masks)
    out, cache = m.forward(p64, x64, eps.astype(np.float64), m64)
    eng = Engine('VAE', h, h, 1, 8, zdim, max_batch=n)
    eng.set_params(p32)
    names = [nm for nm, _, _ in eng.spec]
    bn = {'enc': [f'Encoder/batch_normalization_{i}' for i in range(2)], 'dec_in': 'Decoder/batch_normalization',
          'dec': [f'Decoder/batch_normalization_{i + 1}' for i in range(2)]}
    for math in MODES:
        eng.set_math(math)
        got = eng.forward(x, eps, masks, want_backward=True)
        act, flips = device_activation_pattern(eng, p32, x, got['x_hat'], cache, 2, bn, math=math, xhat_oracle=out['x_hat'])
        eng.backward()
        torch.cuda.synchronize()
        g = m.backward(p64, x64, out, cache, m64, act=act)
        worst = assert_grads_close(eng.get_grads(), g, names, flips=flips)
        _report(f'VAE 32x32 N={n}', math, flips, worst)
    eng.close()


@pytest.mark.parametrize('n', [16, 64])
def test_cevae_gradients_at_baseline_batch(n):
    """C3 (trainers/ceVAE.py:38-51): 16 slices per GPU (batch 128 over 8 GPUs) and 64 per GPU; both branches' patterns are read from the
    handle's [x ; x_ce] rows; the anomaly map L1_vae * |d loss_vae / d x| is held to the same bar."""
    h, zdim, flat = 128, 128, 8 * 8 * 16
    m = ovae.CeVAE(h, h, 1, 8, zdim)
    p32 = ovae.init_params(m.spec, seed=4, perturb=True)
    x = ovae.synthetic_slices(n, h, h, seed=2)
    x_ce = x.copy()
    r = np.random.default_rng(9)
    for i in range(n):                      # 1-3 zeroed 20x20 squares per slice (the per-sample form of CE.py:123-139; the broadcast defect is pinned elsewhere)
        for _ in range(int(r.integers(1, 4))):
            a, b = int(r.integers(30, 78)), int(r.integers(30, 78))
            x_ce[i, a:a + 20, b:b + 20] = 0
    rng = np.random.default_rng(5)
    eps = rng.standard_normal((n, zdim)).astype(np.float32)
    masks = _masks(rng, n, zdim, flat, ('mu', 'sigma', 'dec', 'mu_ce', 'dec_ce'))
    p64, x64, xc64, m64 = _f64(p32), x.astype(np.float64), x_ce.astype(np.float64), _f64(masks)
    out, caches = m.ce_forward(p64, x64, xc64, eps.astype(np.float64), m64)
    ls = m.ce_losses(x64, xc64, out)
    eng = Engine('ceVAE', h, h, 1, 8, zdim, max_batch=n)
    eng.set_params(p32)
    names = [nm for nm, _, _ in eng.spec]
    for math in MODES:
        eng.set_math(math)
        got = eng.forward(x, eps, masks, want_backward=True, x_ce=x_ce)
        act_v, fl_v = device_activation_pattern(eng, p32, x, got['x_hat'], caches[1], 4, VAE_BN, rows=slice(0, n), math=math, xhat_oracle=out['x_hat'])
        act_c, fl_c = device_activation_pattern(eng, p32, x_ce, got['x_hat_ce'], caches[3], 4, VAE_BN, rows=slice(n, 2 * n), math=math, xhat_oracle=out['x_hat_ce'])
        eng.backward()
        torch.cuda.synchronize()
        assert_close(got['x_hat'].cpu().numpy(), out['x_hat'], name='x_hat')
        assert_close(got['x_hat_ce'].cpu().numpy(), out['x_hat_ce'], name='x_hat_ce')
        sc = got['scalars'].cpu().numpy()
        for idx, key in ((0, 'reconstructionLoss'), (1, 'kl'), (2, 'loss'), (4, 'Rec_vae'), (5, 'Rec_ce'), (6, 'loss_vae')):
            assert abs(sc[idx] - ls[key]) <= 1e-4 * abs(ls[key]), (key, sc[idx], ls[key])
        g = m.ce_backward(p64, x64, xc64, out, caches, m64, act_v=act_v, act_c=act_c)
        flips = type(fl_v)({f'vae/{k}': v for k, v in fl_v.items()})
        flips.update({f'ce/{k}': v for k, v in fl_c.items()})
        flips.l1_sign = getattr(fl_v, 'l1_sign', 0) + getattr(fl_c, 'l1_sign', 0)      # counted and bounded (rounding ties only) per branch
        flips.mag = {**{f'vae/{k}': v for k, v in fl_v.mag.items()}, **{f'ce/{k}': v for k, v in fl_c.mag.items()}}
        worst = assert_grads_close(eng.get_grads(), g, names, flips=flips)
        assert_close(got['anomaly'].cpu().numpy(), g['anomaly'], tol=2e-4, name='anomaly')
        tot = _report(f'ceVAE N={n}', math, flips, worst)
        assert tot <= 2e-5 * sum(v.size for k, v in caches[1].items() if 'bn' in k) + 8, flips
    eng.close()


def test_gmvae_spatial_gradients_at_baseline_batch():
    """C5 (models/gaussian_mixture_variational_autoencoder_spatial.py, 256^2, dim_c 9, dim_z = dim_w = 1) at 16 slices: the trunk's
    activation pattern is read from the device; the latent heads' own kinks (the p(z|w,c) ReLU, tf.maximum in the c-prior) have O(1)
    arguments and are left to the oracle."""
    h, n = 256, 16
    m = og.GMVAE(h, h, 1, 8, 9, 1, 1, 1.0)
    p32 = og.init_params(m.spec, seed=7, dtype=np.float32, perturb=True)
    x = ovae.synthetic_slices(n, h, h, seed=0)
    rng = np.random.default_rng(50)
    e_w = rng.standard_normal((n, 8, 8, 1)).astype(np.float32)
    e_z = rng.standard_normal((n, 8, 8, 1)).astype(np.float32)
    p64, x64 = _f64(p32), x.astype(np.float64)
    out, cache = m.forward(p64, x64, e_w.astype(np.float64), e_z.astype(np.float64))
    ls = m.losses(x64, out)
    bn = {'enc': m.bn[:5], 'dec_in': m.bn[5], 'dec': m.bn[6:]}
    eng = Engine('GMVAE_spatial', h, h, 1, 8, max_batch=n, dim_c=9, dim_z=1, dim_w=1, c_lambda=1.0)
    eng.set_params(p32)
    names = [nm for nm, _, _ in eng.spec]
    for math in MODES:
        eng.set_math(math)
        got = eng.gm_forward(x, e_w, e_z, want_backward=True)
        act, flips = device_activation_pattern(eng, p32, x, got['x_hat'], cache, 5, bn, final_kernel='dec_Conv2D_final/kernel', math=math, xhat_oracle=out['xz_mu'])
        eng.backward()
        torch.cuda.synchronize()
        assert_close(got['x_hat'].cpu().numpy(), out['xz_mu'], name='xz_mu')
        sc = got['scalars'].cpu().numpy()
        for idx, key in ((0, 'mean_p_loss'), (1, 'conditional_prior_loss'), (2, 'loss'), (3, 'w_prior_loss'), (4, 'c_prior_loss')):
            assert abs(sc[idx] - ls[key]) <= 1e-4 * max(abs(ls[key]), 1e-3), (key, sc[idx], ls[key])
        g = m.backward(p64, x64, out, cache, act=act)
        worst = assert_grads_close(eng.get_grads(), g, names, flips=flips)
        tot = _report(f'GMVAE_spatial N={n}', math, flips, worst)
        assert tot <= 1e-5 * sum(v.size for k, v in cache.items() if 'bn' in k) + 8, flips
    eng.close()
