"""GPU parity: the whole AE / VAE step through the C-ABI (uad_forward / uad_backward / uad_adam_step) vs the fp64
numpy oracle on identical weights, inputs, eps and dropout masks.  Tolerance 1e-4 max-norm relative (north_star)."""
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


def _setup(arch, h, inter, zdim, n, seed=0, perturb=True):
    m = ovae.Model(arch, h, h, 1, inter, zdim)
    p32 = ovae.init_params(m.spec, seed=3 + seed, dtype=np.float32, perturb=perturb)
    x = ovae.synthetic_slices(n, h, h, seed=seed, dtype=np.float32)
    rng = np.random.default_rng(100 + seed)
    eps = rng.standard_normal((n, zdim)).astype(np.float32)
    flat = inter * inter * p32['Bottleneck/conv2d/kernel'].shape[-1]
    if arch == 'VAE':
        masks = {'mu': onn.make_dropout_mask(rng, (n, zdim), 0.2), 'sigma': onn.make_dropout_mask(rng, (n, zdim), 0.2),
                 'dec': onn.make_dropout_mask(rng, (n, flat), 0.2)}
    else:
        masks = {'z': onn.make_dropout_mask(rng, (n, zdim), 0.2)}
    return m, p32, x, eps, masks


def _f64(d):
    return {k: np.asarray(v, np.float64) for k, v in d.items()}


def test_param_table_matches_oracle_spec():
    eng = Engine('VAE', 128, 128, 1, 8, 128, max_batch=2)
    m = ovae.Model('VAE', 128, 128, 1, 8, 128)
    assert eng.nparams == 1758449
    assert [(n, tuple(s)) for n, s, _ in eng.spec] == [(n, tuple(s)) for n, s, _ in m.spec]
    eng.close()


@pytest.mark.parametrize('math', ['f32', 'bf16x3'])
@pytest.mark.parametrize('arch,h,inter,zdim,n', [('VAE', 32, 8, 16, 2), ('AE', 32, 8, 32, 3), ('VAE', 64, 8, 64, 5),
                                                  ('VAE', 128, 8, 128, 2), ('AE', 128, 8, 128, 1), ('VAE', 64, 16, 128, 2),
                                                  # 80 = one full 64-sample chunk + a ragged one in the fused bottleneck gradient kernel; this width splits
                                                  # no further than one workgroup per sample
                                                  ('VAE', 32, 8, 64, 80)])
def test_forward_backward_parity(arch, h, inter, zdim, n, math):
    """Both math modes must meet the same 1e-4 bar: 'f32' = exact fp32 MFMA, 'bf16x3' = split-bf16 products on the
    bf16 matrix cores with fp32 accumulation (forward and data-gradient k5 s2 contractions)."""
    if n > 16 and math == 'bf16x3':
        # 80 x 64 x 64 ReLU inputs: a few sit within the split-bf16 round-off of the kink and take the other derivative (measured: dense_dec/kernel
        # 1.3e-3 off at n = 64 and 80 alike, seed-independent, with and without the fused gradient kernel, 6e-7 in f32 mode:
        # tools/debug/n80_bottleneck_grad.py); the flip-aware comparison at these sample counts is tests/test_gpu_scale_parity.py
        pytest.skip('needs the flip-aware comparison (tests/test_gpu_scale_parity.py); the ragged-chunk logic under test is math-mode independent')
    m, p32, x, eps, masks = _setup(arch, h, inter, zdim, n)
    p64 = _f64(p32)
    out, cache = m.forward(p64, x.astype(np.float64), eps.astype(np.float64) if arch == 'VAE' else None, _f64(masks))
    ls = m.losses(x.astype(np.float64), out)
    g = m.backward(p64, x.astype(np.float64), out, cache, _f64(masks))

    eng = Engine(arch, h, h, 1, inter, zdim, max_batch=n, math=math)
    eng.set_params(p32)
    got = eng.forward(x, eps if arch == 'VAE' else None, masks, want_backward=True)
    eng.backward()
    torch.cuda.synchronize()
    assert_close(got['x_hat'].cpu().numpy(), out['x_hat'], name='x_hat')
    assert_close(got['L1'].cpu().numpy(), ls['L1'], tol=2e-4, name='L1')
    sc = got['scalars'].cpu().numpy()
    assert abs(sc[0] - ls['reconstructionLoss']) <= 1e-4 * abs(ls['reconstructionLoss'])
    assert abs(sc[2] - ls['loss']) <= 1e-4 * abs(ls['loss'])
    if arch == 'VAE':
        assert abs(sc[1] - ls['kl']) <= 1e-4 * abs(ls['kl'])
        for k in ('z_mu', 'z_log_sigma', 'z_sigma'):
            assert_close(got[k].cpu().numpy(), out[k], name=k)
    else:
        assert_close(got['z'].cpu().numpy(), out['z'], name='z')
    grads = eng.get_grads()
    worst = 0.0
    for name, _, _ in m.spec:
        # gradient parity: 1e-4 of the tensor's max-norm, loosened to 5e-4 for the long-reduction bias/BN sums
        tol = 1e-4 if name.endswith('kernel') else 5e-4
        worst = max(worst, assert_close(grads[name], g[name], tol=tol, name=name))
    eng.close()


@pytest.mark.parametrize('arch,h,zdim,n', [('VAE', 64, 64, 5), ('AE', 32, 32, 3)])
def test_segmented_backward_equals_the_whole_one(arch, h, zdim, n):
    """uad_backward by segments (what parallel.DataParallelStep issues: DECODER, BOTTLENECK, ENCODER_HI, ENCODER_LO) leaves the same bits as
    UAD_SEG_ALL, every segment's slice is final when its call returns (the all-reduce reads it then), and ENCODER_HI | ENCODER_LO tile ENCODER."""
    from unsupervised_anomaly_detection_brain_mri_amd.parallel import SEGMENT_ORDER
    m, p32, x, eps, masks = _setup(arch, h, 8, zdim, n)
    eng = Engine(arch, h, h, 1, 8, zdim, max_batch=n)
    eng.set_params(p32)
    e = eps if arch == 'VAE' else None
    eng.forward(x, e, masks, want_backward=True)
    eng.backward()
    whole = eng.buffer(_lib.BUF_GRADS).clone()
    eng.buffer(_lib.BUF_GRADS).zero_()
    segs = {s: eng.grad_segment(s) for s in SEGMENT_ORDER}
    (o_enc, c_enc), (o_hi, c_hi), (o_lo, c_lo) = eng.grad_segment(_lib.SEG_ENCODER), segs[_lib.SEG_ENCODER_HI], segs[_lib.SEG_ENCODER_LO]
    assert (o_lo, o_lo + c_lo, o_hi + c_hi) == (o_enc, o_hi, o_enc + c_enc) and c_lo > 0
    assert (c_hi > c_lo) if h >= 64 else (c_hi == 0)        # a two-block encoder (32 x 32) has no deep part: ENCODER_HI is empty there
    assert sum(c for _, c in segs.values()) == eng.nparams
    eng.forward(x, e, masks, want_backward=True)
    for s in SEGMENT_ORDER:
        eng.backward(s)
        off, cnt = segs[s]
        torch.cuda.current_stream().synchronize()          # only the caller's stream: the segment must have been joined into it
        assert torch.equal(eng.buffer(_lib.BUF_GRADS)[off:off + cnt], whole[off:off + cnt]), s
    assert torch.equal(eng.buffer(_lib.BUF_GRADS), whole)
    with pytest.raises(Exception):
        eng.backward(_lib.SEG_ENCODER_LO)                   # the forward state was consumed
    # deferred joins (uad_backward_deferred, what DataParallelStep issues since round 4): a segment either is complete in the caller's stream
    # (None) or names the stream it is complete in; the LAST segment always joins, after which the whole buffer is final in the caller's stream
    eng.buffer(_lib.BUF_GRADS).zero_()
    eng.forward(x, e, masks, want_backward=True)
    deferred = 0
    for s in SEGMENT_ORDER:
        ready = eng.backward_deferred(s)
        off, cnt = segs[s]
        if ready is None:
            torch.cuda.current_stream().synchronize()
        else:
            deferred += 1
            assert s != _lib.SEG_ENCODER_LO
            ready.synchronize()                             # ONLY the stream the call named
        assert torch.equal(eng.buffer(_lib.BUF_GRADS)[off:off + cnt], whole[off:off + cnt]), ('deferred', s)
    torch.cuda.current_stream().synchronize()
    assert torch.equal(eng.buffer(_lib.BUF_GRADS), whole)
    assert deferred >= 1                                    # (the decoder segment's reductions always sit on the side stream)
    eng.close()


def test_committed_model_fixture():
    """The handle against tests/golden/model_golden.npz (the oracle frozen at VAE 32 x 32, 2 slices; tests/golden/make_model_golden.py holds the
    inputs' recipe): x_hat, the scalars and every gradient tensor's sum, in both math modes."""
    import os
    from tests.golden import make_model_golden as mk
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'model_golden.npz'))
    m, p, x = mk.case()
    flat = mk.INTER * mk.INTER * p['Bottleneck/conv2d/kernel'].shape[-1]
    eps, masks = mk.noise(np.random.default_rng(100), flat)
    p32 = {k: v.astype(np.float32) for k, v in p.items()}
    for math in ('f32', 'bf16x3'):
        eng = Engine('VAE', mk.H, mk.H, 1, mk.INTER, mk.ZDIM, max_batch=mk.N, math=math)
        eng.set_params(p32)
        got = eng.forward(x.astype(np.float32), eps.astype(np.float32), {k: v.astype(np.float32) for k, v in masks.items()}, want_backward=True)
        eng.backward()
        torch.cuda.synchronize()
        assert_close(got['x_hat'].cpu().numpy(), G['x_hat'], name='x_hat')
        sc = got['scalars'].cpu().numpy()
        np.testing.assert_allclose(sc[:3], G['scalars'], rtol=1e-4)
        grads = eng.get_grads()
        for name, s_ref, a_ref in zip(G['grad_names'], G['grad_sum'], G['grad_abs_sum']):
            g = grads[str(name)].astype(np.float64)
            assert abs(g.sum() - s_ref) <= 5e-4 * a_ref + 1e-9, (math, name)
            assert abs(np.abs(g).sum() - a_ref) <= 5e-4 * a_ref + 1e-9, (math, name)
        eng.close()


def test_train_trajectory_vae_matches_oracle():
    """12 Adam steps from fixed init + fixed eps/masks (SURVEY.md §4 build-side plan): loss trajectory and final
    weights track the fp64 oracle.  lr is kept small enough for a monotone descent: with lr=1e-3 this problem overshoots
    (loss spikes at step 6) and the chaotic dynamics amplify fp32 summation-order noise to 5e-3 within 20 steps; even at
    lr=2e-4 the fp64 oracle itself turns non-monotone after step 16, so the window is 12 steps (agreement ~1e-6)."""
    arch, h, inter, zdim, n = 'VAE', 32, 8, 32, 4
    m, p32, x, eps, masks = _setup(arch, h, inter, zdim, n, seed=2, perturb=False)
    p64 = _f64(p32)
    opt = m.new_opt(p64)
    eng = Engine(arch, h, h, 1, inter, zdim, max_batch=n)
    eng.set_params(p32)
    ref_losses, got_losses = [], []
    for step in range(12):
        _, ls, _ = m.train_step(p64, opt, x.astype(np.float64), eps.astype(np.float64), _f64(masks), lr=2e-4, beta1=0.5)
        ref_losses.append(float(ls['loss']))
        out = eng.train_step(x, eps, masks, lr=2e-4, beta1=0.5)
        got_losses.append(float(out['scalars'][2].item()))
    np.testing.assert_allclose(got_losses, ref_losses, rtol=3e-4)
    assert got_losses[-1] < got_losses[0]
    assert eng.step_count == 12
    flat = eng.get_buffer_host(_lib.BUF_PARAMS)
    ref = ovae.flatten_params(m.spec, p64)
    # Adam's m/sqrt(v) amplifies tiny gradient differences on near-zero-gradient entries; compare in max-norm
    assert np.abs(flat - ref).max() <= 2e-3 * np.abs(ref).max()
    eng.close()


def test_reconstruct_forward_only_is_deterministic_and_matches():
    arch, h, inter, zdim, n = 'VAE', 128, 8, 128, 3
    m, p32, x, eps, _ = _setup(arch, h, inter, zdim, n, seed=4)
    eng = Engine(arch, h, h, 1, inter, zdim, max_batch=4)
    eng.set_params(p32)
    a = eng.forward(x, eps, None, want_backward=False)['x_hat'].cpu().numpy()
    b = eng.forward(x, eps, None, want_backward=False)['x_hat'].cpu().numpy()
    assert np.array_equal(a, b)
    ref = m.reconstruct(_f64(p32), x.astype(np.float64), eps.astype(np.float64))
    assert_close(a, ref['reconstruction'], name='reconstruction')
    res, l1 = eng.residual(x, a)
    assert_close(res.cpu().numpy(), np.maximum(x - a, 0), tol=1e-6)
    assert abs(l1.sum().item() - np.abs(x - a).sum()) <= 1e-4 * np.abs(x - a).sum()
    eng.close()


def test_error_paths():
    with pytest.raises(ValueError):
        Engine('VAE', 128, 128, 3, 8, 128, max_batch=2)       # numChannels != 1 unsupported
    with pytest.raises(ValueError):
        Engine('VAE', 100, 100, 1, 8, 128, max_batch=2)       # not a power of two
    eng = Engine('AE', 32, 32, 1, 8, 16, max_batch=2)
    with pytest.raises(ValueError):
        eng.forward(np.zeros((3, 32, 32, 1), np.float32))     # batch > max_batch
    with pytest.raises(ValueError):
        eng.backward()                                        # no forward(want_backward) before
    eng.close()
