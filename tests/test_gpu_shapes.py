"""GPU: planner coverage.  The host-side planner picks different kernels / split factors as the batch and the image size change
(spatial vs generic implicit GEMM, channel-chunk splits, fused final epilogue, paired filter-gradient blocks).  For a sweep of
batch sizes and resolutions the split-bf16 mode must agree with the exact-fp32 MFMA mode -- two independent kernel families.

Reconstruction and loss are held to the 1e-4 bar.  Gradients only get a gross-error bound (relative L2 3e-2, max-norm 1e-1):
at these batch sizes (10^5-10^6 activations behind the decoder's input ReLU) a few pre-activations lie within the forward
tolerance of zero, their derivative (0 vs 1) flips between ANY two fp32 implementations, and one flipped element moves
single entries of the small dense-layer gradients by 1e-3 of the tensor's max (measured against the fp64 oracle at n = 16:
exact-fp32 mode 3.5e-4, split-bf16 1.8e-3 on dense_dec/kernel, everything else <= 4e-4; flipping, inside the fp64 oracle
itself, every pre-activation with |bn| < 1e-6 moves that tensor by 5e-3 in L2, |bn| < 1e-5 by 1.7e-2).  The strict 1e-4
max-norm parity lives (a) against the oracle in test_gpu_model.py / test_gpu_cevae.py / test_gpu_gmvae.py at batch sizes where
no pre-activation sits on a kink and (b) per contraction at the bench shapes in test_gpu_ops_large.py."""
import numpy as np
import pytest
import torch

from oracle import vae as ovae

pytestmark = pytest.mark.gpu

try:
    from unsupervised_anomaly_detection_brain_mri_amd.engine import Engine
    from tests.gpu_util import assert_close
except Exception:
    Engine = None


@pytest.mark.parametrize('arch,h,n', [('VAE', 128, 1), ('VAE', 128, 7), ('VAE', 128, 33), ('AE', 128, 64), ('VAE', 64, 19),
                                      ('VAE', 256, 3), ('ceVAE', 128, 9), ('AE', 32, 40)])
def test_math_modes_agree_across_planner_paths(arch, h, n):
    oarch = 'VAE' if arch == 'ceVAE' else arch
    m = ovae.Model(oarch, h, h, 1, 8, 128) if arch != 'ceVAE' else ovae.CeVAE(h, h, 1, 8, 128)
    p32 = ovae.init_params(m.spec, seed=11, dtype=np.float32, perturb=True)
    x = ovae.synthetic_slices(n, h, h, seed=n, dtype=np.float32)
    rng = np.random.default_rng(n)
    eps = rng.standard_normal((n, 128)).astype(np.float32)
    res = {}
    for math in ('f32', 'bf16x3'):
        eng = Engine(arch, h, h, 1, 8, 128, max_batch=n, math=math)
        eng.set_params(p32)
        kw = {'x_ce': x * (rng.random(x.shape) > 0.05).astype(np.float32)} if arch == 'ceVAE' and math == 'f32' else {}
        if arch == 'ceVAE':
            kw = {'x_ce': res.get('x_ce', kw.get('x_ce'))}
            res['x_ce'] = kw['x_ce']
        out = eng.forward(x, eps if arch != 'AE' else None, None, want_backward=True, **kw)
        eng.backward()
        torch.cuda.synchronize()
        res[math] = (out['x_hat'].cpu().numpy(), out['scalars'].cpu().numpy(), eng.get_grads(),
                     out['anomaly'].cpu().numpy() if arch == 'ceVAE' else None)
        eng.close()
    xa, sa, ga, aa = res['f32']
    xb, sb, gb, ab = res['bf16x3']
    assert_close(xb, xa, name='x_hat')
    assert abs(sb[2] - sa[2]) <= 1e-4 * abs(sa[2])
    for name, _, _ in m.spec:
        a64, b64 = ga[name].astype(np.float64), gb[name].astype(np.float64)
        l2 = np.linalg.norm(b64 - a64) / max(np.linalg.norm(a64), 1e-30)
        assert l2 <= 3e-2, f'{name}: relative L2 {l2:.2e}'
        assert_close(gb[name], ga[name], tol=1e-1, name=name)
    if aa is not None:
        l2 = np.linalg.norm(ab.astype(np.float64) - aa) / np.linalg.norm(aa.astype(np.float64))
        assert l2 <= 3e-2, f'anomaly: relative L2 {l2:.2e}'
