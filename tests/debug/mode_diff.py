import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import vae as ovae
from unsupervised_anomaly_detection_brain_mri_amd.engine import Engine
h = 128
m = ovae.Model('VAE', h, h, 1, 8, 128)
p32 = ovae.init_params(m.spec, seed=11, dtype=np.float32, perturb=True)
for n in (4, 16, 33, 64):
    x = ovae.synthetic_slices(n, h, h, seed=n, dtype=np.float32)
    eps = np.random.default_rng(n).standard_normal((n, 128)).astype(np.float32)
    res = {}
    for math in ('f32', 'bf16x3'):
        eng = Engine('VAE', h, h, 1, 8, 128, max_batch=n, math=math); eng.set_params(p32)
        out = eng.forward(x, eps, None, want_backward=True); eng.backward(); torch.cuda.synchronize()
        res[math] = eng.get_grads(); eng.close()
    worst = sorted(((np.abs(res['bf16x3'][k] - res['f32'][k]).max() / np.abs(res['f32'][k]).max(), k) for k, _, _ in m.spec), reverse=True)[:4]
    print(n, [(f'{e:.1e}', k.split('/')[1] + '/' + k.split('/')[2]) for e, k in worst])
