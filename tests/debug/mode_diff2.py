import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from oracle import vae as ovae
from unsupervised_anomaly_detection_brain_mri_amd.engine import Engine
h = 128; n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
m = ovae.Model('VAE', h, h, 1, 8, 128)
p32 = ovae.init_params(m.spec, seed=11, dtype=np.float32, perturb=True)
x = ovae.synthetic_slices(n, h, h, seed=n, dtype=np.float32)
eps = np.random.default_rng(n).standard_normal((n, 128)).astype(np.float32)
t0 = time.time()
p64 = {k: v.astype(np.float64) for k, v in p32.items()}
out, cache = m.forward(p64, x.astype(np.float64), eps.astype(np.float64), None)
g = m.backward(p64, x.astype(np.float64), out, cache, None)
print('oracle', round(time.time() - t0, 1), 's')
for math in ('f32', 'bf16x3'):
    eng = Engine('VAE', h, h, 1, 8, 128, max_batch=n, math=math); eng.set_params(p32)
    o = eng.forward(x, eps, None, want_backward=True); eng.backward(); torch.cuda.synchronize()
    gg = eng.get_grads()
    xe = np.abs(o['x_hat'].cpu().numpy() - out['x_hat']).max() / np.abs(out['x_hat']).max()
    worst = sorted(((np.abs(gg[k] - g[k]).max() / np.abs(g[k]).max(), k) for k, _, _ in m.spec), reverse=True)[:5]
    print(math, 'x_hat', f'{xe:.1e}', [(f'{e:.1e}', k.split('/')[1] + '/' + k.split('/')[2]) for e, k in worst])
    eng.close()
