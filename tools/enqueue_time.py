import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from unsupervised_anomaly_detection_brain_mri_amd.engine import Engine
from unsupervised_anomaly_detection_brain_mri_amd.utils.synthetic import synthetic_slices
eng = Engine('VAE', 128, 128, 1, 8, 128, max_batch=64)
rng = np.random.default_rng(3)
flat = np.zeros(eng.nparams, np.float32)
for name, shape, off in eng.spec:
    cnt = int(np.prod(shape))
    if name.endswith('kernel'):
        rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
        lim = np.sqrt(6.0 / (shape[-2] * rf + shape[-1] * rf)); flat[off:off + cnt] = rng.uniform(-lim, lim, cnt)
    elif name.endswith('gamma'): flat[off:off + cnt] = 1.0
eng.set_params(flat)
x = torch.from_numpy(synthetic_slices(64, 128, 128, seed=1)).cuda()
eps = torch.randn(64, 128, device='cuda')
for _ in range(10): eng.train_step(x, eps, None, want_latents=False)
torch.cuda.synchronize()
K = 50
t0 = time.perf_counter()
for _ in range(K): eng.train_step(x, eps, None, want_latents=False)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'enqueue {1e3*(t1-t0)/K:.3f} ms/step, total {1e3*(t2-t0)/K:.3f} ms/step')
