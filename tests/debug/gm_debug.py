import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import gmvae as og, vae as ovae
from unsupervised_anomaly_detection_brain_mri_amd.engine import Engine
h = int(sys.argv[1]) if len(sys.argv) > 1 else 256
math = sys.argv[2] if len(sys.argv) > 2 else 'f32'
n = 1
m = og.GMVAE(h, h, 1, 8, 9, 1, 1, 1.0)
p32 = og.init_params(m.spec, seed=7, dtype=np.float32, perturb=True)
x = ovae.synthetic_slices(n, h, h, seed=0, dtype=np.float32)
rng = np.random.default_rng(50)
e_w = rng.standard_normal((n, 8, 8, 1)).astype(np.float32); e_z = rng.standard_normal((n, 8, 8, 1)).astype(np.float32)
p64 = {k: v.astype(np.float64) for k, v in p32.items()}
out, cache = m.forward(p64, x.astype(np.float64), e_w.astype(np.float64), e_z.astype(np.float64))
g = m.backward(p64, x.astype(np.float64), out, cache)
eng = Engine('GMVAE_spatial', h, h, 1, 8, max_batch=n, math=math, dim_c=9, dim_z=1, dim_w=1, c_lambda=1.0)
eng.set_params(p32)
got = eng.gm_forward(x, e_w, e_z, want_backward=True); eng.backward(); torch.cuda.synchronize()
grads = eng.get_grads()
for name, _, _ in m.spec:
    e = np.abs(grads[name] - g[name]).max() / max(np.abs(g[name]).max(), 1e-30)
    print(f'{name:40s} {e:.2e}', '  <<<' if e > 5e-4 else '')
