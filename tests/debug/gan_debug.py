"""Compare every intermediate of the f-AnoGAN phases (debug buffers of the handle) with the numpy oracle."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from oracle import fanogan as ofa, vae as ovae, nn
from unsupervised_anomaly_detection_brain_mri_amd.gan_engine import GanEngine

h, inter, zdim, n = (int(a) for a in (sys.argv[1:5] if len(sys.argv) > 4 else (32, 8, 16, 2)))
math = sys.argv[5] if len(sys.argv) > 5 else 'f32'
m = ofa.FAnoGAN(h, inter, zdim)
p = ovae.init_params(m.spec, seed=21, dtype=np.float64, perturb=True)
rng = np.random.default_rng(90)
x = ovae.synthetic_slices(n, h, h, seed=0, dtype=np.float64)
z = rng.standard_normal((n, zdim)); alpha = rng.uniform(0, 1, (n, 1))
eng = GanEngine(h, h, 1, inter, zdim, max_batch=n, math=math)
eng.set_params(p)
L = m.npool


def cmp(name, dev, ref):
    ref = np.asarray(ref, np.float64)
    d = dev.detach().cpu().numpy().reshape(-1)[:ref.size].reshape(ref.shape).astype(np.float64)
    print(f'{name:28s} rel {np.abs(d - ref).max() / max(np.abs(ref).max(), 1e-30):.3e}  (max {np.abs(ref).max():.3e})')


def buf(name, lo=0, cnt=None):
    b = eng.debug_buffer(name)
    return b[lo:] if cnt is None else b[lo:lo + cnt]


print('== critic phase')
out = eng.phase('Discriminator', x=x, z=z, alpha=alpha)
torch.cuda.synchronize()
xg, gcache = m.gen_forward(p, z)
cmp('xg', buf('xg'), xg)
for i in range(L + 1):
    cmp(f'gc{i}', buf(f'gc{i}'), gcache['c'][i]); cmp(f'ga{i}', buf(f'ga{i}'), gcache['a'][i])
x_hat = x + alpha.reshape(-1, 1, 1, 1) * (xg - x)
allx = np.concatenate([xg, x, x_hat])
_, d_all, c_all = m.disc_forward(p, allx)
cmp('din', buf('din'), allx)
for i in range(L):
    cmp(f'Da{i+1}', buf(f'Da{i+1}'), c_all['a'][i + 1])
cmp('Dd', buf('Dd'), d_all)
_, _, c_hat = m.disc_forward(p, x_hat)
ddx, tape = m.disc_input_grad(p, c_hat)
cmp('Gx(ddx)', buf('Gx'), ddx)
for i in range(L):
    per = tape[i][1][0].size
    cmp(f'V{i}', buf(f'V{i}'), tape[i][1])
    cmp(f'Dg{i}.tail(dc1)', buf(f'Dg{i}', 3 * n * per), tape[i][2])
pen, gbar = m.gradient_penalty(ddx)
print('penalty', out['penalty'].item(), pen)
cmp('gbar', buf('din', 3 * n * h * h), gbar)
g2, inject = m.disc_penalty_grads(p, c_hat, tape, gbar)
for i in range(L):
    cmp(f'inj{i}', buf(f'inj{i}'), inject[i])
ls, g = m.disc_phase(p, x, z, alpha)
gd = eng.get_grads()
for k, s, _ in m.spec:
    if k.startswith('Discriminator'):
        ref = np.asarray(g.get(k, np.zeros(s))).reshape(s)
        print(f'grad {k:50s} err {np.abs(gd[k] - ref).max():.3e} max {np.abs(ref).max():.3e}')
for k in ('disc_fake', 'disc_real', 'penalty', 'disc_loss'):
    print(k, out[k].item(), ls[k])

print('== generator phase')
out = eng.phase('Generator', z=z)
ls, g = m.gen_phase(p, z)
gd = eng.get_grads()
print('gen_loss', out['gen_loss'].item(), ls['gen_loss'])
for k, s, _ in m.spec:
    if k.startswith('Generator'):
        ref = np.asarray(g.get(k, np.zeros(s))).reshape(s)
        print(f'grad {k:50s} err {np.abs(gd[k] - ref).max():.3e} max {np.abs(ref).max():.3e}')

print('== encoder phase')
out = eng.phase('Encoder', x=x)
ls, g = m.enc_phase(p, x)
gd = eng.get_grads()
for k in ('loss_img', 'loss_fts', 'enc_loss', 'reconstructionLoss'):
    print(k, out[k].item(), ls[k])
cmp('z_enc', out['z_enc'], ls['z_enc']); cmp('x_enc', out['reconstruction'], ls['reconstruction'])
for k, s, _ in m.spec:
    if k.startswith('Encoder'):
        ref = np.asarray(g.get(k, np.zeros(s))).reshape(s)
        print(f'grad {k:50s} err {np.abs(gd[k] - ref).max():.3e} max {np.abs(ref).max():.3e}')
