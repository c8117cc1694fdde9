"""Compare the intermediates of the ResNet f-AnoGAN phases (debug buffers) with the numpy oracle."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from oracle import fanogan_schlegl as ofs, vae as ovae
from unsupervised_anomaly_detection_brain_mri_amd.gan_engine import GanEngine

h, zdim, dim, n = (int(a) for a in (sys.argv[1:5] if len(sys.argv) > 4 else (32, 16, 32, 2)))
math = sys.argv[5] if len(sys.argv) > 5 else 'f32'
m = ofs.FAnoGANSchlegl(h, h // 8, zdim, dim)
p = ovae.init_params(m.spec, seed=31, dtype=np.float64, perturb=True)
rng = np.random.default_rng(190)
x = ovae.synthetic_slices(n, h, h, seed=0, dtype=np.float64)
z = rng.standard_normal((n, zdim)); alpha = rng.uniform(0, 1, (n, 1))
eng = GanEngine(h, h, 1, h // 8, zdim, max_batch=n, math=math, variant='resnet', dim=dim)
eng.set_params(p)


def cmp(name, dev, ref):
    ref = np.asarray(ref, np.float64)
    d = dev.detach().cpu().numpy().reshape(-1)[:ref.size].reshape(ref.shape).astype(np.float64)
    print(f'{name:28s} rel {np.abs(d - ref).max() / max(np.abs(ref).max(), 1e-30):.3e}  (max {np.abs(ref).max():.3e})')


def buf(name, lo=0):
    return eng.debug_buffer(name)[lo:]


def grads(group, g):
    gd = eng.get_grads()
    for k, s, _ in m.spec:
        if k.startswith(group):
            ref = np.asarray(g.get(k, np.zeros(s))).reshape(s)
            print(f'grad {k:50s} err {np.abs(gd[k] - ref).max():.3e} max {np.abs(ref).max():.3e}')


print('== generator phase')
out = eng.phase('Generator', z=z)
c = {}
ls, g = m.gen_phase(p, z, c)
cmp('xg', buf('xg'), ls['generated'])
for k in range(4):
    cmp(f'sg_h1_{k}', buf(f'sg_h1_{k}'), c['gen']['blocks'][k]['h1']); cmp(f'sg_h2_{k}', buf(f'sg_h2_{k}'), c['gen']['blocks'][k]['h2'])
print('gen_loss', out['gen_loss'].item(), ls['gen_loss'])
grads('Generator', g)
print('== critic phase')
out = eng.phase('Discriminator', x=x, z=z, alpha=alpha)
c = {}
ls, g = m.disc_phase(p, x, z, alpha, c)
for k in range(4):
    cmp(f'sd_h1_{k}', buf(f'sd_h1_{k}'), np.concatenate([cc['blocks'][k]['h1'] for cc in c['disc']]))
    cmp(f'sd_h2_{k}', buf(f'sd_h2_{k}'), np.concatenate([cc['blocks'][k]['h2'] for cc in c['disc']]))
cmp('Gx(ddx)', buf('Gx'), ls['ddx'])
for k in ('disc_fake', 'disc_real', 'penalty', 'disc_loss'):
    print(k, out[k].item(), ls[k])
grads('Discriminator', g)
print('== encoder phase')
out = eng.phase('Encoder', x=x)
ls, g = m.enc_phase(p, x)
for k in ('loss_img', 'loss_fts', 'enc_loss', 'reconstructionLoss'):
    print(k, out[k].item(), ls[k])
cmp('z_enc', out['z_enc'], ls['z_enc']); cmp('x_enc', out['reconstruction'], ls['reconstruction'])
grads('Encoder', g)
