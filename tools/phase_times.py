"""Times the phases of the materialised-graph handles (AAE family, AnoVAE-GAN, f-AnoGAN) at 128x128 (64x64 for the ResNet graph),
batch 64 (32): ms per phase incl. its Adam step, HIP-event timed over 20 repetitions."""
import json, sys
import numpy as np, torch
sys.path.insert(0, '.')
from unsupervised_anomaly_detection_brain_mri_amd.gan_engine import GanEngine
from unsupervised_anomaly_detection_brain_mri_amd.utils.synthetic import synthetic_slices


def init(eng):
    rng = np.random.default_rng(3)
    flat = np.zeros(eng.nparams, np.float32)
    for name, shape, off in eng.spec:
        cnt = int(np.prod(shape))
        if name.endswith('kernel'):
            rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
            lim = np.sqrt(6.0 / (shape[-2] * rf + shape[-1] * rf)); flat[off:off + cnt] = rng.uniform(-lim, lim, cnt)
        elif name.endswith('gamma'):
            flat[off:off + cnt] = 1.0
    eng.set_params(flat)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(b) / reps, 3)


res = {}
g = torch.Generator(device='cuda').manual_seed(1)
for kind in ('constrained_ae', 'aae', 'constrained_aae'):
    h, bs, zd = 128, 64, 128
    eng = GanEngine(h, h, 1, 8, zd, max_batch=bs, variant='aae', aae_kind=kind)
    init(eng)
    x = torch.from_numpy(synthetic_slices(bs, h, h, seed=1)).cuda()
    z = torch.randn(bs, zd, device='cuda', generator=g); e = torch.rand(bs, device='cuda', generator=g)
    r = {'AE': timed(lambda: (eng.aae_phase('AE', x, want_images=False), eng.adam('AE', 1e-4)))}
    if kind != 'constrained_ae':
        r['Discriminator'] = timed(lambda: (eng.aae_phase('Discriminator', x, z=z, eps=e), eng.adam('Discriminator', 1e-4)))
        r['Encoder(gen)'] = timed(lambda: (eng.aae_phase('Encoder', x), eng.adam('Encoder', 1e-4)))
    res[kind + '_128_b64_ms'] = r
    eng.close()
eng = GanEngine(128, 128, 1, 8, 128, max_batch=64, variant='anovaegan')
init(eng)
x = torch.from_numpy(synthetic_slices(64, 128, 128, seed=1)).cuda()
z = torch.randn(64, 128, device='cuda', generator=g); e = torch.rand(64, device='cuda', generator=g)
res['anovaegan_128_b64_ms'] = {'VAE': timed(lambda: (eng.phase('Encoder', x=x, eps=z, want_images=False), eng.adam('Encoder', 1e-4))),
                               'Generator': timed(lambda: (eng.phase('Generator', x=x, eps=z, want_images=False), eng.adam('Generator', 1e-4))),
                               'Discriminator': timed(lambda: (eng.phase('Discriminator', x=x, eps=z, alpha=e, want_images=False), eng.adam('Discriminator', 1e-4)))}
eng.close()
for (dc, dz, dw) in ((6, 1, 1), (9, 128, 64)):
    eng = GanEngine(128, 128, 1, 8, zdim=dz, max_batch=64, variant='aae', aae_kind='gmvae', dim=dc, dim_w=dw)
    init(eng)
    ew = torch.randn(64, dw, device='cuda', generator=g); ez = torch.randn(64, dz, device='cuda', generator=g)
    xr = x.clone()
    res[f'gmvae_dense_c{dc}_z{dz}_w{dw}_128_b64_ms'] = {
        'train': timed(lambda: (eng.gm_phase(x, ew, ez, want_l1=False), eng.adam('AE', 1e-4, 0.5, 0.999))),
        'forward': timed(lambda: eng.gm_phase(x, ew, ez, want_backward=False, want_l1=False)),
        'restore_step': timed(lambda: eng.gm_restore_step(xr, ew, ez))}
    eng.close()
for math in ('f32', 'bf16x3_all'):
    eng = GanEngine(128, 128, 1, 8, 128, max_batch=64, variant='aae', aae_kind='vae_zimmerer', math=math)
    init(eng)
    ez = torch.randn(64, 128, device='cuda', generator=g)
    res[f'vae_zimmerer_128_b64_{math}_ms'] = {'train': timed(lambda: (eng.zim_phase(x, ez, want_l1=False), eng.adam('AE', 1e-4, 0.5, 0.999)), reps=10),
                                             'forward': timed(lambda: eng.zim_phase(x, ez, want_backward=False, want_l1=False), reps=10)}
    eng.close()
for math in ('f32', 'bf16x3_all'):
    eng = GanEngine(128, 128, 1, 32, zdim=1, max_batch=64, variant='aae', aae_kind='gmvae_you', dim=9, dim_w=1, math=math)
    init(eng)
    ew = torch.randn(64, 32, 32, 1, device='cuda', generator=g); ez = torch.randn(64, 32, 32, 1, device='cuda', generator=g)
    xr = x.clone()
    res[f'gmvae_you_128_b64_{math}_ms'] = {'train': timed(lambda: (eng.gm_phase(x, ew, ez, want_l1=False), eng.adam('AE', 5e-5, 0.5, 0.999)), reps=10),
                                          'forward': timed(lambda: eng.gm_phase(x, ew, ez, want_backward=False, want_l1=False), reps=10),
                                          'restore_step': timed(lambda: eng.gm_restore_step(xr, ew, ez), reps=10)}
    eng.close()
eng = GanEngine(64, 64, 1, 8, 128, max_batch=32, variant='aae', aae_kind='caae_chen', dim=64, math='f32')
init(eng)
x64 = torch.from_numpy(synthetic_slices(32, 64, 64, seed=1)).cuda()
z32 = torch.randn(32, 128, device='cuda', generator=g); e32 = torch.full((32,), 0.3, device='cuda')
res['caae_chen_64_b32_f32_ms'] = {'AE': timed(lambda: (eng.aae_phase('AE', x64, want_images=False), eng.adam('AE', 1e-4)), reps=5),
                                  'Discriminator': timed(lambda: (eng.aae_phase('Discriminator', x64, z=z32, eps=e32), eng.adam('Discriminator', 1e-4)), reps=5),
                                  'Encoder(gen)': timed(lambda: (eng.aae_phase('Encoder', x64), eng.adam('Encoder', 1e-4)), reps=5)}
eng.close()
# (the hipGraph-replay comparison that lived here went with uad_gan_set_graph_mode in round 4: profiles/r01_m_graph_replay.json holds its numbers)
print(json.dumps(res))
