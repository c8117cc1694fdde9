"""GMVAE 256^2 N=16 bf16x3: which activation pattern did the device's backward use at enc_bn2?  (debug of tests/test_gpu_scale_parity.py)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import gmvae as og, vae as ovae
from tests.gpu_util import device_activation_pattern, BN_MULT
from unsupervised_anomaly_detection_brain_mri_amd.engine import Engine

h, n = 256, 16
m = og.GMVAE(h, h, 1, 8, 9, 1, 1, 1.0)
p32 = og.init_params(m.spec, seed=7, dtype=np.float32, perturb=True)
x = ovae.synthetic_slices(n, h, h, seed=0)
rng = np.random.default_rng(50)
e_w = rng.standard_normal((n, 8, 8, 1)).astype(np.float32); e_z = rng.standard_normal((n, 8, 8, 1)).astype(np.float32)
p64 = {k: v.astype(np.float64) for k, v in p32.items()}; x64 = x.astype(np.float64)
out, cache = m.forward(p64, x64, e_w.astype(np.float64), e_z.astype(np.float64))
bn = {'enc': m.bn[:5], 'dec_in': m.bn[5], 'dec': m.bn[6:]}
eng = Engine('GMVAE_spatial', h, h, 1, 8, max_batch=n, dim_c=9, dim_z=1, dim_w=1, c_lambda=1.0)
eng.set_params(p32)
eng.set_math('bf16x3')
got = eng.gm_forward(x, e_w, e_z, want_backward=True)
act, flips = device_activation_pattern(eng, p32, x, got['x_hat'], cache, 5, bn, final_kernel='dec_Conv2D_final/kernel')
c2 = eng.debug_buffer('enc_c2').cpu().numpy().reshape(cache['enc_bn2'].shape)
eng.backward(); torch.cuda.synchronize()
grads = eng.get_grads()
print('flips', flips)
idx = np.argwhere(act['enc_bn2'] != (cache['enc_bn2'] > 0))
a = (p32[m.bn[2] + '/gamma'] * BN_MULT); b = p32[m.bn[2] + '/beta']
for i in idx:
    i = tuple(i)
    cc = c2[i]; ch = i[3]
    print('flip at', i, 'oracle bn', cache['enc_bn2'][i], 'oracle c', cache['enc_c2'][i], 'dev c', cc, 'dev bn fma64', float(cc) * float(a[ch]) + float(b[ch]),
          'dev bn f32 unfused', np.float32(np.float32(cc) * a[ch]) + b[ch])
names = ['enc_conv2D_2/kernel', 'enc_conv2D_2/bias', m.bn[2] + '/beta', m.bn[2] + '/gamma', 'enc_conv2D_1/kernel', 'enc_conv2D_3/kernel']
def show(tag, g):
    print(tag, {k: f"{np.abs(grads[k] - g[k]).max() / np.abs(g[k]).max():.2e}" for k in names})
    return g
gA = show('A all injected     ', m.backward(p64, x64, out, cache, act=act))
actB = dict(act); del actB['enc_bn2']
gB = show('B enc_bn2 = oracle ', m.backward(p64, x64, out, cache, act=actB))
show('C nothing injected ', m.backward(p64, x64, out, cache))
for k in idx[:, 3]:
    nm = m.bn[2] + '/beta'
    print('channel', k, 'beta dev', grads[nm][k], 'A', gA[nm][k], 'B', gB[nm][k])
# one at a time
for j in range(len(idx)):
    pat = cache['enc_bn2'] > 0
    pat = pat.copy(); pat[tuple(idx[j])] = ~pat[tuple(idx[j])]
    actD = dict(act); actD['enc_bn2'] = pat
    show(f'D only flip {j}      ', m.backward(p64, x64, out, cache, act=actD))
eng.close()
