import numpy as np, torch, sys, os
sys.path.insert(0,'.')
from oracle import vae as ovae, nn as onn
from unsupervised_anomaly_detection_brain_mri_amd.engine import Engine
n=int(os.environ.get('N','80')); seed=int(os.environ.get('SEED','0'))
m = ovae.Model('VAE', 32, 32, 1, 8, 64)
p32 = ovae.init_params(m.spec, seed=3+seed, dtype=np.float32, perturb=True)
x = ovae.synthetic_slices(n, 32, 32, seed=seed, dtype=np.float32)
rng = np.random.default_rng(100+seed)
eps = rng.standard_normal((n, 64)).astype(np.float32)
flat = 8*8*p32['Bottleneck/conv2d/kernel'].shape[-1]
masks = {'mu': onn.make_dropout_mask(rng,(n,64),0.2), 'sigma': onn.make_dropout_mask(rng,(n,64),0.2), 'dec': onn.make_dropout_mask(rng,(n,flat),0.2)}
f64=lambda d:{k:np.asarray(v,np.float64) for k,v in d.items()}
p64=f64(p32)
out,cache=m.forward(p64,x.astype(np.float64),eps.astype(np.float64),f64(masks))
g=m.backward(p64,x.astype(np.float64),out,cache,f64(masks))
for math in ('f32','bf16x3'):
    eng=Engine('VAE',32,32,1,8,64,max_batch=n,math=math); eng.set_params(p32)
    eng.forward(x,eps,masks,want_backward=True); eng.backward(); torch.cuda.synchronize()
    gr=eng.get_grads()
    errs={k:float(np.abs(gr[k]-g[k]).max()/max(np.abs(g[k]).max(),1e-30)) for k,_,_ in m.spec}
    bad={k:f'{v:.1e}' for k,v in errs.items() if v>5e-5}
    print(math, 'worst', max(errs.values()), bad)
    eng.close()
