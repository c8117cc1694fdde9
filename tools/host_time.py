import time, torch, numpy as np, sys
sys.path.insert(0, '.')
from unsupervised_anomaly_detection_brain_mri_amd.engine import Engine, rng_fill
from unsupervised_anomaly_detection_brain_mri_amd.parallel import DataParallelStep
from unsupervised_anomaly_detection_brain_mri_amd.utils.synthetic import synthetic_slices
B=64
eng = Engine('VAE', 128, 128, 1, 8, 128, max_batch=B, device='cuda:0', math='bf16x3')
x = torch.from_numpy(synthetic_slices(B, 128, 128, seed=1)).cuda()
jobs = [('eps', 128, 'normal', 0.0), ('mu', 128, 'keep', 0.2), ('sigma', 128, 'keep', 0.2), ('dec', 8*8*16, 'keep', 0.2)]
dp = DataParallelStep(eng, 1)
def step(i):
    got = rng_fill(jobs, B, 1, i, 0)
    eps = got.pop('eps')
    return dp.train_step(x, eps, got, lr=1e-4, beta1=0.5, want_l1=True, want_latents=False)
for i in range(10): step(i)
torch.cuda.synchronize()
# host enqueue time with an idle GPU queue each step (sync before), i.e. pure host cost
ts=[]
for i in range(30):
    torch.cuda.synchronize()
    t0=time.perf_counter(); step(i); ts.append(time.perf_counter()-t0)
print('host enqueue per step (ms): median %.3f min %.3f' % (np.median(ts)*1e3, min(ts)*1e3))
t0=time.perf_counter()
for i in range(100): step(i)
t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
print('100 steps: enqueue done after %.1f ms, gpu done after %.1f ms' % ((t1-t0)*1e3, (t2-t0)*1e3))
