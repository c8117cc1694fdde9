"""Host enqueue time of one train step on the data-parallel path (RCCL on ONE rank, collectives forced) next to the plain step: is the segmented
backward + per-bucket all-reduce bound by the host or by the device?  python tools/host_time_dp.py  (one GPU)"""
import os, sys, time
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, '.')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29577')
from unsupervised_anomaly_detection_brain_mri_amd.engine import Engine, rng_fill
from unsupervised_anomaly_detection_brain_mri_amd.parallel import DataParallelStep
from unsupervised_anomaly_detection_brain_mri_amd.utils.synthetic import synthetic_slices
B = 64
torch.cuda.set_device(0)
BACKEND = os.environ.get('HT_BACKEND', 'nccl')          # nccl | gloo | none (control: no process group at all -> the plain step only)
eng = None
if os.environ.get('HT_ENGINE_FIRST'):                   # the handle (streams, events, every allocation) exists before the communicator does
    eng = Engine('VAE', 128, 128, 1, 8, 128, max_batch=B, device='cuda:0', math='bf16x3')
if BACKEND == 'nccl':
    if os.environ.get('HT_LAZY'):                       # no device_id: the communicator is only created by the first collective
        dist.init_process_group('nccl', rank=0, world_size=1)
    else:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    if os.environ.get('HT_TOUCH'):                      # create the communicator before anything is timed
        t = torch.ones(4, device='cuda'); dist.all_reduce(t); torch.cuda.synchronize()
elif BACKEND == 'gloo':
    dist.init_process_group('gloo', rank=0, world_size=1)
eng = eng or Engine('VAE', 128, 128, 1, 8, 128, max_batch=B, device='cuda:0', math='bf16x3')
x = torch.from_numpy(synthetic_slices(B, 128, 128, seed=1)).cuda()
jobs = [('eps', 128, 'normal', 0.0), ('mu', 128, 'keep', 0.2), ('sigma', 128, 'keep', 0.2), ('dec', 8 * 8 * 16, 'keep', 0.2)]


def run(tag, dp):
    def step(i):
        got = rng_fill(jobs, B, 1, i, 0)
        eps = got.pop('eps')
        return dp.train_step(x, eps, got, lr=1e-4, beta1=0.5, want_l1=True, want_latents=False)
    for i in range(10):
        step(i)
    torch.cuda.synchronize()
    ts = []
    for i in range(30):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); step(i); ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    for i in range(200):
        step(i)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'{tag:34s} host enqueue per step, idle queue: median {np.median(ts) * 1e3:.3f} ms (min {min(ts) * 1e3:.3f}); 200 steps: enqueue done after '
          f'{(t1 - t0) * 1e3:.1f} ms, device done after {(t2 - t0) * 1e3:.1f} ms -> {(t2 - t0) * 5:.3f} ms/step', flush=True)


run(f'plain (uad_backward ALL) [{BACKEND}]', DataParallelStep(eng, 1, force_collectives=False))
if os.environ.get('HT_ONLY_PLAIN') or BACKEND == 'none':
    if BACKEND != 'none':
        dist.destroy_process_group()
    sys.exit(0)
for defer in (False, True):
    for noar in (True, False):
        dp = DataParallelStep(eng, 1, force_collectives=True, no_allreduce=noar, library_allreduce=False)
        dp.defer = defer
        run(f'torch PG: segmented defer={int(defer)} allreduce={int(not noar)}', dp)
for b in (2, 1):
    dp = DataParallelStep(eng, 1, force_collectives=True, buckets=b, library_allreduce=False)
    run(f'torch PG: defer=1 buckets={b}', dp)
# round 5: the library enqueues ncclAllReduce itself (uad_allreduce_attach / uad_backward_allreduce); UAD_AR_STREAM=own in the environment gives the
# collectives a stream of their own instead of the handle's side stream
if BACKEND == 'nccl':
    for b in (4, 2, 1):
        dp = DataParallelStep(eng, 1, force_collectives=True, buckets=b, library_allreduce=True)
        assert dp.comm is not None
        run(f'library RCCL ({os.environ.get("UAD_AR_STREAM", "side") + " stream"}): buckets={b}', dp)
    run(f'plain again (uad_backward ALL) [{BACKEND}]', DataParallelStep(eng, 1, force_collectives=False))
dist.destroy_process_group()
