"""GPU: RCCL (torch.distributed backend "nccl" on ROCm) on the ENGINE-OWNED buffers, on a one-GPU box.

Every other multi-process test of this repo runs over gloo; the first time ProcessGroupNCCL would see the handle's gradient buffer -- a ctypes
device pointer wrapped as a torch tensor through __cuda_array_interface__ (engine._DevArray), sliced per gradient bucket -- used to be the
driver's 8-GPU run.  Here a process group of world_size 1 is initialised with backend nccl and parallel.DataParallelStep is driven with
force_collectives=True: segmented backward (DECODER, BOTTLENECK, ENCODER_HI, ENCODER_LO), one ASYNC all-reduce per bucket on RCCL's stream
behind the gradients (ProcessGroupNCCL orders its stream after torch's current stream, which is the stream the engine enqueues on), wait,
Adam with grad_scale 1/world, plus the parameter broadcast and the scalar all-reduce.  An all-reduce over one rank is the identity, so after
three steps the parameters must equal those of the plain single-process step BIT FOR BIT, for every bucketing.
Round 5: the same for the LIBRARY-ISSUED path (include/uad_hip.h: uad_rccl_*, uad_allreduce_attach, uad_backward_allreduce) -- a communicator created by
libuad_hip.so from a unique id exchanged over the torch group, ncclAllReduce enqueued by the library on its own stream behind each bucket's slab reductions
-- and a second test that the torch path refuses an engine older than the process group (the hardware-queue order DESIGN.md section 6 measured)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', sys.argv[1])
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)
from unsupervised_anomaly_detection_brain_mri_amd import _lib
from unsupervised_anomaly_detection_brain_mri_amd.engine import Engine
from unsupervised_anomaly_detection_brain_mri_amd.parallel import DataParallelStep, SEGMENT_ORDER
from unsupervised_anomaly_detection_brain_mri_amd.utils.synthetic import synthetic_slices
n, h, z = 8, 128, 128
rng = np.random.default_rng(0)
w0 = None
ends = {}
# lib*: the library-issued path (uad_allreduce_attach: libuad_hip.so enqueues ncclAllReduce on its own communicator and stream); b*: the torch.distributed path
cases = [('plain', None)] + [(f'lib{b}', dict(buckets=b, library_allreduce=True)) for b in (4, 3, 2, 1)] + [(f'b{b}', dict(buckets=b, library_allreduce=False)) for b in (4, 3, 2, 1)]
for tag, kw in cases:
    eng = Engine('VAE', h, h, 1, 8, z, max_batch=n, math='bf16x3')
    if w0 is None:
        w0 = (np.random.default_rng(1).standard_normal(eng.nparams) * 0.05).astype(np.float32)
    eng.set_params(w0)
    dp = DataParallelStep(eng, world=1, force_collectives=kw is not None, **(kw or {}))
    if kw is not None:
        assert dp.grads is not None and dp.grads.is_cuda and dp.grads.data_ptr() == eng.buffer(_lib.BUF_GRADS).data_ptr()
        dp.broadcast_params(0)                                   # RCCL broadcast THROUGH the parameter view (invalidates the packed weights)
        assert (dp.comm is not None) == tag.startswith('lib'), tag
        if dp.comm is not None:                                  # the library's communicator on a plain torch tensor: identity over one rank
            t = torch.arange(1000, dtype=torch.float32, device='cuda'); dp.comm.allreduce_(t); torch.cuda.synchronize()
            assert torch.equal(t.cpu(), torch.arange(1000, dtype=torch.float32))
    sc = None
    for step in range(3):
        x = synthetic_slices(n, h, h, seed=20 + step)
        eps = np.random.default_rng(50 + step).standard_normal((n, z)).astype(np.float32)
        out = dp.train_step(x, eps, None, lr=1e-3, beta1=0.5)
        sc = dp.allreduce_scalars(out['scalars'].clone())
    torch.cuda.synchronize()
    ends[tag] = (eng.get_buffer_host(_lib.BUF_PARAMS).copy(), sc.cpu().numpy().copy())
    eng.close()
ref_w, ref_s = ends['plain']
assert np.isfinite(ref_w).all() and not np.array_equal(ref_w, w0)
for tag in [t for t, kw in cases if kw is not None]:
    w, s = ends[tag]
    assert np.array_equal(w, ref_w), (tag, float(np.abs(w - ref_w).max()))
    assert np.array_equal(s, ref_s), tag
print('BACKEND', dist.get_backend(), 'NCCL_OK')
dist.destroy_process_group()
'''


def test_rccl_allreduce_on_engine_owned_gradient_buffer():
    import socket
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, '-c', _SCRIPT, str(port)], cwd=ROOT, env=dict(os.environ), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert 'BACKEND nccl NCCL_OK' in r.stdout, r.stdout[-2000:]


_ORDER_SCRIPT = r'''
import os, sys, torch, torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', sys.argv[1])
torch.cuda.set_device(0)
from unsupervised_anomaly_detection_brain_mri_amd.engine import Engine
from unsupervised_anomaly_detection_brain_mri_amd.parallel import DataParallelStep
eng = Engine('VAE', 128, 128, 1, 8, 128, max_batch=4)            # handle first, communicator second: the order the torch path must refuse
dist.init_process_group('nccl', rank=0, world_size=1)
try:
    DataParallelStep(eng, world=1, force_collectives=True, library_allreduce=False)
    print('ACCEPTED')
except RuntimeError as e:
    print('REFUSED' if 'BEFORE the engine' in str(e) else 'OTHER ' + str(e))
dp = DataParallelStep(eng, world=1, force_collectives=True, library_allreduce=True)      # the library-issued path has no such constraint
print('LIB_OK' if dp.comm is not None else 'LIB_MISSING')
dist.destroy_process_group()
'''


def test_torch_path_refuses_an_engine_older_than_the_process_group():
    import socket
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, '-c', _ORDER_SCRIPT, str(port)], cwd=ROOT, env=dict(os.environ), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert 'REFUSED' in r.stdout and 'LIB_OK' in r.stdout, r.stdout[-2000:]
