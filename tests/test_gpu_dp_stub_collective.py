"""GPU: ORDERING of the library-issued gradient all-reduce (include/uad_hip.h: uad_allreduce_attach / uad_backward_allreduce), checked with a collective that
really changes the data.  Over the one rank a one-GPU box has, RCCL's in-place all-reduce is a no-op, so tests/test_gpu_dp_nccl.py cannot see a collective that
was enqueued too early.  Here libuad_hip.so binds tests/native/stub_rccl.hip instead of librccl (UAD_RCCL_LIB): its ncclAllReduce is a kernel on the stream it is
given that doubles the buffer -- the sum over two ranks with identical gradients.  With DataParallelStep(world=2) the optimizer scales by 1/2, and x * 2 * 0.5 is
exact in fp32: after three steps the parameters must equal the plain single-process step's BIT FOR BIT, for every bucketing, on the side-stream placement and
on the own-stream one.  A bucket all-reduced before its slab reductions had written it would come out un-doubled (half the update)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r'''
import ctypes, os, sys, numpy as np, torch, torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', sys.argv[1])
torch.cuda.set_device(0)
dist.init_process_group('gloo', rank=0, world_size=1)          # bootstrap channel only: one rank, nothing is sent
from unsupervised_anomaly_detection_brain_mri_amd import _lib
from unsupervised_anomaly_detection_brain_mri_amd.engine import Engine
from unsupervised_anomaly_detection_brain_mri_amd.parallel import DataParallelStep
from unsupervised_anomaly_detection_brain_mri_amd.utils.synthetic import synthetic_slices
stub = ctypes.CDLL(os.environ['UAD_RCCL_LIB'])
stub.stub_rccl_calls.restype = ctypes.c_longlong; stub.stub_rccl_elems.restype = ctypes.c_longlong
n, h, z = 8, 128, 128
w0 = None
ends = {}
for tag, kw in [('plain', None)] + [(f'lib{b}', dict(buckets=b, library_allreduce=True)) for b in (4, 3, 2, 1)]:
    eng = Engine('VAE', h, h, 1, 8, z, max_batch=n, math='bf16x3')
    if w0 is None:
        w0 = (np.random.default_rng(1).standard_normal(eng.nparams) * 0.05).astype(np.float32)
    eng.set_params(w0)
    dp = DataParallelStep(eng, world=1 if kw is None else 2, **(kw or {}))
    assert (dp.comm is not None) == (kw is not None)
    c0, e0 = stub.stub_rccl_calls(), stub.stub_rccl_elems()
    for step in range(3):
        x = synthetic_slices(n, h, h, seed=20 + step)
        eps = np.random.default_rng(50 + step).standard_normal((n, z)).astype(np.float32)
        dp.train_step(x, eps, None, lr=1e-3, beta1=0.5)
    torch.cuda.synchronize()
    if kw is not None:      # every bucket went through the stub once per step, and together they cover the whole gradient buffer
        assert stub.stub_rccl_calls() - c0 == 3 * kw['buckets'], (tag, stub.stub_rccl_calls() - c0)
        assert stub.stub_rccl_elems() - e0 == 3 * eng.nparams, (tag, stub.stub_rccl_elems() - e0, eng.nparams)
    ends[tag] = eng.get_buffer_host(_lib.BUF_PARAMS).copy()
    eng.close()
ref = ends['plain']
assert np.isfinite(ref).all() and not np.array_equal(ref, w0)
for tag, w in ends.items():
    assert np.array_equal(w, ref), (tag, float(np.abs(w - ref).max()), int((w != ref).sum()))
print('STUB_COLLECTIVE_OK', os.environ.get('UAD_AR_STREAM', 'side'))
dist.destroy_process_group()
'''


def _build_stub(tmp_path):
    so = str(tmp_path / 'libstub_rccl.so')
    hipcc = os.environ.get('HIPCC') or '/opt/rocm/bin/hipcc'
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-O2', '-shared', '-fPIC', os.path.join(ROOT, 'tests', 'native', 'stub_rccl.hip'), '-o', so])
    return so


@pytest.mark.parametrize('placement', ['side', 'own'])
def test_library_allreduce_runs_where_the_gradients_are_final(tmp_path, placement):
    import socket
    so = _build_stub(tmp_path)
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, UAD_RCCL_LIB=so, GPU_MAX_HW_QUEUES='8')
    if placement == 'own':
        env['UAD_AR_STREAM'] = 'own'
    r = subprocess.run([sys.executable, '-c', _SCRIPT, str(port)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert 'STUB_COLLECTIVE_OK ' + placement in r.stdout, r.stdout[-2000:]


_GAN_SCRIPT = r'''
import ctypes, os, sys, numpy as np, torch, torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', sys.argv[1])
torch.cuda.set_device(0)
dist.init_process_group('gloo', rank=0, world_size=1)          # bootstrap channel only
from unsupervised_anomaly_detection_brain_mri_amd import _lib
from unsupervised_anomaly_detection_brain_mri_amd.gan_engine import GanEngine
from unsupervised_anomaly_detection_brain_mri_amd.parallel import GanDataParallel
from unsupervised_anomaly_detection_brain_mri_amd.utils.synthetic import synthetic_slices
stub = ctypes.CDLL(os.environ['UAD_RCCL_LIB'])
stub.stub_rccl_calls.restype = ctypes.c_longlong; stub.stub_rccl_elems.restype = ctypes.c_longlong
n, h, zd, dim = 4, 64, 64, 32
w0 = None
ends = {}
for tag in ('plain', 'lib'):
    eng = GanEngine(h, h, 1, 8, zd, max_batch=n, math='bf16x3', variant='resnet', dim=dim)
    if w0 is None:
        w0 = (np.random.default_rng(1).standard_normal(eng.nparams) * 0.05).astype(np.float32)
    eng.set_buffer_host(_lib.BUF_PARAMS, w0) if hasattr(eng, 'set_buffer_host') else eng.set_params(eng.unflatten(w0))
    dp = GanDataParallel(eng, world=1 if tag == 'plain' else 2, library_allreduce=(tag == 'lib'), force_collectives=(tag == 'lib'))
    assert dp.in_phase == (tag == 'lib')
    c0, e0 = stub.stub_rccl_calls(), stub.stub_rccl_elems()
    want = 0
    for it in range(2):
        rng = np.random.default_rng(100 + it)
        x = synthetic_slices(n, h, h, seed=30 + it)
        for k in range(2):                                   # critic steps, then one generator step, then one encoder step: every trained group
            z = rng.standard_normal((n, zd)).astype(np.float32); alpha = rng.random(n).astype(np.float32)
            dp.train_phase('Discriminator', 1e-4, x=x, z=z, alpha=alpha); want += eng.group('Discriminator')[1]
        z = rng.standard_normal((n, zd)).astype(np.float32)
        dp.train_phase('Generator', 1e-4, z=z); want += eng.group('Generator')[1]
        dp.train_phase('Encoder', 1e-4, x=x); want += eng.group('Encoder')[1]
    torch.cuda.synchronize()
    if tag == 'lib':     # every trained slice went through the stub exactly once per phase, in at most four buckets
        calls, elems = stub.stub_rccl_calls() - c0, stub.stub_rccl_elems() - e0
        assert elems == want, (elems, want)
        assert 8 <= calls <= 8 * 4, calls
        assert calls > 8, 'the ResNet phases are expected to go out in several buckets'
    ends[tag] = eng.get_buffer_host(_lib.BUF_PARAMS).copy()
    dp.close(); eng.close()
ref = ends['plain']
assert np.isfinite(ref).all() and not np.array_equal(ref, w0)
assert np.array_equal(ends['lib'], ref), (float(np.abs(ends['lib'] - ref).max()), int((ends['lib'] != ref).sum()))
print('STUB_GAN_COLLECTIVE_OK')
dist.destroy_process_group()
'''


def test_gan_phases_allreduce_their_buckets_where_the_gradients_are_final(tmp_path):
    """uad_gan_allreduce_attach (round 6): the ResNet f-AnoGAN handle all-reduces the trained group's slice itself, in buckets issued per residual block while the
    backward of the earlier blocks still runs.  With the doubling stand-in for librccl and world = 2 (Adam halves: exact) two WGAN iterations -- critic, generator
    and encoder phases -- must end on the plain phases' parameters bit for bit; a bucket sent before its last gradient kernel would come out un-doubled."""
    import socket
    so = _build_stub(tmp_path)
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, UAD_RCCL_LIB=so, GPU_MAX_HW_QUEUES='8')
    r = subprocess.run([sys.executable, '-c', _GAN_SCRIPT, str(port)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert 'STUB_GAN_COLLECTIVE_OK' in r.stdout, r.stdout[-2000:]
