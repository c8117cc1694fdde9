"""GPU: the fallback paths behind the library's environment switches stay correct.  The switches are read once per process, so each case
runs the whole-model parity tests of tests/test_gpu_model.py (forward, every gradient tensor, Adam trajectory vs the oracle) in a fresh
interpreter with the switch set:
  UAD_BOTT_Q1              one workgroup per sample in the fused bottleneck kernels instead of a group of four
  UAD_NO_FUSED_BOTT_WGRAD  bottleneck parameter gradients as batched GEMMs + column sums instead of bottleneck_wgrad_kernel
  UAD_NO_FIRST32           generic first-layer kernels instead of conv_first_fwd32 / conv_first_wgrad32
  UAD_NO_SIDE_PACK         weight repack on the caller's stream inside the next forward instead of on the side stream after the optimizer step
  UAD_NO_INKERNEL_SPLITK   split-K slabs summed by splitk_epilogue_kernel launches instead of the conv kernels' last-arriver reduction
  UAD_NO_D16S              round-2 ConvT-class kernel (LDS-transposed epilogue) instead of the lane = pixel one
  UAD_NO_D16S_T16          fused-final training instance of the lane = pixel kernel on 8 x 16 tiles (one fragment per wave) instead of 16 x 16 (two)
  UAD_NO_REDUCE4           slab reductions (filter-gradient slabs, LayerNorm parameter partials) on the scalar-load kernel instead of the 16-byte streaming one
  UAD_NO_TAIL_SPLIT        one main -> side edge behind both kernels of the last encoder block instead of one behind its filter gradient and one behind its data gradient
  UAD_NO_FUSED_FINAL_F32   exact-fp32 mode: separate final 1x1 conv + loss kernel instead of the last ConvT's fused epilogue (round 4)
  UAD_NO_W5_DB             k5 filter gradient of the 64-column layers on the single-buffered tile loop (two barriers per tile) instead of the double-buffered one
  UAD_NO_ANYORDER          every launch with the AQL barrier bit (no data gradient starting while its layer's filter gradient drains)
  UAD_K3_FORM=0..3         kernel form of the k3 / k1 tap-list launches (round 6: first kernel | pipelined 32 x 32 wave tiles | 64 x 32 wave tiles with 64 / 128
                           output channels per workgroup); the library picks one per launch shape, the forms are bit-identical (test_k3_forms_*)
  UAD_K3_WFORM=0           k3 filter gradient on the first kernel instead of the pipelined twelve-wave one (bit-identical slabs)
  UAD_NO_LN1               two-pass LayerNorm kernels on every map size instead of the one-pass register-resident forms (UAD_NO_LNQ / UAD_NO_LN2Q: only the clustered
                           forward + backward of the 64 x 64 maps / only the second-order adjoint)
(The opt-in experiments of round 3 -- UAD_PG, UAD_PP, UAD_D16S_MF2, UAD_W_TW8, UAD_W5_MINTILES, UAD_STAGGER -- measured slower or neutral and
were removed in round 4; round 6 removed every path that had been the non-default for two rounds -- UAD_NO_F16, UAD_NO_D16, UAD_NO_W_T, UAD_NO_W_TR,
UAD_NO_W2, UAD_NO_FB_BITS, UAD_NO_FB_ON_LOAD, UAD_NO_REDUCE_NT, UAD_NO_PACK8, UAD_EVENT_SYSFENCE -- with their kernels; git history has them.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('knob', ['UAD_BOTT_Q1', 'UAD_NO_FUSED_BOTT_WGRAD', 'UAD_NO_FIRST32', 'UAD_NO_SIDE_PACK', 'UAD_NO_INKERNEL_SPLITK',
                                  'UAD_NO_D16S', 'UAD_NO_D16S_T16', 'UAD_NO_REDUCE4', 'UAD_NO_TAIL_SPLIT', 'UAD_NO_ANYORDER', 'UAD_NO_FUSED_FINAL_F32', 'UAD_NO_PACK_HEAD', 'UAD_NO_W5_DB',
                                  'UAD_SPATIAL_MIN_WGS=256'])
def test_model_parity_with_switch(knob):
    name, _, val = knob.partition('=')
    env = dict(os.environ, **{name: val or '1'})
    sel = 'test_forward_backward_parity or test_train_trajectory_vae_matches_oracle'
    r = subprocess.run([sys.executable, '-m', 'pytest', 'tests/test_gpu_model.py', '-q', '-x', '-m', 'gpu', '-k', sel, '-p', 'no:cacheprovider'],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (knob, r.stdout[-2000:], r.stderr[-1000:])
    assert ' passed' in r.stdout and 'failed' not in r.stdout


_FAULT_SCRIPT = r'''
import numpy as np, torch
from unsupervised_anomaly_detection_brain_mri_amd.engine import Engine
eng = Engine('VAE', 128, 128, 1, 8, 128, max_batch=4)          # a shape whose bottleneck runs as groups of four workgroups per sample
x = np.random.default_rng(0).random((4, 128, 128, 1), dtype=np.float32)
eps = np.zeros((4, 128), np.float32)
eng.forward(x, eps, None)             # UAD_BOTT_FAULT: workgroup 1 of sample 0 never publishes its flag -> its siblings give up after a few seconds
torch.cuda.synchronize()
try:
    eng.forward(x, eps, None)
    print('NO-ERROR')
except RuntimeError as e:
    print('REPORTED' if 'gave up waiting' in str(e) else 'OTHER: ' + str(e))
eng.forward(x, eps, None)             # the report is consumed: the handle keeps working (this launch times out again, which the NEXT call would report)
torch.cuda.synchronize()
print('DONE')
'''


_ANYORDER_SCRIPT = r'''
import sys, numpy as np, torch
from unsupervised_anomaly_detection_brain_mri_amd.engine import Engine
from unsupervised_anomaly_detection_brain_mri_amd.utils.synthetic import synthetic_slices
eng = Engine('VAE', 128, 128, 1, 8, 128, max_batch=16, math='bf16x3')
rng = np.random.default_rng(0)
flat = rng.standard_normal(eng.nparams).astype(np.float32) * 0.05
eng.set_params(flat)
outs = []
for step in range(3):          # fresh inputs every step: a slab left over from the previous step would be a WRONG value, not the same one
    x = synthetic_slices(16, 128, 128, seed=10 + step)
    eps = np.random.default_rng(100 + step).standard_normal((16, 128)).astype(np.float32)
    eng.forward(x, eps, None, want_backward=True)
    eng.backward()
    torch.cuda.synchronize()
    outs.append(eng.get_buffer_host(1).copy())          # BUF_GRADS
np.save(sys.argv[1], np.stack(outs))
'''


def test_any_order_edge_waits_for_the_slower_filter_gradient(tmp_path):
    """a layer's data gradient is launched without the AQL barrier bit behind its filter gradient, and the layer's ONE stream edge is
    recorded after both.  The side stream's slab reduction is only safe if that event waits for EVERY earlier dispatch, not just the last one
    (uad_model.hip: edge).  With UAD_W_ABL=32 every filter-gradient workgroup sleeps ~100 us before its slab store, so the filter gradient
    outlasts the data gradient in every layer; on fresh inputs per step the gradients must equal, bit for bit, a fully ordered run's."""
    res = {}
    for tag, env in (('ordered', {'UAD_NO_ANYORDER': '1'}), ('anyorder_slow_w', {'UAD_W_ABL': '32'})):
        out = str(tmp_path / (tag + '.npy'))
        r = subprocess.run([sys.executable, '-c', _ANYORDER_SCRIPT, out], cwd=ROOT, env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (tag, r.stdout[-2000:], r.stderr[-2000:])
        import numpy as np
        res[tag] = np.load(out)
    assert res['ordered'].shape == res['anyorder_slow_w'].shape and np.isfinite(res['ordered']).all()
    assert np.array_equal(res['ordered'], res['anyorder_slow_w']), 'gradients differ: the layer edge did not wait for the slower filter gradient'
    assert not np.array_equal(res['ordered'][0], res['ordered'][1])          # the steps really had different gradients


def test_bottleneck_sibling_exchange_is_bounded_and_reports():
    """The fused bottleneck kernels exchange partial vectors between the four workgroups of a sample through flags in global memory (uad_bott.hip:
    group_exchange).  A sibling that never arrives must not hang the device: the wait is bounded and the failure reaches the caller."""
    env = dict(os.environ, UAD_BOTT_FAULT='1')
    r = subprocess.run([sys.executable, '-c', _FAULT_SCRIPT], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert 'REPORTED' in r.stdout and 'DONE' in r.stdout, (r.stdout[-2000:], r.stderr[-1000:])


def test_restoration_parity_without_the_pattern_word():
    """Round 5: restoration iterations hand the last block's activation pattern to the data gradient as one word per pixel (uad_model.hip: restore_bits)
    instead of the block's pre-BN output.  UAD_NO_RESTORE_BITS=1 keeps the round-2 path; both must meet the oracle."""
    env = dict(os.environ, UAD_NO_RESTORE_BITS='1')
    r = subprocess.run([sys.executable, '-m', 'pytest', 'tests/test_gpu_gmvae.py', 'tests/test_gpu_vae_you.py', '-q', '-x', '-m', 'gpu', '-k', 'restore', '-p', 'no:cacheprovider'],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-1000:])
    assert ' passed' in r.stdout and 'failed' not in r.stdout


def test_layernorm_two_pass_kernels_still_meet_the_oracle():
    """Round 6: on maps of up to 1024 pixels the LayerNorm forward / backward kernels of the f-AnoGAN graphs keep a workgroup's (sample, 32-channel)
    slice in registers between the statistics sweep and the apply sweep (one pass over HBM, same accumulation order).  UAD_NO_LN1=1 keeps the two-pass
    kernels everywhere (UAD_NO_LNQ=1: on the 64 x 64 maps only, where the one-pass form is a cluster of workgroups that exchange partial sums); that
    path must still meet the oracle on the ResNet critic (first- and second-order LayerNorm adjoints, gamma / beta gradients)
    and on the unified graphs."""
    env = dict(os.environ, UAD_NO_LN1='1')
    r = subprocess.run([sys.executable, '-m', 'pytest', 'tests/test_gpu_fanogan.py', '-q', '-x', '-m', 'gpu', '-k', 'resnet_critic_phase or test_critic_phase or resnet_generator_phase',
                        '-p', 'no:cacheprovider'], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-1000:])
    assert ' passed' in r.stdout and 'failed' not in r.stdout


_LNQ_SCRIPT = r"""
import hashlib, numpy as np, torch
from tests.test_gpu_fanogan import _setup_rn, _engine_rn
from unsupervised_anomaly_detection_brain_mri_amd import _lib
m, p, x, z, alpha = _setup_rn(64, 32, 32, 3, seed=4)
eng = _engine_rn(m, p, 3, 'bf16x3')
h = hashlib.sha256()
for it in range(2):
    eng.phase('Discriminator', x=x, z=z, alpha=alpha)
    h.update(np.ascontiguousarray(eng.get_buffer_host(_lib.BUF_GRADS)).tobytes())
    eng.phase('Generator', z=z)
    h.update(np.ascontiguousarray(eng.get_buffer_host(_lib.BUF_GRADS)).tobytes())
torch.cuda.synchronize()
print('DIGEST', h.hexdigest())
"""


def test_layernorm_cluster_without_a_sibling_recomputes_the_same_bits():
    """Round 6: on the 64 x 64 maps the one-pass LayerNorm is a cluster of workgroups per (sample, 32-channel) slice that exchange their partial sums through
    global memory (bounded poll).  A sibling that never raises its flag (UAD_LNQ_FAULT=1: workgroup 1 of every slice) must cost time only: the others
    recompute its partial vector through the same lane mapping, so the critic's and the generator's gradients come out bit-identical."""
    digests = {}
    for knob in ('', 'UAD_LNQ_FAULT=1'):
        env = dict(os.environ)
        if knob:
            k, v = knob.split('=')
            env[k] = v
        r = subprocess.run([sys.executable, '-c', _LNQ_SCRIPT], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (knob, r.stdout[-2000:], r.stderr[-2000:])
        digests[knob] = [l for l in r.stdout.splitlines() if l.startswith('DIGEST')][0]
    assert digests[''] == digests['UAD_LNQ_FAULT=1'], digests


_K3_SCRIPT = r'''
import ctypes as C, hashlib, os, numpy as np, torch
os.environ['UAD_MATH'] = 'bf16x3'
from unsupervised_anomaly_detection_brain_mri_amd import _lib
from tests.gpu_util import dev, ptr, desc, stream
from oracle import nn as onn
lib = _lib.load()
h = hashlib.sha256()
rng = np.random.default_rng(5)
# (kind, N, H, Cin, Cout, k, s): k3 s1 / s2 convolutions and transposed convolutions, a k1 shortcut; N and Cout large enough for every form's grid rule to matter
for kind, N, H, Cin, Cout, k, s in [('conv', 5, 16, 128, 128, 3, 1), ('conv', 4, 16, 64, 128, 3, 2), ('convT', 4, 8, 128, 128, 3, 2), ('conv', 4, 16, 128, 256, 1, 1),
                                    ('convT', 3, 8, 256, 128, 3, 1), ('conv', 2, 8, 512, 512, 3, 1)]:
    if kind == 'conv':
        x = rng.standard_normal((N, H, H, Cin)); w = rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)
        oh, pt, _ = onn.same_pads(H, k, s)
        g = rng.standard_normal((N, oh, oh, Cout))
        d = desc(N, H, H, Cin, oh, oh, Cout, k, s, pt)
        xd, wd, gd = dev(x), dev(w), dev(g)
        out = torch.empty((N, oh, oh, Cout), device='cuda'); dx = torch.empty((N, H, H, Cin), device='cuda'); dw = torch.empty((k, k, Cin, Cout), device='cuda')
        _lib.check(lib.uad_op_conv_f(C.byref(d), ptr(xd), None, ptr(wd), None, None, None, ptr(out), stream()))
        _lib.check(lib.uad_op_conv_d(C.byref(d), ptr(gd), None, ptr(wd), None, None, None, ptr(dx), stream()))
        _lib.check(lib.uad_op_conv_w(C.byref(d), ptr(xd), None, ptr(gd), None, ptr(dw), stream()))
    else:
        x = rng.standard_normal((N, H, H, Cin)); w = rng.standard_normal((k, k, Cout, Cin)) / np.sqrt(k * k * Cin)
        OH = H * s
        g = rng.standard_normal((N, OH, OH, Cout))
        _, pt, _ = onn.same_pads(OH, k, s)
        d = desc(N, OH, OH, Cout, H, H, Cin, k, s, pt)
        xd, wd, gd = dev(x), dev(w), dev(g)
        out = torch.empty((N, OH, OH, Cout), device='cuda'); dx = torch.empty((N, H, H, Cin), device='cuda'); dw = torch.empty((k, k, Cout, Cin), device='cuda')
        _lib.check(lib.uad_op_conv_d(C.byref(d), ptr(xd), None, ptr(wd), None, None, None, ptr(out), stream()))
        _lib.check(lib.uad_op_conv_f(C.byref(d), ptr(gd), None, ptr(wd), None, None, None, ptr(dx), stream()))
        _lib.check(lib.uad_op_conv_w(C.byref(d), ptr(gd), None, ptr(xd), None, ptr(dw), stream()))
    torch.cuda.synchronize()
    for t in (out, dx, dw):
        a = t.cpu().numpy()
        assert np.isfinite(a).all()
        h.update(a.tobytes())
print('DIGEST', h.hexdigest())
'''


def test_k3_forms_are_bit_identical():
    """Round 6: the pipelined forms of the k3 tap-list kernel and of the k3 filter gradient keep the first kernels' products and summation order --
    forcing any form (UAD_K3_FORM, UAD_K3_WFORM) over forward, data gradient and filter gradient of k3 s1 / s2 / transposed / k1 launches gives the
    SAME BITS as the library's own per-shape choice."""
    digests = {}
    for knob in ('', 'UAD_K3_FORM=0', 'UAD_K3_FORM=1', 'UAD_K3_FORM=2', 'UAD_K3_FORM=3', 'UAD_K3_WFORM=0'):
        env = dict(os.environ)
        if knob:
            name, _, val = knob.partition('=')
            env[name] = val
        r = subprocess.run([sys.executable, '-c', _K3_SCRIPT], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and 'DIGEST' in r.stdout, (knob, r.stdout[-2000:], r.stderr[-2000:])
        digests[knob or 'default'] = r.stdout.split('DIGEST')[1].split()[0]
    assert len(set(digests.values())) == 1, digests


@pytest.mark.parametrize('knob', ['UAD_K3_FORM=0', 'UAD_K3_FORM=2', 'UAD_K3_FORM=3', 'UAD_K3_WFORM=0'])
def test_k3_op_parity_with_forced_form(knob):
    """... and each forced form meets the fp64 oracle on the op-level parity tests of the ResNet geometries (the default choice at these small batches is
    the pipelined 32 x 32 form; the 64 x 32 forms are what the bench-size launches run)."""
    name, _, val = knob.partition('=')
    env = dict(os.environ, **{name: val})
    r = subprocess.run([sys.executable, '-m', 'pytest', 'tests/test_gpu_ops_resnet.py', '-q', '-x', '-m', 'gpu', '-p', 'no:cacheprovider'],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (knob, r.stdout[-2000:], r.stderr[-1000:])
    assert ' passed' in r.stdout and 'failed' not in r.stdout
