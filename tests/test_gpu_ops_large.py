"""GPU parity at the BENCH shapes: the kernels / plans that the headline run actually launches (N = 64, VAE 128x128 layer
geometries; and the 256x256 GMVAE tail at N = 8) are checked here, one contraction at a time, against PyTorch's own fp32 GPU
convolutions (an independent implementation; the numpy oracle would need minutes per case at these sizes).  Single linear
contractions have no activation kink, so the 1e-4 max-norm bar applies at any batch size -- unlike whole-model gradients, whose
ReLU / LeakyReLU derivative flips between any two fp32 implementations once 10^5..10^6 pre-activations are in play
(tests/test_gpu_shapes.py).  Run in both math modes (UAD_MATH is read by the op entry points; the f32 mode is the default)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

try:
    from unsupervised_anomaly_detection_brain_mri_amd import _lib
    from tests.gpu_util import ptr, desc, assert_close, stream
except Exception:
    _lib = None

torch.backends.cudnn.allow_tf32 = False
torch.backends.cuda.matmul.allow_tf32 = False

# (N, H_small, C_big, C_small): big = 2H x 2H x C_big, small = H x H x C_small; encoder conv: big = input; decoder ConvT: big = output
ENC = [(64, 32, 32, 64), (64, 16, 64, 128), (64, 8, 128, 128), (8, 64, 32, 64)]
DEC = [(64, 8, 128, 128), (64, 16, 64, 128), (64, 32, 32, 64), (64, 64, 32, 32), (8, 128, 32, 32)]


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def _nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def _same_conv(x_nhwc, w_hwio, stride=2):
    xp = F.pad(_nchw(x_nhwc), (1, 2, 1, 2))
    return _nhwc(F.conv2d(xp, w_hwio.permute(3, 2, 0, 1).contiguous(), None, stride=stride))


def _same_convT(x_nhwc, w_hwoi):
    y = F.conv_transpose2d(_nchw(x_nhwc), w_hwoi.permute(3, 2, 0, 1).contiguous(), None, stride=2, padding=0)
    h = x_nhwc.shape[1] * 2
    return _nhwc(y[:, :, 1:1 + h, 1:1 + h])


def _rand(*shape, scale=1.0, seed=0):
    g = torch.Generator(device='cuda').manual_seed(seed)
    return torch.randn(*shape, device='cuda', generator=g) * scale


def _mode_env(math):
    if math == 'bf16x3':
        os.environ['UAD_MATH'] = 'bf16x3'
    else:
        os.environ.pop('UAD_MATH', None)


@pytest.mark.parametrize('math', ['f32', 'bf16x3'])
@pytest.mark.parametrize('N,H,CB,CS', ENC)
def test_encoder_conv_fwd_dgrad_wgrad_at_bench_shapes(N, H, CB, CS, math):
    _mode_env(math)
    lib = _lib.load()
    x = _rand(N, 2 * H, 2 * H, CB, seed=1)
    w = _rand(5, 5, CB, CS, scale=1.0 / np.sqrt(25 * CB), seed=2)
    g = _rand(N, H, H, CS, seed=3)
    d = desc(N, 2 * H, 2 * H, CB, H, H, CS, 5, 2, 1)
    # forward (F kind)
    out = torch.empty((N, H, H, CS), device='cuda')
    _lib.check(lib.uad_op_conv_f(C.byref(d), ptr(x), None, ptr(w), None, None, None, ptr(out), stream()))
    ref = _same_conv(x, w)
    assert_close(out.cpu().numpy(), ref.cpu().numpy(), name='fwd')
    # data gradient (D kind) and filter gradient (W kind) from autograd of the torch reference
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    (_same_conv(xr, wr) * g).sum().backward()
    dx = torch.empty_like(x)
    _lib.check(lib.uad_op_conv_d(C.byref(d), ptr(g), None, ptr(w), None, None, None, ptr(dx), stream()))
    assert_close(dx.cpu().numpy(), xr.grad.cpu().numpy(), name='dgrad')
    dw = torch.empty_like(w)
    _lib.check(lib.uad_op_conv_w(C.byref(d), ptr(x), None, ptr(g), None, ptr(dw), stream()))
    assert_close(dw.cpu().numpy(), wr.grad.cpu().numpy(), name='wgrad')
    os.environ.pop('UAD_MATH', None)


@pytest.mark.parametrize('math', ['f32', 'bf16x3'])
@pytest.mark.parametrize('N,H,CB,CS', DEC)
def test_decoder_convT_fwd_dgrad_wgrad_at_bench_shapes(N, H, CB, CS, math):
    _mode_env(math)
    lib = _lib.load()
    a = _rand(N, H, H, CS, seed=4)                       # layer input (small)
    w = _rand(5, 5, CB, CS, scale=1.0 / np.sqrt(25 * CS / 4), seed=5)     # [kh,kw,Cout,Cin]
    g = _rand(N, 2 * H, 2 * H, CB, seed=6)               # d loss / d output (big)
    d = desc(N, 2 * H, 2 * H, CB, H, H, CS, 5, 2, 1)
    out = torch.empty((N, 2 * H, 2 * H, CB), device='cuda')
    _lib.check(lib.uad_op_conv_d(C.byref(d), ptr(a), None, ptr(w), None, None, None, ptr(out), stream()))
    assert_close(out.cpu().numpy(), _same_convT(a, w).cpu().numpy(), name='fwd')
    ar = a.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    (_same_convT(ar, wr) * g).sum().backward()
    da = torch.empty_like(a)
    _lib.check(lib.uad_op_conv_f(C.byref(d), ptr(g), None, ptr(w), None, None, None, ptr(da), stream()))
    assert_close(da.cpu().numpy(), ar.grad.cpu().numpy(), name='dgrad')
    dw = torch.empty_like(w)
    _lib.check(lib.uad_op_conv_w(C.byref(d), ptr(g), None, ptr(a), None, ptr(dw), stream()))
    assert_close(dw.cpu().numpy(), wr.grad.cpu().numpy(), name='wgrad')
    os.environ.pop('UAD_MATH', None)
