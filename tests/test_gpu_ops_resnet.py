"""GPU parity of the F / D / W contractions at the geometries of the ResNet f-AnoGAN graph (models/fanogan_schlegl.py):
k3 s1 and k3 s2 SAME convolutions, k3 s1 / k3 s2 / k1 s2 transposed convolutions, their data and filter gradients, and the
Cin = 1 k3 first layer -- through the C-ABI op entry points vs the fp64 numpy oracle (1e-4 max-norm relative), in BOTH math modes of the op
entry points: 'f32' = the generic exact-fp32 kernels, 'bf16x3' (UAD_MATH=bf16x3, read per call) = the round-4 tap-list spatial kernel
(csrc/uad_convk16.inc) wherever its shape conditions hold, the generic bf16x3 kernels elsewhere."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import nn as onn

pytestmark = pytest.mark.gpu

try:
    from unsupervised_anomaly_detection_brain_mri_amd import _lib
    from tests.gpu_util import dev, ptr, desc, assert_close, stream
except Exception:
    _lib = None


def lib():
    return _lib.load()


@pytest.fixture(params=['f32', 'bf16x3'], autouse=True)
def math_mode(request):
    old = os.environ.get('UAD_MATH')
    if request.param == 'bf16x3':
        os.environ['UAD_MATH'] = 'bf16x3'
    else:
        os.environ.pop('UAD_MATH', None)
    yield request.param
    if old is None:
        os.environ.pop('UAD_MATH', None)
    else:
        os.environ['UAD_MATH'] = old


# Conv2D (big = input): (N, H, Cin, Cout, k, s)
CONV = [(2, 16, 64, 128, 3, 1), (3, 8, 128, 128, 3, 2), (2, 16, 32, 64, 3, 2), (1, 32, 64, 64, 3, 1), (2, 8, 256, 256, 3, 1),
        (2, 16, 64, 128, 1, 1), (2, 16, 128, 256, 3, 2), (1, 64, 64, 128, 3, 1), (2, 8, 512, 512, 3, 1),
        (2, 16, 32, 64, 3, 1), (3, 8, 64, 64, 3, 2), (2, 16, 128, 128, 1, 1)]      # (round 6: one-chunk k3 s1, 64 x 64-channel s2 filter gradient, four-chunk k1)


@pytest.mark.parametrize('N,H,Cin,Cout,k,s', CONV)
def test_conv2d_generic_fwd_dgrad_wgrad(N, H, Cin, Cout, k, s):
    rng = np.random.default_rng(k * 10 + s)
    x = rng.standard_normal((N, H, H, Cin))
    w = rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)
    b = rng.standard_normal(Cout)
    oh, pt, _ = onn.same_pads(H, k, s)
    g = rng.standard_normal((N, oh, oh, Cout))
    ref = onn.conv2d_fwd(x, w, b, s)
    dx_ref, dw_ref, _ = onn.conv2d_bwd(x, w, g, s)
    d = desc(N, H, H, Cin, oh, oh, Cout, k, s, pt)
    xd, wd, bd, gd = dev(x), dev(w), dev(b), dev(g)
    out = torch.empty((N, oh, oh, Cout), device='cuda')
    _lib.check(lib().uad_op_conv_f(C.byref(d), ptr(xd), None, ptr(wd), ptr(bd), None, None, ptr(out), stream()))
    assert_close(out.cpu().numpy(), ref, name='fwd')
    dx = torch.empty((N, H, H, Cin), device='cuda')
    _lib.check(lib().uad_op_conv_d(C.byref(d), ptr(gd), None, ptr(wd), None, None, None, ptr(dx), stream()))
    assert_close(dx.cpu().numpy(), dx_ref, name='dgrad')
    dw = torch.empty((k, k, Cin, Cout), device='cuda')
    _lib.check(lib().uad_op_conv_w(C.byref(d), ptr(xd), None, ptr(gd), None, ptr(dw), stream()))
    assert_close(dw.cpu().numpy(), dw_ref, name='wgrad')


# Conv2DTranspose (big = output): (N, H, Cin, Cout, k, s)
CONVT = [(2, 8, 128, 128, 3, 1), (2, 8, 128, 64, 3, 2), (3, 16, 64, 32, 3, 2), (2, 8, 128, 64, 1, 2), (1, 16, 64, 32, 1, 2), (2, 8, 512, 256, 3, 2),
         (1, 32, 128, 64, 3, 2), (2, 8, 512, 512, 3, 1)]


@pytest.mark.parametrize('N,H,Cin,Cout,k,s', CONVT)
def test_conv2d_transpose_generic_fwd_dgrad_wgrad(N, H, Cin, Cout, k, s):
    rng = np.random.default_rng(100 + k * 10 + s)
    x = rng.standard_normal((N, H, H, Cin))
    w = rng.standard_normal((k, k, Cout, Cin)) / np.sqrt(k * k * Cin)
    b = rng.standard_normal(Cout)
    OH = H * s
    _, pt, _ = onn.same_pads(OH, k, s)
    g = rng.standard_normal((N, OH, OH, Cout))
    ref = onn.conv2d_transpose_fwd(x, w, b, s)
    dx_ref, dw_ref, _ = onn.conv2d_transpose_bwd(x, w, g, s)
    d = desc(N, OH, OH, Cout, H, H, Cin, k, s, pt)
    xd, wd, bd, gd = dev(x), dev(w), dev(b), dev(g)
    out = torch.empty((N, OH, OH, Cout), device='cuda')
    _lib.check(lib().uad_op_conv_d(C.byref(d), ptr(xd), None, ptr(wd), ptr(bd), None, None, ptr(out), stream()))
    assert_close(out.cpu().numpy(), ref, name='convT fwd')
    dx = torch.empty((N, H, H, Cin), device='cuda')
    _lib.check(lib().uad_op_conv_f(C.byref(d), ptr(gd), None, ptr(wd), None, None, None, ptr(dx), stream()))
    assert_close(dx.cpu().numpy(), dx_ref, name='convT dgrad')
    dw = torch.empty((k, k, Cout, Cin), device='cuda')
    _lib.check(lib().uad_op_conv_w(C.byref(d), ptr(gd), None, ptr(xd), None, ptr(dw), stream()))
    assert_close(dw.cpu().numpy(), dw_ref, name='convT wgrad')


@pytest.mark.parametrize('N,H,Cout', [(2, 16, 64), (3, 64, 64), (1, 32, 32)])
def test_first_layer_k3s1(N, H, Cout):
    rng = np.random.default_rng(7)
    x = rng.standard_normal((N, H, H, 1))
    w = rng.standard_normal((3, 3, 1, Cout)) / 3.0
    b = rng.standard_normal(Cout)
    g = rng.standard_normal((N, H, H, Cout))
    ref = onn.conv2d_fwd(x, w, b, 1)
    _, dw_ref, _ = onn.conv2d_bwd(x, w, g, 1)
    d = desc(N, H, H, 1, H, H, Cout, 3, 1, 1)
    xd, wd, bd, gd = dev(x), dev(w), dev(b), dev(g)
    out = torch.empty((N, H, H, Cout), device='cuda')
    _lib.check(lib().uad_op_conv_first_fwd(C.byref(d), ptr(xd), ptr(wd), ptr(bd), ptr(out), stream()))
    assert_close(out.cpu().numpy(), ref, name='first fwd')
    dw = torch.empty((3, 3, 1, Cout), device='cuda')
    _lib.check(lib().uad_op_conv_first_wgrad(C.byref(d), ptr(xd), ptr(gd), ptr(dw), stream()))
    assert_close(dw.cpu().numpy(), dw_ref, name='first wgrad')


@pytest.mark.parametrize('kind,N,H,Cin,Cout,s', [('conv', 2, 16, 128, 128, 1), ('conv', 2, 16, 128, 256, 2), ('convT', 2, 8, 256, 128, 2), ('conv', 1, 8, 512, 512, 1)])
def test_bf16x6_products_are_fp32_grade(kind, N, H, Cin, Cout, s, math_mode):
    """UAD_MATH=bf16x6: the k3 tap-list kernel with THREE bf16 planes per operand and six products (uad_convk16.inc) -- what the ResNet graph's
    exact passes run on.  Its error against the fp64 oracle has to be of the exact-fp32 kernel's order (<= 3x + 1e-7 of the output's max) and
    well below the bf16x3 kernel's (< 1/2) on the same inputs, forward (F or D kind) and data gradient (the other kind)."""
    if math_mode != 'f32':
        pytest.skip('runs its three modes itself')
    rng = np.random.default_rng(N * 1000 + H + Cin)
    k = 3
    if kind == 'conv':
        x = rng.standard_normal((N, H, H, Cin)); w = rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)
        oh, pt, _ = onn.same_pads(H, k, s)
        g = rng.standard_normal((N, oh, oh, Cout))
        ref_f = onn.conv2d_fwd(x, w, None, s); ref_d = onn.conv2d_bwd(x, w, g, s)[0]
        d = desc(N, H, H, Cin, oh, oh, Cout, k, s, pt)
        f_in, d_in = x, g
    else:
        x = rng.standard_normal((N, H, H, Cin)); w = rng.standard_normal((k, k, Cout, Cin)) / np.sqrt(k * k * Cin)
        OH = H * s
        _, pt, _ = onn.same_pads(OH, k, s)
        g = rng.standard_normal((N, OH, OH, Cout))
        ref_d = onn.conv2d_transpose_fwd(x, w, None, s); ref_f = onn.conv2d_transpose_bwd(x, w, g, s)[0]
        d = desc(N, OH, OH, Cout, H, H, Cin, k, s, pt)
        f_in, d_in = g, x
    wd, fi, di = dev(w), dev(f_in), dev(d_in)
    errs = {}
    for mode in ('f32', 'bf16x3', 'bf16x6'):
        if mode == 'f32':
            os.environ.pop('UAD_MATH', None)
        else:
            os.environ['UAD_MATH'] = mode
        of = torch.empty(ref_f.shape, device='cuda'); od = torch.empty(ref_d.shape, device='cuda')
        _lib.check(lib().uad_op_conv_f(C.byref(d), ptr(fi), None, ptr(wd), None, None, None, ptr(of), stream()))
        _lib.check(lib().uad_op_conv_d(C.byref(d), ptr(di), None, ptr(wd), None, None, None, ptr(od), stream()))
        errs[mode] = (np.abs(of.cpu().numpy() - ref_f).max() / np.abs(ref_f).max(), np.abs(od.cpu().numpy() - ref_d).max() / np.abs(ref_d).max())
    os.environ.pop('UAD_MATH', None)
    print(f'\n[{kind} N={N} H={H} {Cin}->{Cout} s{s}] max-norm relative error  F kind: f32 {errs["f32"][0]:.2e} bf16x3 {errs["bf16x3"][0]:.2e} bf16x6 {errs["bf16x6"][0]:.2e}'
          f' | D kind: f32 {errs["f32"][1]:.2e} bf16x3 {errs["bf16x3"][1]:.2e} bf16x6 {errs["bf16x6"][1]:.2e}')
    for i in (0, 1):
        assert errs['bf16x6'][i] <= 3 * errs['f32'][i] + 1e-7, errs
        assert errs['bf16x6'][i] < 0.5 * errs['bf16x3'][i], errs
