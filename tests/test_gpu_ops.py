"""GPU parity: each HIP kernel family (through the C-ABI op entry points) vs the fp64 numpy oracle.
Tolerance: 1e-4 max-norm relative (north_star's fp32 bar)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import nn as onn

pytestmark = pytest.mark.gpu

try:
    from unsupervised_anomaly_detection_brain_mri_amd import _lib
    from tests.gpu_util import dev, ptr, xform, desc, assert_close, stream
except Exception:  # collected on CPU boxes too
    _lib = None


def lib():
    return _lib.load()


def _act(c, scale, shift, alpha):
    bn = c * scale + shift
    return np.where(bn > 0, bn, alpha * bn)


# (N, H, Cin, Cout) for a k5 s2 SAME Conv2D: big = input HxH, small = output H/2
CONV_CASES = [(2, 16, 32, 64), (3, 8, 128, 128), (1, 32, 64, 32), (2, 10, 16, 48), (5, 4, 32, 20)]


@pytest.mark.parametrize('N,H,Cin,Cout', CONV_CASES)
@pytest.mark.parametrize('with_xf', [False, True])
def test_conv_f_k5s2(N, H, Cin, Cout, with_xf):
    rng = np.random.default_rng(1)
    x = rng.standard_normal((N, H, H, Cin))
    w = rng.standard_normal((5, 5, Cin, Cout)) / np.sqrt(25 * Cin)
    b = rng.standard_normal(Cout)
    oh, pt, _ = onn.same_pads(H, 5, 2)
    sc, sh = rng.uniform(0.5, 1.5, Cin), rng.uniform(-0.5, 0.5, Cin)
    xin = _act(x, sc, sh, 0.3) if with_xf else x
    ref = onn.conv2d_fwd(xin, w, b, 2)
    d = desc(N, H, H, Cin, oh, oh, Cout, 5, 2, pt)
    xd, wd, bd = dev(x), dev(w), dev(b)
    out = torch.empty((N, oh, oh, Cout), device='cuda')
    xf, keep = xform(sc, sh, 0.3) if with_xf else (None, ())
    _lib.check(lib().uad_op_conv_f(C.byref(d), ptr(xd), C.byref(xf) if xf else None, ptr(wd), ptr(bd), None, None,
                                   ptr(out), stream()))
    assert_close(out.cpu().numpy(), ref, name='conv_f')


@pytest.mark.parametrize('N,H,Cin,Cout', [(2, 8, 128, 128), (2, 16, 64, 32), (1, 32, 32, 32), (3, 5, 16, 64), (2, 8, 48, 20)])
@pytest.mark.parametrize('with_xf', [False, True])
def test_conv_d_transpose_k5s2(N, H, Cin, Cout, with_xf):
    """Conv2DTranspose forward: small = input HxH (Cin), big = output 2Hx2H (Cout); kernel [5,5,Cout,Cin]."""
    rng = np.random.default_rng(2)
    x = rng.standard_normal((N, H, H, Cin))
    w = rng.standard_normal((5, 5, Cout, Cin)) / np.sqrt(25 * Cin / 4)
    b = rng.standard_normal(Cout)
    sc, sh = rng.uniform(0.5, 1.5, Cin), rng.uniform(-0.5, 0.5, Cin)
    xin = _act(x, sc, sh, 0.0) if with_xf else x
    ref = onn.conv2d_transpose_fwd(xin, w, b, 2)
    d = desc(N, 2 * H, 2 * H, Cout, H, H, Cin, 5, 2, 1)
    xd, wd, bd = dev(x), dev(w), dev(b)
    out = torch.empty((N, 2 * H, 2 * H, Cout), device='cuda')
    xf, keep = xform(sc, sh, 0.0) if with_xf else (None, ())
    _lib.check(lib().uad_op_conv_d(C.byref(d), ptr(xd), C.byref(xf) if xf else None, ptr(wd), ptr(bd), None, None,
                                   ptr(out), stream()))
    assert_close(out.cpu().numpy(), ref, name='conv_d')


@pytest.mark.parametrize('N,H,Cin,Cout', CONV_CASES)
def test_conv_w_conv2d(N, H, Cin, Cout):
    """Conv2D filter gradient: big = activated layer input, small = dL/dc."""
    rng = np.random.default_rng(3)
    x = rng.standard_normal((N, H, H, Cin))
    oh, pt, _ = onn.same_pads(H, 5, 2)
    g = rng.standard_normal((N, oh, oh, Cout))
    sc, sh = rng.uniform(0.5, 1.5, Cin), rng.uniform(-0.5, 0.5, Cin)
    xin = _act(x, sc, sh, 0.3)
    w0 = np.zeros((5, 5, Cin, Cout))
    _, ref, _ = onn.conv2d_bwd(xin, w0, g, 2)
    d = desc(N, H, H, Cin, oh, oh, Cout, 5, 2, pt)
    xd, gd = dev(x), dev(g)
    dw = torch.empty((5, 5, Cin, Cout), device='cuda')
    xf, keep = xform(sc, sh, 0.3)
    _lib.check(lib().uad_op_conv_w(C.byref(d), ptr(xd), C.byref(xf), ptr(gd), None, ptr(dw), stream()))
    assert_close(dw.cpu().numpy(), ref, name='conv_w')


@pytest.mark.parametrize('N,H,Cin,Cout', [(2, 8, 128, 128), (2, 16, 64, 32), (1, 32, 32, 32), (3, 5, 16, 64)])
def test_conv_w_transpose(N, H, Cin, Cout):
    """Conv2DTranspose filter gradient: big = dL/dy (raw), small = activated layer input."""
    rng = np.random.default_rng(4)
    x = rng.standard_normal((N, H, H, Cin))
    g = rng.standard_normal((N, 2 * H, 2 * H, Cout))
    sc, sh = rng.uniform(0.5, 1.5, Cin), rng.uniform(-0.5, 0.5, Cin)
    xin = _act(x, sc, sh, 0.3)
    w0 = np.zeros((5, 5, Cout, Cin))
    _, ref, _ = onn.conv2d_transpose_bwd(xin, w0, g, 2)
    d = desc(N, 2 * H, 2 * H, Cout, H, H, Cin, 5, 2, 1)
    xd, gd = dev(x), dev(g)
    dw = torch.empty((5, 5, Cout, Cin), device='cuda')
    xf, keep = xform(sc, sh, 0.3)
    _lib.check(lib().uad_op_conv_w(C.byref(d), ptr(gd), None, ptr(xd), C.byref(xf), ptr(dw), stream()))
    assert_close(dw.cpu().numpy(), ref, name='conv_w_T')


@pytest.mark.parametrize('N,H,Cin,Cout', [(2, 16, 32, 64), (3, 8, 128, 128), (1, 32, 32, 32)])
def test_conv_d_bwdact_is_conv2d_data_grad(N, H, Cin, Cout):
    """Conv2D data gradient fused with the producer's BN+LeakyReLU backward and its gamma/beta sums."""
    rng = np.random.default_rng(5)
    cprev = rng.standard_normal((N, H, H, Cin))            # pre-BN output of the producer layer
    sc, sh = rng.uniform(0.5, 1.5, Cin), rng.uniform(-0.5, 0.5, Cin)
    bn = cprev * sc + sh
    a = np.where(bn > 0, bn, 0.3 * bn)
    w = rng.standard_normal((5, 5, Cin, Cout)) / np.sqrt(25 * Cin)
    oh, pt, _ = onn.same_pads(H, 5, 2)
    g = rng.standard_normal((N, oh, oh, Cout))
    da, _, _ = onn.conv2d_bwd(a, w, g, 2)
    dbn = np.where(bn > 0, da, 0.3 * da)
    ref_dc = dbn * sc
    ref_s1 = dbn.reshape(-1, Cin).sum(0)
    ref_s2 = (dbn * cprev).reshape(-1, Cin).sum(0)
    d = desc(N, H, H, Cin, oh, oh, Cout, 5, 2, pt)
    gd, wd, cd = dev(g), dev(w), dev(cprev)
    out = torch.empty((N, H, H, Cin), device='cuda')
    s1 = torch.empty(Cin, device='cuda'); s2 = torch.empty(Cin, device='cuda')
    act, keep = xform(sc, sh, 0.3)
    _lib.check(lib().uad_op_conv_d_bwdact(C.byref(d), ptr(gd), ptr(wd), ptr(cd), C.byref(act), ptr(out), ptr(s1),
                                          ptr(s2), stream()))
    assert_close(out.cpu().numpy(), ref_dc, name='d_c')
    assert_close(s1.cpu().numpy(), ref_s1, tol=3e-4, name='S1')
    assert_close(s2.cpu().numpy(), ref_s2, tol=3e-4, name='S2')


@pytest.mark.parametrize('N,H,Cin,Cout', [(2, 8, 128, 128), (2, 16, 64, 32), (1, 32, 32, 32)])
def test_conv_f_bwdact_is_transpose_data_grad(N, H, Cin, Cout):
    rng = np.random.default_rng(6)
    cprev = rng.standard_normal((N, H, H, Cin))
    sc, sh = rng.uniform(0.5, 1.5, Cin), rng.uniform(-0.5, 0.5, Cin)
    bn = cprev * sc + sh
    a = np.where(bn > 0, bn, 0.0)
    w = rng.standard_normal((5, 5, Cout, Cin)) / np.sqrt(25 * Cin / 4)
    g = rng.standard_normal((N, 2 * H, 2 * H, Cout))
    da, _, _ = onn.conv2d_transpose_bwd(a, w, g, 2)
    dbn = np.where(bn > 0, da, 0.0)
    d = desc(N, 2 * H, 2 * H, Cout, H, H, Cin, 5, 2, 1)
    gd, wd, cd = dev(g), dev(w), dev(cprev)
    out = torch.empty((N, H, H, Cin), device='cuda')
    s1 = torch.empty(Cin, device='cuda'); s2 = torch.empty(Cin, device='cuda')
    act, keep = xform(sc, sh, 0.0)
    _lib.check(lib().uad_op_conv_f_bwdact(C.byref(d), ptr(gd), ptr(wd), ptr(cd), C.byref(act), ptr(out), ptr(s1),
                                          ptr(s2), stream()))
    assert_close(out.cpu().numpy(), dbn * sc, name='d_c')
    assert_close(s1.cpu().numpy(), dbn.reshape(-1, Cin).sum(0), tol=3e-4, name='S1')
    assert_close(s2.cpu().numpy(), (dbn * cprev).reshape(-1, Cin).sum(0), tol=3e-4, name='S2')


@pytest.mark.parametrize('rows,K,Nout', [(64, 1024, 128), (8, 128, 1024), (5, 16, 32), (4096, 128, 16), (4096, 16, 128)])
def test_dense_and_1x1_all_three_kernels(rows, K, Nout):
    """Dense / 1x1 conv = the KS=1 S=1 case of F (forward), D (data grad) and W (weight grad); mul/add epilogues."""
    rng = np.random.default_rng(7)
    x = rng.standard_normal((rows, K)); w = rng.standard_normal((K, Nout)) / np.sqrt(K); b = rng.standard_normal(Nout)
    mul = rng.uniform(0, 2, (rows, Nout)); add = rng.standard_normal((rows, K))
    g = rng.standard_normal((rows, Nout))
    d = desc(rows, 1, 1, K, 1, 1, Nout, 1, 1, 0)
    xd, wd, bd, md, ad, gd = dev(x), dev(w), dev(b), dev(mul), dev(add), dev(g)
    y = torch.empty((rows, Nout), device='cuda')
    _lib.check(lib().uad_op_conv_f(C.byref(d), ptr(xd), None, ptr(wd), ptr(bd), ptr(md), None, ptr(y), stream()))
    assert_close(y.cpu().numpy(), (x @ w + b) * mul, name='dense fwd')
    dx = torch.empty((rows, K), device='cuda')
    _lib.check(lib().uad_op_conv_d(C.byref(d), ptr(gd), None, ptr(wd), None, None, ptr(ad), ptr(dx), stream()))
    assert_close(dx.cpu().numpy(), g @ w.T + add, name='dense dgrad')
    dw = torch.empty((K, Nout), device='cuda')
    _lib.check(lib().uad_op_conv_w(C.byref(d), ptr(xd), None, ptr(gd), None, ptr(dw), stream()))
    assert_close(dw.cpu().numpy(), x.T @ g, name='dense wgrad')


@pytest.mark.parametrize('N,H,Cout', [(2, 16, 32), (3, 128, 32), (1, 10, 64), (5, 64, 32), (2, 256, 32), (64, 128, 32)])
def test_conv_first_fwd_and_wgrad(N, H, Cout):
    rng = np.random.default_rng(8)
    x = rng.uniform(0, 1, (N, H, H, 1)); w = rng.standard_normal((5, 5, 1, Cout)) / 5; b = rng.standard_normal(Cout)
    oh, pt, _ = onn.same_pads(H, 5, 2)
    ref = onn.conv2d_fwd(x, w, b, 2)
    g = rng.standard_normal((N, oh, oh, Cout))
    _, ref_dw, _ = onn.conv2d_bwd(x, w, g, 2)
    d = desc(N, H, H, 1, oh, oh, Cout, 5, 2, pt)
    xd, wd, bd, gd = dev(x), dev(w), dev(b), dev(g)
    out = torch.empty((N, oh, oh, Cout), device='cuda')
    _lib.check(lib().uad_op_conv_first_fwd(C.byref(d), ptr(xd), ptr(wd), ptr(bd), ptr(out), stream()))
    assert_close(out.cpu().numpy(), ref, name='first fwd')
    dw = torch.empty((5, 5, 1, Cout), device='cuda')
    _lib.check(lib().uad_op_conv_first_wgrad(C.byref(d), ptr(xd), ptr(gd), ptr(dw), stream()))
    assert_close(dw.cpu().numpy(), ref_dw, name='first wgrad')


def test_adam_matches_tf_form():
    rng = np.random.default_rng(9)
    n = 100003
    p = rng.standard_normal(n).astype(np.float32); g = rng.standard_normal(n).astype(np.float32)
    m = rng.standard_normal(n).astype(np.float32) * 0.1; v = rng.uniform(0, 1, n).astype(np.float32)
    pr, mr, vr = p.astype(np.float64), m.astype(np.float64), v.astype(np.float64)
    t = 7
    onn.adam_tf_step(pr, g.astype(np.float64) * 0.5, mr, vr, t, lr=1e-3, beta1=0.5)
    lr_t = 1e-3 * np.sqrt(1 - 0.999 ** t) / (1 - 0.5 ** t)
    pd, gd, md, vd = dev(p), dev(g), dev(m), dev(v)
    _lib.check(lib().uad_op_adam(ptr(pd), ptr(gd), ptr(md), ptr(vd), n, lr_t, 0.5, 0.999, 1e-8, 0.5, stream()))
    assert_close(pd.cpu().numpy(), pr, tol=1e-6, name='p')
    assert_close(md.cpu().numpy(), mr, tol=1e-6, name='m')
    assert_close(vd.cpu().numpy(), vr, tol=1e-6, name='v')


def test_residual_map_matches_evaluation_formula():
    rng = np.random.default_rng(10)
    n, h = 3, 64
    x = rng.uniform(0, 1, (n, h, h, 1)); xr = rng.uniform(0, 1, (n, h, h, 1))
    mask = (rng.uniform(0, 1, (n, h, h, 1)) > 0.3).astype(np.float64)
    prior = float(np.quantile(x, 0.9))
    for pos_only in (1, 0):
        ref = np.maximum(x - xr, 0) if pos_only else np.abs(x - xr)
        ref = ref * mask
        ref[x.astype(np.float32) < np.float32(prior)] = 0
        xd, rd, md = dev(x), dev(xr), dev(mask)
        out = torch.empty((n, h, h, 1), device='cuda'); l1 = torch.empty(n, device='cuda')
        _lib.check(lib().uad_residual(ptr(xd), ptr(rd), ptr(md), n, h * h, pos_only, prior, ptr(out), ptr(l1), stream()))
        assert_close(out.cpu().numpy(), ref, tol=1e-6, name='residual')
        assert_close(l1.cpu().numpy(), np.abs(x - xr).reshape(n, -1).sum(1), tol=1e-5, name='l1err')
