"""Helpers for the -m gpu parity tests: call the C-ABI (include/uad_hip.h) through ctypes with torch device buffers."""
import ctypes as C

import numpy as np
import torch

from unsupervised_anomaly_detection_brain_mri_amd import _lib

# parity bar from BASELINE.json north_star: "within 1e-4 rel fp32" (norm-relative: max|a-b| <= tol * max|b|)
REL_TOL = 1e-4


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def xform(scale=None, shift=None, alpha=1.0):
    if scale is None:
        return None, ()
    s, h = dev(scale), dev(shift)
    return _lib.UadXform(ptr(s), ptr(h), float(alpha)), (s, h)


def desc(N, HB, WB, CB, HS, WS, CS, KS, S, P):
    return _lib.UadConvDesc(N, HB, WB, CB, HS, WS, CS, KS, S, P)


def assert_close(actual, expected, tol=REL_TOL, name=''):
    actual = np.asarray(actual, np.float64)
    expected = np.asarray(expected, np.float64)
    assert actual.shape == expected.shape, f'{name}: shape {actual.shape} vs {expected.shape}'
    scale = max(np.abs(expected).max(), 1e-30)
    err = np.abs(actual - expected).max() / scale
    assert np.isfinite(actual).all(), f'{name}: non-finite values'
    assert err <= tol, f'{name}: max-norm relative error {err:.3e} > {tol:.1e}'
    return err


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
