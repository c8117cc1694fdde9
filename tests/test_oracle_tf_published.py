"""CPU: the oracle's primitives against expectations published in TensorFlow r1.15's own unit tests (tests/golden/tf_published.json,
written by tests/golden/make_tf_published.py; every entry names the TF test file and case it restates).  This is the pin of SURVEY.md
section 8a's semantics notes that the image allows without TensorFlow: SAME padding puts the odd element at the END (conv_ops_test.py's
Stride2Same / KernelSmallerThanStrideSame tables), Conv2DTranspose is the input-gradient of that SAME conv (conv2d_transpose_test.py),
LeakyReLU / ReLU gradients select on features > 0, tf.losses / tf.image.total_variation values, and the update rules + slot
initialisation of the four optimizers DLMODEL.create_optimizer accepts (trainers/DLMODEL.py:112-131)."""
import json
import os

import numpy as np
import pytest

from oracle import fanogan as ofg
from oracle import gmvae as og
from oracle import nn

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'tf_published.json')) as fh:
    ENTRIES = json.load(fh)['entries']


def _ramp(shape):
    return np.arange(1, int(np.prod(shape)) + 1, dtype=np.float64).reshape(shape)


def _by(op):
    es = [e for e in ENTRIES if e['op'] == op]
    assert es, op
    return [pytest.param(e, id=e['source'].split('::')[-1][:70]) for e in es]


def test_fixture_covers_the_primitives():
    ops = {e['op'] for e in ENTRIES}
    assert ops >= {'conv2d', 'conv2d_backprop_input', 'conv2d_backprop_filter', 'conv2d_transpose_same_ones', 'leaky_relu', 'leaky_relu_grad',
                   'absolute_difference_mean', 'mean_squared_error_mean', 'total_variation', 'sgd', 'momentum', 'rmsprop', 'adam',
                   'batch_norm_inference', 'layer_norm_hw'}
    assert sum(e['kind'] == 'literal' for e in ENTRIES) >= 30 and all(e['source'].startswith('tensorflow/') for e in ENTRIES)


@pytest.mark.parametrize('e', _by('conv2d'))
def test_conv2d_forward(e):
    y = nn.conv2d_fwd(_ramp(e['in_sizes']), _ramp(e['filter_sizes']), None, e['strides'], e['padding'])
    np.testing.assert_allclose(y.ravel(), e['expected'], rtol=1e-12)


@pytest.mark.parametrize('e', _by('conv2d_backprop_input'))
def test_conv2d_backprop_input(e):
    dx, _, _ = nn.conv2d_bwd(np.zeros(e['in_sizes']), _ramp(e['filter_sizes']), _ramp(e['out_sizes']), e['strides'], e['padding'])
    np.testing.assert_allclose(dx.ravel(), e['expected'], rtol=1e-12)


@pytest.mark.parametrize('e', _by('conv2d_backprop_filter'))
def test_conv2d_backprop_filter(e):
    _, dw, _ = nn.conv2d_bwd(_ramp(e['in_sizes']), np.zeros(e['filter_sizes']), _ramp(e['out_sizes']), e['strides'], e['padding'])
    np.testing.assert_allclose(dw.ravel(), e['expected'], rtol=1e-12)


@pytest.mark.parametrize('e', _by('conv2d_transpose_same_ones'))
def test_conv2d_transpose_same(e):
    y = nn.conv2d_transpose_fwd(np.ones(e['x_shape']), np.ones(e['f_shape']), None, e['stride'])
    assert list(y.shape) == e['expected_shape']
    np.testing.assert_allclose(y.ravel(), e['expected'], rtol=1e-12)
    # ... and it is the adjoint of the SAME convolution with the same kernel array ([kh, kw, out_ch, in_ch] read as HWIO of the conv
    # that maps the big image to the small one): <convT(x), g> == <x, conv(g)>
    rng = np.random.default_rng(0)
    x, w = rng.standard_normal(e['x_shape']), rng.standard_normal(e['f_shape'])
    g = rng.standard_normal(e['expected_shape'])
    lhs = (nn.conv2d_transpose_fwd(x, w, None, e['stride']) * g).sum()
    rhs = (x * nn.conv2d_fwd(g, w, None, e['stride'])).sum()
    assert lhs == pytest.approx(rhs, rel=1e-12)


def test_conv2d_transpose_k5_s2_is_the_adjoint_of_the_pinned_same_conv():
    """The layer the reference uses (customlayers.py:34: k5, stride 2, 'same'): its geometry follows from the SAME rule pinned above."""
    rng = np.random.default_rng(1)
    x, w, g = rng.standard_normal((2, 4, 4, 3)), rng.standard_normal((5, 5, 2, 3)), rng.standard_normal((2, 8, 8, 2))
    y = nn.conv2d_transpose_fwd(x, w, None, 2)
    assert y.shape == (2, 8, 8, 2)
    assert (y * g).sum() == pytest.approx((x * nn.conv2d_fwd(g, w, None, 2)).sum(), rel=1e-12)
    assert nn.same_pads(8, 5, 2) == (4, 1, 2) and nn.same_pads(8, 3, 2) == (4, 0, 1) and nn.same_pads(8, 4, 2) == (4, 1, 1)


@pytest.mark.parametrize('e', _by('leaky_relu'))
def test_leaky_relu_values(e):
    np.testing.assert_allclose(nn.leaky_relu_fwd(np.asarray(e['x'], np.float64), e['alpha']), e['expected'], rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize('e', _by('leaky_relu_grad'))
def test_leaky_relu_grad_rule(e):
    x, g = np.asarray(e['x'], np.float64), np.asarray(e['g'], np.float64)
    np.testing.assert_allclose(nn.leaky_relu_bwd(x, g, e['alpha']), e['expected'], rtol=1e-12)
    np.testing.assert_allclose(nn.leaky_relu_bwd(x, g, 0.0), [0, 0, 0, 1, 1])          # ReluGrad


def test_losses():
    (a,), (m,) = [e for e in ENTRIES if e['op'] == 'absolute_difference_mean'], [e for e in ENTRIES if e['op'] == 'mean_squared_error_mean']
    p, l = np.asarray(a['predictions'], np.float64), np.asarray(a['labels'], np.float64)
    assert np.abs(p - l).mean() == pytest.approx(a['expected']) and ((p - l) ** 2).mean() == pytest.approx(m['expected'])


def test_total_variation():
    (e,) = [e for e in ENTRIES if e['op'] == 'total_variation']
    img = np.asarray(e['images'], np.float64)
    np.testing.assert_allclose(og.total_variation(img), e['expected'], rtol=1e-12)
    # its gradient helper is the derivative of that function
    rng = np.random.default_rng(2)
    r, d = rng.standard_normal((2, 5, 6, 1)), rng.standard_normal((2, 5, 6, 1))
    h = 1e-6
    num = (og.total_variation(r + h * d) - og.total_variation(r - h * d)) / (2 * h)
    np.testing.assert_allclose((og.total_variation_grad(r) * d).sum(axis=(1, 2, 3)), num, rtol=1e-6)


def test_sgd_and_momentum():
    (e,) = [e for e in ENTRIES if e['op'] == 'sgd']
    for v, g, want in zip(e['var'], e['grad'], e['expected']):
        v = np.asarray(v, np.float64)
        nn.sgd_tf_step(v, np.asarray(g, np.float64), e['lr'])
        np.testing.assert_allclose(v, want, rtol=1e-12)
    (e,) = [e for e in ENTRIES if e['op'] == 'momentum']
    for v, g, want, wacc in zip(e['var'], e['grad'], e['expected'], e['expected_accum']):
        v, acc = np.asarray(v, np.float64), np.zeros(2)
        for _ in range(e['steps']):
            nn.momentum_tf_step(v, np.asarray(g, np.float64), acc, e['lr'], e['momentum'])
        np.testing.assert_allclose(v, want, rtol=1e-12)
        np.testing.assert_allclose(acc, wacc, rtol=1e-12)


@pytest.mark.parametrize('e', _by('rmsprop'))
def test_rmsprop(e):
    for k, (v, g, want, wrms) in enumerate(zip(e['var'], e['grad'], e['expected'], e['expected_rms'])):
        v, ms, mom = np.asarray(v, np.float64), np.ones(2), np.zeros(2)          # the rms slot starts at ONE
        for _ in range(e['steps']):
            nn.rmsprop_tf_step(v, np.asarray(g, np.float64), ms, mom, e['lr'], momentum=e['momentum'], decay=e['decay'], eps=e['epsilon'])
        np.testing.assert_allclose(v, want, rtol=1e-12)
        np.testing.assert_allclose(ms, wrms, rtol=1e-12)
        if 'expected_mom' in e:
            np.testing.assert_allclose(mom, e['expected_mom'][k], rtol=1e-12)


@pytest.mark.parametrize('e', _by('adam'))
def test_adam(e):
    for k, (v, g) in enumerate(zip(e['var'], e['grad'])):
        v, m, s = np.asarray(v, np.float64), np.zeros(2), np.zeros(2)
        for t in range(1, e['steps'] + 1):
            nn.adam_tf_step(v, np.asarray(g, np.float64), m, s, t, e['lr'], e['beta1'], e['beta2'], e['epsilon'])
            np.testing.assert_allclose(v, e['expected_trajectory'][t - 1][k], rtol=1e-12)


def test_batch_norm_inference_with_initial_moving_statistics():
    (e,) = [e for e in ENTRIES if e['op'] == 'batch_norm_inference']
    y = nn.bn_frozen_fwd(np.asarray(e['x']), np.asarray(e['gamma']), np.asarray(e['beta']), e['epsilon'])
    np.testing.assert_allclose(y, e['expected'], rtol=1e-12)
    assert nn.BN_EPS == e['epsilon']


def test_layer_normalization_hw():
    (e,) = [e for e in ENTRIES if e['op'] == 'layer_norm_hw']
    y, _ = ofg.ln_fwd(np.asarray(e['x']), np.asarray(e['gamma']), np.asarray(e['beta']), e['epsilon'])
    np.testing.assert_allclose(y, e['expected'], rtol=1e-12)
    assert ofg.LN_EPS == e['epsilon']
