"""Writes tests/golden/tf_published.json: expectations PUBLISHED in TensorFlow r1.15's own unit tests for the primitives the reference's
graphs are made of (reference pins tensorflow==1.15.2, requirements.txt:3; TF itself is not installable here).  Two kinds of entries:

  kind "literal"  -- the numbers are the ones written in the TF test source (restated from the r1.15 tree; each entry names file and
                     test).  The conv tables were additionally re-derived here with a brute-force loop from the test's documented
                     inputs (values 1..N in row-major order, `_VerifyValues` / `_RunAndVerifyBackprop*` of conv_ops_test.py): all
                     agree, so the tables below are self-consistent with TF's SAME rule (extra padding at the END).
  kind "np_ref"   -- TF's test compares the op with a numpy reference function defined in the test file (adam_update_numpy,
                     _npBatchNorm, ...); the entry restates THAT function and stores its output on the test's inputs.

tests/test_oracle_tf_published.py holds oracle/nn.py (and the LayerNorm / total-variation helpers of oracle/fanogan.py, oracle/gmvae.py)
to every entry.  Run:  python tests/golden/make_tf_published.py"""
import json
import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
K = 'tensorflow/python/kernel_tests/'
entries = []


def add(**kw):
    entries.append(kw)


# ---------------------------------------------------------------------------------------------------------------------------------
# Conv2D forward: conv_ops_test.py, Conv2DTest._VerifyValues(tensor_in_sizes, filter_in_sizes, strides, padding, expected):
# input x = 1..prod(in) and filter = 1..prod(filter) (row-major), NHWC / HWIO.
# ---------------------------------------------------------------------------------------------------------------------------------
F = K + 'conv_ops_test.py'
for test, i, f, s, pad, exp in [
    ('testConv2D1x1Filter', [1, 2, 3, 3], [1, 1, 3, 3], [1, 1], 'VALID',
     [30.0, 36.0, 42.0, 66.0, 81.0, 96.0, 102.0, 126.0, 150.0, 138.0, 171.0, 204.0, 174.0, 216.0, 258.0, 210.0, 261.0, 312.0]),
    ('testConv2D2x2Filter', [1, 2, 3, 3], [2, 2, 3, 3], [1, 1], 'VALID', [2271.0, 2367.0, 2463.0, 2901.0, 3033.0, 3165.0]),
    ('testConv2D1x2Filter', [1, 2, 3, 3], [1, 2, 3, 3], [1, 1], 'VALID',
     [231.0, 252.0, 273.0, 384.0, 423.0, 462.0, 690.0, 765.0, 840.0, 843.0, 936.0, 1029.0]),
    ('testConv2D2x2FilterStride2', [1, 2, 3, 3], [2, 2, 3, 3], [2, 2], 'VALID', [2271.0, 2367.0, 2463.0]),
    ('testConv2D2x2FilterStride2Same', [1, 2, 3, 3], [2, 2, 3, 3], [2, 2], 'SAME', [2271.0, 2367.0, 2463.0, 1230.0, 1305.0, 1380.0]),
    ('testConv2D2x2FilterStride1x2', [1, 3, 6, 1], [2, 2, 1, 1], [1, 2], 'VALID', [58.0, 78.0, 98.0, 118.0, 138.0, 158.0]),
    ('testConv2DKernelSmallerThanStrideValid', [1, 7, 7, 1], [2, 2, 1, 1], [3, 3], 'VALID', [65, 95, 275, 305]),
    ('testConv2DKernelSmallerThanStrideSame/0', [1, 3, 3, 1], [1, 1, 1, 1], [2, 2], 'SAME', [1, 3, 7, 9]),
    ('testConv2DKernelSmallerThanStrideSame/1', [1, 4, 4, 1], [1, 1, 1, 1], [2, 2], 'SAME', [1, 3, 9, 11]),
    ('testConv2DKernelSmallerThanStrideSame/2', [1, 4, 4, 1], [2, 2, 1, 1], [3, 3], 'SAME', [44, 28, 41, 16]),
    ('testConv2DKernelSizeMatchesInputSize', [1, 2, 2, 1], [2, 2, 1, 2], [1, 1], 'VALID', [50, 60]),
]:
    add(op='conv2d', kind='literal', source=f'{F}::Conv2DTest.{test}', in_sizes=i, filter_sizes=f, strides=s, padding=pad, expected=exp)

# Conv2DBackpropInput: _RunAndVerifyBackpropInput(input_sizes, filter_sizes, output_sizes, strides, padding, expected):
# filter = 1..prod(filter), out_backprop = 1..prod(output).
for test, i, f, o, s, pad, exp in [
    ('testConv2D2x2Depth1ValidBackpropInput', [1, 2, 3, 1], [2, 2, 1, 1], [1, 1, 2, 1], [1, 1], 'VALID', [1.0, 4.0, 4.0, 3.0, 10.0, 8.0]),
    ('testConv2D2x2Depth3ValidBackpropInput', [1, 2, 3, 3], [2, 2, 3, 3], [1, 1, 2, 3], [1, 1], 'VALID',
     [14.0, 32.0, 50.0, 100.0, 163.0, 226.0, 167.0, 212.0, 257.0, 122.0, 140.0, 158.0, 478.0, 541.0, 604.0, 437.0, 482.0, 527.0]),
    ('testConv2D2x2Depth3ValidBackpropInputStride1x2', [1, 3, 6, 1], [2, 2, 1, 1], [1, 2, 3, 1], [1, 2], 'VALID',
     [1.0, 2.0, 2.0, 4.0, 3.0, 6.0, 7.0, 12.0, 11.0, 18.0, 15.0, 24.0, 12.0, 16.0, 15.0, 20.0, 18.0, 24.0]),
    ('testConv2DStrideTwoFilterOneSameBackpropInput', [1, 4, 4, 1], [1, 1, 1, 1], [1, 2, 2, 1], [2, 2], 'SAME',
     [1.0, 0.0, 2.0, 0.0, 0.0, 0.0, 0.0, 0.0, 3.0, 0.0, 4.0, 0.0, 0.0, 0.0, 0.0, 0.0]),
    ('testConv2DKernelSizeMatchesInputSizeBackpropInput', [1, 2, 2, 1], [2, 2, 1, 2], [1, 1, 1, 2], [1, 1], 'VALID', [5.0, 11.0, 17.0, 23.0]),
]:
    add(op='conv2d_backprop_input', kind='literal', source=f'{F}::Conv2DTest.{test}', in_sizes=i, filter_sizes=f, out_sizes=o, strides=s,
        padding=pad, expected=exp)

# Conv2DBackpropFilter: _RunAndVerifyBackpropFilter: input = 1..prod(input), out_backprop = 1..prod(output).
for test, i, f, o, s, pad, exp in [
    ('testConv2D2x2Depth1ValidBackpropFilter', [1, 2, 3, 1], [2, 2, 1, 1], [1, 1, 2, 1], [1, 1], 'VALID', [5.0, 8.0, 14.0, 17.0]),
    ('testConv2D2x2Depth3ValidBackpropFilter', [1, 2, 3, 3], [2, 2, 3, 3], [1, 1, 2, 3], [1, 1], 'VALID',
     [17.0, 22.0, 27.0, 22.0, 29.0, 36.0, 27.0, 36.0, 45.0, 32.0, 43.0, 54.0, 37.0, 50.0, 63.0, 42.0, 57.0, 72.0, 62.0, 85.0, 108.0, 67.0,
      92.0, 117.0, 72.0, 99.0, 126.0, 77.0, 106.0, 135.0, 82.0, 113.0, 144.0, 87.0, 120.0, 153.0]),
    ('testConv2D2x2Depth3ValidBackpropFilterStride1x2', [1, 3, 6, 1], [2, 2, 1, 1], [1, 2, 3, 1], [1, 2], 'VALID', [161.0, 182.0, 287.0, 308.0]),
    ('testConv2DStrideTwoFilterOneSameBackpropFilter', [1, 4, 4, 1], [1, 1, 1, 1], [1, 2, 2, 1], [2, 2], 'SAME', [78.0]),
    ('testConv2DKernelSizeMatchesInputSizeBackpropFilter', [1, 2, 2, 1], [2, 2, 1, 2], [1, 1, 1, 2], [1, 1], 'VALID',
     [1.0, 2.0, 2.0, 4.0, 3.0, 6.0, 4.0, 8.0]),
]:
    add(op='conv2d_backprop_filter', kind='literal', source=f'{F}::Conv2DTest.{test}', in_sizes=i, filter_sizes=f, out_sizes=o, strides=s,
        padding=pad, expected=exp)

# ---------------------------------------------------------------------------------------------------------------------------------
# conv2d_transpose, padding SAME: conv2d_transpose_test.py.  x = ones [2,6,4,3], f = ones [3,3,2,3] ([kh,kw,out_ch,in_ch]); the test
# spells the expected value of every output element with the loops restated below.
# ---------------------------------------------------------------------------------------------------------------------------------
T = K + 'conv2d_transpose_test.py'


def _ct_single_stride():
    x_shape, y_shape, f_shape = [2, 6, 4, 3], [2, 6, 4, 2], [3, 3, 2, 3]
    y = np.zeros(y_shape)
    for n in range(x_shape[0]):
        for k in range(f_shape[2]):
            for w in range(y_shape[2]):
                for h in range(y_shape[1]):
                    target = 4 * 3.0
                    h_in = 0 < h < y_shape[1] - 1
                    w_in = 0 < w < y_shape[2] - 1
                    if h_in and w_in:
                        target += 5 * 3.0
                    elif h_in or w_in:
                        target += 2 * 3.0
                    y[n, h, w, k] = target
    return x_shape, f_shape, 1, y


def _ct_same_stride2():
    x_shape, y_shape, f_shape, strides = [2, 6, 4, 3], [2, 12, 8, 2], [3, 3, 2, 3], [1, 2, 2, 1]
    y = np.zeros(y_shape)
    for n in range(x_shape[0]):
        for k in range(f_shape[2]):
            for w in range(y_shape[2]):
                for h in range(y_shape[1]):
                    target = 3.0
                    # "We add a case for locations divisible by the stride."
                    h_in = h % strides[1] == 0 and 0 < h < y_shape[1] - 1
                    w_in = w % strides[2] == 0 and 0 < w < y_shape[2] - 1
                    if h_in and w_in:
                        target += 9.0
                    elif h_in or w_in:
                        target += 3.0
                    y[n, h, w, k] = target
    return x_shape, f_shape, 2, y


for test, fn in (('testConv2DTransposeSingleStride', _ct_single_stride), ('testConv2DTransposeSame', _ct_same_stride2)):
    xs, fs, stride, y = fn()
    add(op='conv2d_transpose_same_ones', kind='literal', source=f'{T}::Conv2DTransposeTest.{test}', x_shape=xs, f_shape=fs, stride=stride,
        expected_shape=list(y.shape), expected=y.ravel().tolist())

# ---------------------------------------------------------------------------------------------------------------------------------
# activations
# ---------------------------------------------------------------------------------------------------------------------------------
add(op='leaky_relu', kind='literal', source='tensorflow/python/ops/nn_test.py::LeakyReluTest.testValues (tf.nn.leaky_relu default alpha 0.2)',
    alpha=0.2, x=[-2, -1, 0, 1, 2], expected=[-0.4, -0.2, 0.0, 1.0, 2.0])
add(op='leaky_relu', kind='literal', source=K + 'relu_op_test.py::LeakyReluTest.testNpLeakyRelu (alpha 0.1)', alpha=0.1,
    x=[[-0.9, 0.7, -0.5, 0.3, -0.1], [0.1, -0.3, 0.5, -0.7, 0.9]], expected=[[-0.09, 0.7, -0.05, 0.3, -0.01], [0.1, -0.03, 0.5, -0.07, 0.9]])
# gradient rule: core/kernels/relu_op_functor.h LeakyReluGrad = (features > 0).select(gradients, gradients * alpha); ReluGrad = gradients * (features > 0)
add(op='leaky_relu_grad', kind='np_ref', source='tensorflow/core/kernels/relu_op_functor.h::LeakyReluGrad / ReluGrad (features > 0 selects; the kink belongs to the alpha side)',
    alpha=0.3, x=[-1.5, -0.0, 0.0, 1e-30, 2.0], g=[1.0, 1.0, 1.0, 1.0, 1.0], expected=[0.3, 0.3, 0.3, 1.0, 1.0])

# ---------------------------------------------------------------------------------------------------------------------------------
# losses (tf.losses.absolute_difference / mean_squared_error; the trainers use reduction=NONE, i.e. the elementwise term whose
# mean these tests check): losses_test.py
# ---------------------------------------------------------------------------------------------------------------------------------
add(op='absolute_difference_mean', kind='literal', source=K + 'losses_test.py::AbsoluteDifferenceLossTest.testNonZeroLoss',
    predictions=[[4, 8, 12], [8, 1, 3]], labels=[[1, 9, 2], [-5, -2, 6]], expected=5.5)
add(op='mean_squared_error_mean', kind='literal', source=K + 'losses_test.py::MeanSquaredErrorTest.testNonZeroLoss',
    predictions=[[4, 8, 12], [8, 1, 3]], labels=[[1, 9, 2], [-5, -2, 6]], expected=49.5)

# ---------------------------------------------------------------------------------------------------------------------------------
# tf.image.total_variation: image_ops_test.py TotalVariationTest.testTotalVariationHandmade
# ---------------------------------------------------------------------------------------------------------------------------------
r_, g_, b_ = [[1, 2], [4, 7]], [[11, 18], [29, 47]], [[73, 120], [193, 313]]
a = np.dstack((r_, g_, b_)).astype(np.float64)
add(op='total_variation', kind='literal', source='tensorflow/python/ops/image_ops_test.py::TotalVariationTest.testTotalVariationHandmade',
    images=np.stack([a, a + 1, -a, 1.1 * a, 2 * a]).tolist(), expected=[564.0, 564.0, 564.0, 1.1 * 564.0, 2 * 564.0])

# ---------------------------------------------------------------------------------------------------------------------------------
# optimizers: tensorflow/python/training/{gradient_descent,momentum,rmsprop,adam}_test.py, all on var0 = [1, 2], var1 = [3, 4],
# grads0 = [0.1, 0.1], grads1 = [0.01, 0.01]
# ---------------------------------------------------------------------------------------------------------------------------------
TR = 'tensorflow/python/training/'
add(op='sgd', kind='literal', source=TR + 'gradient_descent_test.py::GradientDescentOptimizerTest.testBasic', lr=3.0,
    var=[[1.0, 2.0], [3.0, 4.0]], grad=[[0.1, 0.1], [0.01, 0.01]], steps=1,
    expected=[[1.0 - 3.0 * 0.1, 2.0 - 3.0 * 0.1], [3.0 - 3.0 * 0.01, 4.0 - 3.0 * 0.01]])
add(op='momentum', kind='literal', source=TR + 'momentum_test.py::MomentumOptimizerTest.testBasic', lr=2.0, momentum=0.9,
    var=[[1.0, 2.0], [3.0, 4.0]], grad=[[0.1, 0.1], [0.01, 0.01]], steps=2,
    expected_accum=[[0.9 * 0.1 + 0.1] * 2, [0.9 * 0.01 + 0.01] * 2],
    expected=[[1.0 - (0.1 * 2.0) - ((0.9 * 0.1 + 0.1) * 2.0), 2.0 - (0.1 * 2.0) - ((0.9 * 0.1 + 0.1) * 2.0)],
              [2.98 - ((0.9 * 0.01 + 0.01) * 2.0), 3.98 - ((0.9 * 0.01 + 0.01) * 2.0)]])
# rmsprop_test.py testWithoutMomentum: RMSPropOptimizer(learning_rate=2.0, decay=0.9, momentum=0.0, epsilon=1.0); "the rms accumulators
# where 1. So we should see a normal update" -- the slot starts at ONE and epsilon sits INSIDE the square root
e = 1.0
add(op='rmsprop', kind='literal', source=TR + 'rmsprop_test.py::RMSPropOptimizerTest.testWithoutMomentum', lr=2.0, decay=0.9, momentum=0.0,
    epsilon=e, var=[[1.0, 2.0], [3.0, 4.0]], grad=[[0.1, 0.1], [0.01, 0.01]], steps=2,
    expected_rms=[[0.901 * 0.9 + 0.001] * 2, [0.90001 * 0.9 + 1e-5] * 2],
    expected=[[1.0 - (0.1 * 2.0 / math.sqrt(0.901 + e)) - (0.1 * 2.0 / math.sqrt(0.901 * 0.9 + 0.001 + e)),
               2.0 - (0.1 * 2.0 / math.sqrt(0.901 + e)) - (0.1 * 2.0 / math.sqrt(0.901 * 0.9 + 0.001 + e))],
              [3.0 - (0.01 * 2.0 / math.sqrt(0.90001 + e)) - (0.01 * 2.0 / math.sqrt(0.90001 * 0.9 + 1e-5 + e)),
               4.0 - (0.01 * 2.0 / math.sqrt(0.90001 + e)) - (0.01 * 2.0 / math.sqrt(0.90001 * 0.9 + 1e-5 + e))]])
e = 1e-5
m0a = 0.1 * 2.0 / math.sqrt(0.901 + e)
m0b = 0.5 * m0a + 0.1 * 2.0 / math.sqrt(0.901 * 0.9 + 0.001 + e)
m1a = 0.01 * 2.0 / math.sqrt(0.90001 + e)
m1b = 0.5 * m1a + 0.01 * 2.0 / math.sqrt(0.90001 * 0.9 + 1e-5 + e)
add(op='rmsprop', kind='literal', source=TR + 'rmsprop_test.py::RMSPropOptimizerTest.testWithMomentum', lr=2.0, decay=0.9, momentum=0.5,
    epsilon=e, var=[[1.0, 2.0], [3.0, 4.0]], grad=[[0.1, 0.1], [0.01, 0.01]], steps=2,
    expected_rms=[[0.901 * 0.9 + 0.001] * 2, [0.90001 * 0.9 + 1e-5] * 2], expected_mom=[[m0b] * 2, [m1b] * 2],
    expected=[[1.0 - m0a - m0b, 2.0 - m0a - m0b], [3.0 - m1a - m1b, 4.0 - m1a - m1b]])


def adam_update_numpy(param, g_t, t, m, v, alpha=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8):
    """adam_test.py: the numpy reference AdamOptimizerTest compares tf.train.AdamOptimizer with."""
    alpha_t = alpha * np.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    m_t = beta1 * m + (1 - beta1) * g_t
    v_t = beta2 * v + (1 - beta2) * g_t * g_t
    param_t = param - alpha_t * m_t / (np.sqrt(v_t) + epsilon)
    return param_t, m_t, v_t


for b1, tag in ((0.9, 'testBasic (defaults)'), (0.5, 'testBasic inputs with the reference\'s beta1 = 0.5 (default_config_setup.py:257)')):
    vs = [np.array([1.0, 2.0]), np.array([3.0, 4.0])]
    gs = [np.array([0.1, 0.1]), np.array([0.01, 0.01])]
    ms, vv = [0.0, 0.0], [0.0, 0.0]
    traj = []
    for t in range(1, 4):
        for k in range(2):
            vs[k], ms[k], vv[k] = adam_update_numpy(vs[k], gs[k], t, ms[k], vv[k], beta1=b1)
        traj.append([v.tolist() for v in vs])
    add(op='adam', kind='np_ref', source=TR + f'adam_test.py::adam_update_numpy on AdamOptimizerTest.{tag}', lr=0.001, beta1=b1, beta2=0.999,
        epsilon=1e-8, var=[[1.0, 2.0], [3.0, 4.0]], grad=[[0.1, 0.1], [0.01, 0.01]], steps=3, expected_trajectory=traj)

# ---------------------------------------------------------------------------------------------------------------------------------
# normalisation layers (formula restatements of the r1.15 sources)
# ---------------------------------------------------------------------------------------------------------------------------------
rng = np.random.default_rng(0)


def _npBatchNorm(x, m, v, beta, gamma, epsilon, scale_after_normalization=True, shift_after_normalization=True):
    """nn_batchnorm_test.py::BatchNormalizationTest._npBatchNorm"""
    y = (x - m) / np.sqrt(v + epsilon)
    y = y * gamma if scale_after_normalization else y
    return y + beta if shift_after_normalization else y


x = rng.standard_normal((2, 3, 3, 4))
gamma, beta = rng.uniform(0.5, 1.5, 4), rng.standard_normal(4)
add(op='batch_norm_inference', kind='np_ref',
    source='tensorflow/python/ops/nn_batchnorm_test.py::_npBatchNorm with the layer\'s never-updated moving statistics '
           '(tensorflow/python/layers/normalization.py BatchNormalization: moving_mean_initializer zeros, moving_variance_initializer ones, '
           'epsilon 1e-3; call(inputs, training=False) -> nn.batch_normalization(inputs, moving_mean, moving_variance, beta, gamma, epsilon))',
    x=x.tolist(), gamma=gamma.tolist(), beta=beta.tolist(), epsilon=1e-3,
    expected=_npBatchNorm(x, 0.0, 1.0, beta, gamma, 1e-3).tolist())

x = rng.standard_normal((2, 4, 3, 2))
gamma, beta = rng.uniform(0.5, 1.5, (4, 3)), rng.standard_normal((4, 3))
mean = x.mean(axis=(1, 2), keepdims=True)
var = ((x - mean) ** 2).mean(axis=(1, 2), keepdims=True)        # nn.moments: population variance
inv = gamma[None, :, :, None] / np.sqrt(var + 1e-3)             # nn.batch_normalization: inv = rsqrt(variance + eps) * scale
add(op='layer_norm_hw', kind='np_ref',
    source='tensorflow/python/keras/layers/normalization.py::LayerNormalization(axis=[1, 2]).call (r1.15): param_shape = [H, W], epsilon 1e-3, '
           'mean, variance = nn.moments(inputs, axis, keep_dims=True); nn.batch_normalization(inputs, mean, variance, offset=beta, scale=gamma, '
           'variance_epsilon) with gamma / beta broadcast to [1, H, W, 1]',
    x=x.tolist(), gamma=gamma.tolist(), beta=beta.tolist(), epsilon=1e-3,
    expected=(x * inv + (beta[None, :, :, None] - mean * inv)).tolist())

with open(os.path.join(HERE, 'tf_published.json'), 'w') as fh:
    json.dump({'tensorflow': 'r1.15', 'entries': entries}, fh, indent=0)
print(f'{len(entries)} entries -> tests/golden/tf_published.json')
