import sys, os
sys.path.insert(0, '.')
import tests.test_gpu_ops as T
cases = [('f_bwdact', T.test_conv_f_bwdact_is_transpose_data_grad, (1, 128, 32, 32)),
         ('f_bwdact64', T.test_conv_f_bwdact_is_transpose_data_grad, (1, 64, 32, 32)),
         ('d_bwdact', T.test_conv_d_bwdact_is_conv2d_data_grad, (1, 128, 32, 64)),
         ('d_fwd', T.test_conv_d_transpose_k5s2, (1, 128, 32, 32, True)),
         ('w_T', T.test_conv_w_transpose, (1, 128, 32, 32)),
         ('f_fwd', T.test_conv_f_k5s2, (1, 128, 32, 64, True)),
         ('w_conv', T.test_conv_w_conv2d, (1, 128, 32, 64))]
for name, fn, args in cases:
    try:
        fn(*args); print(name, args, 'OK')
    except AssertionError as e:
        print(name, args, 'FAIL', str(e)[:200])
