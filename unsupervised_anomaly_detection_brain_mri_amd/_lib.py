"""ctypes binding of libuad_hip.so (C-ABI: include/uad_hip.h).  Fails loudly when the library is missing —
there is no CPU / eager fallback anywhere in the product path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('UAD_LIB') or os.path.join(_HERE, 'libuad_hip.so')   # UAD_LIB: A/B builds for kernel tuning

UAD_OK = 0
ARCH_AE, ARCH_VAE, ARCH_CEVAE, ARCH_GMVAE_SPATIAL, ARCH_AE_SPATIAL = 0, 1, 2, 3, 4
BUF_PARAMS, BUF_GRADS, BUF_ADAM_M, BUF_ADAM_V = 0, 1, 2, 3
SEG_DECODER, SEG_BOTTLENECK, SEG_ENCODER, SEG_ENCODER_HI, SEG_ENCODER_LO, SEG_ALL = 0, 1, 2, 3, 4, -1
MATH_F32, MATH_BF16X3, MATH_BF16X3_ALL, MATH_BF16X6 = 0, 1, 2, 3      # (BF16X3_ALL: uad_gan_* handles only; BF16X6: uad_create handles only)

c_float_p = C.c_void_p  # device pointers are passed as integers


class UadConfig(C.Structure):
    _fields_ = [('arch', C.c_int), ('height', C.c_int), ('width', C.c_int), ('channels', C.c_int),
                ('inter_res', C.c_int), ('zdim', C.c_int), ('max_batch', C.c_int),
                ('dim_c', C.c_int), ('dim_z', C.c_int), ('dim_w', C.c_int), ('c_lambda', C.c_float)]


class UadRngJob(C.Structure):
    _fields_ = [('out', C.c_void_p), ('per_sample', C.c_int), ('kind', C.c_int), ('rate', C.c_float), ('stream', C.c_int)]


RNG_NORMAL, RNG_KEEP_MASK = 0, 1


class UadIO(C.Structure):
    _fields_ = [('x', C.c_void_p), ('eps', C.c_void_p), ('mask_mu', C.c_void_p), ('mask_sigma', C.c_void_p),
                ('mask_dec', C.c_void_p), ('x_hat', C.c_void_p), ('l1_map', C.c_void_p), ('z_mu', C.c_void_p),
                ('z_log_sigma', C.c_void_p), ('z_sigma', C.c_void_p), ('scalars', C.c_void_p),
                ('rec_per_sample', C.c_void_p),
                # ceVAE only
                ('x_ce', C.c_void_p), ('mask_mu_ce', C.c_void_p), ('mask_dec_ce', C.c_void_p),
                ('x_hat_ce', C.c_void_p), ('l1_map_ce', C.c_void_p), ('anomaly', C.c_void_p),
                # spatial GMVAE only
                ('eps_w', C.c_void_p), ('eps_z', C.c_void_p), ('w_mu', C.c_void_p), ('w_log_sigma', C.c_void_p),
                ('pc', C.c_void_p)]


class UadGanConfig(C.Structure):
    _fields_ = [('height', C.c_int), ('width', C.c_int), ('channels', C.c_int), ('inter_res', C.c_int), ('zdim', C.c_int),
                ('max_batch', C.c_int), ('scale', C.c_float), ('kappa', C.c_float), ('variant', C.c_int), ('dim', C.c_int),
                ('kl_weight', C.c_float), ('aae_kind', C.c_int), ('rho', C.c_float), ('dim_w', C.c_int), ('c_lambda', C.c_float)]


class UadGanIO(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ('x', 'z', 'alpha', 'mask_z', 'mask_g', 'generated', 'reconstruction', 'z_enc',
                                          'l1_map', 'scalars', 'eps', 'mask_sigma', 'eps_w', 'mask_w_mu', 'mask_w_ls', 'x_ce', 'l1_map_ce', 'anomaly')]


GAN_ENCODER, GAN_GENERATOR, GAN_DISCRIMINATOR = 0, 1, 2
GAN_UNIFIED, GAN_RESNET, GAN_ANOVAEGAN, GAN_AAE = 0, 1, 2, 3
GAN_GROUP_VAE = 3
BUF_ADAM_M2, BUF_ADAM_V2 = 4, 5
GAN_SCALARS = ('gen_loss', 'disc_fake', 'disc_real', 'penalty', 'disc_loss', 'loss_img', 'loss_fts', 'enc_loss',
               'reconstructionLoss', 'kl', 'gm_loss', 'gm_con', 'gm_w', 'gm_c')


class UadConvDesc(C.Structure):
    _fields_ = [(k, C.c_int) for k in ('N', 'HB', 'WB', 'CB', 'HS', 'WS', 'CS', 'KS', 'S', 'P')]


class UadXform(C.Structure):
    _fields_ = [('scale', C.c_void_p), ('shift', C.c_void_p), ('alpha', C.c_float)]


# every symbol include/uad_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    'uad_last_error': (C.c_char_p, []),
    'uad_version': (C.c_char_p, []),
    'uad_create': (C.c_int, [C.POINTER(UadConfig), C.POINTER(C.c_void_p)]),
    'uad_destroy': (C.c_int, [C.c_void_p]),
    'uad_param_count': (C.c_longlong, [C.c_void_p]),
    'uad_num_tensors': (C.c_int, [C.c_void_p]),
    'uad_tensor_info': (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_longlong),
                                  C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'uad_buffer': (C.c_void_p, [C.c_void_p, C.c_int]),
    'uad_grad_segment': (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    'uad_set_params': (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong]),
    'uad_get_params': (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong]),
    'uad_get_buffer': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong]),
    'uad_set_buffer': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong]),
    'uad_reset_optimizer': (C.c_int, [C.c_void_p]),
    'uad_get_step': (C.c_longlong, [C.c_void_p]),
    'uad_set_step': (C.c_int, [C.c_void_p, C.c_longlong]),
    'uad_forward': (C.c_int, [C.c_void_p, C.POINTER(UadIO), C.c_int, C.c_int, C.c_void_p]),
    'uad_backward': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    'uad_backward_deferred': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    'uad_check_fault': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    'uad_rccl_unique_id': (C.c_int, [C.c_void_p, C.c_int]),
    'uad_rccl_comm_create': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    'uad_rccl_comm_destroy': (C.c_int, [C.c_void_p]),
    'uad_rccl_allreduce': (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]),
    'uad_allreduce_attach': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    'uad_backward_allreduce': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    'uad_set_fault_deferred': (C.c_int, [C.c_void_p, C.c_int]),
    'uad_adam_step': (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    'uad_optimizer_step': (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    'uad_train_step': (C.c_int, [C.c_void_p, C.POINTER(UadIO), C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                 C.c_void_p]),
    'uad_restore_step': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float,
                                   C.c_void_p, C.c_void_p]),
    'uad_set_math_mode': (C.c_int, [C.c_void_p, C.c_int]),
    'uad_get_math_mode': (C.c_int, [C.c_void_p]),
    'uad_profile_enable': (C.c_int, [C.c_void_p, C.c_int]),
    'uad_profile_report': (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    'uad_residual': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p,
                               C.c_void_p, C.c_void_p]),
    'uad_gather_slices': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_void_p, C.c_void_p]),
    'uad_gather_mask': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p]),
    'uad_erode_cross': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'uad_median3d': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'uad_mc_stats': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p]),
    'uad_cc_filter': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'uad_scores_create': (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.POINTER(C.c_void_p), C.c_void_p]),
    'uad_scores_auc': (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    'uad_scores_dice': (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_void_p]),
    'uad_scores_destroy': (C.c_int, [C.c_void_p]),
    'uad_rng_fill': (C.c_int, [C.POINTER(UadRngJob), C.c_int, C.c_int, C.c_ulonglong, C.c_ulonglong, C.c_longlong, C.c_void_p]),
    'uad_clock_probe': (C.c_int, [C.c_void_p, C.c_ulonglong, C.c_void_p]),
    'uad_gan_create': (C.c_int, [C.POINTER(UadGanConfig), C.POINTER(C.c_void_p)]),
    'uad_gan_destroy': (C.c_int, [C.c_void_p]),
    'uad_gan_param_count': (C.c_longlong, [C.c_void_p]),
    'uad_gan_num_tensors': (C.c_int, [C.c_void_p]),
    'uad_gan_tensor_info': (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_longlong),
                                      C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'uad_gan_buffer': (C.c_void_p, [C.c_void_p, C.c_int]),
    'uad_gan_group': (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    'uad_gan_set_buffer': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong]),
    'uad_gan_get_buffer': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong]),
    'uad_gan_set_math_mode': (C.c_int, [C.c_void_p, C.c_int]),
    'uad_gan_get_step': (C.c_longlong, [C.c_void_p, C.c_int]),
    'uad_gan_set_step': (C.c_int, [C.c_void_p, C.c_int, C.c_longlong]),
    'uad_gan_phase': (C.c_int, [C.c_void_p, C.c_int, C.POINTER(UadGanIO), C.c_int, C.c_int, C.c_void_p]),
    'uad_gan_adam': (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    'uad_gan_allreduce_attach': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    'uad_gan_reconstruct': (C.c_int, [C.c_void_p, C.POINTER(UadGanIO), C.c_int, C.c_void_p]),
    'uad_gan_restore_step': (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(UadGanIO), C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    'uad_debug_buffer': (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_longlong)]),
    'uad_k3_profile_enable': (C.c_int, [C.c_int]),
    'uad_k3_profile_read': (C.c_int, [C.c_char_p, C.c_int]),
    'uad_gan_debug_buffer': (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_longlong)]),
    'uad_op_conv_f': (C.c_int, [C.POINTER(UadConvDesc), C.c_void_p, C.POINTER(UadXform), C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'uad_op_conv_d': (C.c_int, [C.POINTER(UadConvDesc), C.c_void_p, C.POINTER(UadXform), C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'uad_op_conv_f_bwdact': (C.c_int, [C.POINTER(UadConvDesc), C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.POINTER(UadXform), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'uad_op_conv_d_bwdact': (C.c_int, [C.POINTER(UadConvDesc), C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.POINTER(UadXform), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'uad_op_conv_w': (C.c_int, [C.POINTER(UadConvDesc), C.c_void_p, C.POINTER(UadXform), C.c_void_p,
                                C.POINTER(UadXform), C.c_void_p, C.c_void_p]),
    'uad_op_conv_first_fwd': (C.c_int, [C.POINTER(UadConvDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    'uad_op_conv_first_wgrad': (C.c_int, [C.POINTER(UadConvDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'uad_op_adam': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_float, C.c_float,
                              C.c_float, C.c_float, C.c_float, C.c_void_p]),
}

_lib = None


def load():
    """Load libuad_hip.so and bind every declared symbol.  Raises RuntimeError if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'{LIB_PATH} is missing: the HIP extension has not been built '
            f'(run `python -m unsupervised_anomaly_detection_brain_mri_amd.build` or __graft_entry__.build()). '
            f'There is no CPU fallback.')
    # PyTorch-ROCm first: libuad_hip.so's HIP runtime dependency must resolve to the copy torch ships and initialises -- loaded on its own before torch, the
    # library binds the system runtime instead and the process then holds two, of which the library's reports "no ROCm-capable device" at the first hipMalloc
    # (seen with `python __graft_entry__.py smoke`, whose build() loads the library before smoke() imports torch)
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != UAD_OK:
        msg = load().uad_last_error().decode('utf-8', 'replace')
        if rc in (1, 3):
            raise ValueError(f'uad_hip: {msg}')
        raise RuntimeError(f'uad_hip (code {rc}): {msg}')
