"""Host-side handle around the f-AnoGAN C-ABI (include/uad_hip.h, uad_gan_*): what a tf.Session + the fanogan graph hold in
the reference (trainers/fAnoGAN.py:18-43) -- the three variable groups, their Adam slots and the compiled phases.  PyTorch is
plumbing only (device buffers, the current HIP stream, zero-copy views for the RCCL all-reduce)."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .engine import _DevArray, _EvalOps, _ptr

GROUPS = {'Encoder': _lib.GAN_ENCODER, 'Generator': _lib.GAN_GENERATOR, 'Discriminator': _lib.GAN_DISCRIMINATOR,
          'VAE': _lib.GAN_GROUP_VAE}       # 'VAE': the Encoder + Generator slice (AnoVAE-GAN's optim_vae), buffers / all-reduce only


class GanEngine(_EvalOps):
    def __init__(self, height=128, width=128, channels=1, inter_res=8, zdim=128, max_batch=64, scale=10.0, kappa=1.0,
                 device=None, math='bf16x3', variant='unified', dim=64, kl_weight=1.0, aae_kind='aae', rho=1.0, dim_w=1, c_lambda=1.0):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError('uad_hip needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU fallback')
        self.device = torch.device(device if device is not None else f'cuda:{torch.cuda.current_device()}')
        torch.cuda.set_device(self.device)
        self.h, self.w, self.c, self.inter, self.zdim, self.max_batch = height, width, channels, inter_res, zdim, max_batch
        variants = {'unified': _lib.GAN_UNIFIED, 'resnet': _lib.GAN_RESNET, 'anovaegan': _lib.GAN_ANOVAEGAN, 'aae': _lib.GAN_AAE}
        kinds = {'constrained_ae': 0, 'aae': 1, 'constrained_aae': 2, 'gmvae': 3, 'vae_zimmerer': 4, 'cevae_zimmerer': 5, 'gmvae_you': 6, 'caae_chen': 7}     # 'gmvae': zdim = dim_z, dim = dim_c, dim_w, c_lambda
        if variant == 'aae' and aae_kind not in kinds:
            raise ValueError(f'unknown aae_kind {aae_kind!r}')
        self.aae_kind = aae_kind if variant == 'aae' else None
        if variant not in variants:
            raise ValueError(f'unknown f-AnoGAN variant {variant!r}')
        self.variant, self.dim = variant, int(dim)
        cfg = _lib.UadGanConfig(height, width, channels, inter_res, zdim, max_batch, float(scale), float(kappa), variants[variant],
                                int(dim), float(kl_weight), kinds.get(aae_kind, 1), float(rho), int(dim_w), float(c_lambda))
        self.dim_w, self.dim_c = int(dim_w), int(dim)
        h = C.c_void_p()
        _lib.check(self.lib.uad_gan_create(C.byref(cfg), C.byref(h)))
        self.handle = h
        self.nparams = int(self.lib.uad_gan_param_count(h))
        self.spec = []
        name = C.create_string_buffer(160)
        off, rank, shape = C.c_longlong(), C.c_int(), (C.c_int * 4)()
        for i in range(self.lib.uad_gan_num_tensors(h)):
            _lib.check(self.lib.uad_gan_tensor_info(h, i, name, 160, C.byref(off), C.byref(rank), shape))
            self.spec.append((name.value.decode(), tuple(shape[:rank.value]), int(off.value)))
        dec_dense = {'constrained_ae': 'Bottleneck/dense_1/kernel', 'aae': 'Bottleneck/dense_1/kernel', 'constrained_aae': 'Decoder/dense/kernel',
                     'gmvae': 'Bottleneck/dense_4/kernel', 'vae_zimmerer': 'dense_2/kernel', 'cevae_zimmerer': 'Bottleneck/dense_2/kernel', 'caae_chen': 'Decoder/dense/kernel'}
        if variant == 'aae' and aae_kind == 'gmvae_you':        # fully convolutional: no dense decoder input
            self.flat = None
        else:
            self.flat = [s for n, s, _ in self.spec if n == (dec_dense[aae_kind] if variant == 'aae' else 'Generator/dense/kernel')][0][1]
        self._views = {}
        self.set_math(math)

    def close(self):
        if getattr(self, 'handle', None):
            torch.cuda.synchronize(self.device)
            self.lib.uad_gan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_math(self, math):
        # 'bf16x3_all' (ResNet graph): also the generic k3 / k1 contractions in bf16x3 -- faster, but NOT parity-rated (header)
        modes = {'f32': _lib.MATH_F32, 'bf16x3': _lib.MATH_BF16X3, 'bf16x3_all': _lib.MATH_BF16X3_ALL}
        if math not in modes:
            raise ValueError(f'unknown math mode {math!r}')
        _lib.check(self.lib.uad_gan_set_math_mode(self.handle, modes[math]))
        self.math = math

    # ---------------------------------------------------------------- buffers
    def buffer(self, which=_lib.BUF_PARAMS):
        # Every access of the parameter buffer goes through the library: uad_gan_buffer(PARAMS) waits for a weight repack still running on the
        # handle's side stream and marks the packed copies stale, so a write through the (cached) view -- a second DP broadcast, a checkpoint
        # restore -- can neither race with the repack nor leave the next forward on old packed kernels.
        if which not in self._views or which == _lib.BUF_PARAMS:
            ptr = self.lib.uad_gan_buffer(self.handle, which)
            if which not in self._views:
                self._views[which] = torch.as_tensor(_DevArray(ptr, self.nparams), device=self.device)
        return self._views[which]

    def group(self, group):
        off, cnt = C.c_longlong(), C.c_longlong()
        gid = _lib.GAN_GENERATOR if (self.variant == 'aae' and group == 'AE') else GROUPS.get(group, group)
        _lib.check(self.lib.uad_gan_group(self.handle, gid, C.byref(off), C.byref(cnt)))
        return int(off.value), int(cnt.value)

    def set_params(self, params):
        if isinstance(params, dict):
            flat = np.concatenate([np.asarray(params[n], np.float32).reshape(-1) for n, _, _ in self.spec])
        else:
            flat = np.ascontiguousarray(params, np.float32).reshape(-1)
        _lib.check(self.lib.uad_gan_set_buffer(self.handle, _lib.BUF_PARAMS, flat.ctypes.data_as(C.c_void_p), flat.size))

    def get_buffer_host(self, which=_lib.BUF_PARAMS):
        torch.cuda.synchronize(self.device)
        out = np.empty(self.nparams, np.float32)
        _lib.check(self.lib.uad_gan_get_buffer(self.handle, which, out.ctypes.data_as(C.c_void_p), out.size))
        return out

    def set_buffer_host(self, which, flat):
        flat = np.ascontiguousarray(flat, np.float32).reshape(-1)
        _lib.check(self.lib.uad_gan_set_buffer(self.handle, which, flat.ctypes.data_as(C.c_void_p), flat.size))

    def unflatten(self, flat):
        return {n: flat[o:o + int(np.prod(s))].reshape(s) for n, s, o in self.spec}

    def get_params(self):
        return self.unflatten(self.get_buffer_host(_lib.BUF_PARAMS))

    def get_grads(self):
        return self.unflatten(self.get_buffer_host(_lib.BUF_GRADS))

    def reset_optimizer(self):
        zeros = np.zeros(self.nparams, np.float32)
        self.set_buffer_host(_lib.BUF_ADAM_M, zeros)
        self.set_buffer_host(_lib.BUF_ADAM_V, zeros)
        if self.variant in ('anovaegan', 'aae') and self.aae_kind not in ('vae_zimmerer', 'cevae_zimmerer', 'gmvae_you'):      # the Zimmerer VAE has one optimizer, one pair of slots
            self.set_buffer_host(_lib.BUF_ADAM_M2, zeros)
            self.set_buffer_host(_lib.BUF_ADAM_V2, zeros)
        for g in ('Encoder', 'Generator', 'Discriminator'):
            self.set_step_count(g, 0)

    def _gid(self, group):
        return _lib.GAN_GENERATOR if (self.variant == 'aae' and group == 'AE') else GROUPS.get(group, group)

    def step_count(self, group):
        return int(self.lib.uad_gan_get_step(self.handle, self._gid(group)))

    def set_step_count(self, group, t):
        _lib.check(self.lib.uad_gan_set_step(self.handle, self._gid(group), int(t)))

    def debug_buffer(self, name):
        """tests: torch view of a named intermediate of the last phase."""
        ptr, cnt = C.c_void_p(), C.c_longlong()
        _lib.check(self.lib.uad_gan_debug_buffer(self.handle, name.encode(), C.byref(ptr), C.byref(cnt)))
        return torch.as_tensor(_DevArray(ptr.value, cnt.value), device=self.device)

    def _new(self, shape, zero=False):
        shape = (shape,) if isinstance(shape, int) else tuple(shape)
        return (torch.zeros if zero else torch.empty)(shape, device=self.device, dtype=torch.float32)

    def _dev(self, a, shape=None):
        if a is None:
            return None
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, np.float32))
        t = t.to(self.device, torch.float32).contiguous()
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise ValueError(f'expected shape {tuple(shape)}, got {tuple(t.shape)}')
        return t

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---------------------------------------------------------------- phases
    def aae_phase(self, which, x, z=None, eps=None, mask_z=None, mask_dec=None, mask_rec=None, want_backward=True, want_images=True,
                  want_l1=False):
        """AAE family (variant 'aae').  which: 'AE' (optim_ae: loss = mean(L2 [+ rho Rec_z]) over every autoencoder variable) |
        'Discriminator' (optim_dis; z = prior sample [n,zDim], eps [n] of z_hat = z + eps (z - z_)) | 'Encoder' (optim_gen: -mean d_).
        Returns device tensors: AE -> loss, L2, Rec_z, reconstructionLoss, reconstruction, z, L1; critic -> disc_loss, disc_fake,
        disc_real, penalty; generator -> gen_loss."""
        if self.variant != 'aae':
            raise ValueError('aae_phase needs an AAE-family engine')
        g = {'AE': _lib.GAN_GENERATOR, 'Discriminator': _lib.GAN_DISCRIMINATOR, 'Encoder': _lib.GAN_ENCODER}[which]
        n = x.shape[0]
        img = (n, self.h, self.w, self.c)
        x = self._dev(x, img)
        z = self._dev(z, (n, self.zdim))
        eps = self._dev(None if eps is None else (eps.reshape(-1) if isinstance(eps, torch.Tensor) else np.asarray(eps, np.float32).reshape(-1)), (n,))
        mask_z, mask_rec = self._dev(mask_z, (n, self.zdim)), self._dev(mask_rec, (n, self.zdim))
        mask_dec = self._dev(mask_dec, (n, self.flat))
        out, scal = {}, self._new(16, zero=True)
        io = _lib.UadGanIO()
        io.x, io.z, io.alpha, io.mask_z, io.mask_g, io.mask_sigma, io.scalars = _ptr(x), _ptr(z), _ptr(eps), _ptr(mask_z), _ptr(mask_dec), _ptr(mask_rec), _ptr(scal)
        if g == _lib.GAN_GENERATOR:
            if want_images:
                out['reconstruction'] = self._new(img); io.reconstruction = _ptr(out['reconstruction'])
            out['z'] = self._new((n, self.zdim)); io.z_enc = _ptr(out['z'])
            if want_l1:
                out['L1'] = self._new(img); io.l1_map = _ptr(out['L1'])
        _lib.check(self.lib.uad_gan_phase(self.handle, g, C.byref(io), n, 1 if want_backward else 0, self._stream()))
        S = _lib.GAN_SCALARS
        if g == _lib.GAN_GENERATOR:
            out.update(L2=scal[S.index('loss_img')], Rec_z=scal[S.index('loss_fts')], loss=scal[S.index('enc_loss')],
                       reconstructionLoss=scal[S.index('reconstructionLoss')])
        elif g == _lib.GAN_DISCRIMINATOR:
            out.update({k: scal[S.index(k)] for k in ('disc_fake', 'disc_real', 'penalty', 'disc_loss')})
        else:
            out['gen_loss'] = scal[S.index('gen_loss')]
        self._keep = (x, z, eps, mask_z, mask_dec, mask_rec, scal, out)
        return out

    # ---------------------------------------------------------------- Zimmerer VAE (variant 'aae', aae_kind 'vae_zimmerer')
    def zim_phase(self, x, eps=None, want_backward=True, want_l1=True, x_ce=None, want_anomaly=True):
        """One sess.run of trainers/VAE.py:83-96 on models/variational_autoencoder_Zimmerer.py: forward, reconstructionLoss / kl / loss and
        (want_backward) the gradient of `loss` w.r.t. every variable.  On a 'cevae_zimmerer' engine: one sess.run of trainers/ceVAE.py:95-110
        (x_ce: the context-masked batch, None = x; want_backward True | False | 'data' = the input-gradient chain only) returning in addition
        reconstruction_ce, L1_ce, anomaly, Rec_vae, Rec_ce, loss_vae."""
        if self.aae_kind not in ('vae_zimmerer', 'cevae_zimmerer'):
            raise ValueError('zim_phase needs a Zimmerer-stack engine')
        ce = self.aae_kind == 'cevae_zimmerer'
        if x_ce is not None and not ce:
            raise ValueError('x_ce is an input of the context-encoding VAE')
        n = x.shape[0]
        img = (n, self.h, self.w, self.c)
        x = self._dev(x, img)
        eps = self._dev(eps, (n, self.zdim))
        x_ce = self._dev(x_ce, img)
        scal = self._new(16, zero=True)
        out = {'reconstruction': self._new(img), 'z': self._new((n, self.zdim))}
        io = _lib.UadGanIO()
        io.x, io.eps, io.scalars, io.reconstruction, io.z_enc = _ptr(x), _ptr(eps), _ptr(scal), _ptr(out['reconstruction']), _ptr(out['z'])
        io.x_ce = _ptr(x_ce)
        if want_l1:
            out['L1'] = self._new(img); io.l1_map = _ptr(out['L1'])
        if ce:
            out['reconstruction_ce'] = self._new(img); io.generated = _ptr(out['reconstruction_ce'])
            if want_l1:
                out['L1_ce'] = self._new(img); io.l1_map_ce = _ptr(out['L1_ce'])
            if want_backward and want_anomaly:
                out['anomaly'] = self._new(img); io.anomaly = _ptr(out['anomaly'])
        wb = 2 if want_backward == 'data' else (1 if want_backward else 0)
        if wb == 2 and not ce:
            raise ValueError("want_backward='data' is the ceVAE's anomaly-map mode")
        _lib.check(self.lib.uad_gan_phase(self.handle, _lib.GAN_GENERATOR, C.byref(io), n, wb, self._stream()))
        S = _lib.GAN_SCALARS
        out.update(reconstructionLoss=scal[S.index('reconstructionLoss')], kl=scal[S.index('kl')], loss=scal[S.index('enc_loss')])
        if ce:
            out.update(Rec_vae=scal[S.index('loss_img')], Rec_ce=scal[S.index('loss_fts')], loss_vae=scal[S.index('gm_loss')])
        self._keep = (x, eps, x_ce, scal, out)
        return out

    # ---------------------------------------------------------------- dense GMVAE (variant 'aae', aae_kind 'gmvae')
    def _gm_io(self, n, eps_w, eps_z, masks):
        masks = masks or {}
        if self.aae_kind == 'gmvae_you':        # spatial latents on the H/4 map; the graph has no dropout layer
            if any(v is not None for v in masks.values()):
                raise ValueError('models/gaussian_mixture_variational_autoencoder_You.py has no dropout layers: masks are not accepted')
            r = self.inter
            t = dict(eps_w=self._dev(eps_w, (n, r, r, self.dim_w)), eps=self._dev(eps_z, (n, r, r, self.zdim)))
            io = _lib.UadGanIO()
            for k, v in t.items():
                setattr(io, k, _ptr(v))
            return io, t
        t = dict(eps_w=self._dev(eps_w, (n, self.dim_w)), eps=self._dev(eps_z, (n, self.zdim)),
                 mask_w_mu=self._dev(masks.get('w_mu'), (n, self.dim_w)), mask_w_ls=self._dev(masks.get('w_ls'), (n, self.dim_w)),
                 mask_z=self._dev(masks.get('z_mu'), (n, self.zdim)), mask_g=self._dev(masks.get('dec'), (n, self.flat)))
        io = _lib.UadGanIO()
        for k, v in t.items():
            setattr(io, k, _ptr(v))
        return io, t

    def gm_phase(self, x, eps_w=None, eps_z=None, masks=None, want_backward=True, want_l1=True):
        """One sess.run of trainers/GMVAE.py:122-139: forward + the four loss terms (+ the gradient of `loss` w.r.t. every variable).
        masks: dict with optional 'w_mu', 'w_ls' [n,dim_w], 'z_mu' [n,dim_z], 'dec' [n,flat] keep masks (already / (1 - rate))."""
        if self.aae_kind not in ('gmvae', 'gmvae_you'):
            raise ValueError('gm_phase needs a GMVAE engine (aae_kind gmvae | gmvae_you)')
        n = x.shape[0]
        img = (n, self.h, self.w, self.c)
        x = self._dev(x, img)
        io, keep = self._gm_io(n, eps_w, eps_z, masks)
        scal = self._new(16, zero=True)
        zshape = (n, self.inter, self.inter, self.zdim) if self.aae_kind == 'gmvae_you' else (n, self.zdim)
        out = {'reconstruction': self._new(img), 'z_sampled': self._new(zshape)}
        io.x, io.scalars, io.reconstruction, io.z_enc = _ptr(x), _ptr(scal), _ptr(out['reconstruction']), _ptr(out['z_sampled'])
        if want_l1:
            out['L1'] = self._new(img); io.l1_map = _ptr(out['L1'])
        _lib.check(self.lib.uad_gan_phase(self.handle, _lib.GAN_GENERATOR, C.byref(io), n, 1 if want_backward else 0, self._stream()))
        S = _lib.GAN_SCALARS
        out.update(loss=scal[S.index('gm_loss')], mean_p_loss=scal[S.index('reconstructionLoss')], reconstructionLoss=scal[S.index('reconstructionLoss')],
                   conditional_prior_loss=scal[S.index('gm_con')], w_prior_loss=scal[S.index('gm_w')], c_prior_loss=scal[S.index('gm_c')])
        self._keep = (x, keep, scal, out)
        return out

    def gm_restore_step(self, x_restored, eps_w=None, eps_z=None, masks=None, tv_lambda=1.8, restore_lr=1e-3, want_grads=False):
        """trainers/GMVAE.py:172-184 on device: x_restored (a DEVICE tensor, updated in place) -= restore_lr * d(n loss + sum tv TV_n)/dx."""
        if self.aae_kind not in ('gmvae', 'gmvae_you'):
            raise ValueError('gm_restore_step needs a GMVAE engine (aae_kind gmvae | gmvae_you)')
        if not isinstance(x_restored, torch.Tensor) or x_restored.device != self.device or x_restored.dtype != torch.float32 or not x_restored.is_contiguous():
            raise ValueError('x_restored must be a contiguous fp32 tensor on the engine device (it is updated in place)')
        n = x_restored.shape[0]
        if tuple(x_restored.shape) != (n, self.h, self.w, self.c):
            raise ValueError('x_restored must be [n,H,W,C]')
        io, keep = self._gm_io(n, eps_w, eps_z, masks)
        grads = self._new(tuple(x_restored.shape)) if want_grads else None
        _lib.check(self.lib.uad_gan_restore_step(self.handle, _ptr(x_restored), C.byref(io), n, float(tv_lambda), float(restore_lr),
                                                 _ptr(grads), self._stream()))
        self._keep = (keep, grads)
        return grads

    def phase(self, group, x=None, z=None, alpha=None, mask_z=None, mask_g=None, want_backward=True, want_images=True,
              want_l1=False, eps=None, mask_sigma=None):
        """Run one phase ('Generator' | 'Discriminator' | 'Encoder').  Returns a dict of device tensors: the 0-d losses of the
        phase (trainers/fAnoGAN.py:50-66 names) and, per phase, 'generated' | 'reconstruction', 'z_enc', 'L1'."""
        g = GROUPS[group]
        n = (x if x is not None else z).shape[0]
        img = (n, self.h, self.w, self.c)
        x = self._dev(x, img)
        z = self._dev(z, (n, self.zdim))
        alpha = self._dev(None if alpha is None else np.asarray(alpha, np.float32).reshape(-1) if not isinstance(alpha, torch.Tensor) else alpha.reshape(-1), (n,))
        if self.variant == 'resnet' and (mask_z is not None or mask_g is not None):
            raise ValueError('the ResNet f-AnoGAN graph has no dropout layers (models/fanogan_schlegl.py): masks are not accepted')
        mask_z = self._dev(mask_z, (n, self.zdim))
        mask_g = self._dev(mask_g, (n, self.flat))
        av = self.variant == 'anovaegan'        # AnoVAE-GAN: 'Encoder' = the VAE phase; eps / mask_z (mu head) / mask_sigma; no z, no mask_g
        if not av and (eps is not None or mask_sigma is not None):
            raise ValueError('eps / mask_sigma are AnoVAE-GAN inputs')
        eps = self._dev(eps, (n, self.zdim))
        mask_sigma = self._dev(mask_sigma, (n, self.zdim))
        out = {}
        scal = self._new(16, zero=True)
        io = _lib.UadGanIO()
        io.x, io.z, io.alpha, io.mask_z, io.mask_g = _ptr(x), _ptr(z), _ptr(alpha), _ptr(mask_z), _ptr(mask_g)
        io.scalars = _ptr(scal)
        io.eps, io.mask_sigma = _ptr(eps), _ptr(mask_sigma)
        if av and g != _lib.GAN_ENCODER and want_images:
            out['reconstruction'] = self._new(img)     # the critic's "fake" is the reconstruction
            io.generated = _ptr(out['reconstruction'])
            want_images = False
        if g == _lib.GAN_ENCODER:
            if want_images:
                out['reconstruction'] = self._new(img)
                io.reconstruction = _ptr(out['reconstruction'])
            out['z_enc'] = self._new((n, self.zdim))
            io.z_enc = _ptr(out['z_enc'])
            if want_l1:
                out['L1'] = self._new(img)
                io.l1_map = _ptr(out['L1'])
        elif want_images:
            out['generated'] = self._new(img)
            io.generated = _ptr(out['generated'])
        _lib.check(self.lib.uad_gan_phase(self.handle, g, C.byref(io), n, 1 if want_backward else 0, self._stream()))
        names = {_lib.GAN_GENERATOR: ('gen_loss', 'disc_fake'),
                 _lib.GAN_DISCRIMINATOR: ('gen_loss', 'disc_fake', 'disc_real', 'penalty', 'disc_loss'),
                 _lib.GAN_ENCODER: ('reconstructionLoss', 'kl', 'enc_loss') if av else ('loss_img', 'loss_fts', 'enc_loss', 'reconstructionLoss')}[g]
        for k in names:
            out[k] = scal[_lib.GAN_SCALARS.index(k)]
        if g == _lib.GAN_ENCODER:
            out['loss'] = out['reconstructionLoss']
        self._keep = (x, z, alpha, mask_z, mask_g, scal, out, eps, mask_sigma)
        return out

    def allreduce_attach(self, comm, world):
        """uad_gan_allreduce_attach: comm = a parallel.RcclComm (None detaches).  From then on phase(..., want_backward=True) all-reduces the trained group's
        gradient slice itself, in buckets issued while its backward still runs; adam(..., grad_scale=1 / world) follows on the same stream.
        Raises for the AAE-family graphs (their caller all-reduces the slice)."""
        import ctypes as C
        _lib.check(self.lib.uad_gan_allreduce_attach(self.handle, C.c_void_p(comm.handle) if comm is not None else None, int(world)))
        self._ar_comm = comm

    def adam(self, group, lr, beta1=0.5, beta2=0.9, eps=1e-8, grad_scale=1.0):
        """group: 'Encoder' | 'Generator' | 'Discriminator'; for the AAE family 'AE' (= optim_ae), 'Discriminator', 'Encoder' (= optim_gen)."""
        gid = _lib.GAN_GENERATOR if (self.variant == 'aae' and group == 'AE') else GROUPS[group]
        _lib.check(self.lib.uad_gan_adam(self.handle, gid, lr, beta1, beta2, eps, grad_scale, self._stream()))

    def reconstruct(self, x, mask_z=None, mask_g=None, want_l1=False, eps=None, mask_sigma=None):
        n = x.shape[0]
        img = (n, self.h, self.w, self.c)
        x = self._dev(x, img)
        mask_z = self._dev(mask_z, (n, self.zdim))
        mask_g = self._dev(mask_g, (n, self.flat))
        out = {'reconstruction': self._new(img), 'z_enc': self._new((n, self.zdim))}
        io = _lib.UadGanIO()
        io.x, io.mask_z, io.mask_g = _ptr(x), _ptr(mask_z), _ptr(mask_g)
        eps, mask_sigma = self._dev(eps, (n, self.zdim)), self._dev(mask_sigma, (n, self.zdim))
        io.eps, io.mask_sigma = _ptr(eps), _ptr(mask_sigma)
        io.reconstruction, io.z_enc = _ptr(out['reconstruction']), _ptr(out['z_enc'])
        if want_l1:
            out['L1'] = self._new(img)
            io.l1_map = _ptr(out['L1'])
        _lib.check(self.lib.uad_gan_reconstruct(self.handle, C.byref(io), n, self._stream()))
        self._keep = (x, mask_z, mask_g, out, eps, mask_sigma)
        return out


class ZimmererEngine(GanEngine):
    """The Zimmerer VAE / ceVAE behind the `Engine` surface the AE-family trainers and `parallel.DataParallelStep` drive (forward / backward
    / adam_step / grad_segment): forward(want_backward=...) already produces every gradient (and the ceVAE's anomaly map), so backward() has
    nothing left to do and the whole flat buffer is reported as ONE segment (the last one in `parallel.SEGMENT_ORDER`), all-reduced once."""

    def __init__(self, height, width, channels, inter_res, zdim, max_batch=64, device=None, math='bf16x3_all', cevae=False):
        # bf16x3_all: the generic k4 contractions in split-bf16; parity-rated for THIS stack (tests/test_gpu_zimmerer.py runs the 1e-4 gradient
        # checks in both modes) -- the mode's 3e-4 drift is specific to the 20-layer ResNet critic's penalty scalar
        super().__init__(height, width, channels, inter_res, zdim, max_batch=max_batch, device=device, math=math, variant='aae',
                         aae_kind='cevae_zimmerer' if cevae else 'vae_zimmerer')
        self.arch = 'ceVAE_Zimmerer' if cevae else 'VAE_Zimmerer'
        self.cevae = bool(cevae)

    def forward(self, x, eps=None, masks=None, want_backward=False, want_l1=True, want_latents=True, x_ce=None, want_anomaly=True, **kw):
        if masks:
            raise ValueError('the Zimmerer models have no dropout layers: masks are not accepted')
        o = self.zim_phase(x, eps, want_backward=want_backward, want_l1=want_l1, x_ce=x_ce, want_anomaly=want_anomaly)
        zero = torch.zeros((), device=self.device)
        if self.cevae:      # scalars in the fused ceVAE handle's layout: reconstructionLoss, kl, loss, -, Rec_vae, Rec_ce, loss_vae, -
            out = {'x_hat': o['reconstruction'], 'x_hat_ce': o['reconstruction_ce'], 'z': o['z'],
                   'scalars': torch.stack([o['reconstructionLoss'], o['kl'], o['loss'], zero, o['Rec_vae'], o['Rec_ce'], o['loss_vae'], zero])}
            if want_l1:
                out['L1_vae'], out['L1_ce'] = o['L1'], o['L1_ce']
            if 'anomaly' in o:
                out['anomaly'] = o['anomaly']
            return out
        out = {'x_hat': o['reconstruction'], 'z': o['z'], 'scalars': torch.stack([o['reconstructionLoss'], o['kl'], o['loss']])}
        if want_l1:
            out['L1'] = o['L1']
        return out

    def backward(self, segment=_lib.SEG_ALL):
        return None

    def grad_segment(self, seg):
        return (0, self.nparams) if seg in (_lib.SEG_ENCODER, _lib.SEG_ENCODER_LO) else (0, 0)

    def adam_step(self, lr, beta1=0.5, beta2=0.999, eps=1e-8, grad_scale=1.0):
        self.adam('AE', lr, beta1, beta2, eps, grad_scale)

    def train_step(self, x, eps=None, masks=None, lr=1e-4, beta1=0.5, beta2=0.999, adam_eps=1e-8, **kw):
        out = self.forward(x, eps, masks, want_backward=True, **kw)
        self.adam_step(lr, beta1, beta2, adam_eps)
        return out

    @property
    def step_count(self):
        return int(self.lib.uad_gan_get_step(self.handle, _lib.GAN_GENERATOR))

    @step_count.setter
    def step_count(self, t):
        _lib.check(self.lib.uad_gan_set_step(self.handle, _lib.GAN_GENERATOR, int(t)))
