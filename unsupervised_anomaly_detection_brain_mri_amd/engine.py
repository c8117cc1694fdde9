"""Host-side handle around the C-ABI model (include/uad_hip.h).  PyTorch is plumbing only: device buffers for the
caller-owned inputs/outputs, the current HIP stream, and zero-copy tensor views of the handle-owned flat
parameter / gradient buffers for torch.distributed (RCCL) all-reduce."""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib


class _DevArray:
    """__cuda_array_interface__ shim: lets torch alias a raw device pointer without copying."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {'shape': (int(count),), 'typestr': '<f4', 'data': (int(ptr), False),
                                         'version': 2, 'strides': None}


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def rng_fill(jobs, n, seed, step, sample0=0, device=None, stream=None):
    """One launch of the device noise generator (include/uad_hip.h: uad_rng_fill).  jobs: list of (name, per_sample shape tuple or int,
    kind 'normal' | 'keep', rate); returns {name: device tensor [n, *shape]}.  Sample i gets the numbers of GLOBAL sample sample0 + i
    at this step: the same whichever rank draws them."""
    lib = _lib.load()
    dev = torch.device(device if device is not None else f'cuda:{torch.cuda.current_device()}')
    arr = (_lib.UadRngJob * len(jobs))()
    out = {}
    for k, (name, shape, kind, rate) in enumerate(jobs):
        shape = (shape,) if np.isscalar(shape) else tuple(shape)
        t = torch.empty((n,) + shape, device=dev, dtype=torch.float32)
        out[name] = t
        arr[k] = _lib.UadRngJob(t.data_ptr(), int(np.prod(shape)), _lib.RNG_NORMAL if kind == 'normal' else _lib.RNG_KEEP_MASK,
                                float(rate), k)
    st = C.c_void_p(stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(lib.uad_rng_fill(arr, len(jobs), int(n), int(seed) & 0xFFFFFFFFFFFFFFFF, int(step), int(sample0), st))
    return out


def shader_clock_under(enqueue, window_ms=30.0, device=None):
    """The shader clock [GHz] the GPU sustains while `enqueue()`'s work runs (include/uad_hip.h: uad_clock_probe): one probe wave on a stream
    of its own samples s_memtime against the 100 MHz s_memrealtime for window_ms while the caller's work -- enqueue() is called until about
    1.5 x window_ms of it is queued -- runs beside it on the current stream.  DVFS keeps a power-limited kernel mix well under the 2.4 GHz spec
    clock the MFMA peaks are priced at; bench.py reports its roofline fraction at both."""
    import time
    lib = _lib.load()
    dev = torch.device(device if device is not None else f'cuda:{torch.cuda.current_device()}')
    out = torch.zeros(2, dtype=torch.int64, device=dev)
    side = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    enqueue()
    torch.cuda.synchronize(dev)
    per_call = max(time.perf_counter() - t0, 1e-5)
    calls = int(np.ceil(1.5 * window_ms * 1e-3 / per_call)) + 2
    for _ in range(2):
        enqueue()                                   # the load is running when the probe starts
    _lib.check(lib.uad_clock_probe(C.c_void_p(out.data_ptr()), int(window_ms * 1e5), C.c_void_p(side.cuda_stream)))
    for _ in range(calls):
        enqueue()
    torch.cuda.synchronize(dev)
    cyc, ticks = (int(v) for v in out.cpu().tolist())
    return cyc / max(ticks, 1) / 10.0


class _EvalOps:
    """Model-independent device ops of the evaluation path (erosion, 3-D median, residual maps, sort-based metrics); shared by
    the AE-family Engine and the f-AnoGAN GanEngine.  Needs self.lib, self.device, self._dev, self._stream."""

    # ---------------------------------------------------------------- scoring
    def erode_cross(self, masks, iterations=12):
        """Brain-mask erosion on device (utils/Evaluation.py:84-89): masks [n,H,W] (any dtype, nonzero = set) -> fp32 0/1 tensor."""
        mk = self._dev(np.asarray(masks, np.float32) if not isinstance(masks, torch.Tensor) else masks)
        if mk.dim() != 3:
            raise ValueError(f'masks must be [n,H,W], got {tuple(mk.shape)}')
        out = torch.empty_like(mk)
        _lib.check(self.lib.uad_erode_cross(_ptr(mk), mk.shape[0], mk.shape[1], mk.shape[2], int(iterations), _ptr(out),
                                            self._stream()))
        return out

    def median3d(self, volume, ksize=5):
        """5x5x5 median filter of a [D,H,W] volume, scipy 'reflect' boundary (utils/Evaluation.py:108-110)."""
        v = self._dev(np.asarray(volume, np.float32) if not isinstance(volume, torch.Tensor) else volume)
        if v.dim() != 3:
            raise ValueError(f'volume must be [D,H,W], got {tuple(v.shape)}')
        out = torch.empty_like(v)
        _lib.check(self.lib.uad_median3d(_ptr(v), v.shape[0], v.shape[1], v.shape[2], int(ksize), _ptr(out), self._stream()))
        return out

    def mc_stats(self, recs, mask=None):
        """Monte-Carlo dropout statistics (utils/Evaluation.py:238-266): recs [K, ...] device / host array of K reconstructions, mask
        broadcastable to one sample.  Returns (mean, epistemic variance) of the masked reconstructions as device tensors."""
        r = recs if isinstance(recs, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(recs, np.float32))
        r = r.to(self.device, torch.float32).contiguous()
        K, total = r.shape[0], r[0].numel()
        m = None
        if mask is not None:
            m = mask if isinstance(mask, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(mask, np.float32))
            m = m.to(self.device, torch.float32).expand(r.shape[1:]).contiguous()
        mean, var = torch.empty_like(r[0]), torch.empty_like(r[0])
        _lib.check(self.lib.uad_mc_stats(_ptr(r), _ptr(m), K, total, _ptr(mean), _ptr(var), self._stream()))
        return mean, var

    def cc_filter(self, volume, max_voxels=7):
        """filter_3d_connected_components (utils/Evaluation.py:113-127) of a [D,H,W] volume (bool / float, non-zero = foreground):
        components of at most `max_voxels` voxels are zeroed.  Returns a float device tensor."""
        v = volume if isinstance(volume, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(volume))
        v = v.to(self.device, torch.float32).contiguous()
        if v.dim() != 3:
            raise ValueError('cc_filter expects a [D,H,W] volume')
        out = torch.empty_like(v)
        _lib.check(self.lib.uad_cc_filter(_ptr(v), v.shape[0], v.shape[1], v.shape[2], int(max_voxels), _ptr(out), self._stream()))
        return out

    def scores(self, predictions, labels):
        """One descending device sort of all voxel scores -> Scores object (AUROC, AUPRC, dice at thresholds)."""
        return Scores(self, predictions, labels)

    def residual(self, x, x_rec, mask=None, pos_only=True, prior_thresh=None):
        """Residual anomaly map on device (utils/Evaluation.py:282-289).  Returns (map, l1err_per_sample)."""
        x = self._dev(x)
        xr = self._dev(x_rec, x.shape)
        mk = self._dev(mask, x.shape) if mask is not None else None
        n = x.shape[0]
        hw = int(np.prod(x.shape[1:]))
        out = torch.empty_like(x)
        l1 = torch.empty(n, device=self.device)
        thr = -math.inf if prior_thresh is None else float(prior_thresh)
        _lib.check(self.lib.uad_residual(_ptr(x), _ptr(xr), _ptr(mk), n, hw, 1 if pos_only else 0, thr, _ptr(out),
                                         _ptr(l1), self._stream()))
        return out, l1


class Engine(_EvalOps):
    """One AE / VAE / ceVAE instance on one GPU.  Mirrors what a tf.Session + graph holds in the reference
    (trainers/VAE.py:18-29): variables, optimizer slots and the compiled step."""
    SCALARS_IN_PLACE = True          # forward(scalars_out=...) is honoured (trainers.AEMODEL.process)

    def __init__(self, arch, height=128, width=128, channels=1, inter_res=8, zdim=128, max_batch=64, device=None,
                 math='bf16x3', dim_c=9, dim_z=1, dim_w=1, c_lambda=1.0):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError('uad_hip needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU fallback')
        self.device = torch.device(device if device is not None else f'cuda:{torch.cuda.current_device()}')
        torch.cuda.set_device(self.device)
        self.arch = arch
        self.h, self.w, self.c, self.inter, self.zdim, self.max_batch = height, width, channels, inter_res, zdim, max_batch
        archs = {'AE': _lib.ARCH_AE, 'VAE': _lib.ARCH_VAE, 'ceVAE': _lib.ARCH_CEVAE,
                 'GMVAE_spatial': _lib.ARCH_GMVAE_SPATIAL, 'AE_spatial': _lib.ARCH_AE_SPATIAL}
        if arch not in archs:
            raise ValueError(f'unknown arch {arch!r}')
        self.dim_c, self.dim_z, self.dim_w = int(dim_c), int(dim_z), int(dim_w)
        cfg = _lib.UadConfig(archs[arch], height, width, channels, inter_res, zdim, max_batch, self.dim_c, self.dim_z,
                             self.dim_w, float(c_lambda))
        h = C.c_void_p()
        _lib.check(self.lib.uad_create(C.byref(cfg), C.byref(h)))
        self.handle = h
        # (parallel.DataParallelStep's torch.distributed/nccl path refuses a handle that is older than the process group: hardware-queue order)
        self.created_before_process_group = not (torch.distributed.is_available() and torch.distributed.is_initialized())
        self.nparams = int(self.lib.uad_param_count(h))
        self.spec = []
        name = C.create_string_buffer(160)
        off = C.c_longlong()
        rank = C.c_int()
        shape = (C.c_int * 4)()
        for i in range(self.lib.uad_num_tensors(h)):
            _lib.check(self.lib.uad_tensor_info(h, i, name, 160, C.byref(off), C.byref(rank), shape))
            self.spec.append((name.value.decode(), tuple(shape[:rank.value]), int(off.value)))
        self.flat = inter_res * inter_res * (self._cenc() // 8)
        self._views = {}
        self.set_math(math)

    def _cenc(self):
        n_pool = int(math.log2(self.h) - math.log2(self.inter))
        return min(128, 32 * 2 ** (n_pool - 1))

    def close(self):
        if getattr(self, 'handle', None):
            torch.cuda.synchronize(self.device)
            self._ar_comm = None              # (the communicator belongs to the DataParallelStep that attached it: parallel.DataParallelStep.close)
            self.lib.uad_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---------------------------------------------------------------- buffers
    def buffer(self, which=_lib.BUF_PARAMS, write=True):
        """Zero-copy torch view (1-D fp32) of a handle-owned flat buffer.  write=False: the caller promises only to READ the parameter view."""
        # A WRITE access of the parameter buffer goes through the library every time: uad_buffer(PARAMS) waits for a weight repack still
        # running on the handle's side stream and marks the packed copies stale, so a write through the (cached) view -- a second DP broadcast,
        # a checkpoint restore -- can neither race with the repack nor leave the next forward on old packed kernels.  Readers (write=False)
        # take the cached view and cost nothing; the other buffers are never repacked.
        if which not in self._views or (which == _lib.BUF_PARAMS and write):
            ptr = self.lib.uad_buffer(self.handle, which)
            if which not in self._views:
                self._views[which] = torch.as_tensor(_DevArray(ptr, self.nparams), device=self.device)
        return self._views[which]

    def check_fault(self, sync=True):
        """Raises RuntimeError when a fused bottleneck launch reported a timed-out sibling exchange (include/uad_hip.h: uad_check_fault);
        the optimizer updates behind such a launch were skipped on the device."""
        _lib.check(self.lib.uad_check_fault(self.handle, 1 if sync else 0, self._stream()))

    def set_fault_deferred(self, on=True):
        """Data-parallel runs: forward() / get_buffer_host() stop reporting a pending bottleneck fault; only check_fault() does, so that every rank
        reaches the epoch's agreement collective (include/uad_hip.h: uad_set_fault_deferred)."""
        _lib.check(self.lib.uad_set_fault_deferred(self.handle, 1 if on else 0))

    def grad_segment(self, seg):
        off, cnt = C.c_longlong(), C.c_longlong()
        _lib.check(self.lib.uad_grad_segment(self.handle, seg, C.byref(off), C.byref(cnt)))
        return int(off.value), int(cnt.value)

    def set_params(self, params):
        """params: flat float32 array, or dict name -> array (all tensors of the spec)."""
        if isinstance(params, dict):
            flat = np.concatenate([np.asarray(params[n], np.float32).reshape(-1) for n, _, _ in self.spec])
        else:
            flat = np.ascontiguousarray(params, np.float32).reshape(-1)
        _lib.check(self.lib.uad_set_params(self.handle, flat.ctypes.data_as(C.c_void_p), flat.size))

    def get_buffer_host(self, which=_lib.BUF_PARAMS):
        torch.cuda.synchronize(self.device)
        out = np.empty(self.nparams, np.float32)
        _lib.check(self.lib.uad_get_buffer(self.handle, which, out.ctypes.data_as(C.c_void_p), out.size))
        return out

    def set_buffer_host(self, which, flat):
        flat = np.ascontiguousarray(flat, np.float32).reshape(-1)
        _lib.check(self.lib.uad_set_buffer(self.handle, which, flat.ctypes.data_as(C.c_void_p), flat.size))

    def unflatten(self, flat):
        return {n: flat[o:o + int(np.prod(s))].reshape(s) for n, s, o in self.spec}

    def get_params(self):
        return self.unflatten(self.get_buffer_host(_lib.BUF_PARAMS))

    def get_grads(self):
        return self.unflatten(self.get_buffer_host(_lib.BUF_GRADS))

    def reset_optimizer(self):
        _lib.check(self.lib.uad_reset_optimizer(self.handle))

    @property
    def step_count(self):
        return int(self.lib.uad_get_step(self.handle))

    @step_count.setter
    def step_count(self, t):
        _lib.check(self.lib.uad_set_step(self.handle, int(t)))

    # ---------------------------------------------------------------- compute
    def _dev(self, a, shape=None):
        if a is None:
            return None
        if isinstance(a, torch.Tensor):
            t = a.to(device=self.device, dtype=torch.float32).contiguous()
        else:
            t = torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(self.device)
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise ValueError(f'expected shape {tuple(shape)}, got {tuple(t.shape)}')
        return t

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def forward(self, x, eps=None, masks=None, want_backward=False, want_l1=True, want_latents=True, x_ce=None,
                want_anomaly=True, scalars_out=None):
        """Returns a dict of DEVICE tensors: x_hat, L1 (opt), z_mu/z_log_sigma/z_sigma or z (opt), scalars [8]
        (reconstructionLoss, kl, loss, 0, Rec_vae, Rec_ce, loss_vae, 0), rec_per_sample [n] ([2n] for ceVAE).
        AE: x_ce (optional) = context-encoder training, the network reads x_ce and the L1 term compares with x (trainers/CE.py).
        ceVAE additionally takes x_ce (None = x) and masks 'mu_ce'/'dec_ce', and returns x_hat_ce, L1_vae / L1_ce
        (instead of L1) and -- filled in by backward() -- 'anomaly'.  want_backward: False | True | 'data' (data-gradient
        chain only: the ceVAE anomaly map without parameter gradients).  scalars_out: a contiguous fp32 device tensor of >= 8 elements that
        receives the scalars in place (the trainers' per-epoch table row: no copy per step).  Asynchronous on the current stream."""
        masks = masks or {}
        x = self._dev(x)
        if x.dim() != 4 or tuple(x.shape[1:]) != (self.h, self.w, self.c):
            raise ValueError(f'x must be [n,{self.h},{self.w},{self.c}], got {tuple(x.shape)}')
        n = x.shape[0]
        if n > self.max_batch:
            raise ValueError(f'batch {n} > max_batch {self.max_batch}')
        ce = self.arch == 'ceVAE'
        if x_ce is not None and self.arch not in ('ceVAE', 'AE', 'AE_spatial'):
            raise ValueError('x_ce is a ceVAE input, or the network input of a context-encoder step on an AE engine')
        # spatial AE: the latent (and its dropout mask) is the encoder feature map
        zs = (n, self.inter, self.inter, self._cenc()) if self.arch == 'AE_spatial' else (n, self.zdim)
        eps = self._dev(eps, zs)
        m_mu_ce = m_dec_ce = None
        if self.arch not in ('AE', 'AE_spatial'):
            m_mu, m_sg = self._dev(masks.get('mu'), zs), self._dev(masks.get('sigma'), zs)
            m_dec = self._dev(masks.get('dec'), (n, self.flat))
            if ce:
                m_mu_ce, m_dec_ce = self._dev(masks.get('mu_ce'), zs), self._dev(masks.get('dec_ce'), (n, self.flat))
        else:
            m_mu, m_sg, m_dec = self._dev(masks.get('z'), zs), None, None
        x_ce = self._dev(x_ce, x.shape)
        if scalars_out is not None and not (isinstance(scalars_out, torch.Tensor) and scalars_out.is_cuda and scalars_out.dtype == torch.float32
                                            and scalars_out.is_contiguous() and scalars_out.numel() >= 8):
            raise ValueError('scalars_out must be a contiguous fp32 CUDA tensor of at least 8 elements')
        out = {'x_hat': torch.empty_like(x), 'scalars': torch.empty(8, device=self.device) if scalars_out is None else scalars_out,
               'rec_per_sample': torch.empty(2 * n if ce else n, device=self.device)}
        if want_l1:
            out['L1_vae' if ce else 'L1'] = torch.empty_like(x)
        if ce:
            out['x_hat_ce'] = torch.empty_like(x)
            if want_l1:
                out['L1_ce'] = torch.empty_like(x)
            if want_backward and want_anomaly:
                out['anomaly'] = torch.empty_like(x)
        lat = {}
        if want_latents:
            if self.arch not in ('AE', 'AE_spatial'):
                lat = {k: torch.empty(zs, device=self.device) for k in ('z_mu', 'z_log_sigma', 'z_sigma')}
            else:
                lat = {'z': torch.empty(zs, device=self.device)}
            out.update(lat)
        io = _lib.UadIO(_ptr(x), _ptr(eps), _ptr(m_mu), _ptr(m_sg), _ptr(m_dec), _ptr(out['x_hat']),
                        _ptr(out.get('L1_vae' if ce else 'L1')), _ptr(lat.get('z_mu', lat.get('z'))),
                        _ptr(lat.get('z_log_sigma')), _ptr(lat.get('z_sigma')), _ptr(out['scalars']),
                        _ptr(out['rec_per_sample']), _ptr(x_ce), _ptr(m_mu_ce), _ptr(m_dec_ce), _ptr(out.get('x_hat_ce')),
                        _ptr(out.get('L1_ce')), _ptr(out.get('anomaly')))
        # keep the inputs alive until the (asynchronous) backward has consumed them
        self._keep = (x, eps, m_mu, m_sg, m_dec, x_ce, m_mu_ce, m_dec_ce, out)
        wb = 2 if want_backward == 'data' else (1 if want_backward else 0)
        _lib.check(self.lib.uad_forward(self.handle, C.byref(io), n, wb, self._stream()))
        return out

    # ---------------------------------------------------------------- spatial GMVAE
    def gm_forward(self, x, eps_w=None, eps_z=None, want_backward=False, want_l1=True, want_latents=True):
        """Spatial GMVAE forward + losses.  eps_w [n,r,r,dim_w], eps_z [n,r,r,dim_z] (None = 0).  Returns DEVICE tensors:
        x_hat (= xz_mu), L1 (opt), scalars [8] = (mean_p_loss, conditional_prior_loss, loss, w_prior_loss, c_prior_loss,
        0, 0, 0), rec_per_sample, and (opt) z_mu, z_log_sigma, w_mu, w_log_sigma, pc maps."""
        if self.arch != 'GMVAE_spatial':
            raise ValueError('gm_forward needs a GMVAE_spatial engine')
        x = self._dev(x)
        if x.dim() != 4 or tuple(x.shape[1:]) != (self.h, self.w, self.c):
            raise ValueError(f'x must be [n,{self.h},{self.w},{self.c}], got {tuple(x.shape)}')
        n, r = x.shape[0], self.inter
        if n > self.max_batch:
            raise ValueError(f'batch {n} > max_batch {self.max_batch}')
        eps_w = self._dev(eps_w, (n, r, r, self.dim_w))
        eps_z = self._dev(eps_z, (n, r, r, self.dim_z))
        new = lambda *shape: torch.empty(shape, device=self.device)
        out = {'x_hat': torch.empty_like(x), 'scalars': new(8), 'rec_per_sample': new(n)}
        if want_l1:
            out['L1'] = torch.empty_like(x)
        if want_latents:
            out.update(z_mu=new(n, r, r, self.dim_z), z_log_sigma=new(n, r, r, self.dim_z), w_mu=new(n, r, r, self.dim_w),
                       w_log_sigma=new(n, r, r, self.dim_w), pc=new(n, r, r, self.dim_c))
        io = _lib.UadIO(x=_ptr(x), x_hat=_ptr(out['x_hat']), l1_map=_ptr(out.get('L1')), z_mu=_ptr(out.get('z_mu')),
                        z_log_sigma=_ptr(out.get('z_log_sigma')), scalars=_ptr(out['scalars']),
                        rec_per_sample=_ptr(out['rec_per_sample']), eps_w=_ptr(eps_w), eps_z=_ptr(eps_z),
                        w_mu=_ptr(out.get('w_mu')), w_log_sigma=_ptr(out.get('w_log_sigma')), pc=_ptr(out.get('pc')))
        self._keep = (x, eps_w, eps_z, out)
        _lib.check(self.lib.uad_forward(self.handle, C.byref(io), n, 1 if want_backward else 0, self._stream()))
        return out

    def gm_train_step(self, x, eps_w=None, eps_z=None, lr=5e-5, beta1=0.5, beta2=0.999, adam_eps=1e-8, **kw):
        out = self.gm_forward(x, eps_w, eps_z, want_backward=True, **kw)
        self.backward(_lib.SEG_ALL)
        self.adam_step(lr, beta1, beta2, adam_eps)
        return out

    def restore_step(self, x_restored, eps_w=None, eps_z=None, tv_lambda=1.8, restore_lr=1e-3, want_grads=False):
        """One restoration iteration ON DEVICE (trainers/GMVAE_spatial.py:178-190): x_restored (a contiguous fp32 CUDA
        tensor [n,H,W,1]) is updated in place, x -= restore_lr * d(loss + tv_lambda * TV(x - xz_mu))/dx.  No host sync."""
        if not (isinstance(x_restored, torch.Tensor) and x_restored.is_cuda and x_restored.dtype == torch.float32
                and x_restored.is_contiguous()):
            raise ValueError('x_restored must be a contiguous fp32 CUDA tensor (it is updated in place)')
        n, r = x_restored.shape[0], self.inter
        if tuple(x_restored.shape[1:]) != (self.h, self.w, self.c) or n > self.max_batch:
            raise ValueError(f'x_restored must be [n<={self.max_batch},{self.h},{self.w},{self.c}]')
        if self.arch == 'VAE':      # trainers/VAE_You.py: eps_z is the [n,zDim] reparameterisation noise, the objective is per sample
            eps_w, eps_z = None, self._dev(eps_z, (n, self.zdim))
        else:
            eps_w = self._dev(eps_w, (n, r, r, self.dim_w))
            eps_z = self._dev(eps_z, (n, r, r, self.dim_z))
        grads = torch.empty_like(x_restored) if want_grads else None
        self._keep = (x_restored, eps_w, eps_z, grads)
        _lib.check(self.lib.uad_restore_step(self.handle, _ptr(x_restored), _ptr(eps_w), _ptr(eps_z), n, float(tv_lambda),
                                             float(restore_lr), _ptr(grads), self._stream()))
        return grads

    def backward(self, segment=_lib.SEG_ALL):
        _lib.check(self.lib.uad_backward(self.handle, segment, self._stream()))

    def backward_deferred(self, segment):
        """uad_backward_deferred: runs the segment and returns None when its gradient slice is complete in the current stream's order, else a
        torch stream (the handle's side stream) in whose order it is -- the data-parallel layer issues the slice's all-reduce under that stream
        instead of stalling the compute stream on it."""
        cur = self._stream()
        ready = C.c_void_p()
        _lib.check(self.lib.uad_backward_deferred(self.handle, segment, cur, C.byref(ready)))
        if (ready.value or 0) == (cur.value or 0):
            return None
        cache = self.__dict__.setdefault('_ext_streams', {})
        if ready.value not in cache:
            cache[ready.value] = torch.cuda.ExternalStream(ready.value, device=self.device)
        return cache[ready.value]

    def allreduce_attach(self, comm, world, plan):
        """uad_allreduce_attach: comm = a parallel.RcclComm (or None to detach), plan = [(segment after which to issue, offset, count)] as
        parallel.bucket_plan returns it.  From then on backward_allreduce(segment) enqueues the buckets' RCCL all-reduces itself."""
        if comm is None:
            _lib.check(self.lib.uad_allreduce_attach(self.handle, None, 1, 0, None, None, None))
            self._ar_comm = None
            return
        n = len(plan)
        after = (C.c_int * n)(*[int(p[0]) for p in plan])
        off = (C.c_longlong * n)(*[int(p[1]) for p in plan])
        cnt = (C.c_longlong * n)(*[int(p[2]) for p in plan])
        _lib.check(self.lib.uad_allreduce_attach(self.handle, C.c_void_p(comm.handle), int(world), n, after, off, cnt))
        self._ar_comm = comm                  # keep the communicator alive as long as the handle refers to it

    def backward_allreduce(self, segment):
        """uad_backward_allreduce: one backward segment + the library-issued all-reduce of the buckets that complete with it."""
        _lib.check(self.lib.uad_backward_allreduce(self.handle, segment, self._stream()))

    OPTIMIZERS = {'ADAM': 0, 'SGD': 1, 'MOMENTUM': 2, 'RMS': 3}

    def set_optimizer(self, kind='ADAM', momentum=0.9):
        """Selects what adam_step() applies (DLMODEL.create_optimizer, trainers/DLMODEL.py:113-123).  A fresh RMSProp run gets TF's slot
        initialisation (`rms` = 1); Adam / Momentum slots start at 0."""
        if kind not in self.OPTIMIZERS:
            raise ValueError('Invalid optimizer type')
        self.optimizer, self.momentum = kind, float(momentum)
        if kind == 'RMS' and self.step_count == 0:
            self.set_buffer_host(_lib.BUF_ADAM_V, np.ones(self.nparams, np.float32))

    def adam_step(self, lr, beta1=0.5, beta2=0.999, eps=1e-8, grad_scale=1.0):
        """The optimizer step of the trainers (TF-Adam unless set_optimizer() chose SGD / MOMENTUM / RMS)."""
        kind = getattr(self, 'optimizer', 'ADAM')
        if kind == 'ADAM':
            _lib.check(self.lib.uad_adam_step(self.handle, lr, beta1, beta2, eps, grad_scale, self._stream()))
        else:
            _lib.check(self.lib.uad_optimizer_step(self.handle, self.OPTIMIZERS[kind], lr, self.momentum, 0.9, 1e-10, grad_scale, self._stream()))

    def train_step(self, x, eps=None, masks=None, lr=1e-4, beta1=0.5, beta2=0.999, adam_eps=1e-8, **kw):
        """forward + backward + Adam; ceVAE: pass x_ce=... (its 'anomaly' output is filled by the backward)."""
        out = self.forward(x, eps, masks, want_backward=True, **kw)
        self.backward(_lib.SEG_ALL)
        self.adam_step(lr, beta1, beta2, adam_eps)
        return out

    def debug_buffer(self, name):
        """tests: torch view of a named intermediate of the last forward (include/uad_hip.h: uad_debug_buffer)."""
        ptr, cnt = C.c_void_p(), C.c_longlong()
        _lib.check(self.lib.uad_debug_buffer(self.handle, name.encode(), C.byref(ptr), C.byref(cnt)))
        if not ptr.value:
            return int(cnt.value)           # flag entries ("fused_final")
        return torch.as_tensor(_DevArray(ptr.value, cnt.value), device=self.device)

    def set_math(self, math):
        """'f32' (exact fp32 MFMA), 'bf16x3' (split-bf16 on the bf16 matrix cores, ~2^-17 relative product error) or 'bf16x6' (three bf16 planes per
        operand, six products: fp32-grade results -- tests hold 1e-5 against the fp64 oracle -- at 3/8 of the fp32 MFMA's matrix-pipe time)."""
        modes = {'f32': _lib.MATH_F32, 'bf16x3': _lib.MATH_BF16X3, 'bf16x6': _lib.MATH_BF16X6}
        if math not in modes:
            raise ValueError(f'unknown math mode {math!r}')
        _lib.check(self.lib.uad_set_math_mode(self.handle, modes[math]))
        self.math = math

    def profile(self, on):
        _lib.check(self.lib.uad_profile_enable(self.handle, 1 if on else 0))

    def profile_report(self):
        """{tag: (count, total_ms)} of the launch groups recorded since the last report (synchronises)."""
        buf = C.create_string_buffer(1 << 16)
        _lib.check(self.lib.uad_profile_report(self.handle, buf, len(buf)))
        rep = {}
        for line in buf.value.decode().splitlines():
            tag, cnt, ms = line.split()
            rep[tag] = (int(cnt), float(ms))
        return rep


class Scores:
    """trainers/Metrics.py's threshold metrics from ONE device sort (uad_scores_*): auroc (sklearn roc_curve + auc),
    auprc (sklearn average_precision_score), dice(pred > t, label) for arbitrary thresholds."""

    def __init__(self, engine, predictions, labels):
        self.lib = engine.lib
        self._stream = engine._stream
        p = engine._dev(predictions).reshape(-1)
        if isinstance(labels, torch.Tensor):
            y = labels.to(device=engine.device, dtype=torch.float32).reshape(-1)
        else:
            y = torch.from_numpy(np.ascontiguousarray(np.asarray(labels).reshape(-1) != 0, np.float32)).to(engine.device)
        if p.numel() != y.numel():
            raise ValueError(f'predictions ({p.numel()}) and labels ({y.numel()}) differ in size')
        h = C.c_void_p()
        _lib.check(self.lib.uad_scores_create(_ptr(p), _ptr(y), p.numel(), C.byref(h), self._stream()))
        self.handle = h
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        _lib.check(self.lib.uad_scores_auc(h, C.byref(a), C.byref(b), C.byref(c)))
        self.auroc, self.auprc, self.positives = a.value, b.value, c.value

    def dice_at(self, thresholds):
        t = np.ascontiguousarray(np.atleast_1d(thresholds), np.float64)
        out = np.empty_like(t)
        _lib.check(self.lib.uad_scores_dice(self.handle, t.ctypes.data_as(C.POINTER(C.c_double)), t.size,
                                            out.ctypes.data_as(C.POINTER(C.c_double)), self._stream()))
        return out

    def close(self):
        if getattr(self, 'handle', None):
            self.lib.uad_scores_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
