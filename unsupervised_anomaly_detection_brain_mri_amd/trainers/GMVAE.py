"""trainers/GMVAE.py — dense Gaussian-mixture VAE: the four-term loss of GMVAE.py:56-91 (L1 reconstruction, conditional-prior KL weighted
by p(c|z), w-prior KL, clamped c-prior), Adam on `loss` over every variable, and restoration-mode inference (:93-94, 158-188).
The epoch loop, early stopping, reconstruct() and determine_best_lambda() are those of the spatial trainer; what differs is the
handle (uad_gan_* with UAD_GAN_AAE / aae_kind 3), the latent shapes ([n, dim_w], [n, dim_z]) and that this graph HAS dropout: on w_mu,
w_log_sigma, z_mu and dec_dense (model :37-45; z_log_sigma's Dropout is called without `training` and never fires)."""
import numpy as np
import torch

from ..gan_engine import GanEngine
from ..parallel import GanDataParallel
from .AEMODEL import Phase
from .GMVAE_spatial import GMVAE_spatial


class GMVAE(GMVAE_spatial):
    class Config(GMVAE_spatial.Config):
        def __init__(self):            # GMVAE.py:12-21
            super().__init__()
            self.model_name = self.modelname = 'GMVAE'

    ARCH = 'GMVAE'
    ARCHS = ('GMVAE',)
    HAS_DROPOUT = True
    SCALAR_KEYS = ('reconstructionLoss', 'mean_p_loss', 'conditional_prior_loss', 'w_prior_loss', 'c_prior_loss', 'loss')
    GROUPS = ('AE',)

    def _make_engine(self, device):
        c = self.config
        return GanEngine(c.outputHeight, c.outputWidth, c.numChannels, int(c.intermediateResolutions[0]), zdim=int(c.dim_z),
                         max_batch=max(int(c.batchsize), 1), device=device, variant='aae', aae_kind='gmvae', dim=int(c.dim_c),
                         dim_w=int(c.dim_w), c_lambda=float(c.c_lambda))

    def _make_dp(self, world):
        return GanDataParallel(self.engine, world)

    def _eps_shapes(self, n):
        return (n, self.config.dim_w), (n, self.config.dim_z)

    def _draw(self, n, dropout=False):
        sw, sz = self._eps_shapes(n)
        return self.rng.standard_normal(sw).astype(np.float32), self.rng.standard_normal(sz).astype(np.float32)

    def _masks(self, n, on):
        r = float(self.config.dropout_rate)
        if not on or r <= 0:
            return None
        c = self.config
        keep = lambda shape: (self.rng.random(shape) >= r).astype(np.float32) / (1.0 - r)
        return {'w_mu': keep((n, c.dim_w)), 'w_ls': keep((n, c.dim_w)), 'z_mu': keep((n, c.dim_z)), 'dec': keep((n, self.engine.flat))}

    # ------------------------------------------------------------------ one sess.run of process() (GMVAE.py:116-139)
    def step(self, batch, phase, *, eps=None, fetch_maps=True, masks=None):
        import torch.distributed as dist
        phase = Phase(phase) if not isinstance(phase, Phase) else phase
        train = phase == Phase.TRAIN
        e_w, e_z = self._draw(len(batch)) if eps is None else eps
        if masks is None:
            masks = self._masks(len(batch), train)
        c = self.config
        out = self.engine.gm_phase(batch, e_w, e_z, masks, want_backward=train, want_l1=fetch_maps)
        if train:
            if self.dp.world > 1:
                off, cnt = self.engine.group('AE')
                dist.all_reduce(self.dp.grads[off:off + cnt], op=dist.ReduceOp.SUM)
            self.engine.adam('AE', c.learningrate, c.beta1, 0.999, 1e-8, 1.0 / self.dp.world)
        sc = self.dp.allreduce_scalars(torch.stack([out[k] for k in self.SCALAR_KEYS])).cpu().numpy()
        run = {k: np.float32(v) for k, v in zip(self.SCALAR_KEYS, sc)}
        if fetch_maps:
            run['reconstruction'] = out['reconstruction'].cpu().numpy()
            run['L1'] = out['L1'].cpu().numpy()
            run['L2'] = run['L1'] ** 2
            run['L1_sum'] = run['L1'].reshape(len(batch), -1).sum(axis=1)
            run['L2_sum'] = np.float32(run['L2'].sum())
        return run

    # ------------------------------------------------------------------ restoration (GMVAE.py:158-188)
    def restore_gradients(self, x, eps=None, tv_lambda=None, dropout=False):
        x = np.asarray(x, np.float32)
        xr = torch.from_numpy(np.ascontiguousarray(x)).to(self.engine.device)
        e_w, e_z = self._draw(len(x)) if eps is None else eps
        tv = self.tv_lambda_value if tv_lambda is None else tv_lambda
        return self.engine.gm_restore_step(xr, e_w, e_z, self._masks(len(x), dropout), tv_lambda=tv, restore_lr=0.0, want_grads=True).cpu().numpy()

    def _restore(self, x, steps, tv_lambda, eps=None, dropout=False):
        xr = torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(self.engine.device)
        n, c, dev = len(x), self.config, self.engine.device
        g = torch.Generator(device=dev).manual_seed(int(self.rng.integers(1 << 31)))
        r = float(c.dropout_rate)
        for step in range(steps):
            if eps is None:
                sw, sz = self._eps_shapes(n)
                e_w = torch.randn(sw, device=dev, generator=g)
                e_z = torch.randn(sz, device=dev, generator=g)
            else:
                e_w, e_z = eps(step) if callable(eps) else eps
            masks = None
            if dropout and r > 0 and self.HAS_DROPOUT:        # `dropout` is fed on every restoration run (:176): fresh masks per step, drawn on the device
                keep = lambda shape: (torch.rand(shape, device=dev, generator=g) >= r).float() / (1.0 - r)
                masks = {'w_mu': keep((n, c.dim_w)), 'w_ls': keep((n, c.dim_w)), 'z_mu': keep((n, c.dim_z)), 'dec': keep((n, self.engine.flat))}
            self.engine.gm_restore_step(xr, e_w, e_z, masks, tv_lambda=tv_lambda, restore_lr=self.restore_lr)
        return xr.cpu().numpy()

    def reconstruct(self, x, dropout=False, eps=None):
        x = np.asarray(x, np.float32)
        if x.ndim < 4:
            x = np.expand_dims(x, 0)
        if eps is not None and np.isscalar(eps):
            v = float(eps)
            eps = (lambda step: (None, None)) if v == 0.0 else None
        bs = self.engine.max_batch
        parts = []
        for s0 in range(0, len(x), bs):
            xb = x[s0:s0 + bs]
            if self.restore_steps == 0:
                e_w, e_z = self._draw(len(xb)) if eps is None else (eps(0) if callable(eps) else eps)
                parts.append(self.engine.gm_phase(xb, e_w, e_z, self._masks(len(xb), dropout), want_backward=False, want_l1=False)['reconstruction'].cpu().numpy())
            else:
                parts.append(self._restore(xb, self.restore_steps, self.tv_lambda_value, eps, dropout))
        rec = np.concatenate(parts, axis=0)
        return {'reconstruction': rec, 'l1err': np.sum(np.abs(x - rec)), 'l2err': np.sum(np.sqrt((x - rec) ** 2))}

    def _adam_steps(self):
        return np.array([self.engine.step_count('AE')], np.int64)

    def _set_adam_steps(self, t):
        self.engine.set_step_count('AE', int(np.atleast_1d(t)[0]))


class GMVAE_You(GMVAE):
    """models/gaussian_mixture_variational_autoencoder_You.py under trainers/GMVAE_spatial.py (what `GMVAE_spatial(sess, config, network=
    gaussian_mixture_variational_autoencoder_You)` instantiates): the spatial trainer's losses / restoration on the original architecture
    (k3 layers, latent maps on the H/4 grid, decoder on z_sampled, no dropout layer).  Handle: uad_gan_* with aae_kind 6."""
    Config = GMVAE_spatial.Config
    ARCH = 'GMVAE_You'
    ARCHS = ('GMVAE_You',)
    HAS_DROPOUT = False

    def _make_engine(self, device):
        c = self.config
        if int(c.intermediateResolutions[0]) * 4 != int(c.outputHeight):
            raise ValueError('gaussian_mixture_variational_autoencoder_You: the latent map is outputHeight / 4; set intermediateResolutions accordingly')
        return GanEngine(c.outputHeight, c.outputWidth, c.numChannels, int(c.intermediateResolutions[0]), zdim=int(c.dim_z),
                         max_batch=max(int(c.batchsize), 1), device=device, variant='aae', aae_kind='gmvae_you', dim=int(c.dim_c),
                         dim_w=int(c.dim_w), c_lambda=float(c.c_lambda), math='bf16x3_all')      # parity-rated for this stack: tests/test_gpu_gmvae_you.py, both modes

    def _eps_shapes(self, n):
        r = self.engine.inter
        return (n, r, r, self.config.dim_w), (n, r, r, self.config.dim_z)

    def _masks(self, n, on):
        return None
