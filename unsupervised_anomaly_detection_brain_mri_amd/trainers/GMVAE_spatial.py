"""trainers/GMVAE_spatial.py — spatial Gaussian-mixture VAE (You et al.): four-term loss (L1 reconstruction,
conditional-prior KL weighted by the mixture posterior, w-prior KL, clamped c-prior; GMVAE_spatial.py:61-89), Adam on
`loss`, and restoration-mode inference: restore_steps gradient steps on the INPUT of
loss + tv_lambda * TV(x - reconstruction) (:91-92, 168-199).

The reference runs each restoration step as a batch-1 `sess.run` with a device->host->device round trip of the image
(150 per slice).  Here `uad_restore_step` keeps the image on the device and updates it in place; reconstruct() enqueues
all steps for a whole batch of slices back to back and synchronises once."""
from collections import defaultdict
from math import inf

import numpy as np
import torch

from .AEMODEL import AEMODEL, Phase, indicate_early_stopping
from ..engine import Engine


class GMVAE_spatial(AEMODEL):
    class Config(AEMODEL.Config):
        def __init__(self):            # GMVAE_spatial.py:12-21
            super().__init__('GMVAE_spatial')
            self.dim_c = 6
            self.dim_z = 1
            self.dim_w = 1
            self.c_lambda = 1
            self.restore_lr = 1e-3
            self.restore_steps = 150
            self.tv_lambda = 1.8

    ARCH = 'GMVAE_spatial'
    ARCHS = ('GMVAE_spatial', 'GMVAE_You')

    def __new__(cls, sess=None, config=None, network=None, **kw):
        # the reference pairs this trainer with two spatial models; the original-architecture one runs on the materialised-graph handle
        if cls is GMVAE_spatial and getattr(network, 'arch', None) == 'GMVAE_You':
            from .GMVAE import GMVAE_You
            return object.__new__(GMVAE_You)
        return object.__new__(cls)

    SCALAR_KEYS = ('reconstructionLoss', 'mean_p_loss', 'conditional_prior_loss', 'w_prior_loss', 'c_prior_loss', 'loss')
    _SCALAR_SLOT = {'reconstructionLoss': 0, 'mean_p_loss': 0, 'conditional_prior_loss': 1, 'loss': 2, 'w_prior_loss': 3,
                    'c_prior_loss': 4}

    def __init__(self, sess, config=None, network=None, **kw):
        super().__init__(sess, config, network, **kw)
        c = self.config
        self.dim_c, self.dim_z, self.dim_w, self.c_lambda = c.dim_c, c.dim_z, c.dim_w, c.c_lambda
        self.restore_lr, self.restore_steps, self.tv_lambda_value = c.restore_lr, c.restore_steps, c.tv_lambda

    def _make_engine(self, device):
        c = self.config
        return Engine(self.ARCH, c.outputHeight, c.outputWidth, c.numChannels, int(c.intermediateResolutions[0]),
                      max_batch=max(int(c.batchsize), 1), device=device, dim_c=c.dim_c, dim_z=c.dim_z, dim_w=c.dim_w,
                      c_lambda=float(c.c_lambda))

    def _draw(self, n, dropout=False):
        """The two reparameterisation noises of one sess.run (model :27,32); the graph has no dropout layer."""
        r = self.engine.inter
        return (self.rng.standard_normal((n, r, r, self.config.dim_w)).astype(np.float32),
                self.rng.standard_normal((n, r, r, self.config.dim_z)).astype(np.float32))

    # ------------------------------------------------------------------ one sess.run of process()
    def step(self, batch, phase, *, eps=None, fetch_maps=True):
        """GMVAE_spatial.py:139-156.  Returns reconstruction, L1, L2, L1_sum, L2_sum and the scalar losses.  The reference
        also fetches `restore` / `grads` on every step because they sit in self.losses; they cost a second backward and are
        used by nothing in process(), so they are served by restore_gradients() on demand instead."""
        phase = Phase(phase) if not isinstance(phase, Phase) else phase
        train = phase == Phase.TRAIN
        e_w, e_z = self._draw(len(batch)) if eps is None else eps
        c = self.config
        if train:
            if self.dp.world > 1:
                out = self._dp_train(batch, e_w, e_z, fetch_maps)
            else:
                out = self.engine.gm_train_step(batch, e_w, e_z, lr=c.learningrate, beta1=c.beta1, want_l1=fetch_maps,
                                                want_latents=False)
        else:
            out = self.engine.gm_forward(batch, e_w, e_z, want_backward=False, want_l1=fetch_maps, want_latents=False)
        sc = self.dp.allreduce_scalars(out['scalars'].clone()).cpu().numpy()
        run = {k: np.float32(sc[i]) for k, i in self._SCALAR_SLOT.items()}
        if fetch_maps:
            run['reconstruction'] = out['x_hat'].cpu().numpy()
            run['L1'] = out['L1'].cpu().numpy()
            run['L2'] = run['L1'] ** 2
            run['L1_sum'] = out['rec_per_sample'].cpu().numpy()
            run['L2_sum'] = np.float32(run['L2'].sum())
        return run

    def _dp_train(self, batch, e_w, e_z, fetch_maps):
        from .. import _lib
        import torch.distributed as dist
        eng, c = self.engine, self.config
        out = eng.gm_forward(batch, e_w, e_z, want_backward=True, want_l1=fetch_maps, want_latents=False)
        works = []
        for seg in (_lib.SEG_DECODER, _lib.SEG_BOTTLENECK, _lib.SEG_ENCODER):
            eng.backward(seg)
            off, cnt = self.dp.segs[seg]
            works.append(dist.all_reduce(self.dp.grads[off:off + cnt], op=dist.ReduceOp.SUM, async_op=True))
        for w in works:
            w.wait()
        eng.adam_step(c.learningrate, c.beta1, 0.999, 1e-8, 1.0 / self.dp.world)
        return out

    def process(self, dataset, epoch, phase, optim=None, visualization_keys=None):     # GMVAE_spatial.py:135-166
        phase = Phase(phase) if not isinstance(phase, Phase) else phase
        scalars = defaultdict(list)
        num_batches = self._num_batches(dataset, phase)
        for idx in range(num_batches):
            batch, _, _ = self._shard(dataset, phase)
            run = self.step(batch, phase, fetch_maps=False)
            print(f'Epoch ({phase.value}): [{epoch:2d}] [{idx:4d}/{num_batches:4d}] loss: {run["loss"]:.8f}')
            for k, v in run.items():
                if np.ndim(v) == 0:
                    scalars[k].append(v)
        out = {k: np.mean(v) for k, v in scalars.items()}
        for k, v in out.items():
            self.curves.setdefault(f'{phase.value}/{k}', []).append(float(v))
        self.log_to_tensorboard(epoch, out, None, phase)
        return out

    def train(self, dataset):          # GMVAE_spatial.py:55-133
        self.create_optimizer(type=getattr(self.config, 'optimizer', 'ADAM'))
        best_cost, last_improvement = inf, 0
        last_epoch = self.load_checkpoint()
        for epoch in range(last_epoch, self.config.numEpochs):
            self.process(dataset, epoch, Phase.TRAIN, optim=True)
            last_epoch += 1
            self.save(self.checkpointDir, last_epoch)
            val_scalars = self.process(dataset, epoch, Phase.VAL)
            best_cost, last_improvement, stop = indicate_early_stopping(val_scalars['loss'], best_cost, last_improvement)
            if stop:
                print('Early stopping was triggered due to no improvement over the last 5 epochs')
                break
        if self.tv_lambda_value == -1 and self.restore_steps > 0:
            print('Determining best lambda')
            self.determine_best_lambda(dataset)

    # ------------------------------------------------------------------ restoration
    def restore_gradients(self, x, eps=None, tv_lambda=None):
        """The `grads` fetch (:91-92) at x, without moving x."""
        x = np.asarray(x, np.float32)
        xr = torch.from_numpy(np.ascontiguousarray(x)).to(self.engine.device)
        e_w, e_z = self._draw(len(x)) if eps is None else eps
        tv = self.tv_lambda_value if tv_lambda is None else tv_lambda
        return self.engine.restore_step(xr, e_w, e_z, tv_lambda=tv, restore_lr=0.0, want_grads=True).cpu().numpy()

    def _restore(self, x, steps, tv_lambda, eps=None):
        """steps x (x -= restore_lr * grads) on device for a batch of slices; fresh noise per step like the graph's
        tf.random_normal (eps: optional callable step -> (e_w, e_z), or a fixed pair)."""
        xr = torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(self.engine.device)
        n, r, c = len(x), self.engine.inter, self.config
        g = torch.Generator(device=self.engine.device).manual_seed(int(self.rng.integers(1 << 31)))
        for step in range(steps):
            if eps is None:
                e_w = torch.randn((n, r, r, c.dim_w), device=self.engine.device, generator=g)
                e_z = torch.randn((n, r, r, c.dim_z), device=self.engine.device, generator=g)
            else:
                e_w, e_z = eps(step) if callable(eps) else eps
            self.engine.restore_step(xr, e_w, e_z, tv_lambda=tv_lambda, restore_lr=self.restore_lr)
        return xr.cpu().numpy()

    def reconstruct(self, x, dropout=False, eps=None):     # GMVAE_spatial.py:168-199
        x = np.asarray(x, np.float32)
        if x.ndim < 4:
            x = np.expand_dims(x, 0)
        if eps is not None and np.isscalar(eps):           # eps=0.0: deterministic mode (as AEMODEL.reconstruct)
            v = float(eps)
            eps = (lambda step: (None, None)) if v == 0.0 else None
        bs = self.engine.max_batch
        parts = []
        for s0 in range(0, len(x), bs):
            xb = x[s0:s0 + bs]
            if self.restore_steps == 0:
                e_w, e_z = self._draw(len(xb)) if eps is None else (eps(0) if callable(eps) else eps)
                parts.append(self.engine.gm_forward(xb, e_w, e_z, want_l1=False, want_latents=False)['x_hat'].cpu().numpy())
            else:
                parts.append(self._restore(xb, self.restore_steps, self.tv_lambda_value, eps))
        rec = np.concatenate(parts, axis=0)
        results = {'reconstruction': rec}
        results['l1err'] = np.sum(np.abs(x - rec))
        results['l2err'] = np.sum(np.sqrt((x - rec) ** 2))
        return results

    def determine_best_lambda(self, dataset):               # GMVAE_spatial.py:201-225
        lambdas = np.arange(20) / 10.0
        mean_errors = []
        for tv_lambda in lambdas:
            errors = []
            for _ in range(int(dataset.num_batches(self.config.batchsize, set=Phase.VAL.value) * 0.2)):
                batch, _, _ = dataset.next_batch(self.config.batchsize, set=Phase.VAL.value)
                restored = self._restore(batch, self.restore_steps, float(tv_lambda))
                errors.append(np.sum(np.abs(batch - restored)))
            mean_error = np.mean(errors) if errors else np.nan      # np.mean([]) is nan in the reference too
            mean_errors.append(mean_error)
            print(f'mean_error for lambda {tv_lambda}: {mean_error}')
        self.tv_lambda_value = lambdas[mean_errors.index(min(mean_errors))]
        print(f'Best lambda: {self.tv_lambda_value}')
