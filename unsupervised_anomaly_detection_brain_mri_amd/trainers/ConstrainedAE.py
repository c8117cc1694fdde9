"""trainers/ConstrainedAE.py / AAE.py / ConstrainedAAE.py — the dense-bottleneck autoencoders with a latent re-encoding constraint
and / or a latent WGAN-GP critic.  One shared implementation (`_LatentAE`) on the AAE-family handle (uad_gan_*, UAD_GAN_AAE):
  ConstrainedAE   loss = mean_n(L2_n + rho * Rec_z_n), one Adam (create_optimizer)                       ConstrainedAE.py:36-49
  AAE             optim_ae (mean L2, every AE variable), optim_dis, optim_gen (-mean d_, 'Encoder' vars)  AAE.py:40-67, loop :82-126
  ConstrainedAAE  optim_ae (mean(L2 + rho Rec_z)), optim_dis, optim_gen                                   ConstrainedAAE.py:44-70
Every sess.run is one GanEngine.aae_phase() (+ adam); the prior sample z, the interpolation eps and the dropout masks come from the
trainer's host RNG."""
from collections import defaultdict
from math import inf

import numpy as np
import torch

from ..gan_engine import GanEngine
from ..parallel import GanDataParallel
from .AEMODEL import AEMODEL, Phase, indicate_early_stopping


class _LatentAE(AEMODEL):
    KIND = 'constrained_ae'
    D_ITERS = 20                     # AAE.py:82 / ConstrainedAAE.py:89
    GROUPS = ('Encoder', 'AE', 'Discriminator')

    def _make_engine(self, device):
        c = self.config
        return GanEngine(c.outputHeight, c.outputWidth, c.numChannels, int(c.intermediateResolutions[0]), c.zDim,
                         max_batch=max(int(c.batchsize), 1), scale=float(getattr(c, 'scale', 10.0)), device=device, variant='aae',
                         aae_kind=self.KIND, rho=float(getattr(c, 'rho', 1.0)))

    def _make_dp(self, world):
        return GanDataParallel(self.engine, world)

    def sample_z(self, batch_size=None):         # AAE.py:195
        return self.rng.standard_normal((batch_size if batch_size else self.config.batchsize, self.config.zDim)).astype(np.float32)

    def _keep(self, shape, on):
        r = float(self.config.dropout_rate)
        if not on or r <= 0:
            return None
        return (self.rng.random(shape) >= r).astype(np.float32) / (1.0 - r)

    def _masks(self, n, train):
        """dropout masks of z_, dec_dense and z_rec.  In constrained_adversarial_autoencoder.py the last two Dropout calls carry no
        `training=` (:35,47) and are therefore never active."""
        if self.KIND == 'caae_chen':                 # models/constrained_adversarial_autoencoder_Chen.py has no Dropout layer
            return dict(mask_z=None, mask_dec=None, mask_rec=None)
        both = self.KIND != 'constrained_aae'
        return dict(mask_z=self._keep((n, self.config.zDim), train), mask_dec=self._keep((n, self.engine.flat), train and both),
                    mask_rec=self._keep((n, self.config.zDim), train and both and self.KIND == 'constrained_ae'))

    def _phase(self, which, lr, **kw):
        """GanDataParallel.train_phase for the AAE family (all-reduce of the phase's gradient slice, then its Adam)."""
        import torch.distributed as dist
        out = self.engine.aae_phase(which, want_backward=True, **kw)
        if self.dp.world > 1:
            off, cnt = self.engine.group(which)
            dist.all_reduce(self.dp.grads[off:off + cnt], op=dist.ReduceOp.SUM)
        b1, b2 = (self.config.beta1, 0.999) if self.KIND == 'constrained_ae' else (0.5, 0.9)     # create_optimizer vs the explicit Adams
        self.engine.adam(which, lr, b1, b2, 1e-8, 1.0 / self.dp.world)
        return out

    # ------------------------------------------------------------------ one sess.run of the autoencoder fetches
    def step(self, batch, phase, *, fetch_maps=True, masks=None):
        phase = Phase(phase) if not isinstance(phase, Phase) else phase
        train = phase == Phase.TRAIN
        kw = dict(x=batch, want_images=fetch_maps, want_l1=fetch_maps, **(masks if masks is not None else self._masks(len(batch), train)))
        out = self._phase('AE', self.config.learningrate, **kw) if train else self.engine.aae_phase('AE', want_backward=False, **kw)
        keys = ('loss', 'L2', 'Rec_z', 'reconstructionLoss')
        sc = self.dp.allreduce_scalars(torch.stack([out[k] for k in keys])).cpu().numpy()
        run = {k: np.float32(v) for k, v in zip(keys, sc)}
        if self.KIND == 'aae':
            run.pop('Rec_z')
        if fetch_maps:
            run['reconstruction'] = out['reconstruction'].cpu().numpy()
            run['L1'] = out['L1'].cpu().numpy()
        return run

    def discriminator_step(self, batch, z=None, eps=None, mask_z=None):
        n = len(batch)
        z = self.sample_z(n) if z is None else z
        if eps is None:      # one coefficient per sample; ..._Chen.py:118 draws ONE scalar per run (tf.random_uniform([]))
            eps = (np.full((n,), self.rng.uniform(0.0, 1.0), np.float32) if self.KIND == 'caae_chen'
                   else self.rng.uniform(0.0, 1.0, (n,)).astype(np.float32))
        chen = self.KIND == 'caae_chen'
        out = self._phase('Discriminator', self.config.learningrate, x=batch, z=z, eps=eps,
                          mask_z=None if chen else (self._keep((n, self.config.zDim), True) if mask_z is None else mask_z))
        return {k: np.float32(out[k].item()) for k in ('disc_loss', 'disc_fake', 'disc_real')}

    def generator_step(self, batch, mask_z=None):
        out = self._phase('Encoder', self.config.learningrate, x=batch,
                          mask_z=None if self.KIND == 'caae_chen' else (self._keep((len(batch), self.config.zDim), True) if mask_z is None else mask_z))
        return {'gen_loss': np.float32(out['gen_loss'].item())}

    # ------------------------------------------------------------------ epoch loops
    def train(self, dataset):
        self.create_optimizer(type=getattr(self.config, 'optimizer', 'ADAM'))
        c = self.config
        best_cost, last_improvement = inf, 0
        last_epoch = self.load_checkpoint()
        adversarial = self.KIND != 'constrained_ae'
        for epoch in range(last_epoch, c.numEpochs):
            if not adversarial:
                self.process(dataset, epoch, Phase.TRAIN, optim=True)
            else:
                scalars = defaultdict(list)
                num_batches = self._num_batches(dataset, Phase.TRAIN)
                for idx in range(num_batches):
                    batch, _, _ = self._shard(dataset, Phase.TRAIN)
                    run = {}
                    for _ in range(self.D_ITERS if epoch <= 5 else 1):
                        run = self.step(batch, Phase.TRAIN, fetch_maps=False)
                    for _ in range(self.D_ITERS):
                        run = {**run, **self.discriminator_step(batch)}
                    run = {**run, **self.generator_step(batch)}
                    print(f'Epoch (TRAIN): [{epoch:2d}] [{idx:4d}/{num_batches:4d}] loss: {run["reconstructionLoss"]:.8f}, gen_loss: '
                          f'{run["gen_loss"]:.8f}, disc_loss: {run["disc_loss"]:.8f}')
                    for k, v in run.items():
                        if np.ndim(v) == 0:
                            scalars[k].append(v)
                for k, v in scalars.items():
                    self.curves.setdefault(f'TRAIN/{k}', []).append(float(np.mean(v)))
            last_epoch += 1
            self.save(self.checkpointDir, last_epoch)
            val = self.process(dataset, epoch, Phase.VAL)
            key = 'reconstructionLoss' if adversarial else 'loss'          # AAE.py:164 / ConstrainedAE.py:69
            best_cost, last_improvement, stop = indicate_early_stopping(val[key], best_cost, last_improvement)
            if stop:
                print('Early stopping was triggered due to no improvement over the last 5 epochs')
                break

    def reconstruct(self, x, dropout=False, eps=None):
        x = np.asarray(x, np.float32)
        if x.ndim < 4:
            x = np.expand_dims(x, 0)
        mk = self._masks(len(x), bool(dropout))
        rec = self.engine.reconstruct(x, mask_z=mk['mask_z'], mask_g=mk['mask_dec'])['reconstruction'].cpu().numpy()
        return {'reconstruction': rec, 'l1err': np.sum(np.abs(x - rec)), 'l2err': np.sum(np.sqrt((x - rec) ** 2))}

    def _adam_steps(self):
        return np.array([self.engine.step_count(g) for g in self.GROUPS], np.int64)

    def _set_adam_steps(self, t):
        for g, v in zip(self.GROUPS, np.atleast_1d(t)):
            self.engine.set_step_count(g, int(v))


class ConstrainedAE(_LatentAE):
    class Config(AEMODEL.Config):
        def __init__(self):          # trainers/ConstrainedAE.py:12-15
            super().__init__('ConstrainedAE')
            self.rho = 1

    ARCH = 'ConstrainedAE'
    KIND = 'constrained_ae'
