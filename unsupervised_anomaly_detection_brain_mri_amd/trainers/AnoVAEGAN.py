"""trainers/AnoVAEGAN.py — AnoVAE-GAN: per TRAIN batch one VAE step (optim_vae: reconstructionLoss + kl_weight * kl over the
Encoder and Generator variables), one generator step (optim_gen: -mean D(G(E(x)))) and five critic steps (optim_dis, WGAN-GP)
(:97-150); validation evaluates the VAE fetches (:160-183), early stopping on reconstructionLoss (:187); reconstruct() :191-208.
Losses :50-71, the three Adam(beta1 .5, beta2 .9) optimisers :82-84.  Every sess.run is one GanEngine.phase() (+ adam); the
reparameterisation noise, the interpolation alpha and the dropout masks come from the trainer's host RNG."""
from collections import defaultdict
from math import inf

import numpy as np
import torch

from ..gan_engine import GanEngine
from ..parallel import GanDataParallel
from .AEMODEL import AEMODEL, Phase, indicate_early_stopping


class AnoVAEGAN(AEMODEL):
    class Config(AEMODEL.Config):
        def __init__(self):          # trainers/AnoVAEGAN.py:12-17
            super().__init__('AnoVAEGAN')
            self.scale = 10.0
            self.kappa = 1.0
            self.kl_weight = 1.0

    ARCH = 'AnoVAEGAN'
    D_ITERS = 5                      # :96
    GROUPS = ('Encoder', 'Generator', 'Discriminator')

    def _make_engine(self, device):
        c = self.config
        return GanEngine(c.outputHeight, c.outputWidth, c.numChannels, int(c.intermediateResolutions[0]), c.zDim,
                         max_batch=max(int(c.batchsize), 1), scale=float(getattr(c, 'scale', 10.0)), device=device, variant='anovaegan',
                         kl_weight=float(getattr(c, 'kl_weight', 1.0)))

    def _make_dp(self, world):
        return GanDataParallel(self.engine, world)

    def _draw(self, n, train):
        """eps ~ N(0,1) [n,zDim]; dropout keep-masks of the mu / log-sigma heads (anovaegan.py:31-32).  The dropout on the
        generator's dense output is called without `training=` (anovaegan.py:44) and therefore never active."""
        eps = self.rng.standard_normal((n, self.config.zDim)).astype(np.float32)
        r = float(self.config.dropout_rate)
        if not train or r <= 0:
            return dict(eps=eps)
        keep = lambda: (self.rng.random((n, self.config.zDim)) >= r).astype(np.float32) / (1.0 - r)
        return dict(eps=eps, mask_z=keep(), mask_sigma=keep())

    # ------------------------------------------------------------------ the three sess.runs
    def step(self, batch, phase, *, fetch_maps=True, inputs=None):
        """optimizer_e fetch (:100-113) in TRAIN, the VAL fetch (:167-180) otherwise: reconstruction, reconstructionLoss, L1, enc_loss
        (+ kl)."""
        phase = Phase(phase) if not isinstance(phase, Phase) else phase
        train = phase == Phase.TRAIN
        kw = dict(x=batch, want_images=fetch_maps, want_l1=fetch_maps, **(inputs or self._draw(len(batch), train)))
        if train:
            out = self.dp.train_phase('Encoder', self.config.learningrate, **kw)
        else:
            out = self.engine.phase('Encoder', want_backward=False, **kw)
        keys = ('reconstructionLoss', 'kl', 'enc_loss')
        sc = self.dp.allreduce_scalars(torch.stack([out[k] for k in keys])).cpu().numpy()
        run = {k: np.float32(v) for k, v in zip(keys, sc)}
        run['loss'] = run['reconstructionLoss']
        if fetch_maps:
            run['reconstruction'] = out['reconstruction'].cpu().numpy()
            run['L1'] = out['L1'].cpu().numpy()
        return run

    def generator_step(self, batch, inputs=None):
        out = self.dp.train_phase('Generator', self.config.learningrate, x=batch, want_images=False, **(inputs or self._draw(len(batch), True)))
        return {'gen_loss': np.float32(out['gen_loss'].item())}

    def discriminator_step(self, batch, inputs=None, alpha=None):
        n = len(batch)
        alpha = self.rng.uniform(0.0, 1.0, (n,)).astype(np.float32) if alpha is None else alpha
        out = self.dp.train_phase('Discriminator', self.config.learningrate, x=batch, alpha=alpha, want_images=False,
                                  **(inputs or self._draw(n, True)))
        return {k: np.float32(out[k].item()) for k in ('disc_loss', 'disc_fake', 'disc_real')}

    # ------------------------------------------------------------------ epoch loop (:87-189)
    def train(self, dataset):
        self.create_optimizer(type=getattr(self.config, 'optimizer', 'ADAM'))
        c = self.config
        best_cost, last_improvement = inf, 0
        last_epoch = self.load_checkpoint()
        for epoch in range(last_epoch, c.numEpochs):
            scalars = defaultdict(list)
            num_batches = self._num_batches(dataset, Phase.TRAIN)
            for idx in range(num_batches):
                batch, _, _ = self._shard(dataset, Phase.TRAIN)
                run = self.step(batch, Phase.TRAIN, fetch_maps=False)
                run = {**run, **self.generator_step(batch)}
                for _ in range(self.D_ITERS):
                    run = {**run, **self.discriminator_step(batch)}
                print(f'Epoch (TRAIN): [{epoch:2d}] [{idx:4d}/{num_batches:4d}] gen_loss: {run["gen_loss"]:.8f}, disc_loss: '
                      f'{run["disc_loss"]:.8f}, reconstructionLoss: {run["reconstructionLoss"]:.8f}')
                for k, v in run.items():
                    if np.ndim(v) == 0:
                        scalars[k].append(v)
            for k, v in scalars.items():
                self.curves.setdefault(f'TRAIN/{k}', []).append(float(np.mean(v)))
            last_epoch += 1
            self.save(self.checkpointDir, last_epoch)
            val = self.process(dataset, epoch, Phase.VAL)
            best_cost, last_improvement, stop = indicate_early_stopping(val['reconstructionLoss'], best_cost, last_improvement)
            if stop:
                print('Early stopping was triggered due to no improvement over the last 5 epochs')
                break

    def reconstruct(self, x, dropout=False, eps=None):                 # :191-208 (z is sampled at eval too; eps=0 pins it)
        x = np.asarray(x, np.float32)
        if x.ndim < 4:
            x = np.expand_dims(x, 0)
        inp = self._draw(len(x), bool(dropout))
        if eps is not None:
            inp['eps'] = None if (np.isscalar(eps) and float(eps) == 0.0) else np.broadcast_to(np.asarray(eps, np.float32), (len(x), self.config.zDim)).copy()
        out = self.engine.reconstruct(x, mask_z=inp.get('mask_z'), eps=inp.get('eps'), mask_sigma=inp.get('mask_sigma'))
        rec = out['reconstruction'].cpu().numpy()
        return {'reconstruction': rec, 'l1err': np.sum(np.abs(x - rec)), 'l2err': np.sum(np.sqrt((x - rec) ** 2))}

    # ------------------------------------------------------------------ checkpoint: three Adam step counters (+ the second Generator slots)
    def _adam_steps(self):
        return np.array([self.engine.step_count(g) for g in self.GROUPS], np.int64)

    def _set_adam_steps(self, t):
        for g, v in zip(self.GROUPS, np.atleast_1d(t)):
            self.engine.set_step_count(g, int(v))
