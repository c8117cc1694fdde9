"""trainers/fAnoGAN.py — f-AnoGAN: WGAN-GP training of Generator / Discriminator for numEpochs epochs (1 generator step and
5 critic steps per batch, :87-138), then izi_f encoder training with validation and early stopping on reconstructionLoss for
up to numEpochs more (:140-210); reconstruct() = G(E(x)) (:220-239).
Losses :50-66, the three Adam(beta1 .5, beta2 .9) optimisers :71-77.  Every sess.run of the reference is one
GanEngine.phase() (+ adam) here; z, the interpolation alpha and the dropout masks come from the trainer's host RNG."""
from collections import defaultdict
from math import inf

import numpy as np
import torch

from ..gan_engine import GanEngine
from ..parallel import GanDataParallel
from .AEMODEL import AEMODEL, Phase, indicate_early_stopping


class fAnoGAN(AEMODEL):
    class Config(AEMODEL.Config):
        def __init__(self):          # trainers/fAnoGAN.py:12-16
            super().__init__('fAnoGAN')
            self.scale = 10.0
            self.kappa = 1.0

    ARCH = 'fAnoGAN'
    D_ITERS = 5                      # trainers/fAnoGAN.py:96

    def _make_engine(self, device):
        c = self.config
        return GanEngine(c.outputHeight, c.outputWidth, c.numChannels, int(c.intermediateResolutions[0]), c.zDim,
                         max_batch=max(int(c.batchsize), 1), scale=float(getattr(c, 'scale', 10.0)),
                         kappa=float(getattr(c, 'kappa', 1.0)), device=device, variant=getattr(self.network, 'variant', 'unified'))

    def _make_dp(self, world):
        return GanDataParallel(self.engine, world)

    # ------------------------------------------------------------------ RNG inputs of one sess.run
    def sample_z(self, batch_size=None):         # trainers/fAnoGAN.py:241
        return self.rng.standard_normal((batch_size if batch_size else self.config.batchsize, self.config.zDim)).astype(np.float32)

    def _keep(self, shape, train):
        r = float(self.config.dropout_rate)
        if not train or r <= 0 or self.engine.variant == 'resnet':      # the ResNet graph has no dropout layers
            return None
        return (self.rng.random(shape) >= r).astype(np.float32) / (1.0 - r)

    @staticmethod
    def _scalars(out, keys):
        return {k: np.float32(out[k].item()) for k in keys}

    # ------------------------------------------------------------------ the three sess.runs
    def generator_step(self, batch, z=None, mask_g=None):
        """optimizer_g fetch (:100-113): {'generated', 'gen_loss'} after one Adam step on the Generator variables."""
        n = len(batch)
        z = self.sample_z(n) if z is None else z
        mask_g = self._keep((n, self.engine.flat), True) if mask_g is None else mask_g
        c = self.config
        out = self.dp.train_phase('Generator', c.learningrate, z=z, mask_g=mask_g)
        run = self._scalars(out, ('gen_loss',))
        run['generated'] = out['generated']
        return run

    def discriminator_step(self, batch, z=None, alpha=None, mask_g=None):
        """optimizer_d fetch (:115-130): {'generated', 'disc_loss', 'disc_fake', 'disc_real'}."""
        n = len(batch)
        z = self.sample_z(n) if z is None else z
        alpha = self.rng.uniform(0.0, 1.0, (n,)).astype(np.float32) if alpha is None else alpha
        mask_g = self._keep((n, self.engine.flat), True) if mask_g is None else mask_g
        c = self.config
        out = self.dp.train_phase('Discriminator', c.learningrate, x=batch, z=z, alpha=alpha, mask_g=mask_g)
        run = self._scalars(out, ('disc_loss', 'disc_fake', 'disc_real'))
        run['generated'] = out['generated']
        return run

    def step(self, batch, phase, *, fetch_maps=True, mask_z=None, mask_g=None):
        """Encoder-phase sess.run (:146-166 TRAIN with optimizer_enc, :179-199 VAL): the encoder losses, 'reconstruction',
        'z_enc' (+ 'L1').  The critic-side entries of **self.losses that the reference also evaluates there (gen_loss,
        disc_loss on a fresh z) are not part of this phase's objective and are not computed."""
        phase = Phase(phase) if not isinstance(phase, Phase) else phase
        train = phase == Phase.TRAIN
        n = len(batch)
        mask_z = self._keep((n, self.config.zDim), train) if mask_z is None else mask_z
        mask_g = self._keep((n, self.engine.flat), train) if mask_g is None else mask_g
        c = self.config
        kw = dict(x=batch, mask_z=mask_z, mask_g=mask_g, want_images=fetch_maps, want_l1=fetch_maps)
        if train:
            out = self.dp.train_phase('Encoder', c.learningrate, **kw)
        else:
            out = self.engine.phase('Encoder', want_backward=False, **kw)
        keys = ('loss_img', 'loss_fts', 'enc_loss', 'reconstructionLoss')
        sc = self.dp.allreduce_scalars(torch.stack([out[k] for k in keys])).cpu().numpy()
        run = {k: np.float32(v) for k, v in zip(keys, sc)}
        run['loss'] = run['reconstructionLoss']
        if fetch_maps:
            run['reconstruction'] = out['reconstruction'].cpu().numpy()
            run['L1'] = out['L1'].cpu().numpy()
            run['z_enc'] = out['z_enc'].cpu().numpy()
        return run

    # ------------------------------------------------------------------ epoch loops
    def train(self, dataset):
        self.create_optimizer(type=getattr(self.config, 'optimizer', 'ADAM'))
        c = self.config
        best_cost, last_improvement = inf, 0
        last_epoch = self.load_checkpoint()
        for epoch in range(last_epoch, c.numEpochs):               # WGAN epochs (:87-138)
            scalars = defaultdict(list)
            num_batches = self._num_batches(dataset, Phase.TRAIN)
            for idx in range(num_batches):
                batch, _, _ = self._shard(dataset, Phase.TRAIN)
                run = self.generator_step(batch)
                for _ in range(self.D_ITERS):
                    run = {**run, **self.discriminator_step(batch)}
                print(f'Epoch (TRAIN WGAN): [{epoch:2d}] [{idx:4d}/{num_batches:4d}] gen_loss: {run["gen_loss"]:.8f}, '
                      f'disc_loss: {run["disc_loss"]:.8f}')
                for k, v in run.items():
                    if np.ndim(v) == 0:
                        scalars[k].append(v)
            for k, v in scalars.items():
                self.curves.setdefault(f'TRAIN/wgan_{k}', []).append(float(np.mean(v)))
            last_epoch += 1
            self.save(self.checkpointDir, last_epoch)
        for epoch in range(last_epoch, 2 * c.numEpochs):           # encoder epochs (:140-210)
            self.process(dataset, epoch, Phase.TRAIN, optim=True)
            last_epoch += 1
            self.save(self.checkpointDir, last_epoch)
            val = self.process(dataset, epoch, Phase.VAL)
            best_cost, last_improvement, stop = indicate_early_stopping(val['reconstructionLoss'], best_cost, last_improvement)
            if stop:
                print('Early stopping was triggered due to no improvement over the last 5 epochs')
                break

    def reconstruct(self, x, dropout=False, eps=None):              # trainers/fAnoGAN.py:220-239 (eps: unused, the graph has no sampling)
        x = np.asarray(x, np.float32)
        if x.ndim < 4:
            x = np.expand_dims(x, 0)
        n = len(x)
        out = self.engine.reconstruct(x, self._keep((n, self.config.zDim), bool(dropout)), self._keep((n, self.engine.flat), bool(dropout)))
        rec = out['reconstruction'].cpu().numpy()
        return {'reconstruction': rec, 'l1err': np.sum(np.abs(x - rec)), 'l2err': np.sum(np.sqrt((x - rec) ** 2))}

    # ------------------------------------------------------------------ checkpoint: three Adam step counters
    GROUPS = ('Encoder', 'Generator', 'Discriminator')

    def _adam_steps(self):
        return np.array([self.engine.step_count(g) for g in self.GROUPS], np.int64)

    def _set_adam_steps(self, t):
        for g, v in zip(self.GROUPS, np.atleast_1d(t)):
            self.engine.set_step_count(g, int(v))
