"""trainers/ceVAE.py — context-encoding VAE (Zimmerer et al.): loss = mean(rec_vae + kl + rec_ce) over the two branches
of models/context_encoder_variational_autoencoder.py; anomaly = L1_vae * |d mean(rec_vae + kl) / d x| is a fetch of
EVERY step because it sits in self.losses (ceVAE.py:38-51,95-99).  The reference builds two TF backward
graphs for that (weights via `loss`, input via `loss_vae`); here the input gradient falls out of the one backward pass,
because the VAE-branch samples' d loss / d x equals d loss_vae / d x."""
from collections import defaultdict
from math import inf

import numpy as np

from .AEMODEL import AEMODEL, Phase, indicate_early_stopping
from .CE import retrieve_masked_batch


class ceVAE(AEMODEL):
    class Config(AEMODEL.Config):
        def __init__(self):
            super().__init__('ceVAE')
            self.use_gradient_based_restoration = True      # ceVAE.py:16

    ARCH = 'ceVAE'
    ARCHS = ('ceVAE', 'ceVAE_Zimmerer')      # models/context_encoder_variational_autoencoder.py | ..._Zimmerer.py (no dropout layers)
    SCALAR_KEYS = ('Rec_ce', 'Rec_vae', 'reconstructionLoss', 'kl', 'loss', 'loss_vae')
    _SCALAR_SLOT = {'reconstructionLoss': 0, 'kl': 1, 'loss': 2, 'Rec_vae': 4, 'Rec_ce': 5, 'loss_vae': 6}

    def _make_engine(self, device):
        if self.arch != 'ceVAE_Zimmerer':
            return super()._make_engine(device)
        from ..gan_engine import ZimmererEngine
        c = self.config
        return ZimmererEngine(c.outputHeight, c.outputWidth, c.numChannels, int(c.intermediateResolutions[0]), c.zDim,
                              max_batch=max(int(c.batchsize), 1), device=device, cevae=True)

    def _noise_layout(self, dropout):
        z, f = self.config.zDim, self.engine.flat
        lay = [('eps', z, 'normal')]
        if dropout and self.config.dropout_rate > 0 and self.arch != 'ceVAE_Zimmerer':
            lay += [('mu', z, 'keep'), ('mu_ce', z, 'keep'), ('sigma', z, 'keep'), ('dec', f, 'keep'), ('dec_ce', f, 'keep')]
        return lay

    def _draw(self, n, dropout):
        """eps + the five independent dropout masks of one sess.run: the single Dropout layer object is called on z_mu,
        z_mu_ce, z_log_sigma and on both dec_dense outputs (context_encoder_variational_autoencoder.py:36-43)."""
        z = self.config.zDim
        eps = self.rng.standard_normal((n, z)).astype(np.float32)
        if not dropout or self.config.dropout_rate <= 0 or self.arch == 'ceVAE_Zimmerer':
            return eps, None
        r = float(self.config.dropout_rate)
        keep = lambda shape: (self.rng.random(shape) >= r).astype(np.float32) / (1.0 - r)
        f = self.engine.flat
        return eps, {'mu': keep((n, z)), 'mu_ce': keep((n, z)), 'sigma': keep((n, z)), 'dec': keep((n, f)),
                     'dec_ce': keep((n, f))}

    def step(self, batch, phase, *, masked_batch=None, eps=None, dropout_masks=None, fetch_maps=True):
        """One sess.run of ceVAE.process (ceVAE.py:95-110).  x_ce = masked_batch in TRAIN, the batch itself otherwise
        (:105).  Returns the reference's fetch keys; the maps (reconstruction, reconstruction_ce, L1*, anomaly) only
        when fetch_maps."""
        phase = Phase(phase) if not isinstance(phase, Phase) else phase
        train = phase == Phase.TRAIN
        n = len(batch)
        if eps is None and dropout_masks is None:
            eps, masks = self._noise(n, dropout=train)           # drawn on the device (AEMODEL._noise)
        else:
            d_eps, d_masks = self._draw(n, dropout=train)
            eps = d_eps if eps is None else eps
            masks = d_masks if dropout_masks is None else dropout_masks
        x_ce = masked_batch if (train and masked_batch is not None) else None
        c = self.config
        if train:
            out = self.dp.train_step(batch, eps, masks, lr=c.learningrate, beta1=c.beta1, want_l1=fetch_maps,
                                     x_ce=x_ce, want_anomaly=fetch_maps)
        else:
            out = self.engine.forward(batch, eps, masks, want_backward='data' if fetch_maps else False,
                                      want_l1=fetch_maps, x_ce=x_ce)
            if fetch_maps:
                self.engine.backward()
        sc = self.dp.allreduce_scalars(out['scalars'].clone()).cpu().numpy()
        run = {k: np.float32(sc[i]) for k, i in self._SCALAR_SLOT.items()}
        if fetch_maps:
            run['reconstruction'] = out['x_hat'].cpu().numpy()
            run['reconstruction_ce'] = out['x_hat_ce'].cpu().numpy()
            run['L1_vae'] = out['L1_vae'].cpu().numpy()
            run['L1_ce'] = out['L1_ce'].cpu().numpy()
            run['L1'] = 0.5 * (run['L1_vae'] + run['L1_ce'])                  # ceVAE.py:40
            run['anomaly'] = out['anomaly'].cpu().numpy()
        return run

    def process(self, dataset, epoch, phase, optim=None, visualization_keys=None):       # ceVAE.py:86-117
        phase = Phase(phase) if not isinstance(phase, Phase) else phase
        scalars = defaultdict(list)
        num_batches = self._num_batches(dataset, phase)
        for idx in range(num_batches):
            batch, _, brainmasks = self._shard(dataset, phase, return_brainmask=True)
            masked_batch = retrieve_masked_batch(batch, brainmasks)        # drawn in every phase, used in TRAIN only
            run = self.step(batch, phase, masked_batch=masked_batch, fetch_maps=False)
            print(f'Epoch ({phase.value}): [{epoch:2d}] [{idx:4d}/{num_batches:4d}] loss: {run["loss"]:.8f}')
            for k, v in run.items():
                if np.ndim(v) == 0:
                    scalars[k].append(v)
        out = {k: np.mean(v) for k, v in scalars.items()}
        for k, v in out.items():
            self.curves.setdefault(f'{phase.value}/{k}', []).append(float(v))
        self.log_to_tensorboard(epoch, out, None, phase)
        return out

    def train(self, dataset):       # ceVAE.py:33-84
        self.create_optimizer(type=getattr(self.config, 'optimizer', 'ADAM'))
        best_cost, last_improvement = inf, 0
        last_epoch = self.load_checkpoint()
        keys = ['reconstruction', 'reconstruction_ce', 'anomaly']
        for epoch in range(last_epoch, self.config.numEpochs):
            self.process(dataset, epoch, Phase.TRAIN, optim=True, visualization_keys=keys)
            last_epoch += 1
            self.save(self.checkpointDir, last_epoch)
            val_scalars = self.process(dataset, epoch, Phase.VAL, visualization_keys=keys)
            best_cost, last_improvement, stop = indicate_early_stopping(val_scalars['loss'], best_cost, last_improvement)
            if stop:
                print('Early stopping was triggered due to no improvement over the last 5 epochs')
                break

    RECONSTRUCT_PER_SLICE = True          # utils/Evaluation.evaluate_volume passes per_slice=True

    def reconstruct(self, x, dropout=False, eps=None, per_slice=False):
        """ceVAE.py:119-144: x_ce = x; fetches the reconstruction and every loss incl. `anomaly`; when
        config.use_gradient_based_restoration is truthy the returned 'reconstruction' is x - c * anomaly (sic: 'not the
        real reconstruction but treated like it', :138-141).  eps as in AEMODEL.reconstruct.
        `anomaly` = L1_vae * |d mean_n(rec_vae + kl) / d x| carries the 1/n of the batch mean: a batch of n slices gives every slice's
        map scaled by 1/n relative to n single-slice calls.  That is what the reference's graph returns for a batch too, but its evaluation
        only ever feeds one slice (Evaluation.py:246-250); per_slice=True rescales by n so that a batched call equals the slice-by-slice
        calls (row i of reconstruct(x[:k], per_slice=True) == reconstruct(x[i:i+1]) for the same noise)."""
        x = np.asarray(x, np.float32)
        if x.ndim < 4:
            x = np.expand_dims(x, 0)
        d_eps, masks = self._noise(len(x), dropout=bool(dropout))
        if eps is None:
            eps = d_eps
        elif np.isscalar(eps):
            eps = None if float(eps) == 0.0 else np.full((len(x), self.config.zDim), eps, np.float32)
        out = self.engine.forward(x, eps, masks, want_backward='data', want_l1=True, want_latents=False)
        self.engine.backward()
        sc = out['scalars'].cpu().numpy()
        results = {k: np.float32(sc[i]) for k, i in self._SCALAR_SLOT.items()}
        results['L1_vae'] = out['L1_vae'].cpu().numpy()
        results['L1_ce'] = out['L1_ce'].cpu().numpy()
        results['L1'] = 0.5 * (results['L1_vae'] + results['L1_ce'])
        results['anomaly'] = out['anomaly'].cpu().numpy()
        if per_slice:
            results['anomaly'] = results['anomaly'] * np.float32(len(x))
        rec = out['x_hat'].cpu().numpy()
        c = self.config.use_gradient_based_restoration
        if c:
            rec = x - np.float32(c) * results['anomaly']
        results['reconstruction'] = rec
        results['l1err'] = np.sum(np.abs(x - rec))
        results['l2err'] = np.sum(np.sqrt((x - rec) ** 2))
        return results
