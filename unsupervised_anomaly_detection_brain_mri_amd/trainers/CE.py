"""trainers/CE.py — the context encoder: an autoencoder that READS the context-masked batch and is scored against the clean one
(CE.py:19-21,33-34,87-92), plus `retrieve_masked_batch` (CE.py:123-139), the host-side masking the ceVAE trainer imports too
(ceVAE.py:8,93).  On the device this is the AE handle with io.x_ce set (include/uad_hip.h): encoder input = x_ce, L1 target = x."""
import random as _random
from collections import defaultdict

import numpy as np

from .AE import AE
from .AEMODEL import Phase


def retrieve_masked_batch(batch, brainmasks, rng=None):
    """Zero 1-3 random 20x20 squares inside each slice's brain bounding box.

    Behaviour kept from the reference, defect included: the per-sample loop re-binds the name of the
    mask array, so what multiplies the batch at the end is the LAST sample's [H,W,C] mask, broadcast over all samples
    (the earlier samples' squares are drawn -- consuming RNG -- and then discarded).
    rng: object with randint(a, b), both ends inclusive; default the `random` module like the reference."""
    rng = _random if rng is None else rng
    if hasattr(batch, 'cpu'):            # device tensors of utils.slice_cache.DeviceDataset: the masking RNG and geometry are host-side
        batch = batch.cpu().numpy()
    if hasattr(brainmasks, 'cpu'):
        brainmasks = brainmasks.cpu().numpy()
    batch = np.asarray(batch)
    boxes = []
    for bm in brainmasks:
        rows, cols = np.nonzero(np.asarray(bm).reshape(batch.shape[1], batch.shape[2], -1).any(axis=-1))
        boxes.append((int(rows.min()), int(rows.max()), int(cols.min()), int(cols.max())))
    side = 20
    last = None
    for r0, r1, c0, c1 in boxes:
        last = np.ones(batch.shape[1:], batch.dtype)
        for _ in range(rng.randint(1, 3)):
            if r0 < r1 - side and c0 < c1 - side:
                r = rng.randint(r0, r1 - side)
                c = rng.randint(c0, c1 - side)
                last[r:r + side, c:c + side] = 0
    return batch * last


class CE(AE):
    """loss = reconstructionLoss = mean_n sum |x - network(x_ce)| (CE.py:33-34).  TRAIN feeds the masked batch, VAL / reconstruct() feed
    x_ce = x (:91,112).  Accepts the AE-family networks (autoencoder, autoencoder_spatial): the trainer's loss has no KL term."""

    class Config(AE.Config):
        def __init__(self):            # CE.py:13-15
            super().__init__()
            self.model_name = self.modelname = 'CE'

    def step(self, batch, phase, *, x_ce=None, eps=None, dropout_masks=None, fetch_maps=True):
        phase = Phase(phase) if not isinstance(phase, Phase) else phase
        train = phase == Phase.TRAIN
        masks = self._noise(len(batch), dropout=train)[1] if dropout_masks is None else dropout_masks
        c = self.config
        if train:
            out = self.dp.train_step(batch, None, masks, lr=c.learningrate, beta1=c.beta1, want_l1=fetch_maps, x_ce=x_ce)
        else:
            out = self.engine.forward(batch, None, masks, want_backward=False, want_l1=fetch_maps, x_ce=x_ce)
        sc = self.dp.allreduce_scalars(out['scalars'].clone()).cpu().numpy()
        run = {'reconstructionLoss': np.float32(sc[0]), 'loss': np.float32(sc[0])}
        if fetch_maps:
            run['reconstruction'] = out['x_hat'].cpu().numpy()
            run['L1'] = out['L1'].cpu().numpy()
        return run

    def process(self, dataset, epoch, phase, optim=None):       # CE.py:70-101
        phase = Phase(phase) if not isinstance(phase, Phase) else phase
        scalars = defaultdict(list)
        num_batches = self._num_batches(dataset, phase)
        for idx in range(num_batches):
            batch, _, brainmasks = self._shard(dataset, phase, return_brainmask=True)
            masked_batch = retrieve_masked_batch(batch, brainmasks, rng=getattr(self, 'mask_rng', None))    # drawn in every phase (:77), fed in TRAIN only (:91)
            run = self.step(batch, phase, x_ce=masked_batch if phase == Phase.TRAIN else None, fetch_maps=False)
            print(f'Epoch ({phase.value}): [{epoch:2d}] [{idx:4d}/{num_batches:4d}] loss: {run["loss"]:.8f}')
            for k, v in run.items():
                if np.ndim(v) == 0:
                    scalars[k].append(v)
        out = {k: np.mean(v) for k, v in scalars.items()}
        for k, v in out.items():
            self.curves.setdefault(f'{phase.value}/{k}', []).append(float(v))
        self.log_to_tensorboard(epoch, out, None, phase)
        return out
