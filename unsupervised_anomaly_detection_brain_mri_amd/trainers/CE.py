"""trainers/CE.py — only `retrieve_masked_batch` (CE.py:123-139), the host-side context masking that the ceVAE
trainer imports (ceVAE.py:8,93).  The CE trainer class itself (an AE fed the masked batch) is not on the north-star path."""
import random as _random

import numpy as np


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
