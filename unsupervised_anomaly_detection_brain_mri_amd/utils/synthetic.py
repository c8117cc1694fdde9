"""Synthetic Brainweb-like slices and a dataset object with the reference loaders' duck-type
(dataloaders/BRAINWEB.py:406-478: num_batches / next_batch / num_channels), for tests and bench.py
(no Brainweb / MS data can be downloaded here)."""
import numpy as np


def synthetic_slices(n, h=128, w=128, seed=0, dtype=np.float32, lesions=False):
    """fp32 NHWC [n,h,w,1] in [0,1]: elliptical brain (~45 % of pixels), smooth field + N(0,0.03), background exactly 0
    (mimics skull-strip + background removal + max scaling, utils/default_config_setup.py:237-240, utils/NII.py:67-70).
    lesions=True also returns a label map with planted hyper-intense blobs."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    cy, cx = (h - 1) / 2.0, (w - 1) / 2.0
    out = np.zeros((n, h, w, 1), dtype=dtype)
    lab = np.zeros((n, h, w), dtype=np.int32)
    msk = np.zeros((n, h, w), dtype=np.int32)
    for i in range(n):
        ry = h * rng.uniform(0.36, 0.42)
        rx = w * rng.uniform(0.30, 0.36)
        mask = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
        ph = rng.uniform(0, 2 * np.pi, 4)
        f = (0.55 + 0.2 * np.sin(yy / h * 2 * np.pi * 1.5 + ph[0]) * np.cos(xx / w * 2 * np.pi * 1.2 + ph[1])
             + 0.12 * np.sin(xx / w * 2 * np.pi * 3 + ph[2]) * np.sin(yy / h * 2 * np.pi * 2.5 + ph[3]))
        img = np.clip(f + rng.normal(0, 0.03, f.shape), 0.0, 1.0)
        if lesions:
            for _ in range(rng.integers(1, 4)):
                ly, lx = cy + rng.uniform(-0.5, 0.5) * ry, cx + rng.uniform(-0.5, 0.5) * rx
                r = rng.uniform(2.0, 5.0) * h / 128.0
                blob = (yy - ly) ** 2 + (xx - lx) ** 2 <= r * r
                img = np.where(blob, np.clip(img + 0.35, 0, 1), img)
                lab[i][blob & mask] = 1
        out[i, :, :, 0] = (img * mask).astype(dtype)
        msk[i] = mask
    if lesions:
        return out, lab, msk
    return out


class SyntheticDataset:
    """Reference dataset duck-type: images fp32 [S,H,W,1] in RAM, TRAIN/VAL split, next_batch cursor."""
    SET_TYPES = ['TRAIN', 'VAL', 'TEST']

    def __init__(self, n_train=64, n_val=16, h=128, w=128, seed=0):
        self.num_channels = 1
        self._data = {'TRAIN': synthetic_slices(n_train, h, w, seed), 'VAL': synthetic_slices(n_val, h, w, seed + 1),
                      'TEST': synthetic_slices(max(1, n_val), h, w, seed + 2)}
        self._cursor = {k: 0 for k in self._data}
        self._rng = np.random.default_rng(seed + 7)

    def num_batches(self, batchsize, set='TRAIN'):
        return len(self._data[set]) // batchsize

    def next_batch(self, batch_size, shuffle=True, set='TRAIN', return_brainmask=False):
        data = self._data[set]
        start = self._cursor[set]
        if start + batch_size > len(data):
            if shuffle:
                self._data[set] = data = data[self._rng.permutation(len(data))]
            start = 0
        self._cursor[set] = start + batch_size
        batch = data[start:start + batch_size]
        assert batch.size, "The batch is empty!"
        labels = np.zeros(batch.shape[:3], np.int32)
        if return_brainmask:
            return batch, labels, (batch[..., 0] > 0).astype(np.int32)
        return batch, labels, None
