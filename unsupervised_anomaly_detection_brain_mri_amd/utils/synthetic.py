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


class _Volume:
    """The slice accessors the reference's utils/NII.py wrapper offers to utils/Evaluation._evaluate (:205-220): data [z,y,x],
    shape(), num_slices_along_axis(axis), get_slice(s, axis)."""
    _AX = {'axial': 0, 'coronal': 1, 'sagittal': 2, 0: 0, 1: 1, 2: 2}

    def __init__(self, data):
        self.data = np.asarray(data)

    def shape(self):
        return self.data.shape

    def num_slices_along_axis(self, axis):
        return self.data.shape[self._AX[axis]]

    def get_slice(self, s, axis):
        return np.take(self.data, s, axis=self._AX[axis])


class _Options:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class SyntheticPatientDataset(SyntheticDataset):
    """A patient-structured stand-in for the reference's lesion datasets (dataloaders/BRAINWEB.py & co.) with exactly the members
    utils/Evaluation.py reads: `patients` (dicts with 'name', 'filtered_files', 'groundtruth_filename'),
    `get_patient_idx(split)`, `load_volume_and_groundtruth(nii_filename, patient) -> (volume, segmentation, skullmap)`,
    `options.{sliceStart, sliceEnd, axis, sliceResolution}`, plus the training duck-type of SyntheticDataset.
    Volumes are [slices, native, native] with planted hyper-intense lesions; native != sliceResolution exercises the zoom step."""

    def __init__(self, n_val=2, n_test=2, slices=16, native=None, h=128, w=128, seed=0, slice_start=0, slice_end=None, axis='axial'):
        super().__init__(max(8, slices), 8, h, w, seed)
        native = native or h
        self.options = _Options(sliceStart=slice_start, sliceEnd=slice_end if slice_end is not None else slices, axis=axis,
                                sliceResolution=(h, w), format='raw')
        self.patients, self._vols = [], {}
        self._split = {'TRAIN': [], 'VAL': [], 'TEST': []}
        for k in range(n_val + n_test):
            name = f'synthetic_patient_{k}'
            x, lab, msk = synthetic_slices(slices, native, native, seed=seed + 100 + k, lesions=True)
            self._vols[name] = (x[..., 0].astype(np.float64), lab.astype(np.int32), msk.astype(np.int32))
            self._split['VAL' if k < n_val else 'TEST'].append(k)
            self.patients.append({'name': name, 'fullpath': name, 'filtered_files': [name + '.raw'], 'groundtruth_filename': name + '_seg.raw',
                                  'split': 'VAL' if k < n_val else 'TEST'})

    def get_patient_idx(self, split='TEST'):
        return list(self._split[split])

    def load_volume_and_groundtruth(self, nii_filename, patient):
        x, lab, msk = self._vols[patient['name']]
        return _Volume(x), _Volume(lab), _Volume(msk)

    def num_batches(self, batchsize, set='TRAIN'):
        if set in ('VAL', 'TEST') and self._split[set] and batchsize == 1:
            return sum(self._vols[self.patients[i]['name']][0].shape[0] for i in self._split[set])
        return super().num_batches(batchsize, set)
