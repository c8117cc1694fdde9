"""Slice cache + device-resident dataset.

On disk (replaces the reference's TFRecord + pickle cache, dataloaders/BRAINWEB.py:59-100, utils/tfrecord_utils.py):
    <dir>/slices.f32   raw little-endian fp32, NHWC [N, H, W, C], values in [0, 1]
    <dir>/labels.u8    uint8 label maps [N, H, W] (BRAINWEB.LABELS values), optional
    <dir>/index.json   {"version": 1, "shape": [N,H,W,C], "sets": [per-slice 0|1|2 = TRAIN|VAL|TEST], "patients": [...],
                        "options": {...}}
Both payload files are plain arrays: `np.memmap` reads them without parsing and one `hipMemcpy` puts them in HBM.

In memory: `DeviceDataset` keeps the whole slice set in device memory (a Brainweb-size set is a few GB of the 288 GB) and serves
the reference's dataset duck-type (`num_batches`, `next_batch`) with DEVICE tensors: a batch is one gather kernel over an index
vector (`uad_gather_slices` / `uad_gather_mask`), so a training step has no host-to-device copy.  The cursor / epoch-wrap /
shuffle behaviour is `dataloaders/BRAINWEB.py:411-457` restated, including its quirk that the first epoch is never shuffled
(`:419` compares the epoch dict with 0)."""
import ctypes as C
import json
import os

import numpy as np

SET_TYPES = ['TRAIN', 'VAL', 'TEST']                               # dataloaders/BRAINWEB.py:24
LABELS = {'BACKGROUND': 0, 'CSF': 1, 'GM': 2, 'WM': 3, 'FAT': 4, 'MUSCLE': 5, 'SKIN': 6, 'SKULL': 7, 'GLIALMATTER': 8,
          'CONNECTIVE': 9, 'LESION': 10}                           # :25
NON_BRAIN = ('FAT', 'MUSCLE', 'SKIN', 'SKULL', 'CONNECTIVE', 'BACKGROUND')   # zeroed by next_batch(return_brainmask=True), :466-476


def brainmask_lut():
    lut = np.ones(256, np.uint8)
    for k in NON_BRAIN:
        lut[LABELS[k]] = 0
    return lut


def write_cache(directory, images, sets, labels=None, patients=None, options=None):
    """images [N,H,W,C] float in [0,1]; sets [N] in {0,1,2}; labels [N,H,W] integer maps (optional)."""
    images = np.ascontiguousarray(images, '<f4')
    if images.ndim != 4:
        raise ValueError('images must be [N,H,W,C]')
    sets = np.asarray(sets, np.int64)
    if sets.shape != (images.shape[0],) or sets.min() < 0 or sets.max() > 2:
        raise ValueError('sets must hold one of 0|1|2 per slice')
    os.makedirs(directory, exist_ok=True)
    images.tofile(os.path.join(directory, 'slices.f32'))
    if labels is not None:
        labels = np.ascontiguousarray(labels, np.uint8)
        if labels.shape != images.shape[:3]:
            raise ValueError('labels must be [N,H,W]')
        labels.tofile(os.path.join(directory, 'labels.u8'))
    with open(os.path.join(directory, 'index.json'), 'w') as f:
        json.dump({'version': 1, 'shape': list(images.shape), 'sets': sets.tolist(), 'has_labels': labels is not None,
                   'patients': list(patients) if patients is not None else [], 'options': options or {}}, f)


def read_cache(directory):
    """-> (images memmap [N,H,W,C] fp32, labels memmap [N,H,W] u8 or None, index dict)."""
    with open(os.path.join(directory, 'index.json')) as f:
        index = json.load(f)
    if index.get('version') != 1:
        raise ValueError(f'unsupported slice-cache version {index.get("version")}')
    shape = tuple(index['shape'])
    images = np.memmap(os.path.join(directory, 'slices.f32'), '<f4', 'r', shape=shape)
    labels = None
    if index.get('has_labels'):
        labels = np.memmap(os.path.join(directory, 'labels.u8'), np.uint8, 'r', shape=shape[:3])
    return images, labels, index


def volumes_from_cache(directory, set_name='TEST'):
    """Per-patient volumes of one split for utils/Evaluation.evaluate (the reference evaluates patient by patient, Evaluation.py:205-262):
    -> (volumes [z,H,W] float64 list, lesion maps [z,H,W] {0,1} list, brain masks [z,H,W] {0,1} list, patient names).  Slices keep their cache
    order inside a patient; lesion = label LESION, brain = every label the brain-mask LUT keeps."""
    images, labels, index = read_cache(directory)
    if labels is None:
        raise ValueError('the cache holds no label maps: nothing to evaluate against')
    sets = np.asarray(index['sets'])
    pats = index.get('patients') or []
    if len(pats) != len(sets):
        raise ValueError('the cache does not record which patient a slice belongs to')
    sel = np.where(sets == SET_TYPES.index(set_name))[0]
    lut = brainmask_lut()
    vols, labs, masks, names = [], [], [], []
    for name in dict.fromkeys(pats[i] for i in sel):          # first-seen order
        idx = [i for i in sel if pats[i] == name]
        lab = np.asarray(labels[idx])
        vols.append(np.asarray(images[idx])[..., 0].astype(np.float64))
        labs.append((lab == LABELS['LESION']).astype(np.float64))
        masks.append(lut[lab].astype(np.float64))
        names.append(name)
    return vols, labs, masks, names


class BatchCursor:
    """The index arithmetic of BRAINWEB.next_batch (:411-457) for one split, on index vectors instead of image arrays: returns the
    positions (into the split's slice list) of the next batch."""

    def __init__(self, n_samples, rng):
        self.n, self.rng = int(n_samples), rng
        self.order = np.arange(self.n)            # current arrangement of the split (the reference permutes the arrays in place)
        self.index_in_epoch = 0
        self.epochs_completed = 0

    def next(self, batch_size, shuffle=True):
        start = self.index_in_epoch
        # (:419 `self._epochs_completed == 0` compares a dict with 0 -> never true: no shuffle before the first epoch)
        if start + batch_size > self.n:
            self.epochs_completed += 1
            rest = self.order[start:self.n].copy()
            if shuffle:
                # self._images[set] = self.images[set[perm]] with `images` the property of `_images` (:312): the CURRENT arrangement is permuted
                self.order = self.order[self.rng.permutation(self.n)]
            self.index_in_epoch = batch_size - len(rest)
            return np.concatenate([rest, self.order[:self.index_in_epoch]])
        self.index_in_epoch += batch_size
        return self.order[start:self.index_in_epoch].copy()


class DeviceDataset:
    """Dataset duck-type over an HBM-resident slice cache; batches are device tensors."""
    SET_TYPES = SET_TYPES

    def __init__(self, images, sets, labels=None, seed=0, device=None, patients=None):
        import torch
        from .. import _lib
        self._torch, self._lib = torch, _lib
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError('DeviceDataset needs a ROCm GPU; there is no CPU fallback (use the host dataset classes for CPU work)')
        self.device = torch.device(device if device is not None else f'cuda:{torch.cuda.current_device()}')
        images = np.asarray(images)
        if images.ndim != 4 or (images.shape[1] * images.shape[2] * images.shape[3]) % 4:
            raise ValueError('images must be [N,H,W,C] with H*W*C a multiple of 4')
        self.shape = tuple(images.shape)
        self.num_channels = self.shape[3]
        self.patients = list(patients) if patients is not None else []
        self._images = torch.from_numpy(np.array(images, np.float32, order='C')).to(self.device)      # one copy out of the memmap, then H2D
        self._labels = None if labels is None else torch.from_numpy(np.array(labels, np.uint8, order='C')).to(self.device)
        self._lut = torch.from_numpy(brainmask_lut()).to(self.device)
        sets = np.asarray(sets)
        self._set_idx = {name: np.where(sets == k)[0].astype(np.int32) for k, name in enumerate(SET_TYPES)}
        rng = np.random.default_rng(seed)
        self._cursor = {name: BatchCursor(len(ix), rng) for name, ix in self._set_idx.items()}
        # batch indices travel host -> device through a ring of pinned staging buffers with non-blocking copies: `.to(device)` of a pageable
        # array synchronises the stream, i.e. the host could never run ahead of the GPU and every step paid its launch latency again
        self._ring, self._ring_n = [], 64
        self._ring_i = 0

    @classmethod
    def from_cache(cls, directory, **kw):
        images, labels, index = read_cache(directory)
        return cls(images, index['sets'], labels, patients=index.get('patients'), **kw)

    def num_batches(self, batchsize, set='TRAIN'):
        return len(self._set_idx[set]) // batchsize

    def _stream(self):
        return C.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def _indices_to_device(self, arr):
        torch = self._torch
        arr = np.ascontiguousarray(arr, np.int32)
        n = int(arr.size)
        k = self._ring_i % self._ring_n
        self._ring_i += 1
        if len(self._ring) <= k or self._ring[k][0].numel() < n:
            slot = (torch.empty(max(n, 256), dtype=torch.int32).pin_memory(), torch.empty(max(n, 256), dtype=torch.int32, device=self.device),
                    torch.cuda.Event())
            if len(self._ring) <= k:
                self._ring.append(slot)
            else:
                self._ring[k][2].synchronize()
                self._ring[k] = slot
        else:
            self._ring[k][2].synchronize()          # the copy that used this slot ring_n batches ago has long completed
        pin, dev, ev = self._ring[k]
        pin[:n].copy_(torch.from_numpy(arr))
        dev[:n].copy_(pin[:n], non_blocking=True)
        ev.record(torch.cuda.current_stream(self.device))
        return dev[:n]

    def next_batch(self, batch_size, shuffle=True, set='TRAIN', return_brainmask=False):
        torch, _lib = self._torch, self._lib
        pos = self._cursor[set].next(batch_size, shuffle)
        assert pos.size, "The batch is empty!"
        idx = self._indices_to_device(self._set_idx[set][pos])
        n = int(idx.numel())
        N, H, W, Cc = self.shape
        out = torch.empty((n, H, W, Cc), device=self.device, dtype=torch.float32)
        _lib.check(self.lib.uad_gather_slices(C.c_void_p(self._images.data_ptr()), C.c_void_p(idx.data_ptr()), n, H * W * Cc,
                                              C.c_void_p(out.data_ptr()), self._stream()))
        labels = masks = None
        if self._labels is not None:
            labels = torch.empty((n, H, W), device=self.device, dtype=torch.float32)
            _lib.check(self.lib.uad_gather_mask(C.c_void_p(self._labels.data_ptr()), C.c_void_p(idx.data_ptr()), n, H * W, None,
                                                C.c_void_p(labels.data_ptr()), self._stream()))
            if return_brainmask:
                masks = torch.empty((n, H, W), device=self.device, dtype=torch.float32)
                _lib.check(self.lib.uad_gather_mask(C.c_void_p(self._labels.data_ptr()), C.c_void_p(idx.data_ptr()), n, H * W,
                                                    C.c_void_p(self._lut.data_ptr()), C.c_void_p(masks.data_ptr()), self._stream()))
        self._keep = idx
        return out, labels, masks
