"""Generates tests/golden/cursor_golden.npz: what the reference's BRAINWEB.num_batches / next_batch (dataloaders/BRAINWEB.py:406-478) RETURN,
run HERE on a small seeded slice set.  Only the two methods' text is exec'd in memory into a stand-in class (importing the module needs
TensorFlow, SimpleITK, cv2); nothing of the reference is written to disk -- the fixture holds inputs (split vector, label maps, seeds, batch
sizes) and outputs (which slices each call returned, the brain masks of one call).  Slice i of the set is the constant image i, so a
returned batch identifies the slices it is made of."""
import textwrap

import numpy
import numpy as np

src = open('/root/reference/dataloaders/BRAINWEB.py').read()
body = src[src.index('    def num_batches(self, batchsize'):src.index('    def _convert_patient_split(self)')]
LABELS = {'BACKGROUND': 0, 'CSF': 1, 'GM': 2, 'WM': 3, 'FAT': 4, 'MUSCLE': 5, 'SKIN': 6, 'SKULL': 7, 'GLIALMATTER': 8, 'CONNECTIVE': 9, 'LESION': 10}
ns = {'numpy': numpy, 'np': np}
exec('class BRAINWEB(object):\n'
     "    SET_TYPES = ['TRAIN', 'VAL', 'TEST']\n"
     f'    LABELS = {LABELS!r}\n'
     '    images = property(lambda self: self._images)\n'
     '    labels = property(lambda self: self._labels)\n'
     '    sets = property(lambda self: self._sets)\n' + textwrap.indent(textwrap.dedent(body), '    '), ns)
BRAINWEB = ns['BRAINWEB']


class _Opt:
    addInstanceNoise = False


def make(n_total, sets, seed):
    d = object.__new__(BRAINWEB)
    d.options = _Opt()
    d._images = np.arange(n_total, dtype=np.float32)[:, None, None, None] * np.ones((1, 2, 2, 1), np.float32)
    rng = np.random.default_rng(seed)
    d._labels = rng.integers(0, 11, (n_total, 2, 2)).astype(np.float32)
    d._sets = np.asarray(sets).copy()
    d._index_in_epoch = {'TRAIN': 0, 'VAL': 0, 'TEST': 0}
    d._epochs_completed = {'TRAIN': 0, 'VAL': 0, 'TEST': 0}
    return d


out = {}
for case, (n_total, n_train, n_val, bs, calls, seed, shuffle) in enumerate([(23, 13, 6, 4, 12, 1, True), (16, 16, 0, 8, 5, 2, True),
                                                                             (20, 11, 9, 3, 9, 3, False), (9, 5, 4, 5, 4, 4, True)]):
    sets = np.array([0] * n_train + [1] * n_val + [2] * (n_total - n_train - n_val))
    np.random.default_rng(100 + case).shuffle(sets)          # the splits are interleaved in the cache
    d = make(n_total, sets, seed)
    labels0 = d._labels.copy()
    numpy.random.seed(seed)                                   # next_batch shuffles with the global numpy RNG
    seq, vseq = [], []
    for c in range(calls):
        im, lab, bm = d.next_batch(bs, shuffle=shuffle, set='TRAIN', return_brainmask=(c == 1))
        seq.append(im[:, 0, 0, 0].astype(np.int64))
        assert np.array_equal(lab, labels0[seq[-1]])         # labels travel with their images
        if c == 1:
            out[f'bm{case}'] = bm
            out[f'bm_labels{case}'] = lab
        if n_val and c % 2 == 0:
            vseq.append(d.next_batch(2, shuffle=shuffle, set='VAL')[0][:, 0, 0, 0].astype(np.int64))
    out[f'sets{case}'] = sets
    out[f'cfg{case}'] = np.array([bs, calls, seed, int(shuffle), d.num_batches(bs, 'TRAIN'), d.num_batches(2, 'VAL')])
    out[f'train{case}'] = np.stack(seq)
    out[f'val{case}'] = np.stack(vseq) if vseq else np.zeros((0, 2), np.int64)
np.savez_compressed('tests/golden/cursor_golden.npz', **out)
print({k: v.shape for k, v in out.items()})
print(out['train0'][:6])
