"""utils/tfrecord_utils.py: the reference's TFRecord slice cache read / written without TensorFlow; a hand-assembled Example (packed and
unpacked int64 lists), round trip, crc check, conversion into the plain-array slice cache."""
import struct

import numpy as np
import pytest

from unsupervised_anomaly_detection_brain_mri_amd.utils import tfrecord_utils as tr
from unsupervised_anomaly_detection_brain_mri_amd.utils.slice_cache import read_cache
from unsupervised_anomaly_detection_brain_mri_amd.utils.tf_checkpoint import _mask, crc32c


def _frame(rec):
    hdr = struct.pack('<Q', len(rec))
    return hdr + struct.pack('<I', _mask(crc32c(hdr))) + rec + struct.pack('<I', _mask(crc32c(rec)))


def test_hand_assembled_example(tmp_path):
    img = np.arange(6, dtype=np.float32).reshape(2, 3, 1)
    lab = (img > 2).astype(np.float32)

    def ld(num, b):
        return bytes([(num << 3) | 2, len(b)]) + b
    feat_bytes = lambda b: ld(1, ld(1, b))                        # Feature{bytes_list{value}}
    h_unpacked = ld(3, bytes([0x08, 2]))                          # Int64List{value: 2} written as a repeated varint (tag 0x08)
    w_packed = ld(3, ld(1, bytes([3])))                           # ... and as a packed field
    entries = b''
    for key, feat in ((b'height', h_unpacked), (b'width', w_packed), (b'image', feat_bytes(img.tobytes())), (b'label', feat_bytes(lab.tobytes())),
                      (b'set', feat_bytes(np.int32(2).tobytes()))):
        entries += ld(1, ld(1, key) + ld(2, feat))
    rec = ld(1, entries)
    (tmp_path / 'a.tfrecord').write_bytes(_frame(rec) + _frame(rec))
    images, labels, sets = tr.read_tf_record(str(tmp_path / 'a.tfrecord'))
    assert images.shape == (2, 2, 3, 1) and np.array_equal(images[1], img) and np.array_equal(labels[0], lab) and sets.tolist() == [[2], [2]]
    bad = bytearray(_frame(rec)); bad[20] ^= 1
    (tmp_path / 'b.tfrecord').write_bytes(bytes(bad))
    with pytest.raises(ValueError):
        tr.read_tf_record(str(tmp_path / 'b.tfrecord'))


def test_round_trip_and_cache_conversion(tmp_path):
    rng = np.random.default_rng(0)
    images = rng.random((7, 16, 12, 1)).astype(np.float32)
    images[:, :2] = 0
    labels = (rng.random((7, 16, 12, 1)) > 0.9).astype(np.float32)
    sets = np.array([[0], [0], [1], [2], [2], [0], [1]], np.int32)
    tr.write_tf_record(images, labels, sets, str(tmp_path / 'c.tfrecord'))
    i2, l2, s2 = tr.read_tf_record(str(tmp_path / 'c.tfrecord'))
    assert np.array_equal(i2, images) and np.array_equal(l2, labels) and np.array_equal(s2, sets)
    shape = tr.tfrecord_to_cache(str(tmp_path / 'c.tfrecord'), str(tmp_path / 'cache'))
    ci, cl, index = read_cache(str(tmp_path / 'cache'))
    assert shape == (7, 16, 12, 1) and np.array_equal(np.asarray(ci), images) and index['sets'] == sets.reshape(-1).tolist()
    assert np.array_equal(np.asarray(cl) == 10, labels[..., 0] > 0.5) and not (np.asarray(cl)[:, :2][labels[:, :2, :, 0] <= 0.5]).any()
    # a BRAINWEB-style tissue map (values up to 10) is kept as it is
    tissue = rng.integers(0, 11, (3, 8, 8, 1)).astype(np.float32)
    tr.write_tf_record(rng.random((3, 8, 8, 1)).astype(np.float32), tissue, np.zeros((3, 1), np.int32), str(tmp_path / 'd.tfrecord'))
    tr.tfrecord_to_cache(str(tmp_path / 'd.tfrecord'), str(tmp_path / 'cache2'))
    assert np.array_equal(np.asarray(read_cache(str(tmp_path / 'cache2'))[1]), tissue[..., 0].astype(np.uint8))
