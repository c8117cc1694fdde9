"""utils/tfrecord_utils.py:14-52 without TensorFlow: the reference's slice cache (`<name>.tfrecord` written by the dataset classes when
options.cache is set, e.g. dataloaders/MSLUB.py:103-112) is a TFRecord file of tf.train.Example messages with the features
  height, width (int64), image, label (raw float32 bytes, [H, W, C]), set (raw int32 bytes: 0 | 1 | 2).
`read_tf_record` returns what the reference's function returns; `write_tf_record` writes the same layout (hand-back path);
`tfrecord_to_cache` converts such a file into the plain-array slice cache of utils/slice_cache.py.

Format restated (UNPINNED, no TensorFlow here): TFRecord framing (uint64 length, masked crc32c, payload, masked crc32c -- shared with
utils/logger.py); Example {1: Features {1: map<string, Feature> entries {1: key, 2: Feature {1: BytesList {1: bytes*} | 2: FloatList |
3: Int64List {1: packed or repeated varint}}}}}."""
import struct

import numpy as np

from .tf_checkpoint import _get_varint, _mask, _parse_message, _put_varint, crc32c


def _records(path):
    data = open(path, 'rb').read()
    pos = 0
    while pos < len(data):
        (ln,) = struct.unpack_from('<Q', data, pos)
        if struct.unpack_from('<I', data, pos + 8)[0] != _mask(crc32c(data[pos:pos + 8])):
            raise ValueError(f'{path}: bad record-length crc at byte {pos}')
        rec = data[pos + 12:pos + 12 + ln]
        if len(rec) != ln or struct.unpack_from('<I', data, pos + 12 + ln)[0] != _mask(crc32c(rec)):
            raise ValueError(f'{path}: bad record crc at byte {pos}')
        pos += 16 + ln
        yield rec


def _int64_list(buf):
    """Int64List.value: packed (one length-delimited field) or repeated varints."""
    m = _parse_message(buf)
    out = []
    for v in m.get(1, []):
        if isinstance(v, bytes):
            p = 0
            while p < len(v):
                x, p = _get_varint(v, p)
                out.append(x)
        else:
            out.append(v)
    return [x - (1 << 64) if x >> 63 else x for x in out]


def parse_example(rec):
    """-> {feature name: [bytes] | [int]} (bytes_list and int64_list features)."""
    out = {}
    for feats in _parse_message(rec).get(1, []):
        for entry in _parse_message(feats).get(1, []):
            e = _parse_message(entry)
            key = e[1][0].decode()
            f = _parse_message(e[2][0]) if 2 in e else {}
            if 1 in f:
                out[key] = list(_parse_message(f[1][0]).get(1, []))
            elif 3 in f:
                out[key] = _int64_list(f[3][0])
    return out


def read_tf_record(filename):
    """(images [N,H,W,C] float32, labels [N,H,W,C] float32, sets [N,1] int32) -- the reference's return value (:36-52)."""
    images, labels, sets = [], [], []
    for rec in _records(filename):
        ex = parse_example(rec)
        h, w = int(ex['height'][0]), int(ex['width'][0])
        images.append(np.frombuffer(ex['image'][0], np.float32).reshape(h, w, -1))
        labels.append(np.frombuffer(ex['label'][0], np.float32).reshape(h, w, -1))
        sets.append(np.frombuffer(ex['set'][0], np.int32))
    return np.array(images), np.array(labels), np.array(sets)


def _field(num, wire, payload):
    return _put_varint((num << 3) | wire) + payload


def _ld(num, b):
    return _field(num, 2, _put_varint(len(b)) + b)


def write_tf_record(images, labels, sets, filename):
    """The layout of the reference's writer (:14-33)."""
    with open(filename, 'wb') as f:
        for img, lab, st in zip(images, labels, sets):
            img = np.ascontiguousarray(img, np.float32)
            feats = b''
            for key, feat in (('height', _ld(3, _ld(1, _put_varint(img.shape[0])))), ('width', _ld(3, _ld(1, _put_varint(img.shape[1])))),
                              ('image', _ld(1, _ld(1, img.tobytes()))), ('label', _ld(1, _ld(1, np.ascontiguousarray(lab, np.float32).tobytes()))),
                              ('set', _ld(1, _ld(1, np.asarray(st, np.int32).tobytes())))):
                feats += _ld(1, _ld(1, key.encode()) + _ld(2, feat))
            rec = _ld(1, feats)
            hdr = struct.pack('<Q', len(rec))
            f.write(hdr + struct.pack('<I', _mask(crc32c(hdr))) + rec + struct.pack('<I', _mask(crc32c(rec))))


def tfrecord_to_cache(filename, directory, lesion_threshold=0.5):
    """Converts a reference slice cache into utils/slice_cache.py's plain arrays.  The reference's label maps are float images: BRAINWEB's hold
    the tissue classes (BRAINWEB.LABELS values), the MS sets' a {0, 1} lesion mask -- a map whose maximum is <= 1 is taken as the latter and
    stored as LESION (10) / GM (2, any other non-zero image pixel) / BACKGROUND."""
    from .slice_cache import write_cache
    images, labels, sets = read_tf_record(filename)
    lab = labels[..., 0]
    if lab.max() <= 1.0:
        lab_u8 = np.where(lab > lesion_threshold, 10, np.where(images[..., 0] > 0, 2, 0)).astype(np.uint8)
    else:
        lab_u8 = np.rint(lab).astype(np.uint8)
    write_cache(directory, images[..., :1] if images.shape[-1] != 1 else images, sets.reshape(-1), lab_u8, options={'source': str(filename)})
    return images.shape
