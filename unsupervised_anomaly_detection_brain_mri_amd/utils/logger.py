"""utils/logger.py -- the reference's TensorBoard `Logger` (utils/logger.py:14-64: one FileWriter per phase under <summary_dir>/{TRAIN,VAL,TEST},
`summarize(step, phase, summaries_dict)` writing a scalar summary for values of rank <= 1 and an image summary otherwise) without TensorFlow:
event files are written directly in the tfevents format TensorBoard reads.

File format restated (UNPINNED: no TensorBoard in this image; the reader below round-trips the writer and the PNG payloads decode with zlib):
  records   uint64 length | masked crc32c(length) | payload | masked crc32c(payload)          (TFRecord framing, crc as in tf_checkpoint.py)
  payload   tensorflow.Event  {1: wall_time double, 2: step int64, 3: file_version string | 5: summary}
  Summary   {1: value* {1: tag, 2: simple_value float | 4: image {1: height, 2: width, 3: colorspace, 4: encoded_image_string (PNG)}}}
Image tags follow tf.summary.image: `<tag>/image/<i>` (`<tag>/image` when a single image is written), pixel scaling of a float image to
[0, 255] as tf.summary.image does (all-non-negative: * 255 / max; otherwise shift / scale so that 0 maps to 127.5)."""
import os
import socket
import struct
import time
import zlib
from enum import Enum

import numpy as np

from .tf_checkpoint import _get_varint, _mask, _parse_message, _put_varint, crc32c


class Phase(Enum):
    TRAIN = 'TRAIN'
    VAL = 'VAL'
    TEST = 'TEST'


def _field(num, wire, payload):
    return _put_varint((num << 3) | wire) + payload


def _bytes_field(num, b):
    return _field(num, 2, _put_varint(len(b)) + b)


def encode_png(img_u8):
    """8-bit grayscale [H,W] or RGB [H,W,3] -> PNG bytes (zlib + crc32 only)."""
    a = np.ascontiguousarray(img_u8, np.uint8)
    if a.ndim == 3 and a.shape[2] == 1:
        a = a[..., 0]
    h, w = a.shape[:2]
    ctype = 0 if a.ndim == 2 else 2
    raw = b''.join(b'\x00' + a[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack('>I', len(data)) + tag + data + struct.pack('>I', zlib.crc32(tag + data) & 0xFFFFFFFF)
    return b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, ctype, 0, 0, 0)) + chunk(b'IDAT', zlib.compress(raw, 6)) + chunk(b'IEND', b'')


def _to_u8(img):
    """tf.summary.image's float -> uint8 rule."""
    a = np.asarray(img, np.float64)
    lo, hi = a.min(), a.max()
    if lo >= 0:
        scale, offset = (255.0 / hi if hi > 0 else 1.0), 0.0
    else:
        scale = 127.0 / max(abs(lo), abs(hi)) if max(abs(lo), abs(hi)) > 0 else 1.0
        offset = 128.0
    return np.clip(a * scale + offset, 0, 255).astype(np.uint8)


class EventFileWriter:
    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, 'events.out.tfevents.%010d.%s' % (int(time.time()), socket.gethostname()))
        self._f = open(self.path, 'wb')
        self._write_event(_bytes_field(3, b'brain.Event:2'), step=0)

    def _write_event(self, body, step):
        ev = _field(1, 1, struct.pack('<d', time.time())) + _field(2, 0, _put_varint(int(step))) + body
        hdr = struct.pack('<Q', len(ev))
        self._f.write(hdr + struct.pack('<I', _mask(crc32c(hdr))) + ev + struct.pack('<I', _mask(crc32c(ev))))

    def add_summary(self, values, step):
        """values: [(tag, float) | (tag, uint8 image array)]"""
        body = b''
        for tag, v in values:
            val = _bytes_field(1, tag.encode())
            if isinstance(v, np.ndarray):
                img = _field(1, 0, _put_varint(v.shape[0])) + _field(2, 0, _put_varint(v.shape[1])) + \
                    _field(3, 0, _put_varint(1 if v.ndim == 2 or v.shape[2] == 1 else 3)) + _bytes_field(4, encode_png(v))
                val += _bytes_field(4, img)
            else:
                val += _field(2, 5, struct.pack('<f', float(v)))
            body += _bytes_field(1, val)
        self._write_event(_bytes_field(5, body), step)

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.close()


def read_events(path):
    """-> [(step, {tag: float | ('image', height, width, png bytes)})] of an event file (tests; verifies both crcs)."""
    data = open(path, 'rb').read()
    pos, out = 0, []
    while pos < len(data):
        (ln,) = struct.unpack_from('<Q', data, pos)
        if struct.unpack_from('<I', data, pos + 8)[0] != _mask(crc32c(data[pos:pos + 8])):
            raise ValueError('bad length crc')
        ev = data[pos + 12:pos + 12 + ln]
        if struct.unpack_from('<I', data, pos + 12 + ln)[0] != _mask(crc32c(ev)):
            raise ValueError('bad payload crc')
        pos += 16 + ln
        m = _parse_message(ev)
        step = m.get(2, [0])[0]
        vals = {}
        for sm in m.get(5, []):
            for vb in _parse_message(sm).get(1, []):
                v = _parse_message(vb)
                tag = v[1][0].decode()
                if 2 in v:
                    vals[tag] = struct.unpack('<f', struct.pack('<I', v[2][0]))[0]
                elif 4 in v:
                    im = _parse_message(v[4][0])
                    vals[tag] = ('image', im[1][0], im[2][0], im[4][0])
        out.append((step, vals))
    return out


class Logger:
    """utils/logger.py:14-64.  `sess` is accepted and ignored."""

    def __init__(self, sess, summary_dir):
        self.writers = {p: EventFileWriter(os.path.join(summary_dir, p.value)) for p in Phase}

    def summarize(self, step, phase=Phase.TRAIN, scope='', summaries_dict=None):
        phase = phase if isinstance(phase, Phase) else Phase(getattr(phase, 'value', phase))
        if phase not in self.writers:
            raise ValueError(f'Illegal Argument for summarizer: {phase}')
        if summaries_dict is None:
            return
        values = []
        prefix = scope + '/' if scope else ''
        for tag, value in summaries_dict.items():
            if value is None:
                continue
            value = np.asarray(value)
            if value.ndim <= 1:                                  # scalar summary (logger.py:50-52)
                values.append((prefix + tag, float(np.mean(value))))
            else:                                                # image summary, max_outputs 100 (:53-55)
                imgs = value if value.ndim >= 3 else value[None]
                imgs = imgs[:100]
                for i, im in enumerate(imgs):
                    name = f'{prefix}{tag}/image' + ('' if len(imgs) == 1 else f'/{i}')
                    values.append((name, _to_u8(im)))
        w = self.writers[phase]
        w.add_summary(values, step)
        w.flush()

    def close(self):
        for w in self.writers.values():
            w.close()
