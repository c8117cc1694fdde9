"""utils/tf_checkpoint.py: TF V2 tensor-bundle reader / writer.  No TensorFlow in this image, so the format
is pinned by (a) published crc32c known-answer vectors, (b) a HAND-ASSEMBLED index file built byte by byte from the table format's
definition (independent of write_checkpoint), (c) writer -> reader round trips over several data blocks."""
import struct

import numpy as np
import pytest

from unsupervised_anomaly_detection_brain_mri_amd.utils import tf_checkpoint as tfc


def test_crc32c_known_answers():
    assert tfc.crc32c(b'123456789') == 0xE3069283
    assert tfc.crc32c(bytes(32)) == 0x8A9136AA                    # RFC 3720 B.4
    assert tfc.crc32c(b'\xff' * 32) == 0x62A8AB43
    assert tfc.crc32c(bytes(range(32))) == 0x46DD794E
    # leveldb's mask: rotate right 15, add 0xa282ead8
    c = tfc.crc32c(b'foo')
    assert tfc._mask(c) == ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


def test_varint_and_proto():
    for v in (0, 1, 127, 128, 300, 2 ** 32, 2 ** 63 - 1):
        b = tfc._put_varint(v)
        assert tfc._get_varint(b, 0) == (v, len(b))
    # BundleEntryProto {dtype: DT_FLOAT, shape {dim {size: 5} dim {size: 3}}, offset: 256, size: 60, crc32c: 0xdeadbeef}
    raw = bytes([0x08, 0x01, 0x12, 0x08, 0x12, 0x02, 0x08, 0x05, 0x12, 0x02, 0x08, 0x03, 0x20, 0x80, 0x02, 0x28, 0x3c, 0x35]) + struct.pack('<I', 0xdeadbeef)
    e = tfc._parse_entry(raw)
    assert e == {'dtype': 1, 'shape': (5, 3), 'shard_id': 0, 'offset': 256, 'size': 60, 'crc32c': 0xdeadbeef, 'sliced': False}
    assert tfc._entry_bytes(1, (5, 3), 256, 60, 0xdeadbeef) == raw


def _hand_block(entries):
    """entries: [(shared, key_delta, value)] -> block bytes with one restart at 0."""
    body = b''
    for shared, delta, val in entries:
        body += bytes([shared, len(delta), len(val)]) + delta + val
    return body + struct.pack('<II', 0, 1)


def test_reader_on_hand_assembled_bundle(tmp_path):
    a = np.arange(6, dtype='<f4').reshape(2, 3)
    b = np.array([7.5], '<f4')
    data = a.tobytes() + b.tobytes()
    (tmp_path / 'm.data-00000-of-00001').write_bytes(data)
    ea = bytes([0x08, 0x01, 0x12, 0x08, 0x12, 0x02, 0x08, 0x02, 0x12, 0x02, 0x08, 0x03, 0x28, 24, 0x35]) + struct.pack('<I', tfc._mask(tfc.crc32c(a.tobytes())))
    eb = bytes([0x08, 0x01, 0x12, 0x04, 0x12, 0x02, 0x08, 0x01, 0x20, 24, 0x28, 4, 0x35]) + struct.pack('<I', tfc._mask(tfc.crc32c(b.tobytes())))
    header = bytes([0x08, 0x01, 0x10, 0x00, 0x1a, 0x02, 0x08, 0x01])
    # keys "", "enc/bias", "enc/kernel" with prefix compression: "enc/kernel" shares "enc/" with "enc/bias"
    blk = _hand_block([(0, b'', header), (0, b'enc/bias', eb), (4, b'kernel', ea)])
    out = blk + b'\x00' + struct.pack('<I', tfc._mask(tfc.crc32c(blk + b'\x00')))
    meta = _hand_block([])
    moff = len(out)
    out += meta + b'\x00' + struct.pack('<I', tfc._mask(tfc.crc32c(meta + b'\x00')))
    idx = _hand_block([(0, b'enc/l', bytes([0, len(blk)]))])          # a shortened separator >= last key, as leveldb emits
    ioff = len(out)
    out += idx + b'\x00' + struct.pack('<I', tfc._mask(tfc.crc32c(idx + b'\x00')))
    footer = bytes([moff, len(meta), ioff, len(idx)])
    footer += bytes(40 - len(footer)) + struct.pack('<Q', 0xdb4775248b80fb57)
    (tmp_path / 'm.index').write_bytes(out + footer)
    got = tfc.read_checkpoint(str(tmp_path / 'm'))
    assert set(got) == {'enc/bias', 'enc/kernel'}
    np.testing.assert_array_equal(got['enc/kernel'], a)
    np.testing.assert_array_equal(got['enc/bias'], b)
    # corrupt one data byte -> crc failure
    bad = bytearray(data); bad[3] ^= 1
    (tmp_path / 'm.data-00000-of-00001').write_bytes(bytes(bad))
    with pytest.raises(ValueError, match='crc32c'):
        tfc.read_checkpoint(str(tmp_path / 'm'))
    (tmp_path / 'n.index').write_bytes(b'not a table' * 8)
    with pytest.raises(ValueError, match='magic'):
        tfc.read_index(str(tmp_path / 'n.index'))


def test_round_trip_many_blocks(tmp_path):
    rng = np.random.default_rng(0)
    tensors = {f'Encoder/enc_conv2D_{i}/kernel': rng.standard_normal((5, 5, 1 + i % 3, 4)).astype(np.float32) for i in range(40)}
    tensors.update({f'Encoder/enc_conv2D_{i}/bias': rng.standard_normal(4).astype(np.float32) for i in range(40)})
    tensors['beta1_power'] = np.float32(0.5 ** 7).reshape(())
    tensors['global_step'] = np.int64(7).reshape(())
    tfc.write_checkpoint(str(tmp_path / 'ck'), tensors, entries_per_block=7)
    got = tfc.read_checkpoint(str(tmp_path / 'ck'))
    assert set(got) == set(tensors)
    for k in tensors:
        assert got[k].dtype == tensors[k].dtype and got[k].shape == tensors[k].shape
        np.testing.assert_array_equal(got[k], tensors[k])


def test_spec_mapping(tmp_path):
    rng = np.random.default_rng(1)
    spec = [('Encoder/conv/kernel', (5, 5, 1, 8), 0), ('Encoder/conv/bias', (8,), 200), ('Decoder/dense/kernel', (8, 16), 208)]
    n = 208 + 128
    params, m, v = (rng.standard_normal(n).astype(np.float32) for _ in range(3))
    tensors = tfc.flat_to_bundle(spec, params, m, np.abs(v), adam_t=11)
    tensors['Encoder/bn/moving_mean'] = np.zeros(8, np.float32)            # ignored on the way back
    tfc.write_checkpoint(str(tmp_path / 'ck'), tensors)
    got = tfc.bundle_to_flat(spec, tfc.read_checkpoint(str(tmp_path / 'ck')))
    assert got['missing'] == [] and got['adam_t'] == 11
    np.testing.assert_array_equal(got['params'], params)
    np.testing.assert_array_equal(got['adam_m'], m)
    np.testing.assert_array_equal(got['adam_v'], np.abs(v))
    # weights only: no slots -> Adam state untouched
    tfc.write_checkpoint(str(tmp_path / 'w'), tfc.flat_to_bundle(spec, params))
    got = tfc.bundle_to_flat(spec, tfc.read_checkpoint(str(tmp_path / 'w')))
    assert got['adam_m'] is None and got['adam_t'] is None
    np.testing.assert_array_equal(got['params'], params)
    # a missing variable is reported; a shape mismatch raises
    got = tfc.bundle_to_flat(spec + [('Decoder/dense/bias', (16,), n)], tfc.read_checkpoint(str(tmp_path / 'w')))
    assert got['missing'] == ['Decoder/dense/bias']
    with pytest.raises(ValueError, match='shape'):
        tfc.bundle_to_flat([('Encoder/conv/bias', (4,), 0)], tfc.read_checkpoint(str(tmp_path / 'w')))


def test_adam_step_survives_beta1_power_underflow_and_bn_statistics_are_exported(tmp_path):
    """with the reference's beta1 = 0.5 the fp32 beta1_power is 0 after 149 steps (less than an epoch), so the step count must come
    from beta2_power; and a reference-side tf.train.Saver() restores every global variable, the BatchNormalization moving statistics
    (never updated there: zeros / ones) included."""
    rng = np.random.default_rng(2)
    spec = [('Encoder/enc_conv2D_0/kernel', (5, 5, 1, 8), 0), ('Encoder/batch_normalization_0/gamma', (8,), 200),
            ('Encoder/batch_normalization_0/beta', (8,), 208), ('Decoder/batch_normalization/gamma', (4,), 216), ('Decoder/batch_normalization/beta', (4,), 220)]
    params, m, v = (rng.standard_normal(224).astype(np.float32) for _ in range(3))
    for t in (1, 148, 150, 2000, 50000):
        tensors = tfc.flat_to_bundle(spec, params, m, np.abs(v), adam_t=t)
        if t >= 150:
            assert float(tensors['beta1_power']) == 0.0          # what a long reference run leaves behind
        tfc.write_checkpoint(str(tmp_path / f'ck{t}'), tensors)
        assert tfc.bundle_to_flat(spec, tfc.read_checkpoint(str(tmp_path / f'ck{t}')))['adam_t'] == t
    np.testing.assert_array_equal(tensors['Encoder/batch_normalization_0/moving_mean'], np.zeros(8, np.float32))
    np.testing.assert_array_equal(tensors['Encoder/batch_normalization_0/moving_variance'], np.ones(8, np.float32))
    np.testing.assert_array_equal(tensors['Decoder/batch_normalization/moving_variance'], np.ones(4, np.float32))
    # a bundle with only an underflowed beta1_power gives no step (the caller keeps its own counter) instead of a wrong one
    only_b1 = {k: val for k, val in tensors.items() if k != 'beta2_power'}
    assert tfc.bundle_to_flat(spec, only_b1)['adam_t'] is None
