"""TensorFlow V2 checkpoint ("tensor bundle") interop without TensorFlow: read the `<prefix>.index` +
`<prefix>.data-00000-of-00001` pair that `tf.train.Saver().save` writes in the reference (trainers/DLMODEL.py:63-83) and map its
variables onto an engine's flat parameter / Adam buffers, so that weights trained with the reference can be scored on this path.

Formats restated (UNPINNED: there is no TensorFlow in this image to produce a real file; the writer below exists so that the reader
has a byte-exact round-trip test and so that checkpoints can be handed back):
  index file   = an SSTable in LevelDB's table format (tensorflow/core/lib/io/table*): data blocks of prefix-compressed
                 (shared, non_shared, value_len varint32; key delta; value) entries + uint32 restart offsets + uint32 count; each block
                 is followed by a 5-byte trailer (compression type, masked crc32c); index block of BlockHandles; 48-byte footer
                 (metaindex handle, index handle, padding, magic 0xdb4775248b80fb57).
  key ""       -> BundleHeaderProto {1: num_shards, 2: endianness, 3: version}
  key <name>   -> BundleEntryProto {1: dtype, 2: shape {2: dim {1: size}}, 3: shard_id, 4: offset, 5: size, 6: crc32c (fixed32)}
  data file    = the tensors' raw little-endian bytes at [offset, offset + size)
Variable names are the graph's (`Encoder/enc_conv2D_0/kernel`, ...): exactly the names of `Engine.spec`.  Adam slots are
`<name>/Adam` (m), `<name>/Adam_1` (v); the step follows from `beta1_power` = beta1^t."""
import math
import os
import struct

import numpy as np

_MAGIC = 0xdb4775248b80fb57
_DT = {1: np.dtype('<f4'), 2: np.dtype('<f8'), 3: np.dtype('<i4'), 9: np.dtype('<i8'), 10: np.dtype('bool')}     # DT_FLOAT, DOUBLE, INT32, INT64, BOOL
_DT_INV = {np.dtype('<f4'): 1, np.dtype('<f8'): 2, np.dtype('<i4'): 3, np.dtype('<i8'): 9}


# ------------------------------------------------------------------ crc32c (Castagnoli), masked like leveldb / TF
def _crc_table():
    tab = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tab.append(c)
    return tab


_CRC = _crc_table()


def crc32c(data, crc=0):
    crc ^= 0xFFFFFFFF
    for b in bytes(data):
        crc = _CRC[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def _mask(crc):
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


# ------------------------------------------------------------------ varints / protobuf wire format (the two tiny messages only)
def _get_varint(buf, pos):
    res, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        res |= (b & 0x7F) << shift
        if not b & 0x80:
            return res, pos
        shift += 7


def _put_varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _parse_message(buf):
    """-> {field: [values]}; values are ints (varint / fixed) or bytes (length-delimited)."""
    out, pos = {}, 0
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]; pos += 8
        elif wt == 2:
            ln, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + ln]); pos += ln
        elif wt == 5:
            v = struct.unpack_from('<I', buf, pos)[0]; pos += 4
        else:
            raise ValueError(f'unsupported protobuf wire type {wt}')
        out.setdefault(field, []).append(v)
    return out


def _parse_entry(buf):
    m = _parse_message(buf)
    shape = []
    for sh in m.get(2, []):
        for dim in _parse_message(sh).get(2, []):
            size = _parse_message(dim).get(1, [0])[0]
            shape.append(size - (1 << 64) if size >> 63 else size)
    return {'dtype': m.get(1, [0])[0], 'shape': tuple(shape), 'shard_id': m.get(3, [0])[0], 'offset': m.get(4, [0])[0],
            'size': m.get(5, [0])[0], 'crc32c': m.get(6, [None])[0], 'sliced': 7 in m}


def _entry_bytes(dtype, shape, offset, size, crc):
    dims = b''.join(b'\x12' + _put_varint(len(d)) + d for d in (b'\x08' + _put_varint(s) for s in shape))
    msg = b'\x08' + _put_varint(dtype) + b'\x12' + _put_varint(len(dims)) + dims
    if offset:
        msg += b'\x20' + _put_varint(offset)
    msg += b'\x28' + _put_varint(size) + b'\x35' + struct.pack('<I', crc)
    return msg


# ------------------------------------------------------------------ SSTable
def _read_block(data, offset, size):
    block = data[offset:offset + size]
    ctype = data[offset + size]
    if ctype != 0:
        raise ValueError('compressed index blocks are not supported (TF writes tensor-bundle indices uncompressed)')
    n_restarts = struct.unpack_from('<I', block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    pos, key, out = 0, b'', []
    while pos < end:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared]); pos += non_shared
        out.append((key, bytes(block[pos:pos + vlen]))); pos += vlen
    return out


def read_index(path):
    """-> (header dict, {name: entry dict}) of a `<prefix>.index` file."""
    data = open(path, 'rb').read()
    if len(data) < 48 or struct.unpack_from('<Q', data, len(data) - 8)[0] != _MAGIC:
        raise ValueError(f'{path} is not a TensorFlow checkpoint index (bad table magic)')
    footer = data[-48:]
    _, pos = _get_varint(footer, 0)            # metaindex offset
    _, pos = _get_varint(footer, pos)          # metaindex size
    ioff, pos = _get_varint(footer, pos)
    isize, pos = _get_varint(footer, pos)
    entries, header = {}, None
    for _, handle in _read_block(data, ioff, isize):
        boff, p = _get_varint(handle, 0)
        bsize, p = _get_varint(handle, p)
        for key, val in _read_block(data, boff, bsize):
            if key == b'':
                h = _parse_message(val)
                header = {'num_shards': h.get(1, [1])[0], 'endianness': h.get(2, [0])[0]}
            else:
                entries[key.decode()] = _parse_entry(val)
    if header is None:
        raise ValueError('checkpoint index has no header entry')
    if header['endianness'] != 0:
        raise ValueError('big-endian checkpoints are not supported')
    return header, entries


def read_checkpoint(prefix, verify_crc=True):
    """-> {variable name: numpy array} for every (unsliced, numeric) tensor of the bundle `<prefix>.index` / `.data-*`."""
    header, entries = read_index(prefix + '.index')
    shards = {}
    out = {}
    for name, e in entries.items():
        if e['sliced'] or e['dtype'] not in _DT:
            continue
        sid = e['shard_id']
        if sid not in shards:
            shards[sid] = open('%s.data-%05d-of-%05d' % (prefix, sid, header['num_shards']), 'rb').read()
        raw = shards[sid][e['offset']:e['offset'] + e['size']]
        if len(raw) != e['size']:
            raise ValueError(f'{name}: data shard is truncated')
        if verify_crc and e['crc32c'] is not None and _mask(crc32c(raw)) != e['crc32c']:
            raise ValueError(f'{name}: crc32c mismatch')
        out[name] = np.frombuffer(raw, _DT[e['dtype']]).reshape(e['shape']).copy()
    return out


def write_checkpoint(prefix, tensors, entries_per_block=64):
    """Writes {name: array} as a one-shard bundle (uncompressed data blocks of `entries_per_block` keys) -- the inverse of
    read_checkpoint."""
    names = sorted(tensors)
    data, recs = bytearray(), []
    for name in names:
        a = np.asarray(tensors[name])
        if a.dtype not in _DT_INV:
            a = a.astype('<f4')
        raw = a.tobytes()
        recs.append((name.encode(), _entry_bytes(_DT_INV[a.dtype], a.shape, len(data), len(raw), _mask(crc32c(raw)))))
        data += raw
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        f.write(bytes(data))
    header = b'\x08\x01' + b'\x1a\x02\x08\x01'            # num_shards = 1, version {producer = 1}
    items = [(b'', header)] + recs

    def block(kvs):
        body, restarts, prev = bytearray(), [], b''
        for i, (k, v) in enumerate(kvs):
            shared = 0
            if i % 16 == 0:
                restarts.append(len(body))
            else:
                while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                    shared += 1
            body += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
            prev = k
        body += b''.join(struct.pack('<I', r) for r in restarts) + struct.pack('<I', len(restarts))
        return bytes(body)

    out = bytearray()

    def emit(b):
        off = len(out)
        out.extend(b + b'\x00' + struct.pack('<I', _mask(crc32c(b + b'\x00'))))
        return off, len(b)

    handles = []
    for i in range(0, len(items), entries_per_block):
        chunk = items[i:i + entries_per_block]
        doff, dsize = emit(block(chunk))
        handles.append((chunk[-1][0], _put_varint(doff) + _put_varint(dsize)))          # separator key = the block's last key
    moff, msize = emit(block([]))
    ioff, isize = emit(block(handles))
    footer = _put_varint(moff) + _put_varint(msize) + _put_varint(ioff) + _put_varint(isize)
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', _MAGIC)
    with open(prefix + '.index', 'wb') as f:
        f.write(bytes(out) + footer)


# ------------------------------------------------------------------ spec (flat buffers) <-> bundle
def bundle_to_flat(spec, tensors, beta1=0.5, beta2=0.999):
    """spec: [(name, shape, offset)] of an engine.  -> dict(params, adam_m, adam_v, adam_t, missing): flat fp32 buffers filled from
    the bundle's variables; adam_* are None unless every variable has its `<name>/Adam` (m) and `<name>/Adam_1` (v) slots.  adam_t is
    recovered from AdamOptimizer's non-slot variables: beta2_power = beta2^t first (0.999^t stays a normal fp32 up to t ~ 8.7e4), beta1_power
    = beta1^t as the fallback -- with the reference's beta1 = 0.5 that one underflows to 0 after 149 steps, less than one epoch, and a
    resume from it would restart Adam's bias correction at t = 1.  Tensors the spec does not name (BN moving averages, which the
    reference never updates, global_step, ...) are ignored."""
    total = max(off + int(np.prod(shape)) for _, shape, off in spec)
    params, m, v = (np.zeros(total, np.float32) for _ in range(3))
    missing, slots = [], True
    for name, shape, off in spec:
        cnt = int(np.prod(shape))
        if name not in tensors:
            missing.append(name)
            continue
        t = np.asarray(tensors[name])
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f'{name}: checkpoint shape {tuple(t.shape)} != engine shape {tuple(shape)}')
        params[off:off + cnt] = t.astype(np.float32).reshape(-1)
        if name + '/Adam' in tensors and name + '/Adam_1' in tensors:
            m[off:off + cnt] = np.asarray(tensors[name + '/Adam'], np.float32).reshape(-1)
            v[off:off + cnt] = np.asarray(tensors[name + '/Adam_1'], np.float32).reshape(-1)
        else:
            slots = False
    t = None
    for key, beta in (('beta2_power', beta2), ('beta1_power', beta1)):
        if key in tensors:
            bp = float(np.asarray(tensors[key]).reshape(-1)[0])
            if bp == 1.0:
                t = 0
                break
            # fp32 normal range only: a denormal / zero power carries no usable step count
            if 1.2e-38 < bp < 1.0:
                t = int(round(math.log(bp) / math.log(beta)))
                break
    ok = slots and not missing
    return {'params': params, 'adam_m': m if ok else None, 'adam_v': v if ok else None, 'adam_t': t if ok else None, 'missing': missing}


def flat_to_bundle(spec, params, adam_m=None, adam_v=None, adam_t=0, beta1=0.5, beta2=0.999):
    """The variables a tf.train.Saver() built on the reference's graph expects (DLMODEL.py:63-83 saves ALL global variables): the spec's
    trainables, every BatchNormalization scope's moving_mean (zeros) / moving_variance (ones) -- the reference never updates them
    (SURVEY.md A1), so their initial values are what any of its checkpoints holds -- and, when slots are given, the Adam slots and
    beta powers.  UNVERIFIED against a real TensorFlow restore (none in the image)."""
    tensors = {}
    for name, shape, off in spec:
        cnt = int(np.prod(shape))
        tensors[name] = params[off:off + cnt].reshape(shape)
        if name.endswith('/gamma') and 'batch_normalization' in name.rsplit('/', 2)[-2]:
            scope = name[:-len('/gamma')]
            tensors[scope + '/moving_mean'] = np.zeros(shape, np.float32)
            tensors[scope + '/moving_variance'] = np.ones(shape, np.float32)
        if adam_m is not None and adam_v is not None:
            tensors[name + '/Adam'] = adam_m[off:off + cnt].reshape(shape)
            tensors[name + '/Adam_1'] = adam_v[off:off + cnt].reshape(shape)
    if adam_m is not None:
        tensors['beta1_power'] = np.float32(beta1 ** int(adam_t)).reshape(())
        tensors['beta2_power'] = np.float32(beta2 ** int(adam_t)).reshape(())
    return tensors
