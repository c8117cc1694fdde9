"""utils/logger.py: TensorBoard event files without TensorFlow -- TFRecord framing with both masked crc32c checks, Event / Summary protobufs,
scalar and PNG image summaries (decoded back with zlib), the reference Logger's surface."""
import glob
import struct
import zlib

import numpy as np
import pytest

from unsupervised_anomaly_detection_brain_mri_amd.utils import logger as lg


def _decode_png(png):
    assert png[:8] == b'\x89PNG\r\n\x1a\n'
    pos, idat, hdr = 8, b'', None
    while pos < len(png):
        (ln,) = struct.unpack_from('>I', png, pos)
        tag, data = png[pos + 4:pos + 8], png[pos + 8:pos + 8 + ln]
        assert struct.unpack_from('>I', png, pos + 8 + ln)[0] == zlib.crc32(tag + data) & 0xFFFFFFFF
        if tag == b'IHDR':
            hdr = struct.unpack('>IIBBBBB', data)
        elif tag == b'IDAT':
            idat += data
        pos += 12 + ln
    w, h, depth, ctype = hdr[:4]
    ch = 1 if ctype == 0 else 3
    raw = zlib.decompress(idat)
    rows = [raw[y * (1 + w * ch) + 1:(y + 1) * (1 + w * ch)] for y in range(h)]
    assert all(raw[y * (1 + w * ch)] == 0 for y in range(h)) and depth == 8
    return np.frombuffer(b''.join(rows), np.uint8).reshape((h, w) if ch == 1 else (h, w, 3))


def test_logger_round_trip(tmp_path):
    log = lg.Logger(None, str(tmp_path / 'logs'))
    rng = np.random.default_rng(0)
    imgs = rng.random((3, 8, 6, 1)).astype(np.float32)
    log.summarize(1, lg.Phase.TRAIN, summaries_dict={'loss': np.float32(2.5), 'kl': np.array([1.0, 3.0]), 'x': imgs, 'skip': None})
    log.summarize(2, lg.Phase.TRAIN, summaries_dict={'loss': np.float32(1.25)})
    log.summarize(1, lg.Phase.VAL, scope='val', summaries_dict={'loss': 7.0})
    log.close()
    files = glob.glob(str(tmp_path / 'logs' / 'TRAIN' / 'events.out.tfevents.*'))
    assert len(files) == 1 and len(glob.glob(str(tmp_path / 'logs' / '*'))) == 3
    ev = lg.read_events(files[0])
    assert ev[0] == (0, {}) and [e[0] for e in ev] == [0, 1, 2]
    v = ev[1][1]
    assert v['loss'] == 2.5 and v['kl'] == 2.0 and 'skip' not in v and set(k for k in v if k.startswith('x/')) == {'x/image/0', 'x/image/1', 'x/image/2'}
    kind, hh, ww, png = v['x/image/1']
    assert (kind, hh, ww) == ('image', 8, 6)
    dec = _decode_png(png)
    exp = np.clip(imgs[1, ..., 0].astype(np.float64) * 255.0 / imgs[1].max(), 0, 255).astype(np.uint8)
    assert np.abs(dec.astype(int) - exp.astype(int)).max() <= 1 and dec.max() >= 254        # scaled so that the maximum maps to 255
    assert ev[2][1] == {'loss': 1.25}
    val = lg.read_events(glob.glob(str(tmp_path / 'logs' / 'VAL' / 'events*'))[0])
    assert val[1] == (1, {'val/loss': 7.0})
    # corrupt one payload byte -> crc failure
    data = bytearray(open(files[0], 'rb').read()); data[40] ^= 1
    open(files[0], 'wb').write(bytes(data))
    with pytest.raises(ValueError):
        lg.read_events(files[0])
    with pytest.raises(ValueError):
        lg.Logger(None, str(tmp_path / 'l2')).summarize(0, 'BOGUS', summaries_dict={'a': 1.0})


def test_image_scaling_rule_and_rgb_png():
    a = np.array([[-1.0, 0.0], [0.5, 1.0]])
    u = lg._to_u8(a)                                   # signed image: 0 -> 128, +-max -> 128 +- 127
    assert u.tolist() == [[1, 128], [191, 255]]
    rgb = (np.arange(2 * 3 * 3) * 10 % 256).astype(np.uint8).reshape(2, 3, 3)
    assert np.array_equal(_decode_png(lg.encode_png(rgb)), rgb)


def test_trainer_log_hook_without_engine(tmp_path):
    """AEMODEL.log_to_tensorboard: lazily creates the per-phase writers next to the checkpoint directory, honours config.useTensorboard."""
    import glob as _glob
    from unsupervised_anomaly_detection_brain_mri_amd.models import variational_autoencoder
    from unsupervised_anomaly_detection_brain_mri_amd.trainers import VAE, Phase
    t = object.__new__(VAE)                                   # no engine: host logic only
    t.config = VAE.Config()
    t.config.dataset, t.config.description, t.config.batchsize = 'Synthetic', '', 4
    t.network = variational_autoencoder
    t.checkpointDir = str(tmp_path / 'ck' / 'variational_autoencoder')
    t.logger = None
    t.config.useTensorboard = False
    t.log_to_tensorboard(0, {'loss': 1.0}, None, Phase.TRAIN)
    assert not (tmp_path / 'ck' / 'logs').exists()
    t.config.useTensorboard = True
    t.log_to_tensorboard(0, {'loss': [1.0, 3.0], 'kl': np.float32(0.5)}, None, Phase.TRAIN)
    t.log_to_tensorboard(1, {'loss': 1.5}, [np.ones((2, 4, 4, 1), np.float32)], Phase.VAL)
    d = tmp_path / 'ck' / 'logs' / 'variational_autoencoder' / t.model_dir
    tr = lg.read_events(_glob.glob(str(d / '*' / 'TRAIN' / 'events*'))[0])          # <logs>/<network>/<model_dir>/<YYYYmmdd_HHMMSS>/TRAIN
    assert tr[1] == (0, {'loss': 2.0, 'kl': 0.5})
    va = lg.read_events(_glob.glob(str(d / '*' / 'VAL' / 'events*'))[0])
    assert va[1][0] == 1 and va[1][1]['loss'] == 1.5 and {'x/image/0', 'x/image/1'} <= set(va[1][1])


def test_process_collects_image_strip_when_asked(tmp_path):
    """AEMODEL.process with config.tfSummaryImages: the per-step maps go through trainer_utils.get_summary_dict into one image summary."""
    import glob as _glob
    from unsupervised_anomaly_detection_brain_mri_amd.models import variational_autoencoder
    from unsupervised_anomaly_detection_brain_mri_amd.trainers import VAE, Phase
    from unsupervised_anomaly_detection_brain_mri_amd.utils.synthetic import SyntheticDataset
    t = object.__new__(VAE)
    t.config = VAE.Config()
    t.config.dataset, t.config.description, t.config.batchsize, t.config.useTensorboard, t.config.tfSummaryImages = 'Synthetic', '', 4, True, True
    t.network, t.checkpointDir, t.logger, t.curves = variational_autoencoder, str(tmp_path / 'ck' / 'net'), None, {}
    calls = []

    def fake_step(batch, phase, fetch_maps=True, **kw):
        calls.append(fetch_maps)
        b = np.asarray(batch)
        return {'loss': np.float32(1.0 + len(calls)), 'kl': np.float32(0.1), 'reconstruction': b * 0.5, 'L1': np.abs(b - b * 0.5)}
    t.step = fake_step
    ds = SyntheticDataset(8, 8, 16, 16, seed=0)
    out = t.process(ds, 3, Phase.VAL)
    assert calls == [True, True] and out['loss'] == 2.5 and t.curves['VAL/loss'] == [2.5]
    ev = lg.read_events(_glob.glob(str(tmp_path / 'ck' / 'logs' / 'variational_autoencoder' / t.model_dir / '*' / 'VAL' / 'events*'))[0])
    step, vals = ev[1]
    imgs = [k for k in vals if k.startswith('x/image')]
    assert step == 3 and vals['loss'] == 2.5 and len(imgs) == 8 and vals[imgs[0]][1:3] == (16, 48)       # input | reconstruction | L1 side by side
    t.config.tfSummaryImages = False
    calls.clear()
    t.process(ds, 4, Phase.VAL)
    assert calls == [False, False]
