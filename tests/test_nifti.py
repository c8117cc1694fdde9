"""utils/nifti.py: NIfTI-1 reader (hand-assembled header incl. big-endian and scl_slope), the volume -> slices restatement of
dataloaders/MSLUB.py:146-196,247-275 against a direct numpy / scipy computation, and the patient-level cache builder."""
import gzip
import struct

import numpy as np
import pytest
from scipy.ndimage import zoom

from unsupervised_anomaly_detection_brain_mri_amd.utils import nifti
from unsupervised_anomaly_detection_brain_mri_amd.utils.slice_cache import read_cache


def _raw_nifti(vol_zyx, end='<', dtype='i2', slope=0.0, inter=0.0):
    nz, ny, nx = vol_zyx.shape
    hdr = bytearray(348)
    struct.pack_into(end + 'i', hdr, 0, 348)
    struct.pack_into(end + '8h', hdr, 40, 3, nx, ny, nz, 1, 1, 1, 1)
    struct.pack_into(end + '2h', hdr, 70, {'i2': 4, 'f4': 16, 'u1': 2}[dtype], np.dtype(dtype).itemsize * 8)
    struct.pack_into(end + '3f', hdr, 108, 352.0, slope, inter)
    hdr[344:348] = b'n+1\x00'
    return bytes(hdr) + bytes(4) + np.ascontiguousarray(vol_zyx, end + dtype).tobytes()


def test_read_nifti_variants(tmp_path):
    rng = np.random.default_rng(0)
    vol = rng.integers(0, 1000, (5, 6, 7)).astype(np.int16)
    (tmp_path / 'a.nii').write_bytes(_raw_nifti(vol))
    d, h = nifti.read_nifti(str(tmp_path / 'a.nii'))
    assert d.shape == (5, 6, 7) and np.array_equal(d, vol) and h['dim'][1:4] == (7, 6, 5)
    # x runs fastest in the file: voxel (x=3, y=2, z=1) is byte offset ((1 * 6 + 2) * 7 + 3)
    assert d[1, 2, 3] == vol.reshape(-1)[(1 * 6 + 2) * 7 + 3]
    with gzip.open(tmp_path / 'b.nii.gz', 'wb') as f:
        f.write(_raw_nifti(vol, end='>', slope=2.0, inter=-1.0))
    d2, h2 = nifti.read_nifti(str(tmp_path / 'b.nii.gz'))
    assert h2['endian'] == '>' and np.array_equal(d2, vol * 2.0 - 1.0)
    fv = rng.random((3, 4, 4)).astype(np.float32); fv[0, 0, 0] = np.nan
    (tmp_path / 'c.nii').write_bytes(_raw_nifti(fv, dtype='f4'))
    d3, _ = nifti.read_nifti(str(tmp_path / 'c.nii'))
    assert d3[0, 0, 0] == 0 and np.allclose(d3.reshape(-1)[1:], fv.reshape(-1)[1:])
    nifti.write_nifti(str(tmp_path / 'w.nii.gz'), fv[1:])
    assert np.allclose(nifti.read_nifti(str(tmp_path / 'w.nii.gz'))[0], fv[1:])
    (tmp_path / 'bad.nii').write_bytes(b'\x00' * 400)
    with pytest.raises(ValueError):
        nifti.read_nifti(str(tmp_path / 'bad.nii'))


def _phantom(seed, shape=(12, 40, 40)):
    rng = np.random.default_rng(seed)
    z, y, x = np.meshgrid(*[np.linspace(-1, 1, s) for s in shape], indexing='ij')
    brain = (x ** 2 + y ** 2 + (z * 0.8) ** 2) < 0.7
    vol = (500 + 200 * x + 100 * rng.standard_normal(shape)) * brain + 30 * rng.random(shape)
    seg = ((x - 0.2) ** 2 + (y + 0.1) ** 2 + z ** 2 < 0.03).astype(np.float32)
    return vol, seg, brain.astype(np.float32)


def test_volume_to_slices_matches_direct_computation():
    vol, seg, brain = _phantom(1)
    imgs, labs, kept = nifti.volume_to_slices(vol, seg, brain, axis='axial', slice_start=1, slice_end=11, slice_resolution=(64, 64))
    v = vol * (brain >= 0.1)
    v = v.astype(np.float32)
    q = np.percentile(v, 99.8); v[v > q] = q
    v = v * np.float32(1.0 / v.max())
    exp_kept = [s for s in range(1, 11) if np.percentile(v[s], 90) >= 0.2]
    assert kept == exp_kept and len(kept) > 3 and imgs.shape == (len(kept), 64, 64)
    s = kept[2]
    # a slice SMALLER than the target is zero-padded to it (MSLUB.py:167-178), after which the zoom factor is 1
    np.testing.assert_allclose(imgs[2], np.pad(v[s], 12), rtol=1e-5, atol=1e-6)
    assert np.array_equal(labs[2], np.pad((seg[s] >= 0.9).astype(np.float32), 12)) and set(np.unique(labs)) <= {0.0, 1.0}
    # a LARGER slice is resampled with scipy.ndimage.zoom (order-3 spline; nearest boundary mode for the label map)
    i32, l32, k32 = nifti.volume_to_slices(vol, seg, brain, axis='axial', slice_start=1, slice_end=11, slice_resolution=(32, 32))
    assert k32 == kept
    np.testing.assert_allclose(i32[2], zoom(v[s], 32 / 40.0), rtol=1e-5, atol=1e-6)
    assert np.array_equal(l32[2], (zoom((seg[s] >= 0.9).astype(np.float64), 32 / 40.0, mode='nearest') >= 0.9).astype(np.float32))
    assert imgs.min() >= -0.05 and imgs.max() <= 1.05
    # coronal view walks the y axis; a slice smaller than the target is zero-padded first
    ic, _, kc = nifti.volume_to_slices(vol, seg, brain, axis='coronal', slice_start=10, slice_end=30, slice_resolution=(48, 48), empty_thresh=0.05)
    assert ic.shape[1:] == (48, 48) and all(10 <= k < 30 for k in kc)
    with pytest.raises(NotImplementedError):
        nifti.volume_to_slices(vol, denoise=True)


def test_build_cache_patient_split(tmp_path):
    patients = []
    for i in range(5):
        vol, seg, brain = _phantom(10 + i)
        d = tmp_path / f'p{i}'
        d.mkdir()
        nifti.write_nifti(str(d / 'flair.nii.gz'), vol)
        nifti.write_nifti(str(d / 'gt.nii.gz'), seg, dtype='u1')
        nifti.write_nifti(str(d / 'mask.nii.gz'), brain, dtype='u1')
        patients.append({'name': f'p{i}', 'volume': str(d / 'flair.nii.gz'), 'groundtruth': str(d / 'gt.nii.gz'), 'skullmap': str(d / 'mask.nii.gz')})
    info = nifti.build_cache(str(tmp_path / 'cache'), patients, partition={'TRAIN': 0.6, 'VAL': 0.2, 'TEST': 0.2}, seed=3,
                             slice_start=1, slice_end=11, slice_resolution=(32, 32))
    assert sorted(len(v) for v in info['split'].values()) == [1, 1, 3] and sum(len(v) for v in info['split'].values()) == 5
    images, labels, index = read_cache(str(tmp_path / 'cache'))
    assert images.shape[1:] == (32, 32, 1) and images.shape[0] == info['slices'] == len(index['sets']) == len(index['patients'])
    # slices of one patient never straddle two splits
    by_patient = {}
    for pt, st in zip(index['patients'], index['sets']):
        by_patient.setdefault(pt, set()).add(st)
    assert all(len(v) == 1 for v in by_patient.values()) and len(by_patient) == 5
    assert set(np.unique(labels)) <= {0, 2, 10} and (labels == 10).any() and (labels == 2).any()
    s = nifti.partition_patients(10, {'TRAIN': 0.7, 'VAL': 0.2, 'TEST': 0.1}, np.random.default_rng(0))
    assert [len(s[k]) for k in ('TRAIN', 'VAL', 'TEST')] == [7, 2, 1] and len(set(np.concatenate(list(s.values())))) == 10


def test_volumes_from_cache(tmp_path):
    from unsupervised_anomaly_detection_brain_mri_amd.utils.slice_cache import volumes_from_cache, write_cache
    rng = np.random.default_rng(0)
    images = rng.random((10, 8, 8, 1)).astype(np.float32)
    labels = rng.choice([0, 2, 3, 7, 10], size=(10, 8, 8)).astype(np.uint8)
    sets = [0, 0, 2, 2, 2, 1, 2, 2, 0, 2]
    pats = ['a', 'a', 'b', 'b', 'c', 'c', 'b', 'c', 'd', 'c']
    write_cache(str(tmp_path / 'c'), images, sets, labels, patients=pats)
    vols, labs, masks, names = volumes_from_cache(str(tmp_path / 'c'), 'TEST')
    assert names == ['b', 'c'] and [v.shape for v in vols] == [(3, 8, 8), (3, 8, 8)]
    np.testing.assert_array_equal(vols[0], images[[2, 3, 6], ..., 0].astype(np.float64))
    np.testing.assert_array_equal(labs[1], (labels[[4, 7, 9]] == 10).astype(np.float64))
    # brain mask = every label the LUT keeps (GM 2, WM 3, LESION 10 are brain; BACKGROUND 0 and SKULL 7 are not)
    np.testing.assert_array_equal(masks[0], np.isin(labels[[2, 3, 6]], [2, 3, 10]).astype(np.float64))
    write_cache(str(tmp_path / 'nolab'), images, sets)
    with pytest.raises(ValueError):
        volumes_from_cache(str(tmp_path / 'nolab'))


def test_rotations_and_center_crop():
    from scipy.ndimage import rotate
    vol, seg, brain = _phantom(2)
    base, lb, kept = nifti.volume_to_slices(vol, seg, brain, slice_start=3, slice_end=9, slice_resolution=(40, 40))
    imgs, labs, k2 = nifti.volume_to_slices(vol, seg, brain, slice_start=3, slice_end=9, slice_resolution=(40, 40), rotations=(0, 15), center_crop=(24, 32))
    assert k2 == [s for s in kept for _ in range(2)] and imgs.shape == (2 * len(kept), 32, 24)
    np.testing.assert_allclose(imgs[0], nifti.crop_center(base[0], 24, 32))
    np.testing.assert_allclose(imgs[1], nifti.crop_center(rotate(base[0], 15, reshape=False), 24, 32), rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(labs[1], nifti.crop_center(rotate(lb[0], 15, reshape=False, mode='nearest'), 24, 32))
    assert nifti.crop_center(np.arange(100).reshape(10, 10), 4, 6).shape == (6, 4)


def test_read_nrrd(tmp_path):
    import gzip as _gz
    rng = np.random.default_rng(0)
    vol = rng.integers(-500, 500, (4, 5, 6)).astype(np.int16)               # [x, y, z], x fastest in the file
    head = b'NRRD0004\n# a comment\ntype: short\ndimension: 3\nsizes: 4 5 6\nendian: little\nencoding: raw\nspace: left-posterior-superior\n\n'
    (tmp_path / 'a.nrrd').write_bytes(head + vol.tobytes(order='F'))
    d, h = nifti.read_nrrd(str(tmp_path / 'a.nrrd'))
    assert d.shape == (4, 5, 6) and np.array_equal(d, vol) and h['dimension'] == '3'
    big = b'NRRD0005\r\ntype: float\r\ndimension: 2\r\nsizes: 3 2\r\nendian: big\r\nencoding: gzip\r\n\r\n'
    fv = rng.random((3, 2)).astype('>f4')
    (tmp_path / 'b.nrrd').write_bytes(big + _gz.compress(fv.tobytes(order='F')))
    d2, _ = nifti.read_nrrd(str(tmp_path / 'b.nrrd'))
    assert np.array_equal(d2, fv)
    (tmp_path / 'c.nrrd').write_bytes(b'NRRD0004\ntype: short\nsizes: 2 2\ndata file: x.raw\n\n')
    with pytest.raises(ValueError):
        nifti.read_nrrd(str(tmp_path / 'c.nrrd'))
