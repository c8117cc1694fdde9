"""NIfTI-1 volumes without SimpleITK / nibabel, and the volume -> slice-cache pipeline of the reference's dataset classes.

`read_nifti` parses the 348-byte NIfTI-1 header (`.nii`, `.nii.gz`; single-file, little- or big-endian) and returns the voxel array in the
index order the reference sees: utils/NII.py reads with SimpleITK and takes `sitk.GetArrayFromImage`, i.e. [z, y, x] (the file's x runs
fastest), with `scl_slope * v + scl_inter` applied when the header asks for it and NaNs set to 0 (NII.py:12-16).

`volume_to_slices` restates what dataloaders/MSLUB.py:146-196,247-275 (and its siblings MSISBI2015 / MSSEG2008 / BRAINWEB) do to one volume:
binarise the brain mask at 0.1 and multiply (skull stripping, NII.py:75-81), clamp to the [0, 99.8] percentiles and scale by the maximum
(`normalize('scaling', 0, 99.8)`, NII.py:50-70), walk the slices `sliceStart .. sliceEnd` of the chosen view, drop the "empty" ones
(90th percentile < 0.2, MSLUB.py:161), zero-pad to and `scipy.ndimage.zoom` onto `sliceResolution`, binarise the label slice at 0.9.
NOT restated: `nii.denoise()` (SimpleITK's CurvatureFlow, three iterations) -- an ITK filter with no counterpart here; `denoise=True` raises.

`build_cache` turns a list of patients into the slice cache of utils/slice_cache.py with the reference's patient-level TRAIN / VAL / TEST
partition (a permutation of the patients cut at floor(fraction * n), MSLUB.py:71-90)."""
import gzip
import math
import struct

import numpy as np

_DTYPES = {2: 'u1', 4: 'i2', 8: 'i4', 16: 'f4', 64: 'f8', 256: 'i1', 512: 'u2', 768: 'u4', 1024: 'i8', 1280: 'u8'}
VIEW_MAPPING = {'saggital': 2, 'coronal': 1, 'axial': 0}          # dataset Options.viewMapping (MSLUB.py:53) on the [z, y, x] array


def read_nifti(path):
    """-> (data [z,y,x] float64, header dict with dim, pixdim, datatype, vox_offset, scl_slope, scl_inter)."""
    opener = gzip.open if str(path).endswith('.gz') else open
    with opener(path, 'rb') as f:
        raw = f.read()
    if len(raw) < 352:
        raise ValueError(f'{path}: too short for a NIfTI-1 file')
    end = '<'
    if struct.unpack('<i', raw[:4])[0] != 348:
        end = '>'
        if struct.unpack('>i', raw[:4])[0] != 348:
            raise ValueError(f'{path}: sizeof_hdr is not 348 (not NIfTI-1)')
    magic = raw[344:348]
    if magic not in (b'n+1\x00', b'ni1\x00'):
        raise ValueError(f'{path}: bad NIfTI magic {magic!r}')
    if magic == b'ni1\x00':
        raise ValueError(f'{path}: two-file NIfTI (.hdr / .img) is not supported')
    dim = struct.unpack(end + '8h', raw[40:56])
    datatype, bitpix = struct.unpack(end + '2h', raw[70:74])
    pixdim = struct.unpack(end + '8f', raw[76:108])
    vox_offset, slope, inter = struct.unpack(end + '3f', raw[108:120])
    if datatype not in _DTYPES:
        raise ValueError(f'{path}: unsupported NIfTI datatype {datatype}')
    nd = dim[0]
    if nd < 3 or nd > 4 or (nd == 4 and dim[4] != 1):
        raise ValueError(f'{path}: expected a 3-D volume, header dim = {dim}')
    nx, ny, nz = dim[1:4]
    dt = np.dtype(end + _DTYPES[datatype])
    off = int(vox_offset) if vox_offset >= 352 else 352
    cnt = nx * ny * nz
    if len(raw) < off + cnt * dt.itemsize:
        raise ValueError(f'{path}: voxel data is truncated')
    data = np.frombuffer(raw, dt, cnt, off).reshape(nz, ny, nx).astype(np.float64)      # x fastest in the file = the [z,y,x] C array
    if slope != 0.0 and not (slope == 1.0 and inter == 0.0):
        data = data * slope + inter
    data[np.isnan(data)] = 0
    return data, dict(dim=dim, pixdim=pixdim, datatype=datatype, bitpix=bitpix, vox_offset=off, scl_slope=slope, scl_inter=inter, endian=end)


def write_nifti(path, data_zyx, dtype='f4', pixdim=(1.0, 1.0, 1.0)):
    """Minimal single-file NIfTI-1 writer (tests, exporting results)."""
    a = np.asarray(data_zyx)
    nz, ny, nx = a.shape
    code = {v: k for k, v in _DTYPES.items()}[np.dtype(dtype).str[1:]]
    hdr = bytearray(348)
    struct.pack_into('<i', hdr, 0, 348)
    struct.pack_into('<8h', hdr, 40, 3, nx, ny, nz, 1, 1, 1, 1)
    struct.pack_into('<2h', hdr, 70, code, np.dtype(dtype).itemsize * 8)
    struct.pack_into('<8f', hdr, 76, 1.0, pixdim[0], pixdim[1], pixdim[2], 0, 0, 0, 0)
    struct.pack_into('<3f', hdr, 108, 352.0, 1.0, 0.0)
    hdr[344:348] = b'n+1\x00'
    payload = bytes(hdr) + b'\x00' * 4 + np.ascontiguousarray(a, '<' + np.dtype(dtype).str[1:]).tobytes()
    opener = gzip.open if str(path).endswith('.gz') else open
    with opener(path, 'wb') as f:
        f.write(payload)


_NRRD_TYPES = {'signed char': 'i1', 'int8': 'i1', 'int8_t': 'i1', 'uchar': 'u1', 'unsigned char': 'u1', 'uint8': 'u1', 'uint8_t': 'u1',
               'short': 'i2', 'short int': 'i2', 'signed short': 'i2', 'int16': 'i2', 'int16_t': 'i2', 'ushort': 'u2', 'unsigned short': 'u2',
               'uint16': 'u2', 'uint16_t': 'u2', 'int': 'i4', 'signed int': 'i4', 'int32': 'i4', 'int32_t': 'i4', 'uint': 'u4', 'unsigned int': 'u4',
               'uint32': 'u4', 'uint32_t': 'u4', 'longlong': 'i8', 'long long': 'i8', 'int64': 'i8', 'int64_t': 'i8', 'ulonglong': 'u8',
               'unsigned long long': 'u8', 'uint64': 'u8', 'uint64_t': 'u8', 'float': 'f4', 'double': 'f8'}


def read_nrrd(path):
    """NRRD (attached header; encodings raw / gzip) -> (data, header dict) in the index order of `nrrd.read` that dataloaders/NRRD.py:11 relies
    on: shape = the header's `sizes` (first axis fastest, i.e. Fortran order)."""
    raw = open(path, 'rb').read()
    if not raw.startswith(b'NRRD'):
        raise ValueError(f'{path}: not an NRRD file')
    end = raw.find(b'\n\n')
    crlf = raw.find(b'\r\n\r\n')
    if crlf != -1 and (end == -1 or crlf < end):
        head, body = raw[:crlf], raw[crlf + 4:]
    elif end != -1:
        head, body = raw[:end], raw[end + 2:]
    else:
        raise ValueError(f'{path}: no header terminator')
    hdr = {}
    for line in head.decode('ascii', 'replace').splitlines()[1:]:
        if line.startswith('#') or ':' not in line:
            continue
        k, v = line.split(':', 1)
        hdr[k.strip().lower()] = v.lstrip('=').strip()
    if 'data file' in hdr or 'datafile' in hdr:
        raise ValueError(f'{path}: detached NRRD data files are not supported')
    t = hdr.get('type', '').lower()
    if t not in _NRRD_TYPES:
        raise ValueError(f'{path}: unsupported NRRD type {t!r}')
    sizes = [int(v) for v in hdr['sizes'].split()]
    enc = hdr.get('encoding', 'raw').lower()
    if enc in ('gzip', 'gz'):
        body = gzip.decompress(body)
    elif enc != 'raw':
        raise ValueError(f'{path}: unsupported NRRD encoding {enc!r}')
    dt = np.dtype(('>' if hdr.get('endian', 'little').lower() == 'big' else '<') + _NRRD_TYPES[t])
    cnt = int(np.prod(sizes))
    if len(body) < cnt * dt.itemsize:
        raise ValueError(f'{path}: NRRD data is truncated')
    return np.frombuffer(body, dt, cnt).reshape(sizes, order='F'), hdr


def normalize_scaling(vol, lower=0, upper=99.8):
    """NII.normalize(method='scaling', lowerpercentile, upperpercentile) (NII.py:50-66)."""
    v = vol.astype(np.float32)
    if lower is not None:
        q = np.percentile(v, lower); v[v < q] = q
    if upper is not None:
        q = np.percentile(v, upper); v[v > q] = q
    if v.max() > 0.0:
        v = v * np.float32(1.0 / v.max())
    return v


def crop_center(img, cropx, cropy):
    """utils/image_utils.py:4-12."""
    y, x = img.shape[:2]
    sx, sy = x // 2 - cropx // 2, y // 2 - cropy // 2
    return img[sy:sy + cropy, sx:sx + cropx]


def volume_to_slices(vol, seg=None, brainmask=None, axis='axial', slice_start=0, slice_end=155, slice_resolution=None, skull_stripping=True,
                     view_mapping=None, empty_percentile=90, empty_thresh=0.2, denoise=False, rotations=(0,), center_crop=None):
    """-> (images [k,H,W] float32 in [0,1], labels [k,H,W] float32 in {0,1}, slice indices kept).
    rotations: angles in degrees, one output per angle and slice (dataloaders/BRAINWEB.py:156-162: scipy.ndimage.rotate, reshape False, the label
    map with mode 'nearest'); center_crop (width, height): the `useCrops` / cropType 'center' option (MSLUB.py:206-210)."""
    from scipy.ndimage import rotate, zoom
    if denoise:
        raise NotImplementedError("nii.denoise() is SimpleITK's CurvatureFlow filter (MSLUB.py:257); it is not restated here")
    vm = view_mapping or VIEW_MAPPING
    ax = vm[axis]
    vol = np.array(vol, np.float64)
    vol[np.isnan(vol)] = 0.0
    if seg is None:
        seg = np.zeros_like(vol)
    seg = (np.asarray(seg) >= 0.9).astype(np.float64)                       # MSLUB.py:264-265
    if skull_stripping and brainmask is not None:
        vol = vol * (np.asarray(brainmask) >= 0.1)                          # NII.apply_skullmap
    vol = normalize_scaling(vol)
    imgs, labs, kept = [], [], []
    for s in range(slice_start, min(slice_end, vol.shape[ax])):
        idx = [slice(None)] * 3
        idx[ax] = s
        sd, ss = vol[tuple(idx)], seg[tuple(idx)]
        if np.percentile(sd, empty_percentile) < empty_thresh:              # MSLUB.py:161: skip "empty" slices
            continue
        if slice_resolution is not None:
            H, W = slice_resolution
            py = (math.floor((H - sd.shape[0]) / 2.0), math.ceil((H - sd.shape[0]) / 2.0)) if sd.shape[0] < H else (0, 0)
            px = (math.floor((W - sd.shape[1]) / 2.0), math.ceil((W - sd.shape[1]) / 2.0)) if sd.shape[1] < W else (0, 0)
            if py != (0, 0) or px != (0, 0):
                sd = np.pad(sd, (py, px), 'constant'); ss = np.pad(ss, (py, px), 'constant')
            f = float(H) / float(sd.shape[0])                               # one factor for both axes, as the reference (:179-180)
            sd = zoom(sd, f)
            ss = zoom(ss, f, mode='nearest')
            ss = (ss >= 0.9).astype(np.float64)
        for angle in rotations:
            sdr, ssr = (sd, ss) if angle == 0 else (rotate(sd, angle, reshape=False), rotate(ss, angle, reshape=False, mode='nearest'))
            if center_crop is not None:
                sdr, ssr = crop_center(sdr, center_crop[0], center_crop[1]), crop_center(ssr, center_crop[0], center_crop[1])
            imgs.append(np.asarray(sdr, np.float32)); labs.append(np.asarray(ssr, np.float32)); kept.append(s)
    if not imgs:
        return np.zeros((0, 0, 0), np.float32), np.zeros((0, 0, 0), np.float32), []
    return np.stack(imgs), np.stack(labs), kept


def partition_patients(n_patients, partition=None, rng=None):
    """Patient-level split of MSLUB.py:71-90: a permutation cut at floor(fraction * n) (fractions <= 1) or at absolute counts."""
    partition = partition or {'TRAIN': 0.7, 'VAL': 0.2, 'TEST': 0.1}
    rng = rng or np.random.default_rng(0)
    ridx = rng.permutation(n_patients)
    out, taken = {}, 0
    for split in partition:
        k = math.floor(partition[split] * n_patients) if partition[split] <= 1.0 else int(partition[split])
        k = min(k, n_patients - taken)
        out[split] = ridx[taken:taken + k]
        taken += k
    return out


def build_cache(directory, patients, partition=None, seed=0, **slice_options):
    """patients: [{'name', 'volume': path, 'groundtruth': path or None, 'skullmap': path or None}] -> slice cache in `directory`.
    slice_options: volume_to_slices keywords.  Returns the index dict that was written."""
    from .slice_cache import SET_TYPES, write_cache
    split = partition_patients(len(patients), partition, np.random.default_rng(seed))
    set_of = {}
    for name, ids in split.items():
        for i in ids:
            set_of[int(i)] = SET_TYPES.index(name)
    images, labels, sets, owner = [], [], [], []
    for i, p in enumerate(patients):
        if i not in set_of:
            continue
        vol, _ = read_nifti(p['volume'])
        seg = read_nifti(p['groundtruth'])[0] if p.get('groundtruth') else None
        msk = read_nifti(p['skullmap'])[0] if p.get('skullmap') else None
        im, lb, kept = volume_to_slices(vol, seg, msk, **slice_options)
        if len(kept):
            images.append(im); labels.append(lb); sets += [set_of[i]] * len(kept); owner += [p.get('name', str(i))] * len(kept)
    if not images:
        raise ValueError('no slice survived the filters')
    images = np.concatenate(images)[..., None]
    labels = np.concatenate(labels)
    # label map in BRAINWEB.LABELS values so that the cache's brain-mask LUT works: 10 = LESION, 2 = GM for every other in-brain pixel
    # (non-zero after skull stripping), 0 = BACKGROUND
    lab_u8 = np.where(labels > 0, 10, np.where(images[..., 0] > 0, 2, 0)).astype(np.uint8)
    write_cache(directory, images, sets, lab_u8, patients=owner,
                options={k: (list(v) if isinstance(v, tuple) else v) for k, v in slice_options.items()})
    return {'slices': int(images.shape[0]), 'shape': list(images.shape), 'split': {k: [int(i) for i in v] for k, v in split.items()}}
