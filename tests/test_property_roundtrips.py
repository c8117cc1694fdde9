"""Property tests (hypothesis) of the three file formats written without their reference libraries: TF tensor bundles, NIfTI-1 volumes,
tfevents records -- random content survives write -> read bit for bit."""
import os
import tempfile

import numpy as np
from hypothesis import given, settings, strategies as st
from hypothesis.extra import numpy as hnp

from unsupervised_anomaly_detection_brain_mri_amd.utils import logger as lg
from unsupervised_anomaly_detection_brain_mri_amd.utils import nifti, tf_checkpoint as tfc

names = st.text(alphabet='abcdefghijklmnopqrstuvwxyz_/0123456789', min_size=1, max_size=24).filter(lambda s: not s.startswith('/'))
arrays = hnp.arrays(np.float32, hnp.array_shapes(min_dims=0, max_dims=4, max_side=5), elements=st.floats(-1e6, 1e6, width=32))


@settings(max_examples=40, deadline=None)
@given(st.dictionaries(names, arrays, min_size=1, max_size=12), st.integers(1, 9))
def test_bundle_round_trip(tensors, per_block):
    with tempfile.TemporaryDirectory() as d:
        tfc.write_checkpoint(os.path.join(d, 'ck'), tensors, entries_per_block=per_block)
        got = tfc.read_checkpoint(os.path.join(d, 'ck'))
    assert set(got) == set(tensors)
    for k, v in tensors.items():
        assert got[k].shape == v.shape and np.array_equal(got[k], v)


@settings(max_examples=60, deadline=None)
@given(st.integers(0, 2 ** 64 - 1))
def test_varint_round_trip(v):
    b = tfc._put_varint(v)
    assert tfc._get_varint(b, 0) == (v, len(b)) and len(b) <= 10


@settings(max_examples=25, deadline=None)
@given(hnp.arrays(np.int16, st.tuples(st.integers(1, 6), st.integers(1, 7), st.integers(1, 8)), elements=st.integers(-2000, 2000)), st.booleans())
def test_nifti_round_trip(vol, gz):
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, 'v.nii.gz' if gz else 'v.nii')
        nifti.write_nifti(p, vol, dtype='i2')
        got, hdr = nifti.read_nifti(p)
    assert got.shape == vol.shape and np.array_equal(got, vol) and hdr['dim'][1:4] == vol.shape[::-1]


@settings(max_examples=25, deadline=None)
@given(st.lists(st.tuples(names, st.floats(-1e6, 1e6, width=32)), min_size=1, max_size=8, unique_by=lambda t: t[0]), st.integers(0, 10 ** 6))
def test_event_scalars_round_trip(vals, step):
    with tempfile.TemporaryDirectory() as d:
        w = lg.EventFileWriter(d)
        w.add_summary(vals, step)
        w.close()
        ev = lg.read_events(w.path)
    assert ev[1][0] == step and ev[1][1] == {k: np.float32(v) for k, v in vals}
