"""End-to-end rehearsal of the real-data path on phantom volumes: NIfTI files -> tools/build_cache.py -> run.py --cache (HBM-resident
training set, per-patient TEST volumes through Evaluation.evaluate)."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unsupervised_anomaly_detection_brain_mri_amd.utils import nifti  # noqa: E402


def phantom(seed, shape=(24, 64, 64)):
    rng = np.random.default_rng(seed)
    z, y, x = np.meshgrid(*[np.linspace(-1, 1, s) for s in shape], indexing='ij')
    brain = (x ** 2 + y ** 2 + (z * 0.8) ** 2) < 0.7
    vol = (500 + 200 * x * y + 60 * rng.standard_normal(shape)) * brain + 20 * rng.random(shape)
    seg = ((x - 0.25) ** 2 + (y + 0.1) ** 2 + z ** 2 < 0.02)
    vol = vol + 400 * seg * brain
    return vol, seg.astype(np.float32), brain.astype(np.float32)


def main():
    tmp = tempfile.mkdtemp(prefix='uad_e2e_')
    for i in range(8):
        name = f'patient{i:02d}'
        d = os.path.join(tmp, 'data', name)
        os.makedirs(d)
        vol, seg, brain = phantom(100 + i)
        nifti.write_nifti(os.path.join(d, f'{name}_FLAIR.nii.gz'), vol)
        nifti.write_nifti(os.path.join(d, f'{name}_consensus_gt.nii.gz'), seg, dtype='u1')
        nifti.write_nifti(os.path.join(d, f'{name}_brainmask.nii.gz'), brain, dtype='u1')
    cache = os.path.join(tmp, 'cache')
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools', 'build_cache.py'), os.path.join(tmp, 'data'), cache, '--res', '64', '--start', '2',
                           '--end', '22', '--train', '0.5', '--val', '0.25', '--test', '0.25'])
    trainer, model = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ('AE', 'autoencoder')
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'run.py'), '-t', trainer, '-m', model, '-E', '3', '-b', '8', '-w', '64', '-g', '64', '-z', '64',
                           '--cache', cache, '-c', os.path.join(tmp, 'none.json')], cwd=tmp)


if __name__ == '__main__':
    main()
