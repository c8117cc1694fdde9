"""Builds a slice cache (utils/slice_cache.py) from a directory of NIfTI volumes laid out like the reference's MS datasets
(dataloaders/MSLUB.py:277-322): <root>/<patient>/<patient>_<PROTOCOL>.nii.gz, <patient>_consensus_gt.nii.gz, <patient>_brainmask.nii.gz.

    python tools/build_cache.py <root> <cache_dir> --protocol FLAIR --res 128 --start 15 --end 125

The volume -> slice steps are those of utils/nifti.py (skull stripping, percentile scaling, empty-slice filter, pad / zoom); the ITK
CurvatureFlow denoising of the reference loaders is not applied."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from unsupervised_anomaly_detection_brain_mri_amd.utils import nifti  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('root'); ap.add_argument('cache')
    ap.add_argument('--protocol', default='FLAIR')
    ap.add_argument('--gt', default='consensus_gt'); ap.add_argument('--mask', default='brainmask')
    ap.add_argument('--res', type=int, default=128); ap.add_argument('--axis', default='axial')
    ap.add_argument('--start', type=int, default=0); ap.add_argument('--end', type=int, default=155)
    ap.add_argument('--train', type=float, default=0.7); ap.add_argument('--val', type=float, default=0.2); ap.add_argument('--test', type=float, default=0.1)
    ap.add_argument('--seed', type=int, default=0)
    a = ap.parse_args()
    patients = []
    for name in sorted(os.listdir(a.root)):
        d = os.path.join(a.root, name)
        vol = os.path.join(d, f'{name}_{a.protocol}.nii.gz')
        if not os.path.isfile(vol):
            continue
        gt, mk = os.path.join(d, f'{name}_{a.gt}.nii.gz'), os.path.join(d, f'{name}_{a.mask}.nii.gz')
        patients.append({'name': name, 'volume': vol, 'groundtruth': gt if os.path.isfile(gt) else None, 'skullmap': mk if os.path.isfile(mk) else None})
    if not patients:
        raise SystemExit(f'no <patient>/<patient>_{a.protocol}.nii.gz under {a.root}')
    info = nifti.build_cache(a.cache, patients, partition={'TRAIN': a.train, 'VAL': a.val, 'TEST': a.test}, seed=a.seed, axis=a.axis,
                             slice_start=a.start, slice_end=a.end, slice_resolution=(a.res, a.res))
    print(json.dumps(info))


if __name__ == '__main__':
    main()
