"""Generates tests/golden/scoring_golden.npz by running the REFERENCE's own scoring code in this container:
trainers/Metrics.py (imports cleanly) and the numeric helpers of utils/Evaluation.py (cv2 / imageio / skimage are
stubbed in sys.modules — they are only used by plotting / PNG export / CC labelling, none of which is called).
Run from the repo root:  MPLBACKEND=Agg python tests/golden/make_scoring_golden.py
The reference never travels to the GPU box: only the resulting vectors are committed."""
import os
import sys
import types

import numpy as np

REF = '/root/reference'
os.environ.setdefault('MPLBACKEND', 'Agg')
sys.path.insert(0, REF)
for name in ('cv2', 'imageio', 'skimage', 'skimage.measure'):
    mod = types.ModuleType(name)
    sys.modules[name] = mod
sys.modules['imageio'].imwrite = lambda *a, **k: None
sys.modules['skimage.measure'].regionprops = None
sys.modules['skimage.measure'].label = None
sys.modules['skimage'].measure = sys.modules['skimage.measure']
import scipy.ndimage  # noqa: E402
import scipy.ndimage.morphology  # noqa: E402,F401  (old attribute paths used by the reference)
import scipy.ndimage.filters  # noqa: E402,F401
import scipy.misc  # noqa: E402,F401
import scipy.signal  # noqa: E402,F401
from trainers import Metrics  # noqa: E402
from utils import Evaluation  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def make_volume(seed=0, patients=2, slices=8, h=64, w=64):
    rng = np.random.default_rng(seed)
    S = patients * slices
    yy, xx = np.mgrid[0:h, 0:w]
    x = np.zeros((S, h, w)); xr = np.zeros((S, h, w)); lab = np.zeros((S, h, w), np.int64); bm = np.zeros((S, h, w), np.int64)
    for s in range(S):
        mask = ((yy - h / 2) / (h * 0.42)) ** 2 + ((xx - w / 2) / (w * 0.36)) ** 2 <= 1
        base = 0.5 + 0.2 * np.sin(yy / 7.0 + s) * np.cos(xx / 9.0)
        img = np.clip(base + rng.normal(0, 0.03, base.shape), 0, 1) * mask
        rec = np.clip(base + rng.normal(0, 0.02, base.shape), 0, 1) * mask
        for _ in range(2):
            cy, cx, r = rng.uniform(h / 2 - 9, h / 2 + 9), rng.uniform(w / 2 - 7, w / 2 + 7), rng.uniform(1.5, 4.0)
            blob = ((yy - cy) ** 2 + (xx - cx) ** 2 <= r * r) & mask
            img = np.where(blob, np.clip(img + 0.3, 0, 1), img)
            lab[s][blob] = 1
        x[s], xr[s], bm[s] = img, rec, mask
    # inputs are committed as float32: round first so the reference sees exactly the committed values
    return x.astype(np.float32).astype(np.float64), xr.astype(np.float32).astype(np.float64), lab, bm


def main():
    x, xr, lab, bm = make_volume()
    S = x.shape[0]
    prior = np.quantile(x, 0.9)
    # utils/Evaluation.py:282-289 per slice, then :311-312 per volume
    diffs = np.zeros_like(x)
    eroded = np.zeros_like(x)
    for s in range(S):
        d = np.maximum(x[s] - xr[s], 0)
        d = Evaluation.apply_brainmask(d, bm[s], erode=True)
        eroded[s] = Evaluation.apply_brainmask(np.ones_like(x[s]), bm[s], erode=True)
        d[x[s] < prior] = 0
        diffs[s] = d
    med = np.concatenate([Evaluation.apply_3d_median_filter(diffs[:8]), Evaluation.apply_3d_median_filter(diffs[8:])])
    pred, gt = med.flatten(), lab.astype(bool).flatten()
    auprc, _, _, _ = Metrics.compute_prc(pred, gt)
    aucroc, _, _, _ = Metrics.compute_roc(pred, gt)
    scores, threshs = Metrics.compute_dice_score(pred, lab.flatten(), 5)
    best_score, best_thr = Metrics.compute_dice_curve_recursive(pred, lab.flatten(), granularity=5)
    dice_at = np.array([Metrics.dice(np.where(pred > t, 1, 0), lab.flatten()) for t in (0.05, 0.1, 0.2)])
    squashed = Evaluation.squash_intensities(diffs[3])
    np.savez_compressed(os.path.join(HERE, 'scoring_golden.npz'),
                        x=x.astype(np.float32), xr=xr.astype(np.float32), lab=lab.astype(np.uint8), bm=bm.astype(np.uint8),
                        prior=np.float64(prior), diffs=diffs, eroded=eroded.astype(np.uint8), med=med,
                        auprc=np.float64(auprc), auroc=np.float64(aucroc), dice_scores=np.array(scores),
                        dice_threshs=np.array(threshs), best_score=np.float64(best_score), best_thr=np.float64(best_thr),
                        dice_at=dice_at, squashed=squashed)
    print('auprc', auprc, 'auroc', aucroc, 'best dice', best_score, '@', best_thr, 'sweep points', len(scores))


if __name__ == '__main__':
    main()
