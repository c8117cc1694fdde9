"""Oracle for the residual-map scoring path.

TEST INFRASTRUCTURE ONLY.  PINNED: tests/golden/scoring_golden.npz was produced by importing the reference's own
trainers/Metrics.py and the numeric helpers of utils/Evaluation.py in this container
(tests/golden/make_scoring_golden.py); tests/test_oracle_scoring.py checks every function below against it.
Exception: filter_3d_connected_components needs scikit-image (absent) -> restated with scipy.ndimage, unpinned.

Each function is an independent numpy restatement (no sklearn / scipy morphology calls) of:
  utils/Evaluation.py:84-89   apply_brainmask  (binary erosion, cross structuring element, 12 iterations)
  utils/Evaluation.py:108-110 apply_3d_median_filter (5x5x5, scipy default boundary mode 'reflect')
  utils/Evaluation.py:113-127 filter_3d_connected_components
  utils/Evaluation.py:282-289 residual map (positive residuals, brain mask, hyper-intensity prior)
  trainers/Metrics.py:17-19   AUPRC = sklearn average_precision_score
  trainers/Metrics.py:45-47   AUROC = sklearn roc_curve + auc
  trainers/Metrics.py:67-72   dice ;  :138-162 greedy recursive Dice-threshold sweep
"""
import numpy as np


def binary_erosion_cross(mask, iterations=12):
    """scipy.ndimage.binary_erosion(mask, generate_binary_structure(2,1), iterations), border_value=0."""
    m = np.asarray(mask).astype(bool)
    for _ in range(iterations):
        p = np.pad(m, 1, constant_values=False)
        m = p[1:-1, 1:-1] & p[:-2, 1:-1] & p[2:, 1:-1] & p[1:-1, :-2] & p[1:-1, 2:]
    return m


def apply_brainmask(x, brainmask, erode=True):
    bm = np.squeeze(np.asarray(brainmask))
    if erode:
        bm = binary_erosion_cross(bm, 12)
    return np.multiply(bm, np.squeeze(x))


def median_filter_3d(volume, k=5):
    """scipy.ndimage.median_filter(volume, (k,k,k)) with the default mode='reflect' (== numpy 'symmetric')."""
    r = k // 2
    p = np.pad(volume, r, mode='symmetric')
    win = np.lib.stride_tricks.sliding_window_view(p, (k, k, k))
    return np.median(win.reshape(*volume.shape, -1), axis=-1)


def residual_map(x, x_rec, brainmask=None, keep_only_positive=True, erode=True, prior_quantile=None):
    """utils/Evaluation.py:282-289 for one slice (2-D arrays)."""
    x = np.squeeze(x)
    x_rec = np.squeeze(x_rec)
    d = np.maximum(x - x_rec, 0) if keep_only_positive else np.abs(x - x_rec)
    if brainmask is not None:
        d = apply_brainmask(d, brainmask, erode)
    if prior_quantile is not None:
        d = np.where(x < prior_quantile, 0, d)
    return d


def dice(P, G):
    P = np.asarray(P).reshape(-1)
    G = np.asarray(G).reshape(-1)
    return (2 * np.sum(P * G)) / (np.sum(P) + np.sum(G))


def _xfrange(start, stop, step):
    i = 0
    while start + i * step < stop:
        yield start + i * step
        i += 1


def compute_dice_score(predictions, labels, granularity):
    """trainers/Metrics.py:138-162, restated verbatim in structure (greedy: recurse once per level into the
    interval before the first non-improving threshold)."""
    def inner(start, stop, decimal):
        ths, scs = [], []
        had = False
        if decimal == granularity:
            return ths, scs
        for i, t in enumerate(_xfrange(start, stop, 1.0 / (10.0 ** decimal))):
            s = dice(np.where(predictions > t, 1, 0), labels)
            if i >= 2 and s <= scs[i - 1] and not had:
                st, ss = inner(ths[i - 2], t, decimal + 1)
                ths.extend(st)
                scs.extend(ss)
                had = True
            scs.append(s)
            ths.append(t)
        return ths, scs

    ths, scs = inner(0, 1.0, 1)
    pairs = sorted(zip(ths, scs))
    ths, scs = list(zip(*pairs))
    return scs, ths


def best_dice(predictions, labels, granularity=5):
    scs, ths = compute_dice_score(predictions, labels, granularity)
    i = int(np.argmax(scs))
    return scs[i], ths[i]


def _sorted_counts(predictions, labels):
    p = np.asarray(predictions, np.float64).reshape(-1)
    y = np.asarray(labels).reshape(-1).astype(bool)
    order = np.argsort(-p, kind='mergesort')
    p, y = p[order], y[order]
    distinct = np.r_[np.nonzero(np.diff(p))[0], p.size - 1]
    tps = np.cumsum(y)[distinct].astype(np.float64)
    fps = (1 + distinct - tps).astype(np.float64)
    return tps, fps, y.sum()


def average_precision(predictions, labels):
    """sklearn.metrics.average_precision_score: sum_n (R_n - R_{n-1}) P_n over distinct thresholds."""
    tps, fps, npos = _sorted_counts(predictions, labels)
    prec = tps / (tps + fps)
    rec = tps / npos
    return float(np.sum(np.diff(np.r_[0.0, rec]) * prec))


def auroc(predictions, labels):
    """sklearn roc_curve + auc (trapezoid over distinct thresholds, origin prepended)."""
    tps, fps, npos = _sorted_counts(predictions, labels)
    nneg = np.asarray(labels).size - npos
    tpr = np.r_[0.0, tps / npos]
    fpr = np.r_[0.0, fps / nneg]
    return float(np.trapezoid(tpr, fpr))


def filter_3d_connected_components(volume, min_filled=7):
    """utils/Evaluation.py:113-127 with scipy: 26-connected components (skimage connectivity=3) whose hole-filled
    size is <= 7 voxels are removed.  UNPINNED (skimage is not installed here).  regionprops' filled_area fills holes
    with the FULL 3x3x3 structure (skimage/measure/_regionprops.py: `structure = np.ones((3,) * ndim)`), restated here."""
    import scipy.ndimage as ndi
    vol = np.array(volume, copy=True)
    cc, n = ndi.label(vol, structure=np.ones((3, 3, 3)))
    for lbl, sl in enumerate(ndi.find_objects(cc), start=1):
        if sl is None:
            continue
        region = cc[sl] == lbl
        if ndi.binary_fill_holes(region, structure=np.ones((3, 3, 3))).sum() <= min_filled:
            vol[sl][region] = 0
    return vol
