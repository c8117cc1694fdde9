"""utils/Evaluation.py — residual-map scoring driver (the numeric core of _evaluate :183-365 and evaluate :372-526;
PNG / PDF / NIfTI export dropped).  Slices of a volume are reconstructed in ONE batched call (the reference runs one
sess.run per slice, :246-250), the residual map + brain mask + hyper-intensity prior run in the HIP residual kernel
(uad_residual), erosion and the 5x5x5 median use the same scipy.ndimage calls as the reference (:84-89, :108-110)."""
import time

import numpy as np
import scipy.ndimage

from ..trainers import Metrics


def should(dictionary, key):
    return key in dictionary and dictionary[key]


def erode_brainmask(brainmask):
    strel = scipy.ndimage.generate_binary_structure(2, 1)
    return scipy.ndimage.binary_erosion(np.squeeze(brainmask), structure=strel, iterations=12)


def apply_3d_median_filter(volume, kernelsize=5):
    return scipy.ndimage.median_filter(volume, (kernelsize, kernelsize, kernelsize))


def evaluate_volume(model, volume, brainmasks, options, eps=0.0):
    """volume [S,H,W] in [0,1]; brainmasks [S,H,W].  Returns the post-processed residual sub-volume [S,H,W] and
    per-slice l1 reconstruction errors (utils/Evaluation.py:223-312)."""
    S = volume.shape[0]
    prior = np.quantile(volume, 0.9) if should(options, 'applyHyperIntensityPrior') else None
    masks = np.stack([erode_brainmask(b) if should(options, 'erodeBrainmask') else np.squeeze(b) for b in brainmasks])
    x = volume[..., None].astype(np.float32)
    bs = model.config.batchsize
    diffs = np.zeros(volume.shape, np.float64)
    l1 = np.zeros(S)
    for s0 in range(0, S, bs):
        xb = x[s0:s0 + bs]
        rec = model.reconstruct(xb, eps=eps)['reconstruction']
        d, e = model.engine.residual(xb, rec, masks[s0:s0 + bs, ..., None].astype(np.float32),
                                     pos_only=should(options, 'keepOnlyPositiveResiduals'), prior_thresh=prior)
        diffs[s0:s0 + bs] = d.cpu().numpy()[..., 0]
        l1[s0:s0 + bs] = e.cpu().numpy()
    if should(options, 'medianFiltering'):
        diffs = apply_3d_median_filter(diffs)
    return diffs, l1


def evaluate(volumes, labels, brainmasks, model, options, eps=0.0):
    """volumes/labels/brainmasks: lists of [S,H,W] arrays (one per patient).  Returns the reference's evalPC scalars
    (utils/Evaluation.py:416-461): diff_AUC, diff_AUPRC, bestDiceScore, bestThreshold, per-patient Dice."""
    _time = {'evaluation': time.time()}
    diffs = [evaluate_volume(model, v, b, options, eps)[0] for v, b in zip(volumes, brainmasks)]
    d_all = np.concatenate([d.flatten() for d in diffs])
    l_all = np.concatenate([np.asarray(l).flatten() for l in labels])
    ev = {}
    ev['diff_AUC'], _, _, _ = Metrics.compute_roc(d_all, l_all.astype(bool))
    ev['diff_AUPRC'], _, _, _ = Metrics.compute_prc(d_all, l_all.astype(bool))
    ev['bestDiceScore'], ev['bestThreshold'] = Metrics.compute_dice_curve_recursive(d_all, l_all, granularity=10)
    thr = ev['bestThreshold'] if options.get('threshold', 'bestdice') == 'bestdice' else options['threshold']
    ev['Dice'] = [Metrics.dice((d > thr).astype(np.int64), np.asarray(l)) for d, l in zip(diffs, labels)]
    _time['evaluation'] = time.time() - _time['evaluation']
    ev['time'] = _time
    return ev
