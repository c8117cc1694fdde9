"""utils/Evaluation.py — residual-map scoring driver (the numeric core of _evaluate :183-365 and evaluate :372-526;
PNG / PDF / NIfTI export dropped).  Everything between the reconstruction and the scalar metrics stays on the device:
slices of a volume are reconstructed in ONE batched call (the reference runs one sess.run per slice, :246-250), the brain
masks are eroded (uad_erode_cross), residual map + mask + hyper-intensity prior come from uad_residual, the 5x5x5 median is
uad_median3d, and AUROC / AUPRC / the Dice threshold sweep read one device sort of all voxels (uad_scores_*).
erode_brainmask / apply_3d_median_filter keep the reference's scipy calls for host-side use."""
import time

import numpy as np
import scipy.ndimage
import torch

from ..trainers import Metrics


def should(dictionary, key):
    return key in dictionary and dictionary[key]


def erode_brainmask(brainmask):
    strel = scipy.ndimage.generate_binary_structure(2, 1)
    return scipy.ndimage.binary_erosion(np.squeeze(brainmask), structure=strel, iterations=12)


def apply_3d_median_filter(volume, kernelsize=5):
    return scipy.ndimage.median_filter(volume, (kernelsize, kernelsize, kernelsize))


def evaluate_volume(model, volume, brainmasks, options, eps=0.0, device_out=False):
    """volume [S,H,W] in [0,1]; brainmasks [S,H,W].  Returns the post-processed residual sub-volume [S,H,W] (numpy, or the
    device tensor with device_out=True) and per-slice l1 reconstruction errors (utils/Evaluation.py:223-312)."""
    S = volume.shape[0]
    eng = model.engine
    prior = np.quantile(volume, 0.9) if should(options, 'applyHyperIntensityPrior') else None
    bm = np.stack([np.squeeze(b) for b in brainmasks]).astype(np.float32)
    masks = eng.erode_cross(bm, 12) if should(options, 'erodeBrainmask') else eng._dev(bm)
    x = volume[..., None].astype(np.float32)
    bs = model.config.batchsize
    diffs = torch.empty((S,) + volume.shape[1:], device=eng.device, dtype=torch.float32)
    l1 = np.zeros(S)
    for s0 in range(0, S, bs):
        xb = x[s0:s0 + bs]
        rec = model.reconstruct(xb, eps=eps)['reconstruction']
        d, e = eng.residual(xb, rec, masks[s0:s0 + bs, ..., None], pos_only=should(options, 'keepOnlyPositiveResiduals'),
                            prior_thresh=prior)
        diffs[s0:s0 + bs] = d[..., 0]
        l1[s0:s0 + bs] = e.cpu().numpy()
    if should(options, 'medianFiltering'):
        diffs = eng.median3d(diffs, 5)
    return (diffs if device_out else diffs.cpu().numpy().astype(np.float64)), l1


def evaluate(volumes, labels, brainmasks, model, options, eps=0.0):
    """volumes/labels/brainmasks: lists of [S,H,W] arrays (one per patient).  Returns the reference's evalPC scalars
    (utils/Evaluation.py:416-461): diff_AUC, diff_AUPRC, bestDiceScore, bestThreshold, per-patient Dice."""
    _time = {'evaluation': time.time()}
    diffs = [evaluate_volume(model, v, b, options, eps, device_out=True)[0] for v, b in zip(volumes, brainmasks)]
    d_all = torch.cat([d.reshape(-1) for d in diffs])
    l_all = np.concatenate([np.asarray(l).flatten() for l in labels])
    sc = model.engine.scores(d_all, l_all)
    ev = {'diff_AUC': sc.auroc, 'diff_AUPRC': sc.auprc}
    ev['bestDiceScore'], ev['bestThreshold'] = Metrics.compute_dice_curve_recursive_device(sc, granularity=10)
    sc.close()
    thr = ev['bestThreshold'] if options.get('threshold', 'bestdice') == 'bestdice' else options['threshold']
    ev['Dice'] = [Metrics.dice((d.cpu().numpy().astype(np.float64) > thr).astype(np.int64), np.asarray(l)) for d, l in zip(diffs, labels)]
    _time['evaluation'] = time.time() - _time['evaluation']
    ev['time'] = _time
    return ev
