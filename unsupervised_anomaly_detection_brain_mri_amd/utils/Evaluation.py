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


def apply_brainmask(x, brainmask, erode=True):
    """utils/Evaluation.py:84-89 (host form): x * (12x cross-eroded) brain mask."""
    bm = erode_brainmask(brainmask) if erode else np.squeeze(brainmask)
    return np.multiply(bm, np.squeeze(x))


def squash_intensities(img):
    """utils/Evaluation.py:70-74: logistic squash of reconstruction errors, 2 * (sigmoid(100 x) - 0.5)."""
    return 2.0 * (1.0 / (1.0 + np.exp(-100.0 * np.asarray(img))) - 0.5)


def postprocess_slice(x, x_rec, slice_skullmap=None):
    """utils/Evaluation.py:92-105: positive residual inside the eroded skull map, zeroed where x < 0.6 (fixed prior)."""
    x, x_rec = np.squeeze(x), np.squeeze(x_rec)
    mask = np.ones(x.shape) if slice_skullmap is None else erode_brainmask(slice_skullmap)
    d = np.multiply(mask, x - x_rec)
    d[d < 0] = 0
    d[x < 0.6] = 0
    return d


def filter_3d_connected_components(volume, max_voxels=7):
    """utils/Evaluation.py:113-127: zero every 26-connected component of at most 7 voxels (4-D input is folded to 3-D like the
    reference).  scikit-image is not available here: scipy.ndimage.label with the full 3x3x3 structure is the same labelling."""
    volume = np.asarray(volume)
    shape = volume.shape
    v3 = volume.reshape(shape[0] * shape[1], shape[2], shape[3]) if volume.ndim > 3 else volume
    lab, n = scipy.ndimage.label(v3 != 0, structure=np.ones((3, 3, 3), bool))
    if n:
        sizes = np.bincount(lab.ravel(), minlength=n + 1)
        small = sizes <= max_voxels
        small[0] = False
        v3 = np.where(small[lab], 0, v3)
    return v3.reshape(shape)


def postprocess_volume(volume):
    """utils/Evaluation.py:175-180: 5x5x5 median, then the connected-component filter."""
    return filter_3d_connected_components(apply_3d_median_filter(volume))


def compute_detection_rate(predicted_volume, groundtruth_volume):
    """utils/Evaluation.py:130-172: lesion-wise (TP, FP, FN) counted on 26-connected components in chunks of 20 slices: a TP is a
    component of prediction AND ground truth; predicted components under 8 voxels are ignored; predicted / true components that
    contain no TP are FPs / FNs."""
    full = np.ones((3, 3, 3), bool)
    pred = np.asarray(predicted_volume) != 0
    gt = np.asarray(groundtruth_volume) != 0
    tps = fps = fns = 0
    for s0 in range(0, gt.shape[0], 20):
        p, g = pred[s0:s0 + 20], gt[s0:s0 + 20]
        li, ni = scipy.ndimage.label(p & g, structure=full)
        lp, npred = scipy.ndimage.label(p, structure=full)
        lg, ng = scipy.ndimage.label(g, structure=full)
        psz = np.bincount(lp.ravel(), minlength=npred + 1)
        keep_p = psz >= 8
        keep_p[0] = False
        hit_p = np.zeros(npred + 1, bool)
        hit_g = np.zeros(ng + 1, bool)
        if ni:
            first = scipy.ndimage.find_objects(li)
            for k, sl in enumerate(first, start=1):
                idx = np.argwhere(li[sl] == k)[0]                 # the reference takes the component's first coordinate
                z, y, x = (sl[0].start + idx[0], sl[1].start + idx[1], sl[2].start + idx[2])
                hit_p[lp[z, y, x]] = True
                hit_g[lg[z, y, x]] = True
        tps += ni
        fps += int(np.count_nonzero(keep_p & ~hit_p))
        fns += int(np.count_nonzero(~hit_g[1:]))
    return tps, fps, fns


def determine_threshold_on_labeled_patients(volumes, labels, brainmasks, model, options, eps=0.0):
    """utils/Evaluation.py:529-570: the Dice-optimal threshold of the residual maps of labelled VALIDATION patients
    (granularity-10 sweep), on the device path.  Returns (bestDiceScore, bestThreshold)."""
    diffs = [evaluate_volume(model, v, b, options, eps, device_out=True)[0] for v, b in zip(volumes, brainmasks)]
    sc = model.engine.scores(torch.cat([d.reshape(-1) for d in diffs]), np.concatenate([np.asarray(l).flatten() for l in labels]))
    best = Metrics.compute_dice_curve_recursive_device(sc, granularity=10)
    sc.close()
    return best


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
    K = int(options.get('numMonteCarloSamples') or 0)
    var = torch.empty_like(diffs) if K > 1 else None
    for s0 in range(0, S, bs):
        xb = x[s0:s0 + bs]
        if K > 1:
            # Monte-Carlo dropout (utils/Evaluation.py:238-266): K stochastic passes, the residual is taken against the mean of the
            # brain-masked reconstructions, their per-pixel variance is the epistemic uncertainty
            recs = torch.stack([torch.from_numpy(model.reconstruct(xb, dropout=True)['reconstruction']).to(eng.device) for _ in range(K)])
            rec, v = eng.mc_stats(recs, masks[s0:s0 + bs, ..., None])
            var[s0:s0 + bs] = v[..., 0]
        else:
            rec = model.reconstruct(xb, eps=eps)['reconstruction']
        d, e = eng.residual(xb, rec, masks[s0:s0 + bs, ..., None], pos_only=should(options, 'keepOnlyPositiveResiduals'),
                            prior_thresh=prior)
        diffs[s0:s0 + bs] = d[..., 0]
        l1[s0:s0 + bs] = e.cpu().numpy()
    if var is not None:
        model.last_epistemic_variance = var
    if should(options, 'medianFiltering'):
        diffs = eng.median3d(diffs, 5)
    return (diffs if device_out else diffs.cpu().numpy().astype(np.float64)), l1


def evaluate(volumes, labels, brainmasks, model, options, eps=0.0):
    """volumes/labels/brainmasks: lists of [S,H,W] arrays (one per patient).  Returns the reference's evalPC scalars
    (utils/Evaluation.py:416-470): diff_AUC, diff_AUPRC, bestDiceScore, bestThreshold, DiceScore, DiceScorePerPatient,
    PrecisionPerPatient, RecallPerPatient (after the small-component filter)."""
    _time = {'evaluation': time.time()}
    diffs, variances = [], []
    for v, b in zip(volumes, brainmasks):
        diffs.append(evaluate_volume(model, v, b, options, eps, device_out=True)[0])
        if int(options.get('numMonteCarloSamples') or 0) > 1:
            variances.append(model.last_epistemic_variance.cpu().numpy())
    d_all = torch.cat([d.reshape(-1) for d in diffs])
    l_all = np.concatenate([np.asarray(l).flatten() for l in labels])
    sc = model.engine.scores(d_all, l_all)
    ev = {'diff_AUC': sc.auroc, 'diff_AUPRC': sc.auprc}
    ev['bestDiceScore'], ev['bestThreshold'] = Metrics.compute_dice_curve_recursive_device(sc, granularity=10)
    sc.close()
    thr = ev['bestThreshold'] if options.get('threshold', 'bestdice') == 'bestdice' else options['threshold']
    ev['thresholdType'] = options.get('threshold', 'bestdice')
    # utils/Evaluation.py:452-470: threshold, drop the <= 7-voxel components of the STACKED patient volume (device flood-fill
    # filter), then the overall and per-patient Dice / precision / recall
    stacked = torch.cat(diffs, dim=0)
    pred = model.engine.cc_filter((stacked > float(thr)).to(torch.float32), 7).cpu().numpy() > 0
    gts = [np.asarray(l).reshape(d.shape).astype(bool) for d, l in zip(diffs, labels)]
    ev['DiceScore'] = Metrics.dice(pred, np.concatenate(gts, axis=0))
    ev['DiceScorePerPatient'], ev['PrecisionPerPatient'], ev['RecallPerPatient'] = [], [], []
    s0 = 0
    with np.errstate(divide='ignore', invalid='ignore'):
        for d, g in zip(diffs, gts):
            sub = pred[s0:s0 + d.shape[0]]
            s0 += d.shape[0]
            ev['DiceScorePerPatient'].append(Metrics.dice(sub, g))
            ev['PrecisionPerPatient'].append(Metrics.precision(sub, g))
            ev['RecallPerPatient'].append(Metrics.recall(sub, g))
    ev['Dice'] = ev['DiceScorePerPatient']
    if variances:
        # utils/Evaluation.py:404-408: histogram of the epistemic variances (50 bins, 1e-5 .. their 99.8th percentile)
        ev['epistemic_variance'] = np.concatenate(variances, axis=0)
        pos = ev['epistemic_variance'][ev['epistemic_variance'] >= 0]
        hi = float(np.percentile(pos, 99.8))
        # (the reference's np.histogram raises when every variance is below 1e-5; an empty histogram is returned here instead)
        ev['uncertaintyHistogram'] = (np.histogram(ev['epistemic_variance'], bins=50, range=(1e-5, hi))[0] if hi > 1e-5 else np.zeros(50, np.int64)).tolist()
    _time['evaluation'] = time.time() - _time['evaluation']
    ev['time'] = _time
    return ev
