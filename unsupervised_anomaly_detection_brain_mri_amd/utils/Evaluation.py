"""utils/Evaluation.py — residual-map scoring driver: the reference's entry points `evaluate(datasetPC, model, options, epoch,
description)` (:372-526), `_evaluate(datasetObj, modelObj, sampleDir, options, split)` (:183-365) and
`determine_threshold_on_labeled_patients(dataset_pc, model, options, epoch, description)` (:529-570) on the dataset duck-type
(`patients`, `get_patient_idx`, `load_volume_and_groundtruth`, `options.{sliceStart, sliceEnd, axis, sliceResolution}`), over the
array-level core `evaluate_arrays` / `evaluate_volume` (PNG / PDF / NIfTI export dropped).  Everything between the reconstruction and the scalar metrics stays on the device:
slices of a volume are reconstructed in ONE batched call (the reference runs one sess.run per slice, :246-250), the brain
masks are eroded (uad_erode_cross), residual map + mask + hyper-intensity prior come from uad_residual, the 5x5x5 median is
uad_median3d, and AUROC / AUPRC / the Dice threshold sweep read one device sort of all voxels (uad_scores_*).
erode_brainmask / apply_3d_median_filter keep the reference's scipy calls for host-side use.

Multi-GPU (SURVEY.md 8e "Inference / config 5"): when torch.distributed is initialised the per-patient loop is SHARDED BY PATIENT (patient k of
the walk -> rank k mod world; the 5x5x5 median couples slices of one patient only, so a patient never straddles ranks), every rank
reconstructs and post-processes its patients, and the finished residual volumes are exchanged (`_sharded_map`: one broadcast per patient
from its owner -- RCCL over xGMI under backend nccl -- plus one all_gather_object of the small host-side records) so that EVERY rank holds the
same, identically ordered patient list as a single process; the global AUROC / AUPRC / Dice are then computed from that list exactly as
before (bit-identical to world 1: tests/test_eval_sharded_gloo.py).  Only rank 0 writes evalPC.npy / evalPC.txt."""
import time

import numpy as np
import scipy.ndimage
import torch

from ..trainers import Metrics


def should(dictionary, key):
    return key in dictionary and dictionary[key]


def _dp_rank_world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _sharded_map(n_items, fn, device):
    """Patient-sharded evaluation.  fn(k) -> (residual volume as a float32 tensor on `device`, picklable host record) or None for item k.
    Item k is computed by rank k mod world; returns {k: (tensor, record)} over all items that produced something, identical on every rank
    (tensors broadcast from their owner, records through one all_gather_object).  world == 1: a plain loop, no collective."""
    rank, world = _dp_rank_world()
    local = {}
    for k in range(rank, n_items, world):
        r = fn(k)
        if r is not None:
            local[k] = (r[0].to(torch.float32).contiguous(), r[1])
    if world == 1:
        return local
    import torch.distributed as dist
    meta = [None] * world
    dist.all_gather_object(meta, {k: (tuple(t.shape), rec) for k, (t, rec) in local.items()})
    out = {}
    for owner, m in enumerate(meta):
        for k in m:
            assert k % world == owner, 'patient shard bookkeeping broke'
    for k in sorted(k for m in meta for k in m):
        owner = k % world
        shape, rec = meta[owner][k]
        t = local[k][0] if owner == rank else torch.empty(shape, device=device, dtype=torch.float32)
        dist.broadcast(t, src=owner)
        out[k] = (t, rec)
    return out


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


def determine_threshold_on_arrays(volumes, labels, brainmasks, model, options, eps=None):
    """utils/Evaluation.py:529-570: the Dice-optimal threshold of the residual maps of labelled VALIDATION patients
    (granularity-10 sweep), on the device path.  Returns (bestDiceScore, bestThreshold)."""
    got = _sharded_map(len(volumes), lambda k: (evaluate_volume(model, volumes[k], brainmasks[k], options, eps, device_out=True)[0], None),
                       model.engine.device)
    return _best_dice_of(model, [got[k][0] for k in sorted(got)], labels)


def _best_dice_of(model, diffs, labels):
    sc = model.engine.scores(torch.cat([d.reshape(-1) for d in diffs]), np.concatenate([np.asarray(l).flatten() for l in labels]))
    best = Metrics.compute_dice_curve_recursive_device(sc, granularity=10)
    sc.close()
    return best


def evaluate_volume(model, volume, brainmasks, options, eps=None, device_out=False, prior=None):
    """volume [S,H,W] in [0,1]; brainmasks [S,H,W].  Returns the post-processed residual sub-volume [S,H,W] (numpy, or the
    device tensor with device_out=True) and per-slice l1 reconstruction errors (utils/Evaluation.py:223-312).
    eps: None = z is sampled as the reference's graph does at evaluation too; 0.0 = the deterministic mode.
    prior: the hyper-intensity threshold (the reference takes the 0.9 quantile of the WHOLE loaded volume, :207); default: of `volume`."""
    S = volume.shape[0]
    eng = model.engine
    if not should(options, 'applyHyperIntensityPrior'):
        prior = None
    elif prior is None:
        prior = np.quantile(volume, 0.9)
    # a batched reconstruct() must equal the reference's slice-by-slice calls (:246-250): trainers whose reconstruct() is a batch mean
    # (ceVAE's input-gradient restoration) take per_slice=True
    rkw = {'per_slice': True} if getattr(model, 'RECONSTRUCT_PER_SLICE', False) else {}
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
            recs = torch.stack([torch.from_numpy(model.reconstruct(xb, dropout=True, **rkw)['reconstruction']).to(eng.device) for _ in range(K)])
            rec, v = eng.mc_stats(recs, masks[s0:s0 + bs, ..., None])
            var[s0:s0 + bs] = v[..., 0]
        else:
            rec = model.reconstruct(xb, eps=eps, **rkw)['reconstruction']
        d, e = eng.residual(xb, rec, masks[s0:s0 + bs, ..., None], pos_only=should(options, 'keepOnlyPositiveResiduals'),
                            prior_thresh=prior)
        diffs[s0:s0 + bs] = d[..., 0]
        l1[s0:s0 + bs] = e.cpu().numpy()
    if var is not None:
        model.last_epistemic_variance = var
    if should(options, 'medianFiltering'):
        diffs = eng.median3d(diffs, 5)
    return (diffs if device_out else diffs.cpu().numpy().astype(np.float64)), l1


def evaluate_arrays(volumes, labels, brainmasks, model, options, eps=None, priors=None):
    """volumes/labels/brainmasks: lists of [S,H,W] arrays (one per patient).  Returns the reference's evalPC scalars
    (utils/Evaluation.py:416-470): diff_AUC, diff_AUPRC, bestDiceScore, bestThreshold, DiceScore, DiceScorePerPatient,
    PrecisionPerPatient, RecallPerPatient (after the small-component filter)."""
    _time = {'evaluation': time.time()}
    mc = int(options.get('numMonteCarloSamples') or 0) > 1

    def one(k):
        d, l1 = evaluate_volume(model, volumes[k], brainmasks[k], options, eps, device_out=True, prior=None if priors is None else priors[k])
        return d, (l1, model.last_epistemic_variance.cpu().numpy() if mc else None)
    got = _sharded_map(len(volumes), one, model.engine.device)
    diffs = [got[k][0] for k in sorted(got)]
    l1s = [got[k][1][0] for k in sorted(got)]
    variances = [got[k][1][1] for k in sorted(got)] if mc else []
    ev = _score_diffs(model, diffs, labels, options, variances)
    l1all = np.concatenate(l1s) if l1s else np.zeros(0)
    # (trainers' l2err == l1err, sic: SURVEY.md A4)
    ev['l1reconstructionErrorMean'] = ev['l2reconstructionErrorMean'] = float(np.mean(l1all)) if l1all.size else 0.0
    ev['l1reconstructionErrorVariance'] = ev['l2reconstructionErrorVariance'] = float(np.var(l1all)) if l1all.size else 0.0
    _time['evaluation'] = time.time() - _time['evaluation']
    ev['time'] = _time
    return ev


def _score_diffs(model, diffs, labels, options, variances=None):
    """The metric tail of utils/Evaluation.py:416-500 on per-patient residual volumes (device tensors [S,H,W]) and label maps."""
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
    return ev


# ------------------------------------------------------------------------------------------------------------------------------
# The reference's entry points on the dataset duck-type
# ------------------------------------------------------------------------------------------------------------------------------
def _zoom_factor(resolution, shape):
    return tuple(i / j for (i, j) in zip(resolution, shape))


def collect_patient_volume(datasetObj, patient, nii_filename, options):
    """utils/Evaluation.py:205-232: load the volume, its ground truth and skull map; take slices sliceStart .. min(sliceEnd, #slices) along
    options.axis; zoom every slice to options.sliceResolution -- cubic spline for the image (scipy.ndimage.zoom default order 3), the same
    call with mode='nearest' for the integer label / skull maps, exactly as written there.  Returns (x [S,H,W] float64, seg [S,H,W] int,
    skullmap [S,H,W] int, prior_quantile of the whole loaded volume, slice indices) or None when the volume is too thin (:210-211)."""
    o = datasetObj.options
    nii, nii_seg, nii_skullmap = datasetObj.load_volume_and_groundtruth(nii_filename, patient)
    prior_quantile = np.quantile(nii.data, 0.9)
    if min(nii.shape()) < (o.sliceEnd - o.sliceStart):
        return None
    slice_start = o.sliceStart if o.sliceStart else 0
    n_ax = nii.num_slices_along_axis(o.axis)
    slice_end = min(o.sliceEnd, n_ax) if o.sliceEnd else n_ax
    xs, segs, skulls, idx = [], [], [], []
    for s in range(slice_start, slice_end):
        slice_data = nii.get_slice(s, o.axis)
        slice_seg = nii_seg.get_slice(s, o.axis).astype(int)
        slice_skullmap = nii_skullmap.get_slice(s, o.axis).astype(int)
        if o.sliceResolution is not None:
            zf = _zoom_factor(o.sliceResolution, slice_data.shape)
            slice_data = scipy.ndimage.zoom(slice_data, zf)
            slice_seg = scipy.ndimage.zoom(slice_seg, zf, mode="nearest")
            slice_skullmap = scipy.ndimage.zoom(slice_skullmap, zf, mode="nearest")
        xs.append(slice_data); segs.append(slice_seg); skulls.append(slice_skullmap); idx.append(s)
    return np.asarray(xs, np.float64), np.asarray(segs), np.asarray(skulls), float(prior_quantile), idx


def _evaluate(datasetObj, modelObj, sampleDir, options, split="TEST", eps=None):
    """utils/Evaluation.py:183-365.  Walks the split's patients through the dataset duck-type, reconstructs every patient's slice stack in
    batched device calls and returns (eval_dict, patients): eval_dict['diffs'] [P*S,H,W] post-processed residuals (device tensor under
    '_diffs_device' as well), 'labelmaps', 'x', 'l1reconstructionErrors' and their mean / variance.  The per-slice PNG dumps are not
    written (sampleDir is created like the reference does)."""
    import os
    os.makedirs(sampleDir, exist_ok=True)
    print("Testing {} samples...".format(datasetObj.num_batches(1, set=split)))
    patients = [datasetObj.patients[i] for i in datasetObj.get_patient_idx(split=split)]
    ev = {'x': [], 'labelmaps': [], 'l1reconstructionErrors': [], 'reconstructionTimes': []}
    diffs_dev = []
    variances = []               # numMonteCarloSamples > 1: every patient's epistemic-variance volume (utils/Evaluation.py:238-266,404-408)
    mc = int(options.get('numMonteCarloSamples') or 0) > 1
    used = []

    def one(p):
        patient = patients[p]
        files = patient['filtered_files']
        if type(files) is not list:
            files = [files]
        for nii_filename in files:                               # `if len(_eval_dict['diffs']) == 0` (:203): the first usable file of a patient
            got = collect_patient_volume(datasetObj, patient, nii_filename, options)
            if got is None:
                continue
            x, seg, skull, prior_q, _ = got
            t0 = time.time()
            d, l1 = evaluate_volume(modelObj, x, skull, options, eps, device_out=True, prior=prior_q)
            rec = {'x': x, 'seg': seg, 'l1': list(l1), 'time': (time.time() - t0) / max(len(x), 1),
                   'var': modelObj.last_epistemic_variance.cpu().numpy() if mc else None}
            return d, rec
        return None
    # patients are sharded over the ranks of an initialised process group (module docstring); every rank ends up with the full, ordered list
    got = _sharded_map(len(patients), one, modelObj.engine.device)
    for p in sorted(got):
        d, rec = got[p]
        ev['reconstructionTimes'].append(rec['time'])
        diffs_dev.append(d)
        if mc:
            variances.append(rec['var'])
        ev['x'].append(rec['x']); ev['labelmaps'].append(rec['seg']); ev['l1reconstructionErrors'] += rec['l1']
        used.append(patients[p])
    print("Done.")
    ev['_diffs_device'] = diffs_dev
    ev['_variances'] = variances
    ev['diffs'] = np.concatenate([d.cpu().numpy().astype(np.float64) for d in diffs_dev], axis=0) if diffs_dev else np.zeros((0,))
    ev['x'] = np.concatenate(ev['x'], axis=0) if ev['x'] else np.zeros((0,))
    ev['labelmaps'] = np.concatenate(ev['labelmaps'], axis=0) if ev['labelmaps'] else np.zeros((0,))
    e = np.asarray(ev['l1reconstructionErrors'], np.float64)
    ev['l1reconstructionErrorMean'] = ev['l2reconstructionErrorMean'] = float(e.mean()) if e.size else 0.0
    ev['l1reconstructionErrorVariance'] = ev['l2reconstructionErrorVariance'] = float(e.var()) if e.size else 0.0
    ev['reconstructionTimes'] = float(np.mean(ev['reconstructionTimes'])) if ev['reconstructionTimes'] else 0.0
    return ev, used


def _eval_dir(model, options, epoch, description):
    import os
    d = os.path.join(options['train']['samplesDir'], model.network.__name__, model.model_dir,
                     'eval-' + str(epoch) + '-' + time.strftime('%Y-%m-%d_%H-%M-%S'))
    if description is not None:
        d += "-" + str(description)
    os.makedirs(d, exist_ok=True)
    return d


def evaluate(datasetPC, gan, options, epoch='last', description=None, eps=None):
    """utils/Evaluation.py:372-526 with the reference's signature: evaluates the TEST patients of `datasetPC`, writes evalPC.npy / evalPC.txt
    (+ rocPC.npy / prcPC.npy when options['exportROC'] / ['exportPRC']) under <SAMPLEDIR>/<network>/<model_dir>/eval-<epoch>-<timestamp>[-
    <description>]/ and -- unlike the reference, which returns None -- hands the scalar dictionary back."""
    import os
    t_all = time.time()
    rank0 = _dp_rank_world()[0] == 0          # every rank scores the same gathered patient list; rank 0 alone writes the files
    eval_dir = _eval_dir(gan, options, epoch, description)
    sample_dir = os.path.join(eval_dir, 'samples_test_PC')
    eval_pc, patients_pc = _evaluate(datasetPC, gan, sample_dir, options, split="TEST", eps=eps)
    diffs = eval_pc.pop('_diffs_device')
    variances = eval_pc.pop('_variances')
    labels = [eval_pc['labelmaps'][sum(d.shape[0] for d in diffs[:k]):sum(d.shape[0] for d in diffs[:k + 1])] for k in range(len(diffs))]
    ev = _score_diffs(gan, diffs, labels, options, variances)
    for k in ('l1reconstructionErrorMean', 'l1reconstructionErrorVariance', 'l2reconstructionErrorMean', 'l2reconstructionErrorVariance',
              'reconstructionTimes'):
        ev[k] = eval_pc[k]
    if rank0 and (should(options, 'exportROC') or should(options, 'exportPRC')):
        flat_l = eval_pc['labelmaps'].astype(bool).flatten()
        if should(options, 'exportROC'):
            _, fpr, tpr, th = Metrics.compute_roc(eval_pc['diffs'].flatten(), flat_l)
            np.save(os.path.join(eval_dir, 'rocPC.npy'), {"fpr": fpr, "tpr": tpr, "threshs": th}, allow_pickle=True)
        if should(options, 'exportPRC'):
            _, pr, rc, th = Metrics.compute_prc(eval_pc['diffs'].flatten(), flat_l)
            np.save(os.path.join(eval_dir, 'prcPC.npy'), {"precisions": pr, "recalls": rc, "threshs": th}, allow_pickle=True)
    ev['time'] = {'evaluation': time.time() - t_all}
    out = {k: v for k, v in ev.items() if k not in ('epistemic_variance',)}
    if rank0:
        np.save(os.path.join(eval_dir, 'evalPC.npy'), out)
        with open(os.path.join(eval_dir, 'evalPC.txt'), "w") as f:
            f.write(str(out))
    ev['eval_dir'] = eval_dir
    return ev


def determine_threshold_on_labeled_patients(dataset_pc, model, options, epoch='last', description=None, eps=None):
    """utils/Evaluation.py:529-570: the Dice-optimal threshold (granularity-10 sweep) of the residual maps of the VAL patients of one
    dataset or a list of datasets.  Returns (bestDiceScore, bestThreshold)."""
    import os
    eval_dir = _eval_dir(model, options, epoch, description)
    sample_dir = os.path.join(eval_dir, 'samples_val_PC')
    if not isinstance(dataset_pc, list):
        dataset_pc = [dataset_pc]
    diffs, labels = [], []
    for ds in dataset_pc:
        e, _ = _evaluate(ds, model, sample_dir, options, split="VAL", eps=eps)
        diffs += e['_diffs_device']
        labels.append(e['labelmaps'])
    print("Computing DICE curve for Lesion Validation samples")
    return _best_dice_of(model, diffs, labels)
