"""CPU: the reference's evaluation entry points (utils/Evaluation.py:183-365 `_evaluate`, :372-526 `evaluate(datasetPC, model, options,
epoch, description)`, :529-570 `determine_threshold_on_labeled_patients`) and run.py's evaluate_optimal / evaluate_with_threshold flow
(run.py:58-116), driven through a dataset object that exposes exactly the members the reference reads (`patients`, `get_patient_idx`,
`load_volume_and_groundtruth`, `num_batches`, `options.{sliceStart, sliceEnd, axis, sliceResolution}`) and a stand-in model whose
`engine` implements the device scoring ops with the (reference-pinned) scoring oracle -- so the host logic runs without a GPU."""
import os
import types

import numpy as np
import pytest
import scipy.ndimage
import torch

from oracle import scoring as osc
from unsupervised_anomaly_detection_brain_mri_amd.trainers import Metrics
from unsupervised_anomaly_detection_brain_mri_amd.utils import Evaluation
from unsupervised_anomaly_detection_brain_mri_amd.utils.default_config_setup import Dataset, get_datasets, get_options
from unsupervised_anomaly_detection_brain_mri_amd.utils.synthetic import SyntheticPatientDataset


class _HostScores:
    def __init__(self, p, y):
        self.p, self.y = np.asarray(p, np.float64).reshape(-1), np.asarray(y).reshape(-1).astype(bool)
        self.auroc, self.auprc, self.positives = osc.auroc(self.p, self.y), osc.average_precision(self.p, self.y), float(self.y.sum())

    def dice_at(self, thresholds):
        return np.array([osc.dice(self.p > t, self.y) for t in np.atleast_1d(thresholds)])

    def close(self):
        pass


class HostEvalEngine:
    """The _EvalOps surface of engine.Engine on torch CPU tensors, computed by oracle/scoring.py."""
    device = torch.device('cpu')

    def _dev(self, a):
        return a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, np.float32))

    def erode_cross(self, masks, iterations=12):
        return torch.from_numpy(np.stack([osc.binary_erosion_cross(m, iterations) for m in np.asarray(masks)]).astype(np.float32))

    def median3d(self, volume, ksize=5):
        return torch.from_numpy(osc.median_filter_3d(volume.numpy().astype(np.float64), ksize).astype(np.float32))

    def residual(self, x, x_rec, mask=None, pos_only=True, prior_thresh=None):
        x, r = np.asarray(x, np.float32), np.asarray(x_rec, np.float32)
        d = np.maximum(x - r, 0) if pos_only else np.abs(x - r)
        if mask is not None:
            d = d * mask.numpy()
        if prior_thresh is not None:
            d = np.where(x < np.float32(prior_thresh), 0, d)
        return torch.from_numpy(d.astype(np.float32)), torch.from_numpy(np.abs(x - r).reshape(len(x), -1).sum(1))

    def scores(self, predictions, labels):
        return _HostScores(predictions.numpy() if isinstance(predictions, torch.Tensor) else predictions, labels)

    def mc_stats(self, recs, mask=None):
        r = recs.numpy().astype(np.float64)
        if mask is not None:
            r = r * mask.numpy()
        return torch.from_numpy(r.mean(axis=0).astype(np.float32)), torch.from_numpy(r.var(axis=0).astype(np.float32))

    def cc_filter(self, volume, max_voxels=7):
        return torch.from_numpy(Evaluation.filter_3d_connected_components(volume.numpy(), max_voxels).astype(np.float32))


class BlurModel:
    """reconstruct() = a smoothed copy of the input: lesions (small, bright) leave a positive residual, healthy tissue does not."""

    def __init__(self, tmp, bs=5):
        self.engine = HostEvalEngine()
        self.config = types.SimpleNamespace(batchsize=bs)
        self.network = types.SimpleNamespace(__name__='blur_network')
        self.model_dir = 'Blur_dSynthetic'
        self.calls = []

    def reconstruct(self, x, dropout=False, eps=None):
        x = np.asarray(x, np.float32)
        self.calls.append(x.shape)
        rec = scipy.ndimage.uniform_filter(x, size=(1, 9, 9, 1))
        return {'reconstruction': rec, 'l1err': np.abs(x - rec).sum(), 'l2err': np.abs(x - rec).sum()}


def _opts(tmp_path, h=64):
    return get_options(batchsize=5, learningrate=1e-4, numEpochs=3, zDim=64, outputWidth=h, outputHeight=h, slices_start=0, slices_end=12,
                       config={'CHECKPOINTDIR': str(tmp_path / 'ck'), 'SAMPLEDIR': str(tmp_path / 'smp')})


def test_collect_patient_volume_follows_the_reference_slice_loop(tmp_path):
    ds = SyntheticPatientDataset(n_val=1, n_test=1, slices=14, native=80, h=64, w=64, seed=3, slice_start=2, slice_end=12)
    assert {'patients', 'get_patient_idx', 'load_volume_and_groundtruth', 'num_batches', 'options'} <= set(dir(ds))
    patient = ds.patients[ds.get_patient_idx('TEST')[0]]
    x, seg, skull, prior, idx = Evaluation.collect_patient_volume(ds, patient, patient['filtered_files'][0], _opts(tmp_path))
    nii, nii_seg, nii_skull = ds.load_volume_and_groundtruth(patient['filtered_files'][0], patient)
    assert idx == list(range(2, 12)) and x.shape == (10, 64, 64) and prior == pytest.approx(np.quantile(nii.data, 0.9))
    # the statements of utils/Evaluation.py:223-232 for one slice
    s = 7
    zf = tuple(i / j for (i, j) in zip((64, 64), nii.get_slice(s, 'axial').shape))
    np.testing.assert_array_equal(x[s - 2], scipy.ndimage.zoom(nii.get_slice(s, 'axial'), zf))
    np.testing.assert_array_equal(seg[s - 2], scipy.ndimage.zoom(nii_seg.get_slice(s, 'axial').astype(int), zf, mode='nearest'))
    np.testing.assert_array_equal(skull[s - 2], scipy.ndimage.zoom(nii_skull.get_slice(s, 'axial').astype(int), zf, mode='nearest'))
    # too thin a volume is skipped (:210-211)
    ds.options.sliceEnd = 40
    assert Evaluation.collect_patient_volume(ds, patient, patient['filtered_files'][0], _opts(tmp_path)) is None


def test_evaluate_with_the_reference_signature(tmp_path):
    opt = _opts(tmp_path)
    ds = SyntheticPatientDataset(n_val=2, n_test=2, slices=12, native=80, h=64, w=64, seed=1, slice_start=0, slice_end=12)
    model = BlurModel(tmp_path)
    ev = Evaluation.evaluate(ds, model, opt, epoch='3', description='unit')
    # directory layout and files (:380-395, :519-526)
    assert os.path.basename(os.path.dirname(ev['eval_dir'])) == 'Blur_dSynthetic' and ev['eval_dir'].endswith('-unit')
    assert os.path.basename(ev['eval_dir']).startswith('eval-3-') and os.path.isdir(os.path.join(ev['eval_dir'], 'samples_test_PC'))
    for f in ('evalPC.npy', 'evalPC.txt', 'rocPC.npy', 'prcPC.npy'):
        assert os.path.isfile(os.path.join(ev['eval_dir'], f)), f
    saved = np.load(os.path.join(ev['eval_dir'], 'evalPC.npy'), allow_pickle=True).item()
    assert saved['diff_AUPRC'] == ev['diff_AUPRC'] and len(saved['DiceScorePerPatient']) == 2
    # batched device-style reconstruct calls, 12 slices per patient in batches of 5
    assert model.calls == [(5, 64, 64, 1), (5, 64, 64, 1), (2, 64, 64, 1)] * 2
    # the numbers equal the reference's slice-by-slice recipe written out with its own helpers
    diffs, labs = [], []
    for k in ds.get_patient_idx('TEST'):
        p = ds.patients[k]
        x, seg, skull, prior, _ = Evaluation.collect_patient_volume(ds, p, p['filtered_files'][0], opt)
        sub = np.zeros_like(x)
        for s in range(len(x)):
            xs = x[s].astype(np.float32)
            rec = BlurModel(tmp_path).reconstruct(xs[None, ..., None])['reconstruction'][0, ..., 0]
            d = Evaluation.apply_brainmask(np.maximum(xs - rec, 0), skull[s], erode=True)
            d[xs < np.float32(prior)] = 0
            sub[s] = d
        diffs.append(Evaluation.apply_3d_median_filter(sub)); labs.append(seg)
    dd, ll = np.concatenate(diffs), np.concatenate(labs)
    auc = Metrics.compute_roc(dd.flatten(), ll.astype(bool).flatten())[0]
    auprc = Metrics.compute_prc(dd.flatten(), ll.astype(bool).flatten())[0]
    best, thr = Metrics.compute_dice_curve_recursive(dd.flatten(), ll.flatten(), granularity=10)
    assert ev['diff_AUC'] == pytest.approx(auc, abs=1e-6) and ev['diff_AUPRC'] == pytest.approx(auprc, abs=1e-6)
    assert ev['bestDiceScore'] == pytest.approx(best, abs=1e-6) and ev['bestThreshold'] == pytest.approx(thr, abs=1e-9)
    pred = Evaluation.filter_3d_connected_components(np.squeeze(dd > thr))
    assert ev['DiceScore'] == pytest.approx(Metrics.dice(pred, ll), abs=1e-9)
    assert 0.0 < ev['diff_AUPRC'] <= 1.0 and 0.0 < ev['diff_AUC'] <= 1.0


def test_evaluate_keeps_the_monte_carlo_dropout_outputs(tmp_path):
    """numMonteCarloSamples > 1 through the reference-signature entry point: K dropout passes per batch, and the epistemic variance + its
    50-bin histogram (utils/Evaluation.py:238-266, 404-408) reach the result exactly as they do through evaluate_arrays."""
    class NoisyBlur(BlurModel):
        def __init__(self, tmp, bs=5):
            super().__init__(tmp, bs)
            self.rng = np.random.default_rng(5)

        def reconstruct(self, x, dropout=False, eps=None):
            out = super().reconstruct(x, dropout, eps)
            if dropout:
                out['reconstruction'] = out['reconstruction'] + self.rng.normal(0, 0.05, out['reconstruction'].shape).astype(np.float32)
            return out

    opt = dict(_opts(tmp_path), numMonteCarloSamples=3)
    ds = SyntheticPatientDataset(n_val=1, n_test=2, slices=12, native=80, h=64, w=64, seed=1, slice_start=0, slice_end=12)
    model = NoisyBlur(tmp_path)
    ev = Evaluation.evaluate(ds, model, opt, epoch='1')
    assert ev['epistemic_variance'].shape == (24, 64, 64) and ev['epistemic_variance'].max() > 0
    assert len(ev['uncertaintyHistogram']) == 50 and sum(ev['uncertaintyHistogram']) > 0
    assert len(model.calls) == 3 * 3 * 2                      # K passes x 3 batches x 2 patients
    saved = np.load(os.path.join(ev['eval_dir'], 'evalPC.npy'), allow_pickle=True).item()
    assert saved['uncertaintyHistogram'] == ev['uncertaintyHistogram'] and 'epistemic_variance' not in saved
    ev0 = Evaluation.evaluate(ds, BlurModel(tmp_path), _opts(tmp_path), epoch='1')
    assert 'epistemic_variance' not in ev0 and 'uncertaintyHistogram' not in ev0


def test_threshold_on_validation_patients_and_fixed_threshold_evaluation(tmp_path):
    opt = _opts(tmp_path)
    ds = SyntheticPatientDataset(n_val=2, n_test=1, slices=12, native=64, h=64, w=64, seed=9, slice_start=0, slice_end=12)
    model = BlurModel(tmp_path)
    best, thr = Evaluation.determine_threshold_on_labeled_patients([ds], model, opt, description='VAL')
    assert 0.0 < best <= 1.0 and 0.0 < thr < 1.0
    assert Evaluation.determine_threshold_on_labeled_patients(ds, model, opt) == (best, thr)          # a single dataset is accepted too (:548-549)
    opt['threshold'] = thr
    ev = Evaluation.evaluate(ds, model, opt, epoch='3', description=f'VALthresh_{thr}')
    assert ev['thresholdType'] == thr and len(ev['DiceScorePerPatient']) == 1


def test_run_py_evaluation_flow(tmp_path, monkeypatch):
    import run
    opt = _opts(tmp_path)
    model = BlurModel(tmp_path)
    seen = []
    real = Evaluation.evaluate
    monkeypatch.setattr(Evaluation, 'evaluate', lambda ds, m, o, **kw: (seen.append((kw['description'], o['applyHyperIntensityPrior'], o['threshold'])), real(ds, m, o, **kw))[1])
    opt['applyHyperIntensityPrior'] = True
    ev = run.evaluate_optimal(model, opt, Dataset.MSLUB)
    assert seen[-1] == ('SyntheticPatientDataset-MSLUB_upperbound_bestdice_wPrior', True, 'bestdice') and 'diff_AUPRC' in ev
    run.evaluate_with_threshold(model, opt, 0.05, Dataset.MSISBI2015)
    assert seen[-1] == ('SyntheticPatientDataset-MSISBI2015-VALthresh_0.05', False, 0.05)
    # the stand-in lesion sets differ per Dataset member and have the patient duck-type
    a, b = get_datasets(opt, Dataset.BRAINWEB)[1], get_datasets(opt, Dataset.MSLUB)[1]
    assert len(a.get_patient_idx('VAL')) == 2 and len(a.get_patient_idx('TEST')) == 2
    assert a.load_volume_and_groundtruth(None, a.patients[0])[0].shape() != b.load_volume_and_groundtruth(None, b.patients[0])[0].shape()
    with pytest.raises(ValueError):
        get_datasets(opt, 'Brainweb')
