"""GPU parity of the residual-map scoring kernels against the oracle restatements that are pinned by the
reference-generated golden vectors (tests/golden/scoring_golden.npz): erosion and median are bit-exact (integer / order-statistic
work), the sort-based metrics agree to fp64 round-off."""
import os

import numpy as np
import pytest
import torch

from oracle import scoring as osc

pytestmark = pytest.mark.gpu

try:
    from unsupervised_anomaly_detection_brain_mri_amd.engine import Engine
    from unsupervised_anomaly_detection_brain_mri_amd.trainers import Metrics
except Exception:
    Engine = None

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'scoring_golden.npz'))


@pytest.fixture(scope='module')
def eng():
    e = Engine('AE', 32, 32, 1, 8, 16, max_batch=1)
    yield e
    e.close()


@pytest.mark.parametrize('h,w,n,iters', [(128, 128, 5, 12), (64, 96, 3, 12), (32, 32, 2, 3), (256, 256, 2, 12), (17, 23, 1, 1)])
def test_erode_cross_bit_exact(eng, h, w, n, iters):
    rng = np.random.default_rng(h + w)
    yy, xx = np.mgrid[0:h, 0:w]
    masks = np.zeros((n, h, w), bool)
    for i in range(n):
        cy, cx = rng.uniform(0.35, 0.65) * h, rng.uniform(0.35, 0.65) * w
        masks[i] = ((yy - cy) / (0.45 * h)) ** 2 + ((xx - cx) / (0.42 * w)) ** 2 <= 1
        masks[i] &= rng.random((h, w)) > 0.002                # pinholes: erosion must grow them
    got = eng.erode_cross(masks, iters).cpu().numpy()
    ref = np.stack([osc.binary_erosion_cross(m, iters) for m in masks])
    assert np.array_equal(got.astype(bool), ref)
    assert set(np.unique(got)) <= {0.0, 1.0}


def test_pipeline_matches_reference_golden(eng):
    """tests/golden/scoring_golden.npz holds the outputs of the reference's own Evaluation / Metrics functions on a seeded
    2-patient volume: eroded masks, residual maps, 5x5x5 medians, AUPRC / AUROC, the Dice sweep.  The device pipeline
    (erode -> residual -> median -> one sort) must reproduce them: masks and medians exactly (the fp32 cast is monotone,
    so the order statistic commutes with it), the scalar metrics to 1e-6 (fp64 vs fp32 residuals)."""
    x, xr, bm, lab = G['x'], G['xr'], G['bm'], G['lab']
    er = eng.erode_cross(bm, 12)
    assert np.array_equal(er.cpu().numpy().astype(np.uint8), G['eroded'])
    d, _ = eng.residual(x[..., None], xr[..., None], er[..., None], pos_only=True, prior_thresh=float(G['prior']))
    d = d[..., 0]
    np.testing.assert_allclose(d.cpu().numpy(), G['diffs'], rtol=0, atol=1e-7)
    med = torch.cat([eng.median3d(torch.from_numpy(G['diffs'][:8].astype(np.float32))),
                     eng.median3d(torch.from_numpy(G['diffs'][8:].astype(np.float32)))])
    assert np.array_equal(med.cpu().numpy(), G['med'].astype(np.float32))
    sc = eng.scores(med, lab)
    assert sc.auprc == pytest.approx(float(G['auprc']), rel=1e-6)
    assert sc.auroc == pytest.approx(float(G['auroc']), rel=1e-6)
    np.testing.assert_allclose(sc.dice_at([0.05, 0.1, 0.2]), G['dice_at'], rtol=1e-6)
    best, thr = Metrics.compute_dice_curve_recursive_device(sc, granularity=5)
    assert best == pytest.approx(float(G['best_score']), rel=1e-6) and thr == pytest.approx(float(G['best_thr']))
    sc.close()


@pytest.mark.parametrize('d,h,w', [(20, 32, 32), (7, 19, 33), (3, 8, 8), (1, 16, 16), (110, 128, 128)])
def test_median3d_bit_exact(eng, d, h, w):
    rng = np.random.default_rng(d * 1000 + h)
    vol = rng.random((d, h, w)).astype(np.float32)
    vol[rng.random(vol.shape) < 0.6] = 0.0                    # residual volumes are mostly exact zeros (many ties)
    vol[0, 0, 0] = -1.5                                       # the key transform must order negatives too
    got = eng.median3d(vol).cpu().numpy()
    if d * h * w <= 50000:
        ref = osc.median_filter_3d(vol.astype(np.float64), 5)
    else:
        import scipy.ndimage
        ref = scipy.ndimage.median_filter(vol.astype(np.float64), (5, 5, 5))     # oracle == scipy is pinned by the golden test
    assert np.array_equal(got.astype(np.float64), ref)


def test_scores_match_oracle_and_host_metrics(eng):
    rng = np.random.default_rng(11)
    n = 300000
    lab = rng.random(n) < 0.03
    pred = np.clip(rng.normal(0.15, 0.1, n) + 0.35 * lab * rng.random(n), 0, 1).astype(np.float32)
    pred[rng.random(n) < 0.5] = 0.0                           # heavy ties at 0 like a masked residual volume
    pred = np.round(pred, 3)                                  # and ties elsewhere
    sc = eng.scores(pred, lab)
    assert sc.positives == lab.sum()
    assert sc.auprc == pytest.approx(osc.average_precision(pred, lab), rel=1e-12)
    assert sc.auroc == pytest.approx(osc.auroc(pred, lab), rel=1e-12)
    ts = [0.0, 0.05, 0.1, 0.123, 0.2, 0.35, 0.5, 0.999, 1.0]
    got = sc.dice_at(ts)
    for t, g in zip(ts, got):
        ref = osc.dice((pred.astype(np.float64) > t).astype(np.int64), lab)
        assert (np.isnan(g) and np.isnan(ref)) or g == pytest.approx(ref, rel=1e-12), t
    best, thr = Metrics.compute_dice_curve_recursive_device(sc, granularity=5)
    rbest, rthr = osc.best_dice(pred.astype(np.float64), lab, granularity=5)
    assert best == pytest.approx(rbest, rel=1e-12) and thr == pytest.approx(rthr)
    hb, ht = Metrics.compute_dice_curve_recursive(pred.astype(np.float64), lab, granularity=5)
    assert best == pytest.approx(hb, rel=1e-12) and thr == pytest.approx(ht)
    sc.close()


def test_scores_edge_cases(eng):
    sc = eng.scores(np.array([0.2, 0.2, 0.2, 0.2], np.float32), np.array([1, 0, 1, 0]))
    assert sc.auprc == pytest.approx(0.5) and sc.auroc == pytest.approx(0.5)
    sc.close()
    sc = eng.scores(np.array([0.9, 0.1, 0.8, 0.3], np.float32), np.array([1, 0, 1, 0]))
    assert sc.auprc == pytest.approx(1.0) and sc.auroc == pytest.approx(1.0)
    assert sc.dice_at([0.5])[0] == pytest.approx(1.0)
    sc.close()
    with pytest.raises(ValueError):
        eng.scores(np.zeros(3, np.float32), np.zeros(4))
    with pytest.raises(ValueError):
        eng.median3d(np.zeros((4, 4), np.float32))


def _cc_cases():
    rng = np.random.default_rng(5)
    v = (rng.random((12, 20, 24)) < 0.06).astype(np.float32) * rng.uniform(0.5, 2.0, (12, 20, 24)).astype(np.float32)
    v[2:5, 3:6, 3:6] = 1.5                                  # 27-voxel blob: kept
    v[8, 10, 10:17] = 0.7                                   # 7-voxel line: removed
    v[9, 15, 2:10] = 0.9                                    # 8-voxel line: kept
    v[0, 0, 0] = v[11, 19, 23] = 1.0                        # corners
    octa = np.zeros((12, 20, 24), np.float32)               # 6 face neighbours of an empty centre + 1: area 7, encloses a 6-connected hole
    for dz, dy, dx in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1), (1, 1, 0)):
        octa[6 + dz, 10 + dy, 12 + dx] = 1.0
    clean = np.zeros((12, 20, 24), np.float32)
    clean[2:5, 3:6, 3:6] = 1.5; clean[8, 10, 10:17] = 0.7; clean[9, 15, 2:10] = 0.9; clean[0, 0, 0] = 1.0
    return [v, octa, np.zeros((3, 8, 8), np.float32), np.ones((4, 8, 8), np.float32), clean]


@pytest.mark.parametrize('case', range(5))
def test_cc_filter_bit_exact(eng, case):
    """uad_cc_filter vs the scipy labelling of the oracle / the host helper: exact (integer decisions, values passed through)."""
    from oracle import scoring as osc
    from unsupervised_anomaly_detection_brain_mri_amd.utils import Evaluation
    v = _cc_cases()[case]
    out = eng.cc_filter(v, 7).cpu().numpy()
    assert np.array_equal(out, osc.filter_3d_connected_components(v))
    assert np.array_equal(out, Evaluation.filter_3d_connected_components(v))
    if case == 4:      # 27-voxel blob and 8-voxel line kept with their values, 7-voxel line and the lone corner voxel removed
        assert out[8, 10, 10:17].sum() == 0 and out[0, 0, 0] == 0 and (out[9, 15, 2:10] == np.float32(0.9)).all() and (out[2:5, 3:6, 3:6] == 1.5).all()
    if case == 1:      # area 7 with an enclosed (6-connected) hole: filled with the full structure it is still 7 -> removed
        assert out.sum() == 0
    for mv in (0, 3, 12):
        assert np.array_equal(eng.cc_filter(v, mv).cpu().numpy(), Evaluation.filter_3d_connected_components(v, max_voxels=mv))


def test_mc_stats_matches_metrics(eng):
    """uad_mc_stats vs Metrics.combined_predictive_uncertainty(x_recs, 0) and the mean of the masked reconstructions."""
    from unsupervised_anomaly_detection_brain_mri_amd.trainers import Metrics
    rng = np.random.default_rng(2)
    recs = rng.random((7, 3, 16, 16, 1)).astype(np.float32)
    mask = (rng.random((3, 16, 16, 1)) > 0.3).astype(np.float32)
    mean, var = eng.mc_stats(recs, mask)
    p = recs.astype(np.float64) * mask
    np.testing.assert_allclose(mean.cpu().numpy(), p.mean(axis=0), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(var.cpu().numpy(), Metrics.combined_predictive_uncertainty(p, np.zeros_like(p), axis=0), rtol=1e-4, atol=1e-7)
    m2, _ = eng.mc_stats(recs)
    np.testing.assert_allclose(m2.cpu().numpy(), recs.mean(axis=0), rtol=1e-6, atol=1e-7)


def test_evaluate_with_monte_carlo_dropout(tmp_path):
    """numMonteCarloSamples > 1 (the "Bayesian" AE / VAE of the comparison): K dropout passes per batch, residual against their mean,
    epistemic variance + its histogram in the result (utils/Evaluation.py:238-266, 404-408)."""
    from unsupervised_anomaly_detection_brain_mri_amd.models import autoencoder
    from unsupervised_anomaly_detection_brain_mri_amd.trainers import AE
    from unsupervised_anomaly_detection_brain_mri_amd.utils import Evaluation
    from unsupervised_anomaly_detection_brain_mri_amd.utils.default_config_setup import get_config, get_options
    from unsupervised_anomaly_detection_brain_mri_amd.utils.synthetic import SyntheticDataset, synthetic_slices
    opt = get_options(batchsize=8, learningrate=2e-4, numEpochs=1, zDim=64, outputWidth=64, outputHeight=64, numMonteCarloSamples=4,
                      config={'CHECKPOINTDIR': str(tmp_path / 'ck'), 'SAMPLEDIR': str(tmp_path / 'smp')})
    ds = SyntheticDataset(16, 8, 64, 64, seed=0)
    model = AE(None, get_config(AE, opt, 'ADAM', [8, 8], 0.2, ds), network=autoencoder)
    model.train(ds)
    x, lab, msk = synthetic_slices(10, 64, 64, seed=33, lesions=True)
    ev = Evaluation.evaluate_arrays([x[..., 0].astype(np.float64)], [lab], [msk], model, opt)
    assert ev['epistemic_variance'].shape == (10, 64, 64) and (ev['epistemic_variance'] >= -1e-6).all() and ev['epistemic_variance'].max() > 0
    assert len(ev['uncertaintyHistogram']) == 50 and 0.0 <= ev['diff_AUC'] <= 1.0
    opt1 = dict(opt, numMonteCarloSamples=0)
    ev1 = Evaluation.evaluate_arrays([x[..., 0].astype(np.float64)], [lab], [msk], model, opt1)
    assert 'epistemic_variance' not in ev1
    model.engine.close()
