"""CPU: the scoring oracle (independent numpy restatement) vs golden vectors produced by the REFERENCE's own
trainers/Metrics.py + utils/Evaluation.py helpers (tests/golden/make_scoring_golden.py).  This pins the oracle for
SURVEY.md §8 row a14.  Dice / AUPRC bar: 1e-3 (north_star); we hold 1e-9."""
import os

import numpy as np
import pytest

from oracle import scoring as sc

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'scoring_golden.npz'))


def test_erosion_matches_reference_apply_brainmask():
    for s in range(G['bm'].shape[0]):
        got = sc.apply_brainmask(np.ones(G['bm'][s].shape), G['bm'][s], erode=True)
        assert np.array_equal(got.astype(np.uint8), G['eroded'][s])


def test_residual_map_matches_reference_lines():
    x, xr = G['x'].astype(np.float64), G['xr'].astype(np.float64)
    for s in range(x.shape[0]):
        d = sc.residual_map(x[s], xr[s], G['bm'][s], True, True, float(G['prior']))
        np.testing.assert_allclose(d, G['diffs'][s], rtol=0, atol=1e-15)


def test_median_filter_matches_scipy_reflect():
    med = np.concatenate([sc.median_filter_3d(G['diffs'][:8]), sc.median_filter_3d(G['diffs'][8:])])
    np.testing.assert_allclose(med, G['med'], rtol=0, atol=1e-15)


def test_auprc_auroc_match_sklearn_via_reference():
    pred, gt = G['med'].flatten(), G['lab'].astype(bool).flatten()
    assert abs(sc.average_precision(pred, gt) - float(G['auprc'])) < 1e-9
    assert abs(sc.auroc(pred, gt) - float(G['auroc'])) < 1e-9


def test_dice_and_threshold_sweep_match_reference():
    pred, lab = G['med'].flatten(), G['lab'].astype(np.int64).flatten()
    for t, ref in zip((0.05, 0.1, 0.2), G['dice_at']):
        d = sc.dice(np.where(pred > t, 1, 0), lab)
        assert (np.isnan(d) and np.isnan(ref)) or abs(d - ref) < 1e-12
    scores, threshs = sc.compute_dice_score(pred, lab, 5)
    np.testing.assert_allclose(threshs, G['dice_threshs'], rtol=0, atol=1e-15)
    np.testing.assert_allclose(scores, G['dice_scores'], rtol=0, atol=1e-12)
    bs, bt = sc.best_dice(pred, lab, 5)
    assert abs(bs - float(G['best_score'])) < 1e-12 and abs(bt - float(G['best_thr'])) < 1e-15


def test_cc_filter_removes_small_components_only():
    vol = np.zeros((6, 10, 10), np.int64)
    vol[1, 1:3, 1:3] = 1            # 4 voxels -> removed
    vol[2:5, 5:8, 5:8] = 1          # 27 voxels -> kept
    out = sc.filter_3d_connected_components(vol)
    assert out[1, 1:3, 1:3].sum() == 0 and out[2:5, 5:8, 5:8].sum() == 27
