"""Times the device scoring kernels (row a14) at the reference's evaluation sizes: one patient = [110,128,128], Brainweb TEST = 12 volumes
= 21.6 M voxels; host columns = the scipy / numpy calls the reference makes (bounded samples)."""
import sys, time
import numpy as np, torch, scipy.ndimage
sys.path.insert(0, '.')
from unsupervised_anomaly_detection_brain_mri_amd.engine import Engine
from unsupervised_anomaly_detection_brain_mri_amd.trainers import Metrics

eng = Engine('AE', 32, 32, 1, 8, 16, max_batch=1)
rng = np.random.default_rng(0)
D, H, W = 110, 128, 128


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


yy, xx = np.mgrid[0:H, 0:W]
masks = np.broadcast_to((((yy - 64) / 52) ** 2 + ((xx - 64) / 46) ** 2 <= 1), (D, H, W)).astype(np.float32).copy()
vol = (rng.random((D, H, W)) * (rng.random((D, H, W)) < 0.4)).astype(np.float32)
dm, dv = torch.from_numpy(masks).cuda(), torch.from_numpy(vol).cuda()
res = {}
res['erode_ms'] = timed(lambda: eng.erode_cross(dm, 12))
res['median_ms'] = timed(lambda: eng.median3d(dv))
# thresholded 12-patient stack [1320,128,128] with ~1.5 % foreground (blobs + speckle), like `diffs > bestThreshold`
thr_mask = (scipy.ndimage.uniform_filter(rng.random((12 * D, H, W)).astype(np.float32), 3) > 0.62).astype(np.float32)
dt_ = torch.from_numpy(thr_mask).cuda()
res['cc_filter_ms'] = timed(lambda: eng.cc_filter(dt_, 7))
res['cc_foreground'] = float(thr_mask.mean())
n = 12 * D * H * W
lab = rng.random(n) < 0.02
pred = (rng.random(n) * (rng.random(n) < 0.4) + 0.3 * lab * rng.random(n)).astype(np.float32)
dp, dl = torch.from_numpy(pred).cuda(), torch.from_numpy(lab.astype(np.float32)).cuda()
t0 = time.perf_counter(); sc = eng.scores(dp, dl); res['scores_create_ms'] = (time.perf_counter() - t0) * 1e3
t0 = time.perf_counter(); best = Metrics.compute_dice_curve_recursive_device(sc, granularity=10); res['dice_sweep_ms'] = (time.perf_counter() - t0) * 1e3
res['auprc'], res['auroc'], res['best_dice'] = sc.auprc, sc.auroc, best[0]
# host references (one volume / the same 21.6 M voxels)
t0 = time.perf_counter(); strel = scipy.ndimage.generate_binary_structure(2, 1)
for s in range(D): scipy.ndimage.binary_erosion(masks[s], structure=strel, iterations=12)
res['host_erode_ms'] = (time.perf_counter() - t0) * 1e3
t0 = time.perf_counter(); scipy.ndimage.median_filter(vol.astype(np.float64), (5, 5, 5)); res['host_median_ms'] = (time.perf_counter() - t0) * 1e3
t0 = time.perf_counter(); hp = Metrics.compute_prc(pred.astype(np.float64), lab)[0]; hr = Metrics.compute_roc(pred.astype(np.float64), lab)[0]
res['host_sorted_metrics_ms'] = (time.perf_counter() - t0) * 1e3
from unsupervised_anomaly_detection_brain_mri_amd.utils import Evaluation
t0 = time.perf_counter(); href = Evaluation.filter_3d_connected_components(thr_mask); res['host_cc_filter_ms'] = (time.perf_counter() - t0) * 1e3
res['cc_equal'] = bool(np.array_equal(href, eng.cc_filter(dt_, 7).cpu().numpy()))
res['host_auprc'], res['host_auroc'] = hp, hr
print({k: (round(v, 6) if isinstance(v, float) else v) for k, v in res.items()})
