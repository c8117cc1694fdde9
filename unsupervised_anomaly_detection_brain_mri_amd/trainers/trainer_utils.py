"""trainers/trainer_utils.py:6-18 -- which fetches of a `sess.run` count as scalars, and the image strip TensorBoard gets: per sample the
min-max normalised input next to the visualisation keys' maps (default 'reconstruction', 'L1'), scaled to [0, 255].  `normalize` is
utils/utils.py:74-75 (cv2.normalize NORM_MINMAX to [0, 1]) in numpy; a constant image maps to 0 as cv2 does."""
import numpy as np


def normalize(x):
    x = np.asarray(x, np.float32)
    if x.ndim == 3 and x.shape[-1] == 1:
        x = x[..., 0]
    lo, hi = float(x.min()), float(x.max())
    out = np.zeros_like(x) if hi <= lo else (x - lo) / (hi - lo)
    return out[..., None]


def get_summary_dict(batch, run, visualization_keys=None, *others):
    if visualization_keys is None:
        visualization_keys = ['reconstruction', 'L1']
    run = {k: v for k, v in run.items() if v is not None}
    visuals = np.asarray([255 * np.hstack([normalize(batch[i]), *[normalize(run[key][i]) for key in visualization_keys],
                                           *[normalize(element[i]) for element in others]]) for i in range(len(batch))])
    scalars = {k: v for k, v in run.items() if not (type(v) == float and v != v) and np.ndim(v) == 0}
    return scalars, visuals
