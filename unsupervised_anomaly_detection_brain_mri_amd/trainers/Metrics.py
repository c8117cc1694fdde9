"""trainers/Metrics.py — scoring used by the evaluation path (same function names / return conventions; plotting and
CSV export dropped).  AUPRC / AUROC are computed from one sort of the predictions (what sklearn does underneath:
average_precision_score / roc_curve+auc, Metrics.py:17-19,45-47); Dice and the greedy recursive threshold sweep follow
Metrics.py:67-72,138-162 but evaluate every threshold on the sorted array in O(log n) instead of a full pass."""
import numpy as np


def _sorted_state(predictions, labels):
    p = np.asarray(predictions, np.float64).reshape(-1)
    y = np.asarray(labels).reshape(-1).astype(np.float64)
    order = np.argsort(-p, kind='mergesort')
    p, y = p[order], y[order]
    return p, np.cumsum(y), y.sum()


def compute_prc(predictions, labels, filename=None, plottitle=None):
    p, ctp, npos = _sorted_state(predictions, labels.astype(int))
    distinct = np.r_[np.nonzero(np.diff(p))[0], p.size - 1]
    tps = ctp[distinct]
    fps = 1 + distinct - tps
    precisions = tps / (tps + fps)
    recalls = tps / npos
    auprc = float(np.sum(np.diff(np.r_[0.0, recalls]) * precisions))
    # sklearn's precision_recall_curve ordering: increasing threshold, with the (1, 0) end point appended
    return auprc, np.r_[precisions[::-1], 1.0], np.r_[recalls[::-1], 0.0], p[distinct][::-1]


def compute_roc(predictions, labels, filename=None, plottitle=None):
    p, ctp, npos = _sorted_state(predictions, labels.astype(int))
    distinct = np.r_[np.nonzero(np.diff(p))[0], p.size - 1]
    tps = ctp[distinct]
    fps = 1 + distinct - tps
    tpr = np.r_[0.0, tps / npos]
    fpr = np.r_[0.0, fps / (p.size - npos)]
    return float(np.trapezoid(tpr, fpr)), fpr, tpr, np.r_[np.inf, p[distinct]]


def dice(P, G):
    psum = np.sum(P.flatten())
    gsum = np.sum(G.flatten())
    pgsum = np.sum(np.multiply(P.flatten(), G.flatten()))
    return (2 * pgsum) / (psum + gsum)


def xfrange(start, stop, step):
    i = 0
    while start + i * step < stop:
        yield start + i * step
        i += 1


def _dice_sweep(dice_batch, granularity):
    """The greedy recursive threshold sweep of Metrics.py:138-162 (identical control flow) over a provider
    dice_batch(list of thresholds) -> dice(pred > t, labels) for each: one call per recursion level."""
    def inner(start, stop, decimal):
        _threshs, _scores = [], []
        had_recursion = False
        if decimal == granularity:
            return _threshs, _scores
        ts = list(xfrange(start, stop, (1.0 / (10.0 ** decimal))))
        vals = dice_batch(ts) if ts else []
        for i, t in enumerate(ts):
            score = vals[i]
            if i >= 2 and score <= _scores[i - 1] and not had_recursion:
                st, ss = inner(_threshs[i - 2], t, decimal + 1)
                _threshs.extend(st)
                _scores.extend(ss)
                had_recursion = True
            _scores.append(score)
            _threshs.append(t)
        return _threshs, _scores

    threshs, scores = inner(0, 1.0, 1)
    threshs, scores = list(zip(*sorted(zip(threshs, scores))))
    return scores, threshs


def compute_dice_score(predictions, labels, granularity):
    """Metrics.py:138-162; dice(pred > t, labels) is read off the sorted cumulative sums."""
    p, ctp, gsum = _sorted_state(predictions, labels)
    asc = p[::-1]

    def dice_at(t):
        k = p.size - np.searchsorted(asc, t, side='right')      # number of predictions > t
        tp = ctp[k - 1] if k > 0 else 0.0
        return (2 * tp) / (k + gsum)

    return _dice_sweep(lambda ts: [dice_at(t) for t in ts], granularity)


def compute_dice_curve_recursive_device(scores, granularity=5):
    """compute_dice_curve_recursive on a device-side engine.Scores object (one sort for every threshold of the sweep)."""
    sc, th = _dice_sweep(lambda ts: list(scores.dice_at(ts)), granularity)
    best = int(np.argmax(sc))
    return sc[best], th[best]


def compute_dice_curve_recursive(predictions, labels, filename=None, plottitle=None, granularity=5):
    scores, threshs = compute_dice_score(predictions, labels, granularity)
    i = int(np.argmax(scores))
    return scores[i], threshs[i]


def confusion_matrix(P, G):
    P, G = P.flatten().astype(bool), G.flatten().astype(bool)
    return np.sum(P & G), np.sum(P & ~G), np.sum(~P & ~G), np.sum(~P & G)


def precision(P, G):
    tp, fp, _, _ = confusion_matrix(P, G)
    return tp / (tp + fp)


def recall(P, G):
    tp, _, _, fn = confusion_matrix(P, G)
    return tp / (tp + fn)


def combined_predictive_uncertainty(p, sigmas, axis=-1, log_var=False):     # Metrics.py:170-173
    if log_var:
        sigmas = np.exp(sigmas)
    return np.mean(np.square(p), axis=axis) - np.square(np.mean(p, axis=axis)) + np.mean(sigmas, axis=axis)
