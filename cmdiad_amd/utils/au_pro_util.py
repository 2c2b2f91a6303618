"""AU-PRO (area under the per-region-overlap vs false-positive-rate curve), host side.

Counterpart of the reference's ``utils/au_pro_util.py`` (``calculate_au_pro(gts, predictions,
integration_limit=0.3, num_thresholds=100)``, called at feature_extractors/features.py:323-324), written
from the metric's definition (MVTec 3D-AD evaluation protocol): every ground-truth connected component
(8-connectivity) contributes its covered fraction; PRO(t) is the mean over all components of all
images, FPR(t) is computed over the defect-free pixels of all images; the curve is integrated up to
``integration_limit`` FPR (linear interpolation at the limit) and normalised by the limit.
Evaluation-only code: it stays on the host.
"""
import numpy as np
from scipy.ndimage import label


def _pro_curve(gts, predictions):
    structure = np.ones((3, 3), dtype=int)
    ok_scores, region_scores, region_weights = [], [], []
    n_regions = 0
    for gt, pred in zip(gts, predictions):
        gt = np.asarray(gt)
        pred = np.asarray(pred, dtype=np.float64)
        labeled, n = label(gt > 0, structure)
        n_regions += n
        ok_scores.append(pred[labeled == 0])
        for c in range(1, n + 1):
            sc = pred[labeled == c]
            region_scores.append(sc)
            region_weights.append(np.full(sc.shape, 1.0 / sc.size))
    ok = np.concatenate(ok_scores) if ok_scores else np.zeros(0)
    if n_regions == 0:
        return np.array([0.0, 1.0]), np.array([0.0, 0.0])
    reg = np.concatenate(region_scores)
    w = np.concatenate(region_weights) / n_regions
    scores = np.concatenate([ok, reg])
    fp_inc = np.concatenate([np.full(ok.shape, 1.0 / max(ok.size, 1)), np.zeros(reg.shape)])
    pro_inc = np.concatenate([np.zeros(ok.shape), w])
    order = np.argsort(-scores, kind="stable")
    s_sorted = scores[order]
    fpr = np.cumsum(fp_inc[order])
    pro = np.cumsum(pro_inc[order])
    # keep one point per distinct threshold (the last pixel of every run of equal scores)
    keep = np.append(s_sorted[1:] != s_sorted[:-1], True)
    fpr, pro = fpr[keep], pro[keep]
    fpr = np.concatenate([[0.0], np.clip(fpr, 0.0, 1.0)])
    pro = np.concatenate([[0.0], np.clip(pro, 0.0, 1.0)])
    return fpr, pro


def trapezoid(x, y, x_max=None):
    x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
    if x_max is not None and x_max < x[-1]:
        i = int(np.searchsorted(x, x_max, side="right"))
        y_at = y[i - 1] + (y[i] - y[i - 1]) * (x_max - x[i - 1]) / (x[i] - x[i - 1]) if i < len(x) and x[i] > x[i - 1] else y[i - 1]
        x = np.append(x[:i], x_max)
        y = np.append(y[:i], y_at)
    return float(np.sum(0.5 * (y[1:] + y[:-1]) * (x[1:] - x[:-1])))


def _pro_curve_sampled(gts, predictions, num_thresholds):
    """The reference's sampling of the curve (utils/au_pro_util.py:156-201): ``num_thresholds`` thresholds taken at
    equidistant RANKS of the sorted defect-free scores; at threshold s a pixel is flagged when its score is > s, so
    FPR = 1 - (rank + 1) / n_ok and a component's overlap = the fraction of its scores > s; the point (1, 1) closes the
    curve.  Vectorised with searchsorted instead of the reference's per-component cursor."""
    structure = np.ones((3, 3), dtype=int)
    ok, comps = [], []
    for gt, pred in zip(gts, predictions):
        gt = np.asarray(gt)
        pred = np.asarray(pred, dtype=np.float64)
        labeled, n = label(gt, structure)
        ok.append(pred[labeled == 0])
        comps.extend(np.sort(pred[labeled == c]) for c in range(1, n + 1))
    ok = np.sort(np.concatenate(ok))
    pos = np.linspace(0, len(ok) - 1, num=num_thresholds, dtype=int)
    thr = ok[pos]
    fpr = 1.0 - (pos + 1) / len(ok)
    pro = np.zeros(len(thr))
    for sc in comps:
        pro += 1.0 - np.searchsorted(sc, thr, side="right") / len(sc)
    pro /= max(len(comps), 1)
    return np.concatenate([fpr[::-1], [1.0]]), np.concatenate([pro[::-1], [1.0]])


def calculate_au_pro(gts, predictions, integration_limit=0.3, num_thresholds=100):
    """-> (au_pro normalised to [0,1], (fpr, pro) curve).  With ``num_thresholds`` (default 100, as the reference calls it,
    features.py:323-324) the curve is sampled exactly as the reference samples it, so AU-PRO / AU-PRO@1% are directly
    comparable with reference numbers; ``num_thresholds=None`` integrates the exact curve (one point per distinct score)."""
    if num_thresholds is None:
        fpr, pro = _pro_curve(gts, predictions)
    else:
        fpr, pro = _pro_curve_sampled(gts, predictions, int(num_thresholds))
    au = trapezoid(fpr, pro, x_max=integration_limit) / integration_limit
    return au, (fpr, pro)
