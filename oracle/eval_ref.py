"""CPU restatement of the verification metrics (TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows /root/reference/utils/eval.py: pair_score (:68-99: score = 1 - |a-b|^2/4 accumulated in float64 from float32
differences, hist_idx = int(99999 * score)), performance_roc (:7-51: thresholds 100000 -> 1, FAR / FRR from the
cumulative histograms, first minimum of |FAR - FRR| = EER threshold, best FRR with FAR <= 1e-level per security
level), performance_acc (:54-66), cross_score (:102-137: all pairs j < i, l = i(i-1)/2 + j).  Plain numpy, vectorised where the reference loops.
"""
import numpy as np


def pair_scores(e1, e2):
    d = (e1.astype(np.float32) - e2.astype(np.float32)).astype(np.float64)
    return 1.0 - np.sum(d * d, axis=1) / 4.0


def histograms(scores, labels):
    idx = ((1e5 - 1.0) * scores).astype(np.int64)          # int() truncates toward zero, scores are >= 0 here
    hg = np.bincount(idx[labels.astype(bool)], minlength=100001).astype(np.float64)
    hi = np.bincount(idx[~labels.astype(bool)], minlength=100001).astype(np.float64)
    return idx, hg, hi


def roc(hist_genuine, hist_imposter, min_level=3, max_level=9):
    """-> (eer_threshold, eer, [(frr, threshold) per level])"""
    th = np.arange(int(1e5), 0, -1)
    tg, ti = int(hist_genuine.sum()), int(hist_imposter.sum())
    cg = np.concatenate([[0.0], np.cumsum(hist_genuine[th])[:-1]])      # genuine counted ABOVE the threshold so far
    ci = np.concatenate([[0.0], np.cumsum(hist_imposter[th])[:-1]])
    far = (ci + hist_imposter[th]) / ti
    frr = (tg - cg) / tg
    diff = np.abs(far - frr)
    best, eer_i = 1.0, None
    # strict '<' against a running minimum that starts at 1: first index of the global minimum, if it is below 1
    m = diff.min()
    if m < best:
        eer_i = int(np.argmax(diff == m))
    eer_th = int(th[eer_i]) if eer_i is not None else int(1e5)
    eer = float((far[eer_i] + frr[eer_i]) / 2) if eer_i is not None else None
    levels = []
    for lv in range(min_level, max_level + 1):
        ok = far <= float("1e-%d" % lv)
        if ok.any():
            f = np.where(ok, frr, np.inf)
            j = int(np.argmin(f))               # strict '<' while scanning: first occurrence of the minimum
            levels.append((float(frr[j]), int(th[j])))
        else:
            levels.append((None, None))
    return eer_th, eer, levels


def accuracy(scores, labels, th):
    fr = np.sum((scores <= th / 1e5) & (labels == 1))
    fa = np.sum((scores > th / 1e5) & (labels == 0))
    return (1 - (fa + fr) / len(scores)) * 100


def cross_scores(emb, labels):
    """-> (scores [P], pair labels [P]) over all pairs j < i in the reference's order l = i (i - 1) / 2 + j"""
    emb = emb.astype(np.float32)
    n = emb.shape[0]
    ii, jj = np.tril_indices(n, -1)                         # row-major over (i, j < i): exactly l = i(i-1)/2 + j
    d = (emb[jj] - emb[ii]).astype(np.float64)
    labels = np.asarray(labels).reshape(-1)
    return 1.0 - np.sum(d * d, axis=1) / 4.0, (labels[jj] == labels[ii]).astype(np.float64)
