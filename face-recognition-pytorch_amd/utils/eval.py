"""Drop-in for the reference verification metrics `utils/eval.py`.

Same functions / return shapes as /root/reference/utils/eval.py: `pair_score(embedding_1, embedding_2, labels)` ->
(hist_genuine[100001], hist_imposter[100001], score_list), `performance_roc(hist_genuine, hist_imposter, min_level,
max_level)` -> (roc report string, eer_threshold), `performance_acc(score_list, label_list, th)` -> accuracy in %,
`cross_score(embeddings, labels)` -> (hist_genuine, hist_imposter, score_list, label_list) over all pairs j < i (:102-137).
pair_score runs on the MI355X (frhip_pair_score: float64 accumulation of float32 differences in the reference's
order, so `int(99999*score)` is bit-exact); the ROC scan and accuracy are host logic on 100 001-bin histograms
(the reference runs them on the host too) restated with numpy cumulative sums instead of Python loops.
"""
import numpy as np
import torch


def pair_score(embedding_1, embedding_2, labels, metric="euclidean", min_level=3, max_level=9):
    assert metric in ["euclidean", "cosine"], "Invalid metric !!!"
    from frhip import ops
    if not torch.cuda.is_available():
        raise RuntimeError("utils.eval.pair_score (frhip) needs the MI355X; there is no CPU path")
    e1 = torch.as_tensor(np.asarray(embedding_1), dtype=torch.float32).cuda().contiguous() if not torch.is_tensor(embedding_1) else embedding_1.float().cuda().contiguous()
    e2 = torch.as_tensor(np.asarray(embedding_2), dtype=torch.float32).cuda().contiguous() if not torch.is_tensor(embedding_2) else embedding_2.float().cuda().contiguous()
    lab = torch.as_tensor(np.asarray(labels)).long().cuda().contiguous() if not torch.is_tensor(labels) else labels.long().cuda().contiguous()
    scores, _, hg, hi = ops.pair_score(e1, e2, lab)
    return hg.cpu().numpy().astype(np.float64), hi.cpu().numpy().astype(np.float64), scores.cpu().numpy()


def _dev(x, dtype):
    t = x if torch.is_tensor(x) else torch.as_tensor(np.asarray(x))
    return t.to(dtype).cuda().contiguous()


def cross_score(embeddings, labels, metric="euclidean"):
    """Cross-matching scores of ONE embedding set (reference utils/eval.py:102-137): every pair j < i in the reference's
    order l = i(i-1)/2 + j -> (hist_genuine[100001], hist_imposter[100001], score_list[P], label_list[P]), label 1 where the
    two identities agree.  frhip_cross_score: the reference's float64-of-float32-differences arithmetic, so the histogram
    bins are bit-exact."""
    assert metric in ["euclidean", "cosine"], "Invalid metric !!!"
    from frhip import ops
    if not torch.cuda.is_available():
        raise RuntimeError("utils.eval.cross_score (frhip) needs the MI355X; there is no CPU path")
    scores, plab, _, hg, hi = ops.cross_score(_dev(embeddings, torch.float32), _dev(labels, torch.int64).view(-1))
    return hg.cpu().numpy().astype(np.float64), hi.cpu().numpy().astype(np.float64), scores.cpu().numpy(), plab.cpu().numpy()


def performance_roc(hist_genuine, hist_imposter, min_level=3, max_level=9):
    th = np.arange(int(1e5), 0, -1)
    total_genuine, total_imposter = int(sum(hist_genuine)), int(sum(hist_imposter))
    hg, hi = np.asarray(hist_genuine, dtype=np.float64)[th], np.asarray(hist_imposter, dtype=np.float64)[th]
    cum_g = np.concatenate([[0.0], np.cumsum(hg)[:-1]])
    cum_i = np.concatenate([[0.0], np.cumsum(hi)[:-1]])
    far = (cum_i + hi) / total_imposter
    frr = (total_genuine - cum_g) / total_genuine
    diff = np.abs(far - frr)
    eer, eer_threshold = None, 1e5
    if diff.min() < 1:
        j = int(np.argmax(diff == diff.min()))
        eer, eer_threshold = (far[j] + frr[j]) / 2, int(th[j])
    roc_result = "\n"
    for level in range(min_level, max_level + 1):
        ok = far <= float(f"1e-{level}")
        j = int(np.argmin(np.where(ok, frr, np.inf)))
        roc_result += f"- FRR @ FAR{level} {100 * frr[j]:6.3f}%, (Threshold = {th[j] / 1e5:.5f})  \n"
    roc_result += "- EER {0:6.3f}%, (Threshold = {1:.5f})\n".format(100 * eer, eer_threshold / 1e5)
    roc_result += "- Total count = {:,}\n".format(total_genuine + total_imposter)
    roc_result += "- Total genuine count = {:,}\n".format(total_genuine)
    roc_result += "- Total imposter count = {:,}\n".format(total_imposter)
    return roc_result, eer_threshold


def performance_acc(score_list, label_list, th):
    score_list, label_list = np.asarray(score_list), np.asarray(label_list)
    fr = int(np.sum((score_list <= th / 1e5) & (label_list == 1)))
    fa = int(np.sum((score_list > th / 1e5) & (label_list == 0)))
    return (1 - (fa + fr) / (len(score_list))) * 100
