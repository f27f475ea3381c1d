"""Verification metrics: oracle and drop-in utils.eval against reference-generated vectors."""
import numpy as np
import pytest
import torch

from oracle import eval_ref, recipe


def _pairs(g):
    n, d = int(g["n"]), int(g["d"])
    base, other = recipe.normal(8101, (n, d)), recipe.normal(8102, (n, d))
    labels = g["labels"]
    e1 = torch.nn.functional.normalize(base)
    mix = torch.from_numpy(np.where(labels[:, None] == 1, 0.12, 0.0).astype(np.float32))
    e2 = torch.nn.functional.normalize(mix * base + (1 - mix) * other + 0.0 * recipe.normal(8104, (n, d)))
    return e1.numpy(), e2.numpy(), labels


def test_oracle_eval_matches_reference(golden):
    g = golden("eval_pairs")
    e1, e2, labels = _pairs(g)
    scores = eval_ref.pair_scores(e1, e2)
    idx, hg, hi = eval_ref.histograms(scores, labels)
    np.testing.assert_allclose(scores, g["scores"], rtol=0, atol=1e-12)
    assert np.array_equal(idx, g["hist_idx"])                               # bit-exact integer artefact
    assert np.array_equal(hg, g["hist_genuine"]) and np.array_equal(hi, g["hist_imposter"])
    eer_th, eer, _ = eval_ref.roc(hg, hi, 1, 3)
    assert eer_th == int(g["eer_th"])
    np.testing.assert_allclose(eval_ref.accuracy(scores, labels, eer_th), g["acc"], rtol=1e-12)


def test_host_roc_and_accuracy_match_reference(golden):
    """performance_roc / performance_acc of the drop-in are host logic: checked without a GPU"""
    from utils import eval as ev
    g = golden("eval_pairs")
    roc, eer_th = ev.performance_roc(g["hist_genuine"], g["hist_imposter"], min_level=1, max_level=3)
    assert eer_th == int(g["eer_th"]) and roc == str(g["roc"])
    np.testing.assert_allclose(ev.performance_acc(g["scores"], g["labels"], eer_th), g["acc"], rtol=1e-12)


@pytest.mark.gpu
def test_pair_score_kernel_bit_exact_indices(golden):
    from utils import eval as ev
    g = golden("eval_pairs")
    e1, e2, labels = _pairs(g)
    hg, hi, scores = ev.pair_score(e1, e2, labels)
    idx = ((1e5 - 1.0) * scores).astype(np.int64)
    assert np.array_equal(idx, g["hist_idx"])                               # bit-exact
    assert np.array_equal(hg, g["hist_genuine"]) and np.array_equal(hi, g["hist_imposter"])
    np.testing.assert_allclose(scores, g["scores"], rtol=0, atol=1e-12)
    roc, eer_th = ev.performance_roc(hg, hi, min_level=1, max_level=3)
    assert eer_th == int(g["eer_th"])
    np.testing.assert_allclose(ev.performance_acc(scores, labels, eer_th), g["acc"], rtol=1e-12)
