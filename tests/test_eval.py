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


def _cross_inputs(g):
    ids, per, d = int(g["ids"]), int(g["per"]), int(g["d"])
    centres = recipe.normal(8201, (ids, d))
    emb = torch.nn.functional.normalize(centres.repeat_interleave(per, 0) * 0.35 + recipe.normal(8202, (ids * per, d)))
    perm = recipe.rng(8203).permutation(ids * per)
    return emb[torch.from_numpy(perm)].numpy(), g["labels"]


def test_oracle_cross_score_matches_reference(golden):
    g = golden("cross_eval")
    emb, labels = _cross_inputs(g)
    scores, plab = eval_ref.cross_scores(emb, labels)
    np.testing.assert_allclose(scores, g["scores"], rtol=0, atol=1e-12)
    assert np.array_equal(plab, g["pair_labels"])
    idx, hg, hi = eval_ref.histograms(scores, plab)
    assert np.array_equal(idx, g["hist_idx"])
    assert np.array_equal(hg, g["hist_genuine"]) and np.array_equal(hi, g["hist_imposter"])
    eer_th, _, _ = eval_ref.roc(hg, hi, 1, 3)
    assert eer_th == int(g["eer_th"])
    np.testing.assert_allclose(eval_ref.accuracy(scores, plab, eer_th), g["acc"], rtol=1e-12)


@pytest.mark.gpu
def test_cross_score_kernel_bit_exact_indices(golden):
    from utils import eval as ev
    g = golden("cross_eval")
    emb, labels = _cross_inputs(g)
    hg, hi, scores, plab = ev.cross_score(emb, labels)
    assert np.array_equal(((1e5 - 1.0) * scores).astype(np.int64), g["hist_idx"])       # bit-exact
    assert np.array_equal(plab, g["pair_labels"])
    assert np.array_equal(hg, g["hist_genuine"]) and np.array_equal(hi, g["hist_imposter"])
    np.testing.assert_allclose(scores, g["scores"], rtol=0, atol=1e-12)
    roc, eer_th = ev.performance_roc(hg, hi, min_level=1, max_level=3)
    assert eer_th == int(g["eer_th"]) and roc == str(g["roc"])
    np.testing.assert_allclose(ev.performance_acc(scores, plab, eer_th), g["acc"], rtol=1e-12)


@pytest.mark.gpu
def test_model_cross_test_steps():
    """Model.cross_test_step / cross_test_epoch_end (reference model/FR_PartialFC.py:379-427) end to end on the HIP encoder:
    the epoch-end metrics equal the oracle's on the embeddings the steps returned."""
    import types
    import torch.distributed as dist
    import tempfile, os
    from model.FR_PartialFC import Model
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method="file://" + os.path.join(tempfile.mkdtemp(), "pg"), rank=0, world_size=1)
    conf = types.SimpleNamespace(network="ResNet18", emd_size=512, img_size=112, local_rank=0, world_size=1, sample_rate=1.0,
                                 mixed_precision=True, loss_s=30.0, loss_m=0.35, n_classes=16, optimizer="SGD", lr=0.1, wd=5e-4,
                                 mom=0.9, lr_scheduler=None, frhip_dtype="bf16", ckpt_path=None, cross_test_dataset=["synt"],
                                 min_level=1, max_level=3)
    torch.manual_seed(3)
    model = Model(conf, None, "test")
    gen = torch.Generator().manual_seed(5)
    outs = []
    for k in range(3):
        img = torch.randn((4, 3, 112, 112), generator=gen).clamp_(-1, 1)
        outs.append(model.cross_test_step((img, torch.tensor([0, 1, 0, 2]) + k), 0))
    assert outs[0]["synt_embedding"].shape == (4, 512) and outs[0]["dataset_name"] == "synt"
    res = model.cross_test_epoch_end(outs)
    emb = np.concatenate([o["synt_embedding"].numpy() for o in outs])
    lab = np.concatenate([o["synt_label_list"].numpy() for o in outs])
    scores, plab = eval_ref.cross_scores(emb, lab)
    idx, hg, hi = eval_ref.histograms(scores, plab)
    eer_th, _, _ = eval_ref.roc(hg, hi, 1, 3)
    assert res["eer_th"] == eer_th
    np.testing.assert_allclose(res["acc"], eval_ref.accuracy(scores, plab, eer_th), rtol=1e-12)
