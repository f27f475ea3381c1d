"""Pin oracle/head_ref.py to vectors produced by the real reference (tools/make_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import head_ref, recipe

T = torch.from_numpy


@pytest.mark.parametrize("tag", ["s30_m035", "s64_m05"])
def test_arcface_edge(golden, tag):
    g = golden("arcface_edge_" + tag)
    cos = T(g["logits_in"])
    lab = T(g["labels"]).flatten()
    z, _ = head_ref.arcface_logits(cos, lab, float(g["s"]), float(g["m"]))
    np.testing.assert_allclose(z.numpy(), g["logits_out"], rtol=1e-6, atol=1e-6)


def test_distce(golden):
    g = golden("distce_ws1")
    loss, grads = head_ref.dist_cross_entropy([T(g["z"])], [T(g["labels"]).flatten()], float(g["upstream"]))
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-6)
    np.testing.assert_allclose(grads[0].numpy(), g["grad"], rtol=1e-5, atol=1e-8)


def _head_inputs(g):
    ws, C, B, D = int(g["ws"]), int(g["C"]), int(g["B"]), int(g["D"])
    embs, labs, ws_, us = [], [], [], []
    for r in range(ws):
        embs.append(recipe.normal(100 + r, (B, D)))
        lab = recipe.labels(200 + r, B, C)
        if int(g["dup"]):
            lab[0] = 3
            lab[1] = 3
        labs.append(lab)
        c0, nloc = head_ref.shard_range(C, ws, r)
        assert c0 == int(g["r%d_class_start" % r]) and nloc == int(g["r%d_num_local" % r])
        assert head_ref.num_sample(float(g["rate"]), nloc) == int(g["r%d_num_sample" % r])
        ws_.append(recipe.normal(500 + r, (nloc, D), 0.05))
        us.append(T(g["r%d_u" % r]))
    return embs, labs, ws_, us


@pytest.mark.parametrize("name", ["head_ws1_rate10", "head_ws1_rate03", "head_ws2_rate10",
                                  "head_ws2_rate03", "head_ws8_rate01", "head_ws4_rate01"])
def test_head_matches_reference(golden, name):
    g = golden(name)
    embs, labs, weights, us = _head_inputs(g)
    out = head_ref.head_all_shards(embs, labs, weights, int(g["C"]), float(g["s"]), float(g["m"]),
                                   sample_rate=float(g["rate"]), uniforms=us)
    for r in range(int(g["ws"])):
        np.testing.assert_allclose(out["loss"].item(), g["r%d_loss" % r], rtol=2e-6)
        assert np.array_equal(out["index"][r].numpy(), g["r%d_index" % r])       # bit-exact
        np.testing.assert_allclose(out["d_emb"][r].numpy(), g["r%d_d_emb" % r], rtol=2e-4, atol=2e-7)
        np.testing.assert_allclose(out["d_w_act"][r].numpy(), g["r%d_d_w_act" % r], rtol=2e-4, atol=2e-7)


def test_sampling_same_rng_stream(golden):
    """The reference draws u with torch.rand on the CPU generator (nets/PartialFC.py:110);
    same seed here must give the same draws (same torch build on the GPU box)."""
    g = golden("head_ws2_rate03")
    for r in range(2):
        torch.manual_seed(1000 + r)
        u = torch.rand(int(g["r%d_num_local" % r]))
        assert np.array_equal(u.numpy(), g["r%d_u" % r])


def test_sample_fewer_slots_than_positives():
    ll = torch.tensor([5, 7, 7, -1, 2, 9])
    idx, rel = head_ref.sample_index(ll, 10, 2, torch.rand(10))
    assert idx.tolist() == [2, 5, 7, 9]            # index = positives when they do not fit
    assert rel.tolist() == [1, 2, 2, -1, 0, 3]


def test_shard_arithmetic_covers_all_classes():
    for C, ws in [(86690, 8), (122000, 8), (1003, 2), (7, 8)]:
        spans = [head_ref.shard_range(C, ws, r) for r in range(ws)]
        assert spans[0][0] == 0 and sum(n for _, n in spans) == C
        for (a, n), (b, _) in zip(spans, spans[1:]):
            assert a + n == b
