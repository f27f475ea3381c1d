"""Pin oracle/head_ref.py to vectors produced by the real reference (tools/make_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import head_ref, recipe, train_ref

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


@pytest.mark.parametrize("name", ["head_adamw_ws1_rate03", "head_adamw_ws2_rate03"])
def test_head_adamw_steps_match_reference(golden, name):
    """PartialFCAdamW + torch.optim.AdamW (reference nets/PartialFC.py:235-432, the shipped recipe main/train.sh:12), three steps with
    fresh embeddings each: per-step loss, sampled rows (bit-exact), dE; then the class centres and both moment tables after update()."""
    g = golden(name)
    ws, C, B, D, rate, steps = int(g["ws"]), int(g["C"]), int(g["B"]), int(g["D"]), float(g["rate"]), int(g["steps"])
    weights = [recipe.normal(500 + r, head_ref.shard_range(C, ws, r)[1:] + (D,), 0.05) for r in range(ws)]
    opts = [train_ref.AdamWState(float(g["lr"]), tuple(g["betas"]), float(g["eps"]), float(g["wd"])) for _ in range(ws)]
    for st in range(steps):
        embs, labs, us = [], [], []
        for r in range(ws):
            embs.append(recipe.normal(100 + r + 10 * st, (B, D)))
            lab = recipe.labels(200 + r + 10 * st, B, C)
            lab[0] = 3
            lab[1] = 3
            labs.append(lab)
            torch.manual_seed(1000 + r + 100 * st)              # the draw the reference made on rank r (nets/PartialFC.py:312)
            us.append(torch.rand(weights[r].shape[0]))
        out = head_ref.head_all_shards(embs, labs, weights, C, float(g["s"]), float(g["m"]), sample_rate=rate, uniforms=us)
        for r in range(ws):
            np.testing.assert_allclose(out["loss"].item(), g["r%d_loss_step%d" % (r, st)], rtol=3e-6)
            assert np.array_equal(out["index"][r].numpy(), g["r%d_index_step%d" % (r, st)])
            np.testing.assert_allclose(out["d_emb"][r].numpy(), g["r%d_d_emb_step%d" % (r, st)], rtol=2e-4, atol=2e-7)
            opts[r].head_update(weights[r], out["index"][r], out["d_w_act"][r], rate)
    for r in range(ws):
        np.testing.assert_allclose(weights[r].numpy(), g["r%d_weight" % r], rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(opts[r].m["head"].numpy(), g["r%d_exp_avg" % r], rtol=2e-4, atol=1e-8)
        np.testing.assert_allclose(opts[r].v["head"].numpy(), g["r%d_exp_avg_sq" % r], rtol=4e-4, atol=1e-12)
