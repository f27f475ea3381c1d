"""Pin oracle/alternet_ref.py to vectors produced by the real reference nets/AlterNet_SwinV2_FAN.py."""
import numpy as np
import pytest
import torch

from oracle import alternet_ref, recipe

NOISE = ("proj.bias", "v_bias")


@pytest.mark.parametrize("tag", ["c128_w6", "c512_w3"])
def test_attention_pair(golden, tag):
    g = golden("alternet_pair_" + tag)
    c, heads, ws, res = int(g["c"]), int(g["heads"]), int(g["ws"]), int(g["res"])
    x = recipe.normal(7101, (2, c, res, res)).requires_grad_(True)
    sds, y = [], x
    for j, shift in enumerate((0, ws // 2)):
        spec = alternet_ref.attn_block_spec("blk", c, heads, ws, shift, res)
        sd = alternet_ref.fill_special(recipe.fill_state(spec, 7000 + 10 * heads + j), spec)
        names = [k for k, _, kind in spec if kind in ("linear_w", "linear_b", "bn_w", "bn_b", "logit_scale")]
        for k in names:
            sd[k] = sd[k].clone().requires_grad_(True)
        y = alternet_ref.attn_block(sd, "blk", y, heads, ws, shift, True)
        sds.append((sd, names))
    y.backward(recipe.normal(7102, (2, c, res, res)))
    np.testing.assert_allclose(y.detach().numpy(), g["out"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(x.grad.numpy(), g["dx"], rtol=1e-3, atol=1e-5)
    for j, (sd, names) in enumerate(sds):
        for k in names:
            want = g["b%d.grad.%s" % (j, k[4:])]
            got = sd[k].grad
            got = recipe.summary(got) if (want.shape == (10,) and got.numel() != 10) else got.numpy()
            np.testing.assert_allclose(got, want, rtol=2e-3, atol=2e-4 if k.endswith(NOISE) else 2e-5, err_msg=k)


def test_alternet50_eval(golden):
    g = golden("alternet50_b2_eval")
    spec = alternet_ref.alter_spec("AlterNet50")
    assert len(spec) == int(g["n_keys"])
    sd = alternet_ref.fill_special(recipe.fill_state(spec, 7300), spec)
    with torch.no_grad():
        y = alternet_ref.alter_forward(sd, recipe.images(7301, 2, 192, 192), "AlterNet50", False)
    np.testing.assert_allclose(y.numpy(), g["out"], rtol=1e-3, atol=1e-4)


def test_alternet50_whole_net_training_mode(golden):
    """/root/reference/nets/AlterNet_SwinV2_FAN.py:637-751 in training mode at 192 x 192, batch 8: stride-2 stem, conv <-> (W-MSA, SW-MSA)
    interleave, bn2 -> ReLU -> Dropout(p = 0 here) -> AAP(6,6) -> fc -> bn3 tail"""
    from wholenet import check_whole_net_train
    g = golden("alternet50_b8_train")
    spec = alternet_ref.alter_spec("AlterNet50")
    sd = alternet_ref.fill_special(recipe.fill_state(spec, int(g["seed"])), spec)
    names = [k for k, _, kind in spec if kind in ("conv", "linear_w", "linear_b", "bn_w", "bn_b", "logit_scale")]
    for k in names:
        sd[k].requires_grad_(True)
    y = alternet_ref.alter_forward(sd, recipe.images(int(g["seed"]) + 1, int(g["batch"]), 192, 192), "AlterNet50", True)
    y.backward(recipe.normal(int(g["seed"]) + 2, tuple(y.shape), 0.05))
    assert {"gprobe." + k for k in names} == {k for k in g if k.startswith("gprobe.")}
    check_whole_net_train(g, {k: sd[k].grad for k in names}, y.detach().numpy(), {k: v.detach() for k, v in sd.items()}, noise=("fc.bias",))


def test_shipped_recipe_two_steps(golden):
    """/root/reference/main/train.sh:12 end to end (AlterNet50 @192 + PartialFCAdamW rate 0.3 + AdamW lr 5e-4 + clip 5), two steps on fresh batches"""
    from oracle import train_ref
    g = golden("recipe_alternet50_adamw_rate03")
    C, B, steps, rate, lr = int(g["C"]), int(g["B"]), int(g["steps"]), float(g["rate"]), float(g["lr"])
    spec = alternet_ref.alter_spec("AlterNet50")
    sd = alternet_ref.fill_special(recipe.fill_state(spec, int(g["seed"])), spec)
    names = [k for k, _, kind in spec if kind in ("conv", "linear_w", "linear_b", "bn_w", "bn_b", "logit_scale")]
    W = recipe.normal(9101, (C, 512), 0.01)
    opt = train_ref.AdamWState(lr, tuple(g["betas"]), float(g["eps"]), float(g["wd"]))
    fwd = lambda work, img: alternet_ref.alter_forward(work, img, "AlterNet50", True)      # noqa: E731
    for st in range(steps):
        img, ids = recipe.images(9110 + 10 * st, B, 192, 192), recipe.labels(9111 + 10 * st, B, C)
        torch.manual_seed(9200 + st)
        u = [torch.rand(C)]
        out = train_ref.train_step(sd, W, img, ids, None, C, opt, sample_rate=rate, uniforms=u, forward=fwd, names=names)
        np.testing.assert_allclose(out["loss"].item(), g["losses"][st], rtol=1e-3 if st == 0 else 1e-2)
        np.testing.assert_allclose(out["grad_norm"].item(), g["grad_norms"][st], rtol=5e-3 if st == 0 else 5e-2)
        assert np.array_equal(out["index"].numpy(), g["index_step%d" % st])
        if st == 0:
            for k in [k[6:] for k in g if k.startswith("grad0.")]:
                want = g["grad0." + k]
                np.testing.assert_allclose(recipe.probe(out["grads"][k])[1:], want[1:], rtol=1e-2, atol=5e-2 * want[1] / out["grads"][k].numel() ** 0.5, err_msg=k)
    for k in [k[6:] for k in g if k.startswith("after.") and not k.startswith("after.head")]:
        got, want = recipe.probe(sd[k].float()), g["after." + k]
        np.testing.assert_allclose(got[1], want[1], rtol=2e-3, err_msg=k)
        assert np.abs(got[2:] - want[2:]).max() <= 2.2 * lr * steps, k
