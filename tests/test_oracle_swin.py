"""Pin oracle/swin_ref.py to vectors produced by the real reference nets/SwinV2.py (tools/make_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import recipe, swin_ref

T = torch.from_numpy


def _cmp_grad(got, want, rtol, atol):
    want = np.asarray(want)
    if want.shape == (10,) and got.numel() != 10:
        np.testing.assert_allclose(recipe.summary(got), want, rtol=rtol, atol=atol)
    else:
        np.testing.assert_allclose(got.numpy(), want, rtol=rtol, atol=atol)


@pytest.mark.parametrize("tag", ["c128h4", "c512h16"])
def test_swin_block(golden, tag):
    g = golden("swin_block_" + tag)
    c, heads, hw = int(g["c"]), int(g["heads"]), int(g["hw"])
    spec = swin_ref.block_spec("blk", c, heads)
    sd = swin_ref.fill_special(recipe.fill_state(spec, 6100 + heads), spec)
    names = [k for k, _, kind in spec if kind in ("conv", "linear_w", "linear_b", "bn_w", "bn_b", "logit_scale")]
    for k in names:
        sd[k] = sd[k].clone().requires_grad_(True)
    x = recipe.normal(6101, (3, c, hw, hw)).requires_grad_(True)
    y = swin_ref.swin_block(sd, "blk", x, heads, True)
    y.backward(recipe.normal(6102, (3, c, hw, hw)))
    np.testing.assert_allclose(y.detach().numpy(), g["out"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(x.grad.numpy(), g["dx"], rtol=1e-3, atol=1e-5)
    for k in names:
        # proj.bias / fc2.bias / v_bias only shift the input of a training-mode BatchNorm: their gradient is analytically ZERO and
        # numerically cancellation noise (~1e-5) in the reference too, so only its size is checked
        noise = k.endswith("proj.bias") or k.endswith("fc2.bias") or k.endswith("v_bias")
        _cmp_grad(sd[k].grad, g["grad." + k[4:]], 2e-3, 2e-4 if noise else 2e-5)
    for k in sd:
        if "running" in k:
            np.testing.assert_allclose(sd[k].detach().numpy(), g["after." + k[4:]], rtol=1e-5, atol=1e-6)


def test_swin18_eval_train_and_swin34_eval(golden):
    g = golden("swin18_b2")
    spec = swin_ref.swin_spec("Swin18")
    assert len(spec) == int(g["n_keys"])
    sd0 = swin_ref.fill_special(recipe.fill_state(spec, 6200), spec)
    x = recipe.images(6201, 2)
    with torch.no_grad():
        y = swin_ref.swin_forward({k: v.clone() for k, v in sd0.items()}, x, "Swin18", False)
    np.testing.assert_allclose(y.numpy(), g["eval_out"], rtol=1e-3, atol=1e-4)
    sd = {k: v.clone() for k, v in sd0.items()}
    names = [k for k, _, kind in spec if kind in ("conv", "linear_w", "linear_b", "bn_w", "bn_b", "logit_scale")]
    for k in names:
        sd[k].requires_grad_(True)
    y = swin_ref.swin_forward(sd, x, "Swin18", True)
    y.backward(recipe.normal(6202, (2, 512), 0.05))
    np.testing.assert_allclose(y.detach().numpy(), g["train_out"], rtol=1e-3, atol=1e-4)
    for k in names:
        noise = k.endswith("proj.bias") or k.endswith("fc2.bias") or k.endswith("v_bias") or k == "fc.bias"
        np.testing.assert_allclose(recipe.summary(sd[k].grad), g["gsum." + k], rtol=3e-3, atol=3e-3 if noise else 3e-5, err_msg=k)
    g34 = golden("swin34_b2")
    spec34 = swin_ref.swin_spec("Swin34")
    sd34 = swin_ref.fill_special(recipe.fill_state(spec34, 6300), spec34)
    with torch.no_grad():
        y34 = swin_ref.swin_forward(sd34, recipe.images(6301, 2), "Swin34", False)
    np.testing.assert_allclose(y34.numpy(), g34["eval_out"], rtol=1e-3, atol=1e-4)


from wholenet import check_whole_net_train


def test_swin34_whole_net_training_mode(golden):
    """/root/reference/nets/SwinV2.py:534-565 at depth 34, training mode (batch statistics everywhere), batch 8"""
    g = golden("swin34_b8_train")
    spec = swin_ref.swin_spec("Swin34")
    sd = swin_ref.fill_special(recipe.fill_state(spec, int(g["seed"])), spec)
    names = [k for k, _, kind in spec if kind in ("conv", "linear_w", "linear_b", "bn_w", "bn_b", "logit_scale")]
    for k in names:
        sd[k].requires_grad_(True)
    y = swin_ref.swin_forward(sd, recipe.images(int(g["seed"]) + 1, int(g["batch"])), "Swin34", True)
    y.backward(recipe.normal(int(g["seed"]) + 2, tuple(y.shape), 0.05))
    assert {"gprobe." + k for k in names} == {k for k in g if k.startswith("gprobe.")}
    check_whole_net_train(g, {k: sd[k].grad for k in names}, y.detach().numpy(), {k: v.detach() for k, v in sd.items()}, noise=("fc.bias", "bn2.bias"))
