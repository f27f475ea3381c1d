"""The HIP AlterNet path (nets.AlterNet_SwinV2_FAN drop-in) against reference-generated fixtures."""
import types

import numpy as np
import pytest
import torch

from oracle import alternet_ref, recipe

pytestmark = pytest.mark.gpu
NOISE = ("proj.bias", "v_bias")


@pytest.mark.parametrize("tag", ["c128_w6", "c512_w3"])
def test_attention_pair_fp32_matches_reference_fixture(golden, tag):
    """(W-MSA, SW-MSA) pair: cyclic roll + region mask + 6x6 / 3x3 windows inside the kernel, fwd + bwd"""
    import nets.AlterNet_SwinV2_FAN as A
    from nets._backbone import BackwardCtx
    g = golden("alternet_pair_" + tag)
    c, heads, ws, res = int(g["c"]), int(g["heads"]), int(g["ws"]), int(g["res"])
    blks = []
    for j, shift in enumerate((0, ws // 2)):
        blk = A.SwinTransformerBlock(c, c, heads=heads, input_resolution=(res, res), window_size=ws, shift_size=shift)
        blk.drop_path_rate = 0.0
        spec = alternet_ref.attn_block_spec("blk", c, heads, ws, shift, res)
        sd = alternet_ref.fill_special(recipe.fill_state(spec, 7000 + 10 * heads + j), spec)
        blk.load_state_dict({k[4:]: v for k, v in sd.items()}, strict=True)
        blks.append(blk.cuda().train())
    x = recipe.normal(7101, (2, c, res, res)).permute(0, 2, 3, 1).contiguous().cuda()
    gy = recipe.normal(7102, (2, c, res, res)).permute(0, 2, 3, 1).contiguous().cuda()
    y0, s0 = A.attn_block_forward(blks[0], x, torch.float32, True, True)
    y1, s1 = A.attn_block_forward(blks[1], y0, torch.float32, True, True)
    params = [p for b in blks for p in b.parameters()]
    bc = BackwardCtx(params, x.device)
    d1 = A.attn_block_backward(blks[1], s1, gy, torch.float32, bc)
    d0 = A.attn_block_backward(blks[0], s0, d1, torch.float32, bc)
    grads = bc.join()
    torch.cuda.synchronize()
    np.testing.assert_allclose(y1.permute(0, 3, 1, 2).cpu().numpy(), g["out"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(d0.permute(0, 3, 1, 2).cpu().numpy(), g["dx"], rtol=2e-3, atol=2e-4)
    for j, blk in enumerate(blks):
        for k, p in blk.named_parameters():
            want = g["b%d.grad.%s" % (j, k)]
            got = grads[p].cpu()
            if want.shape == (10,) and got.numel() != 10:
                np.testing.assert_allclose(recipe.summary(got), want, rtol=5e-3, atol=5e-3 * abs(want[1]) + 1e-4, err_msg=k)
            else:
                np.testing.assert_allclose(got.numpy().reshape(want.shape), want, rtol=5e-3,
                                           atol=(2e-3 if k.endswith(NOISE) else 5e-3 * np.abs(want).max() + 1e-5), err_msg=k)


def test_alternet50_fp32_eval(golden):
    import nets.AlterNet_SwinV2_FAN as A
    g = golden("alternet50_b2_eval")
    spec = alternet_ref.alter_spec("AlterNet50")
    sd = alternet_ref.fill_special(recipe.fill_state(spec, 7300), spec)
    net = A.Encoder(types.SimpleNamespace(network="AlterNet50", emd_size=512, img_size=192, frhip_dtype="fp32"))
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    with torch.no_grad():
        y = net(recipe.images(7301, 2, 192, 192).cuda())
    np.testing.assert_allclose(y.cpu().numpy(), g["out"], rtol=1e-3, atol=3e-4)



def _alternet50(dtype, seed):
    import nets.AlterNet_SwinV2_FAN as A
    spec = alternet_ref.alter_spec("AlterNet50")
    sd = alternet_ref.fill_special(recipe.fill_state(spec, seed), spec)
    net = A.Encoder(types.SimpleNamespace(network="AlterNet50", emd_size=512, img_size=192, frhip_dtype=dtype))
    net.load_state_dict(sd, strict=True)
    return net.cuda()


def test_alternet50_whole_net_training_mode_fp32_matches_reference_fixture(golden):
    """BASELINE cfg 5's network (/root/reference/nets/AlterNet_SwinV2_FAN.py:637-751) in training mode against the real reference, batch 8 at
    192 x 192: stride-2 stem (im2col route), conv <-> (W-MSA, SW-MSA) interleave, bn2 -> ReLU -> Dropout(p = 0) -> AAP(6,6) -> fc -> bn3."""
    from wholenet import check_whole_net_train, whole_net_train_on_gpu
    g = golden("alternet50_b8_train")
    grads, out, bufs = whole_net_train_on_gpu(_alternet50("fp32", int(g["seed"])), g, 192, 192)
    check_whole_net_train(g, grads, out, bufs, rtol=2e-3, noise=("fc.bias",), kink_rtol=5e-2,
                          kink_free=("fc.", "bn3.", "bn2.", "layer4.3.norm2.", "layer4.3.attn.proj."))


def test_alternet50_bf16_training_step_tracks_the_reference_fixture(golden):
    """bf16 MFMA mode on the fixture's inputs.  This randomly initialised 50-block network at batch 8 amplifies bf16 STORAGE rounding to a
    20 % embedding error in a plain PyTorch emulation with no HIP code (wholenet.alternet50_bf16_storage_emulation: 0.197); the HIP path must
    not be worse than that by more than 15 %, and its large gradients must point the reference's way."""
    from wholenet import alternet50_bf16_storage_emulation, whole_net_train_on_gpu
    g = golden("alternet50_b8_train")
    grads, out, _ = whole_net_train_on_gpu(_alternet50("bf16", int(g["seed"])), g, 192, 192)
    assert np.isfinite(out).all() and all(torch.isfinite(v).all() for v in grads.values())
    emu = alternet50_bf16_storage_emulation(g)
    err = float(np.linalg.norm(out - g["out"]) / np.linalg.norm(g["out"]))
    assert err <= 1.15 * emu + 5e-3, (err, emu)
    for k in [k[6:] for k in g if k.startswith("gfull.")]:
        want = g["gfull." + k].reshape(-1).astype(np.float64)
        got = grads[k].numpy().reshape(-1).astype(np.float64)
        if want.size >= 16384:              # (the 2 048-element position-bias MLP gradients of stage 2 sit at 0.79 after 40 bf16 blocks)
            cos = float(got @ want / (np.linalg.norm(got) * np.linalg.norm(want)))
            assert cos >= 0.80, (k, cos)


def test_stochastic_depth_reductions_fused_into_the_producing_data_gradient(monkeypatch):
    """AlterNet50 in training mode WITH stochastic depth (drop_path 0.1, the reference default): the BatchNorm-backward sums of an attention
    block's norm2 ride in the data-gradient that produces its incoming gradient (frhip_conv_dgrad_fused_rs) instead of a pass of their own --
    every parameter gradient must equal the unfused route's on the same per-sample keep draws."""
    import nets.AlterNet_SwinV2_FAN as A
    x = recipe.images(7501, 8, 192, 192).cuda()
    gy = recipe.normal(7502, (8, 512), 0.05).cuda()
    res = []
    for fused in (True, False):
        monkeypatch.setattr(A, "_FUSE_BNRED_RS", fused)
        net = _alternet50("fp32", 7500).train()
        net.dropout.p = 0.0
        torch.manual_seed(7503)                      # the keep masks come from torch's device generator
        y = net(x)
        y.backward(gy)
        torch.cuda.synchronize()
        res.append((y.detach().float().cpu(), {k: p.grad.float().cpu() for k, p in net.named_parameters()}))
    (ya, ga), (yb, gb) = res
    assert torch.equal(ya, yb)                       # same forward, same keeps
    dropped = 0
    for k in ga:
        if k.endswith(NOISE) or k == "fc.bias":          # analytically-zero gradients: round-off only
            continue
        a, b = ga[k].double(), gb[k].double()
        denom = float(b.norm()) + 1e-12
        assert float((a - b).norm()) <= 1e-3 * denom + 1e-7, (k, float((a - b).norm()) / denom)
        dropped += 1
    assert dropped > 150
