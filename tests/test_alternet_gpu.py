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


def test_alternet50_fp32_eval_and_bf16_train(golden):
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
    net16 = A.Encoder(types.SimpleNamespace(network="AlterNet50", emd_size=512, img_size=192, frhip_dtype="bf16"))
    net16.load_state_dict(sd, strict=True)
    net16 = net16.cuda().train()
    out = net16(recipe.images(7301, 4, 192, 192).cuda())
    out.sum().backward()
    assert torch.isfinite(out).all() and all(torch.isfinite(p.grad).all() for p in net16.parameters())
