"""The HIP SwinV2 path (nets.SwinV2 drop-in) against reference-generated fixtures and the oracle."""
import types

import numpy as np
import pytest
import torch

from oracle import recipe, swin_ref

pytestmark = pytest.mark.gpu
NOISE = ("proj.bias", "fc2.bias", "v_bias")        # analytically zero gradients (they only shift a train-mode BN input)


def _rb(x):
    """round to bf16 in the forward pass, identity in the backward pass: the casts torch autocast puts in front of `@`"""
    return x + (x.bfloat16().float() - x).detach()


@pytest.mark.parametrize("impl", ["f32", "bf16-valu", "bf16-mfma"])
@pytest.mark.parametrize("shape", [(3, 14, 128, 4), (2, 7, 512, 16)])
def test_window_attention_kernel(impl, shape):
    """frhip_winattn_fwd / _bwd against the oracle's cosine window attention on the same q, k, v.  The bf16 MFMA kernels
    feed both GEMMs bf16 operands exactly where the reference's autocast does (nets/SwinV2.py:160-176: F.normalize and
    softmax are fp32 autocast ops, the two `@` cast their inputs to bf16), so their oracle carries those roundings."""
    from frhip import ops
    from frhip._abi import lib
    dtype = torch.float32 if impl == "f32" else torch.bfloat16
    cast = _rb if impl == "bf16-mfma" else (lambda x: x)
    b, hw, c, heads = shape
    g = torch.Generator().manual_seed(b * hw + c)
    qkv = torch.randn((b * hw * hw, 3 * c), generator=g).to(dtype).float()
    bias = 16 * torch.sigmoid(torch.randn((heads, 49, 49), generator=g))
    scale = torch.exp(torch.randn(heads, generator=g) * 0.3 + 2.0)
    dout = torch.randn((b * hw * hw, c), generator=g).to(dtype).float()
    # oracle on windows
    qr = qkv.clone().requires_grad_(True)
    br, sr = bias.clone().requires_grad_(True), scale.clone().requires_grad_(True)
    xw = swin_ref.to_windows(qr.view(b, hw, hw, 3 * c))                       # [B_, 49, 3C]
    q, k, v = [t.reshape(-1, 49, heads, 32).transpose(1, 2) for t in xw.split(c, dim=-1)]
    attn = cast(torch.nn.functional.normalize(q, dim=-1)) @ cast(torch.nn.functional.normalize(k, dim=-1)).transpose(-2, -1)
    attn = torch.softmax(attn * sr.view(1, heads, 1, 1) + br.unsqueeze(0), dim=-1)
    ref = swin_ref.from_windows((cast(attn) @ v).transpose(1, 2).reshape(-1, 49, c), b, hw, hw).reshape(-1, c)
    ref.backward(dout)
    old = lib().frhip_set_winattn_mfma(1 if impl == "bf16-mfma" else 0)
    try:
        out = ops.winattn_fwd(qkv.to(dtype).cuda(), bias.cuda(), scale.cuda(), b, hw, hw, heads)
        dqkv, dbias, dscale = ops.winattn_bwd(qkv.to(dtype).cuda(), dout.to(dtype).cuda(), bias.cuda(), scale.cuda(), b, hw, hw, heads)
    finally:
        lib().frhip_set_winattn_mfma(old)
    t = dict(rtol=2e-4, atol=2e-5) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-2)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref.detach().numpy(), **t)
    sc = qr.grad.abs().max().item()
    np.testing.assert_allclose(dqkv.float().cpu().numpy(), qr.grad.numpy(), rtol=t["rtol"], atol=t["atol"] * max(sc, 1.0))
    # d(bias), d(scale) are fp32 sums over all windows; the MFMA path's dS differs from the oracle's only by the bf16
    # rounding of dP's operands being applied in a different order -> same tolerance as the fp32-arithmetic kernels
    np.testing.assert_allclose(dbias.cpu().numpy(), br.grad.numpy(), rtol=2e-3, atol=2e-3 * br.grad.abs().max().item())
    np.testing.assert_allclose(dscale.cpu().numpy(), sr.grad.numpy(), rtol=2e-3, atol=2e-3 * sr.grad.abs().max().item())


@pytest.mark.parametrize("geom", [(2, 12, 128, 4, 6, 3), (2, 12, 128, 4, 6, 0), (3, 6, 512, 16, 3, 1), (2, 14, 64, 2, 7, 3)])
def test_window_attention_mfma_matches_valu_kernels(geom):
    """shifted / masked and 6x6, 3x3 windows: the bf16 MFMA kernels against the fp32-arithmetic kernels (which the
    reference fixtures of test_alternet_gpu pin) on the same bf16 operands"""
    from frhip import ops
    from frhip._abi import lib
    b, hw, c, heads, ws, shift = geom
    n = ws * ws
    g = torch.Generator().manual_seed(hw * ws + shift)
    qkv = torch.randn((b * hw * hw, 3 * c), generator=g).bfloat16().cuda()
    dout = torch.randn((b * hw * hw, c), generator=g).bfloat16().cuda()
    bias = (16 * torch.sigmoid(torch.randn((heads, n, n), generator=g))).cuda()
    scale = torch.exp(torch.randn(heads, generator=g) * 0.3 + 1.5).cuda()
    res = {}
    for mode in (0, 1):
        old = lib().frhip_set_winattn_mfma(mode)
        try:
            out = ops.winattn_fwd(qkv, bias, scale, b, hw, hw, heads, ws=ws, shift=shift)
            res[mode] = (out,) + ops.winattn_bwd(qkv, dout, bias, scale, b, hw, hw, heads, ws=ws, shift=shift, want_colsum=True)
        finally:
            lib().frhip_set_winattn_mfma(old)
    torch.cuda.synchronize()
    assert res[0][4] is None                      # the fp32-arithmetic kernels leave the column sums to the caller
    colsum = res[1][4].cpu().numpy()              # fused q_bias / v_bias gradient = column sums of the STORED dqkv
    stored = res[1][1].float().sum(0).cpu().numpy()
    np.testing.assert_allclose(colsum, stored, rtol=1e-3, atol=1e-3 * np.abs(res[1][1].float().cpu().numpy()).sum(0).max())
    # the q / v thirds added straight into existing gradient accumulators (frhip_winattn_bwd_qvbias): same dqkv bits, accumulators = their
    # old contents + the stored tensor's column sums
    gq, gv = torch.full((c,), 0.5, device="cuda"), torch.full((c,), -0.25, device="cuda")
    direct = ops.winattn_bwd(qkv, dout, bias, scale, b, hw, hw, heads, ws=ws, shift=shift, want_colsum=True, qv_grads=(gq, gv))
    assert direct[3] is True and torch.equal(direct[0], res[1][1])
    lim = 1e-3 * np.abs(res[1][1].float().cpu().numpy()).sum(0).max()
    np.testing.assert_allclose(gq.cpu().numpy() - 0.5, stored[:c], rtol=1e-3, atol=lim)
    np.testing.assert_allclose(gv.cpu().numpy() + 0.25, stored[2 * c:], rtol=1e-3, atol=lim)
    names = ("out", "dqkv", "dbias", "dscale")
    for name, a, r in zip(names, res[1], res[0]):
        a, r = a.float().cpu().numpy(), r.float().cpu().numpy()
        tol = 4e-2 if name in ("out", "dqkv") else 2e-2       # operand roundings of the bf16 GEMMs (q^, k^, P, dS)
        np.testing.assert_allclose(a, r, rtol=tol, atol=tol * max(np.abs(r).max(), 1e-3), err_msg=name)


@pytest.mark.parametrize("tag", ["c128h4", "c512h16"])
def test_swin_block_fp32_matches_reference_fixture(golden, tag):
    import nets.SwinV2 as S
    from nets._backbone import BackwardCtx
    g = golden("swin_block_" + tag)
    c, heads, hw = int(g["c"]), int(g["heads"]), int(g["hw"])
    blk = S.SwinTransformerBlock(c, c, heads=heads)
    spec = swin_ref.block_spec("blk", c, heads)
    sd = swin_ref.fill_special(recipe.fill_state(spec, 6100 + heads), spec)
    blk.load_state_dict({k[4:]: v for k, v in sd.items()}, strict=True)
    blk = blk.cuda().train()
    x = recipe.normal(6101, (3, c, hw, hw))
    gy = recipe.normal(6102, (3, c, hw, hw))
    xh = x.permute(0, 2, 3, 1).contiguous().cuda()
    out, s = S.swin_block_forward(blk, xh, torch.float32, True, True)
    params = list(blk.parameters())
    bc = BackwardCtx(params, xh.device)
    dx = S.swin_block_backward(blk, s, gy.permute(0, 2, 3, 1).contiguous().cuda(), torch.float32, bc)
    grads = bc.join()
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.permute(0, 3, 1, 2).cpu().numpy(), g["out"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(dx.permute(0, 3, 1, 2).cpu().numpy(), g["dx"], rtol=2e-3, atol=2e-4)
    for k, p in blk.named_parameters():
        want = g["grad." + k]
        got = grads[p].cpu()
        noise = k.endswith(NOISE)
        if want.shape == (10,) and got.numel() != 10:
            np.testing.assert_allclose(recipe.summary(got), want, rtol=5e-3, atol=5e-3 * abs(want[1]) + 1e-4, err_msg=k)
        else:
            np.testing.assert_allclose(got.numpy().reshape(want.shape), want, rtol=5e-3,
                                       atol=(2e-3 if noise else 5e-3 * np.abs(want).max() + 1e-5), err_msg=k)
    for k, bf in blk.named_buffers():
        if "running" in k:
            np.testing.assert_allclose(bf.cpu().numpy(), g["after." + k], rtol=1e-3, atol=1e-5, err_msg=k)


def _net(name, dtype, seed):
    import nets.SwinV2 as S
    net = S.Encoder(types.SimpleNamespace(network=name, emd_size=512, frhip_dtype=dtype))
    spec = swin_ref.swin_spec(name)
    sd = swin_ref.fill_special(recipe.fill_state(spec, seed), spec)
    net.load_state_dict(sd, strict=True)
    return net.cuda()


def test_swin18_fp32_eval_and_train_match_reference_fixture(golden):
    g = golden("swin18_b2")
    net = _net("Swin18", "fp32", 6200)
    x = recipe.images(6201, 2).cuda()
    net.eval()
    with torch.no_grad():
        np.testing.assert_allclose(net(x).cpu().numpy(), g["eval_out"], rtol=1e-3, atol=3e-4)
    net = _net("Swin18", "fp32", 6200)
    net.train()
    net.dropout.p = 0.0                  # the fixture's RNG-free training pass
    y = net(x)
    y.backward(recipe.normal(6202, (2, 512), 0.05).cuda())
    # training-mode BatchNorm1d over a batch of TWO: each output is +-(f1-f2)/sqrt((f1-f2)^2/4+eps), ill-conditioned
    # wherever the two samples nearly agree -> a handful of entries move by 1e-2 for 1e-6 input noise
    diff = np.abs(y.detach().cpu().numpy() - g["train_out"])
    assert np.median(diff) < 1e-5 and (diff > 3e-4).mean() < 0.02 and diff.max() < 5e-2, (np.median(diff), diff.max())
    # the same ill-conditioning scales every gradient that flows through bn3 (d out / d f ~ 1/|f1-f2|): whole-net
    # gradients are held to 6 % of each tensor's l2 norm here; the tight per-block gradient parity is
    # test_swin_block_fp32_matches_reference_fixture and test_window_attention_kernel
    for k, p in net.named_parameters():
        want = g["gsum." + k]
        if k.endswith(NOISE) or k in ("fc.bias", "bn2.bias"):     # tail: a constant shift in front of train-mode bn3 -> zero gradient
            continue
        got = recipe.summary(p.grad.cpu())
        np.testing.assert_allclose(got[1], want[1], rtol=6e-2, atol=1e-4, err_msg=k)      # atol: analytically-zero grads
        np.testing.assert_allclose(got[2:], want[2:], rtol=6e-2, atol=6e-2 * abs(want[1]) + 1e-4, err_msg=k)


def test_swin34_fp32_eval(golden):
    g = golden("swin34_b2")
    net = _net("Swin34", "fp32", 6300)
    net.eval()
    with torch.no_grad():
        np.testing.assert_allclose(net(recipe.images(6301, 2).cuda()).cpu().numpy(), g["eval_out"], rtol=1e-3, atol=3e-4)


def test_swin34_whole_net_training_mode_fp32_matches_reference_fixture(golden):
    """BASELINE cfg 4's network (/root/reference/nets/SwinV2.py:534-565 at depth 34) in training mode against the real reference, batch 8:
    embeddings, EVERY parameter gradient (l2 + 256 elements at portable positions), full tensors of the stem / downsample convs, the first
    attention block's qkv / position-bias-MLP / logit-scale / q-bias gradients and the tail BatchNorms, running statistics."""
    from wholenet import check_whole_net_train, whole_net_train_on_gpu
    g = golden("swin34_b8_train")
    grads, out, bufs = whole_net_train_on_gpu(_net("Swin34", "fp32", int(g["seed"])), g)
    check_whole_net_train(g, grads, out, bufs, rtol=2e-3, noise=("fc.bias", "bn2.bias"))      # measured: <= 6e-4 of the rms on every tensor


def test_swin34_bf16_training_step_tracks_the_reference_fixture(golden):
    """bf16 MFMA mode on the same inputs: embeddings within 5 %, the large gradients point the reference's way"""
    from wholenet import whole_net_train_on_gpu
    g = golden("swin34_b8_train")
    grads, out, _ = whole_net_train_on_gpu(_net("Swin34", "bf16", int(g["seed"])), g)
    assert np.isfinite(out).all() and all(torch.isfinite(v).all() for v in grads.values())
    assert np.linalg.norm(out - g["out"]) <= 8e-2 * np.linalg.norm(g["out"])          # measured 5.5 % (ResNet50: 4 %)
    for k in [k[6:] for k in g if k.startswith("gfull.")]:
        want = g["gfull." + k].reshape(-1).astype(np.float64)
        got = grads[k].numpy().reshape(-1).astype(np.float64)
        if want.size >= 1024:
            cos = float(got @ want / (np.linalg.norm(got) * np.linalg.norm(want)))
            assert cos >= 0.90, (k, cos)


@pytest.mark.parametrize("ws,heads", [(7, 4), (6, 8), (3, 16)])
def test_position_bias_kernel_matches_torch_formula(ws, heads):
    """csrc/cpb.hip (one launch for a group of blocks, forward and backward) against the reference's formula in torch ops:
    16 * sigmoid(cpb_mlp(coords_table)[relative_position_index]) and exp(min(logit_scale, ln 100))
    (/root/reference/nets/SwinV2.py:150-158, nets/AlterNet_SwinV2_FAN.py:276-283), fp32, 1e-5."""
    import math
    import torch.nn.functional as F
    import nets.AlterNet_SwinV2_FAN as A
    import nets.SwinV2 as S
    from nets._backbone import BackwardCtx
    torch.manual_seed(ws)
    blks = []
    for i in range(3):
        blk = A.SwinTransformerBlock(heads * 32, heads * 32, heads=heads, input_resolution=(2 * ws, 2 * ws), window_size=ws).cuda()
        with torch.no_grad():
            blk.attn.logit_scale.add_(torch.randn_like(blk.attn.logit_scale))        # some heads beyond the ln(100) clamp
            blk.attn.logit_scale[0] = 5.0
        blks.append(blk)
    batch = S.precompute_position_bias(blks, torch.device("cuda"))
    params = [p for b in blks for p in b.attn.cpb_params()]
    bc = BackwardCtx(params, torch.device("cuda"))
    n = ws * ws
    for i, blk in enumerate(blks):
        at = blk.attn
        w0, b0, w2, ls = [p.detach().clone().requires_grad_(True) for p in at.cpb_params()]
        t = F.linear(F.relu(F.linear(at.relative_coords_table, w0, b0)), w2).view(-1, heads)
        bias_ref = 16 * torch.sigmoid(t[at.relative_position_index.view(-1)].view(n, n, heads).permute(2, 0, 1).contiguous())
        scale_ref = torch.clamp(ls, max=math.log(100.0)).exp().reshape(-1)
        _, _, bias, scale, dbias, dscale = S.position_bias(blk)
        np.testing.assert_allclose(bias.cpu().numpy(), bias_ref.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(scale.cpu().numpy(), scale_ref.detach().cpu().numpy(), rtol=1e-6)
        gb, gs = torch.randn_like(bias_ref), torch.randn_like(scale_ref)
        dbias.copy_(gb)
        dscale.copy_(gs)
        want = torch.autograd.grad([bias_ref, scale_ref], [w0, b0, w2, ls], [gb, gs])
        s = types.SimpleNamespace(cpb_batch=batch)
        S.position_bias_backward(blk, s, bc)
        blk._want = want
    grads = bc.join()
    for blk in blks:
        for p, w in zip(blk.attn.cpb_params(), blk._want):
            np.testing.assert_allclose(grads[p].cpu().numpy(), w.cpu().numpy(), rtol=2e-4, atol=2e-5 * float(w.abs().max()) + 1e-7)
