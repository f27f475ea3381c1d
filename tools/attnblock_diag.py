#!/usr/bin/env python3
"""GPU diagnostic: one AlterNet attention block (x + BN(attention)) of the product in fp32 mode against the oracle's restatement on the same
random inputs; prints max |d| / rms per tensor.  Usage (GPU box): python tools/attnblock_diag.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "face-recognition-pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from oracle import alternet_ref, recipe  # noqa: E402
import nets.AlterNet_SwinV2_FAN as A  # noqa: E402
from nets._backbone import BackwardCtx  # noqa: E402


def e(a, r):
    a, r = np.asarray(a, dtype=np.float64), np.asarray(r, dtype=np.float64)
    return np.abs(a - r).max() / max(np.sqrt((r ** 2).mean()), 1e-30)


def run(b, c, heads, ws, res, shift, dt64=False):
    spec = alternet_ref.attn_block_spec("blk", c, heads, ws, shift, res)
    sd = alternet_ref.fill_special(recipe.fill_state(spec, 31 + ws + shift), spec)
    blk = A.SwinTransformerBlock(c, c, heads=heads, input_resolution=(res, res), window_size=ws, shift_size=shift)
    blk.drop_path_rate = 0.0
    blk.load_state_dict({k[4:]: v for k, v in sd.items()}, strict=True)
    blk = blk.cuda().train()
    x = recipe.normal(77, (b, c, res, res))
    gy = recipe.normal(78, (b, c, res, res))
    # oracle
    odt = torch.float64 if dt64 else torch.float32
    osd = {k: (v.to(odt) if v.is_floating_point() else v).clone() for k, v in sd.items()}
    names = [k for k, _, kind in spec if kind in ("conv", "linear_w", "linear_b", "bn_w", "bn_b", "logit_scale")]
    for k in names:
        osd[k].requires_grad_(True)
    xo = x.to(odt).requires_grad_(True)
    yo = alternet_ref.attn_block(osd, "blk", xo, heads, ws, shift, True)
    yo.backward(gy.to(odt))
    # product
    xg = x.permute(0, 2, 3, 1).contiguous().cuda()
    y, s = A.attn_block_forward(blk, xg, torch.float32, True, True)
    bc = BackwardCtx(list(blk.parameters()), xg.device)
    dx = A.attn_block_backward(blk, s, gy.permute(0, 2, 3, 1).contiguous().cuda(), torch.float32, bc)
    grads = bc.join()
    torch.cuda.synchronize()
    msg = ["b%d c%d h%d ws%d res%d shift%d %s:" % (b, c, heads, ws, res, shift, "f64" if dt64 else "f32"),
           "out %.1e" % e(y.permute(0, 3, 1, 2).cpu().numpy(), yo.detach().numpy()), "dx %.1e" % e(dx.permute(0, 3, 1, 2).cpu().numpy(), xo.grad.numpy())]
    for k, p in blk.named_parameters():
        want = osd["blk." + k].grad
        if want is None or float(want.abs().max()) < 1e-6:
            continue
        msg.append("%s %.1e" % (k.replace("attn.", ""), e(grads[p].cpu().numpy().reshape(want.shape), want.numpy())))
    print(" ".join(msg), flush=True)


for dt64 in (False, True):
    run(8, 512, 16, 3, 6, 0, dt64)
    run(8, 512, 16, 3, 6, 1, dt64)
    run(2, 512, 16, 3, 6, 1, dt64)
    run(8, 256, 8, 6, 12, 3, dt64)
    run(4, 128, 4, 6, 24, 0, dt64)
