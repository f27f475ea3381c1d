#!/usr/bin/env python3
"""GPU diagnostic: fp32-arithmetic window attention kernels (csrc/winattn.hip) against a torch autograd restatement, shifted / masked and
small windows included; prints max |d| / rms per output.  Usage (GPU box): python tools/winattn_diag.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "face-recognition-pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from oracle import alternet_ref, swin_ref  # noqa: E402
from frhip import ops  # noqa: E402
from frhip._abi import lib  # noqa: E402


def run(b, hw, c, heads, ws, shift, scale_mu):
    n = ws * ws
    g = torch.Generator().manual_seed(hw * ws + shift + c)
    qkv = torch.randn((b * hw * hw, 3 * c), generator=g)
    dout = torch.randn((b * hw * hw, c), generator=g)
    bias = 16 * torch.sigmoid(torch.randn((heads, n, n), generator=g))
    scale = torch.exp(torch.randn(heads, generator=g) * 0.3 + scale_mu)
    qr, br, sr = qkv.clone().requires_grad_(True), bias.clone().requires_grad_(True), scale.clone().requires_grad_(True)
    y = qr.view(b, hw, hw, 3 * c)
    if shift:
        y = torch.roll(y, shifts=(-shift, -shift), dims=(1, 2))
    xw = swin_ref.to_windows(y, ws)
    q, k, v = [t.reshape(-1, n, heads, 32).transpose(1, 2) for t in xw.split(c, dim=-1)]
    attn = torch.nn.functional.normalize(q, dim=-1) @ torch.nn.functional.normalize(k, dim=-1).transpose(-2, -1)
    attn = attn * sr.view(1, heads, 1, 1) + br.unsqueeze(0)
    if shift:
        mask = alternet_ref.shift_mask(hw, hw, ws, shift)
        nw = mask.shape[0]
        attn = (attn.view(-1, nw, heads, n, n) + mask.unsqueeze(1).unsqueeze(0)).view(-1, heads, n, n)
    attn = torch.softmax(attn, dim=-1)
    o = swin_ref.from_windows((attn @ v).transpose(1, 2).reshape(-1, n, c), b, hw, hw, ws)
    if shift:
        o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
    ref = o.reshape(-1, c)
    ref.backward(dout)
    old = lib().frhip_set_winattn_mfma(0)
    try:
        out = ops.winattn_fwd(qkv.cuda(), bias.cuda(), scale.cuda(), b, hw, hw, heads, ws=ws, shift=shift)
        res = ops.winattn_bwd(qkv.cuda(), dout.cuda(), bias.cuda(), scale.cuda(), b, hw, hw, heads, ws=ws, shift=shift)
    finally:
        lib().frhip_set_winattn_mfma(old)
    dqkv, dbias, dscale = res[0], res[1], res[2]

    def e(a, r):
        a, r = a.float().cpu().numpy().astype(np.float64), r.detach().numpy().astype(np.float64)
        return np.abs(a - r).max() / np.sqrt((r ** 2).mean())
    c3 = dqkv.shape[1] // 3
    print("b%d hw%d c%d h%d ws%d shift%d mu%.1f: out %.2e  dq %.2e dk %.2e dv %.2e  dbias %.2e  dscale %.2e" % (
        b, hw, c, heads, ws, shift, scale_mu, e(out, ref), e(dqkv[:, :c3], qr.grad[:, :c3]), e(dqkv[:, c3:2 * c3], qr.grad[:, c3:2 * c3]),
        e(dqkv[:, 2 * c3:], qr.grad[:, 2 * c3:]), e(dbias, br.grad), e(dscale, sr.grad)))


for mu in (1.5, 2.7):
    for geom in [(2, 14, 128, 4, 7, 0), (2, 12, 128, 4, 6, 0), (2, 12, 128, 4, 6, 3), (8, 6, 512, 16, 3, 0), (8, 6, 512, 16, 3, 1), (2, 14, 64, 2, 7, 3)]:
        run(*geom, mu)
