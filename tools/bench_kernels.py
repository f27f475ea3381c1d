"""Micro-benchmark of the conv kernels on the ResNet50 body shapes (B=512) per NT tile choice.  GPU box only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "face-recognition-pytorch_amd")]
import torch
from frhip import ops
from frhip._abi import lib

B = int(os.environ.get("B", "512"))
SHAPES = [  # (h, c, k, r, stride)
    (56, 64, 64, 3, 1), (56, 64, 128, 3, 2), (28, 128, 128, 3, 1), (28, 128, 256, 3, 2),
    (14, 256, 256, 3, 1), (14, 256, 512, 3, 2), (7, 512, 512, 3, 1), (56, 64, 128, 1, 2),
]


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us


what = sys.argv[1] if len(sys.argv) > 1 else "fwd"
if what == "ew":
    for (h, c) in [(56, 64), (28, 128), (14, 256), (7, 512)]:
        rows = B * h * h
        y = torch.randn(rows, c, device="cuda").bfloat16()
        d = torch.randn(rows, c, device="cuda").bfloat16()
        gamma, beta = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
        st = ops.bn_finalize(ops.colstats(y), rows, gamma, beta, None, None)
        T = rows * c * 2 / 1e6   # MB per tensor
        out = torch.empty_like(y)
        dg, db = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
        nb = lib().frhip_colreduce_blocks(rows, c, 0)
        part = torch.empty((nb, 2, c), device="cuda")
        coef = torch.rand((3, c), device="cuda")
        P = ops._p
        S = ops._s
        t_apply = timeit(lambda: ops.bn_apply(y, st, relu=True, out=out))
        t_apply_res = timeit(lambda: ops.bn_apply(y, st, res=d, out=out))
        t_stats = timeit(lambda: lib().frhip_colstats(0, P(y), rows, c, P(part), S()))
        t_bred = timeit(lambda: lib().frhip_bn_bwd_reduce(0, P(d), P(y), P(st.mean), P(st.invstd), P(st.scale), P(st.shift), rows, c, P(part), S()))
        t_bapp = timeit(lambda: lib().frhip_bn_bwd_apply(0, P(d), P(y), P(coef[0]), P(coef[1]), P(coef[2]), P(st.scale), P(st.shift), P(out), rows, c, S()))
        print("h=%2d c=%3d T=%6.1fMB | apply %6.1fus %4.2fTB/s | apply+res %6.1fus %4.2fTB/s | colstats %6.1fus %4.2fTB/s | bwd_reduce %6.1fus %4.2fTB/s | bwd_apply %6.1fus %4.2fTB/s" % (
            h, c, T, t_apply, 2 * T / t_apply, t_apply_res, 3 * T / t_apply_res, t_stats, T / t_stats, t_bred, 2 * T / t_bred, t_bapp, 3 * T / t_bapp), flush=True)
    sys.exit(0)
if what == "attn":
    for (h, c, heads) in [(56, 64, 2), (28, 128, 4), (14, 256, 8), (7, 512, 16)]:
        rows = B * h * h
        qkv = torch.randn(rows, 3 * c, device="cuda").bfloat16()
        do = torch.randn(rows, c, device="cuda").bfloat16()
        bias = torch.randn(heads, 49, 49, device="cuda")
        scale = torch.rand(heads, device="cuda") * 5 + 5
        tf = timeit(lambda: ops.winattn_fwd(qkv, bias, scale, B, h, h, heads))
        tb = timeit(lambda: ops.winattn_bwd(qkv, do, bias, scale, B, h, h, heads))
        pairs = rows // 49 * heads
        gf = pairs * 2 * 2 * 49 * 49 * 32 / 1e9
        print("h=%2d c=%3d heads=%2d pairs=%6d | fwd %7.1fus %5.1fTF | bwd %7.1fus %5.1fTF" % (h, c, heads, pairs, tf, gf / tf / 1e3, tb, 2.5 * gf / tb / 1e3), flush=True)
    sys.exit(0)
for (h, c, k, r, stride) in SHAPES:
    pad = (r - 1) // 2
    x = torch.randn(B, h, h, c, device="cuda").bfloat16()
    w = (torch.randn(k, r, r, c, device="cuda") * 0.05).bfloat16()
    ho = (h + 2 * pad - r) // stride + 1
    flops = 2.0 * B * ho * ho * k * r * r * c
    line = "h=%3d c=%3d k=%3d r=%d s=%d  GF=%6.1f |" % (h, c, k, r, stride, flops / 1e9)
    if what == "fwd":
        ref = None
        lib().frhip_set_conv_halo(0)
        for tile in (1, 2, 3, 4):
            if tile == 4 and k % 256:
                line += "    --    |"
                continue
            lib().frhip_set_nt_tile(tile)
            y, _ = ops.conv_fwd(x, w, stride, pad)
            if ref is None:
                ref = y.float()
            err = (y.float() - ref).abs().max().item()
            us = timeit(lambda: ops.conv_fwd(x, w, stride, pad))
            line += " t%d %6.1fus %5.0fTF e=%.0e |" % (tile, us, flops / us / 1e6, err)
        lib().frhip_set_nt_tile(0)
        for hm in (1, 2):
            lib().frhip_set_conv_halo(hm)
            y, _ = ops.conv_fwd(x, w, stride, pad)
            err = (y.float() - ref).abs().max().item()
            us = timeit(lambda: ops.conv_fwd(x, w, stride, pad))
            line += " halo%d %6.1fus %5.0fTF e=%.0e |" % (hm, us, flops / us / 1e6, err)
        lib().frhip_set_conv_halo(1)
    elif what == "wgrad":
        dy = torch.randn(B, ho, ho, k, device="cuda").bfloat16()
        dw = torch.zeros(k, r, r, c, device="cuda")
        for mode in (0, 1):      # 0 = per-tap gather kernel, 1 = nine-tap kernel where it applies
            lib().frhip_set_wgrad_taps9(mode)
            us = timeit(lambda: ops.conv_wgrad(dy, x, dw, r, r, stride, pad, 0))
            line += " wgrad[m%d] %6.1fus %5.0fTF |" % (mode, us, flops / us / 1e6)
        lib().frhip_set_wgrad_taps9(1)
    print(line, flush=True)
