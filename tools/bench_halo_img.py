"""Launch-fitted halo tile (csrc/igemm_halo_img.h) against the 4-wave and the 64 x 128-per-wave tiles: bit-identity of the outputs
on small ragged cases, then timing on the ResNet50 body shapes at B = 512.  GPU box only.
   usage: python tools/bench_halo_img.py [check|time|all]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "face-recognition-pytorch_amd")]
import torch
from frhip import ops
from frhip._abi import lib

what = sys.argv[1] if len(sys.argv) > 1 else "all"


def timeit(fn, n=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def run(n, h, c, k, img, wide_mode):
    lib().frhip_set_halo_img(img)
    old = lib().frhip_set_conv_halo(wide_mode)
    g = torch.Generator().manual_seed(5)
    x = torch.randn((n, h, h, c), generator=g).bfloat16().cuda()
    w = (torch.randn((k, 3, 3, c), generator=g) * 0.05).bfloat16().cuda()
    dy = torch.randn((n, h, h, k), generator=g).bfloat16().cuda()
    res = torch.randn((n, h, h, c), generator=g).bfloat16().cuda()
    y_bn = torch.randn((n, h, h, c), generator=g).bfloat16().cuda()
    wt = ops.pack_wt(w.float(), torch.bfloat16)
    rows = n * h * h
    st = ops.bn_finalize(ops.colstats(y_bn.view(rows, c)), rows, torch.ones(c).cuda(), torch.zeros(c).cuda(), None, None)
    y, part = ops.conv_fwd(x, w, 1, 1)
    dx, bpart = ops.conv_dgrad(dy, wt, (n, h, h, c), 3, 3, 1, 1, residual=res, bnred=(y_bn, st, True))
    dx2 = ops.conv_dgrad(dy, wt, (n, h, h, c), 3, 3, 1, 1)
    torch.cuda.synchronize()
    lib().frhip_set_conv_halo(old)
    return y, part, dx, bpart, dx2


if what in ("check", "all"):
    for case in [(6, 28, 128, 128), (7, 14, 256, 256), (11, 7, 512, 512), (9, 14, 128, 256), (300, 7, 256, 128), (3, 20, 128, 384),
                 (16, 14, 256, 256), (64, 14, 256, 256), (1, 7, 128, 128)]:
        a = run(*case, 3, 1)
        b = run(*case, 0, 2 | 64)
        ok = torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]) and torch.equal(a[4], b[4])
        ps = torch.allclose(a[1].sum(0), b[1].sum(0), rtol=1e-4, atol=1e-2) and torch.allclose(a[3].sum(0), b[3].sum(0), rtol=1e-4, atol=1e-2)
        print("case %s: outputs bit-identical %s, partial sums agree %s (rows %d vs %d), max|dy| %.3g" % (
            case, ok, ps, a[1].shape[0], b[1].shape[0], (a[0].float() - b[0].float()).abs().max().item()), flush=True)
        if not ok:
            bad = (a[0] != b[0]).nonzero()
            print("   first mismatches (n,y,x,k):", bad[:8].tolist(), "count", bad.shape[0])

if what in ("time", "all"):
    B = int(os.environ.get("B", "512"))
    for (h, c) in [(28, 128), (14, 256), (7, 512)]:
        g = torch.Generator().manual_seed(5)
        x = torch.randn((B, h, h, c), generator=g).bfloat16().cuda()
        w = (torch.randn((c, 3, 3, c), generator=g) * 0.05).bfloat16().cuda()
        wt = ops.pack_wt(w.float(), torch.bfloat16)
        res = torch.randn((B, h, h, c), generator=g).bfloat16().cuda()
        rows = B * h * h
        st = ops.bn_finalize(ops.colstats(res.view(rows, c)), rows, torch.ones(c).cuda(), torch.zeros(c).cuda(), None, None)
        flops = 2.0 * B * h * h * c * 9 * c
        line = "h=%2d c=%3d GF=%6.1f |" % (h, c, flops / 1e9)
        for name, img, mode, wide_dirs in (("fitted", 3, 1, 3), ("wide", 0, 1, 3), ("4-wave", 0, 2 | 64, 3)):
            lib().frhip_set_halo_img(img)
            oldp = lib().frhip_set_halo_wide_slots(0 | (wide_dirs << 18) | (1 << 21))
            old = lib().frhip_set_conv_halo(mode)
            tf = timeit(lambda: ops.conv_fwd(x, w, 1, 1))
            td = timeit(lambda: ops.conv_dgrad(x, wt, (B, h, h, c), 3, 3, 1, 1, residual=res, bnred=(res, st, True)))
            lib().frhip_set_conv_halo(old)
            lib().frhip_set_halo_wide_slots(oldp)
            line += " %s fwd %6.1fus %5.0fTF dgrad+red %6.1fus %5.0fTF |" % (name, tf, flops / tf / 1e6, td, flops / td / 1e6)
        lib().frhip_set_halo_img(3)
        print(line, flush=True)
        if os.environ.get("ZEROS"):
            # same launches on all-zero operands: the matrix pipes draw less power, the chip holds a higher clock -- tells stalls from DVFS
            xz, wz = torch.zeros_like(x), torch.zeros_like(w)
            line = "   zero operands    |"
            for name, img, mode in (("fitted", 3, 1), ("4-wave", 0, 2 | 64)):
                lib().frhip_set_halo_img(img)
                old = lib().frhip_set_conv_halo(mode)
                tf = timeit(lambda: ops.conv_fwd(xz, wz, 1, 1))
                lib().frhip_set_conv_halo(old)
                line += " %s fwd %6.1fus %5.0fTF |" % (name, tf, flops / tf / 1e6)
            lib().frhip_set_halo_img(3)
            print(line, flush=True)
