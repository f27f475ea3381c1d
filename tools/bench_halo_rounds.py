"""Halo-kernel tile comparison on shapes that fill WHOLE rounds of resident workgroups (no tail quantisation).  GPU box only.
   usage: python tools/bench_halo_rounds.py [fwd|dgrad]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "face-recognition-pytorch_amd")]
import torch
from frhip import ops
from frhip._abi import lib


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


what = sys.argv[1] if len(sys.argv) > 1 else "fwd"
# (batch, h, c): M = batch * h * h is a multiple of 512 * 256 rows
for (b, h, c) in [(2048, 16, 128), (1024, 16, 256), (4096, 8, 512)]:
    x = torch.randn(b, h, h, c, device="cuda").bfloat16()
    w = (torch.randn(c, 3, 3, c, device="cuda") * 0.05).bfloat16()
    flops = 2.0 * b * h * h * c * 9 * c
    line = "B=%4d h=%2d c=%3d GF=%6.1f |" % (b, h, c, flops / 1e9)
    for name, mode, dirs in (("wide", 1, 3), ("4-wave", 2, 0), ("8-wave", 3, 0)):
        lib().frhip_set_conv_halo(mode)
        lib().frhip_set_halo_wide_dirs(dirs)
        if what == "fwd":
            us = timeit(lambda: ops.conv_fwd(x, w, 1, 1))
        else:
            us = timeit(lambda: ops.conv_dgrad(x, w, (b, h, h, c), 3, 3, 1, 1))
        line += " %s %6.1fus %5.0fTF |" % (name, us, flops / us / 1e6)
    lib().frhip_set_conv_halo(1)
    lib().frhip_set_halo_wide_dirs(2)
    print(line, flush=True)
