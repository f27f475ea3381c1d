"""Linear-layer GEMMs of the attention backbones per NT tile (frhip_set_nt_tile): us and effective GB/s of operand + result bytes.
python tools/bench_linear_tiles.py   (on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "face-recognition-pytorch_amd"))
import torch
from frhip import ops
from frhip._abi import lib


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


dt = torch.bfloat16
for m, k, n, name in [(100352, 256, 768, "qkv"), (100352, 256, 256, "proj"), (100352, 256, 1024, "fc1+gelu"), (100352, 1024, 256, "fc2"),
                      (25088, 512, 1536, "qkv4"), (25088, 512, 2048, "fc1+gelu4"), (25088, 2048, 512, "fc2_4")]:
    a = torch.randn(m, k, device="cuda").to(dt)
    w = (torch.randn(n, k, device="cuda") * 0.05).to(dt)
    bias = torch.randn(n, device="cuda")
    act = "gelu" in name
    byts = (m * k + n * k + m * n * (2 if act else 1)) * 2
    row = "%-10s M=%6d K=%4d N=%4d " % (name, m, k, n)
    for tile in (1, 3, 4):
        lib().frhip_set_nt_tile(tile)
        try:
            us = timeit(lambda: ops.linear_fwd(a, w, bias, want_act=act, want_stats=not act))
        finally:
            lib().frhip_set_nt_tile(0)
        row += "| tile %d %7.1f us %5.0f GB/s %5.0f TF " % (tile, us, byts / us / 1e3, 2.0 * m * n * k / us / 1e6)
    print(row)
