"""class-centre gradient of the head at the cfg 2 shape: fused launch (frhip_head_dw) against GEMM + normalise-backward.  GPU box only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "face-recognition-pytorch_amd")]
import torch
from frhip import ops


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for n, classes in ((512, 122000), (512, 12200), (4096, 1525)):
    dt = (torch.randn(n, (classes + 7) // 8 * 8, device="cuda") * 0.01).bfloat16()          # pitch in whole 16-byte vectors, as frhip_head_bwd_dt writes it
    e = torch.randn(n, 512, device="cuda")
    ehat = (e / e.norm(dim=1, keepdim=True)).bfloat16()
    what, wnorm = ops.l2norm_rows(torch.randn(classes, 512, device="cuda") * 0.05, torch.bfloat16)
    g = torch.empty((classes, 512), dtype=torch.float32, device="cuda")

    def two():
        ops.gemm_tn(dt, ehat, g, kc=classes, overwrite=True)
        return ops.l2norm_bwd(g, what, wnorm)

    print("n=%d classes=%d: fused %.1f us, two-pass %.1f us" % (n, classes, timeit(lambda: ops.head_dw(dt, ehat, what, wnorm)), timeit(two)), flush=True)
