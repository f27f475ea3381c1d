"""What does a kernel-to-kernel transition cost on the stream?  Times N launches of A, N of B and N alternations A B A B ... between two
HIP events (everything enqueued ahead, the GPU is never starved): per-pair cost of the alternation minus (A + B) = what the two
transitions cost beyond same-kernel back-to-back launches.  GPU box only.   usage: python tools/gap_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "face-recognition-pytorch_amd")]
import torch
from frhip import ops

B, N = 512, 40


def timed(fns, n):
    for f in fns:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        for f in fns:
            f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


x = torch.randn(B, 14, 14, 256, device="cuda").bfloat16()
w = (torch.randn(256, 3, 3, 256, device="cuda") * 0.05).bfloat16()
y = torch.empty_like(x)
gamma, beta = torch.ones(256, device="cuda"), torch.zeros(256, device="cuda")
rows = B * 14 * 14
st = ops.bn_finalize(ops.colstats(x.view(rows, 256)), rows, gamma, beta, None, None)
part = ops.colstats(x.view(rows, 256))

conv = lambda: ops.conv_fwd(x, w, 1, 1)
apply_ = lambda: ops.bn_apply(x, st, relu=True, out=y)
fin = lambda: ops.bn_finalize(part, rows, gamma, beta, None, None)
stats = lambda: ops.colstats(x.view(rows, 256))

for name, a, b in (("conv / bn_apply", conv, apply_), ("conv / bn_finalize", conv, fin), ("bn_apply / colstats", apply_, stats),
                   ("bn_apply / bn_finalize", apply_, fin)):
    ta, tb, tab = timed([a], N), timed([b], N), timed([a, b], N)
    print("%-24s A %7.1f us  B %7.1f us  A+B alternating %7.1f us  -> transitions cost %+6.1f us per pair" % (name, ta, tb, tab, tab - ta - tb), flush=True)
