"""Swin MLP kernels at the stage shapes of Swin34 (B = 512): stored pre-activation against recompute (csrc/mlp_recompute.hip).  GPU box only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "face-recognition-pytorch_amd")]
import torch
from frhip import ops

def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for (hw, c) in [(56, 64), (28, 128), (14, 256), (7, 512)]:
    m, n = 512 * hw * hw, 4 * c
    x = torch.randn(m, c, device="cuda").bfloat16()
    w1 = (torch.randn(n, c, device="cuda") * 0.1).bfloat16()
    b1 = torch.randn(n, device="cuda") * 0.1
    dy = torch.randn(m, c, device="cuda").bfloat16()
    w2t = (torch.randn(n, c, device="cuda") * 0.1).bfloat16()
    hid, act, _ = ops.linear_fwd(x, w1, b1, want_act=True)
    a = t(lambda: ops.linear_fwd(x, w1, b1, want_act=True))
    b = t(lambda: ops.linear_fwd_act(x, w1, b1))
    c_ = t(lambda: ops.linear_dgrad_gelu(dy, w2t, hid))
    d = t(lambda: ops.linear_dgrad_gelu_rc(dy, w2t, x, w1, b1))
    gb = m * c * 2 / 1e9
    print("tokens %8d C %3d | fc1 fwd: both tensors %7.1f us (%.2f GB), act only %7.1f us (%.2f GB) | fc2 dgrad: stored %7.1f us (%.2f GB), recompute %7.1f us (%.2f GB)"
          % (m, c, a, gb * 9, b, gb * 5, c_, gb * 9, d, gb * 6), flush=True)
