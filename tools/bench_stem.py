"""time the recompute-style stem kernels (B=512, 112x112) -- GPU box only"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "face-recognition-pytorch_amd")]
import torch
from frhip import ops
B = int(os.environ.get("B", "512"))
x = torch.randn(B, 3, 112, 112, device="cuda").clamp(-1, 1)
w = torch.randn(64, 27, device="cuda") * 0.2
wp = ops.pack_stem(w, torch.bfloat16, kp=32)
gamma, beta = torch.ones(64, device="cuda"), torch.zeros(64, device="cuda")
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
st = ops.bn_finalize(ops.stem_stats(x, wp), B * 112 * 112, gamma, beta, None, None)
pooled, arg = ops.stem_fwd(x, wp, st)
dp = torch.randn_like(pooled)
dg, db, dw = torch.zeros(64, device="cuda"), torch.zeros(64, device="cuda"), torch.zeros(64, 27, device="cuda")
print("stats %.0f us | fwd %.0f us | bwd (reduce + finalize + wgrad) %.0f us" % (
    t(lambda: ops.stem_stats(x, wp)), t(lambda: ops.stem_fwd(x, wp, st)),
    t(lambda: ops.stem_bwd(x, wp, dp, arg, st, gamma, dg, db, dw))), flush=True)
