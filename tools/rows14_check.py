"""rows kernel (14 x 14 nine-tap weight gradient) against the pixel-stream kernel and torch: results and time.  GPU box only."""
import ctypes
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

for (n, c, k) in [(1, 64, 64), (3, 64, 128), (8, 256, 256), (37, 128, 64), (512, 256, 256), (512, 256, 512)]:
    g = torch.Generator().manual_seed(n * 1000 + c)
    x = torch.randn(n, 14, 14, c, generator=g).bfloat16().cuda()
    dy = torch.randn(n, 14, 14, k, generator=g).bfloat16().cuda()
    dw = torch.zeros(k, 3, 3, c, device="cuda")
    ops.conv_wgrad(dy, x, dw, 3, 3, 1, 1)
    ref = torch.nn.grad.conv2d_weight(x.float().permute(0, 3, 1, 2), (k, c, 3, 3), dy.float().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    err = (dw - ref).abs().max().item() / ref.abs().max().item()
    t = timeit(lambda: ops.conv_wgrad(dy, x, dw, 3, 3, 1, 1)) if n >= 512 else 0.0
    print("n=%3d c=%3d k=%3d  max err / max |ref| = %.2e   %.1f us  (FRHIP_T9_ROWS=%s)" % (n, c, k, err, t, os.environ.get("FRHIP_T9_ROWS", "1")), flush=True)
    assert err < 2e-3
