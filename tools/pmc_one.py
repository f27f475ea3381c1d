"""One kernel under the counters: python tools/pmc_one.py wgrad|fwd|dgrad h c k  (B=512, bf16).  Run under
rocprofv3 --pmc ... --kernel-trace on the GPU box."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "face-recognition-pytorch_amd")]
import torch
from frhip import ops

what, h, c, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
B = int(os.environ.get("B", "512"))
x = torch.randn(B, h, h, c, device="cuda").bfloat16()
w = (torch.randn(k, 3, 3, c, device="cuda") * 0.05).bfloat16()
dy = torch.randn(B, h, h, k, device="cuda").bfloat16()
dw = torch.zeros(k, 3, 3, c, device="cuda")
wt = ops.pack_wt(w.float(), torch.bfloat16)
for _ in range(4):
    if what == "wgrad":
        ops.conv_wgrad(dy, x, dw, 3, 3, 1, 1, 0)
    elif what == "fwd":
        ops.conv_fwd(x, w, 1, 1)
    else:
        ops.conv_dgrad(dy, wt, (B, h, h, c), 3, 3, 1, 1)
torch.cuda.synchronize()
