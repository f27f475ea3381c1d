"""Host time per eager training step: at a small batch the GPU finishes every kernel long before the next launch arrives, so
wall time / steps = what the Python + HIP launch path costs per step (558 launches for ResNet50).  GPU box only.
usage: python tools/host_time.py [batch] [network]"""
import os
import sys
import tempfile
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "face-recognition-pytorch_amd")]
import torch
import torch.distributed as dist

dist.init_process_group("gloo", init_method="file://" + os.path.join(tempfile.mkdtemp(), "pg"), rank=0, world_size=1)
from model.FR_PartialFC import Model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
net = sys.argv[2] if len(sys.argv) > 2 else "ResNet50"
size = 192 if net.startswith("AlterNet") else 112
conf = types.SimpleNamespace(network=net, emd_size=512, img_size=size, local_rank=0, world_size=1, sample_rate=1.0,
                             mixed_precision=True, loss_s=30.0, loss_m=0.35, n_classes=122000, optimizer="SGD", lr=0.1, wd=5e-4,
                             mom=0.9, loss="PartialFC", lr_scheduler=None, frhip_dtype="bf16", ckpt_path=None)
model = Model(conf, None, "train")
model.sync_loss = False
img = torch.randn(B, 3, size, size).clamp_(-1, 1).cuda()
ids = torch.randint(0, 122000, (B,)).cuda()
for _ in range(5):
    model.training_step((img, ids.clone()))
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20):
        model.training_step((img, ids.clone()))
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s B=%d: host %.2f ms per step to enqueue, %.2f ms per step incl. the final drain" % (net, B, (t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3), flush=True)
