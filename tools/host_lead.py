#!/usr/bin/env python3
"""How far ahead of the GPU is the host?  Enqueue time of ONE training step into an idle stream (no back-pressure) against the step's GPU
time, for the headline configuration.  GPU box only.  usage: python tools/host_lead.py [network] [batch]"""
import os
import sys
import tempfile
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "face-recognition-pytorch_amd")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import torch.distributed as dist

net = sys.argv[1] if len(sys.argv) > 1 else "ResNet50"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 512
img_size = 192 if net.startswith("AlterNet") else 112
dist.init_process_group("gloo", init_method="file://" + os.path.join(tempfile.mkdtemp(), "pg"), rank=0, world_size=1)
from model.FR_PartialFC import Model
conf = types.SimpleNamespace(network=net, emd_size=512, img_size=img_size, local_rank=0, world_size=1, sample_rate=1.0, mixed_precision=True,
                             loss_s=30.0, loss_m=0.35, n_classes=122000, optimizer="SGD", lr=0.1, wd=5e-4, mom=0.9, loss="PartialFC",
                             lr_scheduler=None, frhip_dtype="bf16", ckpt_path=None)
model = Model(conf, None, "train")
model.sync_loss = False
g = torch.Generator().manual_seed(1)
img = torch.randn((batch, 3, img_size, img_size), generator=g).clamp_(-1, 1).cuda()
ids = torch.randint(0, 122000, (batch,), generator=g).cuda()
for _ in range(5):
    model.training_step((img, ids.clone()))
torch.cuda.synchronize()
host, total = [], []
for _ in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.training_step((img, ids.clone()))
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3)
    total.append((t2 - t0) * 1e3)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    model.training_step((img, ids.clone()))
torch.cuda.synchronize()
steady = (time.perf_counter() - t0) / 20 * 1e3
print("%s B=%d: host enqueue of one step into an idle stream %.2f ms (min %.2f), that step start-to-done %.2f ms; steady state %.2f ms/step"
      % (net, batch, sorted(host)[len(host) // 2], min(host), sorted(total)[len(total) // 2], steady))
