"""cProfile of eager training steps (host-side launch overhead).  GPU box only."""
import cProfile
import os
import pstats
import sys
import tempfile
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "face-recognition-pytorch_amd")]
import torch
import torch.distributed as dist

dist.init_process_group("gloo", init_method="file://" + os.path.join(tempfile.mkdtemp(), "pg"), rank=0, world_size=1)
from model.FR_PartialFC import Model

RATE = float(os.environ.get("RATE", "1.0"))      # RATE=0.1 FRHIP_FORCE_COLLECTIVES=1: the multi-GPU code path
conf = types.SimpleNamespace(network="ResNet50", emd_size=512, img_size=112, local_rank=0, world_size=1,
                             force_ddp=RATE < 1.0, sample_rate=RATE, mixed_precision=True, loss_s=30.0, loss_m=0.35, n_classes=122000,
                             optimizer="SGD", lr=0.1, wd=5e-4, mom=0.9, loss="PartialFC", lr_scheduler=None,
                             frhip_dtype="bf16", ckpt_path=None)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
model = Model(conf, None, "train")
model.sync_loss = False
img = torch.randn(B, 3, 112, 112).clamp_(-1, 1).cuda()
ids = torch.randint(0, 122000, (B,)).cuda()
for _ in range(3):
    model.training_step((img, ids.clone()))
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    model.training_step((img, ids.clone()))
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumtime").print_stats(45)
