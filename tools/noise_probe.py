"""run-to-run spread of the fp32 training step's gradients (ResNet18, B = 8, two steps, the set-up of tests/test_nccl_gpu.py without a
process group): prints max |difference| / max |value| between two identical runs in fresh processes.  GPU box only."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, types
import numpy as np, torch
sys.path[:0] = [%r, os.path.join(%r, "face-recognition-pytorch_amd")]
import torch.distributed as dist
dist.init_process_group("gloo", init_method="file://" + sys.argv[1] + ".pg", rank=0, world_size=1)
from model.FR_PartialFC import Model
from oracle import recipe, resnet_ref
C, B = 256, 8
conf = types.SimpleNamespace(network="ResNet18", emd_size=512, img_size=112, local_rank=0, world_size=1, sample_rate=1.0,
                             mixed_precision=False, loss_s=30.0, loss_m=0.35, n_classes=C, optimizer="SGD", lr=0.1, wd=5e-4, mom=0.9,
                             loss="PartialFC", lr_scheduler=None, frhip_dtype="fp32", ckpt_path=None)
torch.manual_seed(5)
model = Model(conf, None, "train")
sd = recipe.fill_state(resnet_ref.resnet_spec(resnet_ref.BLOCKS["ResNet18"]), 777)
enc = model.encoder.module if hasattr(model.encoder, "module") else model.encoder
enc.load_state_dict(sd, strict=True)
img, ids = recipe.images(779, B), recipe.labels(780, B, C)
for st in range(2):
    model.training_step((img, ids.clone()))
out = {k: p.grad.float().cpu().numpy() for k, p in enc.named_parameters() if k in ("conv1.weight", "layer1.0.conv1.weight", "layer3.1.conv2.weight", "fc.weight", "bn1.weight")}
np.savez(sys.argv[1], **out)
''' % (ROOT, ROOT)

N = int(os.environ.get("RUNS", "3"))
with tempfile.TemporaryDirectory() as td:
    files = []
    for i in range(N):
        f = os.path.join(td, "r%d.npz" % i)
        subprocess.check_call([sys.executable, "-c", CHILD, f], env=dict(os.environ))
        files.append(dict(np.load(f)))
    for k in files[0]:
        d = [np.abs(files[0][k] - files[j][k]).max() / np.abs(files[0][k]).max() for j in range(1, N)]
        print("%-24s max|g| %.3e   rel. distance of runs 1.. to run 0: %s   (FRHIP_LIB_PATH=%s)" % (
            k, np.abs(files[0][k]).max(), " ".join("%.1e" % v for v in d), os.environ.get("FRHIP_LIB_PATH", "-")))
