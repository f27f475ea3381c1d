"""Run-to-run spread of the fp32 training step, bounded.

Two fresh processes run the same two SGD steps (ResNet18, B = 8, fp32 mode).  What differs between them is the order of the fp32
atomics some weight-gradient kernels accumulate with: 3e-5 ... 6e-5 of a gradient tensor's largest element -- and, in one or two runs
out of eight, an alternative outcome 5.7e-4 away in the early layers (always the same one: a ReLU / arg-max decision that rounding
tips; `RUNS=8 python tools/noise_probe.py`, the same with the builtin and with the inline-assembly operand loads).  A synchronisation
error in a kernel (a stage read while it is being refilled: csrc/common.h glds16_asm and the `lgkmcnt(0)` in front of the refill
barriers of csrc/igemm_tn.hip) would show as run-dependent errors of the size of the operands' products, not of their rounding."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

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
enc = model.encoder.module if hasattr(model.encoder, "module") else model.encoder
enc.load_state_dict(recipe.fill_state(resnet_ref.resnet_spec(resnet_ref.BLOCKS["ResNet18"]), 777), strict=True)
img, ids = recipe.images(779, B), recipe.labels(780, B, C)
for st in range(2):
    model.training_step((img, ids.clone()))
keys = ("conv1.weight", "bn1.weight", "layer1.0.conv1.weight", "layer2.0.conv2.weight", "layer3.1.conv2.weight", "fc.weight")
np.savez(sys.argv[1], **{k: p.grad.float().cpu().numpy() for k, p in enc.named_parameters() if k in keys})
dist.destroy_process_group()
''' % (ROOT, ROOT)


def test_two_identical_fp32_runs_agree_to_rounding_noise():
    with tempfile.TemporaryDirectory() as td:
        runs = []
        for i in range(2):
            f = os.path.join(td, "r%d.npz" % i)
            subprocess.check_call([sys.executable, "-c", CHILD, f], env=dict(os.environ))
            runs.append(dict(np.load(f)))
        assert set(runs[0]) == set(runs[1]) and len(runs[0]) == 6
        for k in runs[0]:
            a, b = runs[0][k].astype(np.float64), runs[1][k].astype(np.float64)
            assert np.isfinite(a).all() and np.abs(a).max() > 0
            assert np.abs(a - b).max() <= 2e-3 * np.abs(a).max(), (k, float(np.abs(a - b).max()), float(np.abs(a).max()))
