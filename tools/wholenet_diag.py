#!/usr/bin/env python3
"""GPU diagnostic: per-tensor gradient error of the fp32 validation mode against the reference's whole-network training fixtures,
in network order (error that grows smoothly from the tail to the stem = summation-order noise amplified by the BatchNorm chain; a jump at
one layer = a defect there).  Usage (on the GPU box): python tools/wholenet_diag.py [Swin34|AlterNet50|ResNet18] [fp32|bf16]"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "face-recognition-pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from oracle import alternet_ref, recipe, resnet_ref, swin_ref  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "AlterNet50"
dtype = sys.argv[2] if len(sys.argv) > 2 else "fp32"
if name == "AlterNet50":
    import nets.AlterNet_SwinV2_FAN as M
    g = dict(np.load(os.path.join(ROOT, "tests/golden/alternet50_b8_train.npz")))
    spec, fs, hw = alternet_ref.alter_spec(name), alternet_ref.fill_special, 192
elif name == "Swin34":
    import nets.SwinV2 as M
    g = dict(np.load(os.path.join(ROOT, "tests/golden/swin34_b8_train.npz")))
    spec, fs, hw = swin_ref.swin_spec(name), swin_ref.fill_special, 112
else:
    import nets.resnet as M
    g = dict(np.load(os.path.join(ROOT, "tests/golden/resnet18_b4_train.npz")))
    g.update(seed=4242, batch=4)
    spec, fs, hw = resnet_ref.resnet_spec(resnet_ref.BLOCKS[name]), (lambda sd, spec: sd), 112
seed = int(g["seed"])
net = M.Encoder(types.SimpleNamespace(network=name, emd_size=512, img_size=hw, frhip_dtype=dtype))
net.load_state_dict(fs(recipe.fill_state(spec, seed), spec), strict=True)
net = net.cuda().train()
if hasattr(net, "dropout"):
    net.dropout.p = 0.0
for m in net.modules():
    if hasattr(m, "drop_path_rate"):
        m.drop_path_rate = 0.0
y = net(recipe.images(seed + 1, int(g["batch"]), hw, hw).cuda())
y.backward(recipe.normal(seed + 2, tuple(y.shape), 0.05).cuda())
out = y.detach().float().cpu().numpy()
print("embeddings: max |d| / max |ref| = %.3e" % (np.abs(out - g["out"]).max() / np.abs(g["out"]).max()))
print("%-44s %10s %10s %10s" % ("parameter", "max|d|/rms", "l2 rel", "rms"))
for k, p in net.named_parameters():
    want = g["gprobe." + k]
    got = recipe.probe(p.grad.float().cpu())
    rms = want[1] / p.numel() ** 0.5
    if rms < 1e-7:
        continue
    print("%-44s %10.2e %10.2e %10.2e" % (k, np.abs(got[2:] - want[2:]).max() / rms, abs(got[1] - want[1]) / want[1], rms))
