import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "face-recognition-pytorch_amd")]
import torch
import bench
import types, tempfile
import torch.distributed as dist
dist.init_process_group("gloo", init_method="file://" + os.path.join(tempfile.mkdtemp(), "pg"), rank=0, world_size=1)
conf = bench.make_conf(types.SimpleNamespace(network="ResNet18", classes=2000), 0, 1)
from model.FR_PartialFC import Model
from frhip import optim
m = Model(conf, "train")
img = torch.randn(16, 3, 112, 112).cuda(); ids = torch.randint(0, 2000, (16,)).cuda()
m.opt.zero_grad()
m.encoder.train()
loss = m.loss(torch.nn.functional.normalize(m.encoder(img)), ids, m.opt)
loss.backward()
print(type(m.opt), m.opt._fusable())
for gi, g in enumerate(m.opt.param_groups):
    print(gi, {k: v for k, v in g.items() if k != "params"})
    for p in g["params"]:
        if p.grad is None:
            print("  grad None", tuple(p.shape)); continue
        if not optim._dense_same_layout(p.data, p.grad):
            print("  layout", tuple(p.shape), p.stride(), p.grad.stride())
        if p.grad.dtype != torch.float32 or p.dtype != torch.float32:
            print("  dtype", p.dtype, p.grad.dtype)
