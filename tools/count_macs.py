"""Forward MACs per image of AlterNet50 @192 (convolutions, linears, window-attention matmuls) counted with forward hooks on the
REFERENCE modules (build container only; same import shims as tools/make_golden.py).  bench.py's flop_img for AlterNet50 = 6 x this.
   conv 1.9411 G + linear 0.1288 G + attention 0.0325 G = 2.1025 GMAC forward -> 12.6 GFLOP per image and training step."""
import sys, types, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
sys.dont_write_bytecode = True
import make_golden as mg
mg._swin_ref()
import nets.AlterNet_SwinV2_FAN as A
conf = types.SimpleNamespace(network="AlterNet50", emd_size=512, img_size=192)
net = A.AlterNet50(conf).eval()
macs = {"conv": 0, "linear": 0, "attn": 0}
def conv_hook(m, i, o):
    macs["conv"] += o.numel() // o.shape[0] * (m.in_channels // m.groups) * m.kernel_size[0] * m.kernel_size[1]
def lin_hook(m, i, o):
    macs["linear"] += o.numel() // i[0].shape[0] * m.in_features if o.dim() == 2 else o.numel() * m.in_features
for m in net.modules():
    if isinstance(m, torch.nn.Conv2d): m.register_forward_hook(conv_hook)
    if isinstance(m, torch.nn.Linear): m.register_forward_hook(lin_hook)
x = torch.randn(1, 3, 192, 192)
with torch.no_grad(): net(x)
print(macs)
import nets.SwinV2 as S
macs2 = {"attn": 0}
def attn_hook(m, i, o):
    x = i[0]
    macs2["attn"] += 2 * x.shape[0] * x.shape[1] * x.shape[1] * x.shape[2]
cls = [c for n, c in vars(A).items() if isinstance(c, type) and "WindowAttention" in n] + [S.WindowAttention]
for m in net.modules():
    if any(isinstance(m, c) for c in cls): m.register_forward_hook(attn_hook)
with torch.no_grad(): net(x)
print(macs2, cls)
