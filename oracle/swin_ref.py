"""CPU restatement of the SwinV2-style backbone (TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows /root/reference/nets/SwinV2.py:
  * WindowAttention.forward (:139-179): qkv = x W^T + [q_bias, 0, v_bias]; cosine attention
    normalize(q) normalize(k)^T * exp(min(logit_scale, ln 100)); + 16*sigmoid(cpb_mlp(coords_table))[index];
    softmax; @ v; proj.  Tables (:94-125): log-spaced relative coords in [-1,1]*8 -> sign*log2(|.|+1)/log2(8),
    pair index (dy + 6) * 13 + (dx + 6) for 7x7 windows.
  * SwinTransformerBlock.forward (:263-300), shift_size = 0: NCHW in/out, x = x + BN(attn(windows(x)));
    x = x + BN(fc2(gelu(fc1(x)))) with 1x1-conv MLP (bias=True), hidden = 4*dim.
  * Swin (:487-565): ResNet stem, stages = optional Conv2d(k=2,s=2,bias=False) + N blocks, tail
    bn2 -> Dropout(0.5) -> AdaptiveAvgPool(7,7) -> flatten -> fc -> bn3.  num_blocks / heads per :570-643.
Functional over a flat state dict with the reference's key names; windows are gathered by index arithmetic.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import resnet_ref

SWIN = {  # name -> (num_blocks, heads)
    "Swin18": ((0, 1, 1, 1), (2, 4, 8, 16)),
    "Swin34": ((0, 0, 4, 6), (2, 4, 8, 16)),
    "Swin50": ((0, 0, 4, 10), (2, 4, 8, 16)),
    "Swin100": ((0, 0, 6, 14), (2, 4, 8, 16)),
    "Swin200": ((0, 0, 6, 30), (2, 4, 8, 16)),
}
WS = 7


def coords_table(ws=WS):
    r = torch.arange(-(ws - 1), ws, dtype=torch.float32) / (ws - 1) * 8
    t = torch.stack(torch.meshgrid([r, r], indexing="ij")).permute(1, 2, 0).contiguous().unsqueeze(0)
    return torch.sign(t) * torch.log2(torch.abs(t) + 1.0) / np.log2(8)


def position_index(ws=WS):
    c = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij")).flatten(1)
    rel = (c[:, :, None] - c[:, None, :]).permute(1, 2, 0) + (ws - 1)
    return rel[:, :, 0] * (2 * ws - 1) + rel[:, :, 1]


def swin_plan(name, emd_size=512):
    """[(layer_idx(1-based), index in Sequential, kind 'down'|'block', cin, cout, heads)]"""
    nblocks, heads = SWIN[name]
    plan, inplanes = [], 64
    for li, (planes, nb, hd, stride) in enumerate(zip((64, 128, 256, emd_size), nblocks, heads, (1, 2, 2, 2)), start=1):
        idx = 0
        if stride > 1:
            plan.append((li, idx, "down", inplanes, planes, hd))
            idx += 1
        inplanes = planes
        for _ in range(nb):
            plan.append((li, idx, "block", planes, planes, hd))
            idx += 1
    return plan


def _attn_spec(p, c, heads):
    return [(p + ".logit_scale", (heads, 1, 1), "logit_scale"), (p + ".q_bias", (c,), "linear_b"), (p + ".v_bias", (c,), "linear_b"),
            (p + ".relative_coords_table", (1, 13, 13, 2), "coords"), (p + ".relative_position_index", (49, 49), "posidx"),
            (p + ".cpb_mlp.0.weight", (512, 2), "linear_w"), (p + ".cpb_mlp.0.bias", (512,), "linear_b"),
            (p + ".cpb_mlp.2.weight", (heads, 512), "linear_w"), (p + ".qkv.weight", (3 * c, c), "linear_w"),
            (p + ".proj.weight", (c, c), "linear_w"), (p + ".proj.bias", (c,), "linear_b")]


def block_spec(p, c, heads):
    return (_attn_spec(p + ".attn", c, heads) + resnet_ref._bn_spec(p + ".norm2", c) +
            [(p + ".mlp.fc1.weight", (4 * c, c, 1, 1), "conv"), (p + ".mlp.fc1.bias", (4 * c,), "linear_b"),
             (p + ".mlp.fc2.weight", (c, 4 * c, 1, 1), "conv"), (p + ".mlp.fc2.bias", (c,), "linear_b")] +
            resnet_ref._bn_spec(p + ".norm3", c))


def swin_spec(name, emd_size=512):
    spec = [("conv1.weight", (64, 3, 3, 3), "conv")] + resnet_ref._bn_spec("bn1", 64)
    for li, idx, kind, cin, cout, hd in swin_plan(name, emd_size):
        p = "layer%d.%d" % (li, idx)
        if kind == "down":
            spec.append((p + ".weight", (cout, cin, 2, 2), "conv"))
        else:
            spec += block_spec(p, cout, hd)
    spec += resnet_ref._bn_spec("bn2", emd_size)
    spec += [("fc.weight", (emd_size, emd_size * 49), "linear_w"), ("fc.bias", (emd_size,), "linear_b")]
    spec += resnet_ref._bn_spec("bn3", emd_size)
    return spec


def fill_special(sd, spec):
    """tables and logit_scale are not random: reference values (nets/SwinV2.py:88-125)"""
    for k, shape, kind in spec:
        if kind == "coords":
            sd[k] = coords_table()
        elif kind == "posidx":
            sd[k] = position_index()
        elif kind == "logit_scale":
            sd[k] = torch.log(10 * torch.ones(shape)) + 0.05 * torch.arange(shape[0], dtype=torch.float32).view(shape)
    return sd


def relative_bias(sd, p, heads):
    t = F.linear(F.relu(F.linear(sd[p + ".relative_coords_table"], sd[p + ".cpb_mlp.0.weight"], sd[p + ".cpb_mlp.0.bias"])),
                 sd[p + ".cpb_mlp.2.weight"]).view(-1, heads)
    n = sd[p + ".relative_position_index"].shape[0]
    b = t[sd[p + ".relative_position_index"].view(-1)].view(n, n, heads).permute(2, 0, 1)
    return 16 * torch.sigmoid(b)            # [heads, n, n]


def window_attention(sd, p, xw, heads, mask=None):
    """xw [B_, n, C] -> [B_, n, C]; mask [nW, n, n] (0 / -100) for shifted windows"""
    b_, n, c = xw.shape
    bias = torch.cat([sd[p + ".q_bias"], torch.zeros_like(sd[p + ".v_bias"]), sd[p + ".v_bias"]])
    qkv = F.linear(xw, sd[p + ".qkv.weight"], bias).reshape(b_, n, 3, heads, c // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = F.normalize(q, dim=-1) @ F.normalize(k, dim=-1).transpose(-2, -1)
    scale = torch.clamp(sd[p + ".logit_scale"], max=math.log(100.0)).exp()
    attn = attn * scale + relative_bias(sd, p, heads).unsqueeze(0)
    if mask is not None:
        nw = mask.shape[0]
        attn = (attn.view(b_ // nw, nw, heads, n, n) + mask.unsqueeze(1).unsqueeze(0)).view(-1, heads, n, n)
    attn = torch.softmax(attn, dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(b_, n, c)
    return F.linear(out, sd[p + ".proj.weight"], sd[p + ".proj.bias"])


def to_windows(x_nhwc, ws=WS):
    b, h, w, c = x_nhwc.shape
    return x_nhwc.reshape(b, h // ws, ws, w // ws, ws, c).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, c)


def from_windows(xw, b, h, w, ws=WS):
    c = xw.shape[-1]
    return xw.view(b, h // ws, w // ws, ws, ws, c).permute(0, 1, 3, 2, 4, 5).reshape(b, h, w, c)


def swin_block(sd, p, x, heads, training):
    """x NCHW -> NCHW"""
    b, c, h, w = x.shape
    a = window_attention(sd, p + ".attn", to_windows(x.permute(0, 2, 3, 1)), heads)
    a = from_windows(a, b, h, w).permute(0, 3, 1, 2)
    x = x + resnet_ref._bn(sd, p + ".norm2", a, training)
    m = F.conv2d(x, sd[p + ".mlp.fc1.weight"], sd[p + ".mlp.fc1.bias"])
    m = F.conv2d(F.gelu(m), sd[p + ".mlp.fc2.weight"], sd[p + ".mlp.fc2.bias"])
    return x + resnet_ref._bn(sd, p + ".norm3", m, training)


def swin_forward(sd, x, name, training, emd_size=512, dropout_mask=None):
    """dropout_mask: None = no dropout (eval, or the p=0 training fixtures); else a {0, 2}-valued tensor."""
    y = F.conv2d(x, sd["conv1.weight"], None, 1, 1)
    y = F.relu(resnet_ref._bn(sd, "bn1", y, training))
    y = F.max_pool2d(y, 3, 2, 1)
    for li, idx, kind, cin, cout, hd in swin_plan(name, emd_size):
        p = "layer%d.%d" % (li, idx)
        y = F.conv2d(y, sd[p + ".weight"], None, 2, 0) if kind == "down" else swin_block(sd, p, y, hd, training)
    y = resnet_ref._bn(sd, "bn2", y, training)
    if dropout_mask is not None:
        y = y * dropout_mask
    y = F.adaptive_avg_pool2d(y, (7, 7)).reshape(y.shape[0], -1)
    y = F.linear(y, sd["fc.weight"], sd["fc.bias"])
    return resnet_ref._bn(sd, "bn3", y, training)
