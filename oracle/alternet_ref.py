"""CPU restatement of the hybrid conv + window-attention backbone (TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows /root/reference/nets/AlterNet_SwinV2_FAN.py:
  * SwinTransformerBlock (:306-450): attention ONLY (no MLP), x = x + BN(attn(windows(roll(x)))), window sizes
    6/6/6/3, the second block of every pair is shifted by ws//2 with the -100 region mask (:375-397, :420-440);
    dim == dim_out everywhere in AlterNet50, so the 1x1 shortcut branch (:351-356) never exists.
  * AlterNet (:637-751): stem conv3x3 STRIDE 2 + BN + ReLU + MaxPool, four stages of IR BasicBlocks with attention
    pairs inserted by the rule of stack_layers (:685-731), tail bn2 -> ReLU -> Dropout -> AdaptiveAvgPool(6,6)
    -> fc(512*36 -> 512) -> bn3.  AlterNet50 = blocks [3,4,14,4], attention pairs [0,1,4,1], heads (2,4,8,16),
    only constructible at img_size 192.
DropPath(0.1) and Dropout are identity in eval; training fixtures set both to zero.
"""
import torch
import torch.nn.functional as F

from . import resnet_ref, swin_ref

ALTER = {"AlterNet50": ((3, 4, 14, 4), (0, 1, 4, 1), (2, 4, 8, 16))}
WINDOWS = (6, 6, 6, 3)


def coords_table(ws):
    import numpy as np
    r = torch.arange(-(ws - 1), ws, dtype=torch.float32) / (ws - 1) * 8
    t = torch.stack(torch.meshgrid([r, r], indexing="ij")).permute(1, 2, 0).contiguous().unsqueeze(0)
    return torch.sign(t) * torch.log2(torch.abs(t) + 1.0) / np.log2(8)


def position_index(ws):
    c = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij")).flatten(1)
    rel = (c[:, :, None] - c[:, None, :]).permute(1, 2, 0) + (ws - 1)
    return rel[:, :, 0] * (2 * ws - 1) + rel[:, :, 1]


def shift_mask(h, w, ws, shift):
    img = torch.zeros((1, h, w, 1))
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, hs, wsl, :] = cnt
            cnt += 1
    mw = swin_ref.to_windows(img, ws).view(-1, ws * ws)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    return m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)


def alter_plan(name, emd_size=512, img=192):
    """[(layer, idx, kind 'basic'|'attn', cin, cout, stride, has_ds, heads, ws, shift, res)]"""
    blocks, blocks2, heads = ALTER[name]
    plan, inplanes = [], 64
    for li, (planes, nb, nb2, hd, ws, stride) in enumerate(zip((64, 128, 256, emd_size), blocks, blocks2, heads, WINDOWS,
                                                              (1, 2, 2, 2)), start=1):
        res = img // (4 * 2 ** (li - 1))
        num = 2 * (nb // 3) + (nb % 3) - 1
        assert 3 * nb2 <= nb
        alt = [False] * num
        for i in range(nb2):
            alt[-2 * i - 1] = True
        idx = 0
        plan.append((li, idx, "basic", inplanes, planes, stride, stride != 1 or inplanes != planes, hd, ws, 0, res))
        idx += 1
        inplanes = planes
        for is_alt in alt:
            if not is_alt:
                plan.append((li, idx, "basic", planes, planes, 1, False, hd, ws, 0, res))
                idx += 1
            else:
                plan.append((li, idx, "attn", planes, planes, 1, False, hd, ws, 0, res))
                plan.append((li, idx + 1, "attn", planes, planes, 1, False, hd, ws, ws // 2, res))
                idx += 2
    return plan


def attn_block_spec(p, c, heads, ws, shift, res):
    t = 2 * ws - 1
    spec = [(p + ".attn_mask", (res // ws * (res // ws), ws * ws, ws * ws), "attn_mask:%d:%d:%d" % (res, ws, shift))] if shift else []
    spec += [(p + ".attn.logit_scale", (heads, 1, 1), "logit_scale"), (p + ".attn.q_bias", (c,), "linear_b"),
             (p + ".attn.v_bias", (c,), "linear_b"),
             (p + ".attn.relative_coords_table", (1, t, t, 2), "coords:%d" % ws),
             (p + ".attn.relative_position_index", (ws * ws, ws * ws), "posidx:%d" % ws),
             (p + ".attn.cpb_mlp.0.weight", (512, 2), "linear_w"), (p + ".attn.cpb_mlp.0.bias", (512,), "linear_b"),
             (p + ".attn.cpb_mlp.2.weight", (heads, 512), "linear_w"), (p + ".attn.qkv.weight", (3 * c, c), "linear_w"),
             (p + ".attn.proj.weight", (c, c), "linear_w"), (p + ".attn.proj.bias", (c,), "linear_b")]
    return spec + resnet_ref._bn_spec(p + ".norm2", c)


def alter_spec(name, emd_size=512, img=192):
    spec = [("conv1.weight", (64, 3, 3, 3), "conv")] + resnet_ref._bn_spec("bn1", 64)
    for li, idx, kind, cin, cout, stride, ds, hd, ws, shift, res in alter_plan(name, emd_size, img):
        p = "layer%d.%d" % (li, idx)
        if kind == "basic":
            spec.append((p + ".conv1.weight", (cin, cin, 3, 3), "conv"))
            spec += resnet_ref._bn_spec(p + ".bn1", cin)
            spec.append((p + ".conv2.weight", (cout, cin, 3, 3), "conv"))
            spec += resnet_ref._bn_spec(p + ".bn2", cout)
            if ds:
                spec.append((p + ".downsample.0.weight", (cout, cin, 1, 1), "conv"))
                spec += resnet_ref._bn_spec(p + ".downsample.1", cout)
        else:
            spec += attn_block_spec(p, cout, hd, ws, shift, res)
    spec += resnet_ref._bn_spec("bn2", emd_size)
    spec += [("fc.weight", (emd_size, emd_size * 36), "linear_w"), ("fc.bias", (emd_size,), "linear_b")]
    spec += resnet_ref._bn_spec("bn3", emd_size)
    return spec


def fill_special(sd, spec):
    for k, shape, kind in spec:
        if kind.startswith("coords:"):
            sd[k] = coords_table(int(kind.split(":")[1]))
        elif kind.startswith("posidx:"):
            sd[k] = position_index(int(kind.split(":")[1]))
        elif kind.startswith("attn_mask:"):
            _, res, ws, shift = kind.split(":")
            sd[k] = shift_mask(int(res), int(res), int(ws), int(shift))
        elif kind == "logit_scale":
            sd[k] = torch.log(10 * torch.ones(shape)) + 0.05 * torch.arange(shape[0], dtype=torch.float32).view(shape)
    return sd


def attn_block(sd, p, x, heads, ws, shift, training):
    """x NCHW -> NCHW:  x + BN(attention over (shifted) ws x ws windows)"""
    b, c, h, w = x.shape
    y = x.permute(0, 2, 3, 1)
    if shift:
        y = torch.roll(y, shifts=(-shift, -shift), dims=(1, 2))
    a = swin_ref.window_attention(sd, p + ".attn", swin_ref.to_windows(y, ws), heads,
                                  mask=sd[p + ".attn_mask"] if shift else None)
    a = swin_ref.from_windows(a, b, h, w, ws)
    if shift:
        a = torch.roll(a, shifts=(shift, shift), dims=(1, 2))
    return x + resnet_ref._bn(sd, p + ".norm2", a.permute(0, 3, 1, 2), training)


def alter_forward(sd, x, name, training, emd_size=512):
    img = x.shape[-1]
    y = F.conv2d(x, sd["conv1.weight"], None, 2, 1)
    y = F.relu(resnet_ref._bn(sd, "bn1", y, training))
    y = F.max_pool2d(y, 3, 2, 1)
    for li, idx, kind, cin, cout, stride, ds, hd, ws, shift, res in alter_plan(name, emd_size, img):
        p = "layer%d.%d" % (li, idx)
        y = resnet_ref.basic_block(sd, p, y, stride, ds, training) if kind == "basic" else attn_block(sd, p, y, hd, ws, shift, training)
    y = F.relu(resnet_ref._bn(sd, "bn2", y, training))
    y = F.adaptive_avg_pool2d(y, (6, 6)).reshape(y.shape[0], -1)
    y = F.linear(y, sd["fc.weight"], sd["fc.bias"])
    return resnet_ref._bn(sd, "bn3", y, training)
