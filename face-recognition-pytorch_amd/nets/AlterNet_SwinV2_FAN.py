"""MI355X-native drop-in for the reference hybrid conv + window-attention backbone `nets/AlterNet_SwinV2_FAN.py`.

Same surface as /root/reference/nets/AlterNet_SwinV2_FAN.py: `AlterNet50(conf)`, `Encoder(conf)` (:786-839; only
AlterNet50 is constructible in the reference, and only at conf.img_size 192 -- SURVEY.md F9), modules
`WindowAttention`, `SwinTransformerBlock` (attention only, no MLP), `BasicBlock`;
`forward(x: float32[B,3,192,192]) -> float32[B, emd_size]`; state_dict keys/shapes identical (incl. the
`attn_mask` buffers of the shifted blocks and the per-window-size coordinate tables).

Underneath: IR BasicBlocks run on the MFMA conv kernels of nets.resnet; the (W-MSA, SW-MSA) attention pairs on
frhip_winattn_fwd/_bwd with 6x6 / 3x3 windows, where the cyclic roll (:420-440), window partition and the -100
region mask (:375-397) are index arithmetic inside the kernel -- no rolled or partitioned copy of the activation
is ever materialised; qkv / proj are MFMA GEMMs, BN post-norm + residual is the fused bn_apply.
Stem = conv3x3 stride 2 (im2col + GEMM) + fused BN-ReLU-MaxPool; tail = bn2 + ReLU (fused) -> dropout -> fc -> bn3.
DropPath(0.1) (:371) is stochastic depth per sample: identity in eval; in training the per-sample keep mask is
applied to the normalised branch.  One autograd node for the whole net; no CPU fallback.
"""
import numpy as np
import os

import torch
import torch.nn as nn

from frhip import ops

from . import SwinV2 as _S
from ._backbone import (BackwardCtx, BasicBlock, DEFER_EARLY_BLOCKS, Fp8Ctx, Saved, _BN, _Conv, _Linear, basic_block_backward,  # noqa: F401
                        basic_block_forward, bn_forward_state, compute_dtype, encoder_call, prepare_conv_weights,
                        stem_backward,
                        stem_forward, tail_backward, tail_forward, use_fp8)

conv1x1 = lambda cin, cout, stride=1: _Conv(cin, cout, 1, stride)  # noqa: E731


class WindowAttention(_S.WindowAttention):
    """Cosine window attention holder for any window size <= 7 (reference :187-302)."""

    def __init__(self, dim, window_size, num_heads, dim_head=32, qkv_bias=True, attn_drop=0.0, proj_drop=0.0,
                 pretrained_window_size=(0, 0)):
        nn.Module.__init__(self)
        assert dim_head * num_heads == dim, "Not match dim_head * num_heads and hidden_dim"
        ws = int(window_size[0])
        assert ws == int(window_size[1]) and 2 <= ws <= 7
        self.dim, self.window_size, self.num_heads = dim, tuple(window_size), num_heads
        self.logit_scale = nn.Parameter(torch.log(10 * torch.ones((num_heads, 1, 1))))
        self.cpb_mlp = nn.Sequential(nn.Linear(2, 512, bias=True), nn.ReLU(inplace=True), nn.Linear(512, num_heads, bias=False))
        r = torch.arange(-(ws - 1), ws, dtype=torch.float32) / (ws - 1) * 8
        table = torch.stack(torch.meshgrid([r, r], indexing="ij")).permute(1, 2, 0).contiguous().unsqueeze(0)
        self.register_buffer("relative_coords_table", torch.sign(table) * torch.log2(torch.abs(table) + 1.0) / np.log2(8))
        c = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij")).flatten(1)
        rel = (c[:, :, None] - c[:, None, :]).permute(1, 2, 0) + (ws - 1)
        self.register_buffer("relative_position_index", rel[:, :, 0] * (2 * ws - 1) + rel[:, :, 1])
        self.qkv = _Linear(dim, dim * 3, bias=False)
        self.q_bias = nn.Parameter(torch.zeros(dim)) if qkv_bias else None
        self.v_bias = nn.Parameter(torch.zeros(dim)) if qkv_bias else None
        self.proj = _Linear(dim, dim)


class SwinTransformerBlock(nn.Module):
    """x = x + DropPath(BN(attn(x)))  (reference :306-450): parameter holder; dim == dim_out in every AlterNet."""

    def __init__(self, dim, dim_out, heads, input_resolution, window_size=7, shift_size=0, qkv_bias=True, drop=0.0,
                 attn_drop=0.0, drop_path=0.1, norm_layer=None, pretrained_window_size=0, activation=None):
        super().__init__()
        if dim != dim_out:
            raise NotImplementedError("the 1x1 shortcut branch (reference :351-356) is unused by every AlterNet config")
        self.dim, self.num_heads, self.window_size, self.shift_size = dim, heads, window_size, shift_size
        self.drop_path_rate = drop_path
        if shift_size > 0:
            h, w = input_resolution
            img = torch.zeros((1, h, w, 1))
            cnt = 0
            for hs in (slice(0, -window_size), slice(-window_size, -shift_size), slice(-shift_size, None)):
                for wsl in (slice(0, -window_size), slice(-window_size, -shift_size), slice(-shift_size, None)):
                    img[:, hs, wsl, :] = cnt
                    cnt += 1
            mw = img.view(1, h // window_size, window_size, w // window_size, window_size, 1).permute(0, 1, 3, 2, 4, 5)
            mw = mw.reshape(-1, window_size * window_size)
            m = mw.unsqueeze(1) - mw.unsqueeze(2)
            mask = m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)
        else:
            mask = None
        self.register_buffer("attn_mask", mask)          # kept for state_dict parity; the kernel derives it from indices
        self.attn = WindowAttention(dim, (window_size, window_size), heads, qkv_bias=qkv_bias)
        self.norm2 = _BN(dim)


def attn_block_forward(blk, x, dt, training, save, wprep=None, q8=None, keep=None):
    b, h, w, c = x.shape
    m = b * h * w
    at = blk.attn
    x2 = x.view(m, c)
    fp8 = q8 is not None and dt == torch.bfloat16 and Fp8Ctx.eligible(c)        # qkv / proj with both operands in fp8
    wqkv, wqkv_t = _S._lin_operands(at.qkv, dt, wprep)
    qb = _S.qkv_bias(at)
    if fp8:
        qkv, _ = ops.linear_fwd_fp8(q8.take(x).view(m, c), *q8.packs[at.qkv], bias=qb)
    else:
        qkv, _, _ = ops.linear_fwd(x2, wqkv, qb)                              # bias add in the GEMM epilogue
    cpb_batch, _, bias, scale, dbias_buf, dscale_buf = _S.position_bias(blk)
    ao = ops.winattn_fwd(qkv, bias, scale, b, h, w, at.num_heads, blk.window_size, blk.shift_size)
    wproj, wproj_t = _S._lin_operands(at.proj, dt, wprep)
    if fp8:
        po, part2 = ops.linear_fwd_fp8(ops.quant_fp8(ao), *q8.packs[at.proj], bias=at.proj.bias.data, want_stats=training)
    else:
        po, _, part2 = ops.linear_fwd(ao, wproj, at.proj.bias.data, want_stats=training)   # + norm2's batch statistics
    st2 = bn_forward_state(blk.norm2, part2, m, training)
    if not (training and blk.drop_path_rate > 0):
        keep = None
    if training and blk.drop_path_rate > 0:
        # stochastic depth: one Bernoulli(keep) per sample scales the whole normalised branch (timm DropPath); the factor rides in the
        # BatchNorm-apply pass (frhip_bn_apply_rs) instead of separate multiply and add passes over the tensor
        if keep is None:
            kp = 1.0 - blk.drop_path_rate
            keep = (torch.rand(b, device=x.device) < kp).float() / kp
        out = ops.bn_apply(po, st2, res=x2, rowscale=keep, rows_per=h * w).view(b, h, w, c)
    elif fp8:
        out, out8 = ops.bn_apply_q8(po, st2, res=x2)
        out = out.view(b, h, w, c)
        q8.put(out, out8.view(b, h, w, c))
    else:
        out = ops.bn_apply(po, st2, res=x2).view(b, h, w, c)
    s = None
    if save:
        s = Saved()
        (s.x2, s.wqkv, s.qkv, s.cpb_batch, s.dbias, s.dscale, s.bias, s.scale, s.ao, s.wproj, s.po, s.st2, s.keep, s.shape) = (
            x2, wqkv, qkv, cpb_batch, dbias_buf, dscale_buf, bias, scale, ao, wproj, po, st2, keep, (b, h, w, c))
        s.wqkv_t, s.wproj_t = wqkv_t, wproj_t
    return out, s


def attn_block_backward(blk, s, dout, dt, bc, part2=None, next_bn=None):
    """part2: the BatchNorm-backward partial sums of (dout, s.po) when the kernel that produced dout already reduced them (under stochastic
    depth: with the per-sample factor, frhip_conv_dgrad_fused_rs).  next_bn=(y, st[, relu]): the BatchNorm that consumes the returned
    dx; its reduction then rides in the epilogue of the qkv data-gradient and (dx, partial) is returned (as nets.SwinV2.swin_block_backward)."""
    G = bc.G
    b, h, w, c = s.shape
    m = b * h * w
    at = blk.attn
    d2 = dout.reshape(m, c)
    dpo = ops.bn_backward(d2, s.po, s.st2, blk.norm2.weight.data, G(blk.norm2.weight), G(blk.norm2.bias),
                          rowscale=s.keep, rows_per=h * w, part=part2)
    # proj.bias only shifts the input of a training-mode BatchNorm: analytically zero gradient (nets/SwinV2.py), left at zero
    if not _PAIR_HANDOVER:                        # else: proj's and qkv's weight gradients go to the side stream in ONE hand-over below
        bc.on_side(lambda: ops.gemm_tn(dpo, s.ao, G(at.proj.weight)), dpo, s.ao)
    dao = ops.gemm_nt(dpo, _S._transposed(s.wproj, s.wproj_t))
    dqkv, _, _, gsum = ops.winattn_bwd(s.qkv, dao, s.bias, s.scale, b, h, w, at.num_heads, blk.window_size,
                                       blk.shift_size, want_colsum=True, dbias=s.dbias, dscale=s.dscale,
                                       qv_grads=(G(at.q_bias), G(at.v_bias)))
    if gsum is None:                                 # fp32 validation kernels: column sums by a ones-GEMM
        gsum = torch.zeros(3 * c, dtype=torch.float32, device=dout.device)
        _S._colsum_via_gemm(dqkv, gsum)
    if gsum is not True:                             # bf16 MFMA kernel: already added into the two gradient accumulators
        G(at.q_bias).add_(gsum[:c])
        G(at.v_bias).add_(gsum[2 * c:])
    if _PAIR_HANDOVER:
        bc.on_side(lambda: (ops.gemm_tn(dpo, s.ao, G(at.proj.weight)), ops.gemm_tn(dqkv, s.x2, G(at.qkv.weight))), dpo, s.ao, dqkv, s.x2)
    else:
        bc.on_side(lambda: ops.gemm_tn(dqkv, s.x2, G(at.qkv.weight)), dqkv, s.x2)
    part = None
    if next_bn is not None:
        dx, part = _S._dgrad_add(dqkv, s.wqkv, d2, s.wqkv_t,
                                 bnred=(next_bn[0], next_bn[1], len(next_bn) > 2 and bool(next_bn[2])) + tuple(next_bn[3:]))
    else:
        dx = _S._dgrad_add(dqkv, s.wqkv, d2, s.wqkv_t)
    _S.position_bias_backward(blk, s, bc)
    return dx.view(b, h, w, c) if next_bn is None else (dx.view(b, h, w, c), part)


_FUSE_BNRED = os.environ.get("FRHIP_ALT_FUSE_BNRED", "1") == "1"      # 0: conv -> conv transitions only (the round-3 behaviour; A/B switch)
_PAIR_HANDOVER = os.environ.get("FRHIP_ALT_PAIR_HANDOVER", "1") == "1"   # proj's and qkv's weight gradients in ONE hand-over to the side stream
_FUSE_BNRED_RS = os.environ.get("FRHIP_ALT_FUSE_BNRED_RS", "1") == "1"   # 0: attention blocks under stochastic depth keep their own reduction pass


class AlterNet(nn.Module):
    def __init__(self, conf, block, block2, num_blocks, num_blocks2, heads):
        super().__init__()
        self.emd_size = conf.emd_size
        self.dtype = compute_dtype(conf)
        self.fp8 = use_fp8(conf)                   # BASELINE cfg 5: forward GEMMs on the fp8 MFMA path (csrc/igemm_fp8.hip)
        res = (conf.img_size, conf.img_size)
        self.inplanes = 64
        self.conv1 = _Conv(3, 64, 3, 2)
        self.bn1 = _BN(64)
        self.layer1 = self.stack_layers(block, block2, 64, num_blocks[0], num_blocks2[0], heads[0], (res[0] // 4, res[1] // 4), window_size=6)
        self.layer2 = self.stack_layers(block, block2, 128, num_blocks[1], num_blocks2[1], heads[1], (res[0] // 8, res[1] // 8), stride=2, window_size=6)
        self.layer3 = self.stack_layers(block, block2, 256, num_blocks[2], num_blocks2[2], heads[2], (res[0] // 16, res[1] // 16), stride=2, window_size=6)
        self.layer4 = self.stack_layers(block, block2, conf.emd_size, num_blocks[3], num_blocks2[3], heads[3], (res[0] // 32, res[1] // 32), stride=2, window_size=3)
        self.bn2 = _BN(block.expansion * conf.emd_size)
        self.dropout = nn.Dropout()
        self.fc = _Linear(block.expansion * conf.emd_size * 6 * 6, conf.emd_size)
        self.bn3 = _BN(conf.emd_size)
        for m in self.modules():
            if isinstance(m, (_Conv, _Linear, nn.Linear)):
                nn.init.xavier_normal_(m.weight)
                if getattr(m, "bias", None) is not None:
                    nn.init.constant_(m.bias, 0)

    def stack_layers(self, block, block2, planes, blocks, blocks2, heads, input_resolution, window_size=3, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(conv1x1(self.inplanes, planes * block.expansion, stride), _BN(planes * block.expansion))
        num_blocks = 2 * (blocks // 3) + (blocks % 3) - 1
        assert 2 * blocks2 + blocks2 <= blocks, "The number of transformers must not exceed cnn !!!"
        alt_seq = [False] * num_blocks
        for i in range(blocks2):
            alt_seq[-2 * i - 1] = True
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for is_alt in alt_seq:
            if not is_alt:
                layers.append(block(self.inplanes, planes))
            else:
                layers.append(block2(self.inplanes, planes, heads=heads, input_resolution=input_resolution, window_size=window_size))
                layers.append(block2(self.inplanes, planes, heads=heads, input_resolution=input_resolution,
                                     shift_size=window_size // 2, window_size=window_size))
        return nn.Sequential(*layers)

    def _layers(self):
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for mod in layer:
                yield mod

    def forward(self, x):
        return encoder_call(self, x)

    def _forward_impl(self, x, training, save):
        dt = self.dtype
        sv = Saved() if save else None
        layers = list(self._layers())
        lprep = None
        attn = [m for m in layers if not isinstance(m, BasicBlock)]
        _S.precompute_position_bias(attn, x.device)         # every block's bias table and logit scale: one launch
        if training and save:
            lprep = _S.prepare_linear_weights([l for b in attn for l in (b.attn.qkv, b.attn.proj)], dt)
        cur = stem_forward(self, x, training, sv)
        saved = []
        convs = [c for b in layers if isinstance(b, BasicBlock)
                 for c in ((b.conv1, b.conv2) + ((b.downsample[0],) if b.downsample is not None else ()))]
        wprep = prepare_conv_weights(convs, dt) if convs else None
        q8 = None
        if self.fp8 and dt == torch.bfloat16:
            lin = [l for b in layers if not isinstance(b, BasicBlock) for l in (b.attn.qkv, b.attn.proj)]
            q8 = Fp8Ctx([(c, c.physical()) for c in convs if Fp8Ctx.eligible(c.cin)] +
                        [(l, l.weight.data) for l in lin if Fp8Ctx.eligible(l.weight.shape[1])])
        # stochastic-depth draws of every attention block in one go (four launches per step instead of four per block)
        keeps, ki = None, 0
        if training:
            rates = [m.drop_path_rate for m in layers if not isinstance(m, BasicBlock)]
            if any(r > 0 for r in rates):
                kp = getattr(self, "_keep_prob", None)
                if kp is None or kp.device != cur.device or kp.shape[0] != len(rates):
                    kp = self._keep_prob = (1.0 - torch.tensor(rates, dtype=torch.float32).view(-1, 1)).to(cur.device)
                keeps = (torch.rand((len(rates), cur.shape[0]), device=cur.device) < kp).float() / kp
        for mod in layers:
            if isinstance(mod, BasicBlock):
                cur, s = basic_block_forward(mod, cur, dt, training, save, wprep, q8)
            else:
                cur, s = attn_block_forward(mod, cur, dt, training, save, lprep, q8, keep=keeps[ki] if keeps is not None else None)
                ki += 1
            saved.append(s)
        if cur.shape[1] != 6 or cur.shape[2] != 6:
            raise NotImplementedError("AdaptiveAvgPool2d((6,6)) is the identity only for 192x192 inputs")
        mask = None
        if training and self.dropout.p > 0:
            keep = 1.0 - self.dropout.p
            mask = ops.dropout_mask(cur.shape, cur.dtype, keep, cur.device)
        emb = tail_forward(self, cur, training, sv, dropout_mask=mask, relu=True)
        if save:
            sv.layers = saved
        return emb, sv

    def _backward_impl(self, sv, d_emb, params):
        dt = self.dtype
        bc = BackwardCtx(params, d_emb.device, allreduce=getattr(self, "_frhip_allreduce", False))
        dout = tail_backward(self, sv, d_emb, bc)
        layers = list(self._layers())
        part = None
        for i in range(len(layers) - 1, -1, -1):
            mod, s = layers[i], sv.layers[i]
            # the gradient leaving layer i enters the last BatchNorm of layer i-1 (bn2 of a conv block, norm2 of an attention block): that
            # BatchNorm's backward sums ride in layer i's last data-gradient -- unless layer i-1 is an attention block under stochastic
            # depth, whose reduction carries a per-sample factor (frhip_bn_bwd_reduce_rs) the producing kernel does not know
            nxt = None
            if i > 0 and _FUSE_BNRED:
                ps = sv.layers[i - 1]
                if isinstance(layers[i - 1], BasicBlock):
                    nxt = (ps.y2, ps.st2)
                elif ps.keep is None:
                    nxt = (ps.po.view(ps.shape), ps.st2, False)
                elif _FUSE_BNRED_RS:
                    # stochastic depth: the sums carry the per-sample factor (frhip_conv_dgrad_fused_rs)
                    nxt = (ps.po.view(ps.shape), ps.st2, False, ps.keep, ps.shape[1] * ps.shape[2], 1.0 / (1.0 - layers[i - 1].drop_path_rate))
            if isinstance(mod, BasicBlock):
                res = basic_block_backward(mod, s, dout, dt, bc, part2=part, next_bn=nxt)
            else:
                res = attn_block_backward(mod, s, dout, dt, bc, part2=part, next_bn=nxt)
            dout, part = res if nxt is not None else (res, None)
            if len(layers) - 1 - i == DEFER_EARLY_BLOCKS:
                bc.run_deferred()                          # the head's early parameter update: beside the blocks, not beside the tail
        stem_backward(self, sv, dout, bc)
        return bc.join()


def AlterNet50(conf, **kwargs):
    return AlterNet(conf, BasicBlock, SwinTransformerBlock, num_blocks=[3, 4, 14, 4], num_blocks2=[0, 1, 4, 1],
                    heads=(2, 4, 8, 16), **kwargs)


def Encoder(conf):
    """Name dispatch of the reference (:831-839).  AlterNet18/34/100/200 fail the reference's own block-budget assert
    (SURVEY.md F9), so only AlterNet50 exists."""
    if conf.network == "AlterNet50":
        return AlterNet50(conf)
    return None
