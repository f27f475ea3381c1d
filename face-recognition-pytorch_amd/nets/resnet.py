"""MI355X-native drop-in for the reference backbone module `nets/resnet.py`.

Same surface as /root/reference/nets/resnet.py: `ResNet18/34/50/100/200(conf)`, `Encoder(conf)` (:253-316);
each returns an nn.Module whose `forward(x: float32[B,3,H,W]) -> float32[B, conf.emd_size]` and whose
state_dict has exactly the reference's keys/shapes (conv1.weight, bn1.*, layerS.B.{conv1,bn1,conv2,bn2,
downsample.0,downsample.1}.*, bn2.*, fc.*, bn3.*), so reference checkpoints load with strict=True.

What is different underneath (nothing is delegated to torch's conv/BN kernels):
  * activations are NHWC in the compute dtype (bf16, or fp32 "validation mode"); conv weights are
    channels_last Parameters, i.e. already [K][R][S][C] in memory;
  * every convolution is the hand-written MFMA implicit GEMM of libfrhip (forward, data-gradient and
    weight-gradient), BN batch statistics come out of the conv epilogue, BN-apply/ReLU/residual are fused
    element-wise passes, the stem is im2col + GEMM + fused BN-ReLU-MaxPool;
  * the whole backbone is ONE autograd node: forward keeps the activations it needs, backward runs the
    hand-written gradient kernels and returns all parameter gradients.
There is no CPU / eager fallback: without libfrhip.so or without a GPU tensor, forward raises.
"""
import os

import torch
import torch.nn as nn

from frhip import ops

_OVERLAP_WGRAD = os.environ.get("FRHIP_OVERLAP_WGRAD", "1") == "1"
_BLOCKS = {18: (2, 2, 2, 2), 34: (3, 4, 6, 4), 50: (3, 4, 14, 4), 100: (3, 13, 30, 4), 200: (3, 43, 50, 4)}
_DTYPES = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp32": torch.float32, "float32": torch.float32}


def compute_dtype(conf):
    """bf16 MFMA by default; conf.frhip_dtype or $FRHIP_DTYPE = 'fp32' selects the exact-fp32 validation mode."""
    name = getattr(conf, "frhip_dtype", None) or os.environ.get("FRHIP_DTYPE", "bf16")
    return _DTYPES[str(name).lower()]


# ------------------------------------------------------------------------------------------------- containers
class _Conv(nn.Module):
    """Parameter holder with the reference's name/shape ([K,C,R,S]); storage is channels_last = [K][R][S][C]."""

    def __init__(self, cin, cout, k, stride):
        super().__init__()
        self.cin, self.cout, self.k, self.stride, self.pad = cin, cout, k, stride, (k - 1) // 2
        w = torch.empty(cout, cin, k, k).contiguous(memory_format=torch.channels_last)
        self.weight = nn.Parameter(w)

    def physical(self):
        """fp32 [K,R,S,C] view of the weight (a copy only if someone replaced the channels_last storage)."""
        p = self.weight.data.permute(0, 2, 3, 1)
        return p if p.is_contiguous() else p.contiguous()


class _BN(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.eps, self.momentum = 1e-5, 0.1


class _Linear(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin))
        self.bias = nn.Parameter(torch.zeros(cout))


class BasicBlock(nn.Module):
    """conv3x3(inplanes->inplanes) - BN - ReLU - conv3x3(inplanes->planes, stride) - BN, + shortcut
    (reference nets/resnet.py:55-103).  Holds parameters only; the math runs in ResNet.forward."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _Conv(inplanes, inplanes, 3, 1)
        self.bn1 = _BN(inplanes)
        self.conv2 = _Conv(inplanes, planes, 3, stride)
        self.bn2 = _BN(planes)
        self.downsample = downsample
        self.stride = stride


class _Saved:
    pass


# ------------------------------------------------------------------------------------------------- network
class ResNet(nn.Module):
    def __init__(self, block, layers, conf):
        super().__init__()
        self.emd_size = conf.emd_size
        self.dtype = compute_dtype(conf)
        self.inplanes = 64
        self.conv1 = _Conv(3, 64, 3, 1)
        self.bn1 = _BN(64)
        self.layer1 = self.stack_layers(block, 64, layers[0])
        self.layer2 = self.stack_layers(block, 128, layers[1], stride=2)
        self.layer3 = self.stack_layers(block, 256, layers[2], stride=2)
        self.layer4 = self.stack_layers(block, conf.emd_size, layers[3], stride=2)
        self.bn2 = _BN(block.expansion * conf.emd_size)
        self.fc = _Linear(block.expansion * conf.emd_size * 7 * 7, conf.emd_size)
        self.bn3 = _BN(conf.emd_size)
        # same initialisation rule as the reference (nets/resnet.py:201-209)
        for m in self.modules():
            if isinstance(m, (_Conv, _Linear)):
                nn.init.xavier_normal_(m.weight)

    def stack_layers(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(_Conv(self.inplanes, planes * block.expansion, 1, stride),
                                       _BN(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    # ---- parameter order used by the autograd node
    def _blocks(self):
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                yield blk

    def _train_params(self):
        return [p for p in self.parameters()]

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("nets.resnet (frhip): input must live on the MI355X; there is no CPU path "
                               "(the CPU restatement lives in oracle/ and is test-only)")
        x = x.contiguous().float()
        if self.training and torch.is_grad_enabled():
            params = self._train_params()
            return _EncoderFn.apply(self, x, *params)
        out, _ = _forward_impl(self, x, self.training, save=False)
        return out


# ------------------------------------------------------------------------------------------------- forward
def _bn_forward_state(bn, part, count, training):
    if training:
        st = ops.bn_finalize(part, count, bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var,
                             bn.momentum, bn.eps)
        bn.num_batches_tracked += 1
        return st
    return ops.bn_eval_affine(bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var, bn.eps)


def _forward_impl(net, x, training, save):
    dt = net.dtype
    b, _, h, w = x.shape
    sv = _Saved() if save else None
    # ---- stem: conv3x3(3->64) as im2col + GEMM, then fused BN + ReLU + MaxPool(3,2,1)
    col = ops.stem_im2col(x, dt)
    wp0 = ops.pack_stem(net.conv1.physical().reshape(64, 27), dt)
    y0, part = ops.conv_fwd(col.view(b * h * w, 1, 1, col.shape[1]), wp0, 1, 0, want_stats=training)
    y0 = y0.view(b, h, w, 64)
    st0 = _bn_forward_state(net.bn1, part, b * h * w, training)
    cur, arg0 = ops.bn_relu_maxpool_fwd(y0, st0)
    if save:
        sv.col, sv.y0, sv.st0, sv.arg0, sv.blocks = col, y0, st0, arg0, []
    # ---- residual stages
    for blk in net._blocks():
        xin = cur
        w1 = ops.cast_from_f32(blk.conv1.physical(), dt)
        y1, p1 = ops.conv_fwd(xin, w1, 1, 1, want_stats=training)
        st1 = _bn_forward_state(blk.bn1, p1, y1.numel() // y1.shape[3], training)
        a1 = ops.bn_apply(y1, st1, relu=True)
        w2 = ops.cast_from_f32(blk.conv2.physical(), dt)
        y2, p2 = ops.conv_fwd(a1, w2, blk.stride, 1, want_stats=training)
        st2 = _bn_forward_state(blk.bn2, p2, y2.numel() // y2.shape[3], training)
        yd = std = None
        if blk.downsample is not None:
            dconv, dbn = blk.downsample[0], blk.downsample[1]
            wd = ops.cast_from_f32(dconv.physical(), dt)
            yd, pd = ops.conv_fwd(xin, wd, dconv.stride, 0, want_stats=training)
            std = _bn_forward_state(dbn, pd, yd.numel() // yd.shape[3], training)
            cur = ops.bn_apply(y2, st2, res=yd, res_st=std)
        else:
            cur = ops.bn_apply(y2, st2, res=xin)
        if save:
            s = _Saved()
            s.x, s.y1, s.st1, s.a1, s.y2, s.st2, s.yd, s.std = xin, y1, st1, a1, y2, st2, yd, std
            sv.blocks.append(s)
    # ---- tail: bn2 -> flatten (NHWC order; fc columns permuted to match) -> fc -> bn3
    bo, ho, wo, co = cur.shape
    rows = bo * ho * wo
    part = ops.colstats(cur.view(rows, co)) if training else None
    stt = _bn_forward_state(net.bn2, part, rows, training)
    z = ops.bn_apply(cur, stt)
    flat = z.view(bo, ho * wo * co)
    wfc = ops.fc_permute(net.fc.weight.data, co, ho * wo, dt)
    f = ops.gemm_nt(flat, wfc, splits=16, atomic_f32=True)
    ops.add_bias(f, net.fc.bias.data)
    part = ops.colstats(f) if training else None
    st3 = _bn_forward_state(net.bn3, part, bo, training)
    emb = ops.bn_apply(f, st3)
    if save:
        sv.out4, sv.stt, sv.flat, sv.wfc, sv.f, sv.st3 = cur, stt, flat, wfc, f, st3
    return emb, sv


# ------------------------------------------------------------------------------------------------- backward
def _grad_like(p):
    """zero fp32 gradient with the same memory layout as the parameter"""
    return torch.zeros_like(p.data, memory_format=torch.preserve_format)


def _phys_grad(conv, g):
    pg = g.permute(0, 2, 3, 1)
    assert pg.is_contiguous()
    return pg


def _flat_grads(params, device):
    """One zeroed fp32 arena for every parameter gradient of the step (a single fill instead of ~160), carved
    into views that have each parameter's own memory layout (channels_last for conv weights)."""
    total = sum(p.numel() for p in params)
    flat = torch.zeros(total, dtype=torch.float32, device=device)
    views, off = {}, 0
    for p in params:
        n = p.numel()
        chunk = flat[off:off + n]
        if p.dim() == 4 and p.data.permute(0, 2, 3, 1).is_contiguous():
            k, c, r, s = p.shape
            views[p] = chunk.view(k, r, s, c).permute(0, 3, 1, 2)
        elif p.data.is_contiguous():
            views[p] = chunk.view(p.shape)
        else:
            views[p] = _grad_like(p)
        off += n
    return views


_SIDE_STREAMS = {}


def _side_stream(device):
    """one long-lived side stream per device for the weight-gradient GEMMs"""
    key = torch.device(device).index
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


def _backward_impl(net, sv, d_emb, params):
    dt = net.dtype
    grads = _flat_grads(params, d_emb.device)
    # Weight gradients do not feed the rest of the backward chain, so they run on a side HIP stream and fill the
    # gaps (partially filled last rounds, HBM-bound BN passes) of the data-gradient chain on the main stream.
    # Every tensor a side-stream kernel reads is kept referenced in `keep` until the streams are joined again.
    main = torch.cuda.current_stream()
    side = _side_stream(d_emb.device) if _OVERLAP_WGRAD else None
    keep = []

    def wgrad(dy, x, gview, r, s, stride, pad):
        if side is None:
            ops.conv_wgrad(dy, x, gview, r, s, stride, pad)
            return
        keep.append((dy, x, gview))
        side.wait_stream(main)
        with torch.cuda.stream(side):
            ops.conv_wgrad(dy, x, gview, r, s, stride, pad)

    def G(p):
        return grads[p]

    # ---- tail
    df = ops.bn_backward(d_emb.contiguous().float(), sv.f, sv.st3, net.bn3.weight.data, G(net.bn3.weight), G(net.bn3.bias))
    ops.colsum_accumulate(df, G(net.fc.bias))
    dft = ops.cast_from_f32(df, dt)
    b, kfc = sv.flat.shape
    wfct = ops.transpose2d(sv.wfc)                                  # [25088][512]
    dflat = ops.gemm_nt(dft, wfct)                                  # [B][25088]
    dwp = torch.zeros((net.emd_size, kfc), dtype=torch.float32, device=d_emb.device)
    ops.gemm_tn(dft, sv.flat, dwp)
    co = sv.out4.shape[3]
    ops.fc_unpermute_grad(dwp, G(net.fc.weight), co, kfc // co)
    dout = ops.bn_backward(dflat.view(sv.out4.shape), sv.out4, sv.stt, net.bn2.weight.data, G(net.bn2.weight), G(net.bn2.bias))
    # ---- residual stages, last to first
    for blk, s in zip(reversed(list(net._blocks())), reversed(sv.blocks)):
        dy2 = ops.bn_backward(dout, s.y2, s.st2, blk.bn2.weight.data, G(blk.bn2.weight), G(blk.bn2.bias))
        shortcut = dout
        if blk.downsample is not None:
            dconv, dbn = blk.downsample[0], blk.downsample[1]
            dyd = ops.bn_backward(dout, s.yd, s.std, dbn.weight.data, G(dbn.weight), G(dbn.bias))
            wdt = ops.pack_wt(dconv.physical(), dt)
            shortcut = ops.conv_dgrad(dyd, wdt, s.x.shape, 1, 1, dconv.stride, 0)
            wgrad(dyd, s.x, _phys_grad(dconv, G(dconv.weight)), 1, 1, dconv.stride, 0)
        w2t = ops.pack_wt(blk.conv2.physical(), dt)
        da1 = ops.conv_dgrad(dy2, w2t, s.a1.shape, 3, 3, blk.stride, 1)
        wgrad(dy2, s.a1, _phys_grad(blk.conv2, G(blk.conv2.weight)), 3, 3, blk.stride, 1)
        dy1 = ops.bn_backward(da1, s.y1, s.st1, blk.bn1.weight.data, G(blk.bn1.weight), G(blk.bn1.bias), relu_mask=True)
        w1t = ops.pack_wt(blk.conv1.physical(), dt)
        dout = ops.conv_dgrad(dy1, w1t, s.x.shape, 3, 3, 1, 1, residual=shortcut)
        wgrad(dy1, s.x, _phys_grad(blk.conv1, G(blk.conv1.weight)), 3, 3, 1, 1)
    # ---- stem
    da0 = ops.maxpool_bwd(dout, sv.arg0, sv.y0.shape)
    dy0 = ops.bn_backward(da0, sv.y0, sv.st0, net.bn1.weight.data, G(net.bn1.weight), G(net.bn1.bias), relu_mask=True)
    m = sv.col.shape[0]
    kp = sv.col.shape[1]
    dwp0 = torch.zeros((64, 1, 1, kp), dtype=torch.float32, device=d_emb.device)
    ops.conv_wgrad(dy0.view(m, 1, 1, 64), sv.col.view(m, 1, 1, kp), dwp0, 1, 1, 1, 0)
    ops.unpack_stem_grad(dwp0, _phys_grad(net.conv1, G(net.conv1.weight)).view(64, 27))
    if side is not None:
        main.wait_stream(side)
    del keep
    return grads


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x, *params):
        emb, sv = _forward_impl(net, x, True, save=True)
        ctx.net, ctx.sv, ctx.params = net, sv, params
        return emb

    @staticmethod
    def backward(ctx, d_emb):
        grads = _backward_impl(ctx.net, ctx.sv, d_emb, ctx.params)
        ctx.sv = None
        return (None, None) + tuple(grads.get(p) for p in ctx.params)


# ------------------------------------------------------------------------------------------------- constructors
def ResNet18(conf, **kwargs):
    return ResNet(BasicBlock, list(_BLOCKS[18]), conf, **kwargs)


def ResNet34(conf, **kwargs):
    return ResNet(BasicBlock, list(_BLOCKS[34]), conf, **kwargs)


def ResNet50(conf, **kwargs):
    return ResNet(BasicBlock, list(_BLOCKS[50]), conf, **kwargs)


def ResNet100(conf, **kwargs):
    return ResNet(BasicBlock, list(_BLOCKS[100]), conf, **kwargs)


def ResNet200(conf, **kwargs):
    return ResNet(BasicBlock, list(_BLOCKS[200]), conf, **kwargs)


def Encoder(conf):
    """Name dispatch of the reference (nets/resnet.py:308-316) -- plus 'ResNet18', which the reference's
    dispatcher forgets although its constructor exists."""
    table = {"ResNet200": ResNet200, "ResNet100": ResNet100, "ResNet50": ResNet50, "ResNet34": ResNet34,
             "ResNet18": ResNet18}
    if conf.network in table:
        return table[conf.network](conf)
    return None
