"""MI355X-native drop-in for the reference SwinV2-style backbone `nets/SwinV2.py`.

Same surface as /root/reference/nets/SwinV2.py: `Swin18/34/50/100/200(conf)`, `Encoder(conf)` (:570-656), modules
`WindowAttention`, `SwinTransformerBlock`, `Mlp`; `forward(x: float32[B,3,112,112]) -> float32[B, emd_size]`; the
state_dict has the reference's keys and shapes (`layerL.I.attn.{logit_scale,q_bias,v_bias,relative_coords_table,
relative_position_index,cpb_mlp.0.weight,cpb_mlp.0.bias,cpb_mlp.2.weight,qkv.weight,proj.weight,proj.bias}`,
`norm2.*`, `mlp.fc1/fc2.{weight,bias}`, `norm3.*`, the 2x2/s2 stage convolutions as `layerL.0.weight`).

Underneath (window_size 7, shift 0 -- the only configuration the reference can run):
  * activations stay NHWC; window_partition / window_reverse (:35-62) are index arithmetic inside the attention
    kernel, so the permute/contiguous copies of the reference do not exist;
  * qkv / proj / fc1 / fc2 are the MFMA NT GEMM on [B*H*W, C] rows, their weight gradients the TN GEMM, the
    data gradients the NT GEMM with the residual add fused; biases and the exact-erf GELU are one element-wise
    pass; BN post-norm + residual is the fused bn_apply of the ResNet path;
  * cosine attention with the continuous-position bias and the clamped learnable scale is frhip_winattn_fwd/_bwd;
  * the 169-entry continuous-position-bias MLP (2 -> 512 -> heads) is parameter-space work done once per block
    and step; it runs as three torch ops on the device and its gradient through torch.autograd.grad.
The whole backbone is one autograd node (nets/_backbone.EncoderFn).  No CPU fallback.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from frhip import ops

from ._backbone import (BackwardCtx, BasicBlock, DEFER_EARLY_BLOCKS, Saved, _BN, _Conv, _Linear, bn_forward_state, compute_dtype,  # noqa: F401
                        stem_reduction_operands,
                        encoder_call, phys_grad, stem_backward, stem_forward, tail_backward, tail_forward)

LN100 = math.log(1.0 / 0.01)


class Mlp(nn.Module):
    """1x1-conv MLP, hidden = 4*dim, GELU (reference :16-32): parameter holder."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=None, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = _Conv(in_features, hidden_features, 1, 1, bias=True)
        self.fc2 = _Conv(hidden_features, out_features, 1, 1, bias=True)


class WindowAttention(nn.Module):
    """Cosine window attention with continuous relative position bias (reference :65-179): parameter holder +
    the two table buffers, registered in the reference's order."""

    def __init__(self, dim, window_size, num_heads, dim_head=32, qkv_bias=True, attn_drop=0.0, proj_drop=0.0,
                 pretrained_window_size=(0, 0)):
        super().__init__()
        assert dim_head * num_heads == dim, "Not match dim_head * num_heads and hidden_dim"
        assert tuple(window_size) == (7, 7) and dim_head == 32, "frhip window attention: 7x7 windows, head dim 32"
        self.dim, self.window_size, self.num_heads = dim, tuple(window_size), num_heads
        self.logit_scale = nn.Parameter(torch.log(10 * torch.ones((num_heads, 1, 1))))
        self.cpb_mlp = nn.Sequential(nn.Linear(2, 512, bias=True), nn.ReLU(inplace=True), nn.Linear(512, num_heads, bias=False))
        ws = 7
        r = torch.arange(-(ws - 1), ws, dtype=torch.float32) / (ws - 1) * 8
        table = torch.stack(torch.meshgrid([r, r], indexing="ij")).permute(1, 2, 0).contiguous().unsqueeze(0)
        table = torch.sign(table) * torch.log2(torch.abs(table) + 1.0) / np.log2(8)
        self.register_buffer("relative_coords_table", table)
        c = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij")).flatten(1)
        rel = (c[:, :, None] - c[:, None, :]).permute(1, 2, 0) + (ws - 1)
        self.register_buffer("relative_position_index", rel[:, :, 0] * (2 * ws - 1) + rel[:, :, 1])
        self.qkv = _Linear(dim, dim * 3, bias=False)
        if qkv_bias:
            self.q_bias = nn.Parameter(torch.zeros(dim))
            self.v_bias = nn.Parameter(torch.zeros(dim))
        else:
            self.q_bias = self.v_bias = None
        self.proj = _Linear(dim, dim)

    def cpb_params(self):
        return [self.cpb_mlp[0].weight, self.cpb_mlp[0].bias, self.cpb_mlp[2].weight, self.logit_scale]


class SwinTransformerBlock(nn.Module):
    """x = x + BN(attn(x)); x = x + BN(mlp(x))  (reference :183-300, shift_size 0): parameter holder."""

    def __init__(self, dim, dim_out, heads, window_size=7, shift_size=0, qkv_bias=True, drop=0.0, attn_drop=0.0,
                 drop_path=0.0, norm_layer=None, pretrained_window_size=0, activation=None):
        super().__init__()
        if shift_size != 0:
            raise NotImplementedError("shifted windows: the reference itself cannot run them (SURVEY.md F8)")
        self.dim, self.num_heads, self.window_size, self.shift_size = dim, heads, window_size, shift_size
        self.attn = WindowAttention(dim, (window_size, window_size), heads, qkv_bias=qkv_bias)
        self.norm2 = _BN(dim)
        self.mlp = Mlp(in_features=dim, out_features=dim_out, hidden_features=dim * 4)
        self.norm3 = _BN(dim)
        self.register_buffer("attn_mask", None)


# ------------------------------------------------------------------------------------------------- block math
def _w2d(conv_or_lin, dt):
    w = conv_or_lin.weight.data
    return ops.cast_from_f32(w.reshape(w.shape[0], -1).contiguous(), dt)


def prepare_linear_weights(mods, dt):
    """{module: (w [N,K] in dt, w^T [K,N] in dt)} for every Linear / 1x1 weight of a step in ONE launch (the batched
    conv-operand kernel: a Linear is a 1x1 convolution) instead of a cast per forward use and a transpose per backward use"""
    mods = list(mods)
    ws = [m.weight.data.reshape(m.weight.shape[0], 1, 1, -1) for m in mods]
    outs = ops.prep_conv_weights(ws, dt)
    return {m: (wc.view(wc.shape[0], wc.shape[3]), wt.view(wt.shape[0], wt.shape[3])) for m, (wc, wt) in zip(mods, outs)}


def _lin_operands(mod, dt, wprep):
    if wprep is not None and mod in wprep:
        return wprep[mod]
    return _w2d(mod, dt), None


def _transposed(w2d, wt):
    return wt if wt is not None else ops.transpose2d(w2d)


def _colsum_via_gemm(x2d, out_accum):
    """out_accum[c] += sum_rows x2d[:, c] for rows wider than the element-wise reducer handles (3C up to 1536)"""
    ones = torch.ones((x2d.shape[0], 8), dtype=x2d.dtype, device=x2d.device)
    tmp = torch.zeros((x2d.shape[1], 8), dtype=torch.float32, device=x2d.device)
    ops.gemm_tn(x2d, ones, tmp)
    out_accum += tmp[:, 0]


class CpbBatch:
    """Position-bias tables of a group of attention blocks by the libfrhip kernels (csrc/cpb.hip): ONE launch computes
    16*sigmoid(cpb_mlp(coords)[index]) and exp(min(logit_scale, ln 100)) of every block, ONE launch (queued at the end of the
    backward pass) turns all d(bias) / d(scale) into the gradients of cpb_mlp.* and logit_scale.  Stands where torch ran
    ~8 forward and ~12 backward launches per block (tiny rocBLAS GEMMs + element-wise kernels, nets/SwinV2.py:150-158)."""

    _DT = np.dtype([(k, "<u8") for k in ("coords", "index", "w0", "b0", "w2", "ls", "bias", "scale", "dbias", "dscale",
                                         "dw0", "db0", "dw2", "dls")] + [(k, "<i4") for k in ("entries", "tokens", "heads", "pad")])
    _PINS, _TURN = None, 0

    def __init__(self, blocks, device):
        from frhip._abi import check, lib
        self.blocks = list(blocks)
        self.check, self.lib = check, lib
        import ctypes
        lim = [ctypes.c_int(0) for _ in range(3)]             # the kernels' compile-time limits (LDS arrays, grid.y = max heads)
        check(lib().frhip_cpb_limits(*[ctypes.byref(v) for v in lim]), "frhip_cpb_limits")
        max_entries, max_heads, hidden = (v.value for v in lim)
        sizes = []
        for blk in self.blocks:
            at = blk.attn
            n = at.window_size[0] * at.window_size[1]
            entries = (2 * at.window_size[0] - 1) * (2 * at.window_size[1] - 1)
            if entries > max_entries or at.num_heads > max_heads or at.cpb_mlp[0].out_features != hidden or n > 49:
                raise ValueError("CpbBatch: window %s / %d heads / hidden %d exceed the position-bias kernels' limits "
                                 "(%d table entries, %d heads, hidden width %d, 49 tokens)"
                                 % (tuple(at.window_size), at.num_heads, at.cpb_mlp[0].out_features, max_entries, max_heads, hidden))
            idx, tab = at.relative_position_index, at.relative_coords_table
            if idx.dtype != torch.int64 or not idx.is_contiguous() or tab.dtype != torch.float32 or not tab.is_contiguous():
                raise ValueError("CpbBatch: relative_position_index must be contiguous int64 and relative_coords_table contiguous fp32")
            sizes.append((at.num_heads * n * n, at.num_heads))
        pad4 = lambda v: (v + 3) // 4 * 4                     # every view starts 16-byte aligned
        total = sum(pad4(a) + pad4(b) for a, b in sizes)
        self.out = torch.empty(total, dtype=torch.float32, device=device)            # bias / scale of every block
        self.grad = torch.zeros(total, dtype=torch.float32, device=device)           # d bias / d scale (the attention backward adds into them)
        self.views, off = [], 0
        for (nb, nh), blk in zip(sizes, self.blocks):
            at = blk.attn
            n = at.window_size[0] * at.window_size[1]
            o2 = off + pad4(nb)
            self.views.append((self.out[off:off + nb].view(at.num_heads, n, n), self.out[o2:o2 + nh],
                               self.grad[off:off + nb].view(at.num_heads, n, n), self.grad[o2:o2 + nh]))
            off = o2 + pad4(nh)
        self.pending = False
        self._launch(False, None)

    @classmethod
    def _pinned(cls, nbytes):
        if cls._PINS is None or cls._PINS.shape[1] < nbytes:
            cls._PINS = torch.empty((8, max(nbytes, 8192)), dtype=torch.uint8, pin_memory=True)
        cls._TURN += 1
        return cls._PINS[cls._TURN % 8, :nbytes]

    def _launch(self, backward, G):
        tab = np.zeros(len(self.blocks), dtype=self._DT)
        for i, blk in enumerate(self.blocks):
            at = blk.attn
            w0, b0, w2, ls = at.cpb_params()
            bias, scale, dbias, dscale = self.views[i]
            t = tab[i]
            t["coords"], t["index"] = at.relative_coords_table.data_ptr(), at.relative_position_index.data_ptr()
            t["w0"], t["b0"], t["w2"], t["ls"] = w0.data_ptr(), b0.data_ptr(), w2.data_ptr(), ls.data_ptr()
            t["bias"], t["scale"], t["dbias"], t["dscale"] = bias.data_ptr(), scale.data_ptr(), dbias.data_ptr(), dscale.data_ptr()
            if backward:
                t["dw0"], t["db0"], t["dw2"], t["dls"] = (G(w0).data_ptr(), G(b0).data_ptr(), G(w2).data_ptr(), G(ls).data_ptr())
            t["entries"], t["tokens"], t["heads"] = (at.relative_coords_table.numel() // 2,
                                                     at.window_size[0] * at.window_size[1], at.num_heads)
        raw = torch.from_numpy(tab.view(np.uint8))
        pin = self._pinned(raw.numel())
        pin.copy_(raw)
        dev = torch.empty(raw.numel(), dtype=torch.uint8, device=self.out.device)
        dev.copy_(pin, non_blocking=True)
        fn = self.lib().frhip_cpb_bwd if backward else self.lib().frhip_cpb_fwd
        scratch = torch.empty(self.lib().frhip_cpb_scratch_floats(len(self.blocks)), dtype=torch.float32, device=self.out.device)
        self.check(fn(dev.data_ptr(), len(self.blocks), scratch.data_ptr(), ops._s()), "frhip_cpb_bwd" if backward else "frhip_cpb_fwd")
        self._table = dev                                    # alive until the kernel has run (stream-ordered free)

    def flush_backward(self, bc):
        """every block's d(bias), d(scale) -> parameter gradients, one launch (registered to run before bc.join())"""
        if self.pending:
            self.pending = False
            for blk in self.blocks:
                for p in blk.attn.cpb_params():
                    if not p.data.is_contiguous() or not bc.G(p).is_contiguous():
                        raise RuntimeError("frhip: the position-bias parameters and their gradients must be contiguous")
            self._launch(True, bc.G)


def precompute_position_bias(attn_blocks, device):
    """all position-bias tables of the step in one launch (CpbBatch); each block picks its views up in position_bias()"""
    attn_blocks = list(attn_blocks)
    if not attn_blocks:
        return None
    batch = CpbBatch(attn_blocks, device)
    for i, blk in enumerate(attn_blocks):
        blk._cpb_batch = (batch, i)
    return batch


def position_bias(blk):
    """(batch, index, bias [heads,n,n], scale [heads], d bias, d scale) of one block; computed on the spot when the block was
    not part of a precompute_position_bias() group (blocks driven one by one, e.g. from a test)"""
    hit = getattr(blk, "_cpb_batch", None)
    if hit is None:
        hit = (CpbBatch([blk], blk.attn.logit_scale.device), 0)
    blk._cpb_batch = None
    batch, i = hit
    bias, scale, dbias, dscale = batch.views[i]
    return batch, i, bias, scale, dbias, dscale


def position_bias_backward(blk, s, bc):
    """the block's d(bias) / d(scale) sit in the batch's gradient arena (the attention backward kernel added them there):
    queue the batch's one backward launch behind the whole backward pass"""
    batch = s.cpb_batch
    if not batch.pending:
        batch.pending = True
        bc.before_join.append(lambda: batch.flush_backward(bc))


def qkv_bias(at):
    """[q_bias, 0, v_bias] (reference nets/SwinV2.py:141-143 concatenates it in every forward) without a launch: q_bias and v_bias are
    kept as views of ONE 3C buffer whose middle third stays zero, so whatever updates the parameters in place (optimizer,
    load_state_dict, broadcast) updates the buffer.  Re-established when something replaced the parameters' storage (.cuda(), .to())."""
    c = at.q_bias.numel()
    qb = getattr(at, "_qkv_bias_buf", None)
    if (qb is None or qb.device != at.q_bias.device or qb.dtype != at.q_bias.dtype or at.q_bias.data_ptr() != qb.data_ptr()
            or at.v_bias.data_ptr() != qb.data_ptr() + 2 * c * qb.element_size()):
        qb = torch.zeros(3 * c, dtype=at.q_bias.dtype, device=at.q_bias.device)
        qb[:c].copy_(at.q_bias.data)
        qb[2 * c:].copy_(at.v_bias.data)
        at.q_bias.data, at.v_bias.data = qb[:c], qb[2 * c:]
        at._qkv_bias_buf = qb
    return qb


def swin_block_forward(blk, x, dt, training, save, wprep=None):
    b, h, w, c = x.shape
    m = b * h * w
    at = blk.attn
    x2 = x.view(m, c)
    wqkv, wqkv_t = _lin_operands(at.qkv, dt, wprep)
    qb = qkv_bias(at)
    qkv, _, _ = ops.linear_fwd(x2, wqkv, qb)                                  # bias add in the GEMM epilogue
    cpb_batch, _, bias, scale, dbias_buf, dscale_buf = position_bias(blk)
    ao = ops.winattn_fwd(qkv, bias, scale, b, h, w, at.num_heads)
    wproj, wproj_t = _lin_operands(at.proj, dt, wprep)
    po, _, part2 = ops.linear_fwd(ao, wproj, at.proj.bias.data, want_stats=training)   # + norm2's batch statistics
    st2 = bn_forward_state(blk.norm2, part2, m, training)
    x1 = ops.bn_apply(po, st2, res=x2)
    w1, w1_t = _lin_operands(blk.mlp.fc1, dt, wprep)
    hid, act, _ = ops.linear_fwd(x1, w1, blk.mlp.fc1.bias.data, want_act=True)          # bias + GELU, both tensors kept
    w2, w2_t = _lin_operands(blk.mlp.fc2, dt, wprep)
    mo, _, part3 = ops.linear_fwd(act, w2, blk.mlp.fc2.bias.data, want_stats=training)
    st3 = bn_forward_state(blk.norm3, part3, m, training)
    out = ops.bn_apply(mo, st3, res=x1).view(b, h, w, c)
    s = None
    if save:
        s = Saved()
        (s.x2, s.wqkv, s.qkv, s.cpb_batch, s.dbias, s.dscale, s.bias, s.scale, s.ao, s.wproj, s.po, s.st2, s.x1, s.w1, s.hid,
         s.act, s.w2, s.mo, s.st3, s.shape) = (x2, wqkv, qkv, cpb_batch, dbias_buf, dscale_buf, bias, scale, ao, wproj, po, st2,
                                                x1, w1, hid, act, w2, mo, st3, (b, h, w, c))
        s.wqkv_t, s.wproj_t, s.w1_t, s.w2_t = wqkv_t, wproj_t, w1_t, w2_t
    return out, s


# Swin34, same box, three alternating runs: 14.98 ms with one hand-over per weight gradient, 15.29 ms with pairs (fc2's weight gradient then starts a
# whole fc2 data-gradient later and the side queue ends the backward pass behind): off here.  The AlterNet attention blocks (two weight
# gradients, no MLP) have their own switch (nets.AlterNet_SwinV2_FAN._PAIR_HANDOVER: 11.79 against 12.18 ms with pairs, on).
_PAIR_HANDOVER = os.environ.get("FRHIP_PAIR_HANDOVER", "0") == "1"


def _dgrad_add(dy2d, w2d, residual2d, wt=None, bnred=None):
    """dy [M,K] @ w [K,C] + residual [M,C], as a 1x1 data-gradient with the residual add fused.
    bnred=(y [.., C], BN state, relu): also the BN-backward partial sums of the result against y -> (dx, partial)"""
    m, k = dy2d.shape
    c = w2d.shape[1]
    wt = _transposed(w2d, wt)                       # [C][K]: K-contiguous rows of the transposed weight
    if bnred is not None:
        dx, part = ops.conv_dgrad(dy2d.view(m, 1, 1, k), wt.view(c, 1, 1, k), (m, 1, 1, c), 1, 1, 1, 0,
                                  residual=residual2d.view(m, 1, 1, c), bnred=(bnred[0].view(m, 1, 1, c),) + tuple(bnred[1:]))
        return dx.view(m, c), part
    return ops.conv_dgrad(dy2d.view(m, 1, 1, k), wt.view(c, 1, 1, k), (m, 1, 1, c), 1, 1, 1, 0,
                          residual=residual2d.view(m, 1, 1, c)).view(m, c)


def swin_block_backward(blk, s, dout, dt, bc, next_bn=None, part3=None):
    """next_bn=(y, st, relu): the BatchNorm whose upstream gradient the returned dx is (norm3 of the previous block, or the
    stem's for the first block): its backward reduction rides in the last data-gradient and (dx, partial) is returned.
    part3: the partial sums of THIS block's norm3 backward when the producer of dout already reduced them."""
    G = bc.G
    b, h, w, c = s.shape
    m = b * h * w
    at = blk.attn
    d2 = dout.reshape(m, c)
    # ---- MLP branch: x2 = x1 + BN(fc2(gelu(fc1(x1))))
    dmo = ops.bn_backward(d2, s.mo, s.st3, blk.norm3.weight.data, G(blk.norm3.weight), G(blk.norm3.bias), part=part3)
    # fc2.bias (and proj.bias below) only shift the input of a training-mode BatchNorm: their gradient, the column sums of
    # that BatchNorm's input gradient, is analytically zero (sum_rows dy = gamma * invstd * (sum d - N mean(d) - mean(d xhat)
    # * sum xhat) = 0); the reference gets 1e-8-sized round-off there.  Left at the arena's zero: no reduction pass.
    # Hand-overs to the side stream cost the MAIN queue ~6.5 us each (the event record in front of the next launch: timeline, 45 per Swin34
    # step); handing the weight gradients over in PAIRS (_PAIR_HANDOVER) was measured slower here all the same, see above
    if not _PAIR_HANDOVER:
        bc.on_side(lambda: ops.gemm_tn(dmo, s.act, G(blk.mlp.fc2.weight).view(c, 4 * c)), dmo, s.act)
    # [M, 4C]: fc2's data-gradient with gelu'(hid) and fc1.bias's gradient (column sums) fused into its epilogue
    dhid, _ = ops.linear_dgrad_gelu(dmo, _transposed(s.w2, s.w2_t), s.hid, colsum_into=G(blk.mlp.fc1.bias))
    if _PAIR_HANDOVER:
        bc.on_side(lambda: (ops.gemm_tn(dmo, s.act, G(blk.mlp.fc2.weight).view(c, 4 * c)),
                            ops.gemm_tn(dhid, s.x1, G(blk.mlp.fc1.weight).view(4 * c, c))), dmo, s.act, dhid, s.x1)
    else:
        bc.on_side(lambda: ops.gemm_tn(dhid, s.x1, G(blk.mlp.fc1.weight).view(4 * c, c)), dhid, s.x1)
    # dx1 is the upstream gradient of norm2: its backward reduction over (dx1, po) rides in this data-gradient's epilogue
    dx1, part2 = _dgrad_add(dhid, s.w1, d2, s.w1_t, bnred=(s.po, s.st2, False))
    # ---- attention branch: x1 = x + BN(proj(attn(qkv(x))))
    dpo = ops.bn_backward(dx1, s.po, s.st2, blk.norm2.weight.data, G(blk.norm2.weight), G(blk.norm2.bias), part=part2)
    if not _PAIR_HANDOVER:
        bc.on_side(lambda: ops.gemm_tn(dpo, s.ao, G(at.proj.weight)), dpo, s.ao)
    dao = ops.gemm_nt(dpo, _transposed(s.wproj, s.wproj_t))
    dqkv, _, _, gsum = ops.winattn_bwd(s.qkv, dao, s.bias, s.scale, b, h, w, at.num_heads, want_colsum=True,
                                       dbias=s.dbias, dscale=s.dscale,      # d(bias), d(scale) accumulate in the batch's arena
                                       qv_grads=(G(at.q_bias), G(at.v_bias)))
    if gsum is None:                                 # fp32 validation kernels: column sums by a ones-GEMM
        gsum = torch.zeros(3 * c, dtype=torch.float32, device=dout.device)
        _colsum_via_gemm(dqkv, gsum)
    if gsum is not True:                             # bf16 MFMA kernel: already added into the two gradient accumulators
        G(at.q_bias).add_(gsum[:c])
        G(at.v_bias).add_(gsum[2 * c:])
    if _PAIR_HANDOVER:
        bc.on_side(lambda: (ops.gemm_tn(dpo, s.ao, G(at.proj.weight)), ops.gemm_tn(dqkv, s.x2, G(at.qkv.weight))), dpo, s.ao, dqkv, s.x2)
    else:
        bc.on_side(lambda: ops.gemm_tn(dqkv, s.x2, G(at.qkv.weight)), dqkv, s.x2)
    part = None
    if next_bn is not None:
        dx, part = _dgrad_add(dqkv, s.wqkv, dx1, s.wqkv_t, bnred=next_bn)
    else:
        dx = _dgrad_add(dqkv, s.wqkv, dx1, s.wqkv_t)
    # ---- the 169-entry position-bias MLP and the logit scale: one batched kernel launch at the end of the backward pass
    position_bias_backward(blk, s, bc)
    return dx.view(b, h, w, c) if next_bn is None else (dx.view(b, h, w, c), part)


# ------------------------------------------------------------------------------------------------- network
class Swin(nn.Module):
    def __init__(self, conf, block, block2, num_blocks, heads):
        super().__init__()
        self.emd_size = conf.emd_size
        self.dtype = compute_dtype(conf)
        self.inplanes = 64
        self.conv1 = _Conv(3, 64, 3, 1)
        self.bn1 = _BN(64)
        self.layer1 = self.stack_layers(block, block2, 64, num_blocks[0], heads[0])
        self.layer2 = self.stack_layers(block, block2, 128, num_blocks[1], heads[1], stride=2)
        self.layer3 = self.stack_layers(block, block2, 256, num_blocks[2], heads[2], stride=2)
        self.layer4 = self.stack_layers(block, block2, conf.emd_size, num_blocks[3], heads[3], stride=2)
        self.bn2 = _BN(block.expansion * conf.emd_size)
        self.dropout = nn.Dropout()
        self.fc = _Linear(block.expansion * conf.emd_size * 7 * 7, conf.emd_size)
        self.bn3 = _BN(conf.emd_size)
        for m in self.modules():             # same initialisation rule as the reference (:518-532)
            if isinstance(m, (_Conv, _Linear, nn.Linear)):
                nn.init.xavier_normal_(m.weight)
                if getattr(m, "bias", None) is not None:
                    nn.init.constant_(m.bias, 0)

    def stack_layers(self, block, block2, planes, blocks, heads, stride=1):
        layers = []
        if stride > 1:
            layers.append(_Conv(self.inplanes, planes, 2, 2, pad=0))
        self.inplanes = planes * block.expansion
        for _ in range(blocks):
            layers.append(block2(self.inplanes, planes, heads=heads))
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
        wprep = None
        blocks = [m for m in self._layers() if not isinstance(m, _Conv)]
        precompute_position_bias(blocks, x.device)         # every block's bias table and logit scale: one launch
        if training and save:
            wprep = prepare_linear_weights([l for b in blocks for l in (b.attn.qkv, b.attn.proj, b.mlp.fc1, b.mlp.fc2)], dt)
        cur = stem_forward(self, x, training, sv)
        saved = []
        for mod in self._layers():
            if isinstance(mod, _Conv):           # 2x2 / stride-2 stage convolution, no norm (reference :545-546)
                wd = ops.cast_from_f32(mod.physical(), dt)
                nxt, _ = ops.conv_fwd(cur, wd, 2, 0, want_stats=False)
                saved.append(cur if save else None)
                cur = nxt
            else:
                cur, s = swin_block_forward(mod, cur, dt, training, save, wprep)
                saved.append(s)
        if cur.shape[1] != 7 or cur.shape[2] != 7:
            raise NotImplementedError("AdaptiveAvgPool2d((7,7)) is the identity only for 112x112 inputs")
        mask = None
        if training and self.dropout.p > 0:
            keep = 1.0 - self.dropout.p
            mask = ops.dropout_mask(cur.shape, cur.dtype, keep, cur.device)
        emb = tail_forward(self, cur, training, sv, dropout_mask=mask)
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
            if isinstance(mod, _Conv):
                wt = ops.pack_wt(mod.physical(), dt)
                bc.wgrad(dout, s, phys_grad(bc.G(mod.weight)), 2, 2, 2, 0)
                # the first stage convolution's data gradient IS the gradient of the stem's pooled map: the stem's BatchNorm-backward
                # sums ride in its epilogue (stem_reduction_operands) instead of a recompute pass over the input (0.5 ms at B = 512)
                nxt = stem_reduction_operands(self, sv) if i == 0 else None
                if nxt is not None:
                    dout, part = ops.conv_dgrad(dout, wt, s.shape, 2, 2, 2, 0, bnred=nxt)
                else:
                    dout, part = ops.conv_dgrad(dout, wt, s.shape, 2, 2, 2, 0), None
                continue
            # the gradient leaving block i enters norm3 of block i-1 (or, for the first block, the stem's pool / ReLU / BN):
            # that BatchNorm's backward sums ride in block i's last kernel
            if i == 0:
                nxt = stem_reduction_operands(self, sv)
            elif not isinstance(layers[i - 1], _Conv):
                nxt = (sv.layers[i - 1].mo, sv.layers[i - 1].st3, False)
            else:
                nxt = None
            res = swin_block_backward(mod, s, dout, dt, bc, next_bn=nxt, part3=part)
            dout, part = res if nxt is not None else (res, None)
            if len(layers) - 1 - i == DEFER_EARLY_BLOCKS:
                bc.run_deferred()                          # the head's early parameter update: beside the blocks, not beside the tail
        stem_backward(self, sv, dout, bc, part)
        return bc.join()


def _make(conf, blocks, **kw):
    return Swin(conf, BasicBlock, SwinTransformerBlock, num_blocks=blocks, heads=(2, 4, 8, 16), **kw)


def Swin18(conf, **kwargs):
    return _make(conf, [0, 1, 1, 1], **kwargs)


def Swin34(conf, **kwargs):
    return _make(conf, [0, 0, 4, 6], **kwargs)


def Swin50(conf, **kwargs):
    return _make(conf, [0, 0, 4, 10], **kwargs)


def Swin100(conf, **kwargs):
    return _make(conf, [0, 0, 6, 14], **kwargs)


def Swin200(conf, **kwargs):
    return _make(conf, [0, 0, 6, 30], **kwargs)


def Encoder(conf):
    """Name dispatch of the reference (:645-656: Swin200/100/50/34) -- plus 'Swin18', whose constructor exists there
    but which its dispatcher forgets."""
    table = {"Swin200": Swin200, "Swin100": Swin100, "Swin50": Swin50, "Swin34": Swin34, "Swin18": Swin18}
    if conf.network in table:
        return table[conf.network](conf)
    return None
