"""Drop-in for the reference `nets/ArcFace.py` (margin modules).

`ArcFace(s, margin)`, `CosFace(s, m)` and `CombinedMarginLoss(s, m1, m2, m3)` keep the reference's constructors,
attribute names and `forward(logits, labels)` contract (/root/reference/nets/ArcFace.py:5-106): `logits` [N, C] fp32
cosines are modified IN PLACE at the target entries (labels [N,1] or [N] int64, -1 = no target on this shard) and the
scaled tensor is returned.  On the MI355X the work is one HIP row kernel (frhip_margin_fwd / _bwd) wrapped in an
autograd node.

Inside PartialFC the margin is not applied through this `forward`: it lives in the epilogue of the fused cos-theta
MFMA kernel (frhip_head_fwd), which only reads `.scale` / `.margin` / `.kind` from the module, so the [N, C] logits
this `forward` would overwrite never exist in HBM on the training hot path.
"""
import math

import torch


class _MarginFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, s, m, kind):
        from frhip import ops
        if not logits.is_cuda:
            raise RuntimeError("nets.ArcFace (frhip): logits must live on the MI355X; there is no CPU path")
        lab = labels.reshape(-1).long().contiguous()
        buf = logits if (logits.is_contiguous() and logits.dtype == torch.float32) else logits.float().contiguous()
        tsave = ops.margin_fwd(buf, lab, float(s), float(m), kind)
        ctx.mark_dirty(logits) if buf is logits else None
        ctx.save_for_backward(lab, tsave)
        ctx.s, ctx.m, ctx.kind = float(s), float(m), kind
        return buf

    @staticmethod
    def backward(ctx, g):
        from frhip import ops
        lab, tsave = ctx.saved_tensors
        return ops.margin_bwd(g.contiguous().float(), lab, tsave, ctx.s, ctx.m, ctx.kind), None, None, None, None


class ArcFace(torch.nn.Module):
    """Additive angular margin: target logit cos(theta) -> cos(theta + m), everything x s."""
    kind = "arcface"

    def __init__(self, s=64.0, margin=0.5):
        super().__init__()
        self.scale = s
        self.margin = margin
        self.cos_m = math.cos(margin)
        self.sin_m = math.sin(margin)
        self.theta = math.cos(math.pi - margin)
        self.sinmm = math.sin(math.pi - margin) * margin
        self.easy_margin = False

    def forward(self, logits: torch.Tensor, labels: torch.Tensor):
        return _MarginFn.apply(logits, labels, self.scale, self.margin, 0)


class CosFace(torch.nn.Module):
    """Additive cosine margin: target logit t -> t - m, everything x s (reference :94-106)."""
    kind = "cosface"

    def __init__(self, s=64.0, m=0.40):
        super().__init__()
        self.s = s
        self.m = m
        self.scale, self.margin = s, m

    def forward(self, logits: torch.Tensor, labels: torch.Tensor):
        return _MarginFn.apply(logits, labels, self.s, self.m, 1)


class CombinedMarginLoss(torch.nn.Module):
    """(m1, m2, m3) front-end of the reference (:5-61): m1 == 1, m3 == 0 is ArcFace with margin m2; m3 > 0 is CosFace
    with margin m3; anything else raises exactly like the reference.  interclass_filtering_threshold > 0 is not built."""

    def __init__(self, s, m1, m2, m3, interclass_filtering_threshold=0):
        super().__init__()
        self.s, self.m1, self.m2, self.m3 = s, m1, m2, m3
        self.interclass_filtering_threshold = interclass_filtering_threshold
        self.cos_m, self.sin_m = math.cos(m2), math.sin(m2)
        self.theta = math.cos(math.pi - m2)
        self.sinmm = math.sin(math.pi - m2) * m2
        self.easy_margin = False

    def forward(self, logits, labels):
        if self.interclass_filtering_threshold > 0:
            raise NotImplementedError("interclass filtering is not built in the HIP path")
        if self.m1 == 1.0 and self.m3 == 0.0:
            return _MarginFn.apply(logits, labels, self.s, self.m2, 0)
        if self.m3 > 0:
            return _MarginFn.apply(logits, labels, self.s, self.m3, 1)
        raise
