"""Drop-in for the reference `nets/ArcFace.py` (margin hyper-parameter holders).

`ArcFace(s, margin)` keeps the reference's constructor and attribute names
(/root/reference/nets/ArcFace.py:63-72).  Inside the PartialFC head the margin is applied in the epilogue of
the fused cos-theta MFMA kernel (frhip_head_fwd), so the logits this module's reference `forward` would
overwrite in place never exist in HBM; PartialFC only reads `.scale` / `.margin` from it.
"""
import math

import torch


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
