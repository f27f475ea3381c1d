"""CPU restatement of one optimisation step (TEST INFRASTRUCTURE, see oracle/__init__.py).

Composition follows /root/reference/model/FR_PartialFC.py:162-193 (non-mixed branch):
  opt.zero_grad -> encoder.train() -> feat = F.normalize(encoder(img)) (:171)
  -> loss = PartialFC(feat, id, opt) (:175) -> loss.backward() (:186)
  -> clip_grad_norm_(encoder.parameters(), 5) (:187) -> opt.step() (:188)
with ONE optimizer over param groups [encoder, head] (:434-449, SGD momentum + weight decay
on every tensor, BN affine included).  world_size == 1 here (BASELINE cfg 1 / cfg 2).
The SGD update itself is restated by hand (torch.optim.SGD semantics: d = g + wd*p;
buf = d on the first step else mom*buf + d; p -= lr*buf).
"""
import torch
import torch.nn.functional as F

from . import head_ref, resnet_ref


class SGDState:
    def __init__(self, lr, momentum=0.9, weight_decay=5e-4):
        self.lr, self.momentum, self.wd = lr, momentum, weight_decay
        self.buf = {}

    def apply(self, name, p, g):
        d = g + self.wd * p
        if name in self.buf:
            self.buf[name].mul_(self.momentum).add_(d)
        else:
            self.buf[name] = d.clone()
        p.sub_(self.lr * self.buf[name])


def clip_coef(grads, max_norm=5.0):
    """torch.nn.utils.clip_grad_norm_: total L2 over all grads, coef = max_norm/(total+1e-6) capped at 1."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    return torch.clamp(max_norm / (total + 1e-6), max=1.0), total


def train_step(sd, head_w, img, ids, blocks, num_classes, opt, s=30.0, m=0.35,
               emd_size=512, sample_rate=1.0, uniforms=None, head_state=None):
    """One step at world_size 1.  Mutates sd (params + BN stats), head_w and opt in place.
    With sample_rate < 1 the activated rows carry their own momentum rows
    (nets/PartialFC.py:120-129, :142-143), kept in opt.buf['head'] as a full [num_local,D] table.
    Returns dict(loss, feat, grad_norm)."""
    names = resnet_ref.trainable_names(sd)
    leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in names}
    work = dict(sd)
    work.update(leaves)
    raw = resnet_ref.resnet_forward(work, img, blocks, True, emd_size)
    for k in sd:   # running stats / counters were updated on `work`
        if k not in leaves:
            sd[k] = work[k]
    feat = F.normalize(raw)
    h = head_ref.head_all_shards([feat.detach()], [ids], [head_w], num_classes, s, m,
                                 sample_rate=sample_rate, uniforms=uniforms)
    feat.backward(h["d_emb"][0])
    grads = [leaves[k].grad for k in names]
    coef, total = clip_coef(grads)
    with torch.no_grad():
        for k in names:
            opt.apply(k, sd[k], leaves[k].grad * coef)
        # head rows (not clipped: clip is over encoder.parameters() only, :187)
        idx = h["index"][0]
        if "head" not in opt.buf:
            opt.buf["head"] = torch.zeros_like(head_w)
            first = True
        else:
            first = False
        w_rows = head_w[idx]
        d = h["d_w_act"][0] + opt.wd * w_rows
        # torch SGD initialises the buffer to d only when the state is EMPTY; PartialFC with
        # sample_rate<1 always installs a momentum tensor (zeros at first), so the general
        # formula mom*buf + d applies there; at sample_rate==1 the first step sets buf = d.
        if sample_rate < 1 or not first:
            buf_rows = opt.buf["head"][idx] * opt.momentum + d
        else:
            buf_rows = d
        opt.buf["head"][idx] = buf_rows
        head_w[idx] = w_rows - opt.lr * buf_rows
    return dict(loss=h["loss"], feat=feat.detach(), grad_norm=total, index=idx)
