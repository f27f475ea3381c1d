"""CPU restatement of one optimisation step (TEST INFRASTRUCTURE, see oracle/__init__.py).

Composition follows /root/reference/model/FR_PartialFC.py:162-193 (non-mixed branch):
  opt.zero_grad -> encoder.train() -> feat = F.normalize(encoder(img)) (:171)
  -> loss = PartialFC(feat, id, opt) (:175) -> loss.backward() (:186)
  -> clip_grad_norm_(encoder.parameters(), 5) (:187) -> opt.step() (:188)
with ONE optimizer over param groups [encoder, head] (:434-449, SGD momentum + weight decay
on every tensor, BN affine included).  train_step: world_size 1 (BASELINE cfg 1 / cfg 2); train_step_ranks: a world of N
ranks in one process (DDP-averaged backbone gradients, class-sharded head), pinned by the ws-2 reference fixtures.
The updates are restated by hand: SGDState (torch.optim.SGD semantics: d = g + wd*p;
buf = d on the first step else mom*buf + d; p -= lr*buf) and AdamWState (torch.optim.AdamW, the reference's shipped recipe main/train.sh:12).
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

    def head_update(self, head_w, idx, d_w, sample_rate):
        """rows `idx` of the class-centre shard (not clipped: the clip is over encoder.parameters() only, :187)"""
        first = "head" not in self.buf
        if first:
            self.buf["head"] = torch.zeros_like(head_w)
        w_rows = head_w[idx]
        d = d_w + self.wd * w_rows
        # torch SGD initialises the buffer to d only when the state is EMPTY; PartialFC with
        # sample_rate<1 always installs a momentum tensor (zeros at first), so the general
        # formula mom*buf + d applies there; at sample_rate==1 the first step sets buf = d.
        if sample_rate < 1 or not first:
            buf_rows = self.buf["head"][idx] * self.momentum + d
        else:
            buf_rows = d
        self.buf["head"][idx] = buf_rows
        head_w[idx] = w_rows - self.lr * buf_rows


class AdamWState:
    """torch.optim.AdamW (decoupled weight decay; reference model/FR_PartialFC.py:436-442, configs/ms1m_arcface_122.py:222-224),
    restated by hand:  p *= 1 - lr*wd;  m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;  t += 1;
    p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).
    Head rows (PartialFCAdamW, nets/PartialFC.py:309-327, :337-342): exp_avg / exp_avg_sq live in full [num_local, D] tables and travel
    with the sampled rows.  With sampling, sample() counts calls in self.step and writes that count into optimizer.state['step'] BEFORE
    the optimizer bumps it, so the head's bias corrections use t = (#sample calls so far) + 1 -- one ahead of the encoder's."""

    def __init__(self, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-4):
        self.lr, self.b1, self.b2, self.eps, self.wd = lr, float(betas[0]), float(betas[1]), eps, weight_decay
        self.m, self.v, self.t = {}, {}, {}
        self.head_calls = 0

    def _adam(self, p, g, m, v, t):
        p.mul_(1.0 - self.lr * self.wd)
        m.mul_(self.b1).add_(g, alpha=1.0 - self.b1)
        v.mul_(self.b2).addcmul_(g, g, value=1.0 - self.b2)
        bc1, bc2 = 1.0 - self.b1 ** t, 1.0 - self.b2 ** t
        denom = (v.sqrt() / (bc2 ** 0.5)).add_(self.eps)
        p.addcdiv_(m, denom, value=-self.lr / bc1)

    def apply(self, name, p, g):
        if name not in self.m:
            self.m[name], self.v[name], self.t[name] = torch.zeros_like(p), torch.zeros_like(p), 0
        self.t[name] += 1
        self._adam(p, g, self.m[name], self.v[name], self.t[name])

    def head_update(self, head_w, idx, d_w, sample_rate):
        if "head" not in self.m:
            self.m["head"], self.v["head"] = torch.zeros_like(head_w), torch.zeros_like(head_w)
        self.head_calls += 1
        t = self.head_calls + 1 if sample_rate < 1 else self.head_calls
        p, m, v = head_w[idx], self.m["head"][idx], self.v["head"][idx]
        self._adam(p, d_w, m, v, t)
        head_w[idx], self.m["head"][idx], self.v["head"][idx] = p, m, v


def clip_coef(grads, max_norm=5.0):
    """torch.nn.utils.clip_grad_norm_: total L2 over all grads, coef = max_norm/(total+1e-6) capped at 1."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    return torch.clamp(max_norm / (total + 1e-6), max=1.0), total


def train_step(sd, head_w, img, ids, blocks, num_classes, opt, s=30.0, m=0.35,
               emd_size=512, sample_rate=1.0, uniforms=None, head_state=None, forward=None, names=None):
    """One step at world_size 1.  Mutates sd (params + BN stats), head_w and opt in place.
    With sample_rate < 1 the activated rows carry their own momentum rows
    (nets/PartialFC.py:120-129, :142-143), kept in opt.buf['head'] as a full [num_local,D] table.
    Returns dict(loss, feat, grad_norm)."""
    # forward(work, img) -> embeddings and names = trainable keys: another backbone's functional forward (oracle.swin_ref / alternet_ref)
    names = resnet_ref.trainable_names(sd) if names is None else list(names)
    leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in names}
    work = dict(sd)
    work.update(leaves)
    raw = resnet_ref.resnet_forward(work, img, blocks, True, emd_size) if forward is None else forward(work, img)
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
        idx = h["index"][0]
        opt.head_update(head_w, idx, h["d_w_act"][0], sample_rate)
    return dict(loss=h["loss"], feat=feat.detach(), grad_norm=total, index=idx, grads={k: leaves[k].grad * coef for k in names})


def train_step_ranks(sds, head_ws, imgs, idss, blocks, num_classes, opts, s=30.0, m=0.35, emd_size=512,
                     sample_rate=1.0, uniforms=None):
    """One step of a world of len(sds) ranks (reference composition model/FR_PartialFC.py:98, :162-193):
    every rank runs its own backbone on its own images (BatchNorm statistics and buffers stay per rank: the reference wraps the
    encoder in DDP(broadcast_buffers=False)), the class-sharded head sees the gathered embeddings (nets/PartialFC.py:182-186),
    DDP averages the backbone gradients over the ranks, each rank clips the averaged gradient and takes the same SGD step;
    a rank's head rows are updated from its own shard gradient (never reduced: nets/PartialFC.py:208).
    sds / opts: one state dict / SGDState per rank (mutated in place); head_ws: one [num_local_r, D] shard per rank.
    Returns dict(loss, grad_norm, index[r])."""
    ws = len(sds)
    names = resnet_ref.trainable_names(sds[0])
    leaves, feats = [], []
    for r in range(ws):
        lv = {k: sds[r][k].detach().clone().requires_grad_(True) for k in names}
        work = dict(sds[r])
        work.update(lv)
        raw = resnet_ref.resnet_forward(work, imgs[r], blocks, True, emd_size)
        for k in sds[r]:
            if k not in lv:
                sds[r][k] = work[k]
        leaves.append(lv)
        feats.append(F.normalize(raw))
    h = head_ref.head_all_shards([f.detach() for f in feats], idss, head_ws, num_classes, s, m,
                                 sample_rate=sample_rate, uniforms=uniforms)
    for r in range(ws):
        feats[r].backward(h["d_emb"][r])
    avg = {k: sum(leaves[r][k].grad for r in range(ws)) / ws for k in names}        # DDP: mean over the ranks
    coef, total = clip_coef([avg[k] for k in names])
    with torch.no_grad():
        for r in range(ws):
            for k in names:
                opts[r].apply(k, sds[r][k], avg[k] * coef)
            opts[r].head_update(head_ws[r], h["index"][r], h["d_w_act"][r], sample_rate)
    return dict(loss=h["loss"], grad_norm=total, index=h["index"])
