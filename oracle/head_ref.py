"""CPU restatement of the margin-softmax head (TEST INFRASTRUCTURE, see oracle/__init__.py).

Closed-form forward AND backward (no autograd) of the reference chain
  nets/PartialFC.py:146-208   PartialFC.forward
  nets/PartialFC.py:92-131    PartialFC.sample  (positives + random negatives, CPU RNG)
  nets/ArcFace.py:76-91       ArcFace.forward   (additive angular margin, then x s)
  nets/PartialFC.py:441-484   DistCrossEntropyFunc (softmax-CE across class shards)
  nets/PartialFC.py:495-525   AllGatherFunc (all-gather fwd, reduce-to-owner x ws bwd)
All class shards ("ranks") are simulated in ONE process by looping over them, so
multi-rank behaviour is testable without a process group; the fixtures it is
pinned against were produced by the real reference running on gloo with real
processes (tools/make_golden.py).
"""
import math
import torch

NORM_EPS = 1e-12   # F.normalize default eps (nets/PartialFC.py:199-200)


# ----------------------------------------------------------------------------- shard arithmetic
def shard_range(num_classes, world_size, rank):
    """(class_start, num_local) -- nets/PartialFC.py:57-62."""
    q, r = divmod(num_classes, world_size)
    return q * rank + min(rank, r), q + (1 if rank < r else 0)


def num_sample(sample_rate, num_local):
    """nets/PartialFC.py:63."""
    return int(sample_rate * num_local)


def to_local_labels(labels, class_start, num_local):
    """Global labels [N] -> shard-relative, -1 when another shard owns the class
    (nets/PartialFC.py:188-193)."""
    labels = labels.reshape(-1).long()
    mine = (labels >= class_start) & (labels < class_start + num_local)
    return torch.where(mine, labels - class_start, torch.full_like(labels, -1))


def sample_index(local_labels, num_local, n_sample, u):
    """Rows of the shard that take part this step, and the labels re-expressed as
    positions in that row list (nets/PartialFC.py:108-118).

    u: float32 [num_local] uniform draws (the reference draws them with
    torch.rand on the CPU generator, :110).  Positives get score 2.0 so top-k always
    keeps them; the k winners are returned SORTED, which makes the result a set
    operation independent of top-k tie order."""
    pos_mask = local_labels >= 0
    positive = torch.unique(local_labels[pos_mask], sorted=True)
    if n_sample - positive.numel() >= 0:
        score = u.clone()
        score[positive] = 2.0
        index = torch.topk(score, k=n_sample).indices.sort().values
    else:
        index = positive
    relabeled = local_labels.clone()
    relabeled[pos_mask] = torch.searchsorted(index, local_labels[pos_mask])
    return index, relabeled


# ----------------------------------------------------------------------------- pieces
def l2_normalize(x):
    n = x.norm(dim=1, keepdim=True).clamp_min(NORM_EPS)
    return x / n, n


def l2_normalize_bwd(dxh, xh, n):
    """d/dX of X/max(|X|,eps) for |X| > eps."""
    return (dxh - xh * (dxh * xh).sum(dim=1, keepdim=True)) / n


def margin_constants(s, m):
    """nets/ArcFace.py:66-72."""
    return dict(s=float(s), cos_m=math.cos(m), sin_m=math.sin(m),
                theta=math.cos(math.pi - m), sinmm=math.sin(math.pi - m) * m)


def arcface_logits(cos, local_labels, s, m):
    """ArcFace.forward on already-clamped cosines (nets/ArcFace.py:76-91).
    Returns z = s * margin(cos) and d margin / d cos at every entry."""
    k = margin_constants(s, m)
    z = cos.clone()
    slope = torch.ones_like(cos)
    rows = torch.nonzero(local_labels >= 0).flatten()
    if rows.numel():
        cols = local_labels[rows]
        t = cos[rows, cols]
        sin_t = torch.sqrt(1.0 - t * t)
        ctm = t * k["cos_m"] - sin_t * k["sin_m"]
        easy = t > k["theta"]
        z[rows, cols] = torch.where(easy, ctm, t - k["sinmm"])
        slope[rows, cols] = torch.where(easy, k["cos_m"] + t * k["sin_m"] / sin_t,
                                        torch.ones_like(t))
    return z * k["s"], slope


# ----------------------------------------------------------------------------- whole head
def head_all_shards(emb_per_rank, labels_per_rank, weights, num_classes, s, m,
                    sample_rate=1.0, uniforms=None, upstream=1.0):
    """Forward + backward of the head for every rank of a world of len(weights) ranks.

    emb_per_rank   : list of [B,D] local embeddings (what each rank passes to forward)
    labels_per_rank: list of [B] global labels
    weights        : list of [num_local_r, D] full shard weights
    uniforms       : list of [num_local_r] draws for sampling (needed when sample_rate < 1)
    Returns dict(loss, d_emb[r] (grad wrt rank r's local embeddings, incl. the xws of
    AllGatherFunc.backward), index[r], labels[r] (relabelled), d_w_act[r] (grad wrt the
    activated rows), w_act[r]).
    """
    ws = len(weights)
    emb = torch.cat(list(emb_per_rank))          # all_gather order = rank order (:182,:186)
    lab = torch.cat([l.reshape(-1) for l in labels_per_rank]).long()
    n, b = emb.shape[0], emb_per_rank[0].shape[0]
    eh, en = l2_normalize(emb)

    shards = []
    for r in range(ws):
        c0, nloc = shard_range(num_classes, ws, r)
        ll = to_local_labels(lab, c0, nloc)
        if sample_rate < 1:
            idx, ll = sample_index(ll, nloc, num_sample(sample_rate, nloc), uniforms[r])
            w_act = weights[r][idx]
        else:
            idx, w_act = torch.arange(nloc), weights[r]
        wh, wn = l2_normalize(w_act)
        raw = eh @ wh.t()
        cos = raw.clamp(-1.0, 1.0)
        z, slope = arcface_logits(cos, ll, s, m)
        shards.append(dict(idx=idx, ll=ll, w_act=w_act, wh=wh, wn=wn, raw=raw, z=z, slope=slope))

    # distributed softmax-CE forward (nets/PartialFC.py:444-461)
    gmax = torch.stack([sh["z"].max(dim=1).values for sh in shards]).max(dim=0).values
    gsum = sum(torch.exp(sh["z"] - gmax[:, None]).sum(dim=1) for sh in shards)
    q = torch.zeros(n, dtype=emb.dtype)
    for sh in shards:
        rows = torch.nonzero(sh["ll"] >= 0).flatten()
        sh["p"] = torch.exp(sh["z"] - gmax[:, None]) / gsum[:, None]
        q[rows] += sh["p"][rows, sh["ll"][rows]]
    loss = -(q.clamp_min(1e-30).log().mean())

    # backward (nets/PartialFC.py:464-484 then the autograd chain of :198-206)
    out = dict(loss=loss, d_emb=[None] * ws, index=[], labels=[], d_w_act=[], w_act=[])
    d_gathered = []
    for sh in shards:
        dz = sh["p"].clone()
        rows = torch.nonzero(sh["ll"] >= 0).flatten()
        dz[rows, sh["ll"][rows]] -= 1.0
        dz = dz / n * upstream
        inside = (sh["raw"] >= -1.0) & (sh["raw"] <= 1.0)      # clamp passes grad on [-1,1]
        dcos = dz * s * sh["slope"] * inside
        d_eh = dcos @ sh["wh"]
        d_wh = dcos.t() @ eh
        d_gathered.append(l2_normalize_bwd(d_eh, eh, en))
        out["d_w_act"].append(l2_normalize_bwd(d_wh, sh["wh"], sh["wn"]))
        out["index"].append(sh["idx"]); out["labels"].append(sh["ll"]); out["w_act"].append(sh["w_act"])
    total = sum(d_gathered)                                     # reduce(SUM) to the owner (:510-519)
    for r in range(ws):
        out["d_emb"][r] = total[r * b:(r + 1) * b] * ws        # grad_out *= world_size (:521)
    return out


def dist_cross_entropy(z_shards, ll_shards, upstream=1.0):
    """DistCrossEntropyFunc fwd/bwd on given per-shard scaled logits (:441-484)."""
    n = z_shards[0].shape[0]
    gmax = torch.stack([z.max(dim=1).values for z in z_shards]).max(dim=0).values
    gsum = sum(torch.exp(z - gmax[:, None]).sum(dim=1) for z in z_shards)
    q = torch.zeros(n, dtype=z_shards[0].dtype)
    grads = []
    for z, ll in zip(z_shards, ll_shards):
        p = torch.exp(z - gmax[:, None]) / gsum[:, None]
        rows = torch.nonzero(ll >= 0).flatten()
        q[rows] += p[rows, ll[rows]]
        g = p.clone()
        g[rows, ll[rows]] -= 1.0
        grads.append(g / n * upstream)
    return -(q.clamp_min(1e-30).log().mean()), grads
