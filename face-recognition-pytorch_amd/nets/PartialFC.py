"""MI355X-native drop-in for the reference class-sharded classifier `nets/PartialFC.py`.

Interface kept from /root/reference/nets/PartialFC.py: `PartialFC(conf, num_classes, margin_loss=ArcFace)` (:30-90),
`PartialFCAdamW` (:235-342), `.forward(local_embeddings, local_labels, optimizer) -> 0-dim loss` (:146-208),
`.sample` (:92-131), `.update` (:133-143), `.state_dict() -> {"weight"}` / `.load_state_dict` (:210-232), attributes
`rank, world_size, num_local, class_start, num_sample, weight, weight_activated, weight_index`.
torch.distributed must be initialised first (:47-49).

What runs where
  * shard arithmetic, label re-basing and the choice of sampled rows are host-orchestrated integer ops on [N] /
    [num_local] vectors; the uniform draws come from torch's CPU generator exactly like the reference (:110), so
    `weight_index` is reproducible from the CPU seed alone;
  * everything floating-point is libfrhip: row l2-normalise, ONE fused kernel for cos-theta GEMM -> clamp ->
    ArcFace margin -> x s -> per-row max / sum-exp (logits never reach HBM), a recompute kernel that emits
    d loss / d cos once, two MFMA TN GEMMs for dW and dE, normalise-backward, and row gather / scatter of the
    sampled class centres;
  * cross-rank traffic is torch.distributed (backend "nccl" = RCCL over xGMI), FOUR collectives per step on
    preallocated flat buffers: an all-gather of the labels at the start of the step (prepare(): off the critical
    path), an all-gather of the embeddings (:182), ONE all-gather of the packed per-row {max, sum-exp, target logit}
    triples that every rank merges itself (the reference issues all-reduce MAX, SUM, SUM, :448-459), and ONE
    reduce-scatter of dE issued before the dW GEMM so the two overlap (the reference loops world_size reduce() calls,
    :510-519).
The floating-point steps sit behind `HipHeadKernels`; tests on CPU/gloo swap in an oracle-backed double to
exercise the distributed host logic without a GPU.  There is no built-in CPU fallback.
"""
import collections
import weakref
from typing import Callable

import torch
from torch import distributed

from .ArcFace import ArcFace


import os

# test hook: take the multi-rank branch (all-gather, all-reduces, reduce-scatter) even in a 1-rank group, so the
# RCCL call sequence can be exercised on a single-GPU box
_FORCE_COLLECTIVES = os.environ.get("FRHIP_FORCE_COLLECTIVES", "0") == "1"
_PFC_SAMPLE_KERNEL = os.environ.get("FRHIP_PFC_SAMPLE_KERNEL", "1") == "1"
_EARLY_HEAD_UPDATE = os.environ.get("FRHIP_EARLY_HEAD_UPDATE", "1") == "1"
_HEAD_DW_FUSED = os.environ.get("FRHIP_HEAD_DW_FUSED", "1") == "1"       # 0: class-centre gradient as GEMM + normalise-backward pass


# --------------------------------------------------------------------------------------------- kernels
class HipHeadKernels:
    """The floating-point steps of the head on the MI355X (through the C ABI)."""

    def __init__(self, dtype):
        from frhip import ops          # raises FrhipError when libfrhip.so is missing
        self.ops, self.dtype = ops, dtype

    def normalize(self, x):
        return self.ops.l2norm_rows(x.contiguous(), self.dtype)          # (xhat, norms)

    def forward_stats(self, ehat, what, labels_i32, s, m):
        return self.ops.head_fwd(ehat, what, labels_i32, s, m)           # (ztarget, rowmax, rowsum) of this shard

    def rescale(self, rowsum, local_max, global_max):
        self.ops.head_rescale(rowsum, local_max, global_max)

    def target_prob(self, zt, labels_i32, rmax, rsum):
        return self.ops.head_target_prob(zt, labels_i32, rmax, rsum)

    def loss(self, q):
        return self.ops.head_loss(q)

    def pack_stats(self, zt, labels_i32, rmax, rsum):
        return self.ops.head_pack_stats(zt, labels_i32, rmax, rsum)

    def merge_stats(self, gathered):
        return self.ops.head_merge_stats(gathered)                         # global (rowmax, rowsum, q)

    def backward(self, ehat, enorm, what, wnorm, labels_i32, s, m, rmax, rsum, n_global, upstream, e_scale=1.0, on_de=None):
        """-> (d_emb * e_scale, d_weight).  on_de(d_emb) is called as soon as the embedding gradient is enqueued, before the
        weight-gradient GEMM: the caller starts the cross-rank reduce-scatter there and the two overlap."""
        ops = self.ops
        n, d = ehat.shape
        classes = what.shape[0]
        # dT and its transpose from ONE launch (the embedding gradient contracts over classes, the weight gradient over samples)
        dt, dtt = ops.head_bwd_dt(ehat, what, labels_i32, s, m, rmax, rsum, 1.0 / n_global, upstream, transposed=True)
        d_eh = torch.zeros((n, d), dtype=torch.float32, device=ehat.device)
        ops.gemm_tn(dtt, what, d_eh, kc=n)
        d_e = ops.l2norm_bwd(d_eh, ehat, enorm, out_scale=e_scale)
        if on_de is not None:
            on_de(d_e)
        d_w = ops.head_dw(dt, ehat, what, wnorm) if _HEAD_DW_FUSED else None      # GEMM + normalise-backward in one launch (bf16, d = 512)
        if d_w is not None:
            return d_e, d_w
        d_wh = torch.empty((classes, d), dtype=torch.float32, device=ehat.device)
        ops.gemm_tn(dt, ehat, d_wh, kc=classes, overwrite=True)      # 250 MB at 122 000 classes: stored once, never zero-filled
        return d_e, ops.l2norm_bwd(d_wh, what, wnorm)

    def gather_rows(self, table, index):
        return self.ops.gather_rows(table, index)

    def pfc_sample_max_local(self):
        return self.ops.lib().frhip_pfc_sample_max_local()

    def pfc_sample(self, labels, class_start, num_local, u, num_sample, index_out, rel_out, count_out):
        ops = self.ops
        ops.check(ops.lib().frhip_pfc_sample(ops._p(labels), labels.numel(), int(class_start), num_local, ops._p(u), num_sample,
                                             ops._p(index_out), ops._p(rel_out), ops._p(count_out), ops._s()), "frhip_pfc_sample")

    def scatter_rows(self, rows, index, table):
        self.ops.scatter_rows(rows.contiguous(), index, table)


def _default_kernels(conf):
    from ._backbone import compute_dtype
    return HipHeadKernels(compute_dtype(conf))


# --------------------------------------------------------------------------------------------- collectives
def _backend_is_nccl():
    return distributed.get_backend() == "nccl"


def _all_gather_flat(out, inp):
    """all-gather into ONE preallocated buffer (out = [world_size * rows, ...]): no per-rank tensor list, no c10d copies"""
    distributed.all_gather_into_tensor(out, inp)
    return out


def _reduce_scatter_sum(stacked, rank, rows, async_op=False):
    """-> (this rank's [rows, ...] slice of the SUM over ranks of `stacked`, work or None)"""
    if _backend_is_nccl():
        out = torch.empty_like(stacked[:rows])
        work = distributed.reduce_scatter_tensor(out, stacked, op=distributed.ReduceOp.SUM, async_op=async_op)
        return out, (work if async_op else None)
    distributed.all_reduce(stacked, op=distributed.ReduceOp.SUM)       # gloo (tests) has no reduce_scatter
    return stacked[rank * rows:(rank + 1) * rows].clone(), None


class AllGatherFunc(torch.autograd.Function):
    """all_gather with gradient: backward = reduce-scatter(SUM) of the per-chunk gradients, x world_size
    (reference :495-525, which loops world_size reduce() calls).  Kept for the reference's surface
    (`AllGather(tensor, *gather_list)`); PartialFC.forward itself gathers inside its fused autograd node."""

    @staticmethod
    def forward(ctx, tensor, *gather_list):
        gather_list = list(gather_list)
        distributed.all_gather(gather_list, tensor.contiguous())
        return tuple(gather_list)

    @staticmethod
    def backward(ctx, *grads):
        ws, rank = distributed.get_world_size(), distributed.get_rank()
        stacked = torch.cat([g.contiguous() for g in grads])
        out, _ = _reduce_scatter_sum(stacked, rank, grads[rank].shape[0])
        out *= ws
        return (out, *[None for _ in grads])


AllGather = AllGatherFunc.apply


class _MarginSoftmaxFn(torch.autograd.Function):
    """all-gather -> normalise -> cos -> margin -> distributed softmax-CE, as ONE autograd node over the fused kernels,
    from the rank's LOCAL embeddings to the global loss.
    Arithmetic: SURVEY.md Appendix A steps 1-7 (nets/PartialFC.py:182, :198-207, nets/ArcFace.py:76-91, :441-484, :504-522)."""

    @staticmethod
    def forward(ctx, local_embeddings, weight_activated, labels_i32, kern, s, m, world_size, collectives):
        local_embeddings = local_embeddings.contiguous()
        rows, dim = local_embeddings.shape
        if collectives:                                                         # :182 (C1)
            embeddings = _all_gather_flat(local_embeddings.new_empty((world_size * rows, dim)), local_embeddings)
        else:
            embeddings = local_embeddings
        ehat, enorm = kern.normalize(embeddings)
        what, wnorm = kern.normalize(weight_activated)
        zt, rmax, rsum = kern.forward_stats(ehat, what, labels_i32, s, m)
        if collectives:                                                         # :448, :453, :459 (C3-C5) in one exchange
            mine = kern.pack_stats(zt, labels_i32, rmax, rsum)
            allst = _all_gather_flat(mine.new_empty((world_size * mine.shape[0], mine.shape[1])), mine)
            rmax, rsum, q = kern.merge_stats(allst.view(world_size, mine.shape[0], mine.shape[1]))
        else:
            q = kern.target_prob(zt, labels_i32, rmax, rsum)
        loss = kern.loss(q)
        ctx.kern, ctx.s, ctx.m, ctx.world_size, ctx.collectives, ctx.rows = kern, s, m, world_size, collectives, rows
        ctx.save_for_backward(ehat, enorm, what, wnorm, labels_i32, rmax, rsum)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_loss):
        ehat, enorm, what, wnorm, labels_i32, rmax, rsum = ctx.saved_tensors
        up = grad_loss.reshape(1).float().contiguous()
        if not ctx.collectives:
            d_e, d_w = ctx.kern.backward(ehat, enorm, what, wnorm, labels_i32, ctx.s, ctx.m, rmax, rsum, ehat.shape[0], up)
            return d_e, d_w, None, None, None, None, None, None
        # :504-522 (C6): reduce-scatter(SUM) of dE, x world_size (folded into the normalise-backward's scale); issued as soon
        # as dE is enqueued so that it runs beside the dW GEMM
        pending = []
        rank = distributed.get_rank()

        def start(d_e):
            pending.append(_reduce_scatter_sum(d_e, rank, ctx.rows, async_op=True))

        _, d_w = ctx.kern.backward(ehat, enorm, what, wnorm, labels_i32, ctx.s, ctx.m, rmax, rsum, ehat.shape[0], up,
                                   e_scale=float(ctx.world_size), on_de=start)
        d_local, work = pending[0]
        if work is not None:
            work.wait()
        return d_local, d_w, None, None, None, None, None, None


class DistCrossEntropyFunc(torch.autograd.Function):
    """Explicit-logit softmax-CE across class shards (reference :435-484): in place on `logits`, three all-reduces.
    Row kernels of libfrhip (frhip_rows_max / _rows_exp_sum / _rows_normalize / _ce_grad); the backward scale is
    read from the device (no `.item()` host sync).  PartialFC.forward does not use this: its CE is fused."""

    @staticmethod
    def forward(ctx, logits, label):
        from frhip import ops
        if not logits.is_cuda:
            raise RuntimeError("nets.PartialFC.DistCrossEntropy (frhip): logits must live on the MI355X")
        assert logits.dtype == torch.float32 and logits.is_contiguous()
        logits = logits.detach()        # overwritten in place like the reference (:449-455); callers never reuse it
        lab = label.reshape(-1).long().contiguous()
        multi = distributed.is_initialized() and distributed.get_world_size() > 1
        rmax = ops.rows_max(logits)
        if multi:
            distributed.all_reduce(rmax, distributed.ReduceOp.MAX)
        rsum = ops.rows_exp_sum(logits, rmax)
        if multi:
            distributed.all_reduce(rsum, distributed.ReduceOp.SUM)
        q = ops.rows_normalize(logits, rsum, lab)
        if multi:
            distributed.all_reduce(q, distributed.ReduceOp.SUM)
        ctx.save_for_backward(logits, lab)
        return ops.head_loss(q).reshape(())

    @staticmethod
    def backward(ctx, loss_gradient):
        from frhip import ops
        p, lab = ctx.saved_tensors
        return ops.ce_grad(p, lab, 1.0 / p.shape[0], loss_gradient.reshape(1).float().contiguous()), None


class DistCrossEntropy(torch.nn.Module):
    def forward(self, logit_part, label_part):
        return DistCrossEntropyFunc.apply(logit_part, label_part)


# --------------------------------------------------------------------------------------------- module
class _PartialFCBase(torch.nn.Module):
    _version = 1

    def __init__(self, conf, num_classes, margin_loss: Callable = ArcFace, kernels=None):
        super().__init__()
        assert distributed.is_initialized(), "must initialize distributed before create this"
        self.rank = distributed.get_rank()
        self.world_size = distributed.get_world_size()
        self.dist_cross_entropy = DistCrossEntropy()
        self.embedding_size = conf.emd_size
        self.sample_rate: float = conf.sample_rate
        self.fp16 = conf.mixed_precision
        self.num_local: int = num_classes // self.world_size + int(self.rank < num_classes % self.world_size)
        self.class_start: int = num_classes // self.world_size * self.rank + min(self.rank, num_classes % self.world_size)
        self.num_sample: int = int(self.sample_rate * self.num_local)
        self.last_batch_size: int = 0
        self.is_updated: bool = True
        self.init_weight_update: bool = True
        self._state_names = self._optimizer_state_names()
        init = torch.normal(0, 0.01, (self.num_local, self.embedding_size))
        if self.sample_rate < 1:
            self.register_buffer("weight", tensor=init)
            for nm in self._state_names:
                self.register_buffer("weight_" + nm, tensor=torch.zeros_like(init))
            self.register_parameter("weight_activated", param=torch.nn.Parameter(torch.empty(0, 0)))
            for nm in self._state_names:
                self.register_buffer("weight_activated_" + nm, tensor=torch.empty(0, 0))
            self.register_buffer("weight_index", tensor=torch.empty(0, 0))
        else:
            self.weight_activated = torch.nn.Parameter(init)
        if isinstance(margin_loss, Callable):
            self.margin_softmax = margin_loss(conf.loss_s, conf.loss_m)
        else:
            raise
        if getattr(self.margin_softmax, "kind", None) != "arcface":
            raise NotImplementedError("the fused head kernel implements the ArcFace margin (the reference default)")
        self._kernels = kernels
        self._conf = conf
        self.step = 0
        # CPU generator of the sampling draws.  None = torch's default generator, exactly like the reference (:110); a
        # dedicated torch.Generator keeps the draws apart from everything else that consumes the default one
        self.generator = getattr(conf, "sample_generator", None)

    # -- subclass hooks
    def _optimizer_state_names(self):
        raise NotImplementedError

    def _install_optimizer_state(self, optimizer):
        raise NotImplementedError

    @property
    def kernels(self):
        if self._kernels is None:
            self._kernels = _default_kernels(self._conf)
        return self._kernels

    @torch.no_grad()
    def sample(self, labels, index_positive, optimizer, n_positive=None):
        """Choose the rows of this shard that take part in the step and re-express labels as positions in
        that list (reference :92-131 / :309-327).  Mutates `labels` in place like the reference.

        n_positive: number of distinct owned classes in the batch when the caller already knows it (prepare()): the
        usual branch (num_sample >= n_positive) then runs without a single host synchronisation -- no boolean-mask
        indexing, no unique() -- and produces the same index set (a set of rows: independent of top-k tie order)."""
        self.step += 1
        dev = labels.device
        if n_positive is not None and self.num_sample - n_positive >= 0:
            lab, mask = labels.view(-1), index_positive.view(-1)
            perm = self._draw_perm(dev)                          # CPU generator, then moved: same draws as the reference
            ext = torch.cat([perm, perm.new_zeros(1)])
            ext.index_fill_(0, torch.where(mask, lab, torch.full_like(lab, self.num_local)), 2.0)    # perm[positive] = 2
            index = torch.topk(ext[:self.num_local], k=self.num_sample)[1].sort()[0]
            pos = torch.searchsorted(index, lab.clamp(min=0))
            labels.copy_(torch.where(mask, pos, lab).view(labels.shape))
        else:
            positive = torch.unique(labels[index_positive], sorted=True)
            if self.num_sample - positive.size(0) >= 0:
                perm = torch.rand(size=[self.num_local], generator=self.generator).to(dev)
                perm[positive] = 2.0
                index = torch.topk(perm, k=self.num_sample)[1]
                index = index.sort()[0]
            else:
                index = positive
            labels[index_positive] = torch.searchsorted(index, labels[index_positive])
        self._install_sample(index, optimizer)

    def _install_sample(self, index, optimizer):
        """rows `index` of the shard become this step's parameter (reference :120-129)"""
        self.weight_index = index
        k = self.kernels
        self.weight_activated = torch.nn.Parameter(k.gather_rows(self.weight, index))
        for nm in self._state_names:
            setattr(self, "weight_activated_" + nm, k.gather_rows(getattr(self, "weight_" + nm), index))
        self._install_optimizer_state(optimizer)

    def _sample_kernel_ok(self, dev):
        """one launch of frhip_pfc_sample (csrc/pfc_sample.hip) instead of the ~25 torch launches of the label side"""
        k = self.kernels
        return (_PFC_SAMPLE_KERNEL and dev.type == "cuda" and getattr(k, "pfc_sample", None) is not None
                and self.num_local <= k.pfc_sample_max_local())

    def _draw_perm(self, dev):
        """torch.rand(num_local) from the CPU generator (the reference's draws, :110) delivered to the device through a
        rotating set of pinned buffers: a pageable host-to-device copy blocks the host until the GPU reaches it, which
        drains the launch queue in the middle of the step."""
        if dev.type != "cuda":
            return torch.rand(size=[self.num_local], generator=self.generator).to(dev)
        pins = getattr(self, "_perm_pins", None)
        if pins is None:
            pins = self._perm_pins = [torch.empty(self.num_local, pin_memory=True) for _ in range(4)]
            self._perm_turn = 0
        buf = pins[self._perm_turn % len(pins)]
        self._perm_turn += 1
        torch.rand(size=[self.num_local], generator=self.generator, out=buf)
        return buf.to(dev, non_blocking=True)

    @torch.no_grad()
    def prepare(self, local_labels, optimizer=None):
        """Optional, call at the START of a step (before the backbone is enqueued): gathers the labels of all ranks and
        counts this shard's distinct positives.  The one host synchronisation sampling needs (is num_sample >= #positives?,
        reference :112) then happens while the GPU still has the previous step to chew on, and forward() runs without any
        -- a mid-step synchronisation drains the launch queue and costs ~3 ms of idle GPU per step at B = 512.

        With `optimizer` the whole label side of forward() moves here as well: update() of the previous step's rows, the
        shard-relative labels, sample() (index draw, row gathers, optimizer parameter swap).  None of it needs the
        embeddings, and it is ~60 small launches whose HOST cost (30-50 us each) sits between the backbone and the head
        otherwise: with a sampled head (rate 0.1: 60 us of GPU work) the GPU catches up with the host there and idles
        for ~1.5 ms per step.  At the start of a step the host is ahead of the GPU, so the same work is free."""
        lab = local_labels.view(-1).long()
        if self.world_size > 1 or _FORCE_COLLECTIVES:
            labels = _all_gather_flat(lab.new_empty(self.world_size * lab.numel()), lab.contiguous())     # :183 (C2)
        else:
            labels = lab.clone()
        n_pos, check = None, None
        rng_before = self._rng_state() if (self.sample_rate < 1 and optimizer is not None) else None
        if self.sample_rate < 1 and optimizer is not None and self._sample_kernel_ok(labels.device):
            # the whole label side in one kernel: distinct positives, the sampled rows (all positives + the rows with the largest
            # draws, ascending) and the labels re-expressed as positions in that list.  Optimistic like the torch route below: the
            # count travels to a pinned slot and forward() verifies num_sample >= #positives a backbone pass later.
            self.update()
            self.step += 1
            dev = labels.device
            u = self._draw_perm(dev)
            index = torch.empty(self.num_sample, dtype=torch.int64, device=dev)
            ready = torch.empty(labels.numel(), dtype=torch.int32, device=dev)
            count = torch.empty(1, dtype=torch.int64, device=dev)
            self.kernels.pfc_sample(labels, self.class_start, self.num_local, u, self.num_sample, index, ready, count)
            pins = getattr(self, "_npos_pins", None)
            if pins is None:
                pins = self._npos_pins = [torch.zeros(1, dtype=torch.int64, pin_memory=True) for _ in range(4)]
                self._npos_turn = 0
            pin = pins[self._npos_turn % len(pins)]
            self._npos_turn += 1
            pin.copy_(count, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._install_sample(index, optimizer)
            self._prep = (local_labels.data_ptr(), labels, 0, ready, [ev, pin, rng_before, None, None], rng_before)
            return
        if self.sample_rate < 1:
            mask = (self.class_start <= labels) & (labels < self.class_start + self.num_local)
            hits = torch.zeros(self.num_local + 1, dtype=torch.int32, device=labels.device)
            hits.index_fill_(0, torch.where(mask, labels - self.class_start, torch.full_like(labels, self.num_local)), 1)
            count = hits[:self.num_local].sum()
            if optimizer is not None and labels.is_cuda:
                # No synchronisation at all in the steady state: sample on the assumption num_sample >= #positives (with
                # num_sample = 1 525 rows against ~512 distinct owned labels per step it always holds), ship the count to
                # a pinned slot, and let forward() -- a whole backbone pass later, the copy is long done -- verify it and
                # redo the sampling on the reference's other branch (RNG state restored) in the case it does not.  A
                # blocking .item() here stops the host from running ahead across the step boundary: the GPU then starts
                # every step with an empty queue.
                pins = getattr(self, "_npos_pins", None)
                if pins is None:
                    pins = self._npos_pins = [torch.zeros(1, dtype=torch.int64, pin_memory=True) for _ in range(4)]
                    self._npos_turn = 0
                pin = pins[self._npos_turn % len(pins)]
                self._npos_turn += 1
                pin.copy_(count.to(torch.int64).view(1), non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                check = [ev, pin, rng_before, None, None]
                n_pos = 0                                  # optimistic: the sync-free branch of sample()
            else:
                n_pos = int(count.item())
        ready = None
        if optimizer is not None:
            self.update()
            glab = labels.view(-1, 1)
            index_positive = (self.class_start <= glab) & (glab < self.class_start + self.num_local)
            rel = torch.where(index_positive, glab - self.class_start, torch.full_like(glab, -1))
            if self.sample_rate < 1:
                if check is not None:
                    check[3], check[4] = rel.clone(), index_positive
                self.sample(rel, index_positive, optimizer, n_pos)
            ready = rel.view(-1).to(torch.int32).contiguous()
        self._prep = (local_labels.data_ptr(), labels, n_pos, ready, check, rng_before)

    def _rng_state(self):
        return self.generator.get_state() if self.generator is not None else torch.get_rng_state()

    def _set_rng_state(self, state):
        if self.generator is not None:
            self.generator.set_state(state)
        else:
            torch.set_rng_state(state)

    @torch.no_grad()
    def update(self):
        """sampled rows -> full table (reference :133-143)"""
        if self.init_weight_update:
            self.init_weight_update = False
            return
        if self.sample_rate < 1:
            k = self.kernels
            k.scatter_rows(self.weight_activated.data, self.weight_index, self.weight)
            for nm in self._state_names:
                k.scatter_rows(getattr(self, "weight_activated_" + nm), self.weight_index, getattr(self, "weight_" + nm))

    def forward(self, local_embeddings, local_labels, optimizer):
        local_labels.squeeze_()
        prep, self._prep = getattr(self, "_prep", None), None
        if prep is not None and prep[0] != local_labels.data_ptr():
            # a prepare() left over from another step / made for other labels: ignore it.  If it already sampled, hand its
            # draw back to the CPU generator so the sampling below consumes what the reference would
            if prep[3] is not None and self.sample_rate < 1 and prep[5] is not None:
                self._set_rng_state(prep[5])
                self.step -= 1
            prep = None
        local_labels = local_labels.long()
        ready = prep[3] if prep is not None else None
        if ready is None:
            self.update()
        elif prep[4] is not None:                          # verify prepare()'s optimistic sampling (never blocks in practice)
            ev, pin, rng_state, rel, index_positive = prep[4]
            ev.synchronize()
            if int(pin[0]) > self.num_sample:              # more distinct positives than sampled rows: the reference's other branch
                self._set_rng_state(rng_state)             # (it draws nothing; the optimistic draw is handed back)
                self.step -= 1
                with torch.no_grad():
                    if rel is None:                        # the sampling kernel only reported the count: labels from scratch
                        glab = prep[1].view(-1, 1)
                        index_positive = (self.class_start <= glab) & (glab < self.class_start + self.num_local)
                        rel = torch.where(index_positive, glab - self.class_start, torch.full_like(glab, -1))
                    self.sample(rel, index_positive, optimizer, None)
                ready = rel.view(-1).to(torch.int32).contiguous()
        batch_size = local_embeddings.size(0)
        if self.last_batch_size == 0:
            self.last_batch_size = batch_size
        assert self.last_batch_size == batch_size, (
            "last batch size do not equal current batch size: {} vs {}".format(self.last_batch_size, batch_size))
        collectives = self.world_size > 1 or _FORCE_COLLECTIVES
        n_global = batch_size * self.world_size
        s, m = float(self.margin_softmax.scale), float(self.margin_softmax.margin)
        if ready is not None and ready.numel() == n_global:      # everything label-side was done by prepare()
            return _MarginSoftmaxFn.apply(local_embeddings, self.weight_activated, ready, self.kernels, s, m,
                                          self.world_size, collectives)
        n_pos = None
        if prep is not None and prep[1].numel() == n_global:
            labels, n_pos = prep[1], prep[2]                      # gathered at the start of the step by prepare()
        elif collectives:
            labels = _all_gather_flat(local_labels.new_empty(n_global), local_labels.contiguous())      # :183 (C2)
        else:
            labels = local_labels.clone()
        labels = labels.view(-1, 1)
        index_positive = (self.class_start <= labels) & (labels < self.class_start + self.num_local)
        # shard-relative label, -1 when another rank owns the class (reference :188-193); written with where()
        # so no boolean-mask indexing (= no host sync; the step stays capturable in a HIP graph)
        labels = torch.where(index_positive, labels - self.class_start, torch.full_like(labels, -1))
        if self.sample_rate < 1:
            self.sample(labels, index_positive, optimizer, n_pos)
        return _MarginSoftmaxFn.apply(local_embeddings, self.weight_activated, labels.view(-1).to(torch.int32).contiguous(),
                                      self.kernels, s, m, self.world_size, collectives)

    def arm_early_update(self, optimizer):
        """Call between forward() and loss.backward() of a step whose gradient clip leaves the class centres out and that calls
        optimizer.step() exactly once afterwards (model.FR_PartialFC.Model._step does).
        The reference clips the ENCODER's gradients only (model/FR_PartialFC.py:181) and the centres' gradient is final as soon as
        the head's backward has run -- a whole backbone backward before optimizer.step().  With frhip.optim.SGD their update
        (1.25 GB of HBM traffic at 122 000 classes) is then launched from a post-accumulate-grad hook on the weight-gradient side
        stream and runs beside the backbone's backward; step() skips the group.  Same kernel, same arithmetic, only earlier.
        ONE-SHOT: the hook disarms itself, so a backward() that is not followed by step() (gradient accumulation, inspection)
        never changes a parameter.  FRHIP_EARLY_HEAD_UPDATE=0 turns it off."""
        p = self.weight_activated
        if not _EARLY_HEAD_UPDATE or optimizer is None or not hasattr(optimizer, "step_group_early") or not p.is_cuda:
            return
        self._early_armed = optimizer
        if getattr(p, "_frhip_early_hook", None) is not None:
            return

        def hook(param):
            opt, self._early_armed = getattr(self, "_early_armed", None), None
            if opt is None or not opt.param_groups or len(opt.param_groups[-1]["params"]) != 1:
                return
            if opt.param_groups[-1]["params"][0] is not param or torch.cuda.is_current_stream_capturing():
                return
            from . import _backbone as bb

            def launch():
                opt.step_group_early(len(opt.param_groups) - 1, bb.side_stream(param.device))
            if bb.DEFER_EARLY_BLOCKS >= 0:
                bb.park_deferred(opt, launch)      # the backbone's backward pass launches it a few blocks in (or its join() does)
            else:
                launch()

        p._frhip_early_hook = p.register_post_accumulate_grad_hook(hook)

    def state_dict(self, destination=None, prefix="", keep_vars=False):
        if destination is None:
            destination = collections.OrderedDict()
            destination._metadata = collections.OrderedDict()
        for name, module in self._modules.items():
            if module is not None:
                module.state_dict(destination=destination, prefix=prefix + name + ".", keep_vars=keep_vars)
        destination["weight"] = (self.weight if self.sample_rate < 1 else self.weight_activated.data).detach()
        return destination

    def load_state_dict(self, state_dict, strict: bool = True):
        if self.sample_rate < 1:
            self.weight = state_dict["weight"].to(self.weight.device)
            for nm in self._state_names:
                getattr(self, "weight_" + nm).zero_()
                getattr(self, "weight_activated_" + nm).zero_()
            self.weight_activated.data.zero_()
            self.weight_index.zero_()
        else:
            self.weight_activated.data = state_dict["weight"].to(self.weight_activated.data.device)


class PartialFC(_PartialFCBase):
    """SGD flavour: the sampled rows carry their momentum rows (reference :10-232)."""

    def _optimizer_state_names(self):
        return ("mom",)

    def _install_optimizer_state(self, optimizer):
        if isinstance(optimizer, torch.optim.SGD):
            # the params of partial fc must be last in the params list (reference :124)
            optimizer.state.pop(optimizer.param_groups[-1]["params"][0], None)
            optimizer.param_groups[-1]["params"][0] = self.weight_activated
            optimizer.state[self.weight_activated]["momentum_buffer"] = self.weight_activated_mom
        else:
            raise


class PartialFCAdamW(_PartialFCBase):
    """Adam/AdamW flavour: exp_avg / exp_avg_sq rows travel with the sampled rows (reference :235-432)."""

    def _optimizer_state_names(self):
        return ("exp_avg", "exp_avg_sq")

    def _install_optimizer_state(self, optimizer):
        if isinstance(optimizer, (torch.optim.Adam, torch.optim.AdamW)):
            optimizer.state.pop(optimizer.param_groups[-1]["params"][0], None)
            optimizer.param_groups[-1]["params"][0] = self.weight_activated
            optimizer.state[self.weight_activated]["exp_avg"] = self.weight_activated_exp_avg
            optimizer.state[self.weight_activated]["exp_avg_sq"] = self.weight_activated_exp_avg_sq
            optimizer.state[self.weight_activated]["step"] = self.step
        else:
            raise
