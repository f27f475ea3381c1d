"""torch.optim.SGD with the update (and the reference's clip_grad_norm_) running as libfrhip multi-tensor kernels.

Drop-in: same constructor, param_groups, state (`momentum_buffer`) and state_dict as torch.optim.SGD, so the
reference's PartialFC can keep swapping the sampled class-centre parameter and its momentum buffer in and out
(/root/reference/nets/PartialFC.py:120-143) and schedulers keep editing group['lr'].  `.step(clip=(params, max_norm))`
folds torch.nn.utils.clip_grad_norm_(params, max_norm) (/root/reference/model/FR_PartialFC.py:181-190) into the step.
Parameters the kernels cannot take (non-CUDA, non-fp32, sparse, strided differently from their gradient) or options
they do not implement (nesterov, dampening, maximize) make the whole step fall back to torch's own implementation."""
import ctypes

import numpy as np
import torch

from . import ops
from ._abi import check, lib

CHUNK = 65536
_CHUNK_DT = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("n", "<u4"), ("group", "<u4")])


class _Group(ctypes.Structure):
    _fields_ = [("lr", ctypes.c_float), ("weight_decay", ctypes.c_float), ("momentum", ctypes.c_float), ("clip", ctypes.c_float)]


def _dense_same_layout(*ts):
    """same shape, same strides on every dimension that has more than one element, and dense in SOME dimension
    order (contiguous, channels_last, ...): the kernels then walk the storages element by element"""
    shape = ts[0].shape
    dims = [d for d in range(len(shape)) if shape[d] > 1]
    st = [ts[0].stride(d) for d in dims]
    for t in ts[1:]:
        if t.shape != shape or [t.stride(d) for d in dims] != st:
            return False
    expect = 1
    for size, stride in sorted(((shape[d], ts[0].stride(d)) for d in dims), key=lambda x: x[1]):
        if stride != expect:
            return False
        expect *= size
    return True


class SGD(torch.optim.SGD):
    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False, **kw):
        super().__init__(params, lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay,
                         nesterov=nesterov, **kw)
        self._tables = {}            # pointer signature -> (device chunk table, chunk count, per-group clip flags, scratch)
        self._keep = None
        self._coef = self._norm = self._pin = None
        self._pin_next = 0
        self._early = {}             # group index -> stream its update already runs on (step_group_early)

    # ------------------------------------------------------------------------------------------------
    def _options_ok(self):
        if len(self.param_groups) > 8:
            return False
        for g in self.param_groups:
            if g.get("nesterov") or g.get("dampening", 0) != 0 or g.get("maximize") or g.get("differentiable"):
                return False
        return True

    @staticmethod
    def _tensors_ok(entries):
        for p, g, m, _ in entries:
            if not (p.is_cuda and p.dtype == torch.float32 and g.dtype == torch.float32 and not g.is_sparse):
                return False
            if not _dense_same_layout(p.data, g) or (m is not None and not _dense_same_layout(p.data, m)):
                return False
        return True

    def _build_table(self, entries, device):
        """entries: [(p, g, m or None, group index)] -> device chunk table"""
        parts = []
        for p, g, m, gi in entries:
            n = p.numel()
            offs = np.arange(0, n, CHUNK, dtype=np.uint64)
            t = np.empty(len(offs), dtype=_CHUNK_DT)
            t["p"] = p.data_ptr() + 4 * offs
            t["g"] = g.data_ptr() + 4 * offs
            t["m"] = (m.data_ptr() + 4 * offs) if m is not None else 0
            t["n"] = np.minimum(n - offs, CHUNK).astype(np.uint32)
            t["group"] = gi
            parts.append(t)
        host = np.concatenate(parts)
        raw = torch.from_numpy(host.view(np.uint8).copy())
        # staged through a pre-allocated pinned slot with an async copy, so a table built while a HIP graph is being
        # captured (the gradient arena lives at other addresses there) becomes a memcpy node instead of an illegal
        # synchronous copy; the slot stays untouched for as long as the cache entry lives
        if self._pin is None or self._pin.shape[1] < raw.numel():
            self._pin = torch.empty((8, raw.numel()), dtype=torch.uint8, pin_memory=True)
            self._pin_next = 0
        slot = self._pin[self._pin_next % 8, :raw.numel()]
        self._pin_next += 1
        slot.copy_(raw)
        buf = torch.empty(raw.numel(), dtype=torch.uint8, device=device)
        buf.copy_(slot, non_blocking=True)
        return buf, len(host)

    def _group_table(self, gi, g, clip_sig, clip):
        """-> (cached chunk table of parameter group gi or None when it has no gradients, entries); (False, None): not servable"""
        mom = g["momentum"] != 0
        entries = []
        for p in g["params"]:
            gr = p.grad
            if gr is not None:
                entries.append((p, gr, self.state[p].get("momentum_buffer") if mom else None, gi))
        if not entries:
            return None, None
        key = (gi, clip_sig) + tuple((p.data_ptr(), gr.data_ptr(), 0 if m is None else m.data_ptr(), p.numel()) for p, gr, m, _ in entries)
        hit = self._tables.get(key)
        if hit is None:
            fixed = []
            for p, gr, m, _ in entries:
                if mom and (m is None or not _dense_same_layout(p.data, m)):
                    m = self.state[p]["momentum_buffer"] = torch.zeros_like(p.data, memory_format=torch.preserve_format)
                fixed.append((p, gr, m, gi))
            entries = fixed
            clip_ids = set() if clip is None else {id(p) for p in clip[0]}
            inside = [id(p) in clip_ids for p, _, _, _ in entries]
            if not self._tensors_ok(entries) or (any(inside) and not all(inside)):
                return False, None
            device = entries[0][0].device
            table, n = self._build_table(entries, device)
            hit = (table, n, all(inside), torch.empty(n, dtype=torch.float32, device=device))
            if len(self._tables) >= 16:
                self._tables.clear()
            key = (gi, clip_sig) + tuple((p.data_ptr(), gr.data_ptr(), 0 if m is None else m.data_ptr(), p.numel()) for p, gr, m, _ in entries)
            self._tables[key] = hit
            if self._coef is None or self._coef.device != device:
                self._coef = torch.ones(2, dtype=torch.float32, device=device)
        return hit, entries

    @torch.no_grad()
    def step_group_early(self, gi, stream):
        """Update parameter group `gi` NOW, on `stream`, from the gradients it already has -- for a group that takes no part in the
        gradient clip (the PartialFC class centres: their gradient is final when the head's backward has run, a whole backbone
        backward before step(); their 1.25-GB update then runs beside it instead of after it).  step() skips the group and makes
        its stream wait for `stream`.  Returns False (and does nothing) when the fused path cannot serve the group."""
        if not self._options_ok() or gi in self._early:
            return False
        g = self.param_groups[gi]
        hit, entries = self._group_table(gi, g, ("early",), None)
        if not hit:
            return False
        groups = (_Group * len(self.param_groups))()
        for k, gg in enumerate(self.param_groups):
            groups[k] = _Group(float(gg["lr"]), float(gg["weight_decay"]), float(gg["momentum"]), 0.0)
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            check(lib().frhip_sgd_multi(ctypes.c_void_p(hit[0].data_ptr()), hit[1], ctypes.cast(groups, ctypes.c_void_p), len(groups),
                                        None, ops._s()), "frhip_sgd_multi")
        self._early[gi] = stream
        self._keep_early = entries
        return True

    def zero_grad(self, set_to_none=True):
        """A new step begins: an early group update whose step() never came (exception, skipped step, inspection between backward()
        and step()) must not make the NEXT step's hook think its group is already done -- join its stream and forget it."""
        for st in self._early.values():
            torch.cuda.current_stream().wait_stream(st)
        self._early = {}
        # a head update still parked for a backward pass that never reached a frhip backbone belongs to the step that ends here
        self._frhip_step_token = getattr(self, "_frhip_step_token", 0) + 1
        from nets import _backbone as _bb
        _bb.drop_deferred(self)
        return super().zero_grad(set_to_none=set_to_none)

    def _fallback(self, clip, done=()):
        """torch's own step; `done` = group indices step_group_early has already updated (their gradients are hidden meanwhile)"""
        self._norm = torch.nn.utils.clip_grad_norm_(list(clip[0]), float(clip[1])) if clip is not None else None
        hidden = []
        for gi in done:
            for p in self.param_groups[gi]["params"]:
                hidden.append((p, p.grad))
                p.grad = None
        super().step()
        for p, g in hidden:
            p.grad = g

    @torch.no_grad()
    def step(self, closure=None, clip=None):
        """clip=(iterable of parameters, max_norm): scale the gradients of those parameters by
        min(1, max_norm / (total_norm + 1e-6)) inside the update (they must make up whole parameter groups)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if clip is not None:
            clip = (list(clip[0]), float(clip[1]))
        early, self._early = self._early, {}
        for gi, st in early.items():
            torch.cuda.current_stream().wait_stream(st)
            if clip is not None and {id(p) for p in self.param_groups[gi]["params"]} & {id(p) for p in clip[0]}:
                raise RuntimeError("frhip.optim.SGD: group %d was updated by step_group_early but takes part in the gradient clip" % gi)
        if not self._options_ok():
            self._fallback(clip, early)
            return loss
        clip_sig = None if clip is None else (id(clip[0][0]) if clip[0] else 0, len(clip[0]))
        # One chunk table per parameter group, cached by the pointer + size signature of its (parameter, gradient, momentum)
        # triples (the size matters: the caching allocator hands a freed block to a tensor of another size at the same
        # address, and PartialFC's `index = positive` branch changes the row count of its parameter from step to step): PartialFC swaps the sampled class-centre parameter of the LAST group every step, which then rebuilds a
        # ~100-chunk table instead of the whole model's.
        hits, keep = [], []
        for gi, g in enumerate(self.param_groups):
            if gi in early:                                # updated during the backward pass already (step_group_early)
                hits.append(None)
                continue
            hit, entries = self._group_table(gi, g, clip_sig, clip)
            if hit is False:
                self._fallback(clip, early)               # incl. a clip set that cuts through a group
                return loss
            hits.append(hit)
            if hit is not None:
                keep.append(entries)
        if not keep:
            return loss
        self._keep = keep                                  # the tables hold raw pointers: keep their owners alive
        groups = (_Group * len(self.param_groups))()
        clipped = []
        for gi, g in enumerate(self.param_groups):
            c = hits[gi] is not None and hits[gi][2]
            if c:
                clipped.append(gi)
            groups[gi] = _Group(float(g["lr"]), float(g["weight_decay"]), float(g["momentum"]), 1.0 if c else 0.0)
        gptr, coef = ctypes.cast(groups, ctypes.c_void_p), None
        if len(clipped) > 1:
            self._fallback(clip, early)                    # the norm would span several tables: not built (torch does it)
            return loss
        if clipped:
            table, n, _, partial = hits[clipped[0]]
            check(lib().frhip_sgd_clip_coef(ctypes.c_void_p(table.data_ptr()), n, gptr, len(groups), clip[1], ops._p(partial),
                                            ops._p(self._coef), ops._s()), "frhip_sgd_clip_coef")
            coef = ops._p(self._coef)
            self._norm = self._coef[1]
        for hit in hits:
            if hit is not None:
                check(lib().frhip_sgd_multi(ctypes.c_void_p(hit[0].data_ptr()), hit[1], gptr, len(groups), coef, ops._s()),
                      "frhip_sgd_multi")
        return loss

    def last_grad_norm(self):
        """total gradient norm of the clipped groups at the last step (device scalar), as clip_grad_norm_ returns it"""
        return self._norm


# ----------------------------------------------------------------------------------------------------- AdamW
_ACHUNK_DT = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<u4"), ("group", "<u4")])


class _AGroup(ctypes.Structure):
    _fields_ = [(k, ctypes.c_float) for k in ("lr", "beta1", "beta2", "eps", "weight_decay", "bc1", "bc2", "clip")]


def _step_value(st):
    v = st.get("step", 0)
    return int(v.item()) if torch.is_tensor(v) else int(v)


class AdamW(torch.optim.AdamW):
    """torch.optim.AdamW whose step (and an optional clip_grad_norm_) runs as libfrhip multi-tensor kernels.  Same
    state keys (`step`, `exp_avg`, `exp_avg_sq`) as torch, so PartialFCAdamW can keep moving the rows of its sampled
    parameter in and out (/root/reference/nets/PartialFC.py:235-342).  The step counter must be uniform inside a
    parameter group (it is: torch bumps every parameter together, PartialFCAdamW owns a group of its own);
    anything else (amsgrad, maximize, mixed steps, non-fp32 / non-dense tensors) falls back to torch's step."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, **kw):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, **kw)
        self._tables = {}
        self._keep = None
        self._coef = self._norm = self._pin = None
        self._pin_next = 0

    def _options_ok(self):
        if len(self.param_groups) > 8:
            return False
        return not any(g.get("amsgrad") or g.get("maximize") or g.get("differentiable") or g.get("capturable")
                       for g in self.param_groups)

    def _table(self, entries, device):
        parts = []
        for p, g, m, v, gi in entries:
            n = p.numel()
            offs = np.arange(0, n, CHUNK, dtype=np.uint64)
            t = np.empty(len(offs), dtype=_ACHUNK_DT)
            t["p"], t["g"] = p.data_ptr() + 4 * offs, g.data_ptr() + 4 * offs
            t["m"], t["v"] = m.data_ptr() + 4 * offs, v.data_ptr() + 4 * offs
            t["n"] = np.minimum(n - offs, CHUNK).astype(np.uint32)
            t["group"] = gi
            parts.append(t)
        host = np.concatenate(parts)
        raw = torch.from_numpy(host.view(np.uint8).copy())
        if self._pin is None or self._pin.shape[1] < raw.numel():
            self._pin = torch.empty((8, raw.numel()), dtype=torch.uint8, pin_memory=True)
            self._pin_next = 0
        slot = self._pin[self._pin_next % 8, :raw.numel()]
        self._pin_next += 1
        slot.copy_(raw)
        buf = torch.empty(raw.numel(), dtype=torch.uint8, device=device)
        buf.copy_(slot, non_blocking=True)
        return buf, len(host)

    def _fallback(self, clip):
        self._norm = torch.nn.utils.clip_grad_norm_(list(clip[0]), float(clip[1])) if clip is not None else None
        super().step()

    @torch.no_grad()
    def step(self, closure=None, clip=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if clip is not None:
            clip = (list(clip[0]), float(clip[1]))
        if not self._options_ok():
            self._fallback(clip)
            return loss
        entries, steps = [], []
        for gi, g in enumerate(self.param_groups):
            gstep = None
            for p in g["params"]:
                gr = p.grad
                if gr is None:
                    continue
                st = self.state[p]
                sv = _step_value(st)
                if gstep is None:
                    gstep = sv
                elif gstep != sv:
                    self._fallback(clip)               # mixed step counters inside one group
                    return loss
                entries.append((p, gr, st.get("exp_avg"), st.get("exp_avg_sq"), gi))
            steps.append(gstep)
        if not entries:
            return loss
        key = tuple((p.data_ptr(), gr.data_ptr(), 0 if m is None else m.data_ptr(), 0 if v is None else v.data_ptr(), gi, p.numel())
                    for p, gr, m, v, gi in entries)
        clip_sig = None if clip is None else (id(clip[0][0]) if clip[0] else 0, len(clip[0]))
        hit = self._tables.get((key, clip_sig))
        if hit is None:
            ok = all(p.is_cuda and p.dtype == torch.float32 and gr.dtype == torch.float32 and not gr.is_sparse and
                     _dense_same_layout(p.data, gr) for p, gr, _, _, _ in entries)
            clip_ids = set() if clip is None else {id(p) for p in clip[0]}
            group_clip = []
            for gi in range(len(self.param_groups)):
                inside = [id(p) in clip_ids for p, _, _, _, g2 in entries if g2 == gi]
                if any(inside) and not all(inside):
                    ok = False
                group_clip.append(bool(inside) and all(inside))
            if not ok:
                self._fallback(clip)                   # before any state is created: torch initialises empty states itself
                return loss
            fixed = []
            for p, gr, m, v, gi in entries:
                st = self.state[p]
                if "step" not in st:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                if m is None or not _dense_same_layout(p.data, m):
                    m = st["exp_avg"] = torch.zeros_like(p.data, memory_format=torch.preserve_format)
                if v is None or not _dense_same_layout(p.data, v):
                    v = st["exp_avg_sq"] = torch.zeros_like(p.data, memory_format=torch.preserve_format)
                fixed.append((p, gr, m, v, gi))
            entries = fixed
            device = entries[0][0].device
            table, n = self._table(entries, device)
            hit = (table, n, group_clip, torch.empty(n, dtype=torch.float32, device=device))
            if len(self._tables) >= 8:
                self._tables.clear()
            key = tuple((p.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), gi, p.numel()) for p, gr, m, v, gi in entries)
            self._tables[(key, clip_sig)] = hit
            if self._coef is None or self._coef.device != device:
                self._coef = torch.ones(2, dtype=torch.float32, device=device)
        table, n, group_clip, partial = hit
        self._keep = entries
        # bump the step counters the way torch does (tensor or python int, whatever the state holds)
        for p, _, _, _, _ in entries:
            st = self.state[p]
            if torch.is_tensor(st.get("step")):
                st["step"] += 1
            else:
                st["step"] = _step_value(st) + 1
        groups = (_AGroup * len(self.param_groups))()
        for gi, g in enumerate(self.param_groups):
            t = (steps[gi] or 0) + 1
            b1, b2 = g["betas"]
            groups[gi] = _AGroup(float(g["lr"]), float(b1), float(b2), float(g["eps"]), float(g["weight_decay"]),
                                 1.0 - float(b1) ** t, 1.0 - float(b2) ** t, 1.0 if group_clip[gi] else 0.0)
        tbl, gptr, coef = ctypes.c_void_p(table.data_ptr()), ctypes.cast(groups, ctypes.c_void_p), None
        if any(group_clip):
            check(lib().frhip_adamw_clip_coef(tbl, n, gptr, len(groups), clip[1], ops._p(partial), ops._p(self._coef), ops._s()),
                  "frhip_adamw_clip_coef")
            coef = ops._p(self._coef)
            self._norm = self._coef[1]
        check(lib().frhip_adamw_multi(tbl, n, gptr, len(groups), coef, ops._s()), "frhip_adamw_multi")
        return loss

    def last_grad_norm(self):
        return self._norm
