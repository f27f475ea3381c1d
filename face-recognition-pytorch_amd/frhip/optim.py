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
    st = ts[0].stride()
    if any(t.stride() != st or t.shape != ts[0].shape for t in ts[1:]):
        return False
    t = ts[0]
    # dense in SOME dimension order (contiguous or channels_last ...): sorted strides multiply up to numel
    expect = 1
    for size, stride in sorted(zip(t.shape, st), key=lambda x: x[1]):
        if size == 1:
            continue
        if stride != expect:
            return False
        expect *= size
    return True


class SGD(torch.optim.SGD):
    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False, **kw):
        super().__init__(params, lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay,
                         nesterov=nesterov, **kw)
        self._table_key, self._table, self._keep = None, None, None
        self._partial = self._coef = self._norm = None

    # ------------------------------------------------------------------------------------------------
    def _fusable(self):
        if len(self.param_groups) > 8:
            return False
        for g in self.param_groups:
            if g.get("nesterov") or g.get("dampening", 0) != 0 or g.get("maximize") or g.get("differentiable"):
                return False
            for p in g["params"]:
                if p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.grad.dtype == torch.float32 and not p.grad.is_sparse):
                    return False
                if not _dense_same_layout(p.data, p.grad):
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
        buf = torch.from_numpy(host.view(np.uint8).copy()).to(device)
        return buf, len(host)

    @torch.no_grad()
    def step(self, closure=None, clip=None):
        """clip=(iterable of parameters, max_norm): scale the gradients of those parameters by
        min(1, max_norm / (total_norm + 1e-6)) inside the update (they must make up whole parameter groups)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        clip_ids, max_norm = (None, None)
        if clip is not None:
            clip_ids, max_norm = {id(p) for p in clip[0]}, float(clip[1])
        group_clip = []
        ok = self._fusable()
        for g in self.param_groups:
            ps = [p for p in g["params"] if p.grad is not None]
            inside = [clip_ids is not None and id(p) in clip_ids for p in ps]
            if any(inside) and not all(inside):
                ok = False                                # a clip set that cuts through a group: not expressible per group
            group_clip.append(bool(inside) and all(inside))
        if not ok:
            self._norm = torch.nn.utils.clip_grad_norm_(list(clip[0]), max_norm) if clip is not None else None
            super().step()
            return loss

        entries, device = [], None
        for gi, g in enumerate(self.param_groups):
            for p in g["params"]:
                if p.grad is None:
                    continue
                device = p.device
                m = None
                if g["momentum"] != 0:
                    st = self.state[p]
                    m = st.get("momentum_buffer")
                    if m is None or not _dense_same_layout(p.data, m):
                        m = st["momentum_buffer"] = torch.zeros_like(p.data, memory_format=torch.preserve_format)
                entries.append((p.data, p.grad, m, gi))
        if not entries:
            return loss
        key = tuple((p.data_ptr(), g.data_ptr(), 0 if m is None else m.data_ptr(), p.numel(), gi) for p, g, m, gi in entries)
        if key != self._table_key:
            self._table, self._n = self._build_table(entries, device)
            self._table_key = key
            self._partial = torch.empty(self._n, dtype=torch.float32, device=device)
            self._coef = torch.ones(2, dtype=torch.float32, device=device)
        self._keep = entries                               # the table holds raw pointers: keep their owners alive
        groups = (_Group * len(self.param_groups))()
        for gi, g in enumerate(self.param_groups):
            groups[gi] = _Group(float(g["lr"]), float(g["weight_decay"]), float(g["momentum"]), 1.0 if group_clip[gi] else 0.0)
        tbl = ctypes.c_void_p(self._table.data_ptr())
        coef = None
        if any(group_clip):
            check(lib().frhip_sgd_clip_coef(tbl, self._n, ctypes.cast(groups, ctypes.c_void_p), len(groups), max_norm,
                                            ops._p(self._partial), ops._p(self._coef), ops._s()), "frhip_sgd_clip_coef")
            coef = ops._p(self._coef)
            self._norm = self._coef[1]
        check(lib().frhip_sgd_multi(tbl, self._n, ctypes.cast(groups, ctypes.c_void_p), len(groups), coef, ops._s()),
              "frhip_sgd_multi")
        return loss

    def last_grad_norm(self):
        """total gradient norm of the clipped groups at the last step (device scalar), as clip_grad_norm_ returns it"""
        return self._norm
