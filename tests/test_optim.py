"""frhip.optim.SGD: torch.optim.SGD semantics + clip_grad_norm_ as multi-tensor HIP kernels.
Reference call sites: /root/reference/model/FR_PartialFC.py:153-160 (optimizer), :181-190 (clip + step)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [p for p in (ROOT, os.path.join(ROOT, "face-recognition-pytorch_amd")) if p not in sys.path]


def _make(device, seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(64, 3, 3, 3), (64,), (128, 64, 3, 3), (7,), (513, 129), (1,), (300000,), (0, 0, 0, 0), (128, 64, 1, 1)]
    ps = []
    for i, s in enumerate(shapes):
        if s == (0, 0, 0, 0):
            continue
        t = torch.randn(s, generator=g)
        if len(s) == 4 and i % 2 == 0:
            t = t.contiguous(memory_format=torch.channels_last)    # 1x1 kernels: size-1 dims carry arbitrary strides
        ps.append(torch.nn.Parameter(t.to(device)))
    head = torch.nn.Parameter(torch.randn((1000, 512), generator=g).to(device))
    return ps, head


def _grads(ps, head, seed, scale):
    g = torch.Generator().manual_seed(seed)
    for p in ps + [head]:
        gr = (torch.randn(p.shape, generator=g) * scale).to(p.device)
        if p.dim() == 4 and p.data.is_contiguous(memory_format=torch.channels_last) and p.shape[1] > 1:
            k, c, r, s_ = p.shape
            gr = gr.permute(0, 2, 3, 1).contiguous().view(k, r, s_, c).permute(0, 3, 1, 2)    # as nets._backbone.flat_grads carves them
        p.grad = gr


def _run(opt_cls, device, clip, steps=3, fused_clip=False):
    ps, head = _make(device, 1)
    opt = opt_cls([{"params": ps}, {"params": [head], "weight_decay": 0.0}], lr=0.1, momentum=0.9, weight_decay=5e-4)
    norms = []
    for k in range(steps):
        _grads(ps, head, 10 + k, 3.0 if k == 0 else 0.01)      # step 0 clips, later steps do not
        if k == 2:
            opt.param_groups[0]["lr"] = 0.03                    # schedulers edit lr in place
        if fused_clip:
            opt.step(clip=(ps, clip))
            norms.append(float(opt.last_grad_norm()))
        else:
            norms.append(float(torch.nn.utils.clip_grad_norm_(ps, clip)))
            opt.step()
    return [p.detach().cpu() for p in ps + [head]], [opt.state[p]["momentum_buffer"].cpu() for p in ps + [head]], norms


def test_sgd_falls_back_to_torch_on_cpu_tensors():
    from frhip.optim import SGD
    a = _run(torch.optim.SGD, "cpu", 5.0)
    b = _run(SGD, "cpu", 5.0, fused_clip=True)
    for x, y in zip(a[0] + a[1], b[0] + b[1]):
        assert torch.equal(x, y)


@pytest.mark.gpu
def test_sgd_matches_torch_sgd_and_clip_grad_norm():
    from frhip import optim as _o
    from frhip.optim import SGD
    ref = _run(torch.optim.SGD, "cuda", 5.0)
    fell_back = []
    orig = torch.optim.SGD.step
    torch.optim.SGD.step = lambda self, *a, **k: (fell_back.append(1), orig(self, *a, **k))[1]
    try:
        got = _run(SGD, "cuda", 5.0, fused_clip=True)
    finally:
        torch.optim.SGD.step = orig
    assert not fell_back, "the fused kernels must take these parameters (no silent torch fallback)"
    np.testing.assert_allclose(got[2], ref[2], rtol=1e-5)
    for x, y in zip(got[0] + got[1], ref[0] + ref[1]):
        np.testing.assert_allclose(x.numpy(), y.numpy(), rtol=2e-6, atol=2e-6)


@pytest.mark.gpu
def test_sgd_honours_swapped_parameter_and_momentum_buffer():
    """PartialFC swaps the sampled centre rows and their momentum into the last group every step
    (/root/reference/nets/PartialFC.py:120-143)"""
    from frhip.optim import SGD
    outs = []
    for cls in (torch.optim.SGD, SGD):
        w = torch.nn.Parameter(torch.ones(8, 4, device="cuda"))
        opt = cls([{"params": [w]}], lr=0.5, momentum=0.9, weight_decay=0.1)
        sub = torch.nn.Parameter(torch.full((3, 4), 2.0, device="cuda"))
        mom = torch.full((3, 4), 0.25, device="cuda")
        opt.state.pop(opt.param_groups[-1]["params"][0], None)
        opt.param_groups[-1]["params"][0] = sub
        opt.state[sub]["momentum_buffer"] = mom
        sub.grad = torch.full((3, 4), 0.5, device="cuda")
        opt.step()
        outs.append((sub.detach().cpu(), mom.cpu()))
    assert torch.allclose(outs[0][0], outs[1][0]) and torch.allclose(outs[0][1], outs[1][1])


@pytest.mark.gpu
def test_sgd_swapped_parameter_changes_size_at_the_same_address():
    """PartialFC's `index = positive` branch (/root/reference/nets/PartialFC.py:114) changes the row count of the sampled
    parameter from step to step while the caching allocator re-issues the same base addresses: the cached chunk table must
    not be reused for a different size (stale n = rows left un-updated, or writes past the end).  The tensors of the two
    steps are carved from the same storages so that their addresses coincide by construction."""
    from frhip.optim import SGD
    outs = []
    for cls in (torch.optim.SGD, SGD):
        dummy = torch.nn.Parameter(torch.ones(4, device="cuda"))
        opt = cls([{"params": [dummy]}, {"params": [torch.nn.Parameter(torch.ones(1, 4, device="cuda"))]}], lr=0.5, momentum=0.9,
                  weight_decay=0.1)
        store_p = torch.zeros(6 * 4 + 8, device="cuda")
        store_g = torch.zeros(6 * 4 + 8, device="cuda")
        store_m = torch.zeros(6 * 4 + 8, device="cuda")
        res = []
        for rows in (3, 6, 2):
            store_p.fill_(2.0)
            store_g.fill_(0.5)
            store_m.fill_(0.25)
            sub = torch.nn.Parameter(store_p[:rows * 4].view(rows, 4))
            mom = store_m[:rows * 4].view(rows, 4)
            opt.state.pop(opt.param_groups[-1]["params"][0], None)
            opt.param_groups[-1]["params"][0] = sub
            opt.state[sub]["momentum_buffer"] = mom
            sub.grad = store_g[:rows * 4].view(rows, 4)
            dummy.grad = torch.ones(4, device="cuda")
            opt.step()
            res.append((store_p.cpu().clone(), store_m.cpu().clone()))       # whole storages: the tail must stay untouched
        outs.append(res)
    for (p0, m0), (p1, m1) in zip(*outs):
        assert torch.allclose(p0, p1) and torch.allclose(m0, m1)


def _run_adamw(opt_cls, device, steps=3, fused_clip=False):
    ps, head = _make(device, 2)
    opt = opt_cls([{"params": ps}, {"params": [head], "weight_decay": 0.0}], lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.05)
    for k in range(steps):
        _grads(ps, head, 20 + k, 3.0 if k == 0 else 0.01)
        if fused_clip:
            opt.step(clip=(ps, 5.0))
        else:
            torch.nn.utils.clip_grad_norm_(ps, 5.0)
            opt.step()
    st = [opt.state[p] for p in ps + [head]]
    return ([p.detach().cpu() for p in ps + [head]], [s_["exp_avg"].cpu() for s_ in st], [s_["exp_avg_sq"].cpu() for s_ in st],
            [int(s_["step"]) for s_ in st])


def test_adamw_falls_back_to_torch_on_cpu_tensors():
    from frhip.optim import AdamW
    a, b = _run_adamw(torch.optim.AdamW, "cpu"), _run_adamw(AdamW, "cpu", fused_clip=True)
    for x, y in zip(a[0] + a[1] + a[2], b[0] + b[1] + b[2]):
        assert torch.equal(x, y)
    assert a[3] == b[3]


@pytest.mark.gpu
def test_adamw_matches_torch_adamw_and_clip_grad_norm():
    from frhip.optim import AdamW
    ref = _run_adamw(torch.optim.AdamW, "cuda")
    fell_back = []
    orig = torch.optim.AdamW.step
    torch.optim.AdamW.step = lambda self, *a, **k: (fell_back.append(1), orig(self, *a, **k))[1]
    try:
        got = _run_adamw(AdamW, "cuda", fused_clip=True)
    finally:
        torch.optim.AdamW.step = orig
    assert not fell_back, "the fused kernels must take these parameters (no silent torch fallback)"
    assert got[3] == ref[3] == [3] * len(ref[3])
    for x, y in zip(got[0] + got[1] + got[2], ref[0] + ref[1] + ref[2]):
        np.testing.assert_allclose(x.numpy(), y.numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_sgd_step_group_early_equals_plain_step():
    """the last (unclipped) parameter group updated ahead of step(), on another stream: same parameters and momentum buffers as
    one step() over everything -- when step() runs its fused kernels for the rest, and when it has to fall back to torch's own
    step (a clip set that cuts through a group) and must leave the early group alone"""
    from frhip.optim import SGD
    for fused in (True, False):
        torch.manual_seed(3)
        a = [torch.randn(300, 40, device="cuda"), torch.randn(77, device="cuda")]
        h = [torch.randn(1000, 64, device="cuda")]
        models = []
        for _ in range(2):
            pa = [torch.nn.Parameter(t.clone()) for t in a]
            ph = [torch.nn.Parameter(t.clone()) for t in h]
            models.append((pa, ph, SGD([{"params": pa}, {"params": ph}], lr=0.1, momentum=0.9, weight_decay=5e-4)))
        side = torch.cuda.Stream()
        for it in range(3):
            grads = [torch.randn_like(t) * (it + 1) for t in a + h]
            for k, (pa, ph, opt) in enumerate(models):
                for p, g in zip(pa + ph, grads):
                    p.grad = g.clone()
                if k == 0:
                    assert opt.step_group_early(1, side) is True
                    assert opt.step_group_early(1, side) is False          # once per step
                opt.step(clip=(pa if fused else pa[:1], 5.0))             # pa[:1] cuts through group 0 -> torch's step
        torch.cuda.synchronize()
        for p, q in zip(models[0][0] + models[0][1], models[1][0] + models[1][1]):
            mp, mq = models[0][2].state[p]["momentum_buffer"], models[1][2].state[q]["momentum_buffer"]
            if fused:                       # same kernel on both sides: bit-identical
                assert torch.equal(p.data, q.data) and torch.equal(mp, mq)
            else:                           # model 1 went through torch's own arithmetic for every group
                torch.testing.assert_close(p.data, q.data, rtol=1e-5, atol=1e-6)
                torch.testing.assert_close(mp, mq, rtol=1e-5, atol=1e-6)
        pa, ph, opt = models[0]
        for p in pa + ph:
            p.grad = torch.ones_like(p)
        opt.step_group_early(1, side)
        with pytest.raises(RuntimeError):
            opt.step(clip=(pa + ph, 5.0))                                   # a group updated early must not be in the clip set


@pytest.mark.gpu
def test_early_update_marker_does_not_survive_a_skipped_step():
    """backward() armed and ran the early head update, but step() never came (exception, skipped step).  zero_grad() of
    the next step joins the side stream and forgets the marker, so that step's early update runs again and step() does not skip a
    group nobody updated -- against a second optimizer that simply never used the early path."""
    from frhip.optim import SGD
    torch.manual_seed(4)
    a, h = [torch.randn(64, 32, device="cuda")], [torch.randn(500, 64, device="cuda")]
    pa, ph = [torch.nn.Parameter(a[0].clone())], [torch.nn.Parameter(h[0].clone())]
    qa, qh = [torch.nn.Parameter(a[0].clone())], [torch.nn.Parameter(h[0].clone())]
    opt = SGD([{"params": pa}, {"params": ph}], lr=0.1, momentum=0.9, weight_decay=5e-4)
    ref = SGD([{"params": qa}, {"params": qh}], lr=0.1, momentum=0.9, weight_decay=5e-4)
    side = torch.cuda.Stream()
    g0 = [torch.randn_like(a[0]), torch.randn_like(h[0])]
    for p, g in zip(pa + ph, g0):
        p.grad = g.clone()
    assert opt.step_group_early(1, side) is True          # the head group is updated ...
    with torch.no_grad():                                  # ... the reference takes the same head-only update
        for p, g in zip(qh, g0[1:]):
            p.grad = g.clone()
        for p in qa:
            p.grad = None
    ref.step()
    opt.zero_grad()                                        # step() is skipped; the next step begins
    assert not opt._early
    ref.zero_grad()
    g1 = [torch.randn_like(a[0]), torch.randn_like(h[0])]
    for p, q, g in zip(pa + ph, qa + qh, g1):
        p.grad, q.grad = g.clone(), g.clone()
    assert opt.step_group_early(1, side) is True          # not refused as "already done"
    opt.step(clip=(pa, 5.0))
    ref.step(clip=(qa, 5.0))
    torch.cuda.synchronize()
    for p, q in zip(pa + ph, qa + qh):
        torch.testing.assert_close(p.data, q.data, rtol=1e-6, atol=1e-7)
