#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference modules (build container only).

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py [--only NAME]

The reference (/root/reference, read-only) is imported as namespace packages.  Two
container-only shims are applied, exactly as recorded in SURVEY.md section 8c / Appendix B:
  * torch.Tensor.cuda -> identity  (nets/PartialFC.py hard-codes .cuda(); no GPU here)
  * torch.distributed on gloo with a file:// rendezvous (multi-rank cases use real processes)
Inputs come from oracle/recipe.py (numpy PCG64), so fixtures store OUTPUTS (+ the few
RNG draws the reference takes from torch's CPU generator).  Nothing from the reference
is copied into the repo: fixtures are data only.
"""
import argparse
import os
import sys
import tempfile
import types

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

from oracle import recipe, resnet_ref  # noqa: E402  (spec + portable inputs only)


def _ref():
    if REF not in sys.path:
        sys.path.insert(0, REF)
    torch.Tensor.cuda = lambda self, *a, **k: self
    import nets.ArcFace as A
    import nets.PartialFC as P
    import nets.resnet as R
    return A, P, R


def _init_pg(rank, ws, path):
    dist.init_process_group("gloo", init_method="file://" + path, rank=rank, world_size=ws)


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"),
                        **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v))
                           for k, v in arrs.items()})
    print("wrote", name, "(%d arrays)" % len(arrs))


# ----------------------------------------------------------------------------- ArcFace edge cases
def gen_arcface_edge():
    A, _, _ = _ref()
    for tag, (s, m) in {"s30_m035": (30.0, 0.35), "s64_m05": (64.0, 0.5)}.items():
        theta = np.cos(np.pi - m)
        t = torch.tensor([0.3, -0.2, 0.999, -0.999, theta, np.nextafter(np.float32(theta), np.float32(1)),
                          np.nextafter(np.float32(theta), np.float32(-1)), 0.0, 1.0, -1.0],
                         dtype=torch.float32)
        n, c = t.numel() + 2, 12
        logits = recipe.normal(77, (n, c), 0.3).clamp_(-1, 1)
        labels = torch.full((n, 1), -1, dtype=torch.int64)
        for i in range(t.numel()):
            labels[i, 0] = (i * 5) % c
            logits[i, labels[i, 0]] = t[i]
        inp = logits.clone()
        out = A.ArcFace(s, m)(logits, labels)
        save("arcface_edge_" + tag, s=s, m=m, logits_in=inp, labels=labels, logits_out=out)


# ----------------------------------------------------------------------------- head, multi-rank
def _head_worker(rank, ws, path, cfg, ret):
    _, P, _ = _ref()
    _init_pg(rank, ws, path)
    C, B, D, rate, s, m = cfg["C"], cfg["B"], cfg["D"], cfg["rate"], cfg["s"], cfg["m"]
    conf = types.SimpleNamespace(emd_size=D, sample_rate=rate, mixed_precision=False, loss_s=s, loss_m=m)
    pfc = P.PartialFC(conf, C)
    W = recipe.normal(500 + rank, (pfc.num_local, D), 0.05)
    with torch.no_grad():
        if rate < 1:
            pfc.weight.copy_(W)
        else:
            pfc.weight_activated.data.copy_(W)
    dummy = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([{"params": [dummy]}, {"params": pfc.parameters()}], lr=0.1, momentum=0.9)
    emb = recipe.normal(100 + rank, (B, D)).requires_grad_(True)
    lab = recipe.labels(200 + rank, B, C)
    if cfg.get("dup_labels"):      # make two ranks share identities and repeat one inside a rank
        lab[0] = 3
        lab[1] = 3
    torch.manual_seed(1000 + rank)
    u = torch.rand(pfc.num_local) if rate < 1 else torch.zeros(0)
    torch.manual_seed(1000 + rank)
    loss = pfc(emb, lab.clone(), opt)
    loss.backward()
    idx = pfc.weight_index if rate < 1 else torch.arange(pfc.num_local)
    ret[rank] = dict(loss=loss.detach().clone(), d_emb=emb.grad.clone(),
                     d_w_act=pfc.weight_activated.grad.clone(), index=idx.clone().long(), u=u,
                     class_start=pfc.class_start, num_local=pfc.num_local, num_sample=pfc.num_sample)
    dist.destroy_process_group()


def gen_head(ws, rate, C=1003, B=6, D=128, s=30.0, m=0.35, dup=True):
    cfg = dict(C=C, B=B, D=D, rate=rate, s=s, m=m, dup_labels=dup)
    mgr = mp.Manager()
    ret = mgr.dict()
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "pg")
        if ws == 1:
            _head_worker(0, 1, path, cfg, ret)
        else:
            mp.spawn(_head_worker, args=(ws, path, cfg, ret), nprocs=ws, join=True)
    arrs = dict(C=C, B=B, D=D, rate=rate, s=s, m=m, ws=ws, dup=int(dup))
    for r in range(ws):
        for k, v in ret[r].items():
            arrs["r%d_%s" % (r, k)] = v
    save("head_ws%d_rate%s" % (ws, str(rate).replace(".", "")), **arrs)


# ----------------------------------------------------------------------------- dist CE alone
def gen_distce():
    _, P, _ = _ref()
    with tempfile.TemporaryDirectory() as td:
        _init_pg(0, 1, os.path.join(td, "pg"))
        z = (recipe.normal(31, (7, 19)) * 8).requires_grad_(True)
        lab = recipe.labels(32, 7, 19).view(-1, 1)
        lab[2, 0] = -1
        zin = z.detach().clone()
        loss = P.DistCrossEntropy()(z.clone(), lab)
        (loss * 2.5).backward()
        save("distce_ws1", z=zin, labels=lab, loss=loss.detach(), grad=z.grad, upstream=2.5)
        dist.destroy_process_group()


# ----------------------------------------------------------------------------- BasicBlock
def gen_basicblock():
    _, _, R = _ref()
    for tag, (cin, cout, stride, hw) in {"s1": (64, 64, 1, 8), "s2": (64, 128, 2, 8)}.items():
        ds = None
        if stride != 1 or cin != cout:
            ds = torch.nn.Sequential(R.conv1x1(cin, cout, stride), torch.nn.BatchNorm2d(cout))
        blk = R.BasicBlock(cin, cout, stride, ds)
        spec = [("conv1.weight", (cin, cin, 3, 3), "conv")] + resnet_ref._bn_spec("bn1", cin) + \
               [("conv2.weight", (cout, cin, 3, 3), "conv")] + resnet_ref._bn_spec("bn2", cout)
        if ds is not None:
            spec += [("downsample.0.weight", (cout, cin, 1, 1), "conv")] + resnet_ref._bn_spec("downsample.1", cout)
        sd = recipe.fill_state(spec, 900 + stride)
        blk.load_state_dict(sd, strict=True)
        x = recipe.normal(901, (3, cin, hw, hw)).requires_grad_(True)
        g = recipe.normal(902, (3, cout, hw // stride, hw // stride))
        arrs = {}
        blk.train()
        y = blk(x)
        y.backward(g)
        arrs.update(train_out=y.detach(), train_dx=x.grad.clone())
        for k, p in blk.named_parameters():
            arrs["grad." + k] = p.grad.clone()
        for k, b in blk.named_buffers():
            arrs["after." + k] = b.clone()
        blk.load_state_dict(sd, strict=True)
        blk.eval()
        arrs["eval_out"] = blk(x.detach()).detach()
        save("basicblock_" + tag, cin=cin, cout=cout, stride=stride, hw=hw, **arrs)


# ----------------------------------------------------------------------------- whole backbone
def gen_resnet():
    _, _, R = _ref()
    conf = types.SimpleNamespace(network="ResNet18", emd_size=512)
    net = R.ResNet18(conf)
    spec = resnet_ref.resnet_spec(resnet_ref.BLOCKS["ResNet18"])
    assert [k for k, _, _ in spec] == list(net.state_dict().keys())
    sd = recipe.fill_state(spec, 4242)
    net.load_state_dict(sd, strict=True)
    x = recipe.images(4243, 4)
    g = recipe.normal(4244, (4, 512), 0.05)
    net.train()
    y = net(x)
    y.backward(g)
    arrs = dict(out=y.detach())
    for k, p in net.named_parameters():
        arrs["gsum." + k] = recipe.summary(p.grad)
        arrs["gprobe." + k] = recipe.probe(p.grad)            # elements at portable positions: a permutation cannot pass
    arrs["gfull.conv1.weight"] = net.conv1.weight.grad.clone()
    arrs["gfull.layer1.0.conv1.weight"] = net.layer1[0].conv1.weight.grad.clone()
    arrs["gprobe16k.fc.weight"] = recipe.probe(net.fc.weight.grad, 16384)
    for k, b in net.named_buffers():
        arrs["after." + k] = recipe.summary(b.float())
    save("resnet18_b4_train", **arrs)

    net.load_state_dict(sd, strict=True)
    net.eval()
    with torch.no_grad():
        save("resnet18_b4_eval", out=net(x))

    conf50 = types.SimpleNamespace(network="ResNet50", emd_size=512)
    net50 = R.Encoder(conf50)
    spec50 = resnet_ref.resnet_spec(resnet_ref.BLOCKS["ResNet50"])
    assert [k for k, _, _ in spec50] == list(net50.state_dict().keys())
    assert [tuple(s) for _, s, _ in spec50] == [tuple(v.shape) for v in net50.state_dict().values()]
    net50.load_state_dict(recipe.fill_state(spec50, 5050), strict=True)
    net50.eval()
    with torch.no_grad():
        save("resnet50_b2_eval", out=net50(recipe.images(5051, 2)), n_keys=len(spec50))


# ----------------------------------------------------------------------------- training steps
TRAIN_PROBED = ("conv1.weight", "layer2.0.downsample.0.weight", "layer3.1.conv2.weight", "layer4.1.bn2.weight", "fc.weight",
                "bn3.weight", "bn3.running_var", "bn1.running_mean")


def gen_train_steps(fresh=False):
    """fresh: a new synthetic batch every step (train_step_resnet18_c256_fresh_*): the repeated 16-image batch of the original fixtures is
    memorised after one lr-0.1 step (loss 15 -> 1e-5), so their steps 1-2 pin little; with fresh batches every step's loss is O(10) and is
    held to the same tolerance as step 0.  Also stored there: element probes (not 10-number summaries) of six backbone tensors and the head."""
    _, P, R = _ref()
    import torch.nn.functional as F
    for tag, rate in {"rate10": 1.0, "rate03": 0.3}.items():
        with tempfile.TemporaryDirectory() as td:
            _init_pg(0, 1, os.path.join(td, "pg"))
            C, B, steps = 256, 16, 3
            conf = types.SimpleNamespace(network="ResNet18", emd_size=512, sample_rate=rate,
                                         mixed_precision=False, loss_s=30.0, loss_m=0.35)
            enc = R.ResNet18(conf)
            spec = resnet_ref.resnet_spec(resnet_ref.BLOCKS["ResNet18"])
            sd = recipe.fill_state(spec, 777)
            # cfg-1 starts from the reference's init statistics for BN (gamma 1, beta 0, rm 0, rv 1)
            for k, _, kind in spec:
                if kind == "bn_w" or kind == "bn_rv":
                    sd[k].fill_(1.0)
                elif kind in ("bn_b", "bn_rm"):
                    sd[k].zero_()
            enc.load_state_dict(sd, strict=True)
            pfc = P.PartialFC(conf, C)
            W = recipe.normal(778, (C, 512), 0.01)
            with torch.no_grad():
                (pfc.weight if rate < 1 else pfc.weight_activated.data).copy_(W)
            opt = torch.optim.SGD([{"params": enc.parameters()}, {"params": pfc.parameters()}],
                                  lr=0.1, momentum=0.9, weight_decay=5e-4)
            img = recipe.images(779, B)
            ids = recipe.labels(780, B, C)
            arrs = dict(C=C, B=B, steps=steps, rate=rate, lr=0.1, momentum=0.9, wd=5e-4)
            losses, gnorms, us = [], [], []
            for st in range(steps):
                if fresh:
                    img, ids = recipe.images(779 + 10 * st, B), recipe.labels(780 + 10 * st, B, C)
                opt.zero_grad()
                enc.train()
                feat = F.normalize(enc(img))
                torch.manual_seed(3000 + st)
                us.append(torch.rand(C) if rate < 1 else torch.zeros(0))
                torch.manual_seed(3000 + st)
                loss = pfc(feat, ids.clone(), opt)
                loss.backward()
                gn = torch.nn.utils.clip_grad_norm_(enc.parameters(), 5)
                opt.step()
                losses.append(loss.detach().clone())
                gnorms.append(gn.detach().clone())
                if rate < 1:
                    arrs["index_step%d" % st] = pfc.weight_index.clone()
            pfc.update() if rate < 1 else None
            arrs.update(losses=torch.stack(losses), grad_norms=torch.stack(gnorms), u=torch.stack(us))
            for k, v in enc.state_dict().items():
                arrs["after." + k] = recipe.summary(v.float())
            wfin = pfc.weight if rate < 1 else pfc.weight_activated.data
            arrs["after.head_weight"] = recipe.summary(wfin)
            if fresh:
                for k in TRAIN_PROBED:
                    arrs["probe." + k] = recipe.probe(enc.state_dict()[k].float())
                arrs["probe.head_weight"] = recipe.probe(wfin, 4096)
            save("train_step_resnet18_c256_" + ("fresh_" if fresh else "") + tag, **arrs)
            dist.destroy_process_group()


# ----------------------------------------------------------------------------- training steps, two ranks (DDP + PartialFC)
def _train_ws2_worker(rank, ws, path, rate, out_dir):
    """The reference composition of model/FR_PartialFC.py:98, :162-193 at world size 2 on gloo/CPU: torch DDP around the reference
    encoder (broadcast_buffers=False, find_unused_parameters=True), the reference PartialFC shard, one SGD over both."""
    _, P, R = _ref()
    import torch.nn.functional as F
    from torch.nn.parallel import DistributedDataParallel as DDP
    _init_pg(rank, ws, path)
    C, B, steps = 256, 8, 3
    conf = types.SimpleNamespace(network="ResNet18", emd_size=512, sample_rate=rate, mixed_precision=False, loss_s=30.0, loss_m=0.35)
    enc = R.ResNet18(conf)
    spec = resnet_ref.resnet_spec(resnet_ref.BLOCKS["ResNet18"])
    sd = recipe.fill_state(spec, 777)
    for k, _, kind in spec:
        if kind == "bn_w" or kind == "bn_rv":
            sd[k].fill_(1.0)
        elif kind in ("bn_b", "bn_rm"):
            sd[k].zero_()
    enc.load_state_dict(sd, strict=True)
    enc = DDP(enc, broadcast_buffers=False, find_unused_parameters=True)
    pfc = P.PartialFC(conf, C)
    W = recipe.normal(778, (C, 512), 0.01)[pfc.class_start:pfc.class_start + pfc.num_local]
    with torch.no_grad():
        (pfc.weight if rate < 1 else pfc.weight_activated.data).copy_(W)
    opt = torch.optim.SGD([{"params": enc.parameters()}, {"params": pfc.parameters()}], lr=0.1, momentum=0.9, weight_decay=5e-4)
    img = recipe.images(779 + rank, B)
    ids = recipe.labels(780 + rank, B, C)
    arrs = dict(class_start=pfc.class_start, num_local=pfc.num_local, num_sample=pfc.num_sample)
    losses, gnorms = [], []
    for st in range(steps):
        opt.zero_grad()
        enc.train()
        feat = F.normalize(enc(img))
        torch.manual_seed(3000 + st + 50 * rank)
        loss = pfc(feat, ids.clone(), opt)
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(enc.parameters(), 5)
        opt.step()
        losses.append(loss.detach().clone())
        gnorms.append(gn.detach().clone())
        if rate < 1:
            arrs["index_step%d" % st] = pfc.weight_index.clone()
    if rate < 1:
        pfc.update()
    arrs.update(losses=torch.stack(losses), grad_norms=torch.stack(gnorms))
    for k, v in enc.module.state_dict().items():
        arrs["after." + k] = recipe.summary(v.float())
    arrs["after.head_weight"] = recipe.summary(pfc.weight if rate < 1 else pfc.weight_activated.data)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    dist.destroy_process_group()


def gen_train_steps_ws2():
    for tag, rate in {"rate10": 1.0, "rate03": 0.3}.items():
        with tempfile.TemporaryDirectory() as td:
            mp.spawn(_train_ws2_worker, args=(2, os.path.join(td, "pg"), rate, td), nprocs=2, join=True)
            arrs = dict(C=256, B=8, steps=3, rate=rate, lr=0.1, momentum=0.9, wd=5e-4, ws=2)
            for r in range(2):
                for k, v in np.load(os.path.join(td, "rank%d.npz" % r)).items():
                    arrs["r%d_%s" % (r, k)] = v
            save("train_step_resnet18_c256_ws2_" + tag, **arrs)


# ----------------------------------------------------------------------------- AdamW flavour (the reference's shipped recipe, main/train.sh:12)
ADAMW = dict(lr=5e-4, wd=5e-4, eps=1e-8, betas=(0.9, 0.999))      # configs/ms1m_arcface_122.py:222-224, main/train.sh:12


def _tensor_step(opt):
    """Container-only shim: PartialFCAdamW.sample() stores a python int in optimizer.state[...]['step'] (nets/PartialFC.py:327);
    torch >= 2 wants a tensor there.  Same value, tensor type."""
    for st in opt.state.values():
        if "step" in st and not torch.is_tensor(st["step"]):
            st["step"] = torch.tensor(float(st["step"]))


def _head_adamw_worker(rank, ws, path, cfg, out_dir):
    _, P, _ = _ref()
    _init_pg(rank, ws, path)
    C, B, D, rate, steps = cfg["C"], cfg["B"], cfg["D"], cfg["rate"], cfg["steps"]
    conf = types.SimpleNamespace(emd_size=D, sample_rate=rate, mixed_precision=False, loss_s=30.0, loss_m=0.35)
    pfc = P.PartialFCAdamW(conf, C)
    W = recipe.normal(500 + rank, (pfc.num_local, D), 0.05)
    with torch.no_grad():
        (pfc.weight if rate < 1 else pfc.weight_activated.data).copy_(W)
    dummy = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([{"params": [dummy]}, {"params": pfc.parameters()}], lr=ADAMW["lr"], weight_decay=ADAMW["wd"],
                            eps=ADAMW["eps"], betas=ADAMW["betas"])
    arrs = dict(class_start=pfc.class_start, num_local=pfc.num_local, num_sample=pfc.num_sample)
    for st in range(steps):
        opt.zero_grad()
        emb = recipe.normal(100 + rank + 10 * st, (B, D)).requires_grad_(True)
        lab = recipe.labels(200 + rank + 10 * st, B, C)
        lab[0] = 3
        lab[1] = 3
        torch.manual_seed(1000 + rank + 100 * st)
        loss = pfc(emb, lab.clone(), opt)
        loss.backward()
        _tensor_step(opt)
        opt.step()
        arrs["loss_step%d" % st] = loss.detach().clone()
        arrs["d_emb_step%d" % st] = emb.grad.clone()
        arrs["index_step%d" % st] = (pfc.weight_index if rate < 1 else torch.arange(pfc.num_local)).clone().long()
    pfc.update()
    if rate < 1:
        arrs.update(weight=pfc.weight.clone(), exp_avg=pfc.weight_exp_avg.clone(), exp_avg_sq=pfc.weight_exp_avg_sq.clone())
    else:
        stt = opt.state[pfc.weight_activated]
        arrs.update(weight=pfc.weight_activated.data.clone(), exp_avg=stt["exp_avg"].clone(), exp_avg_sq=stt["exp_avg_sq"].clone())
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    dist.destroy_process_group()


def gen_head_adamw(ws, rate=0.3, C=503, B=6, D=128, steps=3):
    cfg = dict(C=C, B=B, D=D, rate=rate, steps=steps)
    with tempfile.TemporaryDirectory() as td:
        if ws == 1:
            _head_adamw_worker(0, 1, os.path.join(td, "pg"), cfg, td)
        else:
            mp.spawn(_head_adamw_worker, args=(ws, os.path.join(td, "pg"), cfg, td), nprocs=ws, join=True)
        arrs = dict(C=C, B=B, D=D, rate=rate, steps=steps, ws=ws, s=30.0, m=0.35, lr=ADAMW["lr"], wd=ADAMW["wd"], eps=ADAMW["eps"],
                    betas=np.asarray(ADAMW["betas"]))
        for r in range(ws):
            for k, v in np.load(os.path.join(td, "rank%d.npz" % r)).items():
                arrs["r%d_%s" % (r, k)] = v
        save("head_adamw_ws%d_rate%s" % (ws, str(rate).replace(".", "")), **arrs)


def gen_train_steps_adamw():
    """model/FR_PartialFC.py:162-193 with the AdamW branch of configure_optimizers (:436-442) and PartialFCAdamW, rate 0.3."""
    _, P, R = _ref()
    import torch.nn.functional as F
    for tag, rate in {"rate03": 0.3, "rate10": 1.0}.items():
        with tempfile.TemporaryDirectory() as td:
            _init_pg(0, 1, os.path.join(td, "pg"))
            C, B, steps = 256, 16, 3
            conf = types.SimpleNamespace(network="ResNet18", emd_size=512, sample_rate=rate, mixed_precision=False, loss_s=30.0, loss_m=0.35)
            enc = R.ResNet18(conf)
            spec = resnet_ref.resnet_spec(resnet_ref.BLOCKS["ResNet18"])
            sd = recipe.fill_state(spec, 777)
            for k, _, kind in spec:
                if kind == "bn_w" or kind == "bn_rv":
                    sd[k].fill_(1.0)
                elif kind in ("bn_b", "bn_rm"):
                    sd[k].zero_()
            enc.load_state_dict(sd, strict=True)
            pfc = P.PartialFCAdamW(conf, C)
            W = recipe.normal(778, (C, 512), 0.01)
            with torch.no_grad():
                (pfc.weight if rate < 1 else pfc.weight_activated.data).copy_(W)
            opt = torch.optim.AdamW([{"params": enc.parameters()}, {"params": pfc.parameters()}], lr=ADAMW["lr"], weight_decay=ADAMW["wd"],
                                    eps=ADAMW["eps"], betas=ADAMW["betas"])
            arrs = dict(C=C, B=B, steps=steps, rate=rate, lr=ADAMW["lr"], wd=ADAMW["wd"], eps=ADAMW["eps"], betas=np.asarray(ADAMW["betas"]))
            losses, gnorms = [], []
            for st in range(steps):
                # a fresh batch per step: one Adam step memorises a repeated 16-image batch (loss 15 -> 1e-6) and later steps would pin nothing
                img = recipe.images(779 + 10 * st, B)
                ids = recipe.labels(780 + 10 * st, B, C)
                opt.zero_grad()
                enc.train()
                feat = F.normalize(enc(img))
                torch.manual_seed(3000 + st)
                loss = pfc(feat, ids.clone(), opt)
                loss.backward()
                gn = torch.nn.utils.clip_grad_norm_(enc.parameters(), 5)
                if st == 0:
                    for k, p_ in enc.named_parameters():
                        if k in TRAIN_PROBED:
                            arrs["grad0." + k] = recipe.probe(p_.grad)      # the (clipped) gradients of step 0: sign-insensitive pin
                _tensor_step(opt)
                opt.step()
                losses.append(loss.detach().clone())
                gnorms.append(gn.detach().clone())
                if rate < 1:
                    arrs["index_step%d" % st] = pfc.weight_index.clone()
            if rate < 1:
                pfc.update()
                arrs.update({"after.head_weight": recipe.probe(pfc.weight, 4096), "after.head_exp_avg": recipe.probe(pfc.weight_exp_avg, 4096),
                             "after.head_exp_avg_sq": recipe.probe(pfc.weight_exp_avg_sq, 4096)})
            else:
                stt = opt.state[pfc.weight_activated]
                arrs.update({"after.head_weight": recipe.probe(pfc.weight_activated.data, 4096), "after.head_exp_avg": recipe.probe(stt["exp_avg"], 4096),
                             "after.head_exp_avg_sq": recipe.probe(stt["exp_avg_sq"], 4096)})
            arrs.update(losses=torch.stack(losses), grad_norms=torch.stack(gnorms))
            esd = enc.state_dict()
            for k in TRAIN_PROBED:
                arrs["after." + k] = recipe.probe(esd[k].float())
            for k, p_ in enc.named_parameters():
                if k in TRAIN_PROBED:
                    arrs["exp_avg." + k] = recipe.probe(opt.state[p_]["exp_avg"])
                    arrs["exp_avg_sq." + k] = recipe.probe(opt.state[p_]["exp_avg_sq"])
            save("train_step_resnet18_c256_adamw_" + tag, **arrs)
            dist.destroy_process_group()


FULL_SWIN34 = ("conv1.weight", "layer2.0.weight", "layer3.0.weight", "layer3.1.attn.qkv.weight", "layer3.1.attn.cpb_mlp.0.weight", "layer3.1.attn.cpb_mlp.2.weight",
               "layer3.1.attn.logit_scale", "layer3.1.attn.q_bias", "bn2.weight",
               "bn3.weight", "bn3.bias")
FULL_ALTERNET50 = ("conv1.weight", "layer1.0.conv1.weight", "layer2.2.attn.qkv.weight", "layer2.2.attn.cpb_mlp.0.weight", "layer2.2.attn.cpb_mlp.2.weight",
                   "layer2.3.attn.qkv.weight", "layer2.3.attn.logit_scale", "layer2.1.conv2.weight", "layer4.0.downsample.0.weight", "bn2.weight", "bn3.weight",
                   "bn3.bias")


def _whole_net_train(name, net, spec, fill_special, seed, x, full):
    """One training-mode forward/backward of a whole reference backbone (BatchNorm in batch-statistics mode, stochastic depth = identity
    through the DropPath stub, tail Dropout p = 0: RNG-free), batch 8 so that the tail BatchNorm1d is well conditioned.  Stored: the
    embeddings, a probe (sum, l2, 256 elements at portable positions) of EVERY parameter gradient, full tensors of the named ones and of
    every attention block's first-in-stage qkv / cpb gradients that fit, probes of the running statistics."""
    assert [k for k, _, _ in spec] == list(net.state_dict().keys()), name
    sd = fill_special(recipe.fill_state(spec, seed), spec)
    net.load_state_dict(sd, strict=True)
    net.train()
    net.dropout.p = 0.0
    y = net(x)
    y.backward(recipe.normal(seed + 2, tuple(y.shape), 0.05))
    arrs = dict(out=y.detach(), batch=x.shape[0], seed=seed)
    for k, p in net.named_parameters():
        arrs["gprobe." + k] = recipe.probe(p.grad)
        if k in full:
            assert p.numel() <= (1 << 20), k
            arrs["gfull." + k] = p.grad.clone()
    assert all(("gfull." + k) in arrs for k in full), [k for k in full if ("gfull." + k) not in arrs]
    arrs["gprobe16k.fc.weight"] = recipe.probe(net.fc.weight.grad, 16384)
    for k, b in net.named_buffers():
        if "running" in k:
            arrs["after." + k] = recipe.probe(b.float())
    save(name, **arrs)


# ----------------------------------------------------------------------------- SwinV2-style backbone
def _swin_ref():
    """reference nets/SwinV2.py needs three symbols of timm.models.layers: stubbed, container only"""
    _ref()
    import torch.nn as nn
    if "timm.models.layers" not in sys.modules:
        class DropPath(nn.Module):
            def __init__(self, p=0.0):
                super().__init__()
            def forward(self, x):
                return x
        ml = types.ModuleType("timm.models.layers")
        ml.DropPath, ml.trunc_normal_ = DropPath, nn.init.trunc_normal_
        ml.to_2tuple = lambda x: tuple(x) if isinstance(x, (tuple, list)) else (x, x)
        sys.modules["timm"] = types.ModuleType("timm")
        sys.modules["timm.models"] = types.ModuleType("timm.models")
        sys.modules["timm.models.layers"] = ml
    import nets.SwinV2 as S
    return S


def gen_swin():
    from oracle import swin_ref
    S = _swin_ref()
    # --- one transformer block, training-mode BN, fwd + bwd
    for tag, (c, heads, hw) in {"c128h4": (128, 4, 14), "c512h16": (512, 16, 7)}.items():
        blk = S.SwinTransformerBlock(c, c, heads=heads)
        spec = swin_ref.block_spec("blk", c, heads)
        sd = swin_ref.fill_special(recipe.fill_state(spec, 6100 + heads), spec)
        blk.load_state_dict({k[4:]: v for k, v in sd.items()}, strict=True)
        x = recipe.normal(6101, (3, c, hw, hw)).requires_grad_(True)
        g = recipe.normal(6102, (3, c, hw, hw))
        blk.train()
        y = blk(x)
        y.backward(g)
        arrs = dict(c=c, heads=heads, hw=hw, out=y.detach(), dx=x.grad.clone())
        for k, p in blk.named_parameters():
            arrs["grad." + k] = p.grad.clone() if p.numel() < 20000 else recipe.summary(p.grad)
        for k, b in blk.named_buffers():
            if "running" in k:
                arrs["after." + k] = b.clone()
        save("swin_block_" + tag, **arrs)
    # --- whole nets
    for name, seed in (("Swin18", 6200), ("Swin34", 6300)):
        conf = types.SimpleNamespace(network=name, emd_size=512)
        net = getattr(S, name)(conf)
        spec = swin_ref.swin_spec(name)
        assert [k for k, _, _ in spec] == list(net.state_dict().keys()), name
        assert [tuple(s) for _, s, _ in spec] == [tuple(v.shape) for v in net.state_dict().values()]
        sd = swin_ref.fill_special(recipe.fill_state(spec, seed), spec)
        net.load_state_dict(sd, strict=True)
        x = recipe.images(seed + 1, 2)
        net.eval()
        with torch.no_grad():
            arrs = dict(eval_out=net(x), n_keys=len(spec))
        if name == "Swin18":
            net.load_state_dict(sd, strict=True)
            net.train()
            net.dropout.p = 0.0          # RNG-free training fixture (the reference's Dropout(0.5) is statistical only)
            y = net(x)
            y.backward(recipe.normal(seed + 2, (2, 512), 0.05))
            arrs["train_out"] = y.detach()
            for k, p in net.named_parameters():
                arrs["gsum." + k] = recipe.summary(p.grad)
            for k, b in net.named_buffers():
                if "running" in k:
                    arrs["after." + k] = recipe.summary(b.float())
        save(name.lower() + "_b2", **arrs)
    _whole_net_train("swin34_b8_train", S.Swin34(types.SimpleNamespace(network="Swin34", emd_size=512)), swin_ref.swin_spec("Swin34"),
                     swin_ref.fill_special, 6400, recipe.images(6401, 8), FULL_SWIN34)


# ----------------------------------------------------------------------------- hybrid AlterNet backbone
def gen_alternet():
    from oracle import alternet_ref
    _swin_ref()        # installs the timm stub
    sys.modules.setdefault("einops", types.ModuleType("einops"))
    if not hasattr(sys.modules["einops"], "rearrange"):
        sys.modules["einops"].rearrange = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("unused"))
        sys.modules["einops"].repeat = sys.modules["einops"].rearrange
    import torch.nn as nn
    import nets.AlterNet_SwinV2_FAN as A
    # --- a (W-MSA, SW-MSA) pair, training-mode BN, DropPath disabled, fwd + bwd
    for tag, (c, heads, ws, res) in {"c128_w6": (128, 4, 6, 12), "c512_w3": (512, 16, 3, 6)}.items():
        arrs = dict(c=c, heads=heads, ws=ws, res=res)
        x = recipe.normal(7101, (2, c, res, res)).requires_grad_(True)
        g = recipe.normal(7102, (2, c, res, res))
        blks, sds = [], []
        for j, shift in enumerate((0, ws // 2)):
            blk = A.SwinTransformerBlock(c, c, heads=heads, input_resolution=(res, res), window_size=ws, shift_size=shift)
            blk.drop_path = nn.Identity()
            spec = alternet_ref.attn_block_spec("blk", c, heads, ws, shift, res)
            sd = alternet_ref.fill_special(recipe.fill_state(spec, 7000 + 10 * heads + j), spec)
            blk.load_state_dict({k[4:]: v for k, v in sd.items()}, strict=True)
            blk.train()
            blks.append(blk)
        y = blks[1](blks[0](x))
        y.backward(g)
        arrs.update(out=y.detach(), dx=x.grad.clone())
        for j, blk in enumerate(blks):
            for k, p in blk.named_parameters():
                arrs["b%d.grad.%s" % (j, k)] = p.grad.clone() if p.numel() < 20000 else recipe.summary(p.grad)
            for k, b in blk.named_buffers():
                if "running" in k:
                    arrs["b%d.after.%s" % (j, k)] = b.clone()
        save("alternet_pair_" + tag, **arrs)
    # --- whole net, eval, 192x192
    conf = types.SimpleNamespace(network="AlterNet50", emd_size=512, img_size=192)
    net = A.AlterNet50(conf)
    spec = alternet_ref.alter_spec("AlterNet50")
    assert [k for k, _, _ in spec] == list(net.state_dict().keys())
    assert [tuple(s) for _, s, _ in spec] == [tuple(v.shape) for v in net.state_dict().values()]
    sd = alternet_ref.fill_special(recipe.fill_state(spec, 7300), spec)
    net.load_state_dict(sd, strict=True)
    net.eval()
    with torch.no_grad():
        save("alternet50_b2_eval", out=net(recipe.images(7301, 2, 192, 192)), n_keys=len(spec))
    _whole_net_train("alternet50_b8_train", A.AlterNet50(conf), spec, alternet_ref.fill_special, 7400, recipe.images(7401, 8, 192, 192), FULL_ALTERNET50)


# ----------------------------------------------------------------------------- verification metrics
def gen_eval():
    _ref()
    nb = types.ModuleType("numba")          # utils/eval.py needs numba (absent): identity decorators, container only
    nb.njit = lambda *a, **k: (lambda f: f)
    nb.prange = range
    sys.modules.setdefault("numba", nb)
    import utils.eval as E
    n, d = 600, 512
    base = recipe.normal(8101, (n, d))
    other = recipe.normal(8102, (n, d))
    labels = (recipe.rng(8103).random(n) < 0.5).astype(np.int64)
    e1 = torch.nn.functional.normalize(base)
    mix = torch.from_numpy(np.where(labels[:, None] == 1, 0.12, 0.0).astype(np.float32))
    e2 = torch.nn.functional.normalize(mix * base + (1 - mix) * other + 0.0 * recipe.normal(8104, (n, d)))
    e1n, e2n = e1.numpy(), e2.numpy()
    hg, hi, scores = E.pair_score(e1n, e2n, labels)
    roc, eer_th = E.performance_roc(hg, hi, min_level=1, max_level=3)
    acc = E.performance_acc(scores, labels, eer_th)
    idx = np.array([int((1e5 - 1.) * s) for s in scores], dtype=np.int64)
    save("eval_pairs", n=n, d=d, labels=labels, scores=scores, hist_idx=idx, hist_genuine=hg, hist_imposter=hi,
         eer_th=eer_th, acc=acc, roc=np.array(roc))


def gen_cross_eval():
    _ref()
    nb = types.ModuleType("numba")
    nb.njit = lambda *a, **k: (lambda f: f)
    nb.prange = range
    sys.modules.setdefault("numba", nb)
    import utils.eval as E
    ids, per, d = 20, 4, 128
    centres = recipe.normal(8201, (ids, d))
    emb = torch.nn.functional.normalize(centres.repeat_interleave(per, 0) * 0.35 + recipe.normal(8202, (ids * per, d)))
    labels = np.repeat(np.arange(ids), per).astype(np.int64)
    perm = recipe.rng(8203).permutation(ids * per)
    emb, labels = emb[torch.from_numpy(perm)], labels[perm]
    hg, hi, scores, plab = E.cross_score(emb.numpy(), labels)
    roc, eer_th = E.performance_roc(hg, hi, min_level=1, max_level=3)
    acc = E.performance_acc(scores, plab, eer_th)
    idx = np.array([int((1e5 - 1.) * s) for s in scores], dtype=np.int64)
    save("cross_eval", ids=ids, per=per, d=d, labels=labels, scores=scores, pair_labels=plab, hist_idx=idx, hist_genuine=hg,
         hist_imposter=hi, eer_th=eer_th, acc=acc, roc=np.array(roc))


# ----------------------------------------------------------------------------- lr schedule
def gen_scheduler():
    _ref()
    import utils.scheduler as S
    p = torch.nn.Parameter(torch.zeros(1))
    out = {}
    for tag, kw in {"c10_w2": dict(first_cycle_steps=10, warmup_steps=2, min_lr=0.001, max_lr=0.1),
                    "c15_w0": dict(first_cycle_steps=15, warmup_steps=0, min_lr=1e-5, max_lr=0.05),
                    "c6_w1_m2_g05": dict(first_cycle_steps=6, warmup_steps=1, min_lr=0.001, max_lr=0.1,
                                         cycle_mult=2.0, gamma=0.5)}.items():
        opt = torch.optim.SGD([p], lr=0.1)
        sch = S.CosineAnnealingWarmupRestarts(opt, **kw)
        lrs = []
        for _ in range(40):
            lrs.append(opt.param_groups[0]["lr"])
            sch.step()
        out[tag] = np.asarray(lrs, dtype=np.float64)
    save("scheduler_lrs", **out)


# ----------------------------------------------------------------------------- the shipped recipe end to end (main/train.sh:12)
def gen_recipe_alternet50():
    """main/train.sh:12 -- `--sample_rate 0.3 --optimizer AdamW --network AlterNet50 --lr 5e-4` -- composed as model/FR_PartialFC.py:162-193 does:
    AlterNet50 @192 encoder, F.normalize, PartialFCAdamW(rate 0.3), AdamW over [encoder, head], clip_grad_norm_(encoder, 5); two steps on fresh
    batches of 8.  RNG-free: tail Dropout p = 0, DropPath = identity (the timm stub)."""
    from oracle import alternet_ref
    _, P, _ = _ref()
    _swin_ref()
    sys.modules.setdefault("einops", types.ModuleType("einops"))
    if not hasattr(sys.modules["einops"], "rearrange"):
        sys.modules["einops"].rearrange = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("unused"))
        sys.modules["einops"].repeat = sys.modules["einops"].rearrange
    import torch.nn.functional as F
    import nets.AlterNet_SwinV2_FAN as A
    with tempfile.TemporaryDirectory() as td:
        _init_pg(0, 1, os.path.join(td, "pg"))
        C, B, steps, rate = 256, 8, 2, 0.3
        conf = types.SimpleNamespace(network="AlterNet50", emd_size=512, img_size=192, sample_rate=rate, mixed_precision=False, loss_s=30.0, loss_m=0.35)
        enc = A.AlterNet50(conf)
        spec = alternet_ref.alter_spec("AlterNet50")
        enc.load_state_dict(alternet_ref.fill_special(recipe.fill_state(spec, 9100), spec), strict=True)
        enc.dropout.p = 0.0
        pfc = P.PartialFCAdamW(conf, C)
        with torch.no_grad():
            pfc.weight.copy_(recipe.normal(9101, (C, 512), 0.01))
        opt = torch.optim.AdamW([{"params": enc.parameters()}, {"params": pfc.parameters()}], lr=ADAMW["lr"], weight_decay=ADAMW["wd"],
                                eps=ADAMW["eps"], betas=ADAMW["betas"])
        arrs = dict(C=C, B=B, steps=steps, rate=rate, lr=ADAMW["lr"], wd=ADAMW["wd"], eps=ADAMW["eps"], betas=np.asarray(ADAMW["betas"]), seed=9100)
        losses, gnorms = [], []
        names = ("conv1.weight", "layer1.0.conv1.weight", "layer2.2.attn.qkv.weight", "layer2.3.attn.cpb_mlp.2.weight", "layer3.5.conv2.weight",
                 "layer4.3.attn.proj.weight", "bn2.weight", "fc.weight", "bn3.weight")
        for st in range(steps):
            img, ids = recipe.images(9110 + 10 * st, B, 192, 192), recipe.labels(9111 + 10 * st, B, C)
            opt.zero_grad()
            enc.train()
            feat = F.normalize(enc(img))
            torch.manual_seed(9200 + st)
            loss = pfc(feat, ids.clone(), opt)
            loss.backward()
            gn = torch.nn.utils.clip_grad_norm_(enc.parameters(), 5)
            if st == 0:
                for k, p_ in enc.named_parameters():
                    if k in names:
                        arrs["grad0." + k] = recipe.probe(p_.grad)
            _tensor_step(opt)
            opt.step()
            losses.append(loss.detach().clone())
            gnorms.append(gn.detach().clone())
            arrs["index_step%d" % st] = pfc.weight_index.clone()
        pfc.update()
        arrs.update(losses=torch.stack(losses), grad_norms=torch.stack(gnorms))
        esd = enc.state_dict()
        for k in names:
            arrs["after." + k] = recipe.probe(esd[k].float())
        arrs.update({"after.head_weight": recipe.probe(pfc.weight, 4096), "after.head_exp_avg": recipe.probe(pfc.weight_exp_avg, 4096),
                     "after.head_exp_avg_sq": recipe.probe(pfc.weight_exp_avg_sq, 4096)})
        save("recipe_alternet50_adamw_rate03", **arrs)
        dist.destroy_process_group()


GENS = {
    "scheduler": gen_scheduler,
    "swin": gen_swin,
    "alternet": gen_alternet,
    "eval": gen_eval,
    "cross_eval": gen_cross_eval,
    "arcface": gen_arcface_edge,
    "distce": gen_distce,
    "head_ws1_rate10": lambda: gen_head(1, 1.0),
    "head_ws1_rate03": lambda: gen_head(1, 0.3),
    "head_ws2_rate10": lambda: gen_head(2, 1.0),
    "head_ws2_rate03": lambda: gen_head(2, 0.3),
    "head_ws8_rate01": lambda: gen_head(8, 0.1, C=4003, B=4),
    "head_ws4_rate01": lambda: gen_head(4, 0.1, C=4003, B=8),
    "basicblock": gen_basicblock,
    "resnet": gen_resnet,
    "train": gen_train_steps,
    "train_ws2": gen_train_steps_ws2,
    "head_adamw_ws1": lambda: gen_head_adamw(1),
    "head_adamw_ws2": lambda: gen_head_adamw(2),
    "train_adamw": gen_train_steps_adamw,
    "train_fresh": lambda: gen_train_steps(fresh=True),
    "recipe_alternet50": gen_recipe_alternet50,
}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    torch.set_num_threads(8)
    for name, fn in GENS.items():
        if a.only is None or a.only == name:
            fn()
