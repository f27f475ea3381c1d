"""bf16 acceptance on the HEADLINE network (BASELINE cfg 2: IR-50-layout ResNet50 + ArcFace head, bf16 MFMA compute):

  * one whole training step of the drop-in Model in bf16 against the oracle's fp32 step on the same inputs
    (/root/reference/model/FR_PartialFC.py:162-193 composition): loss, every parameter gradient, the BatchNorm running
    statistics;
  * a verification proxy for the north_star's "LFW accuracy within +-0.1 %" clause: synthetic genuine / imposter pairs
    through the bf16 encoder and through the fp32-validation encoder, then pair_score -> performance_roc -> performance_acc
    (/root/reference/utils/eval.py:7-99) on both: the accuracies must agree to 0.1 percentage points.

Tolerances (stated up front): loss 2e-2 relative; per-tensor gradient cosine >= 0.99 for every tensor with more than 10 000
elements (>= 0.97 for the rest, which are BatchNorm vectors of 64-512 elements); running statistics 1e-2 of the tensor's
norm; |acc(bf16) - acc(fp32)| <= 0.1."""
import os
import tempfile
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.nn.functional as F

from oracle import head_ref, recipe, resnet_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pg():
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method="file://" + os.path.join(tempfile.mkdtemp(), "pg"), rank=0, world_size=1)
    yield


def _conf(dtype, classes):
    return types.SimpleNamespace(network="ResNet50", emd_size=512, img_size=112, local_rank=0, world_size=1, sample_rate=1.0,
                                 mixed_precision=True, loss_s=30.0, loss_m=0.35, n_classes=classes, optimizer="SGD", lr=0.1,
                                 wd=5e-4, mom=0.9, loss="PartialFC", lr_scheduler=None, frhip_dtype=dtype, ckpt_path=None,
                                 test_dataset=["synt"], min_level=1, max_level=3)


def _state(seed):
    spec = resnet_ref.resnet_spec(resnet_ref.BLOCKS["ResNet50"])
    return recipe.fill_state(spec, seed)


def test_resnet50_bf16_training_step_vs_oracle(pg):
    from model.FR_PartialFC import Model, normalize
    B, C = 16, 1000
    torch.cuda.set_device(0)
    sd = _state(9101)
    W = recipe.normal(9102, (C, 512), 0.01)
    img, ids = recipe.images(9103, B), recipe.labels(9104, B, C)

    # ---- oracle (fp32, CPU): the reference's step composition up to the gradients
    blocks = resnet_ref.BLOCKS["ResNet50"]
    names = resnet_ref.trainable_names(sd)
    work = {k: v.clone() for k, v in sd.items()}
    leaves = {k: work[k].requires_grad_(True) for k in names}
    raw = resnet_ref.resnet_forward(work, img, blocks, True, 512)
    feat = F.normalize(raw)
    h = head_ref.head_all_shards([feat.detach()], [ids], [W], C, 30.0, 0.35)
    feat.backward(h["d_emb"][0])

    # ---- drop-in Model, bf16 MFMA compute
    model = Model(_conf("bf16", C), None, "train")
    model.encoder.load_state_dict(sd, strict=True)
    with torch.no_grad():
        model.loss.weight_activated.data.copy_(W.cuda())
    model.opt.zero_grad()
    model.encoder.train()
    f = normalize(model.forward(img.cuda()))
    loss = model.loss(f, ids.cuda(), model.opt)
    loss.backward()

    np.testing.assert_allclose(float(loss.detach()), float(h["loss"]), rtol=2e-2)
    got = dict(model.encoder.named_parameters())
    rows, bad = [], []
    for k in names:
        a, b = got[k].grad.detach().float().cpu().flatten().double(), leaves[k].grad.flatten().double()
        if k == "fc.bias":
            continue        # a bias in front of the training-mode bn3: analytically zero gradient (round-off on both sides)
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-300))
        ratio = float(a.norm() / (b.norm() + 1e-300))
        floor = 0.99 if a.numel() > 10000 else 0.97
        rows.append((cos, ratio, k, a.numel()))
        if cos < floor or not (0.9 < ratio < 1.1):
            bad.append("%s (%d el.): cosine %.4f (floor %.2f), norm ratio %.3f" % (k, a.numel(), cos, floor, ratio))
    rows.sort()
    print("worst gradient cosines:\n" + "\n".join("  %.5f  ratio %.3f  %s (%d)" % r for r in rows[:12]))
    assert not bad, "bf16 gradients off:\n" + "\n".join(bad)
    gw = model.loss.weight_activated.grad.float().cpu().flatten().double()
    rw = h["d_w_act"][0].flatten().double()
    assert float((gw @ rw) / (gw.norm() * rw.norm())) >= 0.99
    msd = model.encoder.state_dict()
    for k in sd:
        if k.endswith("running_mean") or k.endswith("running_var"):
            a, b = msd[k].float().cpu().double(), work[k].detach().double()
            assert float((a - b).norm() / (b.norm() + 1e-12)) <= 1e-2, k
        elif k.endswith("num_batches_tracked"):
            assert int(msd[k]) == int(work[k]) == 1


def test_bf16_vs_fp32_verification_accuracy_on_synthetic_pairs(pg):
    """Stand-in for the LFW clause: the same 3 000 synthetic pairs (genuine = one image under per-pair noise of varying
    strength, imposter = two images) through the bf16 and the fp32 encoder, metrics by the reference's pipeline."""
    from model.FR_PartialFC import Model
    from utils import eval as ev
    torch.cuda.set_device(0)
    n = 3000
    sd = _state(9201)
    gen = np.random.Generator(np.random.PCG64(9202))
    labels = (gen.random(n) < 0.5).astype(np.int64)
    sigma = gen.uniform(0.2, 2.5, size=n).astype(np.float32)
    accs, ths, scs = {}, {}, {}
    for dtype in ("fp32", "bf16"):
        model = Model(_conf(dtype, 16), None, "test")
        model.encoder.load_state_dict(sd, strict=True)
        outs = []
        for lo in range(0, n, 250):
            hi = min(n, lo + 250)
            a = recipe.images(9300 + lo, hi - lo)
            other = recipe.images(9400 + lo, hi - lo)
            lab = torch.from_numpy(labels[lo:hi])
            noisy = (a + torch.from_numpy(sigma[lo:hi]).view(-1, 1, 1, 1) * recipe.normal(9500 + lo, tuple(a.shape))).clamp_(-1, 1)
            b = torch.where(lab.view(-1, 1, 1, 1) == 1, noisy, other)
            outs.append(model.test_step((torch.stack([a, b], dim=1), lab), 0))
        res = model.test_epoch_end(outs)
        accs[dtype], ths[dtype] = res["acc"], res["eer_th"]
        e1 = np.concatenate([o["synt_embedding_1"] for o in outs])
        e2 = np.concatenate([o["synt_embedding_2"] for o in outs])
        scs[dtype] = ev.pair_score(e1, e2, labels)[2]
        del model
    print("verification proxy: acc fp32 %.3f (th %d)  bf16 %.3f (th %d)  max |dscore| %.2e"
          % (accs["fp32"], ths["fp32"], accs["bf16"], ths["bf16"], float(np.abs(scs["fp32"] - scs["bf16"]).max())))
    assert 60.0 < accs["fp32"] < 99.9, "the proxy must sit where pairs can flip (acc %.2f)" % accs["fp32"]
    assert abs(accs["bf16"] - accs["fp32"]) <= 0.1
