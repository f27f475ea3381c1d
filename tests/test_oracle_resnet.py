"""Pin oracle/resnet_ref.py and oracle/train_ref.py to reference-generated vectors."""
import numpy as np
import pytest
import torch

from oracle import recipe, resnet_ref, train_ref

T = torch.from_numpy


def _summary_close(got, want, rtol, atol):
    np.testing.assert_allclose(recipe.summary(got), want, rtol=rtol, atol=atol)


@pytest.mark.parametrize("tag", ["s1", "s2"])
def test_basicblock(golden, tag):
    g = golden("basicblock_" + tag)
    cin, cout, stride, hw = int(g["cin"]), int(g["cout"]), int(g["stride"]), int(g["hw"])
    has_ds = stride != 1 or cin != cout
    spec = [("conv1.weight", (cin, cin, 3, 3), "conv")] + resnet_ref._bn_spec("bn1", cin) + \
           [("conv2.weight", (cout, cin, 3, 3), "conv")] + resnet_ref._bn_spec("bn2", cout)
    if has_ds:
        spec += [("downsample.0.weight", (cout, cin, 1, 1), "conv")] + resnet_ref._bn_spec("downsample.1", cout)
    sd0 = recipe.fill_state(spec, 900 + stride)
    sd = {"blk." + k: v.clone() for k, v in sd0.items()}
    names = resnet_ref.trainable_names(sd)
    for k in names:
        sd[k].requires_grad_(True)
    x = recipe.normal(901, (3, cin, hw, hw)).requires_grad_(True)
    y = resnet_ref.basic_block(sd, "blk", x, stride, has_ds, True)
    y.backward(recipe.normal(902, y.shape))
    np.testing.assert_allclose(y.detach().numpy(), g["train_out"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(x.grad.numpy(), g["train_dx"], rtol=1e-3, atol=1e-5)
    for k in names:
        np.testing.assert_allclose(sd[k].grad.numpy(), g["grad." + k[4:]], rtol=1e-3, atol=2e-5)
    for k in sd:
        if k not in names:
            np.testing.assert_allclose(sd[k].detach().numpy(), g["after." + k[4:]], rtol=1e-5, atol=1e-6)
    sd = {"blk." + k: v.clone() for k, v in sd0.items()}
    ye = resnet_ref.basic_block(sd, "blk", x.detach(), stride, has_ds, False)
    np.testing.assert_allclose(ye.numpy(), g["eval_out"], rtol=1e-4, atol=1e-5)


def test_spec_matches_reference_key_count(golden):
    g = golden("resnet50_b2_eval")
    assert len(resnet_ref.resnet_spec(resnet_ref.BLOCKS["ResNet50"])) == int(g["n_keys"]) == 336


def test_resnet18_train_and_eval(golden):
    blocks = resnet_ref.BLOCKS["ResNet18"]
    sd0 = recipe.fill_state(resnet_ref.resnet_spec(blocks), 4242)
    x = recipe.images(4243, 4)
    g = golden("resnet18_b4_train")
    sd = {k: v.clone() for k, v in sd0.items()}
    names = resnet_ref.trainable_names(sd)
    for k in names:
        sd[k].requires_grad_(True)
    y = resnet_ref.resnet_forward(sd, x, blocks, True)
    y.backward(recipe.normal(4244, (4, 512), 0.05))
    np.testing.assert_allclose(y.detach().numpy(), g["out"], rtol=1e-3, atol=1e-4)
    for k in names:
        _summary_close(sd[k].grad, g["gsum." + k], 2e-3, 2e-5)
        np.testing.assert_allclose(recipe.probe(sd[k].grad), g["gprobe." + k], rtol=2e-3, atol=2e-5, err_msg=k)    # elementwise: no permutation passes
    for k in ("conv1.weight", "layer1.0.conv1.weight"):
        np.testing.assert_allclose(sd[k].grad.numpy(), g["gfull." + k], rtol=2e-3, atol=2e-5, err_msg=k)
    np.testing.assert_allclose(recipe.probe(sd["fc.weight"].grad, 16384), g["gprobe16k.fc.weight"], rtol=2e-3, atol=2e-5)
    for k in sd:
        if k not in names:
            _summary_close(sd[k].detach().float(), g["after." + k], 1e-4, 1e-6)
    ge = golden("resnet18_b4_eval")
    sd = {k: v.clone() for k, v in sd0.items()}
    with torch.no_grad():
        ye = resnet_ref.resnet_forward(sd, x, blocks, False)
    np.testing.assert_allclose(ye.numpy(), ge["out"], rtol=1e-3, atol=1e-4)


def test_resnet50_eval(golden):
    blocks = resnet_ref.BLOCKS["ResNet50"]
    sd = recipe.fill_state(resnet_ref.resnet_spec(blocks), 5050)
    with torch.no_grad():
        y = resnet_ref.resnet_forward(sd, recipe.images(5051, 2), blocks, False)
    np.testing.assert_allclose(y.numpy(), golden("resnet50_b2_eval")["out"], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("tag", ["rate10", "rate03"])
def test_train_steps(golden, tag):
    """BASELINE cfg 1: ResNet-18 + ArcFace head, 256 ids, fp32 CPU, world_size 1, 3 SGD steps."""
    g = golden("train_step_resnet18_c256_" + tag)
    C, B, steps, rate = int(g["C"]), int(g["B"]), int(g["steps"]), float(g["rate"])
    blocks = resnet_ref.BLOCKS["ResNet18"]
    spec = resnet_ref.resnet_spec(blocks)
    sd = recipe.fill_state(spec, 777)
    for k, _, kind in spec:
        if kind in ("bn_w", "bn_rv"):
            sd[k].fill_(1.0)
        elif kind in ("bn_b", "bn_rm"):
            sd[k].zero_()
    W = recipe.normal(778, (C, 512), 0.01)
    img, ids = recipe.images(779, B), recipe.labels(780, B, C)
    opt = train_ref.SGDState(float(g["lr"]), float(g["momentum"]), float(g["wd"]))
    for st in range(steps):
        u = [T(g["u"][st])] if rate < 1 else None
        out = train_ref.train_step(sd, W, img, ids, blocks, C, opt, sample_rate=rate, uniforms=u)
        np.testing.assert_allclose(out["loss"].item(), g["losses"][st], rtol=2e-3)
        np.testing.assert_allclose(out["grad_norm"].item(), g["grad_norms"][st], rtol=5e-3)
        if rate < 1:
            assert np.array_equal(out["index"].numpy(), g["index_step%d" % st])
    for k in ("conv1.weight", "layer2.0.downsample.0.weight", "layer4.1.bn2.weight", "fc.weight",
              "bn3.running_var", "bn1.running_mean"):
        _summary_close(sd[k].float(), g["after." + k], 5e-3, 5e-5)
    _summary_close(W, g["after.head_weight"], 5e-3, 5e-5)


@pytest.mark.parametrize("tag", ["rate10", "rate03"])
def test_train_steps_two_ranks(golden, tag):
    """The reference composition at world size 2 (torch DDP around the encoder + the class-sharded PartialFC, gloo/CPU, generated by
    tools/make_golden.py train_ws2): per-rank data, one global loss, averaged backbone gradients, per-rank head shards, 3 SGD steps."""
    g = golden("train_step_resnet18_c256_ws2_" + tag)
    C, B, steps, rate, ws = int(g["C"]), int(g["B"]), int(g["steps"]), float(g["rate"]), int(g["ws"])
    blocks = resnet_ref.BLOCKS["ResNet18"]
    spec = resnet_ref.resnet_spec(blocks)
    sds, Ws, imgs, idss, opts = [], [], [], [], []
    for r in range(ws):
        sd = recipe.fill_state(spec, 777)
        for k, _, kind in spec:
            if kind in ("bn_w", "bn_rv"):
                sd[k].fill_(1.0)
            elif kind in ("bn_b", "bn_rm"):
                sd[k].zero_()
        sds.append(sd)
        c0, nloc = int(g["r%d_class_start" % r]), int(g["r%d_num_local" % r])
        Ws.append(recipe.normal(778, (C, 512), 0.01)[c0:c0 + nloc].clone())
        imgs.append(recipe.images(779 + r, B))
        idss.append(recipe.labels(780 + r, B, C))
        opts.append(train_ref.SGDState(float(g["lr"]), float(g["momentum"]), float(g["wd"])))
    for st in range(steps):
        u = None
        if rate < 1:
            u = []
            for r in range(ws):
                torch.manual_seed(3000 + st + 50 * r)           # the draw the reference made on rank r (nets/PartialFC.py:110)
                u.append(torch.rand(int(g["r%d_num_local" % r])))
        out = train_ref.train_step_ranks(sds, Ws, imgs, idss, blocks, C, opts, sample_rate=rate, uniforms=u)
        for r in range(ws):
            # later steps start from ~1e-5 losses of a memorised batch: summation-order noise is amplified (as in the ws-1 test)
            np.testing.assert_allclose(out["loss"].item(), g["r%d_losses" % r][st], rtol=2e-3 if st == 0 else 5e-2)
            np.testing.assert_allclose(out["grad_norm"].item(), g["r%d_grad_norms" % r][st], rtol=5e-3 if st == 0 else 5e-2)
            if rate < 1:
                assert np.array_equal(out["index"][r].numpy(), g["r%d_index_step%d" % (r, st)])
    for r in range(ws):
        for k in ("conv1.weight", "layer2.0.downsample.0.weight", "layer4.1.bn2.weight", "fc.weight", "bn3.running_var", "bn1.running_mean"):
            _summary_close(sds[r][k].float(), g["r%d_after.%s" % (r, k)], 5e-3, 5e-5)
        _summary_close(Ws[r], g["r%d_after.head_weight" % r], 5e-3, 5e-5)


def _adam_close(got, want, lr, steps, err_msg=""):
    """Adam moves every element by ~lr per step whatever the size of its gradient (m/sqrt(v) = +-1 on step one): an element whose gradient
    is rounding noise takes a random direction in any two implementations.  So: nearly all probed elements agree tightly, every one of them
    stays within the distance the steps can cover, and the l2 norm agrees."""
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    d = np.abs(got[2:] - want[2:])
    assert (d <= 2e-5 + 2e-3 * np.abs(want[2:])).mean() >= 0.97, (err_msg, float((d <= 2e-5 + 2e-3 * np.abs(want[2:])).mean()))
    assert d.max() <= 2.2 * lr * steps, (err_msg, d.max())
    np.testing.assert_allclose(got[1], want[1], rtol=2e-3, err_msg=err_msg)


@pytest.mark.parametrize("tag", ["rate03", "rate10"])
def test_train_steps_adamw(golden, tag):
    """The reference's shipped recipe (main/train.sh:12: --optimizer AdamW --sample_rate 0.3 --lr 5e-4) on BASELINE cfg 1's network:
    PartialFCAdamW + torch.optim.AdamW over [encoder, head], 3 steps."""
    g = golden("train_step_resnet18_c256_adamw_" + tag)
    C, B, steps, rate, lr = int(g["C"]), int(g["B"]), int(g["steps"]), float(g["rate"]), float(g["lr"])
    blocks = resnet_ref.BLOCKS["ResNet18"]
    spec = resnet_ref.resnet_spec(blocks)
    sd = recipe.fill_state(spec, 777)
    for k, _, kind in spec:
        if kind in ("bn_w", "bn_rv"):
            sd[k].fill_(1.0)
        elif kind in ("bn_b", "bn_rm"):
            sd[k].zero_()
    W = recipe.normal(778, (C, 512), 0.01)
    opt = train_ref.AdamWState(lr, tuple(g["betas"]), float(g["eps"]), float(g["wd"]))
    for st in range(steps):
        img, ids = recipe.images(779 + 10 * st, B), recipe.labels(780 + 10 * st, B, C)      # a fresh batch per step
        u = None
        if rate < 1:
            torch.manual_seed(3000 + st)
            u = [torch.rand(C)]
        out = train_ref.train_step(sd, W, img, ids, blocks, C, opt, sample_rate=rate, uniforms=u)
        np.testing.assert_allclose(out["loss"].item(), g["losses"][st], rtol=1e-3 if st == 0 else 5e-3)
        np.testing.assert_allclose(out["grad_norm"].item(), g["grad_norms"][st], rtol=5e-3 if st == 0 else 2e-2)
        if rate < 1:
            assert np.array_equal(out["index"].numpy(), g["index_step%d" % st])
        if st == 0:
            for k in [k[6:] for k in g if k.startswith("grad0.")]:
                want = g["grad0." + k]
                np.testing.assert_allclose(recipe.probe(out["grads"][k]), want, rtol=5e-3, atol=5e-3 * want[1] / out["grads"][k].numel() ** 0.5, err_msg=k)
    for k in [k[6:] for k in g if k.startswith("after.") and not k.startswith("after.head")]:
        if "running" in k:
            np.testing.assert_allclose(recipe.probe(sd[k].float()), g["after." + k], rtol=2e-2, atol=1e-5, err_msg=k)
        else:
            _adam_close(recipe.probe(sd[k].float()), g["after." + k], lr, steps, k)
    _adam_close(recipe.probe(W, 4096), g["after.head_weight"], lr, steps, "head weight")
    np.testing.assert_allclose(recipe.probe(opt.m["head"], 4096)[1], g["after.head_exp_avg"][1], rtol=2e-2)
    np.testing.assert_allclose(recipe.probe(opt.v["head"], 4096)[1], g["after.head_exp_avg_sq"][1], rtol=4e-2)


@pytest.mark.parametrize("tag", ["rate10", "rate03"])
def test_train_steps_fresh_batches(golden, tag):
    """three SGD steps on a NEW synthetic batch each (the repeated batch of test_train_steps is memorised after one step): every step's loss
    and gradient norm at the first step's tolerance, sampled rows bit-exact, element probes of six backbone tensors and the head after the steps"""
    g = golden("train_step_resnet18_c256_fresh_" + tag)
    C, B, steps, rate = int(g["C"]), int(g["B"]), int(g["steps"]), float(g["rate"])
    blocks = resnet_ref.BLOCKS["ResNet18"]
    spec = resnet_ref.resnet_spec(blocks)
    sd = recipe.fill_state(spec, 777)
    for k, _, kind in spec:
        if kind in ("bn_w", "bn_rv"):
            sd[k].fill_(1.0)
        elif kind in ("bn_b", "bn_rm"):
            sd[k].zero_()
    W = recipe.normal(778, (C, 512), 0.01)
    opt = train_ref.SGDState(float(g["lr"]), float(g["momentum"]), float(g["wd"]))
    for st in range(steps):
        img, ids = recipe.images(779 + 10 * st, B), recipe.labels(780 + 10 * st, B, C)
        u = [T(g["u"][st])] if rate < 1 else None
        out = train_ref.train_step(sd, W, img, ids, blocks, C, opt, sample_rate=rate, uniforms=u)
        np.testing.assert_allclose(out["loss"].item(), g["losses"][st], rtol=2e-3)
        np.testing.assert_allclose(out["grad_norm"].item(), g["grad_norms"][st], rtol=5e-3)
        if rate < 1:
            assert np.array_equal(out["index"].numpy(), g["index_step%d" % st])
    for k in [k[6:] for k in g if k.startswith("probe.") and k != "probe.head_weight"]:
        want, got = g["probe." + k], recipe.probe(sd[k].float())
        # three clipped lr-0.1 steps move a parameter by up to its own size; ReLU / max-pool kinks make the steps' gradients differ by a few
        # per cent of their rms between any two fp32 evaluations (tests/wholenet.py), hence 5 % of the PARAMETER's rms per element
        np.testing.assert_allclose(got[1], want[1], rtol=2e-3, err_msg=k)
        np.testing.assert_allclose(got[2:], want[2:], rtol=5e-3, atol=5e-2 * want[1] / sd[k].numel() ** 0.5 + 1e-7, err_msg=k)
    want, got = g["probe.head_weight"], recipe.probe(W, 4096)
    np.testing.assert_allclose(got[1:], want[1:], rtol=5e-3, atol=2e-2 * want[1] / W.numel() ** 0.5)
