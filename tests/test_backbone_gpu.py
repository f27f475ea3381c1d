"""The HIP backbone (nets.resnet drop-in) against the reference-generated fixtures and the oracle."""
import types

import numpy as np
import pytest
import torch

from oracle import recipe, resnet_ref

pytestmark = pytest.mark.gpu


def _net(name, dtype, seed):
    import nets.resnet as R
    conf = types.SimpleNamespace(network=name, emd_size=512, frhip_dtype=dtype)
    net = R.Encoder(conf)
    sd = recipe.fill_state(resnet_ref.resnet_spec(resnet_ref.BLOCKS[name]), seed)
    net.load_state_dict(sd, strict=True)          # reference key names / shapes
    return net.cuda(), sd


def test_resnet18_train_fp32_matches_reference_fixture(golden):
    """fp32 validation mode: embeddings, every parameter gradient and the BN running statistics after one
    training-mode forward/backward, against vectors produced by the real reference (resnet18_b4_train)."""
    g = golden("resnet18_b4_train")
    net, _ = _net("ResNet18", "fp32", 4242)
    net.train()
    x = recipe.images(4243, 4).cuda()
    y = net(x)
    y.backward(recipe.normal(4244, (4, 512), 0.05).cuda())
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["out"], rtol=1e-3, atol=2e-4)
    for k, p in net.named_parameters():
        got, want = recipe.summary(p.grad.cpu()), g["gsum." + k]
        np.testing.assert_allclose(got, want, rtol=5e-3, atol=5e-5 + 2e-3 * abs(want[1]), err_msg=k)
        # element by element at 256 portable positions of the LOGICAL [K,C,R,S] / [out,in] order: no layout permutation passes
        want = g["gprobe." + k]
        rms = want[1] / p.numel() ** 0.5
        np.testing.assert_allclose(recipe.probe(p.grad.cpu())[2:], want[2:], rtol=5e-3, atol=1e-2 * rms + 2e-6, err_msg=k + " (probe)")
    for k in ("conv1.weight", "layer1.0.conv1.weight"):                        # whole tensors
        want = g["gfull." + k]
        got = dict(net.named_parameters())[k].grad.cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=5e-3, atol=1e-2 * float(np.sqrt((want.astype(np.float64) ** 2).mean())), err_msg=k)
    want = g["gprobe16k.fc.weight"]                                            # 16 384 elements of the (c,h,w) -> (h,w,c) permuted fc columns
    np.testing.assert_allclose(recipe.probe(net.fc.weight.grad.cpu(), 16384)[2:], want[2:], rtol=5e-3, atol=1e-2 * want[1] / net.fc.weight.numel() ** 0.5)
    for k, b in net.named_buffers():
        np.testing.assert_allclose(recipe.summary(b.float().cpu()), g["after." + k], rtol=1e-3, atol=1e-5, err_msg=k)


def test_resnet18_eval_fp32(golden):
    net, _ = _net("ResNet18", "fp32", 4242)
    net.eval()
    with torch.no_grad():
        y = net(recipe.images(4243, 4).cuda())
    np.testing.assert_allclose(y.cpu().numpy(), golden("resnet18_b4_eval")["out"], rtol=1e-3, atol=2e-4)


def test_resnet50_eval_fp32(golden):
    net, _ = _net("ResNet50", "fp32", 5050)
    net.eval()
    with torch.no_grad():
        y = net(recipe.images(5051, 2).cuda())
    np.testing.assert_allclose(y.cpu().numpy(), golden("resnet50_b2_eval")["out"], rtol=1e-3, atol=2e-4)


@pytest.mark.parametrize("name,batch", [("ResNet18", 16), ("ResNet50", 64)])
def test_eval_with_folded_batchnorm_matches_the_separate_passes(monkeypatch, name, batch):
    """Inference (reference model/FR_PartialFC.py:205-211, encoder.eval()): the eval-mode BatchNorms ride in the store epilogues of the
    convolutions (frhip_conv_fwd_affine; nets._backbone._EVAL_FOLD) -- lean kernels where the batch makes whole tiles, the general
    epilogue elsewhere.  Same arithmetic as conv + BatchNorm-apply on the rounded conv output; only the shortcut convolution's
    BatchNorm is rounded once more (its output is a tensor now: four bf16 roundings of 2^-9 in a ResNet, carried through 25 blocks).
    bf16: embeddings within 2 % of the separate passes (the suite's whole-network bf16 tolerance is 5 %), cosine >= 0.9999."""
    import nets._backbone as bb
    net, _ = _net(name, "bf16", 777)
    net.eval()
    x = recipe.images(778, batch).cuda()
    outs = []
    for fold in (False, True):
        monkeypatch.setattr(bb, "_EVAL_FOLD", fold)
        with torch.no_grad():
            outs.append(net(x).float())
    a, b = outs
    assert torch.isfinite(b).all()
    rel = float((a - b).norm() / a.norm())
    cos = torch.nn.functional.cosine_similarity(a, b, dim=1)
    assert rel <= 2e-2 and float(cos.min()) >= 0.9999, (rel, float(cos.min()))


def test_resnet18_bf16_tracks_fp32():
    """bf16 MFMA path: same code, bf16 storage.  Embeddings stay within bf16 noise of the fp32 path and the
    parameter gradients point the same way (cosine > 0.99 per tensor for the large tensors)."""
    net32, _ = _net("ResNet18", "fp32", 4242)
    net16, _ = _net("ResNet18", "bf16", 4242)
    x = recipe.images(4243, 8).cuda()
    gy = recipe.normal(4244, (8, 512), 0.05).cuda()
    outs = []
    for net in (net32, net16):
        net.train()
        y = net(x)
        y.backward(gy)
        outs.append(y.detach().float().cpu())
    rel = (outs[0] - outs[1]).norm() / outs[0].norm()
    assert rel < 0.05, rel
    for (k, p32), (_, p16) in zip(net32.named_parameters(), net16.named_parameters()):
        if p32.numel() < 1000:
            continue
        a, b = p32.grad.flatten().double(), p16.grad.flatten().double()
        cos = (a @ b) / (a.norm() * b.norm() + 1e-30)
        assert cos > 0.95, (k, cos.item())


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_stem_reduction_fused_into_block0_matches_recompute_pass(dtype, monkeypatch):
    """the stem's BN-backward sums from the (dpool, pooled) reduction fused into block 0's data-gradient epilogue (default)
    against the stem's own recompute reduction pass: same gradients for the stem's parameters and everything else.
    bn1.weight starts near the reference's initial 1.0 here, and one channel is made (near-)dead to exercise the
    regularised 1 / gamma"""
    import nets._backbone as BB
    grads = {}
    for fused in (True, False):
        monkeypatch.setattr(BB, "_STEM_FUSED_REDUCE", fused)
        net, _ = _net("ResNet18", dtype, 4747)
        with torch.no_grad():
            net.bn1.weight[3] = 1e-6
            net.bn1.weight[5] = 0.0
        net.train()
        y = net(recipe.images(4748, 6).cuda())
        y.backward(recipe.normal(4749, (6, 512), 0.05).cuda())
        grads[fused] = {k: p.grad.detach().float().cpu() for k, p in net.named_parameters()}
    live = torch.ones(64, dtype=torch.bool)
    live[3] = live[5] = False
    for k in grads[True]:
        a, b = grads[True][k], grads[False][k]
        assert torch.isfinite(a).all(), k
        if k in ("bn1.weight", "conv1.weight"):    # d(gamma) of the dead channels is regularised towards 0 and their dy carries
            a, b = a[live], b[live]                # gamma * (...) ~ 0 either way: not compared
        if dtype == "fp32":
            scale = max(b.abs().max().item(), 1e-6)
            # + 1e-6: analytically-zero gradients (a bias in front of a train-mode BN) are pure round-off in both runs
            np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=2e-4, atol=2e-4 * scale + 1e-6, err_msg=k)
        elif k in ("bn1.weight", "bn1.bias", "conv1.weight"):
            # bf16: the fp32 atomics of the fc GEMM make two runs differ in the last bit of the embeddings, which now and then
            # flips a bf16 rounding upstream and is amplified by the 6-sample BatchNorms -> compare the stem's tensors in norm
            rel = ((a - b).norm() / b.norm()).item()
            assert rel < 0.25, (k, rel)


def test_checkpoint_roundtrip_with_reference_keys():
    net, sd = _net("ResNet18", "fp32", 4242)
    out = net.state_dict()
    assert list(out.keys()) == list(sd.keys())
    for k in sd:
        assert torch.equal(out[k].cpu(), sd[k]), k


def test_cpu_input_is_refused():
    net, _ = _net("ResNet18", "fp32", 1)
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 3, 112, 112))


def test_data_parallel_gradient_allreduce_covers_the_arena_once(monkeypatch, tmp_path):
    """Data parallel: the backbone hands tail slices of its flat gradient arena to the all-reduce while the backward pass
    runs (nets._backbone.BackwardCtx).  With a recording stand-in for dist.all_reduce: every arena element is reduced
    exactly once, slices come in descending order, and p.grad aliases the arena (so the in-place average IS the gradient)."""
    import torch.distributed as dist
    import nets._backbone as BB
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method="file://" + str(tmp_path / "pg"), rank=0, world_size=1)
    calls = []

    class _Work:
        def wait(self):
            return True

    def fake_all_reduce(t, op=None, async_op=False):
        assert op == (dist.ReduceOp.AVG if dist.get_backend() == "nccl" else dist.ReduceOp.SUM) and async_op
        calls.append((t.data_ptr(), t.numel()))
        t.mul_(0.5)                                    # pretend the other rank contributed zeros
        return _Work()

    net, _ = _net("ResNet18", "fp32", 4242)
    net.train()
    x = recipe.images(4243, 4).cuda()
    dy = recipe.normal(4244, (4, 512), 0.05).cuda()
    net(x).backward(dy)
    ref = {k: p.grad.clone() for k, p in net.named_parameters()}
    net.zero_grad(set_to_none=True)
    for b in net.buffers():                            # same BN statistics path: reset what the first pass changed is not needed for grads
        pass
    monkeypatch.setattr(BB.dist, "all_reduce", fake_all_reduce)
    monkeypatch.setattr(BB.BackwardCtx, "MIN_BYTES", 1 << 20)
    net._frhip_allreduce = True
    net(x).backward(dy)
    total = sum(p.numel() for p in net.parameters())
    assert sum(n for _, n in calls) == total and len(calls) >= 3
    ends = [ptr + 4 * n for ptr, n in calls]
    for (ptr, _), prev_start in zip(calls[1:], [c[0] for c in calls[:-1]]):
        assert ptr + 4 * [n for q, n in calls if q == ptr][0] == prev_start        # contiguous, descending
    lo = min(ptr for ptr, _ in calls)
    for k, p in net.named_parameters():
        assert lo <= p.grad.data_ptr() < lo + 4 * total, k
        np.testing.assert_allclose(p.grad.cpu().numpy(), 0.5 * ref[k].cpu().numpy(), rtol=1e-3, atol=1e-5 * max(1.0, ref[k].abs().max().item()), err_msg=k)


def test_data_parallel_wrapper_has_ddp_surface(tmp_path):
    """nets._backbone.DataParallel stands where the reference has DistributedDataParallel: `.module`, 'module.'-prefixed
    state_dict keys (the reference's checkpoints, model/FR_PartialFC.py:76-90), parameters broadcast from rank 0"""
    import torch.distributed as dist
    import nets._backbone as BB
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method="file://" + str(tmp_path / "pg"), rank=0, world_size=1)
    net, sd = _net("ResNet18", "fp32", 4242)
    dp = BB.DataParallel(net)
    assert dp.module is net and net._frhip_allreduce
    assert set(dp.state_dict().keys()) == {"module." + k for k in net.state_dict().keys()}
    for k, v in net.state_dict().items():
        assert torch.equal(v.cpu(), sd[k].to(v.dtype)), k          # the broadcast round trip left every value intact
    dp.eval()
    with torch.no_grad():
        assert dp(recipe.images(1, 2).cuda()).shape == (2, 512)
    with pytest.raises(TypeError):
        BB.DataParallel(torch.nn.Linear(2, 2))


def test_folded_bn1_relu_whole_net_is_bit_identical(monkeypatch):
    """nets._backbone._FUSE_BN1 (bn1-apply + ReLU inside conv2's forward kernel and inside its weight-gradient kernel, a1 never
    written) against the default separate pass: a ResNet18 bf16 training step must give bit-identical embeddings, parameter
    gradients and BatchNorm buffers."""
    import nets._backbone as bb
    import nets.resnet as R
    conf = types.SimpleNamespace(network="ResNet18", emd_size=512, frhip_dtype="bf16")
    x = recipe.images(4401, 8).cuda()
    gy = recipe.normal(4402, (8, 512)).cuda()
    outs = []
    for fuse in (False, True):
        monkeypatch.setattr(bb, "_FUSE_BN1", fuse)
        torch.manual_seed(4403)
        net = R.ResNet18(conf).cuda().train()
        y = net(x)
        y.backward(gy)
        outs.append((y.detach().clone(), [p.grad.clone() for p in net.parameters()], [b.clone() for b in net.buffers()]))
    (y0, g0, b0), (y1, g1, b1) = outs
    assert torch.equal(y0, y1)                       # forward: bit-identical
    for a, b in zip(b0, b1):
        assert torch.equal(a, b)                     # BatchNorm buffers: bit-identical
    for a, b in zip(g0, g1):
        # gradients: identical arithmetic; tensors below 16 K elements take the fp32-atomic split-K path of the weight-gradient
        # kernels, whose summation ORDER varies from launch to launch (last-bit differences with or without the fusion)
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6 * float(b.abs().max()))
