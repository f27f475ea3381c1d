"""Shared by the oracle (CPU) and the HIP (GPU) whole-network training-mode parity tests (fixtures swin34_b8_train, alternet50_b8_train)."""
import numpy as np
import torch

from oracle import recipe

NOISE_TAILS = ("proj.bias", "fc2.bias", "v_bias")


def check_whole_net_train(g, grads, out, running, rtol=5e-3, noise=()):
    """shared by the oracle (CPU) and the HIP (GPU) whole-network training-mode tests: embeddings, a probe of EVERY parameter gradient
    (sum, l2, 256 elements at portable positions -- a layout permutation cannot pass), full tensors where the fixture holds them, running
    statistics.  Element tolerance = rtol of the value + rtol x the tensor's rms (fp32 summation order through 30-50 BatchNorm'd blocks)."""
    np.testing.assert_allclose(out, g["out"], rtol=rtol, atol=rtol * float(np.abs(g["out"]).max()))
    for key in [k for k in g if k.startswith("gprobe.")]:
        k = key[7:]
        want = g[key]
        got = recipe.probe(grads[k])
        rms = want[1] / max(1.0, float(grads[k].numel())) ** 0.5
        if k.endswith(NOISE_TAILS) or k in noise or rms < 1e-7:           # analytically-zero gradients (a shift in front of a training-mode BatchNorm): size only
            assert np.abs(got[2:]).max() <= 1e-3 + 50 * np.abs(want[2:]).max(), k
            continue
        np.testing.assert_allclose(got[1], want[1], rtol=rtol, atol=1e-7, err_msg=k + " (l2)")
        np.testing.assert_allclose(got[2:], want[2:], rtol=rtol, atol=2 * rtol * rms + 1e-8, err_msg=k)
    for key in [k for k in g if k.startswith("gfull.")]:
        k = key[6:]
        want = g[key]
        rms = float(np.sqrt((want.astype(np.float64) ** 2).mean()))
        np.testing.assert_allclose(grads[k].numpy().reshape(want.shape), want, rtol=rtol, atol=2 * rtol * rms + 1e-8, err_msg=k + " (full)")
    want = g["gprobe16k.fc.weight"]
    np.testing.assert_allclose(recipe.probe(grads["fc.weight"], 16384)[2:], want[2:], rtol=rtol, atol=2 * rtol * want[1] / grads["fc.weight"].numel() ** 0.5)
    for key in [k for k in g if k.startswith("after.")]:
        np.testing.assert_allclose(recipe.probe(running[key[6:]].float()), g[key], rtol=1e-3, atol=1e-5, err_msg=key)


def whole_net_train_on_gpu(net, g, h=112, w=112):
    """one training-mode forward/backward of a product backbone on the fixture's inputs -> (grads by name, embeddings, buffers), CPU"""
    net.train()
    net.dropout.p = 0.0                  # the fixtures' RNG-free training pass (Dropout p = 0, stochastic depth off)
    for m in net.modules():
        if hasattr(m, "drop_path_rate"):
            m.drop_path_rate = 0.0
    seed, batch = int(g["seed"]), int(g["batch"])
    y = net(recipe.images(seed + 1, batch, h, w).cuda())
    y.backward(recipe.normal(seed + 2, tuple(y.shape), 0.05).cuda())
    torch.cuda.synchronize()
    return ({k: p.grad.float().cpu() for k, p in net.named_parameters()}, y.detach().float().cpu().numpy(),
            {k: b.detach().cpu() for k, b in net.named_buffers()})
