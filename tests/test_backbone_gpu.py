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
