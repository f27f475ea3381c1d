"""BASELINE cfg 2 sizes (B = 512) on the MI355X: size-independent properties of the conv kernels where a CPU reference would
take minutes.  A convolution is linear in x and in w, so with y = conv(x, w):
    <y, dy> = <x, dgrad(dy, w)> = <w, wgrad(dy, x)>          (the data- and weight-gradient kernels are its adjoints)
and alternative kernels for the same problem (LDS-halo vs generic implicit GEMM, nine-tap vs per-tap weight gradient) agree."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
B = 512


def _rand(shape, seed, std=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn(shape, generator=g, device="cuda") * std


def _dot(a, b):
    return (a.double().flatten() @ b.double().flatten()).item()


@pytest.mark.parametrize("geom", [(14, 256, 256, 3, 1, 1), (56, 64, 128, 3, 2, 1), (56, 64, 64, 3, 1, 1), (28, 128, 256, 1, 2, 0),
                                  (7, 512, 512, 3, 1, 1)])
def test_conv_gradients_are_adjoints_at_batch_512(geom):
    from frhip import ops
    h, c, k, r, stride, pad = geom
    ho = (h + 2 * pad - r) // stride + 1
    x = _rand((B, h, h, c), 1).bfloat16()
    w32 = _rand((k, r, r, c), 2, 0.05)
    w = w32.bfloat16()
    y, _ = ops.conv_fwd(x, w, stride, pad, want_stats=False)
    # dy correlated with y: <y, dy> is then a large number (~ |y|^2 / 2) and the identity is checked to 1e-3 of it; with an
    # independent dy all three inner products would be round-off sized
    dy = (0.5 * y.float() + 0.5 * y.float().std() * _rand((B, ho, ho, k), 3)).bfloat16()
    dx = ops.conv_dgrad(dy, ops.pack_wt(w.float(), torch.bfloat16), (B, h, h, c), r, r, stride, pad)
    dw = torch.zeros((k, r, r, c), dtype=torch.float32, device="cuda")
    ops.conv_wgrad(dy, x, dw, r, r, stride, pad)
    ref = _dot(y, dy)
    assert ref > 0.2 * (y.double().norm() ** 2).item()
    # bf16 rounding of y / dx (2^-9 relative per element, independent) averages out over 1e7..1e8 terms; dw is fp32
    assert abs(_dot(x, dx) - ref) < 2e-3 * ref, (ref, _dot(x, dx))
    assert abs(_dot(w, dw) - ref) < 2e-3 * ref, (ref, _dot(w, dw))
    assert torch.isfinite(y.float()).all() and torch.isfinite(dx.float()).all() and torch.isfinite(dw).all()


def test_alternative_kernels_agree_at_batch_512():
    from frhip import ops
    from frhip._abi import lib
    x = _rand((B, 14, 14, 256), 11).bfloat16()
    w = _rand((256, 3, 3, 256), 12, 0.05).bfloat16()
    dy = _rand((B, 14, 14, 256), 13).bfloat16()
    old = lib().frhip_set_conv_halo(0)
    try:
        y_nt, p_nt = ops.conv_fwd(x, w, 1, 1, want_stats=True)
    finally:
        lib().frhip_set_conv_halo(old)
    y_h, p_h = ops.conv_fwd(x, w, 1, 1, want_stats=True)
    # same products, fp32 accumulation in a different order, one bf16 rounding at the end
    np.testing.assert_allclose(y_h.float().cpu().numpy(), y_nt.float().cpu().numpy(), rtol=2e-2, atol=2e-2)
    np.testing.assert_allclose(p_h.sum(0).cpu().numpy(), p_nt.sum(0).cpu().numpy(), rtol=1e-3, atol=1.0)
    dws = []
    for mode in (0, 1):                     # per-tap gather kernel, nine-tap kernel
        o = lib().frhip_set_wgrad_taps9(mode)
        try:
            dw = torch.zeros((256, 3, 3, 256), dtype=torch.float32, device="cuda")
            ops.conv_wgrad(dy, x, dw, 3, 3, 1, 1)
            dws.append(dw)
        finally:
            lib().frhip_set_wgrad_taps9(o)
    np.testing.assert_allclose(dws[0].cpu().numpy(), dws[1].cpu().numpy(), rtol=1e-3, atol=1e-3 * dws[0].abs().max().item())
