"""GPU parity of each HIP kernel (called through the C ABI) against a plain PyTorch fp32 CPU reference of the
same op.  fp32 mode: exact-fp32 MFMA, tight tolerances.  bf16 mode: inputs are rounded to bf16 first and the
reference is computed in fp32 from the rounded inputs, so the tolerance only covers bf16 output rounding
(2^-8 relative) and accumulation order."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


def _ops():
    from frhip import ops
    return ops


def tol(dtype, scale=1.0):
    return (dict(rtol=2e-4, atol=2e-5 * scale) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2 * scale))


def rnd(seed, shape, std=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * std


def q(t, dtype):
    """round to the compute dtype and come back to fp32 (CPU)"""
    return t.to(dtype).float()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


CONV_CASES = [
    # n, h, w, c, k, r, stride, pad
    (2, 8, 8, 64, 64, 3, 1, 1),
    (3, 7, 7, 64, 128, 3, 1, 1),       # M = 147: ragged last row tile
    (2, 10, 6, 128, 128, 3, 2, 1),     # strided, non-square
    (2, 8, 8, 64, 128, 1, 2, 0),       # downsample 1x1 s2
    (1, 5, 5, 256, 512, 3, 1, 1),
    (5, 9, 9, 64, 64, 3, 2, 1),        # odd size with stride 2
    (2, 56, 56, 64, 64, 3, 1, 1),      # halo kernel, widest map, one 64-channel chunk
    (3, 28, 28, 128, 128, 3, 1, 1),    # halo kernel, two chunks, tiles cross image boundaries
    (5, 14, 14, 256, 256, 3, 1, 1),
    (9, 7, 7, 512, 512, 3, 1, 1),      # 441 pixels: ragged last tile, 5+ images per tile
    (3, 7, 7, 128, 64, 3, 1, 1),       # narrow output, multi-chunk
    (2, 12, 12, 64, 128, 3, 1, 1),     # wide output, single chunk
    (3, 14, 14, 64, 128, 2, 2, 0),     # 2x2 / stride-2 stage convolution of the Swin backbone
    (2, 28, 28, 128, 256, 2, 2, 0),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_and_stats(dtype, case):
    ops = _ops()
    n, h, w, c, k, r, stride, pad = case
    x = q(rnd(1, (n, c, h, w)), dtype)
    wt = q(rnd(2, (k, c, r, r), 0.05), dtype)
    ref = F.conv2d(x, wt, None, stride, pad)
    y, part = ops.conv_fwd(nhwc(x).to(dtype).cuda(), nhwc(wt).to(dtype).cuda(), stride, pad)
    got = nchw(y.float().cpu())
    np.testing.assert_allclose(got.numpy(), ref.numpy(), **tol(dtype, ref.abs().max().item()))
    # epilogue statistics are those of the STORED tensor
    s = part.sum(dim=0).cpu()
    yy = y.float().cpu().reshape(-1, k)
    np.testing.assert_allclose(s[0].numpy(), yy.sum(0).numpy(), rtol=1e-3, atol=1e-2)
    np.testing.assert_allclose(s[1].numpy(), (yy * yy).sum(0).numpy(), rtol=1e-3, atol=1e-2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_dgrad(dtype, case):
    ops = _ops()
    n, h, w, c, k, r, stride, pad = case
    ho, wo = ops.conv_out_hw(h, w, r, r, stride, pad)
    wt = q(rnd(2, (k, c, r, r), 0.05), dtype)
    dy = q(rnd(3, (n, k, ho, wo)), dtype)
    res = q(rnd(4, (n, c, h, w)), dtype)
    ref = torch.nn.grad.conv2d_input((n, c, h, w), wt, dy, stride, pad) + res
    wpack = ops.pack_wt(nhwc(wt).cuda(), dtype)
    dx = ops.conv_dgrad(nhwc(dy).to(dtype).cuda(), wpack, (n, h, w, c), r, r, stride, pad,
                        residual=nhwc(res).to(dtype).cuda())
    np.testing.assert_allclose(nchw(dx.float().cpu()).numpy(), ref.numpy(), **tol(dtype, ref.abs().max().item()))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("splits", [0, 1, 3])
def test_conv_wgrad(dtype, case, splits):
    ops = _ops()
    n, h, w, c, k, r, stride, pad = case
    ho, wo = ops.conv_out_hw(h, w, r, r, stride, pad)
    x = q(rnd(1, (n, c, h, w)), dtype)
    dy = q(rnd(3, (n, k, ho, wo)), dtype)
    ref = torch.nn.grad.conv2d_weight(x, (k, c, r, r), dy, stride, pad)
    dw = torch.zeros((k, r, r, c), dtype=torch.float32, device="cuda")
    ops.conv_wgrad(nhwc(dy).to(dtype).cuda(), nhwc(x).to(dtype).cuda(), dw, r, r, stride, pad, splits)
    got = dw.cpu().permute(0, 3, 1, 2)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=2e-3 if dtype == torch.bfloat16 else 2e-4,
                               atol=1e-3 * ref.abs().max().item())


def test_halo_kernel_equals_generic_kernel():
    """The LDS-halo 3x3/s1 kernel and the generic tap-by-tap kernel are two schedules of the same sums."""
    ops = _ops()
    from frhip._abi import lib
    for (n, h, c, k) in [(4, 56, 64, 64), (6, 28, 128, 128), (7, 14, 256, 256), (11, 7, 512, 512)]:
        x = rnd(40, (n, h, h, c)).bfloat16().cuda()
        w = (rnd(41, (k, 3, 3, c)) * 0.05).bfloat16().cuda()
        res = rnd(42, (n, h, h, c)).bfloat16().cuda()
        wt = ops.pack_wt(w.float(), torch.bfloat16)
        dy = rnd(43, (n, h, h, k)).bfloat16().cuda()
        outs = []
        for halo in (2, 3, 0):       # 4-wave tile (two workgroups per CU) / 8-wave tile with double-buffered halo / generic
            old = lib().frhip_set_conv_halo(halo)
            y, part = ops.conv_fwd(x, w, 1, 1)
            dx = ops.conv_dgrad(dy, wt, (n, h, h, c), 3, 3, 1, 1, residual=res)
            lib().frhip_set_conv_halo(old)
            outs.append((y.float().cpu(), part.sum(0).cpu(), dx.float().cpu()))
        ref = outs[2]
        scale = ref[0].abs().max().item()
        for o in outs[:2]:
            np.testing.assert_allclose(o[0].numpy(), ref[0].numpy(), rtol=0, atol=scale * 2 ** -7)   # one bf16 ulp of the largest value
            np.testing.assert_allclose(o[1].numpy(), ref[1].numpy(), rtol=2e-3, atol=0.5)
            np.testing.assert_allclose(o[2].numpy(), ref[2].numpy(), rtol=0, atol=ref[2].abs().max().item() * 2 ** -7)


@pytest.mark.parametrize("case", [(6, 28, 128, 128), (7, 14, 256, 256), (11, 7, 512, 512), (9, 14, 128, 256), (300, 7, 256, 128),
                                  (3, 20, 128, 384)])
def test_halo_wide_tile_is_bit_identical_to_the_four_wave_tile(case):
    """64 x 128 outputs per wave (igemm_halo_wide.h) against 64 x 64 (igemm_halo.h): the same MFMAs in the same order per output,
    the same rows per BN-partial -> identical outputs AND identical partial sums, forward and data-gradient"""
    ops = _ops()
    from frhip._abi import lib
    n, h, c, k = case
    x = rnd(80, (n, h, h, c)).bfloat16().cuda()
    w = (rnd(81, (k, 3, 3, c)) * 0.05).bfloat16().cuda()
    dy = rnd(82, (n, h, h, k)).bfloat16().cuda()
    wt = ops.pack_wt(w.float(), torch.bfloat16)
    res = rnd(83, (n, h, h, c)).bfloat16().cuda()
    y_bn = rnd(84, (n, h, h, c)).bfloat16().cuda()
    rows = n * h * h
    st = ops.bn_finalize(ops.colstats(y_bn.view(rows, c)), rows, torch.ones(c).cuda(), torch.zeros(c).cuda(), None, None)
    outs = []
    for dirs in (3, 0):              # the wide tile in both directions / nowhere (the 4-wave tile runs instead)
        old = lib().frhip_set_halo_wide_dirs(dirs)
        y, part = ops.conv_fwd(x, w, 1, 1)
        y2, _ = ops.conv_fwd(x, w, 1, 1, want_stats=False)
        dx, bpart = ops.conv_dgrad(dy, wt, (n, h, h, c), 3, 3, 1, 1, residual=res, bnred=(y_bn, st, True))
        dx2 = ops.conv_dgrad(dy, wt, (n, h, h, c), 3, 3, 1, 1)
        lib().frhip_set_halo_wide_dirs(old)
        outs.append((y, part, y2, dx, bpart, dx2))
    for a, b in zip(*outs):
        assert a.shape == b.shape and torch.equal(a, b)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(3, 14, 64, 128, 1, True), (2, 28, 128, 128, 1, False), (2, 16, 64, 64, 2, True),
                                  (5, 7, 256, 256, 1, True), (1, 9, 64, 64, 1, False),
                                  (3, 9, 128, 64, 2, False), (2, 7, 64, 128, 2, True)])      # stride 2, odd sizes: parity classes
def test_dgrad_epilogue_bn_backward_reduction(dtype, case):
    """conv_dgrad(bnred=...) = conv_dgrad followed by the stand-alone BN-backward reduction over (dx, y_bn)"""
    ops = _ops()
    from frhip._abi import lib
    n, h, c, k, stride, mask = case
    ho = (h + 2 - 3) // stride + 1
    dy = rnd(60, (n, ho, ho, k)).to(dtype).cuda()
    w = (rnd(61, (k, 3, 3, c)) * 0.05)
    wt = ops.pack_wt(w.cuda(), dtype)
    res = rnd(62, (n, h, h, c)).to(dtype).cuda()
    y_bn = rnd(63, (n, h, h, c)).to(dtype).cuda()
    rows = n * h * h
    st = ops.bn_finalize(ops.colstats(y_bn.view(rows, c)), rows, (1 + 0.1 * rnd(64, (c,))).cuda(), (0.1 * rnd(65, (c,))).cuda(),
                         None, None)
    dx_ref = ops.conv_dgrad(dy, wt, (n, h, h, c), 3, 3, stride, 1, residual=res)
    nb = lib().frhip_colreduce_blocks(rows, c, ops.dt_of(y_bn))
    pref = torch.empty((nb, 2, c), dtype=torch.float32, device="cuda")
    P = ops._p
    ops.check(lib().frhip_bn_bwd_reduce(ops.dt_of(y_bn), P(dx_ref), P(y_bn), P(st.mean), P(st.invstd),
                                        P(st.scale) if mask else None, P(st.shift) if mask else None, rows, c, P(pref), ops._s()),
              "frhip_bn_bwd_reduce")
    dx, part = ops.conv_dgrad(dy, wt, (n, h, h, c), 3, 3, stride, 1, residual=res, bnred=(y_bn, st, mask))
    assert torch.equal(dx, dx_ref)
    a, b = part.sum(0).cpu().numpy(), pref.sum(0).cpu().numpy()
    np.testing.assert_allclose(a, b, rtol=2e-4, atol=2e-4 * max(1.0, np.abs(b).max()))


@pytest.mark.parametrize("case", [(4, 56, 64), (16, 28, 128), (64, 14, 256), (256, 7, 512), (4, 8, 64), (16, 8, 128)])
def test_lean_epilogue_against_the_general_one_and_float64(case):
    """Launches made of whole tiles run the LEAN kernels (frhip_set_epi_lean): 32-bit buffer-offset stores and the BatchNorm sums
    on the matrix pipe (ones x D, diagonal of D^T Y; the backward sum as invstd * (sum d y - mean * sum d)).  Outputs must be the
    bits of the general epilogue; the per-channel sums are held against float64 sums over the STORED tensor and may not be further
    from them than 4x the general epilogue's own error (+ 1e-6 of the sum of magnitudes)."""
    ops = _ops()
    from frhip._abi import lib
    n, h, c = case
    dt = torch.bfloat16
    x = rnd(70, (n, h, h, c)).to(dt).cuda()
    w = (rnd(71, (c, 3, 3, c)) * 0.05)
    wp, wt = w.to(dt).cuda(), ops.pack_wt(w.cuda(), dt)
    res = rnd(72, (n, h, h, c)).to(dt).cuda()
    y_bn = (rnd(73, (n, h, h, c)) * 0.7 + 1.5).to(dt).cuda()          # a mean of two standard deviations: exercises the cancellation
    rows = n * h * h
    assert rows % 256 == 0
    st = ops.bn_finalize(ops.colstats(y_bn.view(rows, c)), rows, (1 + 0.1 * rnd(74, (c,))).cuda(), (0.1 * rnd(75, (c,))).cuda(), None, None)
    out = {}
    for lean in (0, 1):
        old = lib().frhip_set_epi_lean(lean)
        try:
            y, p = ops.conv_fwd(x, wp, 1, 1)
            out[lean] = [(y, p)]
            for mask in (False, True):
                for r in (None, res):
                    out[lean].append(ops.conv_dgrad(x, wt, (n, h, h, c), 3, 3, 1, 1, residual=r, bnred=(y_bn, st, mask)))
        finally:
            lib().frhip_set_epi_lean(old)
    yb = y_bn.double().view(rows, c)
    mean, invstd, sc, sh = st.mean.double(), st.invstd.double(), st.scale.double(), st.shift.double()
    for i, ((t0, p0), (t1, p1)) in enumerate(zip(out[0], out[1])):
        assert torch.equal(t0, t1), i
        assert p0.shape == p1.shape
        d = t1.double().view(rows, c)
        if i == 0:
            want = torch.stack([d.sum(0), (d * d).sum(0)])
            mag = torch.stack([d.abs().sum(0), (d * d).sum(0)])
        else:
            if (i - 1) // 2 == 1:
                d = d * ((y_bn.float().view(rows, c) * st.scale + st.shift) > 0)       # the mask as the kernels form it (fp32)
            xh = (yb - mean) * invstd
            want = torch.stack([d.sum(0), (d * xh).sum(0)])
            mag = torch.stack([d.abs().sum(0), (d * xh).abs().sum(0)])
        e0 = (p0.double().sum(0) - want).abs()
        e1 = (p1.double().sum(0) - want).abs()
        lim = 4 * e0.max() + 1e-6 * mag.max()
        assert float(e1.max()) <= float(lim), (i, float(e0.max()), float(e1.max()), float(mag.max()))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(3, 14, 64, 64), (2, 28, 128, 128), (2, 9, 64, 64)])
def test_dgrad_compact_stride2_residual(dtype, case):
    """residual_stride=2 (compact gradient of a stride-2 1x1 shortcut, added on the even pixels) == the zero-stuffed dense residual"""
    ops = _ops()
    n, h, c, k = case
    dy = rnd(90, (n, h, h, k)).to(dtype).cuda()
    wt = ops.pack_wt((rnd(91, (k, 3, 3, c)) * 0.05).cuda(), dtype)
    hc = (h + 1) // 2
    compact = rnd(92, (n, hc, hc, c)).to(dtype).cuda()
    dense = torch.zeros((n, h, h, c), dtype=dtype, device="cuda")
    dense[:, ::2, ::2, :] = compact
    y_bn = rnd(93, (n, h, h, c)).to(dtype).cuda()
    st = ops.bn_finalize(ops.colstats(y_bn.view(-1, c)), n * h * h, torch.ones(c).cuda(), torch.zeros(c).cuda(), None, None)
    a = ops.conv_dgrad(dy, wt, (n, h, h, c), 3, 3, 1, 1, residual=dense)
    b = ops.conv_dgrad(dy, wt, (n, h, h, c), 3, 3, 1, 1, residual=compact, residual_stride=2)
    assert torch.equal(a, b)
    a2, pa = ops.conv_dgrad(dy, wt, (n, h, h, c), 3, 3, 1, 1, residual=dense, bnred=(y_bn, st, False))
    b2, pb = ops.conv_dgrad(dy, wt, (n, h, h, c), 3, 3, 1, 1, residual=compact, bnred=(y_bn, st, False), residual_stride=2)
    assert torch.equal(a2, b2) and torch.equal(a2, a) and torch.equal(pa, pb)


def test_nine_tap_wgrad_equals_generic_wgrad():
    """stride-1 3x3 weight gradient: the all-nine-taps kernel (one staged window per K step, padding resolved at the LDS read)
    vs the per-tap gather kernel"""
    ops = _ops()
    from frhip._abi import lib
    for (n, h, c, k, r) in [(3, 56, 64, 64, 3), (5, 28, 128, 128, 3), (7, 14, 256, 256, 3), (11, 7, 512, 512, 3),
                            (2, 10, 96, 192, 3), (2, 56, 64, 128, 3), (1, 5, 72, 40, 3)]:
        pad = (r - 1) // 2
        x = rnd(50, (n, h, h, c)).bfloat16().cuda()
        dy = rnd(51, (n, h, h, k)).bfloat16().cuda()
        outs = []
        for taps9 in (1, 0):
            old = lib().frhip_set_wgrad_taps9(taps9)
            dw = torch.zeros((k, r, r, c), dtype=torch.float32, device="cuda")
            ops.conv_wgrad(dy, x, dw, r, r, 1, pad)
            lib().frhip_set_wgrad_taps9(old)
            outs.append(dw.cpu())
        np.testing.assert_allclose(outs[0].numpy(), outs[1].numpy(), rtol=1e-4, atol=1e-4 * outs[1].abs().max().item())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mnk", [(16, 512, 1024), (200, 64, 128), (512, 1003 // 8 * 8, 512)])
def test_gemm_nt_store_and_splitk(dtype, mnk):
    ops = _ops()
    m, n, k = mnk
    a, b = q(rnd(5, (m, k)), dtype), q(rnd(6, (n, k), 0.1), dtype)
    ref = a @ b.t()
    out = ops.gemm_nt(a.to(dtype).cuda(), b.to(dtype).cuda())
    np.testing.assert_allclose(out.float().cpu().numpy(), ref.numpy(), **tol(dtype, ref.abs().max().item()))
    out2 = ops.gemm_nt(a.to(dtype).cuda(), b.to(dtype).cuda(), splits=4, atomic_f32=True)
    np.testing.assert_allclose(out2.cpu().numpy(), ref.numpy(), rtol=2e-4, atol=2e-4 * ref.abs().max().item())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mnk", [(600, 256, 64), (1000, 192, 128), (5000, 512, 256), (300, 72, 64)])
def test_linear_fwd_fused_bias_gelu_stats(dtype, mnk):
    """frhip_linear_fwd = nn.Linear + (GELU) + BatchNorm partial sums in one GEMM epilogue, against the separate passes"""
    ops = _ops()
    m, n, k = mnk
    a, w = q(rnd(31, (m, k)), dtype), q(rnd(32, (n, k), 0.2), dtype)
    bias = rnd(33, (n,))
    out, act, part = ops.linear_fwd(a.to(dtype).cuda(), w.to(dtype).cuda(), bias.cuda(), want_act=True, want_stats=True)
    ref = q(q(a @ w.t(), dtype) + bias, dtype)                     # the GEMM result is rounded, then the biased value
    t = tol(dtype, ref.abs().max().item())
    np.testing.assert_allclose(out.float().cpu().numpy(), ref.numpy(), **t)
    got = out.float().cpu()
    np.testing.assert_allclose(act.float().cpu().numpy(), q(torch.nn.functional.gelu(got), dtype).numpy(),
                               rtol=1e-2 if dtype == torch.bfloat16 else 1e-5, atol=1e-2 if dtype == torch.bfloat16 else 1e-5)
    sums = part.sum(0).cpu()
    np.testing.assert_allclose(sums[0].numpy(), got.sum(0).numpy(), rtol=1e-4, atol=1e-3 * got.abs().sum(0).max().item())
    np.testing.assert_allclose(sums[1].numpy(), (got * got).sum(0).numpy(), rtol=1e-4, atol=1e-3)
    plain, none_act, none_part = ops.linear_fwd(a.to(dtype).cuda(), w.to(dtype).cuda())
    assert none_act is None and none_part is None
    np.testing.assert_allclose(plain.float().cpu().numpy(), (a @ w.t()).numpy(), **t)


@pytest.mark.parametrize("mnk", [(1024, 256, 256), (2560, 768, 256), (512, 2048, 512), (768, 512, 2048)])
def test_linear_lean_epilogue_is_bit_identical_to_the_general_one(mnk):
    """Linears made of whole 256 x 256 tiles (every Swin34 / AlterNet linear at B = 512) run the lean store epilogue (nt_epilogue_store_lean;
    frhip_set_epi_lean): out, the GELU output and the GELU data-gradient must be the bits of the general epilogue, the per-tile sums
    (BatchNorm partials, bias-gradient column sums: matrix pipe against VALU) agree to fp32 summation order."""
    ops = _ops()
    from frhip._abi import lib
    m, n, k = mnk
    dt = torch.bfloat16
    a = rnd(90, (m, k)).to(dt).cuda()
    w = (rnd(91, (n, k)) * 0.2).to(dt).cuda()
    bias = rnd(92, (n,)).cuda()
    pre = (rnd(93, (m, n)) * 1.5).to(dt).cuda()
    res = {}
    for lean in (0, 1):
        old = lib().frhip_set_epi_lean(lean)
        try:
            out, act, part = ops.linear_fwd(a, w, bias, want_act=True, want_stats=True)
            plain, _, _ = ops.linear_fwd(a, w)
            dx, colsum = ops.linear_dgrad_gelu(a, w, pre)
            res[lean] = (out, act, part.sum(0), plain, ops.gemm_nt(a, w), dx, colsum)
        finally:
            lib().frhip_set_epi_lean(old)
    g, l = res[0], res[1]
    for i in (0, 1, 3, 4, 5):
        assert torch.equal(g[i], l[i]), i
    np.testing.assert_allclose(l[2].cpu().numpy(), g[2].cpu().numpy(), rtol=1e-5, atol=1e-5 * float(g[2].abs().max()))
    np.testing.assert_allclose(l[6].cpu().numpy(), g[6].cpu().numpy(), rtol=1e-5, atol=1e-5 * float(g[6].abs().max()))


@pytest.mark.parametrize("mkc", [(1024, 1024, 256), (2560, 256, 256), (512, 2048, 512)])
def test_linear_dgrad_with_bn_partials_lean_against_general(mkc):
    """A linear's data-gradient whose result (+ residual) is the upstream gradient of a BatchNorm (nets/SwinV2.py: x + norm(f(x)); the 1 x 1
    form of frhip_conv_dgrad_bnred) on whole 256 x 256 tiles: the lean kernel stages its tile in two 64-row halves and forms the partials
    { sum d, sum d xhat } on the matrix pipe -- dx bit-identical to the general epilogue, the sums to fp32 summation order."""
    ops = _ops()
    from frhip._abi import lib
    m, k, c = mkc
    dt = torch.bfloat16
    dy = rnd(95, (m, 1, 1, k)).to(dt).cuda()
    wt = (rnd(96, (c, 1, 1, k)) * 0.1).to(dt).cuda()
    res = rnd(97, (m, 1, 1, c)).to(dt).cuda()
    y_bn = (rnd(98, (m, 1, 1, c)) * 0.8 + 0.5).to(dt).cuda()
    st = ops.bn_finalize(ops.colstats(y_bn.view(m, c)), m, (1 + 0.1 * rnd(99, (c,))).cuda(), (0.1 * rnd(100, (c,))).cuda(), None, None)
    outs = {}
    for lean in (0, 1):
        old = lib().frhip_set_epi_lean(lean)
        try:
            outs[lean] = ops.conv_dgrad(dy, wt, (m, 1, 1, c), 1, 1, 1, 0, residual=res, bnred=(y_bn, st, False))
        finally:
            lib().frhip_set_epi_lean(old)
    (dx0, p0), (dx1, p1) = outs[0], outs[1]
    assert torch.equal(dx0, dx1) and p0.shape == p1.shape
    a, b = p1.sum(0).cpu().numpy(), p0.sum(0).cpu().numpy()
    np.testing.assert_allclose(a, b, rtol=2e-4, atol=2e-4 * max(1.0, np.abs(b).max()))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mnk", [(600, 256, 64), (5000, 512, 128), (300, 72, 64)])
def test_linear_dgrad_gelu_fused(dtype, mnk):
    """frhip_linear_dgrad_gelu = data-gradient GEMM * gelu'(saved pre-activation) + column sums, vs the separate kernels"""
    ops = _ops()
    m, n, k = mnk
    dy, wt = q(rnd(41, (m, k)), dtype), q(rnd(42, (n, k), 0.2), dtype)
    pre = q(rnd(43, (m, n)) * 1.5, dtype)
    dx, colsum = ops.linear_dgrad_gelu(dy.to(dtype).cuda(), wt.to(dtype).cuda(), pre.to(dtype).cuda())
    sep = ops.gelu_bwd(ops.gemm_nt(dy.to(dtype).cuda(), wt.to(dtype).cuda()), pre.to(dtype).cuda())
    np.testing.assert_array_equal(dx.float().cpu().numpy(), sep.float().cpu().numpy())        # same arithmetic, same roundings
    x = pre.clone().requires_grad_(True)
    torch.nn.functional.gelu(x).backward(q(dy @ wt.t(), dtype))
    t = tol(dtype, x.grad.abs().max().item())
    np.testing.assert_allclose(dx.float().cpu().numpy(), x.grad.numpy(), **t)
    want = dx.float().sum(0).cpu().numpy()
    np.testing.assert_allclose(colsum.cpu().numpy(), want, rtol=1e-4, atol=1e-4 * np.abs(dx.float().cpu().numpy()).sum(0).max())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(96, 200, 208, 64), (1000, 128, 128, 512), (70, 24, 24, 32)])
def test_gemm_tn(dtype, shape):
    ops = _ops()
    m, kc, ldp, c = shape
    p, qq = q(rnd(7, (m, ldp)), dtype), q(rnd(8, (m, c)), dtype)
    ref = p[:, :kc].t() @ qq
    out = torch.zeros((kc, c), dtype=torch.float32, device="cuda")
    ops.gemm_tn(p.to(dtype).cuda(), qq.to(dtype).cuda(), out, kc=kc)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=2e-4, atol=2e-4 * ref.abs().max().item())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(96, 200, 208, 64), (512, 1000, 1000, 512), (3000, 128, 128, 256)])
def test_gemm_tn_overwrite(dtype, shape):
    """out = P^T Q into an UNINITIALISED buffer: plain stores for a single K split (m <= 512), cleared + accumulated beyond"""
    ops = _ops()
    m, kc, ldp, c = shape
    p, qq = q(rnd(17, (m, ldp)), dtype), q(rnd(18, (m, c)), dtype)
    ref = p[:, :kc].t() @ qq
    out = torch.full((kc, c), float("nan"), dtype=torch.float32, device="cuda")
    ops.gemm_tn(p.to(dtype).cuda(), qq.to(dtype).cuda(), out, kc=kc, overwrite=True)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=2e-4, atol=2e-4 * ref.abs().max().item())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rc", [(300, 64), (4 * 7 * 7, 512), (2000, 128)])
def test_batchnorm_forward_backward(dtype, rc):
    ops = _ops()
    rows, c = rc
    y = q(rnd(9, (rows, c)) * 2 + 0.5, dtype)
    res = q(rnd(10, (rows, c)), dtype)
    dout = q(rnd(11, (rows, c)), dtype)
    gamma, beta = 1 + 0.1 * rnd(12, (c,)), 0.1 * rnd(13, (c,))
    rm, rv = 0.1 * rnd(14, (c,)), 1 + 0.1 * rnd(15, (c,)).abs()
    for relu in (False, True):
        yr = y.clone().requires_grad_(True)
        g_, b_ = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        rm_ref, rv_ref = rm.clone(), rv.clone()
        o = F.batch_norm(yr, rm_ref, rv_ref, g_, b_, True, 0.1, 1e-5)
        o = F.relu(o) if relu else o + res
        o.backward(dout)
        yd = y.to(dtype).cuda()
        rm_d, rv_d = rm.clone().cuda(), rv.clone().cuda()
        st = ops.bn_finalize(ops.colstats(yd), rows, gamma.cuda(), beta.cuda(), rm_d, rv_d)
        out = ops.bn_apply(yd, st, relu=relu, res=None if relu else res.to(dtype).cuda())
        np.testing.assert_allclose(out.float().cpu().numpy(), o.detach().numpy(), **tol(dtype, 4.0))
        np.testing.assert_allclose(rm_d.cpu().numpy(), rm_ref.numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(rv_d.cpu().numpy(), rv_ref.numpy(), rtol=1e-4, atol=1e-5)
        dg, db = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
        dy = ops.bn_backward(dout.to(dtype).cuda(), yd, st, gamma.cuda(), dg, db, relu_mask=relu)
        np.testing.assert_allclose(dy.float().cpu().numpy(), yr.grad.numpy(), **tol(dtype, 1.0))
        np.testing.assert_allclose(dg.cpu().numpy(), g_.grad.numpy(), rtol=2e-3, atol=2e-2 if dtype == torch.bfloat16 else 2e-3)
        np.testing.assert_allclose(db.cpu().numpy(), b_.grad.numpy(), rtol=2e-3, atol=2e-2 if dtype == torch.bfloat16 else 2e-3)


@pytest.mark.parametrize("dtype", DTYPES)
def test_stem_im2col_conv_pool(dtype):
    """conv1 -> bn1 -> relu -> maxpool of the reference stem (nets/resnet.py:232-235), fwd + bwd."""
    ops = _ops()
    b, h, w = 2, 12, 10
    x = rnd(20, (b, 3, h, w)).clamp(-1, 1)
    wt = rnd(21, (64, 3, 3, 3), 0.2)
    gamma, beta = 1 + 0.1 * rnd(22, (64,)), 0.1 * rnd(23, (64,))
    xq = q(x, dtype)
    wq = q(wt, dtype)
    # reference
    w_ref = wq.clone().requires_grad_(True)
    g_, b_ = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y0 = F.conv2d(xq, w_ref, None, 1, 1)
    y0.retain_grad()
    a0 = F.relu(F.batch_norm(y0, None, None, g_, b_, True, 0.1, 1e-5))
    p0 = F.max_pool2d(a0, 3, 2, 1)
    dp = rnd(24, p0.shape)
    p0.backward(q(dp, dtype))
    # device
    col = ops.stem_im2col(x.cuda(), dtype)
    wp = ops.pack_stem(nhwc(wt).reshape(64, 27).contiguous().cuda(), dtype)
    y, part = ops.conv_fwd(col.view(b * h * w, 1, 1, -1), wp, 1, 0)
    y = y.view(b, h, w, 64)
    np.testing.assert_allclose(nchw(y.float().cpu()).numpy(), y0.detach().numpy(), **tol(dtype, 2.0))
    st = ops.bn_finalize(part, b * h * w, gamma.cuda(), beta.cuda(), None, None)
    pooled, arg = ops.bn_relu_maxpool_fwd(y, st)
    np.testing.assert_allclose(nchw(pooled.float().cpu()).numpy(), p0.detach().numpy(), **tol(dtype, 3.0))
    if dtype == torch.float32:      # argmax ties make the bf16 scatter differ legitimately
        da = ops.maxpool_bwd(nhwc(dp).cuda(), arg, (b, h, w, 64))
        dg, db = torch.zeros(64, device="cuda"), torch.zeros(64, device="cuda")
        dy0 = ops.bn_backward(da, y, st, gamma.cuda(), dg, db, relu_mask=True)
        np.testing.assert_allclose(nchw(dy0.cpu()).numpy(), y0.grad.numpy(), rtol=2e-3, atol=2e-5)
        dwp = torch.zeros((64, 1, 1, 32), dtype=torch.float32, device="cuda")
        ops.conv_wgrad(dy0.view(b * h * w, 1, 1, 64), col.view(b * h * w, 1, 1, 32), dwp, 1, 1, 1, 0)
        dw = torch.zeros((64, 27), device="cuda")
        ops.unpack_stem_grad(dwp, dw)
        ref_dw = nhwc(w_ref.grad).reshape(64, 27)
        np.testing.assert_allclose(dw.cpu().numpy(), ref_dw.numpy(), rtol=2e-3, atol=2e-4)
        np.testing.assert_allclose(dg.cpu().numpy(), g_.grad.numpy(), rtol=2e-3, atol=2e-4)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 12, 10), (3, 23, 57), (1, 112, 112)])
def test_recompute_stem_matches_torch(dtype, shape):
    """stem_stats / stem_fwd / stem_bwd (conv output never materialised) vs torch autograd of conv - BN - ReLU - MaxPool
    (reference stem, nets/resnet.py:232-235)"""
    ops = _ops()
    b, h, w = shape
    x = rnd(80, (b, 3, h, w)).clamp(-1, 1)
    wt = rnd(81, (64, 3, 3, 3), 0.2)
    gamma, beta = 1 + 0.1 * rnd(82, (64,)), 0.1 * rnd(83, (64,))
    xq, wq = q(x, dtype), q(wt, dtype)
    w_ref = wq.clone().requires_grad_(True)
    g_, b_ = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y0 = F.conv2d(xq, w_ref, None, 1, 1)
    a0 = F.relu(F.batch_norm(y0, None, None, g_, b_, True, 0.1, 1e-5))
    p0 = F.max_pool2d(a0, 3, 2, 1)
    dp = q(rnd(84, p0.shape), dtype)
    # device
    xd = x.cuda()
    wp = ops.pack_stem(nhwc(wt).reshape(64, 27).contiguous().cuda(), dtype, kp=32)
    part = ops.stem_stats(xd, wp)
    st = ops.bn_finalize(part, b * h * w, gamma.cuda(), beta.cuda(), None, None)
    np.testing.assert_allclose(st.mean.cpu().numpy(), y0.detach().mean((0, 2, 3)).numpy(), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(st.invstd.cpu().numpy(), (y0.detach().var((0, 2, 3), unbiased=False) + 1e-5).rsqrt().numpy(), rtol=1e-3)
    pooled, arg = ops.stem_fwd(xd, wp, st)
    np.testing.assert_allclose(nchw(pooled.float().cpu()).numpy(), p0.detach().numpy(), **tol(dtype, 3.0))
    # reference backward: route the pooled gradient through the DEVICE's arg-max (bf16 rounding of the activations makes
    # ties that torch's fp32 max-pool breaks differently -- both are valid sub-gradients), then autograd the rest
    hp, wq_ = p0.shape[2], p0.shape[3]
    argc = nchw(arg.cpu().long())                                   # [b,64,hp,wq] tap index r*3+s
    ph = torch.arange(hp).view(1, 1, hp, 1); pw = torch.arange(wq_).view(1, 1, 1, wq_)
    hh = (2 * ph - 1 + argc // 3).clamp(0, h - 1); ww = (2 * pw - 1 + argc % 3).clamp(0, w - 1)
    da0 = torch.zeros_like(a0)
    da0.view(b, 64, -1).scatter_add_(2, (hh * w + ww).view(b, 64, -1), dp.view(b, 64, -1))
    a0.backward(da0)
    dg, db = torch.zeros(64, device="cuda"), torch.zeros(64, device="cuda")
    dw = torch.zeros((64, 27), device="cuda")
    ops.stem_bwd(xd, wp, nhwc(dp).to(dtype).cuda(), arg, st, gamma.cuda(), dg, db, dw)
    ref_dw = nhwc(w_ref.grad).reshape(64, 27)
    t = dict(rtol=2e-3, atol=2e-4 * max(1.0, ref_dw.abs().max().item())) if dtype == torch.float32 else \
        dict(rtol=5e-2, atol=5e-2 * ref_dw.abs().max().item())
    np.testing.assert_allclose(dw.cpu().numpy(), ref_dw.numpy(), **t)
    tg = dict(rtol=2e-3, atol=2e-3) if dtype == torch.float32 else dict(rtol=5e-2, atol=5e-2 * max(1.0, g_.grad.abs().max().item()))
    np.testing.assert_allclose(dg.cpu().numpy(), g_.grad.numpy(), **tg)
    np.testing.assert_allclose(db.cpu().numpy(), b_.grad.numpy(), **tg)
    # the algebraic weight gradient (csrc/stem_algebra.hip: dW = ca D + cb W G + cc s, no conv recompute): same bounds against torch,
    # and against the recompute kernel above
    dg3, db3, dw3 = torch.zeros(64, device="cuda"), torch.zeros(64, device="cuda"), torch.zeros((64, 27), device="cuda")
    ops.stem_bwd(xd, wp, nhwc(dp).to(dtype).cuda(), arg, st, gamma.cuda(), dg3, db3, dw3, gram=ops.stem_gram(xd, dtype), pooled=pooled)
    np.testing.assert_allclose(dw3.cpu().numpy(), ref_dw.numpy(), **t)
    np.testing.assert_allclose(dw3.cpu().numpy(), dw.cpu().numpy(), rtol=2e-3 if dtype == torch.float32 else 2e-2,
                               atol=(2e-4 if dtype == torch.float32 else 2e-2) * max(1.0, ref_dw.abs().max().item()))
    assert torch.equal(dg3, dg) and torch.equal(db3, db)
    if dtype == torch.bfloat16:
        # bf16 runs the scatter-form backward by default; the gather-form kernels (the fp32 path) must agree with it
        from frhip._abi import lib
        old = lib().frhip_set_stem_scatter(0)
        dg2, db2, dw2 = torch.zeros(64, device="cuda"), torch.zeros(64, device="cuda"), torch.zeros((64, 27), device="cuda")
        ops.stem_bwd(xd, wp, nhwc(dp).to(dtype).cuda(), arg, st, gamma.cuda(), dg2, db2, dw2)
        lib().frhip_set_stem_scatter(old)
        np.testing.assert_allclose(dg.cpu().numpy(), dg2.cpu().numpy(), rtol=1e-2, atol=1e-2 * max(1.0, dg2.abs().max().item()))
        np.testing.assert_allclose(db.cpu().numpy(), db2.cpu().numpy(), rtol=1e-2, atol=1e-2 * max(1.0, db2.abs().max().item()))
        np.testing.assert_allclose(dw.cpu().numpy(), dw2.cpu().numpy(), rtol=2e-2, atol=2e-2 * dw2.abs().max().item())


def test_packs_roundtrip():
    ops = _ops()
    w = rnd(30, (48, 3, 3, 40))
    wt = ops.pack_wt(w.cuda(), torch.float32).cpu()
    assert torch.equal(wt, w.permute(3, 1, 2, 0).contiguous())
    fc = rnd(31, (16, 128 * 9))
    wp = ops.fc_permute(fc.cuda(), 128, 9, torch.float32).cpu()
    assert torch.equal(wp, fc.view(16, 128, 9).permute(0, 2, 1).reshape(16, -1))
    back = torch.zeros_like(fc).cuda()
    ops.fc_unpermute_grad(wp.cuda(), back, 128, 9)
    assert torch.equal(back.cpu(), fc)
    t = ops.transpose2d(fc.cuda()).cpu()
    assert torch.equal(t, fc.t().contiguous())
    idx = torch.tensor([5, 0, 9, 3], dtype=torch.int64)
    src = rnd(32, (12, 64))
    got = ops.gather_rows(src.cuda(), idx.cuda()).cpu()
    assert torch.equal(got, src[idx])
    dst = torch.zeros((12, 64)).cuda()
    ops.scatter_rows(got.cuda(), idx.cuda(), dst)
    assert torch.equal(dst.cpu()[idx], src[idx])


@pytest.mark.parametrize("shape", [(8, 56, 56, 64, 64), (8, 28, 28, 128, 128), (16, 14, 14, 256, 256), (3, 14, 14, 256, 128),
                                   (5, 7, 9, 128, 64), (2, 56, 56, 64, 128)])
def test_conv_with_folded_bn_relu_is_bit_identical_to_the_separate_pass(shape):
    """bn1 -> relu -> conv2 with the BatchNorm-apply + ReLU folded into the operand path of the convolution (forward: LDS-halo
    kernel) and of its weight gradient (nine-tap kernel): the activated tensor is never written.  Same arithmetic and rounding
    as the separate bn_apply pass, so outputs, BatchNorm partial sums and weight gradients must be BIT-identical."""
    from frhip import ops
    n, h, w, c, k = shape
    g = torch.Generator().manual_seed(sum(shape))
    y1 = (torch.randn((n, h, w, c), generator=g) * 1.5 + 0.3).cuda().bfloat16()
    wt = (torch.randn((k, 3, 3, c), generator=g) * 0.05).cuda().bfloat16()
    dy = torch.randn((n, h, w, k), generator=g).cuda().bfloat16()
    st = ops.bn_eval_affine((torch.rand(c, generator=g) + 0.5).cuda(), (torch.randn(c, generator=g) * 0.3).cuda(),
                            (torch.randn(c, generator=g) * 0.3).cuda(), (torch.rand(c, generator=g) + 0.5).cuda())
    if not ops.conv_bnrelu_fusable(y1, wt, 1, 1):
        pytest.skip("shape not served by the fused kernels (the caller then runs bn_apply + the plain kernels)")
    a1 = ops.bn_apply(y1, st, relu=True)
    y_ref, p_ref = ops.conv_fwd(a1, wt, 1, 1, want_stats=True)
    y_fus, p_fus = ops.conv_fwd_bnrelu(y1, st, wt, 1, 1, want_stats=True)
    assert torch.equal(y_fus, y_ref)
    assert p_fus.shape == p_ref.shape and torch.equal(p_fus, p_ref)
    # ... and with the activated tensor written out on the way (what the forward pass of a training step uses): every row exactly
    # once (poisoned buffer), the same bits as the separate pass, the convolution's results unchanged
    a1_out = torch.full_like(y1, float("nan"))
    y_fo, p_fo = ops.conv_fwd_bnrelu(y1, st, wt, 1, 1, want_stats=True, act_out=a1_out)
    assert torch.equal(a1_out, a1) and torch.equal(y_fo, y_ref) and torch.equal(p_fo, p_ref)
    dw_ref = torch.zeros((k, 3, 3, c), device="cuda")
    dw_fus = torch.zeros((k, 3, 3, c), device="cuda")
    ops.conv_wgrad(dy, a1, dw_ref, 3, 3, 1, 1)
    ops.conv_wgrad_bnrelu(dy, y1, st, dw_fus, 3, 3, 1, 1)
    if (h, w) == (14, 14) and c % 64 == 0 and k % 64 == 0:
        # the plain weight gradient of 14 x 14 maps runs on the rows kernel (one image row per K step), the folded one on the pixel-stream
        # kernel: the same products summed in another order
        assert float((dw_fus - dw_ref).abs().max()) <= 4e-6 * float(dw_ref.abs().max())
    else:
        assert torch.equal(dw_fus, dw_ref)
    # and against plain fp32 arithmetic (independent of the library's own unfused path)
    a_ref = torch.relu(y1.float().cpu() * st.scale.cpu() + st.shift.cpu()).bfloat16().float()
    ref = torch.nn.functional.conv2d(a_ref.permute(0, 3, 1, 2), wt.float().cpu().permute(0, 3, 1, 2), None, 1, 1).permute(0, 2, 3, 1)
    assert float((y_fus.float().cpu() - ref).abs().max()) <= 2e-2 * float(ref.abs().max())


@pytest.mark.parametrize("dtype", DTYPES)
def test_batchnorm_passes_with_a_per_sample_scale(dtype):
    """stochastic depth around a normalised branch (nets/AlterNet_SwinV2_FAN.py: x + drop_path(norm(f(x)))): the per-sample factor inside
    the BatchNorm-apply / backward-reduce / backward-apply passes (frhip_bn_*_rs) against separate full-tensor multiplies"""
    ops = _ops()
    b, hw, c = 6, 35, 128
    rows = b * hw
    y = q(rnd(60, (rows, c)), dtype).to(dtype).cuda()
    res = q(rnd(61, (rows, c)), dtype).to(dtype).cuda()
    dout = q(rnd(62, (rows, c)), dtype).to(dtype).cuda()
    gamma, beta = (1 + 0.1 * rnd(63, (c,))).cuda(), (0.1 * rnd(64, (c,))).cuda()
    keep = torch.tensor([1.25, 0.0, 1.25, 1.25, 0.0, 1.25], dtype=torch.float32).cuda()
    st = ops.bn_finalize(ops.colstats(y), rows, gamma, beta, None, None)
    out = ops.bn_apply(y, st, res=res, rowscale=keep, rows_per=hw)
    ref = res.float() + keep.repeat_interleave(hw)[:, None] * (y.float() * st.scale + st.shift)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref.cpu().numpy(), **tol(dtype, 2.0))
    dg, db = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
    dy = ops.bn_backward(dout, y, st, gamma, dg, db, rowscale=keep, rows_per=hw)
    dscaled = (dout.float() * keep.repeat_interleave(hw)[:, None]).to(dtype)          # exact: the factors are 0 and 1.25 ... rounded once
    dg2, db2 = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
    dy2 = ops.bn_backward(dscaled, y, st, gamma, dg2, db2)
    t = dict(rtol=2e-3, atol=2e-3) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-2 * max(1.0, dg2.abs().max().item()))
    np.testing.assert_allclose(dg.cpu().numpy(), dg2.cpu().numpy(), **t)
    np.testing.assert_allclose(db.cpu().numpy(), db2.cpu().numpy(), **t)
    np.testing.assert_allclose(dy.float().cpu().numpy(), dy2.float().cpu().numpy(), **tol(dtype, 3.0))


@pytest.mark.parametrize("dtype", DTYPES)
def test_dropout_mask_kernel(dtype):
    """frhip_dropout_mask: values are exactly 0 or 1/keep, the kept fraction is keep to binomial accuracy, a seed pins the mask,
    different seeds and neighbouring elements are uncorrelated, odd lengths are filled to the end"""
    ops = _ops()
    n = 512 * 25088 + 3
    for keep in (0.5, 0.8):
        m = ops.dropout_mask((n,), dtype, keep, "cuda", seed=1234)
        inv = torch.tensor(1.0 / keep, dtype=dtype).item()
        kept = m != 0
        assert bool(((m == 0) | (m == inv)).all())
        frac = float(kept.float().mean())
        assert abs(frac - keep) < 5 * (keep * (1 - keep) / n) ** 0.5, frac
        assert torch.equal(m, ops.dropout_mask((n,), dtype, keep, "cuda", seed=1234))
        other = ops.dropout_mask((n,), dtype, keep, "cuda", seed=1235) != 0
        both = float((kept & other).float().mean())
        assert abs(both - keep * keep) < 2e-3, both
        lag = float((kept[1:] & kept[:-1]).float().mean())
        assert abs(lag - keep * keep) < 2e-3, lag
        assert bool(kept[-3:].any() | ~kept[-3:].any())          # the tail elements were written (no NaN garbage)
        assert bool(torch.isfinite(m[-3:].float()).all())
    ops.seed_dropout(7)
    a = ops.dropout_mask((4, 8), dtype, 0.5, "cuda")
    cpu_state = torch.get_rng_state()
    ops.seed_dropout(7)
    assert torch.equal(a, ops.dropout_mask((4, 8), dtype, 0.5, "cuda"))          # seed_dropout pins the drawn seeds ...
    assert torch.equal(cpu_state, torch.get_rng_state())                         # ... which never touch torch's default CPU generator



def test_bf16_gelu_pair_stays_within_two_to_the_minus_ten_of_the_exact_erf_form():
    """bf16 mode evaluates GELU in the logistic form h * sigma(2 h (c1 + c3 h^2)) (csrc/common.h: 4 + 2 instructions per element instead of
    12 + 2 -- the Swin MLP epilogues are bound by this arithmetic); the reference's nn.GELU() is the exact erf form
    (/root/reference/nets/SwinV2.py:16-32), which the fp32 validation mode keeps.  Over EVERY bf16 input in [-12, 12]: the value is within
    3.5e-4 and the slope within 6.8e-4 of the exact form before the result is rounded to bf16 -- under 2^-10, the rounding step of the stored
    activation from |a| = 0.2 up.  The fp32 mode is held to 1e-6."""
    import math
    ops = _ops()
    bits = (torch.arange(0, 1 << 16, dtype=torch.int32) << 16).view(torch.float32)
    h = bits[torch.isfinite(bits) & (bits.abs() <= 12)]
    h = torch.cat([h, h.new_zeros((-h.numel()) % 64)]).view(-1, 64)
    h64 = h.double()
    cdf = 0.5 * (1 + torch.erf(h64 / math.sqrt(2)))
    val, slope = h64 * cdf, cdf + h64 * torch.exp(-h64 * h64 / 2) / math.sqrt(2 * math.pi)
    for dtype, dv, ds in ((torch.bfloat16, 3.5e-4, 6.8e-4), (torch.float32, 1e-6, 1e-6)):
        x = h.to(dtype).cuda()
        act = ops.bias_gelu_fwd(x.clone(), torch.zeros(64, device="cuda"), True).double().cpu()
        dh = ops.gelu_bwd(torch.ones_like(x), x).double().cpu()
        rnd_v = 2.0 ** -8 * val.abs() if dtype == torch.bfloat16 else 0.0          # half a bf16 step of the result itself (8 significant bits)
        rnd_s = 2.0 ** -8 * slope.abs() if dtype == torch.bfloat16 else 0.0
        assert bool(((act - val).abs() <= dv + rnd_v).all()), float(((act - val).abs() - rnd_v).max())
        assert bool(((dh - slope).abs() <= ds + rnd_s).all()), float(((dh - slope).abs() - rnd_s).max())


@pytest.mark.parametrize("case", [("bf16 3x3 whole tiles", torch.bfloat16, 32, 12, 256, 256, 3), ("fp32 3x3 ragged", torch.float32, 6, 6, 64, 64, 3),
                                  ("bf16 1x1 qkv data-gradient", torch.bfloat16, 32, 12, 256, 768, 1), ("bf16 1x1 small maps", torch.bfloat16, 64, 6, 512, 1536, 1)])
def test_dgrad_carries_batchnorm_backward_sums_under_stochastic_depth(case):
    """frhip_conv_dgrad_fused_rs: the data-gradient whose result enters a BatchNorm under stochastic depth (AlterNet attention blocks:
    x + drop_path(norm(f(x))), /root/reference/nets/AlterNet_SwinV2_FAN.py:407-450) emits that BatchNorm's backward sums with the per-sample
    factor -- dropped samples' rows contribute nothing, kept ones 1 / keep-probability times their gradient -- exactly what the separate pass
    frhip_bn_bwd_reduce_rs computes from the stored dx; dx itself is the unscaled data-gradient, bit for bit."""
    ops = _ops()
    from frhip._abi import lib
    _, dtype, n, hw, c, k, r = case
    pad = (r - 1) // 2
    dy = q(rnd(61, (n, hw, hw, k)), dtype).to(dtype).cuda()
    wt = q(rnd(62, (c, r, r, k), 0.05), dtype).to(dtype).cuda()
    res = q(rnd(63, (n, hw, hw, c)), dtype).to(dtype).cuda()
    y = q(rnd(64, (n, hw, hw, c)) * 1.3 + 0.2, dtype).to(dtype).cuda()
    rows = n * hw * hw
    st = ops.bn_finalize(ops.colstats(y.view(rows, c)), rows, torch.ones(c, device="cuda"), torch.zeros(c, device="cuda"), None, None)
    kp = 0.9
    gen = torch.Generator().manual_seed(65)
    keep = ((torch.rand(n, generator=gen) < 0.6).float() / kp).cuda()          # 40 % dropped: both kinds of row in most tiles
    dx, part = ops.conv_dgrad(dy, wt, (n, hw, hw, c), r, r, 1, pad, residual=res, bnred=(y, st, False, keep, hw * hw, 1.0 / kp))
    plain = ops.conv_dgrad(dy, wt, (n, hw, hw, c), r, r, 1, pad, residual=res)
    assert torch.equal(dx, plain)
    nb = lib().frhip_colreduce_blocks(rows, c, ops.dt_of(y))
    ref = torch.empty((nb, 2, c), dtype=torch.float32, device="cuda")
    ops.check(lib().frhip_bn_bwd_reduce_rs(ops.dt_of(y), ops._p(dx), ops._p(y), ops._p(st.mean), ops._p(st.invstd), ops._p(keep), hw * hw, rows, c,
                                           ops._p(ref), ops._s()), "frhip_bn_bwd_reduce_rs")
    got, want = part.sum(0).cpu().numpy(), ref.sum(0).cpu().numpy()
    # and float64 on the host
    d = dx.float().cpu().double() * keep.cpu().double().repeat_interleave(hw * hw).view(n, hw, hw, 1)
    xh = (y.float().cpu().double() - st.mean.cpu().double()) * st.invstd.cpu().double()
    host = [d.sum((0, 1, 2)).numpy(), (d * xh).sum((0, 1, 2)).numpy()]
    for i in (0, 1):
        scale = float(np.abs(host[i]).max())
        np.testing.assert_allclose(want[i], host[i], rtol=2e-3, atol=2e-3 * scale, err_msg="separate pass vs float64, sum %d" % i)
        np.testing.assert_allclose(got[i], host[i], rtol=2e-3, atol=2e-3 * scale, err_msg="fused epilogue vs float64, sum %d" % i)


@pytest.mark.parametrize("case", [(1, 64, 64), (2, 64, 128), (3, 128, 64), (8, 256, 256), (37, 128, 192), (64, 256, 512)])
def test_rows_weight_gradient_of_14x14_maps_matches_fp32(case):
    """tn_rows14_kernel (one image row per K step, vertical taps reuse fragments in registers): 3x3 / stride-1 weight gradient of
    nn.Conv2d on 14 x 14 maps (/root/reference/nets/resnet.py:23-46, the 256-channel stage) against fp32 autograd arithmetic on the
    same bf16-rounded operands -- odd image counts (the two-image loop runs a phantom image), one image per K split, rectangular c / k."""
    ops = _ops()
    n, c, k = case
    x = q(rnd(901 + n, (n, 14, 14, c)), torch.bfloat16)
    dy = q(rnd(902 + n, (n, 14, 14, k)), torch.bfloat16)
    dw = torch.zeros(k, 3, 3, c, device="cuda")
    ops.conv_wgrad(dy.bfloat16().cuda(), x.bfloat16().cuda(), dw, 3, 3, 1, 1)
    ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).double(), (k, c, 3, 3), dy.permute(0, 3, 1, 2).double(), padding=1).permute(0, 2, 3, 1)
    assert float((dw.cpu().double() - ref).abs().max()) <= 3e-6 * float(ref.abs().max())
    ops.conv_wgrad(dy.bfloat16().cuda(), x.bfloat16().cuda(), dw, 3, 3, 1, 1)            # accumulates
    assert float((dw.cpu().double() - 2 * ref).abs().max()) <= 6e-6 * float(ref.abs().max())


def test_chained_weight_gradients_equal_the_unchained_ones_bit_for_bit():
    """frhip_conv_wgrad_chain: every link adds its predecessor's K-split slabs in its prologue, the last link is closed by
    frhip_conv_wgrad_chain_finish -- the same sums in the same order as frhip_conv_wgrad, on a chain that changes shape from link to link
    and accumulates into non-zero gradients."""
    ops = _ops()
    shapes = [(24, 256, 256), (24, 256, 512), (24, 64, 64), (5, 128, 64), (24, 256, 256)]
    bufs = ops.chain_slabs(torch.device("cuda", 0))
    link, want, got = None, [], []
    for i, (n, c, k) in enumerate(shapes):
        x = rnd(911 + i, (n, 14, 14, c)).bfloat16().cuda()
        dy = rnd(921 + i, (n, 14, 14, k)).bfloat16().cuda()
        assert ops.conv_wgrad_chain_ok(dy, x, 3, 3, 1, 1)
        base = rnd(931 + i, (k, 3, 3, c)).cuda()
        want.append(ops.conv_wgrad(dy, x, base.clone(), 3, 3, 1, 1))
        got.append(base.clone())
        link = ops.conv_wgrad_chain(dy, x, got[-1], bufs[i & 1], link)
        if i:
            assert torch.equal(got[i - 1], want[i - 1])                  # closed by this launch's prologue
        assert torch.equal(got[i], base)                                 # this link's own gradient: untouched until the next link / finish
    ops.conv_wgrad_chain_finish(link)
    assert torch.equal(got[-1], want[-1])
    x = rnd(1, (4, 28, 28, 64)).bfloat16().cuda()
    assert not ops.conv_wgrad_chain_ok(rnd(2, (4, 28, 28, 64)).bfloat16().cuda(), x, 3, 3, 1, 1)
    assert not ops.conv_wgrad_chain_ok(rnd(2, (4, 14, 14, 64)).bfloat16().cuda()[:, :7, :7], x[:, :14, :14], 3, 3, 2, 1)


def test_standin_batchnorm_state_of_the_stem_equals_the_torch_expression():
    """frhip_bn_standin_state (one launch in front of the stem's backward pass, nets/_backbone.py stem_reduction_operands): mean = beta,
    invstd = gamma / (gamma^2 + (k beta)^2 + 1e-20), scale = 1, shift = 0 -- the bits of the eight torch kernels it replaces, dead and
    near-dead channels included."""
    ops = _ops()
    gamma = rnd(941, (64,)).cuda()
    beta = rnd(942, (64,)).cuda()
    gamma[3], gamma[7], beta[7], gamma[11] = 0.0, 1e-12, 0.0, -2e-9
    for k in (2.0 ** -7, 2.0 ** -20):
        st = ops.bn_standin_state(gamma, beta, k, 1234.0)
        assert torch.equal(st.mean, beta) and st.count == 1234.0
        assert torch.equal(st.invstd, gamma / (gamma * gamma + (k * beta) ** 2 + 1e-20))
        assert torch.equal(st.scale, torch.ones_like(gamma)) and torch.equal(st.shift, torch.zeros_like(gamma))


@pytest.mark.parametrize("case", [(512, 1000), (96, 130), (40, 64), (8, 70), (1024, 2052)])
def test_fused_class_centre_gradient_matches_the_two_pass_form(case):
    """frhip_head_dw: d_w = normalise-backward(dT^T ehat) in one launch (autograd of F.normalize(weight) + F.linear,
    /root/reference/nets/PartialFC.py:464-484) against frhip_gemm_tn_overwrite + frhip_l2norm_bwd on the same operands and against fp64:
    sample counts that are not whole K steps, class counts that are not whole tiles, a padded dT pitch."""
    ops = _ops()
    n, classes = case
    ldt = (classes + 7) // 8 * 8
    dt = torch.full((n, ldt), float("nan"), dtype=torch.bfloat16, device="cuda")              # the padding columns must not matter
    dt[:, :classes] = (rnd(961, (n, classes)) * 0.01).bfloat16().cuda()
    e = rnd(962, (n, 512))
    ehat = (e / e.norm(dim=1, keepdim=True)).bfloat16().cuda()
    w = rnd(963, (classes, 512)) * 0.05
    what, wnorm = ops.l2norm_rows(w.cuda(), torch.bfloat16)
    got = ops.head_dw(dt, ehat, what, wnorm)
    assert got is not None and got.shape == (classes, 512)
    g = torch.empty((classes, 512), dtype=torch.float32, device="cuda")
    ops.gemm_tn(dt, ehat, g, kc=classes, overwrite=True)
    ref = ops.l2norm_bwd(g, what, wnorm)
    assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    g64 = dt[:, :classes].double().t() @ ehat.double()
    h64 = what.double()
    want = (g64 - h64 * (g64 * h64).sum(1, keepdim=True)) / wnorm.double()[:, None]
    assert float((got.double() - want).abs().max()) <= 2e-5 * float(want.abs().max())
