"""CPU restatement of the reference backbone (TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows /root/reference/nets/resnet.py:
  * stem  conv3x3(3->64,s1) -> BN -> ReLU -> MaxPool(3,2,1)           (:186-189, :232-235)
  * four stages of BasicBlock                                          (:191-194, :211-229)
      out = bn2(conv2(relu(bn1(conv1(x))))) + (downsample(x) or x)     (:89-103)
      conv1 keeps the width, conv2 changes width/stride                (:80-85)
      downsample = conv1x1(stride) + BN when stride!=1 or width change (:214-218)
  * tail  bn2 -> flatten(NCHW order) -> fc -> bn3 (BatchNorm1d)        (:196-199, :242-246)
  * block counts: 18=[2,2,2,2] 34=[3,4,6,4] 50=[3,4,14,4] 100=[3,13,30,4] 200=[3,43,50,4] (:253-306)

Written as a stateless function over a flat state dict (reference key names) so
it shares no structure with the reference's nn.Module classes.
"""
import torch
import torch.nn.functional as F

BLOCKS = {
    "ResNet18": (2, 2, 2, 2),
    "ResNet34": (3, 4, 6, 4),
    "ResNet50": (3, 4, 14, 4),
    "ResNet100": (3, 13, 30, 4),
    "ResNet200": (3, 43, 50, 4),
}
BN_EPS = 1e-5       # nn.BatchNorm2d default
BN_MOMENTUM = 0.1   # nn.BatchNorm2d default


def _bn_spec(prefix, c):
    return [
        (prefix + ".weight", (c,), "bn_w"),
        (prefix + ".bias", (c,), "bn_b"),
        (prefix + ".running_mean", (c,), "bn_rm"),
        (prefix + ".running_var", (c,), "bn_rv"),
        (prefix + ".num_batches_tracked", (), "bn_nbt"),
    ]


def stage_plan(blocks, emd_size=512):
    """[(stage_idx(1-based), block_idx, inplanes, planes, stride, has_downsample)]"""
    plan = []
    inplanes = 64
    for si, (planes, n, stride) in enumerate(
            zip((64, 128, 256, emd_size), blocks, (1, 2, 2, 2)), start=1):
        for bi in range(n):
            s = stride if bi == 0 else 1
            ds = bi == 0 and (s != 1 or inplanes != planes)
            plan.append((si, bi, inplanes, planes, s, ds))
            inplanes = planes
    return plan


def resnet_spec(blocks, emd_size=512, spatial=7):
    """(name, shape, kind) in the reference's state_dict order."""
    spec = [("conv1.weight", (64, 3, 3, 3), "conv")] + _bn_spec("bn1", 64)
    for si, bi, cin, cout, s, ds in stage_plan(blocks, emd_size):
        p = "layer%d.%d" % (si, bi)
        spec.append((p + ".conv1.weight", (cin, cin, 3, 3), "conv"))
        spec += _bn_spec(p + ".bn1", cin)
        spec.append((p + ".conv2.weight", (cout, cin, 3, 3), "conv"))
        spec += _bn_spec(p + ".bn2", cout)
        if ds:
            spec.append((p + ".downsample.0.weight", (cout, cin, 1, 1), "conv"))
            spec += _bn_spec(p + ".downsample.1", cout)
    spec += _bn_spec("bn2", emd_size)
    spec.append(("fc.weight", (emd_size, emd_size * spatial * spatial), "linear_w"))
    spec.append(("fc.bias", (emd_size,), "linear_b"))
    spec += _bn_spec("bn3", emd_size)
    return spec


def _bn(sd, prefix, x, training):
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    y = F.batch_norm(x, rm, rv, sd[prefix + ".weight"], sd[prefix + ".bias"],
                     training, BN_MOMENTUM, BN_EPS)
    if training:
        sd[prefix + ".num_batches_tracked"] += 1
    return y


class _StorageCast(torch.autograd.Function):
    """straight-through rounding to a storage dtype: the value AND its gradient pass through that dtype"""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.dtype = dtype
        return x.to(dtype).float()

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dtype).float(), None


def storage_cast(dtype):
    """q for resnet_forward(..., q=): emulates tensors kept in `dtype` (e.g. torch.bfloat16) between fp32-accumulating ops --
    what a mixed-precision implementation of the same network does by construction.  Isolates the error that comes from the
    STORAGE precision (which any such implementation has) from the error of a particular kernel."""
    return lambda t: _StorageCast.apply(t, dtype)


def _id(t):
    return t


FP8_MAX = 448.0


def fp8_e4m3(t):
    """value of t after a round trip through OCP e4m3 (clamped at +-448 like the kernels' packer)"""
    return t.clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).float()


class _Fp8FwdConv(torch.autograd.Function):
    """Emulation of the fp8 WEIGHT path of the build (cfg 5) -- test infrastructure, plain PyTorch CPU ops: the forward
    convolution multiplies e4m3 operands (activations under the static scale 1.0, weights under one scale per output channel =
    amax / 448), the backward pass is the unquantised one (the build's backward kernels read the bf16 tensors)."""

    @staticmethod
    def forward(ctx, x, w, stride, pad):
        ctx.save_for_backward(x, w)
        ctx.sp = (stride, pad)
        sc = w.abs().amax(dim=(1, 2, 3), keepdim=True) / FP8_MAX
        sc = torch.where(sc > 0, sc, torch.ones_like(sc))
        return F.conv2d(fp8_e4m3(x), fp8_e4m3(w / sc) * sc, None, stride, pad)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        stride, pad = ctx.sp
        gx = torch.nn.grad.conv2d_input(x.shape, w, g, stride=stride, padding=pad)
        gw = torch.nn.grad.conv2d_weight(x, w.shape, g, stride=stride, padding=pad)
        return gx, gw, None, None


def _conv(x, w, stride, pad, fp8):
    """fp8: the build's eligibility rule (input width a multiple of 128: one K step of the fp8 kernels)"""
    if fp8 and x.shape[1] % 128 == 0:
        return _Fp8FwdConv.apply(x, w, stride, pad)
    return F.conv2d(x, w, None, stride, pad)


def basic_block(sd, p, x, stride, has_ds, training, q=_id, fp8=False):
    y = q(_conv(x, q(sd[p + ".conv1.weight"]), 1, 1, fp8))
    y = q(F.relu(_bn(sd, p + ".bn1", y, training)))
    y = q(_conv(y, q(sd[p + ".conv2.weight"]), stride, 1, fp8))
    y = _bn(sd, p + ".bn2", y, training)
    if has_ds:
        r = q(_conv(x, q(sd[p + ".downsample.0.weight"]), stride, 0, fp8))
        r = _bn(sd, p + ".downsample.1", r, training)
    else:
        r = x
    return q(y + r)


def resnet_forward(sd, x, blocks, training, emd_size=512, q=_id, fp8=False):
    """x float32 [B,3,H,W] -> [B, emd_size]; running stats in `sd` updated in place when training.
    q: storage cast applied where a mixed-precision implementation keeps a tensor (conv outputs, activation outputs, block
    outputs, weights as GEMM operands); identity = the reference's fp32 arithmetic.
    fp8: emulate the build's fp8 forward weight path in the body convolutions with >= 128 input channels (_Fp8FwdConv)."""
    y = F.conv2d(q(x), q(sd["conv1.weight"]), None, 1, 1)
    y = F.relu(_bn(sd, "bn1", y, training))
    y = q(F.max_pool2d(y, 3, 2, 1))
    for si, bi, cin, cout, s, ds in stage_plan(blocks, emd_size):
        y = basic_block(sd, "layer%d.%d" % (si, bi), y, s, ds, training, q, fp8)
    y = q(_bn(sd, "bn2", y, training))
    y = y.reshape(y.shape[0], -1)
    y = F.linear(y, q(sd["fc.weight"]), sd["fc.bias"])
    y = _bn(sd, "bn3", y, training)
    return y


def trainable_names(sd):
    return [k for k in sd if not (k.endswith("running_mean") or k.endswith("running_var")
                                  or k.endswith("num_batches_tracked"))]
