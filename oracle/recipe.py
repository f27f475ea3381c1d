"""Portable deterministic tensors for fixtures (numpy PCG64; no torch RNG streams).

Both tools/make_golden.py (which feeds the real reference) and the tests (which
feed the oracle and the HIP path) build their inputs with these functions, so a
fixture only has to store outputs.
"""
import math
import numpy as np
import torch


def rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def normal(seed, shape, std=1.0, mean=0.0):
    a = rng(seed).standard_normal(size=tuple(shape), dtype=np.float64) * std + mean
    return torch.from_numpy(a.astype(np.float32))


def images(seed, b, h=112, w=112):
    """float32 [b,3,h,w] ~ N(0,1) clipped to [-1,1] (post Normalize(0.5,0.5) range,
    reference utils/data_partial.py:151)."""
    return normal(seed, (b, 3, h, w)).clamp_(-1.0, 1.0)


def labels(seed, n, num_classes):
    a = rng(seed).integers(0, num_classes, size=(n,), dtype=np.int64)
    return torch.from_numpy(a)


def fill_state(spec, seed):
    """spec: list of (name, shape, kind).  Returns an OrderedDict name -> tensor.

    kinds: conv / linear_w (xavier-normal std, as reference nets/resnet.py:201-209 but
    from the portable stream), linear_b, bn_w, bn_b, bn_rm, bn_rv, bn_nbt.
    BN tensors are perturbed away from (1, 0, 0, 1) so eval-mode parity is not vacuous.
    """
    from collections import OrderedDict
    g = rng(seed)
    out = OrderedDict()
    for name, shape, kind in spec:
        shape = tuple(shape)
        if kind in ("conv", "linear_w"):
            recept = int(np.prod(shape[2:])) if len(shape) > 2 else 1
            fan_in, fan_out = shape[1] * recept, shape[0] * recept
            std = math.sqrt(2.0 / (fan_in + fan_out))
            a = g.standard_normal(size=shape) * std
        elif kind == "linear_b":
            a = g.standard_normal(size=shape) * 0.01
        elif kind == "bn_w":
            a = 1.0 + 0.1 * g.standard_normal(size=shape)
        elif kind == "bn_b":
            a = 0.1 * g.standard_normal(size=shape)
        elif kind == "bn_rm":
            a = 0.1 * g.standard_normal(size=shape)
        elif kind == "bn_rv":
            a = 1.0 + 0.1 * np.abs(g.standard_normal(size=shape))
        elif kind == "bn_nbt":
            out[name] = torch.zeros((), dtype=torch.int64)
            continue
        elif kind in ("coords", "posidx", "logit_scale") or ":" in kind:     # deterministic tables: see oracle.*_ref.fill_special
            out[name] = torch.zeros(shape)
            continue
        else:
            raise ValueError(kind)
        out[name] = torch.from_numpy(np.asarray(a, dtype=np.float32))
    return out


def summary(t, k=8):
    """Compact checksum of a tensor for fixtures: [sum, l2, first k elements]."""
    t = t.detach().double().flatten()
    head = t[:k]
    if head.numel() < k:
        head = torch.cat([head, torch.zeros(k - head.numel(), dtype=torch.float64)])
    return torch.cat([t.sum().view(1), t.norm().view(1), head]).numpy()


def probe_positions(numel, k=256):
    """portable positions a probe() reads: every element of a small tensor, else k positions drawn from PCG64 seeded by the size"""
    if numel <= k:
        return np.arange(numel, dtype=np.int64)
    return rng(0x9E3779B1 ^ numel).integers(0, numel, size=(k,), dtype=np.int64)


def probe(t, k=256):
    """Checksum of a tensor that a permutation of its elements cannot pass: [sum, l2, the elements at probe_positions()]
    of the tensor flattened in its LOGICAL (row-major over .shape) order."""
    t = t.detach().double().reshape(-1)
    pos = torch.from_numpy(probe_positions(t.numel(), k))
    return torch.cat([t.sum().view(1), t.norm().view(1), t[pos]]).numpy()
