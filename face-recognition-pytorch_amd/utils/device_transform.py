"""Input pipeline on the device: the reference's albumentations chain
(/root/reference/utils/data_partial.py:134-164 -- RandomGamma, MotionBlur, ISONoise, Resize, HorizontalFlip, Normalize(0.5, 0.5),
CoarseDropout, ToTensorV2) as libfrhip kernels on a uint8 HWC batch, so the loader only has to decode JPEGs and hand over bytes:
one fused kernel for gamma / resize / flip / normalize / dropout, preceded by a blur pass and a noise pass (two launches) when
those two augmentations are configured.

    t = DeviceTransform(conf)                       # conf.img_size, conf.data_augmentation, conf.img_augmenation.*
    x = t(batch_u8_hwc_cuda)                        # -> float32 [B,3,S,S] for Model.training_step

The random decisions are drawn on the host with the semantics of albumentations' transforms (RandomGamma: with probability
gamma_p an exponent uniform(gamma_s) / 100, applied as a 256-entry table; HorizontalFlip p = 0.5; CoarseDropout: with
probability erase_p, randint(min_holes, max_holes) holes of height/width randint(1..max_h/max_w) at uniform positions) from a
numpy Generator the caller can seed.  MotionBlur: with probability blur_p a k x k kernel (k in {3, 5, 7}) holding a random
one-pixel line (cv2.line between two random grid points), normalised, applied as cv2.filter2D.  ISONoise: with probability iso_p
color_shift ~ uniform(c_shift), intensity ~ uniform(intensity); the per-pixel Poisson / normal draws of that transform are made
ON THE DEVICE from a per-image seed (12.8 M draws per batch of 512 would bind the host): the noise has the transform's
distributions, not numpy's stream -- statistical parity only, like DropPath / Dropout.
There is no CPU fallback: the batch must live on the MI355X."""
import numpy as np
import torch

from frhip import ops
from frhip._abi import check, lib


class DeviceTransform:
    def __init__(self, conf, seed=None, train=True):
        self.size = int(conf.img_size)
        aug = list(getattr(conf, "data_augmentation", [])) if train else []
        self.flip = "RandomHorizontalFlip" in aug
        self.erase = "RandomErasing" in aug
        self.gamma = "RandomGammaContrast" in aug
        self.blur = "RandomMotionBlur" in aug
        self.iso = "ISONoise" in aug
        ia = getattr(conf, "img_augmenation", None)
        if self.blur:
            self.blur_p = float(ia.blur_p)
        if self.iso:
            self.iso_p = float(ia.iso_p)
            self.c_shift = tuple(float(v) for v in ia.c_shift)
            self.intensity = tuple(float(v) for v in ia.intensity)
        if self.gamma:
            self.gamma_p = float(ia.gamma_p)
            self.gamma_lo, self.gamma_hi = (float(v) for v in ia.gamma_s)
        if self.erase:
            self.erase_p = float(ia.erase_p)
            self.min_holes, self.max_holes = int(ia.erase_min_holes), int(ia.erase_max_holes)
            self.max_h, self.max_w = int(ia.erase_max_h), int(ia.erase_max_w)
        self.rng = np.random.default_rng(seed)

    def draw_gamma(self, batch):
        """-> float64 [B] exponents (NaN = transform not applied to that image) or None"""
        if not self.gamma:
            return None
        g = np.full(batch, np.nan)
        for n in range(batch):
            if self.rng.random() < self.gamma_p:
                g[n] = self.rng.uniform(self.gamma_lo, self.gamma_hi) / 100.0
        return g

    @staticmethod
    def gamma_tables(gamma):
        """uint8 [B,256]: albumentations' cv2.LUT table per image (identity where gamma is NaN)"""
        lut = np.tile(np.arange(256, dtype=np.uint8), (len(gamma), 1))
        for n, gv in enumerate(gamma):
            if np.isfinite(gv) and gv > 0:
                lut[n] = (np.power(np.arange(0, 256.0 / 255, 1.0 / 255), float(gv)) * 255).astype(np.uint8)[:256]
        return lut

    @staticmethod
    def line_kernel(ksize, xs, ys, xe, ye):
        """float32 [7,7]: albumentations MotionBlur's kernel -- cv2.line(zeros(k, k), (xs, ys), (xe, ye), 1, thickness=1) / sum,
        zero-padded to 7 x 7 with the centre at (3, 3).  The line is OpenCV's 8-connected LineIterator (integer Bresenham:
        one step along the major axis per point, a minor-axis step whenever the error term is negative)."""
        k = np.zeros((ksize, ksize), dtype=np.float32)
        dx, dy = xe - xs, ye - ys
        sx, sy = (1 if dx >= 0 else -1), (1 if dy >= 0 else -1)
        dx, dy = abs(dx), abs(dy)
        steep = dy > dx
        if steep:
            dx, dy = dy, dx
        err, x, y = dx - 2 * dy, xs, ys
        for _ in range(dx + 1):
            k[y, x] = 1.0
            minor = err < 0
            err += -2 * dy + (2 * dx if minor else 0)
            if steep:
                y += sy
                x += sx if minor else 0
            else:
                x += sx
                y += sy if minor else 0
        k /= k.sum()
        out = np.zeros((7, 7), dtype=np.float32)
        o = 3 - ksize // 2
        out[o:o + ksize, o:o + ksize] = k
        return out

    def draw_blur(self, batch):
        """-> (ksize int32 [B] (0 = not applied), kernels float32 [B,7,7]) or None"""
        if not self.blur:
            return None
        ks = np.zeros(batch, dtype=np.int32)
        kern = np.zeros((batch, 7, 7), dtype=np.float32)
        for n in range(batch):
            if self.rng.random() >= self.blur_p:
                continue
            k = int(self.rng.choice([3, 5, 7]))                              # blur_limit (3, 7), odd sizes
            xs, xe = int(self.rng.integers(0, k)), int(self.rng.integers(0, k))
            if xs == xe:
                ys, ye = (int(v) for v in self.rng.choice(k, size=2, replace=False))
            else:
                ys, ye = int(self.rng.integers(0, k)), int(self.rng.integers(0, k))
            ks[n], kern[n] = k, self.line_kernel(k, xs, ys, xe, ye)
        return ks, kern

    def draw_iso(self, batch):
        """-> (params float32 [B,2] = (color_shift, intensity), intensity 0 = not applied; seeds uint64 [B]) or None"""
        if not self.iso:
            return None
        params = np.zeros((batch, 2), dtype=np.float32)
        for n in range(batch):
            if self.rng.random() < self.iso_p:
                params[n] = (self.rng.uniform(*self.c_shift), self.rng.uniform(*self.intensity))
        seeds = self.rng.integers(0, 2 ** 63, size=batch, dtype=np.uint64)
        return params, seeds

    def blur_noise(self, images, gamma=None, blur=None, iso=None, iso_noise=None):
        """the chain's first three transforms on the uint8 HWC batch: RandomGamma (table) -> MotionBlur -> ISONoise -> uint8 HWC.
        iso_noise = (int32 [B,H,W] Poisson draws, float32 [B,H,W] scaled normal draws) replaces the device generator (tests)."""
        b, h, w, _ = images.shape
        dev = images.device

        def put(a, dt):
            return torch.as_tensor(np.ascontiguousarray(a, dtype=dt)).to(dev, non_blocking=True)

        lut = None if gamma is None else torch.as_tensor(self.gamma_tables(gamma)).to(dev, non_blocking=True)
        x = images
        if blur is not None or lut is not None:
            ks = kern = None
            if blur is not None:
                ks, kern = put(blur[0], np.int32), put(blur[1], np.float32)
            y = torch.empty_like(x)
            check(lib().frhip_motion_blur_u8(ops._p(x), ops._p(y), ops._p(lut), ops._p(kern), ops._p(ks), b, h, w, ops._s()),
                  "frhip_motion_blur_u8")
            x = y
        if iso is not None:
            params = put(iso[0], np.float32)
            seeds = torch.as_tensor(np.ascontiguousarray(iso[1]).view(np.int64)).to(dev, non_blocking=True)
            ln = cn = None
            if iso_noise is not None:
                ln, cn = put(iso_noise[0], np.int32), put(iso_noise[1], np.float32)
            scratch = torch.empty(lib().frhip_iso_noise_scratch_doubles(b), dtype=torch.float64, device=dev)
            y = torch.empty_like(x)
            check(lib().frhip_iso_noise_u8(ops._p(x), ops._p(y), ops._p(scratch), ops._p(params), ops._p(ln), ops._p(cn),
                                           ops._p(seeds), b, h, w, ops._s()), "frhip_iso_noise_u8")
            x = y
        return x

    def draw(self, batch):
        """-> (flip int32 [B] or None, holes int32 [B,K,4] or None) for one batch"""
        flip = holes = None
        if self.flip:
            flip = (self.rng.random(batch) < 0.5).astype(np.int32)
        if self.erase:
            k = max(self.max_holes, 1)
            holes = np.zeros((batch, k, 4), dtype=np.int32)
            for n in range(batch):
                if self.rng.random() >= self.erase_p:
                    continue
                for j in range(int(self.rng.integers(self.min_holes, self.max_holes + 1))):
                    hh, ww = int(self.rng.integers(1, self.max_h + 1)), int(self.rng.integers(1, self.max_w + 1))
                    y1, x1 = int(self.rng.integers(0, self.size - hh + 1)), int(self.rng.integers(0, self.size - ww + 1))
                    holes[n, j] = (x1, y1, x1 + ww, y1 + hh)
        return flip, holes

    def apply(self, images, flip=None, holes=None, gamma=None, blur=None, iso=None, iso_noise=None):
        """images uint8 [B,H,W,3] on the GPU, explicit decisions -> float32 [B,3,S,S]"""
        if not (images.is_cuda and images.dtype == torch.uint8 and images.dim() == 4 and images.shape[3] == 3):
            raise RuntimeError("DeviceTransform: a uint8 [B,H,W,3] batch on the MI355X is required (there is no CPU path)")
        images = images.contiguous()
        if blur is not None or iso is not None:          # gamma is folded into the blur pass (chain order: gamma -> blur -> noise)
            images, gamma = self.blur_noise(images, gamma, blur, iso, iso_noise), None
        b, h, w, _ = images.shape
        dev = images.device
        out = torch.empty((b, 3, self.size, self.size), dtype=torch.float32, device=dev)
        f = None if flip is None else torch.as_tensor(np.ascontiguousarray(flip, dtype=np.int32)).to(dev, non_blocking=True)
        hl = None if holes is None else torch.as_tensor(np.ascontiguousarray(holes, dtype=np.int32)).to(dev, non_blocking=True)
        nh = 0 if hl is None else int(hl.shape[1])
        if gamma is not None:
            lut = torch.as_tensor(self.gamma_tables(gamma)).to(dev, non_blocking=True)
            check(lib().frhip_augment_u8_lut(ops._p(images), ops._p(lut), ops._p(out), ops._p(f), ops._p(hl), nh, b, h, w,
                                             self.size, ops._s()), "frhip_augment_u8_lut")
        else:
            check(lib().frhip_augment_u8(ops._p(images), ops._p(out), ops._p(f), ops._p(hl), nh, b, h, w, self.size, ops._s()),
                  "frhip_augment_u8")
        return out

    def __call__(self, images):
        gamma = self.draw_gamma(images.shape[0])          # chain order: RandomGamma first
        blur = self.draw_blur(images.shape[0])
        iso = self.draw_iso(images.shape[0])
        flip, holes = self.draw(images.shape[0])
        return self.apply(images, flip, holes, gamma, blur, iso)
