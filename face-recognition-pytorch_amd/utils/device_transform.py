"""Input pipeline on the device: the reference's albumentations chain
(/root/reference/utils/data_partial.py:134-164 -- RandomGamma, Resize, HorizontalFlip, Normalize(0.5, 0.5), CoarseDropout,
ToTensorV2) as one libfrhip kernel on a uint8 HWC batch, so the loader only has to decode JPEGs and hand over bytes.

    t = DeviceTransform(conf)                       # conf.img_size, conf.data_augmentation, conf.img_augmenation.*
    x = t(batch_u8_hwc_cuda)                        # -> float32 [B,3,S,S] for Model.training_step

The random decisions are drawn on the host with the semantics of albumentations' transforms (RandomGamma: with probability
gamma_p an exponent uniform(gamma_s) / 100, applied as a 256-entry table; HorizontalFlip p = 0.5; CoarseDropout: with
probability erase_p, randint(min_holes, max_holes) holes of height/width randint(1..max_h/max_w) at uniform positions) from a
numpy Generator the caller can seed.  MotionBlur / ISONoise of the reference chain (a random line-kernel convolution and
HLS-space Poisson noise on the decoded image) stay with the decoder on the CPU (not built here).
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
        ia = getattr(conf, "img_augmenation", None)
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

    def apply(self, images, flip=None, holes=None, gamma=None):
        """images uint8 [B,H,W,3] on the GPU, explicit decisions -> float32 [B,3,S,S]"""
        if not (images.is_cuda and images.dtype == torch.uint8 and images.dim() == 4 and images.shape[3] == 3):
            raise RuntimeError("DeviceTransform: a uint8 [B,H,W,3] batch on the MI355X is required (there is no CPU path)")
        images = images.contiguous()
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
        flip, holes = self.draw(images.shape[0])
        return self.apply(images, flip, holes, gamma)
