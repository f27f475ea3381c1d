"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's input pipeline without MotionBlur / ISONoise
(/root/reference/utils/data_partial.py:134-164): [alb.RandomGamma] -> alb.Resize -> alb.HorizontalFlip ->
alb.Normalize(0.5, 0.5) -> alb.CoarseDropout -> ToTensorV2, with the random decisions (gammas, flip flags, hole rectangles) as
explicit inputs.

PARITY UNPINNED: albumentations and OpenCV are not installed in the build container and the reference holds no fixture
for this path, so this file restates the published algorithms (cv2.resize INTER_LINEAR on 8-bit images: 11-bit fixed-point
weights, horizontal then vertical pass, ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2; albumentations Normalize =
(img - mean*255) / (std*255); CoarseDropout fill_value 0 applied after Normalize) and the HIP kernel is checked against
it.  With equal input/output sizes (the reference's face crops are already 112x112) Resize is the identity."""
import numpy as np


def _taps(n_dst, n_src):
    scale = np.float32(n_src) / np.float32(n_dst)
    d = np.arange(n_dst, dtype=np.float32)
    f = (d + np.float32(0.5)) * scale - np.float32(0.5)
    i = np.floor(f).astype(np.int64)
    f = (f - i.astype(np.float32)).astype(np.float32)
    lo = i < 0
    i[lo], f[lo] = 0, 0.0
    hi = i >= n_src - 1
    i[hi], f[hi] = n_src - 1, 0.0
    i1 = np.minimum(i + 1, n_src - 1)
    a0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int64)
    a1 = np.rint(f * np.float32(2048.0)).astype(np.int64)
    return i, i1, a0, a1


def resize_linear_u8(img, size):
    """img uint8 [H,W,3] -> uint8 [size,size,3] (cv2.INTER_LINEAR fixed-point arithmetic)"""
    h, w, _ = img.shape
    if h == size and w == size:
        return img.copy()
    y0, y1, b0, b1 = _taps(size, h)
    x0, x1, a0, a1 = _taps(size, w)
    src = img.astype(np.int64)
    s0 = src[y0][:, x0] * a0[None, :, None] + src[y0][:, x1] * a1[None, :, None]
    s1 = src[y1][:, x0] * a0[None, :, None] + src[y1][:, x1] * a1[None, :, None]
    d = (((b0[:, None, None] * (s0 >> 4)) >> 16) + ((b1[:, None, None] * (s1 >> 4)) >> 16) + 2) >> 2
    return np.clip(d, 0, 255).astype(np.uint8)


def gamma_table(gamma):
    """albumentations RandomGamma on uint8 (functional.gamma_transform): the cv2.LUT table for exponent `gamma`
    (= uniform(gamma_limit) / 100): ((arange(0, 256/255, 1/255)) ** gamma * 255) truncated to uint8"""
    return (np.power(np.arange(0, 256.0 / 255, 1.0 / 255), gamma) * 255).astype(np.uint8)[:256]


def augment(images, size, flip=None, holes=None, gamma=None):
    """images uint8 [B,H,W,3]; flip bool/int [B] or None; holes int [B,K,4] (x1,y1,x2,y2 exclusive, x2<=x1 unused) or None;
    gamma float [B] or None (NaN / <= 0: that image is left alone) -> float32 [B,3,size,size]"""
    out = np.empty((images.shape[0], 3, size, size), dtype=np.float32)
    for n, img in enumerate(images):
        if gamma is not None and np.isfinite(gamma[n]) and gamma[n] > 0:
            img = gamma_table(float(gamma[n]))[img]
        r = resize_linear_u8(img, size)
        if flip is not None and flip[n]:
            r = r[:, ::-1]
        x = (r.astype(np.float32) - np.float32(127.5)) / np.float32(127.5)
        if holes is not None:
            for x1, y1, x2, y2 in holes[n]:
                if x2 > x1 and y2 > y1:
                    x[max(y1, 0):y2, max(x1, 0):x2] = 0.0
        out[n] = x.transpose(2, 0, 1)
    return out
