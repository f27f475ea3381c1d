"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's input pipeline
(/root/reference/utils/data_partial.py:134-164): [alb.RandomGamma] -> [alb.MotionBlur] -> [alb.ISONoise] -> alb.Resize ->
alb.HorizontalFlip -> alb.Normalize(0.5, 0.5) -> alb.CoarseDropout -> ToTensorV2, with the random decisions (gammas, blur
kernels, noise draws, flip flags, hole rectangles) as explicit inputs.

PARITY UNPINNED: albumentations and OpenCV are not installed in the build container and the reference holds no fixture
for this path, so this file restates the published algorithms (cv2.resize INTER_LINEAR on 8-bit images: 11-bit fixed-point
weights, horizontal then vertical pass, ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2; albumentations Normalize =
(img - mean*255) / (std*255); CoarseDropout fill_value 0 applied after Normalize) and the HIP kernel is checked against
it.  With equal input/output sizes (the reference's face crops are already 112x112) Resize is the identity.
MotionBlur = albumentations 1.x `MotionBlur.get_params` (cv2.line on a k x k grid, normalised) + `cv2.filter2D` (correlation,
BORDER_REFLECT_101, cvRound); ISONoise = albumentations 1.x `functional.iso_noise` over OpenCV's float RGB<->HLS
(imgproc color_hsv: RGB2HLS_f / HLS2RGB_f) -- restated from those packages' published sources, equally unpinned."""
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


def line_points(xs, ys, xe, ye):
    """pixels of cv2.line((xs, ys) -> (xe, ye), thickness 1, 8-connected): OpenCV LineIterator -- count = max(|dx|, |dy|) + 1 points,
    err starts at dmajor - 2 dminor; per point: step the minor axis iff err < 0, then err += -2 dminor (+ 2 dmajor if stepped)"""
    dx, dy = xe - xs, ye - ys
    ax, ay = abs(dx), abs(dy)
    major_is_y = ay > ax
    dmaj, dmin = (ay, ax) if major_is_y else (ax, ay)
    smaj = (1 if dy >= 0 else -1) if major_is_y else (1 if dx >= 0 else -1)
    smin = (1 if dx >= 0 else -1) if major_is_y else (1 if dy >= 0 else -1)
    pts, maj, mino, err = [], (ys if major_is_y else xs), (xs if major_is_y else ys), dmaj - 2 * dmin
    for _ in range(dmaj + 1):
        pts.append((mino, maj) if major_is_y else (maj, mino))
        if err < 0:
            mino += smin
            err += 2 * dmaj
        err -= 2 * dmin
        maj += smaj
    return pts


def motion_kernel(ksize, xs, ys, xe, ye):
    """float32 [ksize, ksize]: the normalised line kernel of alb.MotionBlur"""
    k = np.zeros((ksize, ksize), dtype=np.float32)
    for x, y in line_points(xs, ys, xe, ye):
        k[y, x] = 1.0
    return k / k.sum()


def _reflect101(i, n):
    i = np.asarray(i)
    if n == 1:
        return np.zeros_like(i)
    period = 2 * (n - 1)
    i = np.mod(i, period)
    return np.where(i >= n, period - i, i)


def motion_blur(img, kernel):
    """cv2.filter2D(img uint8 [H,W,3], -1, kernel [k,k] float32): correlation, anchor centre, reflect-101 border, float32
    accumulation over the non-zero taps in row-major order, round-half-even, saturate"""
    h, w, _ = img.shape
    k = kernel.shape[0]
    a = k // 2
    acc = np.zeros(img.shape, dtype=np.float32)
    src = img.astype(np.float32)
    for i in range(k):
        yy = _reflect101(np.arange(h) + i - a, h)
        for j in range(k):
            if kernel[i, j] == 0:
                continue
            xx = _reflect101(np.arange(w) + j - a, w)
            acc = (acc + np.float32(kernel[i, j]) * src[yy][:, xx]).astype(np.float32)
    return np.clip(np.rint(acc), 0, 255).astype(np.uint8)


def rgb2hls(rgb):
    """float32 [...,3] in [0,1] -> (h [0,360], l, s): OpenCV RGB2HLS_f"""
    r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    vmax, vmin = np.maximum(np.maximum(r, g), b), np.minimum(np.minimum(r, g), b)
    diff = (vmax - vmin).astype(np.float32)
    l = ((vmax + vmin) * np.float32(0.5)).astype(np.float32)
    live = diff > np.float32(1.1920929e-07)
    safe = np.where(live, diff, np.float32(1))
    with np.errstate(divide="ignore", invalid="ignore"):
        s = np.where(l < np.float32(0.5), safe / (vmax + vmin), safe / (np.float32(2) - vmax - vmin)).astype(np.float32)
    d = (np.float32(60) / safe).astype(np.float32)
    h = np.where(vmax == r, (g - b) * d, np.where(vmax == g, (b - r) * d + np.float32(120), (r - g) * d + np.float32(240))).astype(np.float32)
    h = np.where(h < 0, h + np.float32(360), h).astype(np.float32)
    return np.where(live, h, np.float32(0)).astype(np.float32), l, np.where(live, s, np.float32(0)).astype(np.float32)


def hls2rgb(h, l, s):
    """OpenCV HLS2RGB_f -> float32 [...,3]"""
    p2 = np.where(l <= np.float32(0.5), l * (np.float32(1) + s), l + s - l * s).astype(np.float32)
    p1 = (np.float32(2) * l - p2).astype(np.float32)
    hh = (h * np.float32(6.0 / 360.0)).astype(np.float32)
    for _ in range(4):                      # "do h += 6 while h < 0" / "do h -= 6 while h >= 6": inputs are within a few periods
        hh = np.where(hh < 0, hh + np.float32(6), np.where(hh >= np.float32(6), hh - np.float32(6), hh)).astype(np.float32)
    sector = np.floor(hh).astype(np.int64)
    f = (hh - sector.astype(np.float32)).astype(np.float32)
    bad = (sector < 0) | (sector >= 6)
    sector, f = np.where(bad, 0, sector), np.where(bad, np.float32(0), f).astype(np.float32)
    tab = np.stack([p2, p1, (p1 + (p2 - p1) * (np.float32(1) - f)).astype(np.float32), (p1 + (p2 - p1) * f).astype(np.float32)], axis=-1)
    sd = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])          # (b, g, r) table slots per sector
    pick = sd[sector]
    b = np.take_along_axis(tab, pick[..., 0:1], axis=-1)[..., 0]
    g = np.take_along_axis(tab, pick[..., 1:2], axis=-1)[..., 0]
    r = np.take_along_axis(tab, pick[..., 2:3], axis=-1)[..., 0]
    grey = s == 0
    return np.stack([np.where(grey, l, r), np.where(grey, l, g), np.where(grey, l, b)], axis=-1).astype(np.float32)


def iso_lambda(img, intensity):
    """Poisson mean of the luminance noise: std of the L channel (cv2.meanStdDev: population std, float64) * intensity * 255"""
    _, l, _ = rgb2hls(img.astype(np.float32) * np.float32(1.0 / 255.0))
    return float(np.sqrt(max(np.mean(l.astype(np.float64) ** 2) - np.mean(l.astype(np.float64)) ** 2, 0.0)) * float(np.float32(intensity)) * 255.0)


def iso_noise(img, lum_noise, color_noise):
    """albumentations functional.iso_noise with the draws as inputs: lum_noise int [H,W] ~ Poisson(iso_lambda), color_noise
    float32 [H,W] ~ N(0, color_shift * 360 * intensity) -> uint8 [H,W,3]"""
    h, l, s = rgb2hls(img.astype(np.float32) * np.float32(1.0 / 255.0))
    h = (h + color_noise.astype(np.float32)).astype(np.float32)
    h = np.where(h < 0, h + np.float32(360), h).astype(np.float32)
    h = np.where(h > np.float32(360), h - np.float32(360), h).astype(np.float32)
    l = (l + (lum_noise.astype(np.float32) / np.float32(255)) * (np.float32(1) - l)).astype(np.float32)
    rgb = hls2rgb(h, l, s) * np.float32(255)
    return np.clip(rgb, 0, 255).astype(np.uint8)          # astype(uint8) truncates; the clip only guards draws > 255


def augment(images, size, flip=None, holes=None, gamma=None, blur=None, iso=None):
    """images uint8 [B,H,W,3]; flip bool/int [B] or None; holes int [B,K,4] (x1,y1,x2,y2 exclusive, x2<=x1 unused) or None;
    gamma float [B] or None (NaN / <= 0: that image is left alone); blur = list of [k,k] kernels or None per image;
    iso = list of (lum_noise, color_noise) or None per image -> float32 [B,3,size,size]"""
    out = np.empty((images.shape[0], 3, size, size), dtype=np.float32)
    for n, img in enumerate(images):
        if gamma is not None and np.isfinite(gamma[n]) and gamma[n] > 0:
            img = gamma_table(float(gamma[n]))[img]
        if blur is not None and blur[n] is not None:
            img = motion_blur(img, blur[n])
        if iso is not None and iso[n] is not None:
            img = iso_noise(img, iso[n][0], iso[n][1])
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
