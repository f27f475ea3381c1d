// Device input pipeline: the reference's albumentations chain without the two convolution /
// noise transforms,
//   [RandomGamma] -> Resize(img_size) -> HorizontalFlip -> Normalize(0.5, 0.5) -> CoarseDropout -> ToTensorV2
// (/root/reference/utils/data_partial.py:134-164) as ONE kernel on uint8 HWC images, writing the NCHW fp32 batch the
// backbones consume.  The random decisions (flip?, which holes) are inputs: the host draws them (as albumentations
// does), the kernel is deterministic.  HBM-bound: reads 3 B, writes 12 B per output pixel.
//
// Resize = OpenCV INTER_LINEAR for 8-bit images restated from its published algorithm (albumentations calls
// cv2.resize): source coordinate (d + 0.5) * scale - 0.5, taps clamped to the image, weights quantised to 11 bits,
// horizontal pass in int32, vertical pass ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.  With equal sizes
// it is the identity.  RandomGamma on uint8 images is a 256-entry look-up table per image (albumentations:
// cv2.LUT(img, ((arange(256) / 255) ** gamma * 255).astype(uint8))): the host builds the tables, the kernel applies them to
// the SOURCE bytes, i.e. before the resize, as the chain does.  cv2 / albumentations are not available in the build container: this restatement is NOT pinned
// against them (oracle/augment_ref.py carries the same note).
#include "common.h"
#include "frhip.h"

namespace frhip {

struct ResizeTap { int i0, i1; int a0, a1; };      // source indices and 11-bit weights

__device__ __forceinline__ ResizeTap resize_tap(int d, int n_src, float scale) {
    float f = ((float)d + 0.5f) * scale - 0.5f;
    int i = (int)floorf(f);
    f -= (float)i;
    if (i < 0) { i = 0; f = 0.f; }
    if (i >= n_src - 1) { i = n_src - 1; f = 0.f; }
    ResizeTap t;
    t.i0 = i; t.i1 = i + 1 < n_src ? i + 1 : i;
    // saturate_cast<short>(x * 2048): round to nearest even like cvRound
    t.a0 = (int)rintf((1.f - f) * 2048.f);
    t.a1 = (int)rintf(f * 2048.f);
    return t;
}

__global__ __launch_bounds__(256) void augment_kernel(const uint8_t* __restrict__ in, float* __restrict__ out,
                                                      const int32_t* __restrict__ flip, const int32_t* __restrict__ holes,
                                                      const uint8_t* __restrict__ lut, int nholes, int B, int Hin, int Win,
                                                      int S) {
    const float sy = (float)Hin / (float)S, sx = (float)Win / (float)S;
    const size_t total = (size_t)B * S * S;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int x = (int)(idx % S), y = (int)((idx / S) % S), n = (int)(idx / ((size_t)S * S));
        const int xs = (flip && flip[n]) ? S - 1 - x : x;          // flip acts on the resized image
        bool dropped = false;
        for (int hIdx = 0; hIdx < nholes; ++hIdx) {
            const int32_t* hb = holes + ((size_t)n * nholes + hIdx) * 4;      // x1, y1, x2, y2 (exclusive), x2 <= x1: unused slot
            dropped |= (x >= hb[0] && x < hb[2] && y >= hb[1] && y < hb[3]);
        }
        float v[3] = {0.f, 0.f, 0.f};
        if (!dropped) {
            const uint8_t* img = in + (size_t)n * Hin * Win * 3;
            const uint8_t* tb = lut ? lut + (size_t)n * 256 : nullptr;      // RandomGamma table of this image
            auto px = [&](const uint8_t* p) -> int { return tb ? (int)tb[*p] : (int)*p; };
            if (Hin == S && Win == S) {
                const uint8_t* p = img + ((size_t)y * Win + xs) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) v[c] = (float)px(p + c);
            } else {
                const ResizeTap ty = resize_tap(y, Hin, sy), tx = resize_tap(xs, Win, sx);
                const uint8_t* r0 = img + (size_t)ty.i0 * Win * 3;
                const uint8_t* r1 = img + (size_t)ty.i1 * Win * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int s0 = px(r0 + tx.i0 * 3 + c) * tx.a0 + px(r0 + tx.i1 * 3 + c) * tx.a1;
                    const int s1 = px(r1 + tx.i0 * 3 + c) * tx.a0 + px(r1 + tx.i1 * 3 + c) * tx.a1;
                    int d = ((((ty.a0 * (s0 >> 4)) >> 16) + ((ty.a1 * (s1 >> 4)) >> 16) + 2) >> 2);
                    d = d < 0 ? 0 : (d > 255 ? 255 : d);
                    v[c] = (float)d;
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = (v[c] - 127.5f) / 127.5f;     // Normalize(mean 0.5, std 0.5, max_pixel_value 255)
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) out[(((size_t)n * 3 + c) * S + y) * S + x] = v[c];
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// MotionBlur and ISONoise of the reference chain (/root/reference/utils/data_partial.py:139-143: alb.MotionBlur(p),
// alb.ISONoise(p, color_shift, intensity)), on the uint8 HWC batch BEFORE the resize, as the chain has them.
// Both restate published algorithms of packages that are not installed here (albumentations 1.x functional.py, OpenCV
// filter2D / cvtColor): PARITY UNPINNED, like the rest of this file.

// MotionBlur = cv2.filter2D(img, -1, kernel): correlation with a k x k (k = 3, 5, 7) kernel holding a normalised one-pixel line,
// anchor at the centre, BORDER_REFLECT_101, float accumulation over the non-zero taps in row-major order, cvRound + saturate to
// uint8.  The host draws the line (as albumentations does) and passes it zero-padded to 7 x 7 with the centre at (3, 3);
// ksize[n] == 0: image n is copied.  `lut` (RandomGamma table per image, may be null) is applied to the source bytes first.
__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}

__global__ __launch_bounds__(256) void motion_blur_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                          const uint8_t* __restrict__ lut, const float* __restrict__ kernels,
                                                          const int32_t* __restrict__ ksize, int B, int H, int W) {
    const size_t total = (size_t)B * H * W;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int x = (int)(idx % W), y = (int)((idx / W) % H), n = (int)(idx / ((size_t)W * H));
        const uint8_t* img = in + (size_t)n * H * W * 3;
        const uint8_t* tb = lut ? lut + (size_t)n * 256 : nullptr;
        const int k = ksize ? ksize[n] : 0;
        uint8_t* o = out + idx * 3;
        if (k <= 0) {
            const uint8_t* p = img + ((size_t)y * W + x) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) o[c] = tb ? tb[p[c]] : p[c];
            continue;
        }
        const float* kn = kernels + (size_t)n * 49;
        const int a = k >> 1;
        float acc[3] = {0.f, 0.f, 0.f};
        for (int i = -a; i <= a; ++i) {
            const int yy = reflect101(y + i, H);
            for (int j = -a; j <= a; ++j) {
                const float wgt = kn[(i + 3) * 7 + (j + 3)];
                if (wgt == 0.f) continue;
                const uint8_t* p = img + ((size_t)yy * W + reflect101(x + j, W)) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[c] += wgt * (float)(tb ? tb[p[c]] : p[c]);
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float r = rintf(acc[c]);
            o[c] = (uint8_t)(r < 0.f ? 0.f : (r > 255.f ? 255.f : r));
        }
    }
}

// OpenCV's float RGB <-> HLS (H in degrees [0, 360], L and S in [0, 1]), restated from imgproc's RGB2HLS_f / HLS2RGB_f
struct Hls { float h, l, s; };
__device__ __forceinline__ Hls rgb2hls(float r, float g, float b) {
    float vmax = r, vmin = r;
    if (vmax < g) vmax = g;
    if (vmax < b) vmax = b;
    if (vmin > g) vmin = g;
    if (vmin > b) vmin = b;
    float diff = vmax - vmin;
    Hls o; o.l = (vmax + vmin) * 0.5f; o.h = 0.f; o.s = 0.f;
    if (diff > 1.1920929e-07f) {
        o.s = o.l < 0.5f ? diff / (vmax + vmin) : diff / (2.f - vmax - vmin);
        diff = 60.f / diff;
        if (vmax == r) o.h = (g - b) * diff;
        else if (vmax == g) o.h = (b - r) * diff + 120.f;
        else o.h = (r - g) * diff + 240.f;
        if (o.h < 0.f) o.h += 360.f;
    }
    return o;
}
__device__ __forceinline__ void hls2rgb(float h, float l, float s, float& r, float& g, float& b) {
    if (s == 0.f) { r = g = b = l; return; }
    const float p2 = l <= 0.5f ? l * (1.f + s) : l + s - l * s;
    const float p1 = 2.f * l - p2;
    h *= (6.f / 360.f);
    if (h < 0.f) do h += 6.f; while (h < 0.f);
    else if (h >= 6.f) do h -= 6.f; while (h >= 6.f);
    int sector = (int)floorf(h);
    h -= (float)sector;
    if ((unsigned)sector >= 6u) { sector = 0; h = 0.f; }
    float tab[4];
    tab[0] = p2; tab[1] = p1; tab[2] = p1 + (p2 - p1) * (1.f - h); tab[3] = p1 + (p2 - p1) * h;
    const int sd[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
    b = tab[sd[sector][0]]; g = tab[sd[sector][1]]; r = tab[sd[sector][2]];
}

// ISONoise, pass 1: per-image partial sums of the L channel (cv2.meanStdDev accumulates in double): partial[n][blk][2]
constexpr int ISO_BLOCKS = 16;
__global__ __launch_bounds__(256) void iso_stats_kernel(const uint8_t* __restrict__ img, double* __restrict__ partial, int H, int W) {
    const int n = blockIdx.y, blk = blockIdx.x;
    const uint8_t* p = img + (size_t)n * H * W * 3;
    const float inv = (float)(1.0 / 255.0);
    double s1 = 0.0, s2 = 0.0;
    for (int i = blk * 256 + threadIdx.x; i < H * W; i += ISO_BLOCKS * 256) {
        const Hls v = rgb2hls((float)p[3 * i] * inv, (float)p[3 * i + 1] * inv, (float)p[3 * i + 2] * inv);
        s1 += (double)v.l; s2 += (double)v.l * (double)v.l;
    }
    __shared__ double red[2][256];
    red[0][threadIdx.x] = s1; red[1][threadIdx.x] = s2;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) { red[0][threadIdx.x] += red[0][threadIdx.x + st]; red[1][threadIdx.x] += red[1][threadIdx.x + st]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[((size_t)n * ISO_BLOCKS + blk) * 2] = red[0][0]; partial[((size_t)n * ISO_BLOCKS + blk) * 2 + 1] = red[1][0]; }
}

// counter-based generator for the device-drawn noise: Philox-4x32-10 (Salmon et al. 2011), key = per-image seed, counter = pixel
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t k1, uint32_t (&o)[4]) {
    uint32_t c[4] = {c0, c1, 0u, 0u};
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c[0]; o[1] = c[1]; o[2] = c[2]; o[3] = c[3];
}
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.f / 16777216.f); }      // (0, 1)

// ISONoise, pass 2 (albumentations functional.iso_noise): hue += N(0, color_shift * 360 * intensity) (wrapped into [0, 360]),
// L += Poisson(std(L) * intensity * 255) / 255 * (1 - L), back to RGB, * 255, truncated to uint8.
// params[n] = (color_shift, intensity); intensity <= 0: image n is copied.  The draws are either explicit inputs (lum_noise int32 /
// color_noise float32, one per pixel: parity tests) or made here from seeds[n] (production).
__global__ __launch_bounds__(256) void iso_noise_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                        const double* __restrict__ partial, const float* __restrict__ params,
                                                        const int32_t* __restrict__ lum_noise, const float* __restrict__ color_noise,
                                                        const uint64_t* __restrict__ seeds, int B, int H, int W) {
    const size_t total = (size_t)B * H * W;
    const float inv = (float)(1.0 / 255.0);
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int n = (int)(idx / ((size_t)W * H));
        const uint8_t* p = in + idx * 3;
        uint8_t* o = out + idx * 3;
        const float cshift = params[2 * n], intensity = params[2 * n + 1];
        if (!(intensity > 0.f)) { o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; continue; }
        double s1 = 0.0, s2 = 0.0;
        for (int k = 0; k < ISO_BLOCKS; ++k) { s1 += partial[((size_t)n * ISO_BLOCKS + k) * 2]; s2 += partial[((size_t)n * ISO_BLOCKS + k) * 2 + 1]; }
        const double cnt = (double)H * (double)W, mean = s1 / cnt;
        double var = s2 / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const float lam = (float)(sqrt(var) * (double)intensity * 255.0);
        const float sigma = cshift * 360.f * intensity;
        float cn, ln;
        if (lum_noise) { ln = (float)lum_noise[idx]; cn = color_noise[idx]; }
        else {
            uint32_t r[4];
            const uint64_t pix = idx - (size_t)n * W * H;
            philox4x32((uint32_t)pix, (uint32_t)(pix >> 32), (uint32_t)seeds[n], (uint32_t)(seeds[n] >> 32), r);
            const float ra = sqrtf(-2.f * logf(u01(r[0]))), th = 6.2831853f * u01(r[1]);
            cn = sigma * ra * cosf(th);
            if (lam >= 10.f) {                         // normal approximation with continuity correction
                const float z = sqrtf(-2.f * logf(u01(r[2]))) * cosf(6.2831853f * u01(r[3]));
                ln = floorf(lam + sqrtf(lam) * z + 0.5f);
                if (ln < 0.f) ln = 0.f;
            } else {                                   // inversion by sequential search
                const float u = u01(r[2]);
                float pk = expf(-lam), cdf = pk;
                int k = 0;
                while (u > cdf && k < 64) { ++k; pk *= lam / (float)k; cdf += pk; }
                ln = (float)k;
            }
        }
        Hls v = rgb2hls((float)p[0] * inv, (float)p[1] * inv, (float)p[2] * inv);
        v.h += cn;
        if (v.h < 0.f) v.h += 360.f;
        if (v.h > 360.f) v.h -= 360.f;
        v.l = v.l + (ln / 255.f) * (1.f - v.l);
        float r_, g_, b_;
        hls2rgb(v.h, v.l, v.s, r_, g_, b_);
        const float rgb[3] = {r_ * 255.f, g_ * 255.f, b_ * 255.f};
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = (uint8_t)(rgb[c] < 0.f ? 0.f : (rgb[c] > 255.f ? 255.f : rgb[c]));      // astype(uint8): truncation
    }
}

// Dropout mask of the backbones' tail (reference nets/SwinV2.py:559, nets/AlterNet_SwinV2_FAN.py:743: nn.Dropout() before fc):
// mask[i] = 1 / keep with probability keep, else 0, four elements per Philox call (key = seed, counter = i / 4).  One launch where
// torch.rand -> compare -> cast -> divide were four passes over the [B, 25 088] tensor.
template <typename T>
__global__ __launch_bounds__(256) void dropout_mask_kernel(T* __restrict__ mask, size_t n, float keep, float inv_keep, uint64_t seed) {
    const size_t n4 = (n + 3) / 4;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += (size_t)gridDim.x * 256) {
        uint32_t r[4];
        philox4x32((uint32_t)q, (uint32_t)(q >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), r);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * q + e < n) mask[4 * q + e] = (T)(u01(r[e]) < keep ? inv_keep : 0.f);
    }
}

}  // namespace frhip

using namespace frhip;

extern "C" int frhip_dropout_mask(int dtype, void* mask, size_t n, float keep, long long seed, hipStream_t stream) {
    if (!(keep > 0.f) || keep > 1.f) { set_error("frhip_dropout_mask: keep probability %g outside (0, 1]", (double)keep); return FRHIP_EINVAL; }
    if (n == 0) return FRHIP_OK;
    size_t blocks = ((n + 3) / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (dtype == FRHIP_DT_BF16)
        hipLaunchKernelGGL(dropout_mask_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, stream, (bf16_t*)mask, n, keep, 1.f / keep, (uint64_t)seed);
    else if (dtype == FRHIP_DT_F32)
        hipLaunchKernelGGL(dropout_mask_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, stream, (float*)mask, n, keep, 1.f / keep, (uint64_t)seed);
    else { set_error("frhip_dropout_mask: bad dtype %d", dtype); return FRHIP_EINVAL; }
    return check_launch("frhip_dropout_mask");
}

static int augment_run(const uint8_t* in, float* out, const int32_t* flip, const int32_t* holes, const uint8_t* lut, int nholes,
                       int b, int hin, int win, int size, hipStream_t stream) {
    if (!in || !out || b <= 0 || hin <= 0 || win <= 0 || size <= 0 || nholes < 0 || (nholes > 0 && !holes)) {
        set_error("frhip_augment_u8: bad arguments (b=%d hin=%d win=%d size=%d nholes=%d)", b, hin, win, size, nholes);
        return FRHIP_EINVAL;
    }
    const size_t total = (size_t)b * size * size;
    size_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(augment_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, in, out, flip, holes, lut, nholes, b, hin, win, size);
    return check_launch("frhip_augment_u8");
}

extern "C" int frhip_augment_u8(const uint8_t* in, float* out, const int32_t* flip, const int32_t* holes, int nholes,
                                int b, int hin, int win, int size, hipStream_t stream) {
    return augment_run(in, out, flip, holes, nullptr, nholes, b, hin, win, size, stream);
}

extern "C" int frhip_augment_u8_lut(const uint8_t* in, const uint8_t* lut, float* out, const int32_t* flip, const int32_t* holes,
                                    int nholes, int b, int hin, int win, int size, hipStream_t stream) {
    return augment_run(in, out, flip, holes, lut, nholes, b, hin, win, size, stream);
}

extern "C" int frhip_motion_blur_u8(const uint8_t* in, uint8_t* out, const uint8_t* lut, const float* kernels, const int32_t* ksize,
                                    int b, int h, int w, hipStream_t stream) {
    if (!in || !out || in == out || b <= 0 || h <= 0 || w <= 0 || (ksize && !kernels)) {
        set_error("frhip_motion_blur_u8: bad arguments (b=%d h=%d w=%d; out must not alias in)", b, h, w);
        return FRHIP_EINVAL;
    }
    size_t blocks = ((size_t)b * h * w + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(motion_blur_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, in, out, lut, kernels, ksize, b, h, w);
    return check_launch("frhip_motion_blur_u8");
}

extern "C" int frhip_iso_noise_scratch_doubles(int b) { return b * ISO_BLOCKS * 2; }

extern "C" int frhip_iso_noise_u8(const uint8_t* in, uint8_t* out, double* scratch, const float* params, const int32_t* lum_noise,
                                  const float* color_noise, const uint64_t* seeds, int b, int h, int w, hipStream_t stream) {
    if (!in || !out || !scratch || !params || b <= 0 || h <= 0 || w <= 0 || ((lum_noise == nullptr) != (color_noise == nullptr)) ||
        (!lum_noise && !seeds)) {
        set_error("frhip_iso_noise_u8: bad arguments (b=%d h=%d w=%d; explicit noise needs both arrays, device noise needs seeds)", b, h, w);
        return FRHIP_EINVAL;
    }
    hipLaunchKernelGGL(iso_stats_kernel, dim3(ISO_BLOCKS, b), dim3(256), 0, stream, in, scratch, h, w);
    int rc = check_launch("frhip_iso_noise_u8(stats)");
    if (rc) return rc;
    size_t blocks = ((size_t)b * h * w + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(iso_noise_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, in, out, scratch, params, lum_noise, color_noise,
                       seeds, b, h, w);
    return check_launch("frhip_iso_noise_u8");
}
