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

}  // namespace frhip

using namespace frhip;

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
