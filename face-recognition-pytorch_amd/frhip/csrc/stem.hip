// Stem of the backbone on gfx950: conv3x3(3->64, s1) is lowered to im2col (K = 27 padded to one 128-byte K step)
// + the shared NT GEMM; BN-apply + ReLU + MaxPool(3,2,1) are fused into one HBM pass that also records the argmax
// tap so the backward pass is a gather.  Reference: /root/reference/nets/resnet.py:186-189, :232-235.
#include "common.h"
#include "frhip.h"

namespace frhip {

// x: NCHW fp32 [B,3,H,W]  ->  col: [B*Ho*Wo][KP] of T (3x3, pad 1, stride 1 or 2), k = (r*3+s)*3 + ci for k < 27,
// zero for k >= 27.
template <typename T>
__global__ __launch_bounds__(256) void stem_im2col_kernel(const float* __restrict__ x, T* __restrict__ col,
                                                          int B, int H, int W, int stride) {
    constexpr int EPV = 16 / (int)sizeof(T);
    constexpr int VPP = 8;                                // vectors per pixel (KP = 8*EPV: 64 bf16 / 32 f32)
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const size_t total = (size_t)B * Ho * Wo * VPP;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
        const int v = (int)(t % VPP);
        const size_t pix = t / VPP;
        const int w = (int)(pix % Wo) * stride, h = (int)((pix / Wo) % Ho) * stride, n = (int)(pix / ((size_t)Wo * Ho));
        Vec16<T> o;
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
            const int k = v * EPV + e;
            float val = 0.f;
            if (k < 27) {
                const int tap = k / 3, ci = k - tap * 3, r = tap / 3, s = tap - r * 3;
                const int hi = h + r - 1, wi = w + s - 1;
                if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W)
                    val = x[(((size_t)n * 3 + ci) * H + hi) * W + wi];
            }
            o.set(e, val);
        }
        *reinterpret_cast<Vec16<T>*>(col + t * EPV) = o;
    }
}

// a = relu(y*scale+shift); out = maxpool3x3/s2/p1(a); arg = first tap (row-major) reaching the max.
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_maxpool_fwd_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, T* __restrict__ out,
                                                                  uint8_t* __restrict__ arg, int B, int H, int W, int C) {
    constexpr int EPV = 16 / (int)sizeof(T);
    const int Hp = (H + 2 - 3) / 2 + 1, Wp = (W + 2 - 3) / 2 + 1, vpr = C / EPV;
    const size_t total = (size_t)B * Hp * Wp * vpr;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
        const int cv = (int)(t % vpr);
        const size_t pp = t / vpr;
        const int pw = (int)(pp % Wp), ph = (int)((pp / Wp) % Hp), n = (int)(pp / ((size_t)Wp * Hp));
        float best[EPV]; int bi[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) { best[e] = -INFINITY; bi[e] = 0; }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int hi = 2 * ph - 1 + r, wi = 2 * pw - 1 + s;
                if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W) {
                    const Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(y + (((size_t)n * H + hi) * W + wi) * C + cv * EPV);
#pragma unroll
                    for (int e = 0; e < EPV; ++e) {
                        float a = v.get(e) * scale[cv * EPV + e] + shift[cv * EPV + e];
                        a = a > 0.f ? a : 0.f;
                        if (a > best[e]) { best[e] = a; bi[e] = r * 3 + s; }
                    }
                }
            }
        Vec16<T> o;
#pragma unroll
        for (int e = 0; e < EPV; ++e) { o.set(e, best[e]); arg[t * EPV + e] = (uint8_t)bi[e]; }
        *reinterpret_cast<Vec16<T>*>(out + t * EPV) = o;
    }
}

// da[n,h,w,c] = sum over pooling windows that contain (h,w) and whose argmax is (h,w) of dpool[window]
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dpool, const uint8_t* __restrict__ arg,
                                                          T* __restrict__ da, int B, int H, int W, int C) {
    constexpr int EPV = 16 / (int)sizeof(T);
    const int Hp = (H + 2 - 3) / 2 + 1, Wp = (W + 2 - 3) / 2 + 1, vpr = C / EPV;
    const size_t total = (size_t)B * H * W * vpr;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
        const int cv = (int)(t % vpr);
        const size_t pix = t / vpr;
        const int w = (int)(pix % W), h = (int)((pix / W) % H), n = (int)(pix / ((size_t)W * H));
        float acc[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) acc[e] = 0.f;
        // window ph covers rows 2ph-1 .. 2ph+1
        const int ph_lo = h >> 1, ph_hi = (h + 1) >> 1, pw_lo = w >> 1, pw_hi = (w + 1) >> 1;
        for (int ph = ph_lo; ph <= ph_hi; ++ph) {
            if (ph >= Hp) continue;
            const int r = h - (2 * ph - 1);
            for (int pw = pw_lo; pw <= pw_hi; ++pw) {
                if (pw >= Wp) continue;
                const int s = w - (2 * pw - 1);
                const int tap = r * 3 + s;
                const size_t o = ((((size_t)n * Hp + ph) * Wp + pw) * vpr + cv) * EPV;
                const Vec16<T> d = *reinterpret_cast<const Vec16<T>*>(dpool + o);
#pragma unroll
                for (int e = 0; e < EPV; ++e)
                    if (arg[o + e] == tap) acc[e] += d.get(e);
            }
        }
        Vec16<T> ov;
#pragma unroll
        for (int e = 0; e < EPV; ++e) ov.set(e, acc[e]);
        *reinterpret_cast<Vec16<T>*>(da + t * EPV) = ov;
    }
}

static int stem_grid(size_t total) {
    size_t b = (total + 255) / 256;
    if (b > 8192) b = 8192;
    return (int)(b < 1 ? 1 : b);
}

}  // namespace frhip

using namespace frhip;

extern "C" int frhip_stem_im2col(int dtype, const float* x, void* col, int b, int h, int w, int stride, hipStream_t stream) {
    if (stride != 1 && stride != 2) { set_error("frhip_stem_im2col: stride must be 1 or 2"); return FRHIP_EINVAL; }
    const size_t total = (size_t)b * ((h - 1) / stride + 1) * ((w - 1) / stride + 1) * 8;
    if (dtype == FRHIP_DT_BF16)
        hipLaunchKernelGGL(stem_im2col_kernel<bf16_t>, dim3(stem_grid(total)), dim3(256), 0, stream, x, (bf16_t*)col, b, h, w, stride);
    else if (dtype == FRHIP_DT_F32)
        hipLaunchKernelGGL(stem_im2col_kernel<float>, dim3(stem_grid(total)), dim3(256), 0, stream, x, (float*)col, b, h, w, stride);
    else { set_error("frhip_stem_im2col: bad dtype %d", dtype); return FRHIP_EINVAL; }
    return check_launch("frhip_stem_im2col");
}

extern "C" int frhip_bn_relu_maxpool_fwd(int dtype, const void* y, const float* scale, const float* shift, void* out,
                                         uint8_t* argmax, int b, int h, int w, int c, hipStream_t stream) {
    const int epv = dtype == FRHIP_DT_BF16 ? 8 : 4;
    if ((dtype != FRHIP_DT_BF16 && dtype != FRHIP_DT_F32) || (c % epv)) { set_error("frhip_bn_relu_maxpool_fwd: bad dtype/channels"); return FRHIP_EINVAL; }
    const int hp = (h - 1) / 2 + 1, wp = (w - 1) / 2 + 1;
    const size_t total = (size_t)b * hp * wp * (c / epv);
    if (dtype == FRHIP_DT_BF16)
        hipLaunchKernelGGL(bn_relu_maxpool_fwd_kernel<bf16_t>, dim3(stem_grid(total)), dim3(256), 0, stream,
                           (const bf16_t*)y, scale, shift, (bf16_t*)out, argmax, b, h, w, c);
    else
        hipLaunchKernelGGL(bn_relu_maxpool_fwd_kernel<float>, dim3(stem_grid(total)), dim3(256), 0, stream,
                           (const float*)y, scale, shift, (float*)out, argmax, b, h, w, c);
    return check_launch("frhip_bn_relu_maxpool_fwd");
}

extern "C" int frhip_maxpool_bwd(int dtype, const void* dpool, const uint8_t* argmax, void* da, int b, int h, int w,
                                 int c, hipStream_t stream) {
    const int epv = dtype == FRHIP_DT_BF16 ? 8 : 4;
    if ((dtype != FRHIP_DT_BF16 && dtype != FRHIP_DT_F32) || (c % epv)) { set_error("frhip_maxpool_bwd: bad dtype/channels"); return FRHIP_EINVAL; }
    const size_t total = (size_t)b * h * w * (c / epv);
    if (dtype == FRHIP_DT_BF16)
        hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, dim3(stem_grid(total)), dim3(256), 0, stream,
                           (const bf16_t*)dpool, argmax, (bf16_t*)da, b, h, w, c);
    else
        hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(stem_grid(total)), dim3(256), 0, stream,
                           (const float*)dpool, argmax, (float*)da, b, h, w, c);
    return check_launch("frhip_maxpool_bwd");
}
