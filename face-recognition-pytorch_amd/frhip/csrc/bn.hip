// BatchNorm (training + eval) and the element-wise glue of the NHWC backbone, for gfx950.
// HBM-bound kernels: 16 bytes per lane, channels fastest (NHWC), per-channel parameters in registers.
// Reference semantics: nn.BatchNorm2d / nn.BatchNorm1d defaults (eps 1e-5, momentum 0.1, biased batch variance
// for normalisation, unbiased for running_var) as used by /root/reference/nets/resnet.py:81-86, :187, :196-199,
// and the residual add of BasicBlock.forward (:89-103).
#include <cstdlib>
#include "common.h"
#include "frhip.h"

namespace frhip {

constexpr int EW_THREADS = 256;
#ifndef EW_UNROLL
#define EW_UNROLL 2
#endif
#ifndef EW_ROWS
#define EW_ROWS 16
#endif
constexpr int RED_GROUPS = 16;      // second-level partial count

template <typename T> struct EW {
    static constexpr int EPV = 16 / (int)sizeof(T);
};

// ------------------------------------------------------------------------------------------------
// Column statistics of a [rows][C] tensor: partial[blk][0][c] = sum x, partial[blk][1][c] = sum x^2.
// Optional second operand turns it into the BN-backward reduction:
//   d_eff = dout * (relu ? (y*scale+shift > 0) : 1);  partial = { sum d_eff, sum d_eff * (y-mean)*invstd }.
// RS: every group of `rows_per` consecutive rows (a sample) carries a factor rowscale[group] on the gradient (stochastic depth: the
// branch was scaled by it in the forward pass, nets/AlterNet_SwinV2_FAN.py DropPath) -- d = dout * rowscale[r / rows_per].
template <typename T, bool BWD, bool RS = false>
__global__ __launch_bounds__(EW_THREADS) void colreduce_kernel(const T* __restrict__ x, const T* __restrict__ y,
                                                               const float* __restrict__ mean, const float* __restrict__ invstd,
                                                               const float* __restrict__ mscale, const float* __restrict__ mshift,
                                                               float* __restrict__ partial, int rows, int C,
                                                               const float* __restrict__ rowscale = nullptr, int rows_per = 1) {
    constexpr int EPV = EW<T>::EPV;
    const int vpr = C / EPV;                     // vectors per row (<= 256, divides 256)
    const int cg = threadIdx.x % vpr, rl = threadIdx.x / vpr, rlanes = EW_THREADS / vpr;
    float s1[EPV], s2[EPV], mu[EPV], is[EPV], ms[EPV], mb[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
        s1[e] = 0.f; s2[e] = 0.f;
        if (BWD) {
            mu[e] = mean[cg * EPV + e]; is[e] = invstd[cg * EPV + e];
            ms[e] = mscale ? mscale[cg * EPV + e] : 0.f; mb[e] = mscale ? mshift[cg * EPV + e] : 1.f;
        }
    }
    for (int r = blockIdx.x * rlanes + rl; r < rows; r += gridDim.x * rlanes) {
        const size_t idx = (size_t)r * C + cg * EPV;
        const Vec16<T> a = *reinterpret_cast<const Vec16<T>*>(x + idx);
        if (BWD) {
            const Vec16<T> b = *reinterpret_cast<const Vec16<T>*>(y + idx);
            float ks = 1.f;
            if constexpr (RS) ks = rowscale[r / rows_per];
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
                const float yy = b.get(e);
                const float d = (yy * ms[e] + mb[e] > 0.f) ? a.get(e) * ks : 0.f;
                s1[e] += d; s2[e] += d * (yy - mu[e]) * is[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < EPV; ++e) { const float v = a.get(e); s1[e] += v; s2[e] += v * v; }
        }
    }
    __shared__ float red[2][EW_THREADS][EPV + 1];
#pragma unroll
    for (int e = 0; e < EPV; ++e) { red[0][threadIdx.x][e] = s1[e]; red[1][threadIdx.x][e] = s2[e]; }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * C; t += EW_THREADS) {
        const int st = t / C, c = t - st * C;
        float acc = 0.f;
        for (int k = 0; k < rlanes; ++k) acc += red[st][k * vpr + c / EPV][c % EPV];
        partial[((size_t)blockIdx.x * 2 + st) * C + c] = acc;
    }
}

// partial[nparts][2][C] -> out[gridDim.y][2][C]   (block = 64 columns x 4 partial-lanes).  A lane's rows are ALL requested before the first
// add (up to RP_MAX independent loads in flight): the kernel is a chain of L2 / HBM round trips, not bandwidth -- with one load in flight per
// trip the 6 272-row fold of a 56 x 56 layer took 8 - 10 us on the conv -> BatchNorm critical path of every such layer (round 4: 256 groups,
// <= 16 loads per lane and trip).
constexpr int RP_MAX = 16;
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ in, float* __restrict__ out, int nparts, int C) {
    __shared__ float red[4][64];
    const int tl = threadIdx.x & 63, pl = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + tl;                        // over 2*C
    float acc = 0.f;
    if (t < 2 * C) {
        const int stride = gridDim.y * 4;
        for (int p0 = blockIdx.y * 4 + pl; p0 < nparts; p0 += stride * RP_MAX) {
            float v[RP_MAX];
#pragma unroll
            for (int u = 0; u < RP_MAX; ++u) { const int p = p0 + u * stride; v[u] = p < nparts ? in[(size_t)p * 2 * C + t] : 0.f; }
#pragma unroll
            for (int u = 0; u < RP_MAX; ++u) acc += v[u];      // fixed order: deterministic
        }
    }
    red[pl][tl] = acc;
    __syncthreads();
    if (pl == 0 && t < 2 * C) out[(size_t)blockIdx.y * 2 * C + t] = red[0][tl] + red[1][tl] + red[2][tl] + red[3][tl];
}

// sum of partial[p][st][c] over p, by PL = blockDim.x / 64 lanes per channel (block = 64 channels x PL lanes, PL <= 16);
// result valid on lane 0
constexpr int FIN_MAX_PL = 16;
__device__ __forceinline__ void sum_parts(const float* __restrict__ parts, int nparts, int C, int c, int pl,
                                          float (*red)[2][64], float& s1, float& s2) {
    const int PL = blockDim.x >> 6;
    float a1 = 0.f, a2 = 0.f;
    if (c < C) {
        int p = pl;
        for (; p + 7 * PL < nparts; p += 8 * PL) {        // eight independent row pairs in flight (sixteen loads per round trip)
            float x[8], y[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const float* r = parts + (size_t)(p + u * PL) * 2 * C; x[u] = r[c]; y[u] = r[C + c]; }
            a1 += ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
            a2 += ((y[0] + y[1]) + (y[2] + y[3])) + ((y[4] + y[5]) + (y[6] + y[7]));
        }
        for (; p + 3 * PL < nparts; p += 4 * PL) {
            const float* r0 = parts + (size_t)p * 2 * C, *r1 = r0 + (size_t)PL * 2 * C, *r2 = r1 + (size_t)PL * 2 * C,
                        *r3 = r2 + (size_t)PL * 2 * C;
            const float x0 = r0[c], y0 = r0[C + c], x1 = r1[c], y1 = r1[C + c], x2 = r2[c], y2 = r2[C + c], x3 = r3[c], y3 = r3[C + c];
            a1 += (x0 + x1) + (x2 + x3); a2 += (y0 + y1) + (y2 + y3);
        }
        for (; p < nparts; p += PL) { a1 += parts[(size_t)p * 2 * C + c]; a2 += parts[(size_t)p * 2 * C + C + c]; }
    }
    red[pl][0][threadIdx.x & 63] = a1; red[pl][1][threadIdx.x & 63] = a2;
    __syncthreads();
    const int cl = threadIdx.x & 63;
    s1 = 0.f; s2 = 0.f;
    for (int l = 0; l < PL; ++l) { s1 += red[l][0][cl]; s2 += red[l][1][cl]; }
}

// The same fold by a 256-thread block of 16 channels x 16 lanes (grid = C / 16).  The 1024-thread blocks above need sixteen free wave
// slots and their registers on ONE CU: beside the persistent linear / weight-gradient workgroups of the Swin and AlterNet steps a
// finalize launch then waits for a whole workgroup of the other stream to retire (timeline, Swin34: bn_bwd_finalize 27.6 us per launch
// inside the step against 5.1 us alone, 0.5 ms of the step on its critical path).  A 4-wave block fits the leftover slots at once.
__device__ __forceinline__ void sum_parts16(const float* __restrict__ parts, int nparts, int C, int c, int pl,
                                            float (*red)[2][16], float& s1, float& s2) {
    float a1 = 0.f, a2 = 0.f;
    if (c < C) {
        int p = pl;
        for (; p + 7 * 16 < nparts; p += 8 * 16) {
            float x[8], y[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const float* r = parts + (size_t)(p + u * 16) * 2 * C; x[u] = r[c]; y[u] = r[C + c]; }
            a1 += ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
            a2 += ((y[0] + y[1]) + (y[2] + y[3])) + ((y[4] + y[5]) + (y[6] + y[7]));
        }
        for (; p < nparts; p += 16) { a1 += parts[(size_t)p * 2 * C + c]; a2 += parts[(size_t)p * 2 * C + C + c]; }
    }
    red[pl][0][threadIdx.x & 15] = a1; red[pl][1][threadIdx.x & 15] = a2;
    __syncthreads();
    const int cl = threadIdx.x & 15;
    s1 = 0.f; s2 = 0.f;
#pragma unroll
    for (int l = 0; l < 16; ++l) { s1 += red[l][0][cl]; s2 += red[l][1][cl]; }
}

// Finalise forward batch statistics.
template <bool SMALL>
__global__ __launch_bounds__(SMALL ? 256 : 1024) void bn_finalize_kernel(const float* __restrict__ parts, int nparts, int C, float count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float momentum, float eps, float* __restrict__ mean_out,
                                   float* __restrict__ invstd_out, float* __restrict__ scale, float* __restrict__ shift) {
    __shared__ float red[FIN_MAX_PL][2][64];
    int c, pl;
    float s1, s2;
    if constexpr (SMALL) {
        c = blockIdx.x * 16 + (threadIdx.x & 15); pl = threadIdx.x >> 4;
        sum_parts16(parts, nparts, C, c, pl, reinterpret_cast<float (*)[2][16]>(&red[0][0][0]), s1, s2);
    } else {
        c = blockIdx.x * 64 + (threadIdx.x & 63); pl = threadIdx.x >> 6;
        sum_parts(parts, nparts, C, c, pl, red, s1, s2);
    }
    if (pl != 0 || c >= C) return;
    const float mu = s1 / count;
    float var = s2 / count - mu * mu;
    var = var < 0.f ? 0.f : var;
    const float is = rsqrtf(var + eps);
    mean_out[c] = mu; invstd_out[c] = is;
    const float sc = gamma[c] * is;
    scale[c] = sc; shift[c] = beta[c] - mu * sc;
    if (running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
        const float unbiased = count > 1.f ? var * count / (count - 1.f) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
}

// Eval-mode scale/shift from running statistics.
__global__ void bn_eval_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                               float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] * rsqrtf(rv[c] + eps);
    scale[c] = sc; shift[c] = beta[c] - rm[c] * sc;
}

// Finalise backward sums: dgamma, dbeta and the per-channel affine of  dy = ca*d_eff + cb*y + cc.
template <bool SMALL>
__global__ __launch_bounds__(SMALL ? 256 : 1024) void bn_bwd_finalize_kernel(const float* __restrict__ parts, int nparts, int C, float count,
                                       const float* __restrict__ gamma, const float* __restrict__ mean,
                                       const float* __restrict__ invstd, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, float* __restrict__ ca, float* __restrict__ cb,
                                       float* __restrict__ cc) {
    __shared__ float red[FIN_MAX_PL][2][64];
    int c, pl;
    float s1, s2;
    if constexpr (SMALL) {
        c = blockIdx.x * 16 + (threadIdx.x & 15); pl = threadIdx.x >> 4;
        sum_parts16(parts, nparts, C, c, pl, reinterpret_cast<float (*)[2][16]>(&red[0][0][0]), s1, s2);
    } else {
        c = blockIdx.x * 64 + (threadIdx.x & 63); pl = threadIdx.x >> 6;
        sum_parts(parts, nparts, C, c, pl, red, s1, s2);
    }
    if (pl != 0 || c >= C) return;
    dgamma[c] += s2; dbeta[c] += s1;                 // accumulate into (caller-zeroed) .grad
    const float gi = gamma[c] * invstd[c], m2 = s2 / count, m1 = s1 / count;
    ca[c] = gi; cb[c] = -gi * invstd[c] * m2; cc[c] = gi * (mean[c] * invstd[c] * m2 - m1);
}

// out = act( y*scale + shift  [+ res  |  + res*rscale + rshift] )
// Each thread owns one 16-byte channel group for its whole life, so the per-channel vectors are read once into
// registers and the row loop is pure streaming.
// Streaming accesses of the element-wise passes.  Every operand of bn_apply / bn_bwd_apply is read exactly once, and the
// producer wrote it a whole conv earlier: non-temporal loads (no allocation in L2 / the last-level cache on the way in)
// are worth 0.45 ms of the 29.5-ms ResNet50 step (tools/nt_ab.sh, two passes: 30.54 / 30.67 -> 30.15 / 30.15 ms on one box,
// 29.46 / 29.52 -> 29.06 / 29.10 on another).  A micro-benchmark that re-reads ONE buffer says the opposite for tensors
// below ~100 MB (they sit in the 256-MB last-level cache between its iterations) -- the step is the judge.  Non-temporal
// STORES measured neutral to slightly negative in the step (the consumer conv wants those lines).
// g_ew_nt bits: 1 = non-temporal loads (residual / backward passes), 4 = also in the plain apply, 2 = non-temporal stores;
// FRHIP_EW_NT_MB = smallest tensor (MB) that gets them.
template <typename T, bool NT> __device__ __forceinline__ Vec16<T> ew_load(const T* p) {
    Vec16<T> r;
    if constexpr (NT) r.v = __builtin_nontemporal_load(reinterpret_cast<const decltype(r.v)*>(p));
    else r = *reinterpret_cast<const Vec16<T>*>(p);
    return r;
}
template <typename T, bool NT> __device__ __forceinline__ void ew_store(T* p, const Vec16<T>& x) {
    if constexpr (NT) __builtin_nontemporal_store(x.v, reinterpret_cast<decltype(x.v)*>(p));
    else *reinterpret_cast<Vec16<T>*>(p) = x;
}
static size_t EW_NT_BYTES = (size_t)(getenv("FRHIP_EW_NT_MB") ? atoi(getenv("FRHIP_EW_NT_MB")) : 0) << 20;
// bit 0: non-temporal loads in the residual / backward-apply passes, bit 2: in the plain BatchNorm-apply pass too, bit 1: non-temporal
// stores.  Default 1.  (Round 1 had 5; with the round-2 tile choices the plain apply pass is better off with cached loads of the
// tensor the convolution has just written: 26.34 vs 26.40 ms over four same-box A/B rounds, 26.67 vs 26.73 over two more.)
static int g_ew_nt = getenv("FRHIP_EW_NT") ? atoi(getenv("FRHIP_EW_NT")) : 1;

// INPLACE: out IS y (frhip_conv_fwd_affine's unfused route normalises the convolution's output where it lies).  Two __restrict__ pointers
// to one buffer would be undefined behaviour -- the compiler may then move a later row's load across an earlier row's store -- so this
// instantiation reads through `out` itself.
template <typename T, bool NTL, bool NTS, bool RS = false, bool INPLACE = false>
__global__ __launch_bounds__(EW_THREADS) void bn_apply_kernel(const T* __restrict__ y_in, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, const T* __restrict__ res,
                                                              const float* __restrict__ rscale, const float* __restrict__ rshift,
                                                              int relu, T* __restrict__ out, int rows, int C,
                                                              const float* __restrict__ rowscale = nullptr, int rows_per = 1) {
    constexpr int EPV = EW<T>::EPV;
    const int vpr = C / EPV;
    const int cg = threadIdx.x % vpr, rl = threadIdx.x / vpr, rlanes = EW_THREADS / vpr;
    float sc[EPV], sh[EPV], rs[EPV], rb[EPV];
    const T* y = INPLACE ? out : y_in;
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
        sc[e] = scale[cg * EPV + e]; sh[e] = shift[cg * EPV + e];
        rs[e] = rscale ? rscale[cg * EPV + e] : 1.f; rb[e] = rscale ? rshift[cg * EPV + e] : 0.f;
    }
#pragma unroll EW_UNROLL
    for (int r = blockIdx.x * rlanes + rl; r < rows; r += gridDim.x * rlanes) {
        const size_t idx = (size_t)r * C + cg * EPV;
        Vec16<T> a = ew_load<T, NTL>(y + idx);
        if constexpr (RS) {           // out = res + rowscale[sample] * (y * scale + shift): the normalised branch under stochastic depth
            const float ks = rowscale[r / rows_per];
            Vec16<T> rv;
            if (res) rv = ew_load<T, NTL>(res + idx);
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
                float o = (a.get(e) * sc[e] + sh[e]) * ks + (res ? rv.get(e) * rs[e] + rb[e] : 0.f);
                a.set(e, relu ? fmaxf(o, 0.f) : o);
            }
        } else if (res) {
            const Vec16<T> rv = ew_load<T, NTL>(res + idx);
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
                float o = a.get(e) * sc[e] + sh[e] + (rv.get(e) * rs[e] + rb[e]);
                a.set(e, relu ? fmaxf(o, 0.f) : o);
            }
        } else {
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
                float o = a.get(e) * sc[e] + sh[e];
                a.set(e, relu ? fmaxf(o, 0.f) : o);
            }
        }
        ew_store<T, NTS>(out + idx, a);
    }
}

// dy = ca * d_eff + cb * y + cc,  d_eff = dout * mask
template <typename T, bool NTL, bool NTS, bool RS = false>
__global__ __launch_bounds__(EW_THREADS) void bn_bwd_apply_kernel(const T* __restrict__ dout, const T* __restrict__ y,
                                                                  const float* __restrict__ ca, const float* __restrict__ cb,
                                                                  const float* __restrict__ cc, const float* __restrict__ mscale,
                                                                  const float* __restrict__ mshift, T* __restrict__ dy,
                                                                  int rows, int C, const float* __restrict__ rowscale = nullptr,
                                                                  int rows_per = 1) {
    constexpr int EPV = EW<T>::EPV;
    const int vpr = C / EPV;
    const int cg = threadIdx.x % vpr, rl = threadIdx.x / vpr, rlanes = EW_THREADS / vpr;
    float a_[EPV], b_[EPV], c_[EPV], ms[EPV], mb[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
        a_[e] = ca[cg * EPV + e]; b_[e] = cb[cg * EPV + e]; c_[e] = cc[cg * EPV + e];
        ms[e] = mscale ? mscale[cg * EPV + e] : 0.f; mb[e] = mscale ? mshift[cg * EPV + e] : 1.f;
    }
#pragma unroll EW_UNROLL
    for (int r = blockIdx.x * rlanes + rl; r < rows; r += gridDim.x * rlanes) {
        const size_t idx = (size_t)r * C + cg * EPV;
        const Vec16<T> d = ew_load<T, NTL>(dout + idx);
        Vec16<T> b = ew_load<T, NTL>(y + idx);
        float ks = 1.f;
        if constexpr (RS) ks = rowscale[r / rows_per];
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
            const float yy = b.get(e);
            const float de = (yy * ms[e] + mb[e] > 0.f) ? d.get(e) * ks : 0.f;
            b.set(e, a_[e] * de + b_[e] * yy + c_[e]);
        }
        ew_store<T, NTS>(dy + idx, b);
    }
}

// out[c] += sum_p partial[p][which][c]      (block = 64 channels x 16 lanes; was one thread per channel walking all rows:
// 199 us per call on the Swin bias gradients with ~1000 partial rows)
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partial, int nparts, int C, int which,
                                                           float* __restrict__ out) {
    // 16 channels x 16 lanes per block (256 threads: fits beside the other stream's persistent workgroups, see bn_finalize_kernel<true>)
    __shared__ float red[16][16];
    const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float acc = 0.f;
    if (c < C) {
        int p = pl;
        for (; p + 7 * 16 < nparts; p += 8 * 16) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[((size_t)(p + u * 16) * 2 + which) * C + c];
            acc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        for (; p < nparts; p += 16) acc += partial[((size_t)p * 2 + which) * C + c];
    }
    red[pl][cl] = acc;
    __syncthreads();
    if (pl == 0 && c < C) {
        float a = 0.f;
#pragma unroll
        for (int l = 0; l < 16; ++l) a += red[l][cl];
        out[c] += a;
    }
}

// x[rows][C] += bias[C]  (fp32 tail of the backbone: fc bias)
__global__ void add_bias_kernel(float* __restrict__ x, const float* __restrict__ bias, size_t n, int C) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        x[i] += bias[i % C];
}

template <typename T>
__global__ void cast_from_f32_kernel(const float* __restrict__ src, T* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = from_f32<T>(src[i]);
}
template <typename T>
__global__ void cast_to_f32_kernel(const T* __restrict__ src, float* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = to_f32<T>(src[i]);
}

static int grid_for(size_t work_items, int per_block) {
    size_t b = (work_items + per_block - 1) / per_block;
    if (b > 2048) b = 2048;            // 256 CUs x 8 blocks, grid-stride the rest
    if (b < 1) b = 1;
    return (int)b;
}

// blocks for the row-streaming element-wise kernels: 256 CUs x 8 resident blocks, each thread >= 1 row
static int ew_row_blocks(int rows, int c, int dtype) {
    const int epv = dtype == FRHIP_DT_BF16 ? 8 : 4;
    const int rlanes = EW_THREADS / (c / epv);
    // >= 8 rows per thread so the per-channel vectors (up to 40 floats per thread) are amortised
    int b = (rows + rlanes * EW_ROWS - 1) / (rlanes * EW_ROWS);
    if (b > 2048) b = 2048;
    return b < 1 ? 1 : b;
}

static bool shape_ok(int dtype, int C, const char* who) {
    const int epv = dtype == FRHIP_DT_BF16 ? 8 : 4;
    if ((dtype != FRHIP_DT_BF16 && dtype != FRHIP_DT_F32) || C <= 0 || (C % epv) || (C / epv) > EW_THREADS ||
        (EW_THREADS % (C / epv))) {
        set_error("%s: unsupported dtype/channels (dtype=%d C=%d)", who, dtype, C);
        return false;
    }
    return true;
}

}  // namespace frhip

using namespace frhip;

extern "C" int frhip_colreduce_blocks(int rows, int c, int dtype) {
    // 0 = this (dtype, width) is not served by the element-wise kernels (the same condition shape_ok() reports with a message);
    // callers size their partial-sum buffer with the result, so it must never divide by zero or come out as zero rows silently
    const int epv = dtype == FRHIP_DT_BF16 ? 8 : 4;
    if ((dtype != FRHIP_DT_BF16 && dtype != FRHIP_DT_F32) || rows <= 0 || c < epv || (c % epv) || c / epv > EW_THREADS ||
        (EW_THREADS % (c / epv)))
        return 0;
    const int rlanes = EW_THREADS / (c / epv);
    int b = (rows + rlanes * 8 - 1) / (rlanes * 8);
    if (b > 1024) b = 1024;
    if (b < 1) b = 1;
    return b;
}

extern "C" int frhip_colstats(int dtype, const void* x, int rows, int c, float* partial, hipStream_t stream) {
    if (!shape_ok(dtype, c, "frhip_colstats")) return FRHIP_EINVAL;
    const int blocks = frhip_colreduce_blocks(rows, c, dtype);
    if (dtype == FRHIP_DT_BF16)
        hipLaunchKernelGGL((colreduce_kernel<bf16_t, false>), dim3(blocks), dim3(EW_THREADS), 0, stream,
                           (const bf16_t*)x, nullptr, nullptr, nullptr, nullptr, nullptr, partial, rows, c);
    else
        hipLaunchKernelGGL((colreduce_kernel<float, false>), dim3(blocks), dim3(EW_THREADS), 0, stream,
                           (const float*)x, nullptr, nullptr, nullptr, nullptr, nullptr, partial, rows, c);
    return check_launch("frhip_colstats");
}

extern "C" int frhip_bn_bwd_reduce(int dtype, const void* dout, const void* y, const float* mean, const float* invstd,
                                   const float* mask_scale, const float* mask_shift, int rows, int c, float* partial,
                                   hipStream_t stream) {
    if (!shape_ok(dtype, c, "frhip_bn_bwd_reduce")) return FRHIP_EINVAL;
    const int blocks = frhip_colreduce_blocks(rows, c, dtype);
    if (dtype == FRHIP_DT_BF16)
        hipLaunchKernelGGL((colreduce_kernel<bf16_t, true>), dim3(blocks), dim3(EW_THREADS), 0, stream,
                           (const bf16_t*)dout, (const bf16_t*)y, mean, invstd, mask_scale, mask_shift, partial, rows, c);
    else
        hipLaunchKernelGGL((colreduce_kernel<float, true>), dim3(blocks), dim3(EW_THREADS), 0, stream,
                           (const float*)dout, (const float*)y, mean, invstd, mask_scale, mask_shift, partial, rows, c);
    return check_launch("frhip_bn_bwd_reduce");
}

// Up to FOLD_LIMIT partial rows go straight into the finalize kernel (16 lanes per channel, 4 rows in flight per lane);
// beyond that a first pass folds them to RED_GROUPS rows.
constexpr int FOLD_LIMIT = 512;
static const float* fold_partials(const float* partial, int& nparts, int c, float* scratch, hipStream_t stream) {
    if (nparts <= FOLD_LIMIT) return partial;
    // scratch holds 64 rows of [2][c]: always fold to 64 rows -- a lane of the fold kernel then owns <= 7 rows of a 28 x 28 layer (one round
    // trip of loads) and 25 of a 56 x 56 layer (two)
    const int groups = 64;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((2 * c + 63) / 64, groups), dim3(256), 0, stream,
                       partial, scratch, nparts, c);
    nparts = groups;
    return scratch;
}
static int fin_threads(int nparts) { return nparts > 64 ? 1024 : 256; }
// FRHIP_BN_FIN_SMALL (default 1): folds of more than 64 rows by 256-thread blocks of 16 channels (0: 1024-thread blocks of 64 channels)
static const int g_fin_small = getenv("FRHIP_BN_FIN_SMALL") ? atoi(getenv("FRHIP_BN_FIN_SMALL")) : 1;
static bool fin_small(int nparts) { return g_fin_small == 1 ? nparts > 64 : (g_fin_small >= 2 ? nparts >= g_fin_small : false); }   // >= 2: a row threshold

extern "C" int frhip_bn_finalize(const float* partial, int nparts, float* scratch, int c, float count,
                                 const float* gamma, const float* beta, float* running_mean, float* running_var,
                                 float momentum, float eps, float* mean, float* invstd, float* scale, float* shift,
                                 hipStream_t stream) {
    // scratch: 64*2*c floats
    const float* p = fold_partials(partial, nparts, c, scratch, stream);
    if (fin_small(nparts))
        hipLaunchKernelGGL(bn_finalize_kernel<true>, dim3((c + 15) / 16), dim3(256), 0, stream, p, nparts, c, count, gamma, beta,
                           running_mean, running_var, momentum, eps, mean, invstd, scale, shift);
    else
        hipLaunchKernelGGL(bn_finalize_kernel<false>, dim3((c + 63) / 64), dim3(fin_threads(nparts)), 0, stream, p, nparts, c, count, gamma, beta,
                           running_mean, running_var, momentum, eps, mean, invstd, scale, shift);
    return check_launch("frhip_bn_finalize");
}

extern "C" int frhip_bn_eval_affine(int c, const float* gamma, const float* beta, const float* running_mean,
                                    const float* running_var, float eps, float* scale, float* shift, hipStream_t stream) {
    hipLaunchKernelGGL(bn_eval_kernel, dim3((c + 63) / 64), dim3(64), 0, stream, c, gamma, beta, running_mean,
                       running_var, eps, scale, shift);
    return check_launch("frhip_bn_eval_affine");
}

// Stand-in BatchNorm state for the stem's backward reduction over the POOLED map (nets/_backbone.py stem_reduction_operands):
// mean := beta, invstd := gamma / (gamma^2 + (k beta)^2 + 1e-20) (a regularised 1 / gamma), scale := 1, shift := 0 -- one launch
// instead of eight one-workgroup torch kernels on the main stream in front of the stem's backward pass.  Separately rounded operations
// (no contraction): the same values as the torch expression it replaces.
__global__ void bn_standin_kernel(int c, const float* __restrict__ gamma, const float* __restrict__ beta, float k,
                                  float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ scale, float* __restrict__ shift) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= c) return;
    const float g = gamma[i], b = beta[i], kb = __fmul_rn(k, b);
    const float den = __fadd_rn(__fadd_rn(__fmul_rn(g, g), __fmul_rn(kb, kb)), 1e-20f);
    mean[i] = b; invstd[i] = __fdiv_rn(g, den); scale[i] = 1.f; shift[i] = 0.f;
}
extern "C" int frhip_bn_standin_state(int c, const float* gamma, const float* beta, float k, float* mean, float* invstd,
                                      float* scale, float* shift, hipStream_t stream) {
    if (c <= 0 || !gamma || !beta || !mean || !invstd || !scale || !shift) { frhip::set_error("frhip_bn_standin_state: bad arguments"); return FRHIP_EINVAL; }
    hipLaunchKernelGGL(bn_standin_kernel, dim3((c + 63) / 64), dim3(64), 0, stream, c, gamma, beta, k, mean, invstd, scale, shift);
    return check_launch("frhip_bn_standin_state");
}

extern "C" int frhip_bn_bwd_finalize(const float* partial, int nparts, float* scratch, int c, float count,
                                     const float* gamma, const float* mean, const float* invstd, float* dgamma,
                                     float* dbeta, float* ca, float* cb, float* cc, hipStream_t stream) {
    const float* p = fold_partials(partial, nparts, c, scratch, stream);
    if (fin_small(nparts))
        hipLaunchKernelGGL(bn_bwd_finalize_kernel<true>, dim3((c + 15) / 16), dim3(256), 0, stream, p, nparts, c, count, gamma,
                           mean, invstd, dgamma, dbeta, ca, cb, cc);
    else
        hipLaunchKernelGGL(bn_bwd_finalize_kernel<false>, dim3((c + 63) / 64), dim3(fin_threads(nparts)), 0, stream, p, nparts, c, count, gamma,
                           mean, invstd, dgamma, dbeta, ca, cb, cc);
    return check_launch("frhip_bn_bwd_finalize");
}

// Stochastic-depth variants (nets/AlterNet_SwinV2_FAN.py: x + drop_path(norm(branch))): rowscale[g] multiplies the normalised branch
// (forward) / the incoming gradient (backward) of the rows_per consecutive rows of sample g.  Plain-load instantiations only.
extern "C" int frhip_bn_apply_rs(int dtype, const void* y, const float* scale, const float* shift, const void* res,
                                 const float* rowscale, int rows_per, void* out, int rows, int c, hipStream_t stream) {
    if (!shape_ok(dtype, c, "frhip_bn_apply_rs")) return FRHIP_EINVAL;
    if (!rowscale || rows_per <= 0) { set_error("frhip_bn_apply_rs: rowscale and rows_per are required"); return FRHIP_EINVAL; }
    const int blocks = ew_row_blocks(rows, c, dtype);
    if (dtype == FRHIP_DT_BF16)
        hipLaunchKernelGGL((bn_apply_kernel<bf16_t, false, false, true>), dim3(blocks), dim3(EW_THREADS), 0, stream, (const bf16_t*)y, scale,
                           shift, (const bf16_t*)res, nullptr, nullptr, 0, (bf16_t*)out, rows, c, rowscale, rows_per);
    else
        hipLaunchKernelGGL((bn_apply_kernel<float, false, false, true>), dim3(blocks), dim3(EW_THREADS), 0, stream, (const float*)y, scale,
                           shift, (const float*)res, nullptr, nullptr, 0, (float*)out, rows, c, rowscale, rows_per);
    return check_launch("frhip_bn_apply_rs");
}

extern "C" int frhip_bn_bwd_reduce_rs(int dtype, const void* dout, const void* y, const float* mean, const float* invstd,
                                      const float* rowscale, int rows_per, int rows, int c, float* partial, hipStream_t stream) {
    if (!shape_ok(dtype, c, "frhip_bn_bwd_reduce_rs")) return FRHIP_EINVAL;
    if (!rowscale || rows_per <= 0) { set_error("frhip_bn_bwd_reduce_rs: rowscale and rows_per are required"); return FRHIP_EINVAL; }
    const int blocks = frhip_colreduce_blocks(rows, c, dtype);
    if (dtype == FRHIP_DT_BF16)
        hipLaunchKernelGGL((colreduce_kernel<bf16_t, true, true>), dim3(blocks), dim3(EW_THREADS), 0, stream, (const bf16_t*)dout,
                           (const bf16_t*)y, mean, invstd, nullptr, nullptr, partial, rows, c, rowscale, rows_per);
    else
        hipLaunchKernelGGL((colreduce_kernel<float, true, true>), dim3(blocks), dim3(EW_THREADS), 0, stream, (const float*)dout,
                           (const float*)y, mean, invstd, nullptr, nullptr, partial, rows, c, rowscale, rows_per);
    return check_launch("frhip_bn_bwd_reduce_rs");
}

extern "C" int frhip_bn_bwd_apply_rs(int dtype, const void* dout, const void* y, const float* ca, const float* cb, const float* cc,
                                     const float* rowscale, int rows_per, void* dy, int rows, int c, hipStream_t stream) {
    if (!shape_ok(dtype, c, "frhip_bn_bwd_apply_rs")) return FRHIP_EINVAL;
    if (!rowscale || rows_per <= 0) { set_error("frhip_bn_bwd_apply_rs: rowscale and rows_per are required"); return FRHIP_EINVAL; }
    const int blocks = ew_row_blocks(rows, c, dtype);
    if (dtype == FRHIP_DT_BF16)
        hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, false, false, true>), dim3(blocks), dim3(EW_THREADS), 0, stream, (const bf16_t*)dout,
                           (const bf16_t*)y, ca, cb, cc, nullptr, nullptr, (bf16_t*)dy, rows, c, rowscale, rows_per);
    else
        hipLaunchKernelGGL((bn_bwd_apply_kernel<float, false, false, true>), dim3(blocks), dim3(EW_THREADS), 0, stream, (const float*)dout,
                           (const float*)y, ca, cb, cc, nullptr, nullptr, (float*)dy, rows, c, rowscale, rows_per);
    return check_launch("frhip_bn_bwd_apply_rs");
}

extern "C" int frhip_bn_apply(int dtype, const void* y, const float* scale, const float* shift, const void* res,
                              const float* res_scale, const float* res_shift, int relu, void* out, int rows, int c,
                              hipStream_t stream) {
    if (!shape_ok(dtype, c, "frhip_bn_apply")) return FRHIP_EINVAL;
    const int blocks = ew_row_blocks(rows, c, dtype);
    const size_t bytes = (size_t)rows * c * (dtype == FRHIP_DT_BF16 ? 2 : 4);
    const bool ntl = (g_ew_nt & 1) && bytes >= EW_NT_BYTES && (res != nullptr || (g_ew_nt & 4)), nts = (g_ew_nt & 2) && bytes >= EW_NT_BYTES / 2;
#define BN_APPLY_GO(T, L, S)                                                                                                  \
    hipLaunchKernelGGL((bn_apply_kernel<T, L, S>), dim3(blocks), dim3(EW_THREADS), 0, stream, (const T*)y, scale, shift,     \
                       (const T*)res, res_scale, res_shift, relu, (T*)out, rows, c)
#define BN_APPLY_PICK(T)                                                                                                      \
    do { if (ntl && nts) BN_APPLY_GO(T, true, true); else if (ntl) BN_APPLY_GO(T, true, false);                               \
         else if (nts) BN_APPLY_GO(T, false, true); else BN_APPLY_GO(T, false, false); } while (0)
    if (y == out) {            // in place: the instantiation that reads through `out` (plain loads and stores)
        if (dtype == FRHIP_DT_BF16)
            hipLaunchKernelGGL((bn_apply_kernel<bf16_t, false, false, false, true>), dim3(blocks), dim3(EW_THREADS), 0, stream, nullptr, scale, shift,
                               (const bf16_t*)res, res_scale, res_shift, relu, (bf16_t*)out, rows, c);
        else
            hipLaunchKernelGGL((bn_apply_kernel<float, false, false, false, true>), dim3(blocks), dim3(EW_THREADS), 0, stream, nullptr, scale, shift,
                               (const float*)res, res_scale, res_shift, relu, (float*)out, rows, c);
        return check_launch("frhip_bn_apply");
    }
    if (dtype == FRHIP_DT_BF16) BN_APPLY_PICK(bf16_t); else BN_APPLY_PICK(float);
#undef BN_APPLY_PICK
#undef BN_APPLY_GO
    return check_launch("frhip_bn_apply");
}

extern "C" int frhip_bn_bwd_apply(int dtype, const void* dout, const void* y, const float* ca, const float* cb,
                                  const float* cc, const float* mask_scale, const float* mask_shift, void* dy,
                                  int rows, int c, hipStream_t stream) {
    if (!shape_ok(dtype, c, "frhip_bn_bwd_apply")) return FRHIP_EINVAL;
    const int blocks = ew_row_blocks(rows, c, dtype);
    const size_t bytes = (size_t)rows * c * (dtype == FRHIP_DT_BF16 ? 2 : 4);
    const bool ntl = (g_ew_nt & 1) && bytes >= EW_NT_BYTES, nts = (g_ew_nt & 2) && bytes >= EW_NT_BYTES / 2;
#define BN_BWD_GO(T, L, S)                                                                                                    \
    hipLaunchKernelGGL((bn_bwd_apply_kernel<T, L, S>), dim3(blocks), dim3(EW_THREADS), 0, stream, (const T*)dout, (const T*)y, \
                       ca, cb, cc, mask_scale, mask_shift, (T*)dy, rows, c)
#define BN_BWD_PICK(T)                                                                                                        \
    do { if (ntl && nts) BN_BWD_GO(T, true, true); else if (ntl) BN_BWD_GO(T, true, false);                                   \
         else if (nts) BN_BWD_GO(T, false, true); else BN_BWD_GO(T, false, false); } while (0)
    if (dtype == FRHIP_DT_BF16) BN_BWD_PICK(bf16_t); else BN_BWD_PICK(float);
#undef BN_BWD_PICK
#undef BN_BWD_GO
    return check_launch("frhip_bn_bwd_apply");
}

extern "C" int frhip_sum_partials(const float* partial, int nparts, int c, int which, float* out_accum, hipStream_t stream) {
    hipLaunchKernelGGL(sum_partials_kernel, dim3((c + 15) / 16), dim3(256), 0, stream, partial, nparts, c, which, out_accum);
    return check_launch("frhip_sum_partials");
}

extern "C" int frhip_add_bias(float* x, const float* bias, int rows, int c, hipStream_t stream) {
    const size_t n = (size_t)rows * c;
    hipLaunchKernelGGL(add_bias_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, x, bias, n, c);
    return check_launch("frhip_add_bias");
}

extern "C" int frhip_cast_from_f32(int dtype, const float* src, void* dst, size_t n, hipStream_t stream) {
    if (dtype == FRHIP_DT_BF16)
        hipLaunchKernelGGL(cast_from_f32_kernel<bf16_t>, dim3(grid_for(n, 256)), dim3(256), 0, stream, src, (bf16_t*)dst, n);
    else
        hipLaunchKernelGGL(cast_from_f32_kernel<float>, dim3(grid_for(n, 256)), dim3(256), 0, stream, src, (float*)dst, n);
    return check_launch("frhip_cast_from_f32");
}

extern "C" int frhip_cast_to_f32(int dtype, const void* src, float* dst, size_t n, hipStream_t stream) {
    if (dtype == FRHIP_DT_BF16)
        hipLaunchKernelGGL(cast_to_f32_kernel<bf16_t>, dim3(grid_for(n, 256)), dim3(256), 0, stream, (const bf16_t*)src, dst, n);
    else
        hipLaunchKernelGGL(cast_to_f32_kernel<float>, dim3(grid_for(n, 256)), dim3(256), 0, stream, (const float*)src, dst, n);
    return check_launch("frhip_cast_to_f32");
}
