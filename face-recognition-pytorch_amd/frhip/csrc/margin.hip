// Explicit-logit versions of the margin and the class-sharded softmax cross-entropy, for callers that use the
// reference modules stand-alone (nets/ArcFace.py:76-105 ArcFace/CosFace.forward, nets/PartialFC.py:441-484
// DistCrossEntropyFunc).  The fused head (head.hip) never materialises logits; these kernels exist so that the
// drop-in nets.ArcFace / nets.PartialFC.DistCrossEntropy keep the reference's tensor-in / tensor-out contract.
// HBM-bound row kernels: one 256-thread block per row, 16-byte accesses where the row pitch allows.
#include "common.h"
#include "frhip.h"

namespace frhip {

__device__ __forceinline__ float block_reduce(float v, bool is_max, float* red) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const float o = __shfl_xor(v, d); v = is_max ? fmaxf(v, o) : v + o; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
    return r;
}

// out = s * margin(logits): kind 0 = ArcFace (cos(theta+m), fallback t - m sin(pi-m) below cos(pi-m)), 1 = CosFace (t - m).
// tsave[row] keeps the raw target cosine for the backward slope.  In place on `logits`.
__global__ __launch_bounds__(256) void margin_fwd_kernel(float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                         int C, float s, float cos_m, float sin_m, float theta,
                                                         float sinmm, float m3, int kind, float* __restrict__ tsave) {
    const int row = blockIdx.x;
    float* x = logits + (size_t)row * C;
    const int64_t lab = labels[row];
    for (int j = threadIdx.x; j < C; j += 256) {
        float t = x[j];
        if (j == lab) {
            tsave[row] = t;
            if (kind == 0) {
                const float sin_t = sqrtf(1.f - t * t);
                t = t > theta ? t * cos_m - sin_t * sin_m : t - sinmm;
            } else {
                t = t - m3;
            }
        }
        x[j] = t * s;
    }
}

__global__ __launch_bounds__(256) void margin_bwd_kernel(const float* __restrict__ gout, const int64_t* __restrict__ labels,
                                                         const float* __restrict__ tsave, int C, float s, float cos_m,
                                                         float sin_m, float theta, int kind, float* __restrict__ gin) {
    const int row = blockIdx.x;
    const int64_t lab = labels[row];
    for (int j = threadIdx.x; j < C; j += 256) {
        float gg = gout[(size_t)row * C + j] * s;
        if (j == lab && kind == 0) {
            const float t = tsave[row];
            if (t > theta) gg *= cos_m + t * sin_m / sqrtf(1.f - t * t);
        }
        gin[(size_t)row * C + j] = gg;
    }
}

__global__ __launch_bounds__(256) void rows_max_kernel(const float* __restrict__ x, int C, float* __restrict__ rowmax) {
    __shared__ float red[4];
    const float* r = x + (size_t)blockIdx.x * C;
    float m = -INFINITY;
    for (int j = threadIdx.x; j < C; j += 256) m = fmaxf(m, r[j]);
    m = block_reduce(m, true, red);
    if (threadIdx.x == 0) rowmax[blockIdx.x] = m;
}

// x <- exp(x - rowmax) in place, rowsum = sum
__global__ __launch_bounds__(256) void rows_exp_sum_kernel(float* __restrict__ x, int C, const float* __restrict__ rowmax,
                                                           float* __restrict__ rowsum) {
    __shared__ float red[4];
    float* r = x + (size_t)blockIdx.x * C;
    const float m = rowmax[blockIdx.x];
    float s = 0.f;
    for (int j = threadIdx.x; j < C; j += 256) { const float e = __expf(r[j] - m); r[j] = e; s += e; }
    s = block_reduce(s, false, red);
    if (threadIdx.x == 0) rowsum[blockIdx.x] = s;
}

// x <- x / rowsum (probabilities); ptarget[row] = p at the label (0 when the label is -1)
__global__ __launch_bounds__(256) void rows_normalize_kernel(float* __restrict__ x, int C, const float* __restrict__ rowsum,
                                                             const int64_t* __restrict__ labels, float* __restrict__ ptarget) {
    float* r = x + (size_t)blockIdx.x * C;
    const float inv = 1.f / rowsum[blockIdx.x];
    const int64_t lab = labels[blockIdx.x];
    if (threadIdx.x == 0 && lab < 0) ptarget[blockIdx.x] = 0.f;
    for (int j = threadIdx.x; j < C; j += 256) {
        const float p = r[j] * inv;
        r[j] = p;
        if (j == lab) ptarget[blockIdx.x] = p;
    }
}

// grad = (p - onehot) / N * upstream   (in place on p)
__global__ __launch_bounds__(256) void ce_grad_kernel(float* __restrict__ p, int C, const int64_t* __restrict__ labels,
                                                      float inv_n, const float* __restrict__ upstream) {
    float* r = p + (size_t)blockIdx.x * C;
    const int64_t lab = labels[blockIdx.x];
    const float g = inv_n * upstream[0];
    for (int j = threadIdx.x; j < C; j += 256) r[j] = (r[j] - (j == lab ? 1.f : 0.f)) * g;
}

}  // namespace frhip

using namespace frhip;

extern "C" int frhip_margin_fwd(float* logits, const int64_t* labels, int n, int c, float s, float m, int kind,
                                float* tsave, hipStream_t stream) {
    const double pi = 3.14159265358979323846;
    hipLaunchKernelGGL(margin_fwd_kernel, dim3(n), dim3(256), 0, stream, logits, labels, c, s, (float)cos((double)m),
                       (float)sin((double)m), (float)cos(pi - (double)m), (float)(sin(pi - (double)m) * (double)m), m, kind, tsave);
    return check_launch("frhip_margin_fwd");
}

extern "C" int frhip_margin_bwd(const float* gout, const int64_t* labels, const float* tsave, int n, int c, float s,
                                float m, int kind, float* gin, hipStream_t stream) {
    const double pi = 3.14159265358979323846;
    hipLaunchKernelGGL(margin_bwd_kernel, dim3(n), dim3(256), 0, stream, gout, labels, tsave, c, s, (float)cos((double)m),
                       (float)sin((double)m), (float)cos(pi - (double)m), kind, gin);
    return check_launch("frhip_margin_bwd");
}

extern "C" int frhip_rows_max(const float* x, int n, int c, float* rowmax, hipStream_t stream) {
    hipLaunchKernelGGL(rows_max_kernel, dim3(n), dim3(256), 0, stream, x, c, rowmax);
    return check_launch("frhip_rows_max");
}

extern "C" int frhip_rows_exp_sum(float* x, int n, int c, const float* rowmax, float* rowsum, hipStream_t stream) {
    hipLaunchKernelGGL(rows_exp_sum_kernel, dim3(n), dim3(256), 0, stream, x, c, rowmax, rowsum);
    return check_launch("frhip_rows_exp_sum");
}

extern "C" int frhip_rows_normalize(float* x, int n, int c, const float* rowsum, const int64_t* labels, float* ptarget,
                                    hipStream_t stream) {
    hipLaunchKernelGGL(rows_normalize_kernel, dim3(n), dim3(256), 0, stream, x, c, rowsum, labels, ptarget);
    return check_launch("frhip_rows_normalize");
}

extern "C" int frhip_ce_grad(float* p, int n, int c, const int64_t* labels, float inv_n, const float* upstream,
                             hipStream_t stream) {
    hipLaunchKernelGGL(ce_grad_kernel, dim3(n), dim3(256), 0, stream, p, c, labels, inv_n, upstream);
    return check_launch("frhip_ce_grad");
}

// ---- verification pair scores (/root/reference/utils/eval.py:68-99): score = 1 - |a-b|^2 / 4 accumulated in float64
// from float32 differences in index order (the reference's arithmetic, so hist_idx = int(99999*score) is
// bit-exact), plus the two 100001-bin histograms by integer atomics.  One thread per pair.
namespace frhip {
__global__ void pair_score_kernel(const float* __restrict__ e1, const float* __restrict__ e2, const int64_t* __restrict__ labels,
                                  int n, int d, double* __restrict__ scores, int* __restrict__ hist_idx,
                                  int* __restrict__ hist_genuine, int* __restrict__ hist_imposter) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double sum = 0.0;
    for (int k = 0; k < d; ++k) {
        const float df = e1[(size_t)i * d + k] - e2[(size_t)i * d + k];
        const double dd = (double)df;
        sum += dd * dd;
    }
    const double score = 1.0 - sum / 4.0;
    const int idx = (int)((1e5 - 1.0) * score);
    scores[i] = score;
    hist_idx[i] = idx;
    if (idx >= 0 && idx <= 100000) atomicAdd(labels[i] ? hist_genuine + idx : hist_imposter + idx, 1);
}

// ---- cross-matching scores (/root/reference/utils/eval.py:102-137): every unordered pair (j < i) of one embedding set, in
// the reference's order l = i (i - 1) / 2 + j; same float64-of-float32-differences arithmetic as pair_score, label 1 where
// the identities agree.  One thread per pair, 16 x 16 pairs per block.
__global__ __launch_bounds__(256) void cross_score_kernel(const float* __restrict__ e, const int64_t* __restrict__ labels, int n, int d,
                                                          double* __restrict__ scores, double* __restrict__ pair_labels,
                                                          int* __restrict__ hist_idx, int* __restrict__ hist_genuine,
                                                          int* __restrict__ hist_imposter) {
    const int i = blockIdx.y * 16 + (threadIdx.x >> 4), j = blockIdx.x * 16 + (threadIdx.x & 15);
    if (i >= n || j >= i) return;
    double sum = 0.0;
    for (int k = 0; k < d; ++k) {
        const float df = e[(size_t)j * d + k] - e[(size_t)i * d + k];
        const double dd = (double)df;
        sum += dd * dd;
    }
    const double score = 1.0 - sum / 4.0;
    const int idx = (int)((1e5 - 1.0) * score);
    const size_t l = (size_t)i * (i - 1) / 2 + j;
    const bool genuine = labels[j] == labels[i];
    scores[l] = score;
    pair_labels[l] = genuine ? 1.0 : 0.0;
    hist_idx[l] = idx;
    if (idx >= 0 && idx <= 100000) atomicAdd(genuine ? hist_genuine + idx : hist_imposter + idx, 1);
}
}  // namespace frhip

extern "C" int frhip_pair_score(const float* e1, const float* e2, const int64_t* labels, int n, int d, double* scores,
                                int* hist_idx, int* hist_genuine, int* hist_imposter, hipStream_t stream) {
    if (n <= 0) return FRHIP_OK;
    hipLaunchKernelGGL(frhip::pair_score_kernel, dim3((n + 127) / 128), dim3(128), 0, stream, e1, e2, labels, n, d, scores,
                       hist_idx, hist_genuine, hist_imposter);
    return frhip::check_launch("frhip_pair_score");
}

extern "C" int frhip_cross_score(const float* e, const int64_t* labels, int n, int d, double* scores, double* pair_labels,
                                 int* hist_idx, int* hist_genuine, int* hist_imposter, hipStream_t stream) {
    if (n <= 1) return FRHIP_OK;
    if (n > 65535 * 16) { frhip::set_error("frhip_cross_score: n = %d too large", n); return FRHIP_EINVAL; }
    const int t = (n + 15) / 16;
    hipLaunchKernelGGL(frhip::cross_score_kernel, dim3(t, t), dim3(256), 0, stream, e, labels, n, d, scores, pair_labels,
                       hist_idx, hist_genuine, hist_imposter);
    return frhip::check_launch("frhip_cross_score");
}
