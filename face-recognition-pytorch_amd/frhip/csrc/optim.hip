// Multi-tensor SGD (momentum + weight decay) with the global-norm gradient clip folded in, for gfx950.
// Replaces torch.optim.SGD.step() + torch.nn.utils.clip_grad_norm_ of the reference's training step
// (/root/reference/model/FR_PartialFC.py:153-160 optimizer, :181-190 clip + step): ~110 small ATen launches become
// three kernels that stream every parameter, gradient and momentum buffer exactly once (HBM-bound: 20 B per element).
// Semantics = torch.optim.SGD with dampening 0, nesterov off: d = g*clip + wd*p; buf = mom*buf + d; p -= lr*buf
// (a zero-initialised buffer makes the first step buf = d, as torch's clone does).
#include "common.h"
#include "frhip.h"

namespace frhip {

struct SgdGroups { frhip_sgd_group g[FRHIP_SGD_MAX_GROUPS]; };

__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// partial[chunk] = sum of squares of the chunk's gradient (0 for chunks of groups that are not clipped)
__global__ __launch_bounds__(256) void sgd_sumsq_kernel(const frhip_sgd_chunk* __restrict__ chunks, SgdGroups groups,
                                                        float* __restrict__ partial) {
    __shared__ float red[4];
    const frhip_sgd_chunk c = chunks[blockIdx.x];
    float acc = 0.f;
    if (groups.g[c.group].clip != 0.f) {
        const float* g = c.g;
        if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
            const uint32_t n4 = c.n >> 2;
            for (uint32_t i = threadIdx.x; i < n4; i += 256) {
                const f32x4_t v = reinterpret_cast<const f32x4_t*>(g)[i];
                acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
            }
            for (uint32_t i = (n4 << 2) + threadIdx.x; i < c.n; i += 256) acc += g[i] * g[i];
        } else {
            for (uint32_t i = threadIdx.x; i < c.n; i += 256) acc += g[i] * g[i];
        }
    }
    const float s = block_sum_256(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// out[0] = min(1, max_norm / (norm + 1e-6)), out[1] = norm   (torch.nn.utils.clip_grad_norm_)
__global__ __launch_bounds__(256) void clip_coef_kernel(const float* __restrict__ partial, int n, float max_norm,
                                                        float* __restrict__ out) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
    const float s = block_sum_256(acc, red);
    if (threadIdx.x == 0) {
        const float norm = sqrtf(s);
        const float coef = max_norm / (norm + 1e-6f);
        out[0] = coef < 1.f ? coef : 1.f;
        out[1] = norm;
    }
}

__global__ __launch_bounds__(256) void sgd_multi_kernel(const frhip_sgd_chunk* __restrict__ chunks, SgdGroups groups,
                                                        const float* __restrict__ clip) {
    const frhip_sgd_chunk c = chunks[blockIdx.x];
    const frhip_sgd_group gr = groups.g[c.group];
    const float coef = (gr.clip != 0.f && clip) ? clip[0] : 1.f;
    const float lr = gr.lr, wd = gr.weight_decay, mom = gr.momentum;
    float* __restrict__ p = c.p;
    const float* __restrict__ g = c.g;
    float* __restrict__ m = c.m;
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m)) & 15) == 0;
    uint32_t done = 0;
    if (vec) {
        const uint32_t n4 = c.n >> 2;
        for (uint32_t i = threadIdx.x; i < n4; i += 256) {
            f32x4_t pv = reinterpret_cast<f32x4_t*>(p)[i];
            const f32x4_t gv = reinterpret_cast<const f32x4_t*>(g)[i];
            f32x4_t mv = m ? reinterpret_cast<f32x4_t*>(m)[i] : f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = gv[e] * coef + wd * pv[e];
                mv[e] = mom * mv[e] + d;
                pv[e] -= lr * (m ? mv[e] : d);
            }
            if (m) reinterpret_cast<f32x4_t*>(m)[i] = mv;
            reinterpret_cast<f32x4_t*>(p)[i] = pv;
        }
        done = n4 << 2;
    }
    for (uint32_t i = done + threadIdx.x; i < c.n; i += 256) {
        const float d = g[i] * coef + wd * p[i];
        float upd = d;
        if (m) { const float b = mom * m[i] + d; m[i] = b; upd = b; }
        p[i] -= lr * upd;
    }
}

struct AdamGroups { frhip_adamw_group g[FRHIP_SGD_MAX_GROUPS]; };

// torch.optim.AdamW (amsgrad off): p *= 1 - lr*wd; m = b1*m + (1-b1)*g; v = b2*v + (1-b2)*g^2;
// p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)      with g scaled by the clip coefficient first
__global__ __launch_bounds__(256) void adamw_multi_kernel(const frhip_adamw_chunk* __restrict__ chunks, AdamGroups groups,
                                                          const float* __restrict__ clip) {
    const frhip_adamw_chunk c = chunks[blockIdx.x];
    const frhip_adamw_group gr = groups.g[c.group];
    const float coef = (gr.clip != 0.f && clip) ? clip[0] : 1.f;
    const float decay = 1.f - gr.lr * gr.weight_decay, step_size = gr.lr / gr.bc1, inv_sqrt_bc2 = rsqrtf(gr.bc2);
    const float b1 = gr.beta1, b2 = gr.beta2, eps = gr.eps;
    float* __restrict__ p = c.p;
    const float* __restrict__ g = c.g;
    float* __restrict__ m = c.m;
    float* __restrict__ v = c.v;
    auto upd = [&](float& pe, float ge, float& me, float& ve) {
        ge *= coef;
        pe *= decay;
        me = b1 * me + (1.f - b1) * ge;
        ve = b2 * ve + (1.f - b2) * ge * ge;
        pe -= step_size * me / (sqrtf(ve) * inv_sqrt_bc2 + eps);
    };
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                       reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    uint32_t done = 0;
    if (vec) {
        const uint32_t n4 = c.n >> 2;
        for (uint32_t i = threadIdx.x; i < n4; i += 256) {
            f32x4_t pv = reinterpret_cast<f32x4_t*>(p)[i], mv = reinterpret_cast<f32x4_t*>(m)[i], vv = reinterpret_cast<f32x4_t*>(v)[i];
            const f32x4_t gv = reinterpret_cast<const f32x4_t*>(g)[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) { float pe = pv[e], me = mv[e], ve = vv[e]; upd(pe, gv[e], me, ve); pv[e] = pe; mv[e] = me; vv[e] = ve; }
            reinterpret_cast<f32x4_t*>(p)[i] = pv; reinterpret_cast<f32x4_t*>(m)[i] = mv; reinterpret_cast<f32x4_t*>(v)[i] = vv;
        }
        done = n4 << 2;
    }
    for (uint32_t i = done + threadIdx.x; i < c.n; i += 256) upd(p[i], g[i], m[i], v[i]);
}

// the clip norm of the AdamW table (same reduction as sgd_sumsq_kernel, other chunk layout)
__global__ __launch_bounds__(256) void adamw_sumsq_kernel(const frhip_adamw_chunk* __restrict__ chunks, AdamGroups groups,
                                                          float* __restrict__ partial) {
    __shared__ float red[4];
    const frhip_adamw_chunk c = chunks[blockIdx.x];
    float acc = 0.f;
    if (groups.g[c.group].clip != 0.f)
        for (uint32_t i = threadIdx.x; i < c.n; i += 256) acc += c.g[i] * c.g[i];
    const float s = block_sum_256(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

}  // namespace frhip

using namespace frhip;

static int pack_groups(const frhip_sgd_group* groups_host, int ngroups, SgdGroups& out, const char* who) {
    if (!groups_host || ngroups < 1 || ngroups > FRHIP_SGD_MAX_GROUPS) {
        set_error("%s: 1..%d parameter groups expected, got %d", who, FRHIP_SGD_MAX_GROUPS, ngroups);
        return FRHIP_EINVAL;
    }
    for (int i = 0; i < FRHIP_SGD_MAX_GROUPS; ++i) out.g[i] = groups_host[i < ngroups ? i : 0];
    return FRHIP_OK;
}

extern "C" int frhip_sgd_clip_coef(const frhip_sgd_chunk* chunks, int nchunks, const frhip_sgd_group* groups_host,
                                   int ngroups, float max_norm, float* partial, float* coef_out, hipStream_t stream) {
    SgdGroups gs;
    int rc = pack_groups(groups_host, ngroups, gs, "frhip_sgd_clip_coef");
    if (rc) return rc;
    if (!chunks || nchunks < 1 || !partial || !coef_out) { set_error("frhip_sgd_clip_coef: missing buffers"); return FRHIP_EINVAL; }
    hipLaunchKernelGGL(sgd_sumsq_kernel, dim3(nchunks), dim3(256), 0, stream, chunks, gs, partial);
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, stream, partial, nchunks, max_norm, coef_out);
    return check_launch("frhip_sgd_clip_coef");
}

extern "C" int frhip_sgd_multi(const frhip_sgd_chunk* chunks, int nchunks, const frhip_sgd_group* groups_host, int ngroups,
                               const float* clip_coef, hipStream_t stream) {
    SgdGroups gs;
    int rc = pack_groups(groups_host, ngroups, gs, "frhip_sgd_multi");
    if (rc) return rc;
    if (!chunks || nchunks < 1) { set_error("frhip_sgd_multi: empty chunk table"); return FRHIP_EINVAL; }
    hipLaunchKernelGGL(sgd_multi_kernel, dim3(nchunks), dim3(256), 0, stream, chunks, gs, clip_coef);
    return check_launch("frhip_sgd_multi");
}

static int pack_adam_groups(const frhip_adamw_group* groups_host, int ngroups, AdamGroups& out, const char* who) {
    if (!groups_host || ngroups < 1 || ngroups > FRHIP_SGD_MAX_GROUPS) {
        set_error("%s: 1..%d parameter groups expected, got %d", who, FRHIP_SGD_MAX_GROUPS, ngroups);
        return FRHIP_EINVAL;
    }
    for (int i = 0; i < FRHIP_SGD_MAX_GROUPS; ++i) out.g[i] = groups_host[i < ngroups ? i : 0];
    return FRHIP_OK;
}

extern "C" int frhip_adamw_clip_coef(const frhip_adamw_chunk* chunks, int nchunks, const frhip_adamw_group* groups_host,
                                     int ngroups, float max_norm, float* partial, float* coef_out, hipStream_t stream) {
    AdamGroups gs;
    int rc = pack_adam_groups(groups_host, ngroups, gs, "frhip_adamw_clip_coef");
    if (rc) return rc;
    if (!chunks || nchunks < 1 || !partial || !coef_out) { set_error("frhip_adamw_clip_coef: missing buffers"); return FRHIP_EINVAL; }
    hipLaunchKernelGGL(adamw_sumsq_kernel, dim3(nchunks), dim3(256), 0, stream, chunks, gs, partial);
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, stream, partial, nchunks, max_norm, coef_out);
    return check_launch("frhip_adamw_clip_coef");
}

extern "C" int frhip_adamw_multi(const frhip_adamw_chunk* chunks, int nchunks, const frhip_adamw_group* groups_host,
                                 int ngroups, const float* clip_coef, hipStream_t stream) {
    AdamGroups gs;
    int rc = pack_adam_groups(groups_host, ngroups, gs, "frhip_adamw_multi");
    if (rc) return rc;
    if (!chunks || nchunks < 1) { set_error("frhip_adamw_multi: empty chunk table"); return FRHIP_EINVAL; }
    hipLaunchKernelGGL(adamw_multi_kernel, dim3(nchunks), dim3(256), 0, stream, chunks, gs, clip_coef);
    return check_launch("frhip_adamw_multi");
}
