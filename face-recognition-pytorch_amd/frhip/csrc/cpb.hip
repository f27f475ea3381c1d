// Continuous position bias of the SwinV2-style window attention, all attention blocks of a step in ONE launch.
//   bias[h][i][j] = 16 * sigmoid( cpb_mlp(coords_table)[index[i][j]][h] ),  cpb_mlp = Linear(2,512,bias) -> ReLU -> Linear(512,heads)
//   scale[h]      = exp(min(logit_scale[h], ln 100))
// Reference: /root/reference/nets/SwinV2.py:88-125 (tables), :150-158 (use), nets/AlterNet_SwinV2_FAN.py:210-248, :276-283.
// The MLP runs on a (2 ws - 1)^2-entry table (169 entries for 7x7 windows): a few hundred thousand MACs per block.  As torch
// ops it is ~8 launches per block forward and ~12 backward (tiny rocBLAS GEMMs + element-wise kernels) whose HOST cost
// dominates a SwinV2 step; here forward and backward are one workgroup per block, fp32, deterministic.
#include "common.h"
#include "frhip.h"

namespace frhip {

constexpr int CPB_HIDDEN = 512;
constexpr int CPB_MAX_T = 169, CPB_MAX_HEADS = 32;

__device__ __forceinline__ const frhip_cpb_block& cpb_desc(const frhip_cpb_block* tab) { return tab[blockIdx.x]; }

// t[T][heads] into LDS: thread (e, h) pairs strided over the workgroup; the hidden row is recomputed per pair (2 FMA + max)
__device__ __forceinline__ void cpb_mlp_table(const frhip_cpb_block& d, float* __restrict__ t_lds) {
    const float* tab = reinterpret_cast<const float*>(d.coords);
    const float* w0 = reinterpret_cast<const float*>(d.w0);
    const float* b0 = reinterpret_cast<const float*>(d.b0);
    const float* w2 = reinterpret_cast<const float*>(d.w2);
    for (int p = threadIdx.x; p < d.entries * d.heads; p += blockDim.x) {
        const int e = p / d.heads, h = p - e * d.heads;
        const float c0 = tab[2 * e], c1 = tab[2 * e + 1];
        float acc = 0.f;
        for (int j = 0; j < CPB_HIDDEN; ++j) {
            const float hid = fmaxf(fmaf(c0, w0[2 * j], fmaf(c1, w0[2 * j + 1], b0[j])), 0.f);
            acc = fmaf(hid, w2[h * CPB_HIDDEN + j], acc);
        }
        t_lds[p] = acc;
    }
}

__global__ __launch_bounds__(256) void cpb_fwd_kernel(const frhip_cpb_block* __restrict__ blocks) {
    __shared__ float t_lds[CPB_MAX_T * CPB_MAX_HEADS];
    const frhip_cpb_block& d = cpb_desc(blocks);
    cpb_mlp_table(d, t_lds);
    __syncthreads();
    const int64_t* index = reinterpret_cast<const int64_t*>(d.index);
    float* bias = reinterpret_cast<float*>(d.bias);
    const int nn = d.tokens * d.tokens;
    for (int p = threadIdx.x; p < d.heads * nn; p += blockDim.x) {
        const int h = p / nn, ij = p - h * nn;
        const float v = t_lds[(int)index[ij] * d.heads + h];
        bias[p] = 16.f / (1.f + expf(-v));
    }
    const float* ls = reinterpret_cast<const float*>(d.logit_scale);
    float* scale = reinterpret_cast<float*>(d.scale);
    for (int h = threadIdx.x; h < d.heads; h += blockDim.x) scale[h] = expf(fminf(ls[h], 4.605170185988092f));
}

// gradients of the four parameters, ADDED into their (caller-zeroed) gradient tensors
__global__ __launch_bounds__(256) void cpb_bwd_kernel(const frhip_cpb_block* __restrict__ blocks) {
    __shared__ float dt_lds[CPB_MAX_T * CPB_MAX_HEADS];
    const frhip_cpb_block& d = cpb_desc(blocks);
    const int nn = d.tokens * d.tokens, TH = d.entries * d.heads;
    for (int p = threadIdx.x; p < TH; p += blockDim.x) dt_lds[p] = 0.f;
    __syncthreads();
    // d t[index[i][j]][h] += dbias[h][i][j] * 16 s (1 - s), s = bias / 16
    const int64_t* index = reinterpret_cast<const int64_t*>(d.index);
    const float* bias = reinterpret_cast<const float*>(d.bias);
    const float* dbias = reinterpret_cast<const float*>(d.dbias);
    for (int p = threadIdx.x; p < d.heads * nn; p += blockDim.x) {
        const int h = p / nn, ij = p - h * nn;
        const float s = bias[p] * (1.f / 16.f);
        atomicAdd(&dt_lds[(int)index[ij] * d.heads + h], dbias[p] * 16.f * s * (1.f - s));
    }
    __syncthreads();
    const float* tab = reinterpret_cast<const float*>(d.coords);
    const float* w0 = reinterpret_cast<const float*>(d.w0);
    const float* b0 = reinterpret_cast<const float*>(d.b0);
    const float* w2 = reinterpret_cast<const float*>(d.w2);
    float* dw0 = reinterpret_cast<float*>(d.dw0);
    float* db0 = reinterpret_cast<float*>(d.db0);
    float* dw2 = reinterpret_cast<float*>(d.dw2);
    // one hidden unit j per thread (two rounds of 256): walk the table entries, recompute hid[e][j]
    for (int j = threadIdx.x; j < CPB_HIDDEN; j += blockDim.x) {
        const float wa = w0[2 * j], wb = w0[2 * j + 1], bb = b0[j];
        float g0 = 0.f, g1 = 0.f, gb = 0.f;
        for (int h = 0; h < d.heads; ++h) {
            const float w2hj = w2[h * CPB_HIDDEN + j];
            float gw2 = 0.f;
            for (int e = 0; e < d.entries; ++e) {
                const float pre = fmaf(tab[2 * e], wa, fmaf(tab[2 * e + 1], wb, bb));
                const float dth = dt_lds[e * d.heads + h];
                gw2 = fmaf(dth, fmaxf(pre, 0.f), gw2);
                if (pre > 0.f) {
                    const float dh = dth * w2hj;
                    g0 = fmaf(dh, tab[2 * e], g0); g1 = fmaf(dh, tab[2 * e + 1], g1); gb += dh;
                }
            }
            dw2[h * CPB_HIDDEN + j] += gw2;
        }
        dw0[2 * j] += g0; dw0[2 * j + 1] += g1; db0[j] += gb;
    }
    const float* ls = reinterpret_cast<const float*>(d.logit_scale);
    const float* scale = reinterpret_cast<const float*>(d.scale);
    const float* dscale = reinterpret_cast<const float*>(d.dscale);
    float* dls = reinterpret_cast<float*>(d.dlogit_scale);
    for (int h = threadIdx.x; h < d.heads; h += blockDim.x)
        dls[h] += ls[h] <= 4.605170185988092f ? dscale[h] * scale[h] : 0.f;      // clamp(max) passes the gradient on <=
}

static int cpb_check(int nblocks, const char* who) {
    if (nblocks <= 0) { set_error("%s: no blocks", who); return FRHIP_EINVAL; }
    return FRHIP_OK;
}

}  // namespace frhip

using namespace frhip;

extern "C" int frhip_cpb_limits(int* max_entries, int* max_heads, int* hidden) {
    if (max_entries) *max_entries = CPB_MAX_T;
    if (max_heads) *max_heads = CPB_MAX_HEADS;
    if (hidden) *hidden = CPB_HIDDEN;
    return FRHIP_OK;
}

extern "C" int frhip_cpb_fwd(const frhip_cpb_block* blocks_dev, int nblocks, hipStream_t stream) {
    int rc = cpb_check(nblocks, "frhip_cpb_fwd");
    if (rc) return rc;
    hipLaunchKernelGGL(cpb_fwd_kernel, dim3(nblocks), dim3(256), 0, stream, blocks_dev);
    return check_launch("frhip_cpb_fwd");
}

extern "C" int frhip_cpb_bwd(const frhip_cpb_block* blocks_dev, int nblocks, hipStream_t stream) {
    int rc = cpb_check(nblocks, "frhip_cpb_bwd");
    if (rc) return rc;
    hipLaunchKernelGGL(cpb_bwd_kernel, dim3(nblocks), dim3(256), 0, stream, blocks_dev);
    return check_launch("frhip_cpb_bwd");
}
