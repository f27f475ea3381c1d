// Continuous position bias of the SwinV2-style window attention, all attention blocks of a step in ONE launch.
//   bias[h][i][j] = 16 * sigmoid( cpb_mlp(coords_table)[index[i][j]][h] ),  cpb_mlp = Linear(2,512,bias) -> ReLU -> Linear(512,heads)
//   scale[h]      = exp(min(logit_scale[h], ln 100))
// Reference: /root/reference/nets/SwinV2.py:88-125 (tables), :150-158 (use), nets/AlterNet_SwinV2_FAN.py:210-248, :276-283.
// The MLP runs on a (2 ws - 1)^2-entry table (169 entries for 7x7 windows): a few hundred thousand MACs per block.  As torch
// ops it is ~8 launches per block forward and ~12 backward (tiny rocBLAS GEMMs + element-wise kernels) whose HOST cost
// dominates a SwinV2 step; here forward and backward are two launches each for ALL blocks, fp32, deterministic (no atomics):
// a first version with one workgroup per block took 0.31 + 0.63 ms of a 20-ms Swin34 step (serial loops over global memory);
// this one spreads (block, table entry) / (block, head) / (block, 64 hidden units) over the grid.
#include "common.h"
#include "frhip.h"

namespace frhip {

constexpr int CPB_HIDDEN = 512;
constexpr int CPB_MAX_T = 169, CPB_MAX_HEADS = 32;

// ---- forward, kernel 1: t[e][h] = cpb_mlp(coords[e])[h].  One workgroup of 128 threads per (block, table entry): the 512
//      hidden units are spread over the threads (4 each, coalesced rows of w2), heads are reduced over the workgroup.
__global__ __launch_bounds__(128) void cpb_table_kernel(const frhip_cpb_block* __restrict__ blocks, float* __restrict__ t_all,
                                                        int t_stride) {
    __shared__ float red[2][CPB_MAX_HEADS];
    const frhip_cpb_block& d = blocks[blockIdx.y];
    const int e = blockIdx.x;
    if (e >= d.entries) return;
    const float* tab = reinterpret_cast<const float*>(d.coords);
    const float* w0 = reinterpret_cast<const float*>(d.w0);
    const float* b0 = reinterpret_cast<const float*>(d.b0);
    const float* w2 = reinterpret_cast<const float*>(d.w2);
    const float c0 = tab[2 * e], c1 = tab[2 * e + 1];
    float hid[CPB_HIDDEN / 128];
#pragma unroll
    for (int r = 0; r < CPB_HIDDEN / 128; ++r) {
        const int j = r * 128 + threadIdx.x;
        hid[r] = fmaxf(fmaf(c0, w0[2 * j], fmaf(c1, w0[2 * j + 1], b0[j])), 0.f);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int h = 0; h < d.heads; ++h) {
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < CPB_HIDDEN / 128; ++r) acc = fmaf(hid[r], w2[h * CPB_HIDDEN + r * 128 + threadIdx.x], acc);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) red[wave][h] = acc;
    }
    __syncthreads();
    for (int h = threadIdx.x; h < d.heads; h += 128)
        t_all[(size_t)blockIdx.y * t_stride + e * d.heads + h] = red[0][h] + red[1][h];
}

// ---- forward, kernel 2: bias[h][i][j] = 16 sigmoid(t[index[i][j]][h]); scale[h] = exp(min(logit_scale[h], ln 100))
__global__ __launch_bounds__(256) void cpb_gather_kernel(const frhip_cpb_block* __restrict__ blocks, const float* __restrict__ t_all,
                                                         int t_stride) {
    const frhip_cpb_block& d = blocks[blockIdx.y];
    const int nn = d.tokens * d.tokens;
    const int64_t* index = reinterpret_cast<const int64_t*>(d.index);
    const float* t = t_all + (size_t)blockIdx.y * t_stride;
    float* bias = reinterpret_cast<float*>(d.bias);
    for (int p = blockIdx.x * 256 + threadIdx.x; p < d.heads * nn; p += gridDim.x * 256) {
        const int h = p / nn, ij = p - h * nn;
        bias[p] = 16.f / (1.f + expf(-t[(int)index[ij] * d.heads + h]));
    }
    if (blockIdx.x == 0) {
        const float* ls = reinterpret_cast<const float*>(d.logit_scale);
        float* scale = reinterpret_cast<float*>(d.scale);
        for (int h = threadIdx.x; h < d.heads; h += 256) scale[h] = expf(fminf(ls[h], 4.605170185988092f));
    }
}

// ---- backward, kernel 1: dt[e][h] = sum over the (i, j) with index[i][j] == e of dbias[h][i][j] * 16 s (1 - s), s = bias / 16.
//      One workgroup per (block, head); thread e walks all positions in order (deterministic, no atomics); the index table
//      sits in LDS.  Also d logit_scale.
__global__ __launch_bounds__(256) void cpb_dt_kernel(const frhip_cpb_block* __restrict__ blocks, float* __restrict__ dt_all, int t_stride) {
    constexpr int NNP = (49 * 49 + 15) / 16 * 16;                  // positions padded to whole 16-element vectors (zero gradient)
    __shared__ __attribute__((aligned(16))) unsigned char idx_lds[NNP];
    __shared__ __attribute__((aligned(16))) float g_lds[NNP];
    const frhip_cpb_block& d = blocks[blockIdx.y];
    const int h = blockIdx.x;
    if (h >= d.heads) return;
    const int nn = d.tokens * d.tokens;
    const int64_t* index = reinterpret_cast<const int64_t*>(d.index);
    const float* bias = reinterpret_cast<const float*>(d.bias) + (size_t)h * nn;
    const float* dbias = reinterpret_cast<const float*>(d.dbias) + (size_t)h * nn;
    const int nnp = (nn + 15) & ~15;
    for (int p = threadIdx.x; p < nnp; p += 256) {
        if (p < nn) {
            idx_lds[p] = (unsigned char)index[p];
            const float s = bias[p] * (1.f / 16.f);
            g_lds[p] = dbias[p] * 16.f * s * (1.f - s);
        } else {
            idx_lds[p] = 0; g_lds[p] = 0.f;
        }
    }
    __syncthreads();
    // every thread reads the same LDS address (broadcast): 16 positions per trip from one 16-byte index read and four 16-byte
    // gradient reads, four independent partial sums (a scalar walk here was latency-bound: 160 us for Swin34's ten blocks)
    for (int e = threadIdx.x; e < d.entries; e += 256) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int p = 0; p < nnp; p += 16) {
            const u32x4_t iv = *reinterpret_cast<const u32x4_t*>(idx_lds + p);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4_t gv = *reinterpret_cast<const f32x4_t*>(g_lds + p + 4 * q);
                const uint32_t w = iv[q];
                a0 += (int)(w & 0xff) == e ? gv[0] : 0.f;
                a1 += (int)((w >> 8) & 0xff) == e ? gv[1] : 0.f;
                a2 += (int)((w >> 16) & 0xff) == e ? gv[2] : 0.f;
                a3 += (int)(w >> 24) == e ? gv[3] : 0.f;
            }
        }
        dt_all[(size_t)blockIdx.y * t_stride + e * d.heads + h] = (a0 + a1) + (a2 + a3);
    }
    if (threadIdx.x == 0) {
        const float ls = reinterpret_cast<const float*>(d.logit_scale)[h];
        float* dls = reinterpret_cast<float*>(d.dlogit_scale);
        dls[h] += ls <= 4.605170185988092f ? reinterpret_cast<const float*>(d.dscale)[h] * reinterpret_cast<const float*>(d.scale)[h] : 0.f;
    }
}

// ---- backward, kernel 2: parameter gradients, ADDED into the (caller-zeroed) gradient tensors.  One workgroup per (block,
//      64 hidden units): thread (j, el) walks the table entries e = el, el + 4, ... with hid[e][j] recomputed, the four
//      entry-lanes are folded through LDS in a fixed order.
__global__ __launch_bounds__(256) void cpb_param_kernel(const frhip_cpb_block* __restrict__ blocks, const float* __restrict__ dt_all,
                                                        int t_stride) {
    __shared__ float red[4][64][CPB_MAX_HEADS + 3];
    __shared__ float dt_lds[CPB_MAX_T * CPB_MAX_HEADS];            // the block's dt table: broadcast LDS reads in the entry loop
    const frhip_cpb_block& d = blocks[blockIdx.y];
    const int jl = threadIdx.x & 63, el = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + jl;
    const float* tab = reinterpret_cast<const float*>(d.coords);
    const float* dt = dt_all + (size_t)blockIdx.y * t_stride;
    const float* w2 = reinterpret_cast<const float*>(d.w2);
    for (int i = threadIdx.x; i < d.entries * d.heads; i += 256) dt_lds[i] = dt[i];
    const float wa = reinterpret_cast<const float*>(d.w0)[2 * j], wb = reinterpret_cast<const float*>(d.w0)[2 * j + 1];
    const float bb = reinterpret_cast<const float*>(d.b0)[j];
    float gw2[CPB_MAX_HEADS], w2r[CPB_MAX_HEADS];
#pragma unroll
    for (int h = 0; h < CPB_MAX_HEADS; ++h) { gw2[h] = 0.f; w2r[h] = h < d.heads ? w2[h * CPB_HIDDEN + j] : 0.f; }
    __syncthreads();
    float g0 = 0.f, g1 = 0.f, gb = 0.f;
    for (int e = el; e < d.entries; e += 4) {
        const float c0 = tab[2 * e], c1 = tab[2 * e + 1];
        const float pre = fmaf(c0, wa, fmaf(c1, wb, bb));
        const float hid = fmaxf(pre, 0.f);
        float dh = 0.f;
#pragma unroll
        for (int h = 0; h < CPB_MAX_HEADS; ++h) {
            if (h < d.heads) {
                const float dth = dt_lds[e * d.heads + h];
                gw2[h] = fmaf(dth, hid, gw2[h]);
                dh = fmaf(dth, w2r[h], dh);
            }
        }
        if (pre > 0.f) { g0 = fmaf(dh, c0, g0); g1 = fmaf(dh, c1, g1); gb += dh; }
    }
#pragma unroll
    for (int h = 0; h < CPB_MAX_HEADS; ++h) red[el][jl][h] = gw2[h];
    red[el][jl][CPB_MAX_HEADS] = g0; red[el][jl][CPB_MAX_HEADS + 1] = g1; red[el][jl][CPB_MAX_HEADS + 2] = gb;
    __syncthreads();
    if (el == 0) {
        float* dw0 = reinterpret_cast<float*>(d.dw0);
        float* db0 = reinterpret_cast<float*>(d.db0);
        float* dw2 = reinterpret_cast<float*>(d.dw2);
        for (int h = 0; h < d.heads; ++h)
            dw2[h * CPB_HIDDEN + j] += (red[0][jl][h] + red[1][jl][h]) + (red[2][jl][h] + red[3][jl][h]);
        const int a = CPB_MAX_HEADS;
        dw0[2 * j] += (red[0][jl][a] + red[1][jl][a]) + (red[2][jl][a] + red[3][jl][a]);
        dw0[2 * j + 1] += (red[0][jl][a + 1] + red[1][jl][a + 1]) + (red[2][jl][a + 1] + red[3][jl][a + 1]);
        db0[j] += (red[0][jl][a + 2] + red[1][jl][a + 2]) + (red[2][jl][a + 2] + red[3][jl][a + 2]);
    }
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

extern "C" int frhip_cpb_scratch_floats(int nblocks) { return nblocks * CPB_MAX_T * CPB_MAX_HEADS; }

extern "C" int frhip_cpb_fwd(const frhip_cpb_block* blocks_dev, int nblocks, float* scratch, hipStream_t stream) {
    // scratch: frhip_cpb_scratch_floats(nblocks) floats (the MLP outputs per table entry and head)
    int rc = cpb_check(nblocks, "frhip_cpb_fwd");
    if (rc) return rc;
    if (!scratch) { set_error("frhip_cpb_fwd: scratch required"); return FRHIP_EINVAL; }
    const int ts = CPB_MAX_T * CPB_MAX_HEADS;
    hipLaunchKernelGGL(cpb_table_kernel, dim3(CPB_MAX_T, nblocks), dim3(128), 0, stream, blocks_dev, scratch, ts);
    hipLaunchKernelGGL(cpb_gather_kernel, dim3(16, nblocks), dim3(256), 0, stream, blocks_dev, (const float*)scratch, ts);
    return check_launch("frhip_cpb_fwd");
}

extern "C" int frhip_cpb_bwd(const frhip_cpb_block* blocks_dev, int nblocks, float* scratch, hipStream_t stream) {
    int rc = cpb_check(nblocks, "frhip_cpb_bwd");
    if (rc) return rc;
    if (!scratch) { set_error("frhip_cpb_bwd: scratch required"); return FRHIP_EINVAL; }
    const int ts = CPB_MAX_T * CPB_MAX_HEADS;
    hipLaunchKernelGGL(cpb_dt_kernel, dim3(CPB_MAX_HEADS, nblocks), dim3(256), 0, stream, blocks_dev, scratch, ts);
    hipLaunchKernelGGL(cpb_param_kernel, dim3(CPB_HIDDEN / 64, nblocks), dim3(256), 0, stream, blocks_dev, (const float*)scratch, ts);
    return check_launch("frhip_cpb_bwd");
}
