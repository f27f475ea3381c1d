// Swin / AlterNet MLP (reference nets/SwinV2.py:16-32: fc1 -> GELU -> fc2) without the hidden PRE-activation in HBM.
//
// The unfused pair keeps two [tokens][4C] tensors per block: hid = fc1(x) + b1 (needed by gelu' in the backward pass) and act = gelu(hid)
// (fc2's operand, also fc2's weight-gradient operand).  These linears have K = C = 64 ... 512: they are HBM-bound, and at the first
// stage (1.6 M tokens, C = 64) each of the two tensors is 822 MB.  hid costs one K = C GEMM to recompute, so
//   * frhip_linear_fwd_act       writes act only;
//   * frhip_linear_dgrad_gelu_rc forms fc2's data-gradient tile dg = dy W2 AND recomputes the hid tile of the same rows and columns from x
//                                (two GEMMs of identical geometry through the same LDS stages), then dh = dg * gelu'(hid) in registers.
// Arithmetic is that of the unfused kernels, rounding included -- hid is rounded to T where frhip_linear_fwd stores it, dg where
// frhip_linear_dgrad_gelu stages it -- so act, dh and the column sums of dh (= fc1.bias's gradient) are bit-identical.
// Traffic per block at stage 1: fc1 forward 1.85 -> 1.03 GB, fc2 data-gradient 1.85 -> 1.23 GB.
#include "igemm_nt.h"
#include "frhip.h"

namespace frhip {

static int mlp_geom(NtGeom& g, int dtype, int m, int n, int k, const char* who) {
    if (dtype != FRHIP_DT_BF16) { set_error("%s: bf16 only", who); return FRHIP_EINVAL; }
    if (m <= 0 || n <= 0 || k <= 0 || (k % 64) || (n % 8)) { set_error("%s: unsupported shape m=%d n=%d k=%d", who, m, n, k); return FRHIP_EINVAL; }
    if (1LL * m * k * 2 > 0x7fffffffLL || 1LL * n * k * 2 > 0x7fffffffLL) { set_error("%s: operand exceeds the 2 GiB buffer window", who); return FRHIP_EINVAL; }
    g.H = 1; g.W = 1; g.C = k; g.Ho = 1; g.Wo = 1; g.R = 1; g.S = 1; g.stride = 1; g.pad = 0; g.mode = 0;
    g.M = m; g.Nout = n; g.Ktot = k;
    g.ksteps = k / 64; g.ksteps_per_split = g.ksteps;
    g.a_bytes = (uint32_t)(1LL * m * k * 2); g.b_bytes = (uint32_t)(1LL * n * k * 2);
    g.par_a = -1; g.par_b = 0; g.hc = 0; g.wc = 0; g.par_r0 = 0; g.par_s0 = 0;
    return 0;
}

// the wave's 64 x 64 tile from its LDS staging area (bf16 rows of pitch P): rows m0 .., channels n0 ..; 8 rows per instruction
template <typename F>
__device__ __forceinline__ void mlp_rows(const char* mine, int P, int M, int Nout, int m0, int n0, F&& body) {
    const int lane = lane_id();
    const int chunk = lane & 7, rsub = lane >> 3;
    const int n = n0 + chunk * 8;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int row = it * 8 + rsub, m = m0 + row;
        Vec16<bf16_t> v = *reinterpret_cast<const Vec16<bf16_t>*>(mine + row * P + chunk * 16);
        if (m < M && n < Nout) body(m, n, v);
    }
}

// act[m][n] = gelu(round(round(a w^T) + bias)): frhip_linear_fwd's act output without its `out`
__global__ __launch_bounds__(256, 2) void mlp_fwd_act_kernel(NtGeom g, const void* __restrict__ a, const void* __restrict__ w,
                                                            const float* __restrict__ bias, bf16_t* __restrict__ act, int mtiles, int ntiles) {
    typedef NtTile<bf16_t, 2, 2, 4> Tile;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lin = xcd_remap(blockIdx.x, gridDim.x);
    const int ntile = (int)(lin % (uint32_t)ntiles), mtile = (int)(lin / (uint32_t)ntiles);
    NtMainloop<bf16_t, 2, 2, 4> ml;
    ml.run(g, a, w, smem, mtile, ntile, 0, g.ksteps);
    const int wave = wave_id();
    const int m0 = mtile * Tile::BM + (wave >> 1) * 64, n0 = ntile * Tile::BN + (wave & 1) * 64;
    const char* mine = ml.template stage_out<bf16_t>(smem);
    mlp_rows(mine, Tile::template stage_pitch<bf16_t>(), g.M, g.Nout, m0, n0, [&](int m, int n, Vec16<bf16_t>& v) {
        Vec16<bf16_t> ga;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v.set(e, v.get(e) + bias[n + e]);                 // the pre-activation as frhip_linear_fwd stores it
            const float hr = v.get(e);
            float cdf, pdf;
            gelu_parts(hr, cdf, pdf);
            ga.set(e, hr * cdf);
        }
        *reinterpret_cast<Vec16<bf16_t>*>(act + (size_t)m * g.Nout + n) = ga;
    });
}

// dx[m][n] = round(dy wt^T) * gelu'(hid[m][n]), hid = round(round(x w1^T) + bias1) recomputed; stats[tile][0][n] = column sums of dx
__global__ __launch_bounds__(256, 2) void mlp_dgrad_rc_kernel(NtGeom g, const void* __restrict__ dy, const void* __restrict__ wt,
                                                             const void* __restrict__ x, const void* __restrict__ w1,
                                                             const float* __restrict__ bias1, bf16_t* __restrict__ dx,
                                                             float* __restrict__ stats, int mtiles, int ntiles) {
    typedef NtTile<bf16_t, 2, 2, 4> Tile;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lin = xcd_remap(blockIdx.x, gridDim.x);
    const int ntile = (int)(lin % (uint32_t)ntiles), mtile = (int)(lin / (uint32_t)ntiles);
    const int lane = lane_id(), wave = wave_id();
    const int fg = lane >> 4;
    const int m0 = mtile * Tile::BM + (wave >> 1) * 64, n0 = ntile * Tile::BN + (wave & 1) * 64;
    NtMainloop<bf16_t, 2, 2, 4> ml;
    // ---- hid tile: acc[nt][mt][e] = channel n0 + 16 nt + 4 fg + e of pixel mt * 16 + (lane & 15)
    ml.run(g, x, w1, smem, mtile, ntile, 0, g.ksteps);
    f32x4_t fac[4][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        float bb[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int c = n0 + nt * 16 + 4 * fg + e; bb[e] = c < g.Nout ? bias1[c] : 0.f; }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float h = (float)(bf16_t)((float)(bf16_t)ml.acc[nt][mt][e] + bb[e]);      // the stored pre-activation, bit for bit
                float cdf, pdf;
                gelu_parts(h, cdf, pdf);
                fac[nt][mt][e] = cdf + h * pdf;
            }
    }
    __syncthreads();                       // every wave is done reading the operand stages of the first GEMM
    // ---- data-gradient tile of fc2 over the same rows and columns
    ml.run(g, dy, wt, smem, mtile, ntile, 0, g.ksteps);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e) ml.acc[nt][mt][e] = (float)(bf16_t)ml.acc[nt][mt][e] * fac[nt][mt][e];
    const char* mine = ml.template stage_out<bf16_t>(smem);
    float s1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = 0.f;
    mlp_rows(mine, Tile::template stage_pitch<bf16_t>(), g.M, g.Nout, m0, n0, [&](int m, int n, Vec16<bf16_t>& v) {
        *reinterpret_cast<Vec16<bf16_t>*>(dx + (size_t)m * g.Nout + n) = v;
#pragma unroll
        for (int e = 0; e < 8; ++e) s1[e] += v.get(e);
    });
    if (stats) {
        // column sums per 128-row tile, the layout frhip_linear_dgrad_gelu writes ([tile][2][n], second row unused): lanes that share a
        // 16-byte chunk (bits 3..5 of the lane), then the two waves that share the channel half
#pragma unroll
        for (int e = 0; e < 8; ++e) s1[e] = lane_sum_bit5(lane_sum_bit4(lane_sum_bit3(s1[e])));
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);          // [wave][64]
        if (lane < 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) red[wave * 64 + lane * 8 + e] = s1[e];
        }
        __syncthreads();
        if (threadIdx.x < 128) {
            const int c = threadIdx.x & 63, wn = threadIdx.x >> 6;
            const int nn = ntile * Tile::BN + wn * 64 + c;
            if (nn < g.Nout) {
                stats[((size_t)mtile * 2 + 0) * g.Nout + nn] = red[(0 * 2 + wn) * 64 + c] + red[(1 * 2 + wn) * 64 + c];
                stats[((size_t)mtile * 2 + 1) * g.Nout + nn] = 0.f;
            }
        }
    }
}

template <typename K> static int mlp_attr(K kern, int lds) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
        set_error("mlp_recompute: cannot raise dynamic LDS to %d bytes", lds);
        return FRHIP_ELAUNCH;
    }
    return 0;
}

}  // namespace frhip

using namespace frhip;

extern "C" int frhip_mlp_stat_rows(int m) { return (m + 127) / 128; }

extern "C" int frhip_linear_fwd_act(int dtype, const void* a, const void* w, const float* bias, void* act_out, int m, int n, int k,
                                    hipStream_t stream) {
    if (!bias || !act_out) { set_error("frhip_linear_fwd_act: bias and act_out are required"); return FRHIP_EINVAL; }
    NtGeom g;
    int rc = mlp_geom(g, dtype, m, n, k, "frhip_linear_fwd_act");
    if (rc) return rc;
    typedef NtTile<bf16_t, 2, 2, 4> Tile;
    const int lds = Tile::lds_bytes<bf16_t>();
    static bool done = false;
    if (!done) { if ((rc = mlp_attr(mlp_fwd_act_kernel, lds))) return rc; done = true; }
    const int mtiles = (m + Tile::BM - 1) / Tile::BM, ntiles = (n + Tile::BN - 1) / Tile::BN;
    hipLaunchKernelGGL(mlp_fwd_act_kernel, dim3(mtiles * ntiles), dim3(256), lds, stream, g, a, w, bias, (bf16_t*)act_out, mtiles, ntiles);
    return check_launch("frhip_linear_fwd_act");
}

extern "C" int frhip_linear_dgrad_gelu_rc(int dtype, const void* dy, const void* wt, const void* x, const void* w1, const float* bias1,
                                          void* dx, float* stats_partial, int m, int n, int k, hipStream_t stream) {
    if (!x || !w1 || !bias1) { set_error("frhip_linear_dgrad_gelu_rc: x, w1 and bias1 are required"); return FRHIP_EINVAL; }
    NtGeom g;
    int rc = mlp_geom(g, dtype, m, n, k, "frhip_linear_dgrad_gelu_rc");
    if (rc) return rc;
    typedef NtTile<bf16_t, 2, 2, 4> Tile;
    const int lds = Tile::lds_bytes<bf16_t>();
    static bool done = false;
    if (!done) { if ((rc = mlp_attr(mlp_dgrad_rc_kernel, lds))) return rc; done = true; }
    const int mtiles = (m + Tile::BM - 1) / Tile::BM, ntiles = (n + Tile::BN - 1) / Tile::BN;
    hipLaunchKernelGGL(mlp_dgrad_rc_kernel, dim3(mtiles * ntiles), dim3(256), lds, stream, g, dy, wt, x, w1, bias1, (bf16_t*)dx,
                       stats_partial, mtiles, ntiles);
    return check_launch("frhip_linear_dgrad_gelu_rc");
}
