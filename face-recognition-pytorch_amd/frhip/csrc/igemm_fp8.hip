// fp8 weight path (BASELINE cfg 5: "hybrid backbone ..., fp8 MFMA weight path"), gfx950.
//   Forward convolutions / linears with BOTH MFMA operands in OCP fp8 (e4m3fn) on the block-scaled matrix instruction
//   v_mfma_scale_f32_16x16x128_f8f6f4 (2x the bf16 rate, half the operand bytes through L2 -> LDS), fp32 accumulation.
//   Weights: fp8 pack [K][R][S][C] + one fp32 scale per output channel (amax / 448), quantised once per step from the fp32
//   master weights.  Activations: fp8 copy with one static scale per tensor, written by the producing BatchNorm-apply pass
//   (frhip_bn_apply_q8) or by frhip_quant_fp8; post-BatchNorm activations are O(1), far inside e4m3's range (448).
//   The accumulators are multiplied by act_scale * w_scale[k] and stored as bf16; the BatchNorm partial sums come out of the
//   same store epilogue as in the bf16 kernels.  Backward passes stay on the bf16 kernels (they read the bf16 tensors).
// The reference has no fp8 arithmetic: parity is stated against the bf16 path of this library.
// Reference call sites served: nets/AlterNet_SwinV2_FAN.py:520-568 (BasicBlock convs), :263-302 (qkv / proj), nets/resnet.py:23-46.
#include "igemm_halo.h"
#include "frhip.h"

namespace frhip {
extern int g_epi_lean;      // igemm_halo.hip: frhip_set_epi_lean / FRHIP_EPI_LEAN

constexpr float FP8_MAX = 448.f;

__device__ __forceinline__ uint32_t pack4_fp8(float a, float b, float c, float d) {
    int v = 0;
    v = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(a, -FP8_MAX), FP8_MAX), fminf(fmaxf(b, -FP8_MAX), FP8_MAX), v, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(c, -FP8_MAX), FP8_MAX), fminf(fmaxf(d, -FP8_MAX), FP8_MAX), v, true);
    return (uint32_t)v;
}

// one workgroup per output channel: scale[k] = amax(w[k][:]) / 448 (1 when the row is all zero), w8 = round(w / scale)
__global__ __launch_bounds__(256) void quant_w8_kernel(const float* __restrict__ w, uint8_t* __restrict__ w8,
                                                       float* __restrict__ scale, int rowlen) {
    __shared__ float red[256];
    const float* row = w + (size_t)blockIdx.x * rowlen;
    float amax = 0.f;
    for (int i = threadIdx.x; i < rowlen; i += 256) amax = fmaxf(amax, fabsf(row[i]));
    red[threadIdx.x] = amax;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) { if ((int)threadIdx.x < d) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + d]); __syncthreads(); }
    amax = red[0];
    const float sc = amax > 0.f ? amax / FP8_MAX : 1.f, inv = 1.f / sc;
    if (threadIdx.x == 0) scale[blockIdx.x] = sc;
    uint32_t* o = reinterpret_cast<uint32_t*>(w8 + (size_t)blockIdx.x * rowlen);
    for (int i = threadIdx.x; i < rowlen / 4; i += 256) {
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(row + 4 * i);
        o[i] = pack4_fp8(v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv);
    }
}

// the same for every weight tensor of a step in one launch: one workgroup per output channel of the whole list, the tensor found by
// a scan of the (short) table of first rows.  44 launches of 4.5 us each became one in the AlterNet50 fp8 step.
__global__ __launch_bounds__(256) void quant_w8_multi_kernel(const frhip_q8w* __restrict__ table, int ntensors) {
    __shared__ float red[256];
    int t = 0;
    while (t + 1 < ntensors && (int)blockIdx.x >= table[t + 1].row_begin) ++t;
    const frhip_q8w d = table[t];
    const int r = (int)blockIdx.x - d.row_begin;
    const float* row = d.w + (size_t)r * d.rowlen;
    float amax = 0.f;
    for (int i = threadIdx.x; i < d.rowlen; i += 256) amax = fmaxf(amax, fabsf(row[i]));
    red[threadIdx.x] = amax;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]); __syncthreads(); }
    amax = red[0];
    const float sc = amax > 0.f ? amax / FP8_MAX : 1.f, inv = 1.f / sc;
    if (threadIdx.x == 0) d.scale[r] = sc;
    uint32_t* o = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(d.w8) + (size_t)r * d.rowlen);
    for (int i = threadIdx.x; i < d.rowlen / 4; i += 256) {
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(row + 4 * i);
        o[i] = pack4_fp8(v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv);
    }
}

// Debug counter of saturated activations: pack4_fp8 clamps at +-448 without a trace, and the activation scale is a static
// 1.0.  frhip_fp8_saturation(1) arms it: the activation quantisers then count the elements whose magnitude exceeds e4m3's range before
// the clamp (one atomic per thread that saw any); production runs pass a null pointer and pay nothing.
__device__ unsigned int g_fp8_sat_count;
static unsigned int* g_fp8_sat_ptr = nullptr;
template <int N> __device__ __forceinline__ void fp8_count_sat(unsigned int* sat, const float (&f)[N]) {
    if (!sat) return;
    int n = 0;
#pragma unroll
    for (int e = 0; e < N; ++e) n += fabsf(f[e]) > FP8_MAX ? 1 : 0;
    if (n) atomicAdd(sat, (unsigned int)n);
}

// x8 = fp8(x * inv_scale), 16 elements per thread
template <typename T>
__global__ __launch_bounds__(256) void quant_act8_kernel(const T* __restrict__ x, uint8_t* __restrict__ x8, size_t n16, float inv_scale,
                                                         unsigned int* __restrict__ sat) {
    constexpr int EPV = 16 / (int)sizeof(T), NV = 16 / EPV;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        float f[16];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const Vec16<T> a = *reinterpret_cast<const Vec16<T>*>(x + i * 16 + v * EPV);
#pragma unroll
            for (int e = 0; e < EPV; ++e) f[v * EPV + e] = a.get(e) * inv_scale;
        }
        fp8_count_sat(sat, f);
        u32x4_t o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = pack4_fp8(f[4 * q], f[4 * q + 1], f[4 * q + 2], f[4 * q + 3]);
        *reinterpret_cast<u32x4_t*>(x8 + i * 16) = o;
    }
}

// out = act(y * scale + shift [+ res]) as T AND as fp8(out * inv_q) -- the BatchNorm-apply pass that feeds an fp8 GEMM
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_q8_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const T* __restrict__ res,
                                                          const float* __restrict__ rscale, const float* __restrict__ rshift,
                                                          int relu, T* __restrict__ out, uint8_t* __restrict__ out8, float inv_q,
                                                          int rows, int C, unsigned int* __restrict__ sat) {
    constexpr int EPV = 16 / (int)sizeof(T);
    const int vpr = C / EPV;
    const int cg = threadIdx.x % vpr, rl = threadIdx.x / vpr, rlanes = 256 / vpr;
    float sc[EPV], sh[EPV], rs[EPV], rb[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
        sc[e] = scale[cg * EPV + e]; sh[e] = shift[cg * EPV + e];
        rs[e] = rscale ? rscale[cg * EPV + e] : 1.f; rb[e] = rscale ? rshift[cg * EPV + e] : 0.f;
    }
    for (int r = blockIdx.x * rlanes + rl; r < rows; r += gridDim.x * rlanes) {
        const size_t idx = (size_t)r * C + cg * EPV;
        Vec16<T> a = *reinterpret_cast<const Vec16<T>*>(y + idx);
        Vec16<T> rv;
        if (res) rv = *reinterpret_cast<const Vec16<T>*>(res + idx);
        float f[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
            float o = a.get(e) * sc[e] + sh[e];
            if (res) o += rv.get(e) * rs[e] + rb[e];
            o = relu ? fmaxf(o, 0.f) : o;
            a.set(e, o);
            f[e] = a.get(e) * inv_q;                       // quantise the value as STORED in T
        }
        *reinterpret_cast<Vec16<T>*>(out + idx) = a;
        fp8_count_sat(sat, f);
        uint32_t* o8 = reinterpret_cast<uint32_t*>(out8 + idx);
#pragma unroll
        for (int q = 0; q < EPV / 4; ++q) o8[q] = pack4_fp8(f[4 * q], f[4 * q + 1], f[4 * q + 2], f[4 * q + 3]);
    }
}

// ---- fp8 x fp8 -> bf16 implicit GEMM on the shared NT main loop (one K step = one filter tap x 128 channels)
// LEAN: whole 256-row / 256-channel tiles and a dense layout (nt_lean_ok): the lean store epilogue of igemm_nt.h
template <int WM, int WN, int MT, bool LEAN = false>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN) / 4) void nt8_kernel(NtGeom g, const void* __restrict__ a,
                                                                         const void* __restrict__ b, const float* __restrict__ wscale,
                                                                         float ascale, void* __restrict__ out, float* __restrict__ stats,
                                                                         EpiBnRed br, int mtiles, int ntiles) {
    typedef NtTile<fp8_t, WM, WN, MT> Tile;
    constexpr int WROWS = Tile::WROWS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lin = xcd_remap(blockIdx.x, gridDim.x);
    const int ntile = (int)(lin % (uint32_t)ntiles), mtile = (int)(lin / (uint32_t)ntiles);
    NtMainloop<fp8_t, WM, WN, MT> ml;
    ml.run(g, a, b, smem, mtile, ntile, 0, g.ksteps);
    const int lane = lane_id(), wave = wave_id();
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = mtile * Tile::BM + wm * WROWS, n0 = ntile * Tile::BN + wn * 64;
    // D rows = output channels nt*16 + 4*(lane>>4) + e: fold act_scale * w_scale[channel] into the accumulators
    const int fg = lane >> 4;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        float sc[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int n = n0 + nt * 16 + 4 * fg + e; sc[e] = n < g.Nout ? ascale * wscale[n] : 0.f; }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e) ml.acc[nt][mt][e] *= sc[e];
    }
    const char* mine = ml.template stage_out<bf16_t>(smem);
    if constexpr (LEAN) {
        nt_epilogue_store_lean<WM, WN, WROWS, Tile::THREADS, Tile::BN>(mine, Tile::template stage_pitch<bf16_t>(), smem, g.M, g.Nout, out,
                                                                       nullptr, stats, br, mtile, ntile, m0, n0);
    } else {
        EpiOperands<bf16_t, WROWS> eo;
        eo.fetch(nullptr, nullptr, g.M, g.Nout, m0, n0, 0, 0, &br.map);
        nt_epilogue_store<bf16_t, WM, WN, WROWS, Tile::THREADS, Tile::BN>(mine, Tile::template stage_pitch<bf16_t>(), smem, g.M, g.Nout,
                                                                          out, false, stats, br, eo, mtile, ntile, m0, n0);
    }
}

template <int WM, int WN, int MT, bool LEAN = false>
static int nt8_launch(const NtGeom& g, const void* a, const void* b, const float* wscale, float ascale, void* out, float* stats,
                      const EpiBnRed& br, hipStream_t stream) {
    typedef NtTile<fp8_t, WM, WN, MT> Tile;
    const int mtiles = (g.M + Tile::BM - 1) / Tile::BM, ntiles = (g.Nout + Tile::BN - 1) / Tile::BN;
    const int lds = Tile::template lds_bytes<bf16_t>();
    auto kern = nt8_kernel<WM, WN, MT, LEAN>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            set_error("igemm_fp8: cannot raise dynamic LDS to %d bytes", lds);
            return FRHIP_ELAUNCH;
        }
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(mtiles * ntiles), dim3(Tile::THREADS), lds, stream, g, a, b, wscale, ascale, out, stats, br, mtiles, ntiles);
    return check_launch("igemm_fp8");
}

// ---- 3x3 / stride-1 / pad-1 fp8 convolution on the LDS-halo main loop (igemm_halo.h): activation window resident in LDS per
//      128-channel chunk, weights streamed per tap; non-scaled fp8 MFMA (see Mma<fp8n_t>): half the L2 -> LDS bytes per MAC of the
//      bf16 kernel, which is what bounds it.
template <int WM, int WN, int MT, int HBUFS, bool LEAN = false>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN) / 4) void halo8_kernel(HaloGeom g, const void* __restrict__ a,
                                                                           const void* __restrict__ b, const float* __restrict__ wscale,
                                                                           float ascale, void* __restrict__ out, float* __restrict__ stats,
                                                                           EpiBnRed br, int mtiles, int ntiles) {
    typedef HaloTile<fp8n_t, WM, WN, MT, HBUFS> Tile;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lin = xcd_remap(blockIdx.x, gridDim.x);
    const int ntile = (int)(lin % (uint32_t)ntiles), mtile = (int)(lin / (uint32_t)ntiles);
    HaloMainloop<fp8n_t, WM, WN, MT, HBUFS> ml;
    ml.run(g, a, b, smem, mtile, ntile);
    const int lane = lane_id(), wave = wave_id();
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = mtile * Tile::BM + wm * Tile::WROWS, n0 = ntile * Tile::BN + wn * 64;
    const int fg = lane >> 4;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        float sc[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int n = n0 + nt * 16 + 4 * fg + e; sc[e] = n < g.Nout ? ascale * wscale[n] : 0.f; }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e) ml.acc[nt][mt][e] *= sc[e];
    }
    EpiOperands<bf16_t, Tile::WROWS> eo;
    const char* mine = ml.template stage_out<bf16_t>(smem);
    if constexpr (LEAN) {
        eo.fetch_fast(nullptr, nullptr, g.M, g.Nout, m0, n0);
        nt_epilogue_store_fast<bf16_t, WM, WN, Tile::WROWS, Tile::THREADS, Tile::BN>(mine, Tile::template stage_pitch<bf16_t>(), smem, g.M,
                                                                                     g.Nout, out, false, stats, br, eo, mtile, ntile);
    } else {
        eo.fetch(nullptr, nullptr, g.M, g.Nout, m0, n0, 0, 0, &br.map);
        nt_epilogue_store<bf16_t, WM, WN, Tile::WROWS, Tile::THREADS, Tile::BN>(mine, Tile::template stage_pitch<bf16_t>(), smem, g.M, g.Nout,
                                                                                out, false, stats, br, eo, mtile, ntile, m0, n0);
    }
}

static int g_fp8_halo = 1;      // 3x3 / stride-1 layers on the halo main loop (0: generic NT kernel; test hook frhip_set_fp8_halo)

static bool halo8_applicable(int h, int w, int c, int k, int r, int s, int stride, int pad) {
    return r == 3 && s == 3 && stride == 1 && pad == 1 && w <= 56 && (c % 128) == 0 && (k % 8) == 0;
}

static int halo8_run(const void* x8, const void* w8, const float* wscale, float ascale, void* y, float* stats, int n, int h, int w,
                     int c, int k, hipStream_t stream) {
    typedef HaloTile<fp8n_t, 4, 1, 4, 1> Tile;
    const long long ab = 1LL * n * h * w * c, bb = 1LL * k * 9 * c;
    if (ab > 0x7fffffffLL || bb > 0x7fffffffLL) { set_error("igemm_fp8(halo): tensor exceeds the 2 GiB buffer window"); return FRHIP_EINVAL; }
    HaloGeom g;
    g.H = h; g.W = w; g.C = c; g.M = n * h * w; g.Nout = k; g.Ktot = 9 * c; g.sign = +1;
    g.a_bytes = (uint32_t)ab; g.b_bytes = (uint32_t)bb;
    g.xf_scale = nullptr; g.xf_shift = nullptr; g.xf_out = nullptr;
    g.d_hw = make_fastdiv((uint32_t)(h * w)); g.d_w = make_fastdiv((uint32_t)w);
    const int mtiles = (g.M + Tile::BM - 1) / Tile::BM, ntiles = (k + Tile::BN - 1) / Tile::BN;
    const int lds = Tile::template lds_bytes<bf16_t>();
    static const EpiBnRed none = {nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, {0, 0, 0, 0, 0, 0}, nullptr, nullptr, 0};
    const bool lean = g_epi_lean && epi_lean_ok(true, g.M, k, Tile::BM, Tile::BN, none);
    auto kern = lean ? halo8_kernel<4, 1, 4, 1, true> : halo8_kernel<4, 1, 4, 1, false>;
    static bool attr_done[2] = {false, false};
    if (!attr_done[lean]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            set_error("igemm_fp8(halo): cannot raise dynamic LDS to %d bytes", lds);
            return FRHIP_ELAUNCH;
        }
        attr_done[lean] = true;
    }
    hipLaunchKernelGGL(kern, dim3(mtiles * ntiles), dim3(Tile::THREADS), lds, stream, g, x8, w8, wscale, ascale, y, stats, none, mtiles, ntiles);
    return check_launch("igemm_fp8(halo)");
}

static bool fp8_wide(const NtGeom& g) { return (g.Nout % 256) == 0 && g.M >= 256 * 64; }

static int fp8_geom(NtGeom& g, int n, int h, int w, int c, int k, int r, int s, int stride, int pad, const char* who) {
    if (n <= 0 || h <= 0 || w <= 0 || c <= 0 || k <= 0 || (c % 128) != 0 || (k % 8) != 0 || (stride != 1 && stride != 2)) {
        set_error("%s: unsupported shape n=%d h=%d w=%d c=%d k=%d stride=%d (c must be a multiple of 128, k of 8)", who, n, h, w, c, k, stride);
        return FRHIP_EINVAL;
    }
    const int ho = (h + 2 * pad - r) / stride + 1, wo = (w + 2 * pad - s) / stride + 1;
    const long long a_bytes = 1LL * n * h * w * c, b_bytes = 1LL * k * r * s * c;
    if (a_bytes > 0x7fffffffLL || b_bytes > 0x7fffffffLL || 1LL * n * ho * wo > 0x7fffffffLL) {
        set_error("%s: tensor exceeds the 2 GiB buffer-addressing window", who);
        return FRHIP_EINVAL;
    }
    g.H = h; g.W = w; g.C = c; g.Ho = ho; g.Wo = wo; g.R = r; g.S = s; g.stride = stride; g.pad = pad; g.mode = 0;
    g.M = n * ho * wo; g.Nout = k; g.Ktot = r * s * c;
    g.ksteps = r * s * (c / 128); g.ksteps_per_split = g.ksteps;
    g.a_bytes = (uint32_t)a_bytes; g.b_bytes = (uint32_t)b_bytes;
    g.par_a = -1; g.par_b = -1; g.hc = 0; g.wc = 0; g.par_r0 = 0; g.par_s0 = 0;
    return FRHIP_OK;
}

static const EpiBnRed NO_EPI = {nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, {0, 0, 0, 0, 0, 0}, nullptr, nullptr, 0};

static int fp8_run(const NtGeom& g, const void* x8, const void* w8, const float* wscale, float ascale, void* y, float* stats,
                   const EpiBnRed& br, hipStream_t stream) {
    if (fp8_wide(g)) {
        if (g_epi_lean && nt_lean_ok(true, g.M, g.Nout, br, stats)) return nt8_launch<2, 4, 8, true>(g, x8, w8, wscale, ascale, y, stats, br, stream);
        return nt8_launch<2, 4, 8>(g, x8, w8, wscale, ascale, y, stats, br, stream);
    }
    if ((g.Nout % 128) != 0 && g.Nout <= 256) return nt8_launch<4, 1, 4>(g, x8, w8, wscale, ascale, y, stats, br, stream);
    return nt8_launch<2, 2, 4>(g, x8, w8, wscale, ascale, y, stats, br, stream);
}

}  // namespace frhip

using namespace frhip;

extern "C" int frhip_quant_fp8_weights(const float* w, void* w8, float* scale, int k, int rowlen, hipStream_t stream) {
    if (k <= 0 || rowlen <= 0 || (rowlen % 4)) { set_error("frhip_quant_fp8_weights: bad shape k=%d rowlen=%d", k, rowlen); return FRHIP_EINVAL; }
    hipLaunchKernelGGL(quant_w8_kernel, dim3(k), dim3(256), 0, stream, w, (uint8_t*)w8, scale, rowlen);
    return check_launch("frhip_quant_fp8_weights");
}

extern "C" int frhip_quant_fp8_weights_multi(const frhip_q8w* table, int ntensors, int nrows, hipStream_t stream) {
    if (!table || ntensors <= 0 || nrows <= 0) { set_error("frhip_quant_fp8_weights_multi: empty table"); return FRHIP_EINVAL; }
    hipLaunchKernelGGL(quant_w8_multi_kernel, dim3(nrows), dim3(256), 0, stream, table, ntensors);
    return check_launch("frhip_quant_fp8_weights_multi");
}

extern "C" int frhip_quant_fp8(int dtype, const void* x, void* x8, size_t n, float inv_scale, hipStream_t stream) {
    if (n % 16) { set_error("frhip_quant_fp8: n must be a multiple of 16"); return FRHIP_EINVAL; }
    const size_t n16 = n / 16;
    const unsigned grid = (unsigned)((n16 + 255) / 256 > 65536 ? 65536 : (n16 + 255) / 256);
    if (dtype == FRHIP_DT_BF16) hipLaunchKernelGGL(quant_act8_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream, (const bf16_t*)x, (uint8_t*)x8, n16, inv_scale, g_fp8_sat_ptr);
    else if (dtype == FRHIP_DT_F32) hipLaunchKernelGGL(quant_act8_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)x, (uint8_t*)x8, n16, inv_scale, g_fp8_sat_ptr);
    else { set_error("frhip_quant_fp8: bad dtype %d", dtype); return FRHIP_EINVAL; }
    return check_launch("frhip_quant_fp8");
}

extern "C" int frhip_bn_apply_q8(int dtype, const void* y, const float* scale, const float* shift, const void* res,
                                 const float* res_scale, const float* res_shift, int relu, void* out, void* out8, float inv_q,
                                 int rows, int c, hipStream_t stream) {
    const int epv = dtype == FRHIP_DT_BF16 ? 8 : 4;
    if (dtype != FRHIP_DT_BF16 || c % epv || c / epv > 256 || 256 % (c / epv) || !out8) {
        set_error("frhip_bn_apply_q8: bf16 only, c/8 must divide 256 (c=%d)", c);
        return FRHIP_EINVAL;
    }
    const int rlanes = 256 / (c / epv);
    int grid = (rows + rlanes * 16 - 1) / (rlanes * 16);
    grid = grid < 1 ? 1 : (grid > 16384 ? 16384 : grid);
    hipLaunchKernelGGL(bn_apply_q8_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream, (const bf16_t*)y, scale, shift, (const bf16_t*)res,
                       res_scale, res_shift, relu, (bf16_t*)out, (uint8_t*)out8, inv_q, rows, c, g_fp8_sat_ptr);
    return check_launch("frhip_bn_apply_q8");
}

extern "C" int frhip_fp8_saturation(int op) {
    // op 1: zero the counter and arm it; op 0: disarm; op 2: read it (device-synchronising copy).  Returns the count (op 2, capped at INT_MAX), 0, or < 0
    void* sym = nullptr;
    if (hipGetSymbolAddress(&sym, HIP_SYMBOL(frhip::g_fp8_sat_count)) != hipSuccess) { set_error("frhip_fp8_saturation: no counter symbol"); return FRHIP_ELAUNCH; }
    unsigned int v = 0;
    if (op == 1) {
        if (hipMemcpy(sym, &v, sizeof(v), hipMemcpyHostToDevice) != hipSuccess) return FRHIP_ELAUNCH;
        frhip::g_fp8_sat_ptr = reinterpret_cast<unsigned int*>(sym);
        return 0;
    }
    if (op == 0) { frhip::g_fp8_sat_ptr = nullptr; return 0; }
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&v, sym, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return FRHIP_ELAUNCH;
    return v > 0x7fffffffu ? 0x7fffffff : (int)v;
}

extern "C" int frhip_conv_fwd_fp8(const void* x8, const void* w8, const float* wscale, float act_scale, void* y,
                                  float* stats_partial, int n, int h, int wd, int c, int k, int r, int s, int stride, int pad,
                                  hipStream_t stream) {
    NtGeom g;
    int rc = fp8_geom(g, n, h, wd, c, k, r, s, stride, pad, "frhip_conv_fwd_fp8");
    if (rc) return rc;
    if (g_fp8_halo && halo8_applicable(h, wd, c, k, r, s, stride, pad))
        return halo8_run(x8, w8, wscale, act_scale, y, stats_partial, n, h, wd, c, k, stream);
    return fp8_run(g, x8, w8, wscale, act_scale, y, stats_partial, NO_EPI, stream);
}

extern "C" int frhip_set_fp8_halo(int enabled) { const int old = g_fp8_halo; if (enabled >= 0) g_fp8_halo = enabled; return old; }

extern "C" int frhip_fp8_conv_stat_rows(int m, int k, int h, int w, int c, int r, int s, int stride, int pad) {
    if (g_fp8_halo && halo8_applicable(h, w, c, k, r, s, stride, pad)) return (m + 255) / 256;
    return frhip_fp8_stat_rows(m, k);
}

extern "C" int frhip_fp8_stat_rows(int m, int k) {
    NtGeom g; g.M = m; g.Nout = k;
    const int bm = fp8_wide(g) ? 256 : (((k % 128) != 0 && k <= 256) ? 256 : 128);
    return (m + bm - 1) / bm;
}

extern "C" int frhip_linear_fwd_fp8(const void* a8, const void* w8, const float* wscale, float act_scale, const float* bias,
                                    void* out, float* stats_partial, int m, int n, int k, hipStream_t stream) {
    NtGeom g;
    int rc = fp8_geom(g, m, 1, 1, k, n, 1, 1, 1, 0, "frhip_linear_fwd_fp8");
    if (rc) return rc;
    EpiBnRed br = NO_EPI;
    br.bias = bias;
    return fp8_run(g, a8, w8, wscale, act_scale, out, stats_partial, br, stream);
}
