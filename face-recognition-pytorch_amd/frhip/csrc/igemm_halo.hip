// Launcher of the halo-tile 3x3/s1/p1 convolution (see igemm_halo.h).  Reached through frhip_conv_fwd /
// frhip_conv_dgrad when the geometry matches; the generic NT kernel covers everything else.
#include <cstdlib>
#include "igemm_halo.h"
#include "igemm_halo_wide.h"
#include "frhip.h"

namespace frhip {

// Diagnostic build only (-DFRHIP_CLOCK_STAMP=1, tools/clock_probe.py): lane 0 of wave 0 stamps the shader-clock counter
// (s_memtime) and the constant 100-MHz counter (s_memrealtime) around the main loop; the quotient is the clock the chip holds
// INSIDE the kernel (MI355X_MICROARCH "DVFS give-back" item 6).  The stamps go to a buffer of their own that nothing else reads.
#ifndef FRHIP_CLOCK_STAMP
#define FRHIP_CLOCK_STAMP 0
#endif
#if FRHIP_CLOCK_STAMP
__device__ unsigned long long g_clock_stamps[2 * 8192];
#endif

template <typename T, int WM, int WN, int MT, int HBUFS, bool XF = false, bool LEAN = false>
__device__ __forceinline__ void halo_body(const HaloGeom& g, const void* __restrict__ a, const void* __restrict__ b,
                                          void* __restrict__ out, const void* __restrict__ res, float* __restrict__ stats,
                                          const EpiBnRed& br, char* smem, int mtile, int ntile) {
    typedef HaloTile<T, WM, WN, MT, HBUFS> Tile;
    HaloMainloop<T, WM, WN, MT, HBUFS, XF> ml;
#if FRHIP_CLOCK_STAMP
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#endif
    ml.run(g, a, b, smem, mtile, ntile);
#if FRHIP_CLOCK_STAMP
    if (threadIdx.x == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        g_clock_stamps[2 * (blockIdx.x & 8191)] = t1 - t0;
        g_clock_stamps[2 * (blockIdx.x & 8191) + 1] = r1 - r0;
    }
#endif
    if constexpr (FRHIP_ABL & 16) {
        if (g.M >= 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j) asm volatile("" :: "v"(ml.acc[i][j]));
            return;
        }
    }
    const int wave = wave_id();
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = mtile * Tile::BM + wm * Tile::WROWS, n0 = ntile * Tile::BN + wn * 64;
    EpiOperands<T, Tile::WROWS> eo;
    if constexpr (LEAN) {
        eo.fetch_fast(res, stats ? br.y : nullptr, g.M, g.Nout, m0, n0);
        const char* mine = ml.template stage_out<T>(smem);
        nt_epilogue_store_fast<T, WM, WN, Tile::WROWS, Tile::THREADS, Tile::BN>(mine, Tile::template stage_pitch<T>(), smem, g.M, g.Nout,
                                                                                out, res != nullptr, stats, br, eo, mtile, ntile);
    } else {
        eo.fetch(res, (stats || br.gelu_bwd) ? br.y : nullptr, g.M, g.Nout, m0, n0, br.res_h, br.res_w, &br.map);
        const char* mine = ml.template stage_out<T>(smem);
        nt_epilogue_store<T, WM, WN, Tile::WROWS, Tile::THREADS, Tile::BN>(mine, Tile::template stage_pitch<T>(), smem, g.M, g.Nout,
                                                                           out, res != nullptr, stats, br, eo, mtile, ntile, m0, n0);
    }
}

// LEAN: the host found every tile whole and the layout dense (epi_lean_ok): lean store epilogue only
template <typename T, int WM, int WN, int MT, int HBUFS, bool XF = false, bool LEAN = false>
__global__ __launch_bounds__(64 * WM * WN, 2) void halo_kernel(HaloGeom g, const void* __restrict__ a,
                                                                         const void* __restrict__ b, void* __restrict__ out,
                                                                         const void* __restrict__ res, float* __restrict__ stats,
                                                                         EpiBnRed br, int mtiles, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lin = xcd_remap(blockIdx.x, gridDim.x);
    const int ntile = (int)(lin % (uint32_t)ntiles), mtile = (int)(lin / (uint32_t)ntiles);
    halo_body<T, WM, WN, MT, HBUFS, XF, LEAN>(g, a, b, out, res, stats, br, smem, mtile, ntile);
}

// 4 waves, 256 pixels x 128 channels, a 64 x 128 tile per wave (igemm_halo_wide.h).  The store epilogue is the shared one, run
// once per 64-channel half of the tile: to it the workgroup looks like two <4 x 1>-wave tiles of 64 channels.
template <bool LEAN>
__global__ __launch_bounds__(256, 2) void halo_wide_kernel(HaloGeom g, const void* __restrict__ a, const void* __restrict__ b,
                                                           void* __restrict__ out, const void* __restrict__ res,
                                                           float* __restrict__ stats, EpiBnRed br, int mtiles, int ntiles) {
    typedef HaloWideTile Tile;
    typedef bf16_t T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lin = xcd_remap(blockIdx.x, gridDim.x);
    const int ntile = (int)(lin % (uint32_t)ntiles), mtile = (int)(lin / (uint32_t)ntiles);
    const int m_tile0 = mtile * Tile::BM;
    HaloWideMainloop ml;
#if FRHIP_CLOCK_STAMP
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#endif
    ml.run(g, a, b, smem, m_tile0, ntile);
#if FRHIP_CLOCK_STAMP
    if (threadIdx.x == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        g_clock_stamps[2 * (blockIdx.x & 8191)] = t1 - t0;
        g_clock_stamps[2 * (blockIdx.x & 8191) + 1] = r1 - r0;
    }
#endif
    const int m0 = m_tile0 + wave_id() * Tile::WROWS;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int n0 = ntile * Tile::BN + half * 64;
        EpiOperands<T, Tile::WROWS> eo;
        if constexpr (LEAN) {
            eo.fetch_fast(res, stats ? br.y : nullptr, g.M, g.Nout, m0, n0);
            const char* mine = ml.stage_out(smem, half);
            nt_epilogue_store_fast<T, 4, 1, Tile::WROWS, Tile::THREADS, 64>(mine, Tile::stage_pitch, smem, g.M, g.Nout, out, res != nullptr,
                                                                            stats, br, eo, mtile, ntile * 2 + half);
        } else {
            eo.fetch(res, (stats || br.gelu_bwd) ? br.y : nullptr, g.M, g.Nout, m0, n0, br.res_h, br.res_w, &br.map);
            const char* mine = ml.stage_out(smem, half);
            nt_epilogue_store<T, 4, 1, Tile::WROWS, Tile::THREADS, 64>(mine, Tile::stage_pitch, smem, g.M, g.Nout, out, res != nullptr,
                                                                       stats, br, eo, mtile, ntile * 2 + half, m0, n0);
        }
    }
}

// FRHIP_EPI_LEAN=0: always the general store epilogue (A/B switch; the lean kernels are bit-identical)
int g_epi_lean = getenv("FRHIP_EPI_LEAN") ? atoi(getenv("FRHIP_EPI_LEAN")) : 1;

static int halo_wide_launch(const HaloGeom& g, const void* a, const void* b, void* out, const void* res, float* stats,
                            const EpiBnRed& br, hipStream_t stream) {
    typedef HaloWideTile Tile;
    const int mtiles = (g.M + Tile::BM - 1) / Tile::BM, ntiles = (g.Nout + Tile::BN - 1) / Tile::BN;
    const bool lean = g_epi_lean && epi_lean_ok(true, g.M, g.Nout, Tile::BM, Tile::BN, br);
    auto kern = lean ? halo_wide_kernel<true> : halo_wide_kernel<false>;
    static bool attr_done[2] = {false, false};
    if (!attr_done[lean]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Tile::LDS) != hipSuccess) {
            set_error("igemm_halo: cannot raise dynamic LDS to %d bytes", Tile::LDS);
            return FRHIP_ELAUNCH;
        }
        attr_done[lean] = true;
    }
    hipLaunchKernelGGL(kern, dim3(mtiles * ntiles), dim3(Tile::THREADS), Tile::LDS, stream, g, a, b, out, res, stats, br, mtiles, ntiles);
    return check_launch("igemm_halo_wide");
}

template <typename T, int WM, int WN, int MT, int HBUFS, bool XF = false>
static int halo_launch(const HaloGeom& g, const void* a, const void* b, void* out, const void* res, float* stats,
                       const EpiBnRed& br, hipStream_t stream) {
    typedef HaloTile<T, WM, WN, MT, HBUFS> Tile;
    const int mtiles = (g.M + Tile::BM - 1) / Tile::BM, ntiles = (g.Nout + Tile::BN - 1) / Tile::BN;
    const int lds = Tile::template lds_bytes<T>();
    // lean epilogue: the 4-wave bf16 tile (what the training step runs); the other configurations keep the general one
    constexpr bool HAS_LEAN = sizeof(T) == 2 && WM == 4 && WN == 1 && MT == 4 && HBUFS == 1;
    bool lean = false;
    if constexpr (HAS_LEAN) lean = g_epi_lean && epi_lean_ok(true, g.M, g.Nout, Tile::BM, Tile::BN, br);
    auto kern = halo_kernel<T, WM, WN, MT, HBUFS, XF, false>;
    if constexpr (HAS_LEAN) { if (lean) kern = halo_kernel<T, WM, WN, MT, HBUFS, XF, true>; }
    static bool attr_done[2] = {false, false};
    if (!attr_done[lean]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            set_error("igemm_halo: cannot raise dynamic LDS to %d bytes", lds);
            return FRHIP_ELAUNCH;
        }
        attr_done[lean] = true;
    }
    hipLaunchKernelGGL(kern, dim3(mtiles * ntiles), dim3(Tile::THREADS), lds, stream, g, a, b, out, res, stats, br, mtiles, ntiles);
    return check_launch("igemm_halo");
}

static int g_halo_enabled = getenv("FRHIP_HALO_MODE") ? atoi(getenv("FRHIP_HALO_MODE")) & 3 : 1;     // 2 / 3: force the 4-wave / 8-wave tile

bool halo_applicable(int dtype, int h, int w, int c, int k, int r, int s, int stride, int pad) {
    const int bke = NT_ROWB / (dtype == FRHIP_DT_BF16 ? 2 : 4);
    return g_halo_enabled && r == 3 && s == 3 && stride == 1 && pad == 1 && w <= 56 && (c % bke) == 0 && (k % 8) == 0 &&
           (dtype == FRHIP_DT_BF16 || dtype == FRHIP_DT_F32);
}

// Which kernel configuration a problem gets:
//   0 = <4,1,4,1>: 4 waves, 256 x 64 tile, 64 x 64 per wave, two workgroups per CU (igemm_halo.h)
//   1 = <8,1,2,2>, 2 = <4,2,4,2>: 8 waves, 256 x 128 tile, double-buffered window (fp32 validation mode; bf16 only when forced)
//   3 = the 64 x 128-per-wave tile (igemm_halo_wide.h): bf16, W <= 28, output channels in whole 128s, input channels >= 128,
//       automatic mode -- and only in the directions g_halo_wide_dirs names (bit 0 forward, bit 1 data-gradient).
//       Default: data-gradient only.  One box, two A/B rounds each: both directions 26.48 ms, forward only 26.94, data-gradient only
//       26.33, nowhere 27.0 -- the wide tile earns its keep beside the weight-gradient workgroups of the backward pass; the forward
//       launches, alone on the chip, are better off with twice as many 64-wide tiles.
// Every configuration has 256-row tiles: one BN-partial row per 256 output pixels.
static int g_halo_wide_dirs = getenv("FRHIP_HALO_WIDE_DIRS") ? atoi(getenv("FRHIP_HALO_WIDE_DIRS")) & 3 : 2;
static int halo_config(int dtype, int c, int k);
static int halo_config_w(int dtype, int w, int c, int k, int sign) {
    if ((g_halo_wide_dirs & (sign > 0 ? 1 : 2)) && dtype == FRHIP_DT_BF16 && w <= HaloWideTile::MAXW &&
        (k % 128) == 0 && (c % 64) == 0 && c >= 128 && (g_halo_enabled & 3) == 1)
        return 3;
    return halo_config(dtype, c, k);
}
static int halo_config(int dtype, int c, int k) {
    const int es = dtype == FRHIP_DT_BF16 ? 2 : 4;
    const int nchunks = c / (NT_ROWB / es);
    const bool narrow = (k % 128) != 0;
    if (dtype == FRHIP_DT_BF16) {
        // measured (tools/bench_kernels.py fwd, B=512): up to 256 input channels the 4-wave 256x64 tile with TWO
        // workgroups per CU wins (117 vs 144 us at 128 ch, 106 vs 124 us at 256 ch): the second workgroup's MFMAs cover
        // the first one's prologue, halo reloads, barriers and store epilogue.  From 512 channels the 8-wave 256x128
        // tile with the double-buffered halo is level or ahead.  g_halo_enabled: 2 forces the former, 3 the latter.
        // Round 2: inside the training step the 4-wave tile also wins at 512 channels (27.6 -> 27.35 ms, two A/B rounds): a
        // 73-KB workgroup fits a CU beside the side stream's 82-KB weight-gradient workgroup, the 142-KB 8-wave tile does not.
        const int mode = g_halo_enabled & 3;
        const bool two_per_cu = mode == 2 || (mode != 3 && nchunks <= 8);
        if (two_per_cu || narrow) return (nchunks == 1 || two_per_cu) ? 0 : 1;
        return 2;
    }
    return narrow ? 1 : 2;
}

// would halo_run take a LEAN kernel for this forward problem?  (frhip_conv_fwd_affine: the folded BatchNorm lives in the lean epilogue)
bool halo_lean_applies(int dtype, int n, int h, int w, int c, int k, int sign) {
    if (!g_epi_lean || dtype != FRHIP_DT_BF16) return false;
    const long long M = 1LL * n * h * w;
    const int cfg = halo_config_w(dtype, w, c, k, sign) == 3 ? 3 : halo_config(dtype, c, k);
    const int bn = cfg == 3 ? HaloWideTile::BN : 64;
    if (cfg != 3 && cfg != 0) return false;
    return M % 256 == 0 && k % bn == 0 && M * k * 2 < 0x7fffffffLL;
}

// rows of the BN-partial buffer a halo launch writes for an output of m pixels
int halo_stat_rows(int, int m, int, int, int, int) { return (m + 255) / 256; }

// a: activations [n,h,w,c] (forward: x, data-gradient: dy), b: [k][3][3][c] K-contiguous pack, out [n,h,w,k]
// bn1 -> relu -> conv2 with the BatchNorm-apply + ReLU folded into the operand path: bf16, the 4-wave single-buffer tile only
bool halo_xf_applicable(int dtype, int h, int w, int c, int k, int r, int s, int stride, int pad) {
    return dtype == FRHIP_DT_BF16 && halo_applicable(dtype, h, w, c, k, r, s, stride, pad) && halo_config(dtype, c, k) == 0;
}

int halo_run(int dtype, const void* a, const void* b, void* out, const void* res, float* stats, const EpiBnRed& br,
             int n, int h, int w, int c, int k, int sign, hipStream_t stream, const float* xf_scale, const float* xf_shift, void* xf_out) {
    const int es = dtype == FRHIP_DT_BF16 ? 2 : 4;
    const long long ab = 1LL * n * h * w * c * es, bb = 1LL * k * 9 * c * es;
    if (ab > 0x7fffffffLL || bb > 0x7fffffffLL) { set_error("igemm_halo: tensor exceeds the 2 GiB buffer window"); return FRHIP_EINVAL; }
    HaloGeom g;
    g.H = h; g.W = w; g.C = c; g.M = n * h * w; g.Nout = k; g.Ktot = 9 * c; g.sign = sign;
    g.a_bytes = (uint32_t)ab; g.b_bytes = (uint32_t)bb;
    g.xf_scale = xf_scale; g.xf_shift = xf_shift; g.xf_out = xf_out;
    g.d_hw = make_fastdiv((uint32_t)(h * w)); g.d_w = make_fastdiv((uint32_t)w);
    const int cfg = halo_config(dtype, c, k);
    if (xf_scale) {
        if (dtype != FRHIP_DT_BF16 || cfg != 0) { set_error("igemm_halo: operand transform not available for this shape"); return FRHIP_EINVAL; }
        return halo_launch<bf16_t, 4, 1, 4, 1, true>(g, a, b, out, res, stats, br, stream);
    }
    if (halo_config_w(dtype, w, c, k, sign) == 3) return halo_wide_launch(g, a, b, out, res, stats, br, stream);
    if (dtype == FRHIP_DT_BF16) {
        if (cfg == 0) return halo_launch<bf16_t, 4, 1, 4, 1>(g, a, b, out, res, stats, br, stream);
        if (cfg == 1) return halo_launch<bf16_t, 8, 1, 2, 2>(g, a, b, out, res, stats, br, stream);
        return halo_launch<bf16_t, 4, 2, 4, 2>(g, a, b, out, res, stats, br, stream);
    }
    if (cfg == 1) return halo_launch<float, 8, 1, 2, 2>(g, a, b, out, res, stats, br, stream);
    return halo_launch<float, 4, 2, 4, 2>(g, a, b, out, res, stats, br, stream);
}

}  // namespace frhip

#if FRHIP_CLOCK_STAMP
// diagnostic build only: copies the (shader-clock ticks, 100-MHz ticks) pairs of the last launch's workgroups to the host
extern "C" int frhip_dbg_clock_read(unsigned long long* host, int pairs) {
    if (pairs > 8192) pairs = 8192;
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(frhip::g_clock_stamps), sizeof(unsigned long long) * 2 * pairs) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int frhip_set_epi_lean(int enabled) {
    const int old = frhip::g_epi_lean;
    if (enabled >= 0) frhip::g_epi_lean = enabled != 0;
    return old;
}

extern "C" int frhip_set_halo_wide_dirs(int dirs) {
    // which launches may use the 64 x 128-per-wave tile: bit 0 forward, bit 1 data-gradient (default 2); < 0 queries
    const int old = frhip::g_halo_wide_dirs;
    if (dirs >= 0) frhip::g_halo_wide_dirs = dirs & 3;
    return old;
}

extern "C" int frhip_set_conv_halo(int enabled) {
    // 0 off (generic NT kernel), 1 auto, 2 force the 4-wave tile, 3 force the 8-wave tile
    const int old = frhip::g_halo_enabled;
    frhip::g_halo_enabled = enabled & 3;
    return old;
}
