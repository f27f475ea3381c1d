// Launcher of the halo-tile 3x3/s1/p1 convolution (see igemm_halo.h).  Reached through frhip_conv_fwd /
// frhip_conv_dgrad when the geometry matches; the generic NT kernel covers everything else.
#include <cstdlib>
#include "igemm_halo.h"
#include "igemm_halo_wide.h"
#include "frhip.h"

namespace frhip {

template <typename T, int WM, int WN, int MT, int HBUFS, bool XF = false>
__device__ __forceinline__ void halo_body(const HaloGeom& g, const void* __restrict__ a, const void* __restrict__ b,
                                          void* __restrict__ out, const void* __restrict__ res, float* __restrict__ stats,
                                          const EpiBnRed& br, char* smem, int mtile, int ntile) {
    typedef HaloTile<T, WM, WN, MT, HBUFS> Tile;
    HaloMainloop<T, WM, WN, MT, HBUFS, XF> ml;
    ml.run(g, a, b, smem, mtile, ntile);
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
    const int m0 = g.m_origin + mtile * Tile::BM + wm * Tile::WROWS, n0 = ntile * Tile::BN + wn * 64;
    EpiOperands<T, Tile::WROWS> eo;
    eo.fetch(res, (stats || br.gelu_bwd) ? br.y : nullptr, g.M, g.Nout, m0, n0, br.res_h, br.res_w, &br.map);
    const char* mine = ml.template stage_out<T>(smem);
    nt_epilogue_store<T, WM, WN, Tile::WROWS, Tile::THREADS, Tile::BN>(mine, Tile::template stage_pitch<T>(), smem, g.M, g.Nout,
                                                                       out, res != nullptr, stats, br, eo, g.stat_row0 + mtile, ntile, m0, n0);
}

template <typename T, int WM, int WN, int MT, int HBUFS, bool XF = false>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN) / 4) void halo_kernel(HaloGeom g, const void* __restrict__ a,
                                                                         const void* __restrict__ b, void* __restrict__ out,
                                                                         const void* __restrict__ res, float* __restrict__ stats,
                                                                         EpiBnRed br, int mtiles, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lin = xcd_remap(blockIdx.x, gridDim.x);
    const int ntile = (int)(lin % (uint32_t)ntiles), mtile = (int)(lin / (uint32_t)ntiles);
    halo_body<T, WM, WN, MT, HBUFS, XF>(g, a, b, out, res, stats, br, smem, mtile, ntile);
}

// The 4-wave 256 x 64 tile with MIXED tile heights (see halo_mixed_plan below): m-tiles [0, wide_big) have 256 rows (MT = 4), the rest
// 192 (MT = 3); both bodies in one kernel behind a workgroup-uniform branch.  m0 = m_origin + mtile * BM inside the body, so the
// 192-row body gets m_origin = 64 * big: 64 big + 192 mtile = 256 big + 192 (mtile - big).
__global__ __launch_bounds__(256, 2) void halo_mixed_kernel(HaloGeom g, const void* __restrict__ a, const void* __restrict__ b,
                                                            void* __restrict__ out, const void* __restrict__ res,
                                                            float* __restrict__ stats, EpiBnRed br, int mtiles, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lin = xcd_remap(blockIdx.x, gridDim.x);
    const int ntile = (int)(lin % (uint32_t)ntiles), mtile = (int)(lin / (uint32_t)ntiles);
    if (mtile < g.wide_big) halo_body<bf16_t, 4, 1, 4, 1>(g, a, b, out, res, stats, br, smem, mtile, ntile);
    else {
        g.m_origin = 64 * g.wide_big;
        halo_body<bf16_t, 4, 1, 3, 1>(g, a, b, out, res, stats, br, smem, mtile, ntile);
    }
}

// 4 waves, 256 (or 192) pixels x 128 channels, a 64 (48) x 128 tile per wave (igemm_halo_wide.h).  The store epilogue is the
// shared one, run once per 64-channel half of the tile: to it the workgroup looks like two <4 x 1>-wave tiles of 64 channels.
template <int MT>
__device__ __forceinline__ void halo_wide_body(const HaloGeom& g, const void* __restrict__ a, const void* __restrict__ b,
                                               void* __restrict__ out, const void* __restrict__ res, float* __restrict__ stats,
                                               const EpiBnRed& br, char* smem, int mtile, int ntile, int m_tile0) {
    typedef HaloWideTile<MT> Tile;
    typedef bf16_t T;
    HaloWideMainloop<MT> ml;
    ml.run(g, a, b, smem, m_tile0, ntile);
    const int m0 = m_tile0 + wave_id() * Tile::WROWS;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int n0 = ntile * Tile::BN + half * 64;
        EpiOperands<T, Tile::WROWS> eo;
        eo.fetch(res, (stats || br.gelu_bwd) ? br.y : nullptr, g.M, g.Nout, m0, n0, br.res_h, br.res_w, &br.map);
        const char* mine = ml.stage_out(smem, half);
        nt_epilogue_store<T, 4, 1, Tile::WROWS, Tile::THREADS, 64>(mine, Tile::stage_pitch, smem, g.M, g.Nout, out, res != nullptr,
                                                                   stats, br, eo, mtile, ntile * 2 + half, m0, n0);
    }
}

__global__ __launch_bounds__(256, 2) void halo_wide_kernel(HaloGeom g, const void* __restrict__ a, const void* __restrict__ b,
                                                           void* __restrict__ out, const void* __restrict__ res,
                                                           float* __restrict__ stats, EpiBnRed br, int mtiles, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lin = xcd_remap(blockIdx.x, gridDim.x);
    const int ntile = (int)(lin % (uint32_t)ntiles), mtile = (int)(lin / (uint32_t)ntiles);
    // the few 256-row tiles come first (they start first and take longest), then the 192-row ones: a workgroup-uniform branch
    if (mtile < g.wide_big) halo_wide_body<4>(g, a, b, out, res, stats, br, smem, mtile, ntile, mtile * 256);
    else halo_wide_body<3>(g, a, b, out, res, stats, br, smem, mtile, ntile, g.wide_big * 256 + (mtile - g.wide_big) * 192);
}

// Tile plan of a wide launch.  With equal 256-row tiles a launch of T workgroups on 512 resident ones takes ceil(T / 512)
// rounds and the last one is mostly empty at the ResNet shapes (B = 512: 784 workgroups at 14x14x256, 1568 at 28x28x128,
// 392 at 7x7x512 -- on shapes that fill whole rounds the same kernel runs 15-25 % faster, tools/bench_halo_rounds.py).
// Instead: make the number of workgroups a MULTIPLE of the resident ones and fit the rows with two tile heights,
// `big` tiles of 256 rows and the rest of 192 (MT = 3): 256 big + 192 (T - big) >= M.
struct HaloWidePlan { int mtiles, big; };
static int g_wide_slots = getenv("FRHIP_HALO_WIDE_SLOTS") ? atoi(getenv("FRHIP_HALO_WIDE_SLOTS")) : 512;   // 0: equal tiles only
// Which launches get the mixed plan: bit 0 forward, bit 1 data-gradient.  OFF by default.  Measured inside the ResNet50 step (A/B
// rounds on one box each): every planned launch is faster on an empty chip (14x14x256: 110 -> 103 us on the wide tile), and with
// the wide tile in both directions the plan gained 0.2 ms in the forward launches and lost 0.15 ms in the data-gradient ones (in the
// backward pass each CU also holds a weight-gradient workgroup of the side stream: the resident-slot arithmetic does not apply
// and the extra tiles only add prologues).  But the forward pass is better off on the 4-wave 64-wide tile altogether (wide tile
// forward + data-gradient 26.48 ms, forward only 26.94, data-gradient only 26.33), and on THAT tile, with twice the tiles per
// launch, equal heights win: 26.26 (off) / 26.42 (forward) / 26.72 ms (both).
static int g_wide_mix = getenv("FRHIP_HALO_WIDE_MIX") ? atoi(getenv("FRHIP_HALO_WIDE_MIX")) : 0;
// ntiles = column tiles of the kernel that will run (128 channels wide for the wide tile, 64 for the 4-wave tile)
static HaloWidePlan halo_mixed_plan(int M, int ntiles, int sign) {
    const int even = (M + 255) / 256;
    HaloWidePlan p = {even, even};
    if (!(g_wide_mix & (sign > 0 ? 1 : 2))) return p;
    if (g_wide_slots <= 0 || ntiles > g_wide_slots || (g_wide_slots % ntiles) != 0) return p;
    const int per = g_wide_slots / ntiles;                       // m-tiles per round
    const double x = (double)even * ntiles / g_wide_slots;       // rounds of equal tiles
    const double frac = x - (double)(long long)x;
    const double cost_even = (double)(long long)x + (frac == 0.0 ? 0.0 : (frac <= 0.5 ? 0.7 : 1.0));   // a half-empty round: one workgroup per CU, faster
    for (int r = 1; r <= 64; ++r) {
        const long long T = 1LL * per * r;
        if (T * 256 < M) continue;                               // not enough rows even with big tiles only
        if (T * 192 > M) break;                                  // small tiles alone overshoot: nothing to balance
        const int big = (int)((M - T * 192 + 63) / 64);
        const double cost = ((double)big + 0.8 * (double)(T - big)) / (double)per;      // a 192-row tile costs ~0.8 of a 256-row one
        if (cost < 0.97 * cost_even) { p.mtiles = (int)T; p.big = big; }
        break;
    }
    return p;
}

static int halo_wide_launch(HaloGeom g, const void* a, const void* b, void* out, const void* res, float* stats,
                            const EpiBnRed& br, hipStream_t stream) {
    typedef HaloWideTile<4> Tile;
    const int ntiles = (g.Nout + Tile::BN - 1) / Tile::BN;
    const HaloWidePlan p = halo_mixed_plan(g.M, ntiles, g.sign);
    g.wide_big = p.big;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(halo_wide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, Tile::LDS) != hipSuccess) {
            set_error("igemm_halo: cannot raise dynamic LDS to %d bytes", Tile::LDS);
            return FRHIP_ELAUNCH;
        }
        attr_done = true;
    }
    hipLaunchKernelGGL(halo_wide_kernel, dim3(p.mtiles * ntiles), dim3(Tile::THREADS), Tile::LDS, stream, g, a, b, out, res, stats, br, p.mtiles, ntiles);
    return check_launch("igemm_halo_wide");
}

static int halo_mixed_launch(HaloGeom g, const HaloWidePlan& p, const void* a, const void* b, void* out, const void* res, float* stats,
                             const EpiBnRed& br, hipStream_t stream) {
    typedef HaloTile<bf16_t, 4, 1, 4, 1> Tile;
    const int ntiles = (g.Nout + Tile::BN - 1) / Tile::BN;
    const int lds = Tile::template lds_bytes<bf16_t>();
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(halo_mixed_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            set_error("igemm_halo: cannot raise dynamic LDS to %d bytes", lds);
            return FRHIP_ELAUNCH;
        }
        attr_done = true;
    }
    g.wide_big = p.big; g.m_origin = 0; g.stat_row0 = 0;
    hipLaunchKernelGGL(halo_mixed_kernel, dim3(p.mtiles * ntiles), dim3(Tile::THREADS), lds, stream, g, a, b, out, res, stats, br, p.mtiles, ntiles);
    return check_launch("igemm_halo_mixed");
}

// Tail balancing.  A launch of T = mtiles x ntiles equal tiles on `slots` resident workgroups takes ceil(T / slots) rounds;
// with T = 3.06 x slots (256-channel layers at B = 512) the fourth round is 94 % empty.  Plan: run only the FULL rounds
// with 256-row tiles, then cover the remaining rows of every column with at most slots / ntiles smaller tiles
// (h x 64 rows, h = 1..3, a second launch of the same kernel instantiated with MT = h): the tail then costs ~h/4 of a round.
struct HaloPlan {
    int nb;        // 256-row m-tiles per column in the first launch
    int h;         // tail tile height in 64-row units (0 = no tail launch)
    int ns;        // tail m-tiles per column
    int rows() const { return nb + ns; }
};

static HaloPlan halo_plan(int M, int ntiles, int slots, bool allow_tail) {
    const int mtiles = (M + 255) / 256;
    HaloPlan p = {mtiles, 0, 0};
    const long long T = 1LL * mtiles * ntiles;
    if (!allow_tail || ntiles > slots || (slots % ntiles) != 0 || T <= slots) return p;
    const long long full = (T / slots) * slots;
    if (full == T) return p;
    const int nb = (int)(full / ntiles);                  // slots % ntiles == 0 -> exact
    const int units = (M - nb * 256 + 63) / 64;           // 64-row units left per column
    const int per_col = slots / ntiles;
    const int h = (units + per_col - 1) / per_col;        // <= 4 because fewer than `slots` full tiles are left
    if (h >= 4) return p;                                 // the tail would be (almost) a whole round anyway
    p.nb = nb; p.h = h; p.ns = (units + h - 1) / h;
    return p;
}

template <typename T, int WM, int WN, int MT, int HBUFS, bool XF = false>
static int halo_launch_one(HaloGeom g, const void* a, const void* b, void* out, const void* res, float* stats,
                           const EpiBnRed& br, int mtiles, int m_origin, int stat_row0, hipStream_t stream) {
    typedef HaloTile<T, WM, WN, MT, HBUFS> Tile;
    const int ntiles = (g.Nout + Tile::BN - 1) / Tile::BN;
    const int lds = Tile::template lds_bytes<T>();
    auto kern = halo_kernel<T, WM, WN, MT, HBUFS, XF>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            set_error("igemm_halo: cannot raise dynamic LDS to %d bytes", lds);
            return FRHIP_ELAUNCH;
        }
        attr_done = true;
    }
    g.m_origin = m_origin; g.stat_row0 = stat_row0;
    hipLaunchKernelGGL(kern, dim3(mtiles * ntiles), dim3(Tile::THREADS), lds, stream, g, a, b, out, res, stats, br, mtiles, ntiles);
    return check_launch("igemm_halo");
}

// Off by default.  Measured on the ResNet50 step (B = 512): the tail launches of 64-row tiles take 12 us each (prologue-
// bound) and save about as much in the main launch (122 vs 134 us with two workgroups per CU, whose last partial round
// already runs at twice the per-workgroup speed; 111 vs 115 us for the 8-wave tile): no net gain.  frhip_set_conv_halo
// bit 5 turns it on (tests keep both paths honest).
static int g_halo_tail = getenv("FRHIP_HALO_TAIL") ? atoi(getenv("FRHIP_HALO_TAIL")) : 0;

// WGPC = workgroups of this configuration one CU holds (LDS / register limited): 2 for the 4-wave tile, 1 for the 8-wave one
template <typename T, int WM, int WN, int HBUFS, int WGPC>
static int halo_launch(const HaloGeom& g, const void* a, const void* b, void* out, const void* res, float* stats,
                       const EpiBnRed& br, hipStream_t stream) {
    static_assert(WM == 4, "256-row tiles");
    const int ntiles = (g.Nout + WN * 64 - 1) / (WN * 64);
    const HaloPlan p = halo_plan(g.M, ntiles, 256 * WGPC, g_halo_tail != 0);
    int rc = halo_launch_one<T, WM, WN, 4, HBUFS>(g, a, b, out, res, stats, br, p.nb, 0, 0, stream);
    if (rc || !p.h) return rc;
    switch (p.h) {
        case 1: return halo_launch_one<T, WM, WN, 1, HBUFS>(g, a, b, out, res, stats, br, p.ns, p.nb * 256, p.nb, stream);
        case 2: return halo_launch_one<T, WM, WN, 2, HBUFS>(g, a, b, out, res, stats, br, p.ns, p.nb * 256, p.nb, stream);
        default: return halo_launch_one<T, WM, WN, 3, HBUFS>(g, a, b, out, res, stats, br, p.ns, p.nb * 256, p.nb, stream);
    }
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
//       tail launches off, automatic mode -- and only in the directions g_halo_wide_dirs names (bit 0 forward, bit 1 data-gradient).
//       Default: data-gradient only.  One box, two A/B rounds each: both directions 26.48 ms, forward only 26.94, data-gradient only
//       26.33, nowhere 27.0 -- the wide tile earns its keep beside the weight-gradient workgroups of the backward pass; the forward
//       launches, alone on the chip, are better off with twice as many 64-wide tiles.
static int g_halo_wide = getenv("FRHIP_HALO_WIDE") ? atoi(getenv("FRHIP_HALO_WIDE")) : 1;
static int g_halo_wide_minc = getenv("FRHIP_HALO_WIDE_MINC") ? atoi(getenv("FRHIP_HALO_WIDE_MINC")) : 128;
static int g_halo_wide_dirs = getenv("FRHIP_HALO_WIDE_DIRS") ? atoi(getenv("FRHIP_HALO_WIDE_DIRS")) : 2;
static int halo_config(int dtype, int c, int k);
static int halo_config_w(int dtype, int w, int c, int k, int sign) {
    if (g_halo_wide && (g_halo_wide_dirs & (sign > 0 ? 1 : 2)) && dtype == FRHIP_DT_BF16 && w <= HaloWideTile<4>::MAXW &&
        (k % 128) == 0 && (c % 64) == 0 && c >= g_halo_wide_minc && !g_halo_tail && (g_halo_enabled & 3) == 1)
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

// rows of the BN-partial buffer a halo launch writes for an output of m pixels
int halo_stat_rows(int dtype, int m, int w, int c, int k, int sign) {
    if (halo_config_w(dtype, w, c, k, sign) == 3) return halo_mixed_plan(m, (k + 127) / 128, sign).mtiles;
    const int cfg = halo_config(dtype, c, k);
    if (cfg == 0 && dtype == FRHIP_DT_BF16 && !g_halo_tail) return halo_mixed_plan(m, (k + 63) / 64, sign).mtiles;
    if (cfg == 1) return (m + 255) / 256;
    const int bn = cfg == 0 ? 64 : 128;
    return halo_plan(m, (k + bn - 1) / bn, cfg == 0 ? 512 : 256, g_halo_tail != 0).rows();
}

// a: activations [n,h,w,c] (forward: x, data-gradient: dy), b: [k][3][3][c] K-contiguous pack, out [n,h,w,k]
// bn1 -> relu -> conv2 with the BatchNorm-apply + ReLU folded into the operand path: bf16, the 4-wave single-buffer tile only
bool halo_xf_applicable(int dtype, int h, int w, int c, int k, int r, int s, int stride, int pad) {
    return dtype == FRHIP_DT_BF16 && halo_applicable(dtype, h, w, c, k, r, s, stride, pad) && halo_config(dtype, c, k) == 0;
}

int halo_run(int dtype, const void* a, const void* b, void* out, const void* res, float* stats, const EpiBnRed& br,
             int n, int h, int w, int c, int k, int sign, hipStream_t stream, const float* xf_scale, const float* xf_shift) {
    const int es = dtype == FRHIP_DT_BF16 ? 2 : 4;
    const long long ab = 1LL * n * h * w * c * es, bb = 1LL * k * 9 * c * es;
    if (ab > 0x7fffffffLL || bb > 0x7fffffffLL) { set_error("igemm_halo: tensor exceeds the 2 GiB buffer window"); return FRHIP_EINVAL; }
    HaloGeom g;
    g.H = h; g.W = w; g.C = c; g.M = n * h * w; g.Nout = k; g.Ktot = 9 * c; g.sign = sign;
    g.m_origin = 0; g.stat_row0 = 0;
    g.a_bytes = (uint32_t)ab; g.b_bytes = (uint32_t)bb;
    g.xf_scale = xf_scale; g.xf_shift = xf_shift; g.wide_big = 0;
    g.d_hw = make_fastdiv((uint32_t)(h * w)); g.d_w = make_fastdiv((uint32_t)w);
    static const int prio = getenv("FRHIP_HALO_PRIO") ? atoi(getenv("FRHIP_HALO_PRIO")) : 0;
    g.wave_prio = prio;
    const int cfg = halo_config(dtype, c, k);
    if (xf_scale) {
        if (dtype != FRHIP_DT_BF16 || cfg != 0 || g_halo_tail) { set_error("igemm_halo: operand transform not available for this shape"); return FRHIP_EINVAL; }
        const int wrote = (g.M + 255) / 256;
        const int rc = halo_launch_one<bf16_t, 4, 1, 4, 1, true>(g, a, b, out, res, stats, br, wrote, 0, 0, stream);
        // the caller sized the BN-partial buffer with halo_stat_rows, which may plan more (smaller) tiles than this kernel has
        const int want = halo_stat_rows(dtype, g.M, w, c, k, sign);
        if (!rc && stats && want > wrote &&
            hipMemsetAsync(stats + (size_t)wrote * 2 * k, 0, (size_t)(want - wrote) * 2 * k * sizeof(float), stream) != hipSuccess) {
            set_error("igemm_halo: cannot clear the unused partial rows");
            return FRHIP_ELAUNCH;
        }
        return rc;
    }
    if (halo_config_w(dtype, w, c, k, sign) == 3) return halo_wide_launch(g, a, b, out, res, stats, br, stream);
    if (dtype == FRHIP_DT_BF16) {
        if (cfg == 0 && !g_halo_tail) {
            const HaloWidePlan p = halo_mixed_plan(g.M, (k + 63) / 64, sign);
            if (p.big != p.mtiles) return halo_mixed_launch(g, p, a, b, out, res, stats, br, stream);
        }
        if (cfg == 0) return halo_launch<bf16_t, 4, 1, 1, 2>(g, a, b, out, res, stats, br, stream);
        if (cfg == 1) return halo_launch_one<bf16_t, 8, 1, 2, 2>(g, a, b, out, res, stats, br, (g.M + 255) / 256, 0, 0, stream);
        return halo_launch<bf16_t, 4, 2, 2, 1>(g, a, b, out, res, stats, br, stream);
    }
    if (cfg == 1) return halo_launch_one<float, 8, 1, 2, 2>(g, a, b, out, res, stats, br, (g.M + 255) / 256, 0, 0, stream);
    return halo_launch_one<float, 4, 2, 4, 2>(g, a, b, out, res, stats, br, (g.M + 255) / 256, 0, 0, stream);
}

}  // namespace frhip

extern "C" int frhip_set_halo_wide_slots(int slots) {
    // low 16 bits: resident workgroups to balance for; bit 20 set: bits 16-17 = which launches get the mixed-height plan
    // (1 forward, 2 data-gradient); bit 21 set: bits 18-19 = which launches may use the 64 x 128-per-wave tile.
    // The returned old value has bits 20 and 21 set, so passing it back restores everything.
    const int old = frhip::g_wide_slots | (frhip::g_wide_mix << 16) | (frhip::g_halo_wide_dirs << 18) | (3 << 20);
    frhip::g_wide_slots = slots & 0xffff;
    if (slots & (1 << 20)) frhip::g_wide_mix = (slots >> 16) & 3;
    if (slots & (1 << 21)) frhip::g_halo_wide_dirs = (slots >> 18) & 3;
    return old;
}

extern "C" int frhip_set_conv_halo(int enabled) {
    // bits 0-1: 0 off, 1 auto, 2 force the 4-wave tile, 3 force the 8-wave tile; bit 5 set: tail-balancing launch on;
    // bit 6 set: the 64 x 128-per-wave tile OFF (it is on by default in auto mode)
    const int old = frhip::g_halo_enabled | (frhip::g_halo_tail ? 32 : 0) | (frhip::g_halo_wide ? 0 : 64);
    frhip::g_halo_enabled = enabled & 3; frhip::g_halo_tail = (enabled & 32) ? 1 : 0; frhip::g_halo_wide = (enabled & 64) ? 0 : 1;
    return old;
}
