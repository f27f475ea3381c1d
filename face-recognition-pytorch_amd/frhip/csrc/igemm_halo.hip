// Launcher of the halo-tile 3x3/s1/p1 convolution (see igemm_halo.h).  Reached through frhip_conv_fwd /
// frhip_conv_dgrad when the geometry matches; the generic NT kernel covers everything else.
#include "igemm_halo.h"
#include "frhip.h"

namespace frhip {

template <typename T, int WM, int WN, int MT, int HBUFS>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN) / 4) void halo_kernel(HaloGeom g, const void* __restrict__ a,
                                                                         const void* __restrict__ b, void* __restrict__ out,
                                                                         const void* __restrict__ res, float* __restrict__ stats,
                                                                         EpiBnRed br, int mtiles, int ntiles) {
    typedef HaloTile<T, WM, WN, MT, HBUFS> Tile;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lin = xcd_remap(blockIdx.x, gridDim.x);
    const int ntile = (int)(lin % (uint32_t)ntiles), mtile = (int)(lin / (uint32_t)ntiles);
    HaloMainloop<T, WM, WN, MT, HBUFS> ml;
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
    const int m0 = mtile * Tile::BM + wm * Tile::WROWS, n0 = ntile * Tile::BN + wn * 64;
    EpiOperands<T, Tile::WROWS> eo;
    eo.fetch(res, stats ? br.y : nullptr, g.M, g.Nout, m0, n0);
    const char* mine = ml.template stage_out<T>(smem);
    nt_epilogue_store<T, WM, WN, Tile::WROWS, Tile::THREADS, Tile::BN>(mine, Tile::template stage_pitch<T>(), smem, g.M, g.Nout,
                                                                       out, res != nullptr, stats, br, eo, mtile, ntile, m0, n0);
}

template <typename T, int WM, int WN, int MT, int HBUFS>
static int halo_launch(const HaloGeom& g, const void* a, const void* b, void* out, const void* res, float* stats,
                       const EpiBnRed& br, hipStream_t stream) {
    typedef HaloTile<T, WM, WN, MT, HBUFS> Tile;
    const int mtiles = (g.M + Tile::BM - 1) / Tile::BM, ntiles = (g.Nout + Tile::BN - 1) / Tile::BN;
    const int lds = Tile::template lds_bytes<T>();
    auto kern = halo_kernel<T, WM, WN, MT, HBUFS>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            set_error("igemm_halo: cannot raise dynamic LDS to %d bytes", lds);
            return FRHIP_ELAUNCH;
        }
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(mtiles * ntiles), dim3(Tile::THREADS), lds, stream, g, a, b, out, res, stats, br, mtiles, ntiles);
    return check_launch("igemm_halo");
}

static int g_halo_enabled = 1;

bool halo_applicable(int dtype, int h, int w, int c, int k, int r, int s, int stride, int pad) {
    const int bke = NT_ROWB / (dtype == FRHIP_DT_BF16 ? 2 : 4);
    return g_halo_enabled && r == 3 && s == 3 && stride == 1 && pad == 1 && w <= 56 && (c % bke) == 0 && (k % 8) == 0 &&
           (dtype == FRHIP_DT_BF16 || dtype == FRHIP_DT_F32);
}

int halo_block_m() { return 256; }

// a: activations [n,h,w,c] (forward: x, data-gradient: dy), b: [k][3][3][c] K-contiguous pack, out [n,h,w,k]
int halo_run(int dtype, const void* a, const void* b, void* out, const void* res, float* stats, const EpiBnRed& br,
             int n, int h, int w, int c, int k, int sign, hipStream_t stream) {
    const int es = dtype == FRHIP_DT_BF16 ? 2 : 4;
    const long long ab = 1LL * n * h * w * c * es, bb = 1LL * k * 9 * c * es;
    if (ab > 0x7fffffffLL || bb > 0x7fffffffLL) { set_error("igemm_halo: tensor exceeds the 2 GiB buffer window"); return FRHIP_EINVAL; }
    HaloGeom g;
    g.H = h; g.W = w; g.C = c; g.M = n * h * w; g.Nout = k; g.Ktot = 9 * c; g.sign = sign;
    g.a_bytes = (uint32_t)ab; g.b_bytes = (uint32_t)bb;
    const int nchunks = c / (NT_ROWB / es);
    const bool narrow = (k % 128) != 0;
    if (dtype == FRHIP_DT_BF16) {
        // measured (tools/bench_kernels.py fwd, B=512): up to 256 input channels the 4-wave 256x64 tile with TWO
        // workgroups per CU wins (117 vs 144 us at 128 ch, 106 vs 124 us at 256 ch): the second workgroup's MFMAs cover
        // the first one's prologue, halo reloads, barriers and store epilogue.  From 512 channels the 8-wave 256x128
        // tile with the double-buffered halo is level or ahead.  g_halo_enabled: 2 forces the former, 3 the latter.
        const bool two_per_cu = g_halo_enabled == 2 || (g_halo_enabled != 3 && nchunks <= 4);
        if (two_per_cu || narrow) {
            if (nchunks == 1 || two_per_cu) return halo_launch<bf16_t, 4, 1, 4, 1>(g, a, b, out, res, stats, br, stream);
            return halo_launch<bf16_t, 8, 1, 2, 2>(g, a, b, out, res, stats, br, stream);
        }
        return halo_launch<bf16_t, 4, 2, 4, 2>(g, a, b, out, res, stats, br, stream);
    }
    if (narrow) return halo_launch<float, 8, 1, 2, 2>(g, a, b, out, res, stats, br, stream);
    return halo_launch<float, 4, 2, 4, 2>(g, a, b, out, res, stats, br, stream);
}

}  // namespace frhip

extern "C" int frhip_set_conv_halo(int enabled) { const int old = frhip::g_halo_enabled; frhip::g_halo_enabled = enabled; return old; }
