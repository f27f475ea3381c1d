// Convolution forward / data-gradient and plain NT GEMM on the shared implicit-GEMM main loop.
// Replaces what the reference gets from cuDNN/cuBLAS behind nn.Conv2d / F.linear:
//   /root/reference/nets/resnet.py:23-46 (conv3x3 / conv1x1), :89-103 (BasicBlock), :244 (fc).
#include "igemm_nt.h"
#include "frhip.h"

namespace frhip {

bool halo_applicable(int dtype, int h, int w, int c, int k, int r, int s, int stride, int pad);
int halo_stat_rows(int dtype, int m, int w, int c, int k, int sign);
bool halo_xf_applicable(int dtype, int h, int w, int c, int k, int r, int s, int stride, int pad);
bool halo_lean_applies(int dtype, int n, int h, int w, int c, int k, int sign);
int halo_run(int dtype, const void* a, const void* b, void* out, const void* res, float* stats, const EpiBnRed& br,
             int n, int h, int w, int c, int k, int sign, hipStream_t stream, const float* xf_scale = nullptr,
             const float* xf_shift = nullptr, void* xf_out = nullptr);

enum { EPI_STORE = 0, EPI_ATOMIC = 1, EPI_SLAB = 2, EPI_LEAN = 3 };    // SLAB: K split y stores its fp32 partial tile to slab y of `out`; LEAN: nt_epilogue_store_lean
extern int g_epi_lean;      // igemm_halo.hip: frhip_set_epi_lean / FRHIP_EPI_LEAN

// one output tile (logical id t of `total`)
template <typename T, int WM, int WN, int MT, int EPI>
__device__ __forceinline__ void nt_tile(const NtGeom& g, const void* __restrict__ a, const void* __restrict__ b, void* __restrict__ out,
                                        const void* __restrict__ res, float* __restrict__ stats, const EpiBnRed& br, int ntiles,
                                        uint32_t t, uint32_t total, char* smem) {
    typedef NtTile<T, WM, WN, MT> Tile;
    constexpr int WROWS = Tile::WROWS, THREADS = Tile::THREADS;
    const uint32_t lin = xcd_remap(t, total);
    const int ntile = (int)(lin % (uint32_t)ntiles), mtile = (int)(lin / (uint32_t)ntiles);
    const int ks_begin = blockIdx.y * g.ksteps_per_split;
    const int ks_end = min(g.ksteps, ks_begin + g.ksteps_per_split);

    NtMainloop<T, WM, WN, MT> ml;
    ml.run(g, a, b, smem, mtile, ntile, ks_begin, ks_end);

    const int lane = lane_id(), wave = wave_id();
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = mtile * Tile::BM + wm * WROWS, n0 = ntile * Tile::BN + wn * 64;

    if constexpr (EPI == EPI_LEAN) {
        if (stats && br.y && !br.gelu_bwd) {
            // BatchNorm-backward partials (a data-gradient whose result is the upstream gradient of a BatchNorm): two 64-row halves
            constexpr int P = Tile::template stage_pitch<T>();
            float accs[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                EpiOperands<T, 64> eo;
                eo.fetch_fast(res, br.y, g.M, g.Nout, m0 + 64 * half, n0);
                const char* mine = ml.stage_half(smem, half);
                nt_epilogue_store_fast<T, WM, WN, 64, THREADS, Tile::BN>(mine, P, smem, g.M, g.Nout, out, res != nullptr, stats, br, eo, mtile, ntile,
                                                                         64 * P, accs);
            }
            epi_stats_tail_mfma<WM, WN, THREADS, Tile::BN>(accs[0], accs[1], smem, g.Nout, stats, mtile, ntile);
        } else {
            const char* mine = ml.template stage_out<T>(smem);
            nt_epilogue_store_lean<WM, WN, WROWS, THREADS, Tile::BN>(mine, Tile::template stage_pitch<T>(), smem, g.M, g.Nout, out, res, stats, br,
                                                                     mtile, ntile, m0, n0);
        }
    } else if constexpr (EPI == EPI_STORE) {
        EpiOperands<T, WROWS> eo;
        eo.fetch(res, (stats || br.gelu_bwd) ? br.y : nullptr, g.M, g.Nout, m0, n0, br.res_h, br.res_w, &br.map);
        const char* mine = ml.template stage_out<T>(smem);
        nt_epilogue_store<T, WM, WN, WROWS, THREADS, Tile::BN>(mine, Tile::template stage_pitch<T>(), smem, g.M, g.Nout, out,
                                                               res != nullptr, stats, br, eo, mtile, ntile, m0, n0);
    } else {
        constexpr int P = Tile::template stage_pitch<float>();
        const char* mine = ml.template stage_out<float>(smem);
        float* o = reinterpret_cast<float*>(out);
        const int n = n0 + lane;
        if constexpr (EPI == EPI_SLAB) o += (size_t)blockIdx.y * g.M * g.Nout;
        for (int row = 0; row < WROWS; ++row) {
            const int m = m0 + row;
            if (m < g.M && n < g.Nout) {
                const float v = *reinterpret_cast<const float*>(mine + row * P + lane * 4);
                if constexpr (EPI == EPI_SLAB) o[(size_t)m * g.Nout + n] = v;
                else atomicAdd(o + (size_t)m * g.Nout + n, v);
            }
        }
    }
}

// EPI_LEAN launches are PERSISTENT: gridDim.x = one workgroup per CU (this tile's LDS allows no second one) and every workgroup walks the
// tiles t = blockIdx.x, + gridDim.x, ...  A workgroup that ends cannot free its CU before its stores have drained, and the next one then
// starts from an empty memory pipeline; in the loop the drain runs under the next tile's first operand loads.  All other epilogues: one
// tile per workgroup.
template <typename T, int WM, int WN, int MT, int EPI>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN) / 4) void nt_kernel(NtGeom g, const void* __restrict__ a,
                                                           const void* __restrict__ b, void* __restrict__ out,
                                                           const void* __restrict__ res, float* __restrict__ stats,
                                                           EpiBnRed br, int mtiles, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t total = (uint32_t)(mtiles * ntiles);
    if constexpr (EPI == EPI_LEAN) {
        for (uint32_t t = blockIdx.x; t < total; t += gridDim.x) {
            nt_tile<T, WM, WN, MT, EPI>(g, a, b, out, res, stats, br, ntiles, t, total, smem);
            if (t + gridDim.x < total) {                     // the epilogue's LDS reads are done before the next tile's operand loads land
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // there; the global stores are NOT waited for
                __builtin_amdgcn_s_barrier();
            }
        }
    } else {
        nt_tile<T, WM, WN, MT, EPI>(g, a, b, out, res, stats, br, ntiles, blockIdx.x, total, smem);
    }
}

// FRHIP_NT_PERSIST=0: one workgroup per tile also for the lean launches (A/B switch)
static const int g_nt_persist = getenv("FRHIP_NT_PERSIST") ? atoi(getenv("FRHIP_NT_PERSIST")) : 1;
static int nt_cus() {
    static int cus[16] = {0};                      // per device: a process that drives a second GPU sizes its grids for THAT chip
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { (void)hipGetLastError(); return 256; }
    if (!cus[dev]) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cus[dev] = v;
    }
    return cus[dev];
}

template <typename T, int WM, int WN, int MT, int EPI>
static int nt_launch_cfg(const NtGeom& g, const void* a, const void* b, void* out, const void* res,
                         float* stats, const EpiBnRed& br, int splits, hipStream_t stream) {
    typedef NtTile<T, WM, WN, MT> Tile;
    const int mtiles = (g.M + Tile::BM - 1) / Tile::BM, ntiles = (g.Nout + Tile::BN - 1) / Tile::BN;
    const int lds = (EPI == EPI_STORE || EPI == EPI_LEAN) ? Tile::template lds_bytes<T>() : Tile::template lds_bytes<float>();     // ATOMIC / SLAB stage fp32
    auto kern = nt_kernel<T, WM, WN, MT, EPI>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            set_error("igemm_nt: cannot raise dynamic LDS to %d bytes", lds);
            return FRHIP_ELAUNCH;
        }
        attr_done = true;
    }
    dim3 grid(mtiles * ntiles, splits);
    if (EPI == EPI_LEAN && g_nt_persist) {
        // a multiple of 8: a workgroup's tiles stay on its XCD's share of the remap.  FRHIP_NT_PERSIST_WGS < CU count leaves CUs to the
        // other stream for the length of the launch (the 8-wave workgroups take the whole register file of the CU they sit on)
        static const int env_wgs = getenv("FRHIP_NT_PERSIST_WGS") ? atoi(getenv("FRHIP_NT_PERSIST_WGS")) : 0;
        const unsigned wgs = (unsigned)(env_wgs > 0 ? env_wgs : nt_cus()) & ~7u;
        if (wgs >= 8 && grid.x > wgs) grid.x = wgs;
    }
    hipLaunchKernelGGL(kern, grid, dim3(Tile::THREADS), lds, stream, g, a, b, out, res, stats, br, mtiles, ntiles);
    return check_launch("igemm_nt");
}

// Tile choice.  0 = automatic; tests / micro-benchmarks can force one with frhip_set_nt_tile().
//   1: 128x128 (4 waves 2x2)   2: 256x64 (4 waves 4x1)   3: 256x128 (8 waves 4x2)   4: 256x256 (8 waves 2x4, bf16 only)
static int g_nt_tile = 0;

static int nt_pick_tile(int dtype, const NtGeom& g) {
    // fp32 has the 128x128 and 256x64 tiles only: a forced 3 / 4 resolves to the tile nt_dispatch really launches (1), so that
    // the BM the callers size their partial-sum buffers with is the dispatched one
    if (g_nt_tile) return (dtype != FRHIP_DT_BF16 && g_nt_tile >= 3) ? 1 : g_nt_tile;
    if ((g.Nout % 128) != 0 && g.Nout <= 256) return 2;
    // measured on MI355X (tools/bench_kernels.py, B=512): 256x256 beats 128x128 by 15-25 % once Cout % 256 == 0
    if (dtype == FRHIP_DT_BF16 && (g.Nout % 256) == 0 && g.M >= 256 * 64) return 4;
    return 1;
}

static const EpiBnRed NO_BNRED = {nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, {0, 0, 0, 0, 0, 0}, nullptr, nullptr, 0};

static int nt_dispatch(int dtype, const NtGeom& g, const void* a, const void* b, void* out, const void* res,
                       float* stats, const EpiBnRed& br, int splits, bool atomic, hipStream_t stream) {
    const int tile = nt_pick_tile(dtype, g);
#define NT_GO(T, WM, WN, MT)                                                                              \
    return atomic ? nt_launch_cfg<T, WM, WN, MT, EPI_ATOMIC>(g, a, b, out, res, stats, br, splits, stream)     \
                  : nt_launch_cfg<T, WM, WN, MT, EPI_STORE>(g, a, b, out, res, stats, br, splits, stream)
    if (dtype == FRHIP_DT_BF16) {
        switch (tile) {
            case 2: NT_GO(bf16_t, 4, 1, 4);
            case 3: NT_GO(bf16_t, 4, 2, 4);
            case 4: if (!atomic && splits == 1 && g_epi_lean && nt_lean_ok(true, g.M, g.Nout, br, stats))
                        return nt_launch_cfg<bf16_t, 2, 4, 8, EPI_LEAN>(g, a, b, out, res, stats, br, splits, stream);
                    if (!atomic) return nt_launch_cfg<bf16_t, 2, 4, 8, EPI_STORE>(g, a, b, out, res, stats, br, splits, stream);
                    NT_GO(bf16_t, 4, 2, 4);
            default: NT_GO(bf16_t, 2, 2, 4);
        }
    }
    if (dtype == FRHIP_DT_F32) {
        switch (tile) {
            case 2: NT_GO(float, 4, 1, 4);
            default: NT_GO(float, 2, 2, 4);
        }
    }
#undef NT_GO
    set_error("igemm_nt: bad dtype %d", dtype);
    return FRHIP_EINVAL;
}

static int esize(int dtype) { return dtype == FRHIP_DT_BF16 ? 2 : 4; }

static int fill_geom(NtGeom& g, int dtype, int n, int h, int w, int c, int ho, int wo, int k, int r, int s,
                     int stride, int pad, int mode, const char* who) {
    const int bke = NT_ROWB / esize(dtype);
    if (n <= 0 || h <= 0 || w <= 0 || c <= 0 || k <= 0 || (c % bke) != 0 || (k % 8) != 0 || (stride != 1 && stride != 2)) {
        set_error("%s: unsupported shape n=%d h=%d w=%d c=%d k=%d stride=%d (c must be a multiple of %d, k of 8)",
                  who, n, h, w, c, k, stride, bke);
        return FRHIP_EINVAL;
    }
    const long long a_bytes = 1LL * n * h * w * c * esize(dtype), b_bytes = 1LL * k * r * s * c * esize(dtype);
    if (a_bytes > 0x7fffffffLL || b_bytes > 0x7fffffffLL || 1LL * n * ho * wo > 0x7fffffffLL) {
        set_error("%s: tensor exceeds the 2 GiB buffer-addressing window", who);
        return FRHIP_EINVAL;
    }
    g.H = h; g.W = w; g.C = c; g.Ho = ho; g.Wo = wo; g.R = r; g.S = s; g.stride = stride; g.pad = pad; g.mode = mode;
    g.M = n * ho * wo; g.Nout = k; g.Ktot = r * s * c;
    g.ksteps = r * s * (c / bke); g.ksteps_per_split = g.ksteps;
    g.a_bytes = (uint32_t)a_bytes; g.b_bytes = (uint32_t)b_bytes;
    g.par_a = -1; g.par_b = -1; g.hc = 0; g.wc = 0; g.par_r0 = 0; g.par_s0 = 0;
    return FRHIP_OK;
}

// Stride-2 3x3 (pad 1) data-gradient as four parity-class launches (see NtGeom::par_a): 1 + 2 + 2 + 4 taps over a
// quarter of the rows each, instead of 9 taps over all rows of which 3 in 4 gather only zeros.
static bool dgrad_by_parity(const NtGeom& g) { return g.mode == 1 && g.stride == 2 && g.R == 3 && g.S == 3 && g.pad == 1; }

static int nt_block_rows(int dtype, const NtGeom& g) {
    static const int bm_of[5] = {128, 128, 256, 256, 256};
    return bm_of[nt_pick_tile(dtype, g)];
}

// g: the full-grid geometry of the data-gradient.  Fills the geometry / output map of class (a, b); returns its row count
static int parity_class(const NtGeom& g, int dtype, int a, int b, NtGeom& gc, OutMap& map) {
    gc = g;
    gc.par_a = a; gc.par_b = b;
    gc.hc = (g.Ho - a + 1) / 2; gc.wc = (g.Wo - b + 1) / 2;
    gc.par_r0 = (a + g.pad) & 1; gc.par_s0 = (b + g.pad) & 1;
    const int nr = (g.R - gc.par_r0 + 1) / 2, ns = (g.S - gc.par_s0 + 1) / 2;
    const int n_img = g.M / (g.Ho * g.Wo);
    gc.M = n_img * gc.hc * gc.wc;
    gc.ksteps = nr * ns * (g.C / (NT_ROWB / (dtype == FRHIP_DT_BF16 ? 2 : 4)));
    gc.ksteps_per_split = gc.ksteps;
    map.hc = gc.hc; map.wc = gc.wc; map.ho = g.Ho; map.wo = g.Wo; map.a = a; map.b = b;
    return gc.M;
}

}  // namespace frhip

using namespace frhip;

extern "C" int frhip_nt_block_m(int nout) {
    // smallest BM any tile choice uses for this width: callers size the BN-partial buffer with it
    return ((nout % 128) == 0 || nout > 256) ? 128 : 256;
}
extern "C" int frhip_conv_stat_rows(int dtype, int m, int k, int h, int w, int c, int r, int s, int stride, int pad) {
    // rows of the stats_partial buffer frhip_conv_fwd writes for an output of m pixels x k channels
    if (halo_applicable(dtype, h, w, c, k, r, s, stride, pad)) return halo_stat_rows(dtype, m, w, c, k, +1);
    NtGeom g; g.M = m; g.Nout = k;
    static const int bm_of[5] = {128, 128, 256, 256, 256};
    const int bm = bm_of[nt_pick_tile(dtype, g)];
    return (m + bm - 1) / bm;
}
extern "C" int frhip_set_nt_tile(int tile) { const int old = g_nt_tile; g_nt_tile = tile; return old; }

extern "C" int frhip_conv_fwd(int dtype, const void* x, const void* w, void* y, float* stats_partial,
                              int n, int h, int wd, int c, int k, int r, int s, int stride, int pad,
                              hipStream_t stream) {
    NtGeom g;
    const int ho = (h + 2 * pad - r) / stride + 1, wo = (wd + 2 * pad - s) / stride + 1;
    int rc = fill_geom(g, dtype, n, h, wd, c, ho, wo, k, r, s, stride, pad, 0, "frhip_conv_fwd");
    if (rc) return rc;
    if (halo_applicable(dtype, h, wd, c, k, r, s, stride, pad))
        return halo_run(dtype, x, w, y, nullptr, stats_partial, NO_BNRED, n, h, wd, c, k, +1, stream);
    return nt_dispatch(dtype, g, x, w, y, nullptr, stats_partial, NO_BNRED, 1, false, stream);
}

extern "C" int frhip_conv_fwd_affine(int dtype, const void* x, const void* w, void* y, const float* scale, const float* shift, int relu,
                                     const void* residual, int n, int h, int wd, int c, int k, int r, int s, int stride, int pad,
                                     hipStream_t stream) {
    // y = [relu](conv(x, w) * scale[k] + shift[k] + residual): eval-mode BatchNorm (frhip_bn_eval_affine) folded into the store epilogue
    if (!scale || !shift) { set_error("frhip_conv_fwd_affine: scale and shift are required"); return FRHIP_EINVAL; }
    NtGeom g;
    const int ho = (h + 2 * pad - r) / stride + 1, wo = (wd + 2 * pad - s) / stride + 1;
    int rc = fill_geom(g, dtype, n, h, wd, c, ho, wo, k, r, s, stride, pad, 0, "frhip_conv_fwd_affine");
    if (rc) return rc;
    if (halo_applicable(dtype, h, wd, c, k, r, s, stride, pad) && halo_lean_applies(dtype, n, h, wd, c, k, +1)) {
        EpiBnRed br = NO_BNRED;
        br.aff_scale = scale; br.aff_shift = shift; br.aff_relu = relu;
        return halo_run(dtype, x, w, y, residual, nullptr, br, n, h, wd, c, k, +1, stream);
    }
    // not a launch of whole tiles on a lean kernel (strided / 1x1 convolutions, small batches, fp32 validation mode): the unfused pair,
    // the BatchNorm-apply pass in place on the convolution's output -- same values
    rc = frhip_conv_fwd(dtype, x, w, y, nullptr, n, h, wd, c, k, r, s, stride, pad, stream);
    if (rc) return rc;
    return frhip_bn_apply(dtype, y, scale, shift, residual, nullptr, nullptr, relu, y, n * ho * wo, k, stream);
}

extern "C" int frhip_conv_bnrelu_fusable(int dtype, int h, int wd, int c, int k, int r, int s, int stride, int pad) {
    return halo_xf_applicable(dtype, h, wd, c, k, r, s, stride, pad) ? 1 : 0;
}

extern "C" int frhip_conv_fwd_bnrelu(int dtype, const void* x, const float* in_scale, const float* in_shift, const void* w, void* y,
                                     float* stats_partial, void* act_out, int n, int h, int wd, int c, int k, int r, int s, int stride,
                                     int pad, hipStream_t stream) {
    // y = conv(relu(x * in_scale[c] + in_shift[c]), w): BatchNorm-apply + ReLU of the operand inside the kernel (bn1 -> relu -> conv2);
    // act_out (optional, shaped like x): the activated tensor, written on the way for the backward pass
    if (!in_scale || !in_shift || !halo_xf_applicable(dtype, h, wd, c, k, r, s, stride, pad)) {
        set_error("frhip_conv_fwd_bnrelu: not fusable for this shape (ask frhip_conv_bnrelu_fusable first)");
        return FRHIP_EINVAL;
    }
    return halo_run(dtype, x, w, y, nullptr, stats_partial, NO_BNRED, n, h, wd, c, k, +1, stream, in_scale, in_shift, act_out);
}

static int dgrad_run(int dtype, const void* dy, const void* wt, void* dx, const void* residual, float* stats,
                     const EpiBnRed& br, int n, int h, int wd, int c, int k, int r, int s, int stride, int pad,
                     hipStream_t stream, const char* who) {
    // dx is [n,h,wd,c]; dy is [n,ho,wo,k]; wt is the transposed pack [c][r][s][k].
    NtGeom g;
    const int ho = (h + 2 * pad - r) / stride + 1, wo = (wd + 2 * pad - s) / stride + 1;
    int rc = fill_geom(g, dtype, n, ho, wo, k, h, wd, c, r, s, stride, pad, 1, who);
    if (rc) return rc;
    if (halo_applicable(dtype, h, wd, k, c, r, s, stride, pad))       // gathered tensor = dy [n,h,w,k] -> dx [n,h,w,c]
        return halo_run(dtype, dy, wt, dx, residual, stats, br, n, h, wd, k, c, -1, stream);
    if (dgrad_by_parity(g)) {
        float* st = stats;
        for (int cls = 0; cls < 4; ++cls) {
            NtGeom gc; EpiBnRed bc = br;
            if (parity_class(g, dtype, cls >> 1, cls & 1, gc, bc.map) == 0) continue;
            rc = nt_dispatch(dtype, gc, dy, wt, dx, residual, st, bc, 1, false, stream);
            if (rc) return rc;
            if (st) st += (size_t)((gc.M + nt_block_rows(dtype, gc) - 1) / nt_block_rows(dtype, gc)) * 2 * gc.Nout;
        }
        return FRHIP_OK;
    }
    return nt_dispatch(dtype, g, dy, wt, dx, residual, stats, br, 1, false, stream);
}

extern "C" int frhip_conv_dgrad(int dtype, const void* dy, const void* wt, void* dx, const void* residual,
                                int n, int h, int wd, int c, int k, int r, int s, int stride, int pad,
                                hipStream_t stream) {
    return dgrad_run(dtype, dy, wt, dx, residual, nullptr, NO_BNRED, n, h, wd, c, k, r, s, stride, pad, stream, "frhip_conv_dgrad");
}

extern "C" int frhip_dgrad_stat_rows(int dtype, int n, int h, int wd, int c, int k, int r, int s, int stride, int pad) {
    const int m = n * h * wd;
    if (halo_applicable(dtype, h, wd, k, c, r, s, stride, pad)) return halo_stat_rows(dtype, m, wd, k, c, -1);
    NtGeom g; g.M = m; g.Nout = c;
    g.mode = 1; g.stride = stride; g.R = r; g.S = s; g.pad = pad; g.Ho = h; g.Wo = wd; g.C = k;
    if (dgrad_by_parity(g)) {
        int rows = 0;
        for (int cls = 0; cls < 4; ++cls) {
            NtGeom gc; OutMap map;
            const int mc = parity_class(g, dtype, cls >> 1, cls & 1, gc, map);
            rows += (mc + nt_block_rows(dtype, gc) - 1) / nt_block_rows(dtype, gc);
        }
        return rows;
    }
    const int bm = nt_block_rows(dtype, g);
    return (m + bm - 1) / bm;
}

extern "C" int frhip_conv_dgrad_fused(int dtype, const void* dy, const void* wt, void* dx, const void* residual,
                                      int residual_stride, const void* y_bn, const float* mean, const float* invstd,
                                      const float* mask_scale, const float* mask_shift, float* stats_partial, int n, int h,
                                      int wd, int c, int k, int r, int s, int stride, int pad, hipStream_t stream) {
    if (residual_stride != 1 && residual_stride != 2) { set_error("frhip_conv_dgrad_fused: residual_stride must be 1 or 2"); return FRHIP_EINVAL; }
    if (y_bn && (!mean || !invstd || !stats_partial || (mask_scale && !mask_shift))) {
        set_error("frhip_conv_dgrad_fused: y_bn needs mean, invstd and stats_partial");
        return FRHIP_EINVAL;
    }
    EpiBnRed br = {y_bn, mean, invstd, mask_scale, mask_shift, 0, 0, {0, 0, 0, 0, 0, 0}, nullptr, nullptr, 0};
    if (residual && residual_stride == 2) { br.res_h = h; br.res_w = wd; }
    return dgrad_run(dtype, dy, wt, dx, residual, y_bn ? stats_partial : nullptr, br, n, h, wd, c, k, r, s, stride, pad, stream,
                     "frhip_conv_dgrad_fused");
}

extern "C" int frhip_conv_dgrad_fused_rs(int dtype, const void* dy, const void* wt, void* dx, const void* residual,
                                         int residual_stride, const void* y_bn, const float* mean, const float* invstd,
                                         const float* mask_scale, const float* mask_shift, const float* rowscale, int rows_per,
                                         float keep_scale, float* stats_partial, int n, int h, int wd, int c, int k, int r, int s,
                                         int stride, int pad, hipStream_t stream) {
    // frhip_conv_dgrad_fused whose BatchNorm sits under stochastic depth: the sums describe dx * rowscale[row / rows_per]
    if (residual_stride != 1 && residual_stride != 2) { set_error("frhip_conv_dgrad_fused_rs: residual_stride must be 1 or 2"); return FRHIP_EINVAL; }
    if (!y_bn || !mean || !invstd || !stats_partial || (mask_scale && !mask_shift) || !rowscale || rows_per <= 0 ||
        (long long)n * h * wd % rows_per != 0) {
        set_error("frhip_conv_dgrad_fused_rs: y_bn, mean, invstd, stats_partial, rowscale and a rows_per that divides the rows are required");
        return FRHIP_EINVAL;
    }
    EpiBnRed br = {y_bn, mean, invstd, mask_scale, mask_shift, 0, 0, {0, 0, 0, 0, 0, 0}, nullptr, nullptr, 0};
    br.rowkeep = rowscale; br.rows_per = rows_per; br.keep_scale = keep_scale;
    if (residual && residual_stride == 2) { br.res_h = h; br.res_w = wd; }
    return dgrad_run(dtype, dy, wt, dx, residual, stats_partial, br, n, h, wd, c, k, r, s, stride, pad, stream, "frhip_conv_dgrad_fused_rs");
}

extern "C" int frhip_conv_dgrad_bnred(int dtype, const void* dy, const void* wt, void* dx, const void* residual,
                                      const void* y_bn, const float* mean, const float* invstd, const float* mask_scale,
                                      const float* mask_shift, float* stats_partial, int n, int h, int wd, int c, int k,
                                      int r, int s, int stride, int pad, hipStream_t stream) {
    if (!y_bn || !mean || !invstd || !stats_partial || (mask_scale && !mask_shift)) {
        set_error("frhip_conv_dgrad_bnred: y_bn, mean, invstd and stats_partial are required");
        return FRHIP_EINVAL;
    }
    return frhip_conv_dgrad_fused(dtype, dy, wt, dx, residual, 1, y_bn, mean, invstd, mask_scale, mask_shift, stats_partial,
                                  n, h, wd, c, k, r, s, stride, pad, stream);
}

extern "C" int frhip_linear_fwd(int dtype, const void* a, const void* w, const float* bias, void* out, void* act_out,
                                float* stats_partial, int m, int n, int k, hipStream_t stream) {
    // out[m][n] = sum_k a[m][k] * w[n][k] + bias[n]; act_out = gelu(out); stats_partial as frhip_conv_fwd (a 1x1 conv)
    NtGeom g;
    int rc = fill_geom(g, dtype, m, 1, 1, k, 1, 1, n, 1, 1, 1, 0, 0, "frhip_linear_fwd");
    if (rc) return rc;
    EpiBnRed br = NO_BNRED;
    br.bias = bias; br.act = act_out;
    return nt_dispatch(dtype, g, a, w, out, nullptr, stats_partial, br, 1, false, stream);
}

extern "C" int frhip_linear_dgrad_gelu(int dtype, const void* dy, const void* wt, const void* pre, void* dx,
                                       float* stats_partial, int m, int n, int k, hipStream_t stream) {
    // dx[m][n] = (sum_k dy[m][k] * wt[n][k]) * gelu'(pre[m][n]); stats_partial[.][0][n] sums to the column sums of dx
    if (!pre) { set_error("frhip_linear_dgrad_gelu: the saved pre-activation is required"); return FRHIP_EINVAL; }
    NtGeom g;
    int rc = fill_geom(g, dtype, m, 1, 1, k, 1, 1, n, 1, 1, 1, 0, 0, "frhip_linear_dgrad_gelu");
    if (rc) return rc;
    EpiBnRed br = NO_BNRED;
    br.y = pre; br.gelu_bwd = 1;
    return nt_dispatch(dtype, g, dy, wt, dx, nullptr, stats_partial, br, 1, false, stream);
}

namespace frhip {
// out[i] = bias[i % n] + sum over the K splits of slabs[s][i], in split order (deterministic)
__global__ __launch_bounds__(256) void nt_slab_reduce_kernel(const float* __restrict__ slabs, int splits, size_t elems,
                                                             const float* __restrict__ bias, int n, float* __restrict__ out) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= elems) return;
    f32x4_t acc = *reinterpret_cast<const f32x4_t*>(slabs + i);
    for (int s = 1; s < splits; ++s) acc += *reinterpret_cast<const f32x4_t*>(slabs + (size_t)s * elems + i);
    if (bias) acc += *reinterpret_cast<const f32x4_t*>(bias + (i % (size_t)n));
    *reinterpret_cast<f32x4_t*>(out + i) = acc;
}
}  // namespace frhip

extern "C" int frhip_gemm_nt_splitk(int dtype, const void* a, const void* b, const float* bias, float* out, int m, int n,
                                    int k, int splits, float* slabs, size_t slab_bytes, hipStream_t stream) {
    // out[m][n] fp32 = sum_k a[m][k] * b[n][k] + bias[n]: K split `splits` ways, every split stores a private fp32 slab
    // (plain stores), one pass adds the slabs in split order -> run-to-run identical (no float atomics)
    NtGeom g;
    int rc = fill_geom(g, dtype, m, 1, 1, k, 1, 1, n, 1, 1, 1, 0, 0, "frhip_gemm_nt_splitk");
    if (rc) return rc;
    if (n % 4) { set_error("frhip_gemm_nt_splitk: n must be a multiple of 4"); return FRHIP_EINVAL; }
    if (splits < 1) splits = 1;
    if (splits > g.ksteps) splits = g.ksteps;
    g.ksteps_per_split = (g.ksteps + splits - 1) / splits;
    splits = (g.ksteps + g.ksteps_per_split - 1) / g.ksteps_per_split;
    const size_t elems = (size_t)m * n;
    if (!slabs || slab_bytes < elems * 4 * (size_t)splits) { set_error("frhip_gemm_nt_splitk: workspace too small (%zu bytes needed)", elems * 4 * (size_t)splits); return FRHIP_EINVAL; }
    if (dtype == FRHIP_DT_BF16) rc = nt_launch_cfg<bf16_t, 2, 2, 4, EPI_SLAB>(g, a, b, slabs, nullptr, nullptr, NO_BNRED, splits, stream);
    else if (dtype == FRHIP_DT_F32) rc = nt_launch_cfg<float, 2, 2, 4, EPI_SLAB>(g, a, b, slabs, nullptr, nullptr, NO_BNRED, splits, stream);
    else { set_error("frhip_gemm_nt_splitk: bad dtype %d", dtype); return FRHIP_EINVAL; }
    if (rc) return rc;
    hipLaunchKernelGGL(nt_slab_reduce_kernel, dim3((unsigned)((elems / 4 + 255) / 256)), dim3(256), 0, stream, slabs, splits, elems, bias, n, out);
    return check_launch("frhip_gemm_nt_splitk/reduce");
}

extern "C" int frhip_gemm_nt(int dtype, const void* a, const void* b, void* out, int m, int n, int k,
                             int splits, int atomic_f32, hipStream_t stream) {
    // out[m][n] = sum_k a[m][k] * b[n][k].  atomic_f32 == 0: out has dtype and is overwritten;
    // atomic_f32 == 1: out is fp32, zeroed by the caller, K is split and partial sums are added atomically.
    NtGeom g;
    int rc = fill_geom(g, dtype, m, 1, 1, k, 1, 1, n, 1, 1, 1, 0, 0, "frhip_gemm_nt");
    if (rc) return rc;
    if (!atomic_f32 || splits < 1) splits = 1;
    if (splits > g.ksteps) splits = g.ksteps;
    g.ksteps_per_split = (g.ksteps + splits - 1) / splits;
    splits = (g.ksteps + g.ksteps_per_split - 1) / g.ksteps_per_split;
    return nt_dispatch(dtype, g, a, b, out, nullptr, nullptr, NO_BNRED, splits, atomic_f32 != 0, stream);
}
