// Implicit-GEMM "NT" main loop for gfx950:  OUT[m][n] = sum_k A[m][k] * B[n][k]
//   A rows are gathered from an NHWC activation tensor by convolution geometry (one filter tap x one
//   128-byte channel slice per K step), B rows are K-contiguous packed weights [n][tap][c].
//   Covers: conv forward, conv data-gradient (transposed-stride gather), plain GEMM (R=S=1, H=W=1).
//
// Tile: 4 waves (256 threads), each wave owns a 64 (m) x 64 (n) sub-tile = 4x4 MFMA 16x16 tiles.
//   WM x WN waves: 2x2 -> 128x128, 4x1 -> 256x64.
// LDS: two stages of [BM + BN] rows x 128 bytes, filled by LDS-DMA (buffer_load ... lds, 16 B per lane,
//   out-of-range lanes read 0 = convolution zero padding), 16-byte chunks XOR-swizzled by (row & 7) on the
//   SOURCE side so the lane-linear LDS image is conflict-free for ds_read_b128.
// Loop (one barrier per K step):  issue DMA for step t+1 -> MFMA on step t -> vmcnt(0) -> s_barrier.
// MFMA operand roles are swapped (weights = "A", pixels = "B") so a lane ends up holding 4 consecutive
// output CHANNELS of one pixel: the epilogue packs them, stages the wave's 64x64 tile through LDS and
// writes whole 128/256-byte rows.
#pragma once
#include "common.h"

namespace frhip {

struct NtGeom {
    // activation tensor (the gathered operand)
    int H, W, C;            // spatial size and channels of the tensor being gathered
    int Ho, Wo;             // spatial size of the GEMM rows (output pixels)
    int R, S, stride, pad;
    int mode;               // 0: forward gather (hi = ho*stride - pad + r); 1: data-grad gather (hi = (ho + pad - r)/stride)
    int M, Nout, Ktot;      // GEMM sizes: rows, cols, taps*C
    int ksteps, ksteps_per_split;
    uint32_t a_bytes, b_bytes;
    // stride-2 data-gradient, one PARITY CLASS per launch (par_a >= 0): the GEMM rows are the output pixels
    // (2i + par_a, 2j + par_b), i < hc, j < wc, and only the taps that reach them (r = par_r0, par_r0 + 2, ...; likewise s)
    // are walked -- the full-grid formulation multiplies 3 of 4 (row, tap) pairs by zeros.
    int par_a, par_b, hc, wc, par_r0, par_s0;
};

// dense output row of GEMM row m (identity unless a parity class is active)
struct OutMap {
    int hc, wc, ho, wo, a, b;        // wc == 0: identity
    __device__ __forceinline__ long long row(int m) const {
        if (wc == 0) return m;
        const int per = hc * wc;
        const int img = m / per, rem = m - img * per;
        const int i = rem / wc, j = rem - i * wc;
        return ((long long)img * ho + 2 * i + a) * wo + 2 * j + b;
    }
};

constexpr int NT_ROWB = 128;                 // bytes per LDS row = one K step

// Optional BatchNorm-backward reduction fused into the store epilogue of a data-gradient kernel: the tensor being
// written (dx) is the upstream gradient of a BatchNorm whose saved input is `y` (same shape).  Instead of
// {sum x, sum x^2} the per-tile partials become { sum d, sum d * (y - mean) * invstd } with
// d = dx * (y*mscale + mshift > 0) when the BN is followed by a ReLU (mscale != NULL), d = dx otherwise --
// exactly what frhip_bn_bwd_reduce computes in a separate pass over dx and y.
struct EpiBnRed {
    const void* y;
    const float* mean;
    const float* invstd;
    const float* mscale;
    const float* mshift;
    // residual layout: res_w == 0: dense, same shape as the output.  res_w > 0: the output is [n][res_h][res_w][c] and the
    // residual is the COMPACT gradient of a stride-2 1x1 shortcut, [n][(res_h+1)/2][(res_w+1)/2][c]: it lands on the even
    // pixels only (everything else receives nothing), so the zero-stuffed full-size tensor is never built.
    int res_h, res_w;
    // linear-layer epilogue (forward only, no BN-backward partials): out = gemm + bias[n], rounded to T and stored;
    // act (optional) = gelu(out) of the STORED value (exact erf form); forward statistics then describe out
    OutMap map;
    const float* bias;
    void* act;
    // gelu_bwd != 0 (data-gradient of a Linear that feeds a GELU): out = gemm o gelu'(y), y = the saved pre-activation
    // (same shape as out); the forward-style statistics then carry the column sums of out = the bias gradient
    int gelu_bwd;
    // eval-mode BatchNorm folded into the store (inference, model/FR_PartialFC.py:205-211: encoder.eval()): the convolution result,
    // rounded to T as the unfused pair would store it, goes through out = [relu](v * aff_scale[n] + aff_shift[n] + residual) -- the
    // arithmetic of bn_apply_kernel -- and only that tensor is written.  No statistics in this mode.  LEAN kernels only: launches
    // that are not made of whole tiles run conv + frhip_bn_apply (frhip_conv_fwd_affine decides).
    const float* aff_scale;
    const float* aff_shift;
    int aff_relu;
    // stochastic depth around the BatchNorm whose backward sums ride here (nets/AlterNet_SwinV2_FAN.py: x + drop_path(norm(f(x)))): the
    // gradient that enters the BatchNorm is dx * rowkeep[row / rows_per] (0 for a dropped sample, keep_scale = 1 / keep-probability for
    // a kept one).  The sums are taken over the kept samples' rows and multiplied by keep_scale -- what frhip_bn_bwd_reduce_rs computes in a
    // pass of its own.  dx itself is stored unscaled.  rowkeep == NULL: no stochastic depth.
    const float* rowkeep;
    int rows_per;
    float keep_scale;
};

// WM x WN waves; each wave owns (MT*16) pixel rows x 64 channels (4 MFMA tiles wide).
template <typename T, int WM, int WN, int MT = 4>
struct NtTile {
    static constexpr int WAVES = WM * WN, THREADS = 64 * WAVES;
    static constexpr int WROWS = MT * 16;                         // pixel rows per wave
    static constexpr int BM = WM * WROWS, BN = WN * 64;
    static constexpr int BKE = NT_ROWB / (int)sizeof(T);          // K elements per step
    static constexpr int STAGE_BYTES = (BM + BN) * NT_ROWB;
    static constexpr int A_PIECES = BM / 8 / WAVES, B_PIECES = BN / 8 / WAVES;   // 1-KiB (8-row) DMA pieces per wave
    static_assert(A_PIECES * 8 * WAVES == BM && B_PIECES * 8 * WAVES == BN, "tile rows must split evenly over the waves");
    template <typename TS> static constexpr int stage_pitch() { return 64 * (int)sizeof(TS) + 16; }
    template <typename TS> static constexpr int lds_bytes() {
        return (2 * STAGE_BYTES > WAVES * WROWS * stage_pitch<TS>()) ? 2 * STAGE_BYTES : WAVES * WROWS * stage_pitch<TS>();
    }
};

// Per-thread bookkeeping of the rows this thread DMA-loads.
template <int NPIECE>
struct RowSet {
    int pixbase[NPIECE];     // n*H*W (element-row index of image n), or -1 when the GEMM row is out of range
    int hb[NPIECE], wb[NPIECE];
};

template <typename T, int WM, int WN, int MT = 4>
struct NtMainloop {
    typedef NtTile<T, WM, WN, MT> Tile;
    typedef typename Mma<T>::Frag Frag;
    static constexpr int BM = Tile::BM, BN = Tile::BN, BKE = Tile::BKE;

    // acc[nt][mt]: D rows = channels (nt*16 + 4*(lane>>4) + reg), D col = pixel (mt*16 + (lane&15))
    f32x4_t acc[4][MT];

    __device__ __forceinline__ void run(const NtGeom& g, const void* __restrict__ a_ptr,
                                        const void* __restrict__ b_ptr, char* smem,
                                        int mtile, int ntile, int ks_begin, int ks_end) {
        const int lane = lane_id();
        const int wave = wave_id();
        const int wm = wave / WN, wn = wave % WN;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (ks_begin >= ks_end) return;

        const __amdgpu_buffer_rsrc_t ra = make_rsrc(a_ptr, g.a_bytes);
        const __amdgpu_buffer_rsrc_t rb = make_rsrc(b_ptr, g.b_bytes);

        // ---- rows this thread stages.  piece q covers tile rows 8q..8q+7; lane -> row 8q + (lane>>3),
        //      physical chunk (lane&7), logical (source) chunk (lane&7) ^ (row&7).
        const int sub = lane >> 3;
        const uint32_t chunk_bytes = (uint32_t)(((lane & 7) ^ sub) * 16);
        RowSet<Tile::A_PIECES> ar;
        const int rowlen = g.par_a >= 0 ? g.wc : g.Wo;
        const int HoWo = g.par_a >= 0 ? g.hc * g.wc : g.Ho * g.Wo;
#pragma unroll
        for (int j = 0; j < Tile::A_PIECES; ++j) {
            const int row = (wave * Tile::A_PIECES + j) * 8 + sub;
            const int m = mtile * BM + row;
            if (m < g.M) {
                const int n = m / HoWo, rem = m - n * HoWo;
                int ho = rem / rowlen, wo = rem - ho * rowlen;
                if (g.par_a >= 0) { ho = 2 * ho + g.par_a; wo = 2 * wo + g.par_b; }
                ar.pixbase[j] = n * g.H * g.W;
                if (g.mode == 0) { ar.hb[j] = ho * g.stride - g.pad; ar.wb[j] = wo * g.stride - g.pad; }
                else             { ar.hb[j] = ho + g.pad;            ar.wb[j] = wo + g.pad; }
            } else {
                ar.pixbase[j] = -1; ar.hb[j] = 0; ar.wb[j] = 0;
            }
        }
        uint32_t brow_off[Tile::B_PIECES];
#pragma unroll
        for (int j = 0; j < Tile::B_PIECES; ++j) {
            const int row = (wave * Tile::B_PIECES + j) * 8 + sub;
            const int n = ntile * BN + row;
            brow_off[j] = (n < g.Nout) ? (uint32_t)n * (uint32_t)g.Ktot * (uint32_t)sizeof(T) + chunk_bytes
                                       : OOB_OFFSET;
        }

        const int cchunks = g.C / BKE;
        int tap = ks_begin / cchunks;
        int c0 = (ks_begin - tap * cchunks) * BKE;
        int fr = tap / g.S, fs = tap - fr * g.S;
        const int tstep = g.par_a >= 0 ? 2 : 1, fs0 = g.par_a >= 0 ? g.par_s0 : 0;
        if (g.par_a >= 0) {                                  // `tap` counted the class's taps: (ir, is) -> (r0 + 2ir, s0 + 2is)
            const int ns = (g.S - g.par_s0 + 1) >> 1;
            const int ir = tap / ns, is = tap - ir * ns;
            fr = g.par_r0 + 2 * ir; fs = g.par_s0 + 2 * is;
            tap = fr * g.S + fs;
        }

        auto stage = [&](int buf) {
            char* sa = smem + buf * Tile::STAGE_BYTES;
            char* sb = sa + BM * NT_ROWB;
#pragma unroll
            for (int j = 0; j < Tile::A_PIECES; ++j) {
                int hi, wi; bool ok = ar.pixbase[j] >= 0;
                if (g.mode == 0) {
                    hi = ar.hb[j] + fr; wi = ar.wb[j] + fs;
                } else {
                    const int th = ar.hb[j] - fr, tw = ar.wb[j] - fs;
                    ok = ok && th >= 0 && tw >= 0;
                    if (g.stride == 2) { ok = ok && !((th | tw) & 1); hi = th >> 1; wi = tw >> 1; }
                    else { hi = th; wi = tw; }
                }
                ok = ok && (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W;
                const uint32_t off = ok ? (uint32_t)((ar.pixbase[j] + hi * g.W + wi) * g.C + c0) * (uint32_t)sizeof(T) + chunk_bytes
                                        : OOB_OFFSET;
                glds16(ra, sa + (wave * Tile::A_PIECES + j) * 1024, off);
            }
            const uint32_t kb = (uint32_t)(tap * g.C + c0) * (uint32_t)sizeof(T);
#pragma unroll
            for (int j = 0; j < Tile::B_PIECES; ++j) {
                const uint32_t off = brow_off[j] == OOB_OFFSET ? OOB_OFFSET : brow_off[j] + kb;
                glds16(rb, sb + (wave * Tile::B_PIECES + j) * 1024, off);
            }
            // advance to the next K step
            c0 += BKE;
            if (c0 == g.C) { c0 = 0; fs += tstep; if (fs >= g.S) { fs = fs0; fr += tstep; } tap = fr * g.S + fs; }
        };

        // fragment read addresses (bytes inside a stage): row = base + i, chunk (g + 4s) ^ (row & 7)
        const int fi = lane & 15, fg = lane >> 4;
        const int xoff = ((wm * Tile::WROWS + fi) * NT_ROWB);        // pixel rows (MFMA "B" operand)
        const int woff = BM * NT_ROWB + ((wn * 64 + fi) * NT_ROWB);  // weight rows (MFMA "A" operand)
        const int sw = fi & 7;   // (row & 7) == (fi & 7) because tile bases are multiples of 16

        auto compute = [&](int buf) {
            const char* base = smem + buf * Tile::STAGE_BYTES;
#pragma unroll
            for (int s = 0; s < RowFrag<T>::KSUB; ++s) {
                Frag xf[MT], wf[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) wf[t] = RowFrag<T>::load(base + woff + t * 16 * NT_ROWB, fg, sw, s);
#pragma unroll
                for (int t = 0; t < MT; ++t) xf[t] = RowFrag<T>::load(base + xoff + t * 16 * NT_ROWB, fg, sw, s);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) Mma<T>::run(wf[nt], xf[mt], acc[nt][mt]);
            }
        };

        stage(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int cur = 0;
        for (int ks = ks_begin; ks < ks_end - 1; ++ks) {
            stage(cur ^ 1);
            compute(cur);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            cur ^= 1;
        }
        compute(cur);
    }

    // rows [64 h, 64 h + 64) of this wave's tile (MT == 8) to rows 0 .. 63 of its staging area, as bf16: the upper 64 rows of the area stay free
    // for a second operand tile (lean BatchNorm-backward partials on the 256 x 256 tile)
    __device__ __forceinline__ char* stage_half(char* smem, int h) {
        constexpr int P = Tile::template stage_pitch<bf16_t>();
        const int lane = lane_id();
        const int fi = lane & 15, fg = lane >> 4;
        char* mine = smem + wave_id() * Tile::WROWS * P;
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                bf16x4_t v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (bf16_t)(h ? acc[nt][(MT > 4 ? 4 : 0) + mt][e] : acc[nt][mt][e]);
                *reinterpret_cast<bf16x4_t*>(mine + (mt * 16 + fi) * P + (nt * 16 + 4 * fg) * 2) = v;
            }
        __syncthreads();
        return mine;
    }

    // Write this wave's (MT*16)x64 tile to its private LDS staging area as TS (row = pixel, 64 channels).
    template <typename TS>
    __device__ __forceinline__ char* stage_out(char* smem) {
        constexpr int P = Tile::template stage_pitch<TS>();
        const int lane = lane_id();
        const int fi = lane & 15, fg = lane >> 4;
        char* mine = smem + wave_id() * Tile::WROWS * P;
        __syncthreads();        // every wave is done reading the operand stages
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                char* p = mine + (mt * 16 + fi) * P + (nt * 16 + 4 * fg) * (int)sizeof(TS);
                if constexpr (sizeof(TS) == 4) {
                    *reinterpret_cast<f32x4_t*>(p) = acc[nt][mt];
                } else {
                    bf16x4_t v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (bf16_t)acc[nt][mt][e];
                    *reinterpret_cast<bf16x4_t*>(p) = v;
                }
            }
        __syncthreads();
        return mine;
    }
};

// Global operands of the store epilogue (residual rows, saved BN input rows), fetched into registers BEFORE the
// accumulators are staged through LDS so their latency hides behind the staging instead of stalling the store loop.
template <typename T, int WROWS>
struct EpiOperands {
    static constexpr int EPV = 16 / (int)sizeof(T), LPR = 64 / EPV, RPI = 64 / LPR, ITERS = WROWS / RPI;
    static constexpr bool PRE = ITERS <= 8;          // 128-row wave tiles would need 128 registers: those load in the loop
    Vec16<T> rv[PRE ? ITERS : 1], yv[PRE ? ITERS : 1];
    const T* r; const T* y;
    int M, Nout, m0, n, rsub;
    int res_h, res_w;
    OutMap map;
    // Lean path (LEAN kernels: bf16, every tile whole, dense layout -- what every body convolution of a ResNet step is): the wave's
    // rows are addressed as 32-bit buffer offsets -- one lane offset + a scalar row step -- instead of a 64-bit index per row and
    // lane (PMC: the general epilogue was ~1 070 VALU instructions per wave and tile against 288 MFMAs in the 64-channel layers).
    uint32_t voff, vstep;
    static constexpr bool CAN_FAST = PRE && sizeof(T) == 2;
    // element index of residual row for output row m, or -1 when that pixel receives no residual
    __device__ __forceinline__ long long res_index(int m) const {
        if (res_w == 0) return (long long)m * Nout + n;
        const int hw = res_h * res_w;
        const int img = m / hw, rem = m - img * hw;
        const int h = rem / res_w, w = rem - h * res_w;
        if ((h | w) & 1) return -1;
        const int hc = (res_h + 1) >> 1, wc = (res_w + 1) >> 1;
        return ((long long)(img * hc + (h >> 1)) * wc + (w >> 1)) * Nout + n;
    }
    __device__ __forceinline__ void fetch(const void* __restrict__ res, const void* __restrict__ ybn, int M_, int Nout_,
                                          int m0_, int n0, int res_h_ = 0, int res_w_ = 0, const OutMap* map_ = nullptr) {
        const int lane = lane_id();
        M = M_; Nout = Nout_; m0 = m0_; res_h = res_h_; res_w = res_w_;
        if (map_) map = *map_; else map.wc = 0;
        n = n0 + (lane % LPR) * EPV; rsub = lane / LPR;
        r = reinterpret_cast<const T*>(res);
        y = reinterpret_cast<const T*>(ybn);
        if constexpr (PRE) {
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int m = m0 + it * RPI + rsub;
                const bool ok = m < M && n < Nout;
                const size_t idx = (size_t)map.row(m) * Nout + n;
                if (r && ok) {
                    const long long ri = res_index((int)map.row(m));
                    if (ri >= 0) rv[it] = *reinterpret_cast<const Vec16<T>*>(r + ri);
                    else {
#pragma unroll
                        for (int e = 0; e < EPV; ++e) rv[it].set(e, 0.f);
                    }
                }
                if (y && ok) yv[it] = *reinterpret_cast<const Vec16<T>*>(y + idx);
            }
        }
    }
    // LEAN kernels only (the host checked: bf16, every tile whole, dense rows, no bias / GELU / row map, < 2 GiB): residual and
    // saved-BatchNorm-input rows by 32-bit buffer offsets
    __device__ __forceinline__ void fetch_fast(const void* __restrict__ res, const void* __restrict__ ybn, int M_, int Nout_,
                                               int m0_, int n0) {
        static_assert(CAN_FAST, "lean epilogue: bf16 tiles of at most 64 rows per wave");
        const int lane = lane_id();
        M = M_; Nout = Nout_; m0 = m0_; res_h = 0; res_w = 0; map.wc = 0;
        n = n0 + (lane % LPR) * EPV; rsub = lane / LPR;
        r = reinterpret_cast<const T*>(res);
        y = reinterpret_cast<const T*>(ybn);
        const uint32_t bytes = (uint32_t)M * (uint32_t)Nout * (uint32_t)sizeof(T);
        voff = ((uint32_t)(m0 + rsub) * (uint32_t)Nout + (uint32_t)n) * (uint32_t)sizeof(T);
        vstep = (uint32_t)RPI * (uint32_t)Nout * (uint32_t)sizeof(T);
        if (r) {
            const __amdgpu_buffer_rsrc_t rr = make_rsrc(res, bytes);
#pragma unroll
            for (int it = 0; it < ITERS; ++it)
                rv[it].v = __builtin_bit_cast(decltype(rv[it].v), __builtin_amdgcn_raw_buffer_load_b128(rr, voff, it * vstep, 0));
        }
        if (y) {
            const __amdgpu_buffer_rsrc_t ry = make_rsrc(ybn, bytes);
#pragma unroll
            for (int it = 0; it < ITERS; ++it)
                yv[it].v = __builtin_bit_cast(decltype(yv[it].v), __builtin_amdgcn_raw_buffer_load_b128(ry, voff, it * vstep, 0));
        }
    }
    // valid for rows inside the tensor only
    __device__ __forceinline__ Vec16<T> res_row(int it) const {
        if constexpr (PRE) return rv[it];
        else {
            const long long ri = res_index((int)map.row(m0 + it * RPI + rsub));
            Vec16<T> z;
#pragma unroll
            for (int e = 0; e < EPV; ++e) z.set(e, 0.f);
            return ri >= 0 ? *reinterpret_cast<const Vec16<T>*>(r + ri) : z;
        }
    }
    __device__ __forceinline__ Vec16<T> y_row(int it) const {
        if constexpr (PRE) return yv[it];
        else return *reinterpret_cast<const Vec16<T>*>(y + (size_t)map.row(m0 + it * RPI + rsub) * Nout + n);
    }
};

// host side: may a launch use the LEAN kernels?  (bm x bn = the workgroup's tile)
inline bool epi_lean_ok(bool bf16, long long M, int Nout, int bm, int bn, const EpiBnRed& br) {
    return bf16 && M % bm == 0 && Nout % bn == 0 && br.map.wc == 0 && br.res_w == 0 && !br.bias && !br.act && !br.gelu_bwd &&
           M * Nout * 2 < 0x7fffffffLL;
}

// Output rows of the store epilogue go out with non-temporal stores: a conv / linear output is written once and read one
// kernel later by a streaming pass (which reads it non-temporally as well), so it should not push the weights and the next
// operand out of L2 on its way.  Step-level A/B (tools/ab_libs.sh, three passes on one box): ResNet50 28.06 / 27.90 / 28.07 ->
// 27.83 / 27.85 / 27.87 ms, Swin34 19.31 -> 19.0 ms.  -DEPI_NT_STORE=0 builds the plain-store variant.
#ifndef EPI_NT_STORE
#define EPI_NT_STORE 1
#endif
template <typename T> __device__ __forceinline__ void epi_store_row(T* p, const Vec16<T>& v) {
    if constexpr (EPI_NT_STORE) __builtin_nontemporal_store(v.v, reinterpret_cast<decltype(v.v)*>(p));
    else *reinterpret_cast<Vec16<T>*>(p) = v;
}

// ---- per-channel sums over a wave's staged tile on the MATRIX pipe (LEAN kernels, bf16).
// The tile sits in LDS as [64 rows][64 channels] bf16 with pitch P.  A transposed read (ds_read_b64_tr_b16) hands lane (g, j) the
// eight rows 8g .. 8g+7 (of a 32-row K group) of channel c0 + j -- which is both the A and the B operand of an MFMA that
// contracts over rows: with F the fragment of tile X and F' the fragment of tile Z (same rows, same channels),
//     mfma(F, F')[i][j]  = sum_rows X[row][c0+i] * Z[row][c0+j]   -> diagonal = sum x*z per channel (products exact, fp32 sums)
//     mfma(ONES, F)[i][j] = sum_rows X[row][c0+j]                 -> every row i holds the column sums
// 16 transposed reads + 16 MFMAs per wave replace 64 x (unpack, add, fma) per lane and the 48-step cross-lane reduction: the
// general epilogue's statistics were ~300 of its VALU instructions.  D layout: lane (g, j) holds rows 4g .. 4g+3 of column j, so
// the diagonal element of channel c0 + j is register j & 3 of the lane with g == j >> 2; the column sum is in every lane.
__device__ __forceinline__ bf16x8_t epi_tr_frag(const char* tile, int P, int r0, int c0, int lane) {
    const int g = lane >> 4, j = lane & 15, q = j >> 2, p = j & 3;
    const char* pa = tile + (r0 + 8 * g + q) * P + (c0 + 4 * p) * 2;
    typedef __attribute__((address_space(3))) i16x4_t* lds_p;
    const i16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)LDS_ADDR(pa));
    const i16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)LDS_ADDR(pa + 4 * P));
    typedef __attribute__((ext_vector_type(8))) short i16x8_t;
    return __builtin_bit_cast(bf16x8_t, (i16x8_t)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
// S[cb][*] = column sums of X (channels 16 cb + j), G[cb] = 16 x 16 block of X^T Z around the diagonal; X at tx, Z at tz (may be tx)
template <int WROWS>
__device__ __forceinline__ void epi_mfma_sums(const char* tx, const char* tz, int P, f32x4_t (&S)[4], f32x4_t (&G)[4]) {
    const int lane = lane_id();
    bf16x8_t ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) { S[cb] = f32x4_t{0.f, 0.f, 0.f, 0.f}; G[cb] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int ks = 0; ks < WROWS / 32; ++ks)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const bf16x8_t fx = epi_tr_frag(tx, P, 32 * ks, 16 * cb, lane);
            const bf16x8_t fz = (tz == tx) ? fx : epi_tr_frag(tz, P, 32 * ks, 16 * cb, lane);
            S[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fx, S[cb], 0, 0, 0);
            G[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx, fz, G[cb], 0, 0, 0);
        }
}
// this lane's diagonal element of G (valid in the lanes with (lane & 15) >> 2 == lane >> 4)
__device__ __forceinline__ float epi_diag(const f32x4_t& g, int lane) {
    const int e = lane & 3;
    return e == 0 ? g[0] : (e == 1 ? g[1] : (e == 2 ? g[2] : g[3]));
}
// per-wave sums -> the workgroup's row of per-tile partials (sum over the WM waves that share wn, fixed order)
template <int WM, int WN, int THREADS, int BN>
__device__ __forceinline__ void epi_stats_tail_mfma(const float (&a1)[4], const float (&a2)[4], char* smem, int Nout,
                                                    float* __restrict__ stats, int mtile, int ntile) {
    const int lane = lane_id(), wave = wave_id();
    const int g = lane >> 4, j = lane & 15;
    __syncthreads();                                // staging areas are free again
    float* red = reinterpret_cast<float*>(smem);   // [wave][2][64]
    if ((j >> 2) == g) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            red[(wave * 2 + 0) * 64 + 16 * cb + j] = a1[cb];
            red[(wave * 2 + 1) * 64 + 16 * cb + j] = a2[cb];
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < WN * 2 * 64; t += THREADS) {
        const int c = t & 63, st = (t >> 6) & 1, w_n = t >> 7;
        float a = 0.f;
#pragma unroll
        for (int w_m = 0; w_m < WM; ++w_m) a += red[((w_m * WN + w_n) * 2 + st) * 64 + c];
        const int nn = ntile * BN + w_n * 64 + c;
        if (nn < Nout) stats[((size_t)mtile * 2 + st) * Nout + nn] = a;
    }
}

// Lean store epilogue (LEAN kernels, operands by EpiOperands::fetch_fast): bf16, every tile whole, dense rows, no bias / GELU /
// row map.  Same values, same summation order as the general path below (results and partial sums are bit-identical); what is gone
// is the per-row 64-bit index arithmetic, the per-row bounds checks and the layout cases: one buffer descriptor, one lane offset, a
// scalar row step.
// yoff: byte offset (from `mine`) of the wave's second staging area for the saved BatchNorm-input rows; acc (optional, [2][4]): the wave's
// sums are ADDED there instead of going to the workgroup's partial row -- the caller runs epi_stats_tail_mfma itself (a tile staged in halves)
template <typename T, int WM, int WN, int WROWS, int THREADS, int BN>
__device__ __forceinline__ void nt_epilogue_store_fast(const char* mine, int P, char* smem, int M, int Nout, void* __restrict__ out,
                                                       bool has_res, float* __restrict__ stats, const EpiBnRed& br,
                                                       const EpiOperands<T, WROWS>& ops, int mtile, int ntile,
                                                       int yoff = WM * WN * WROWS * (64 * (int)sizeof(T) + 16), float (*acc)[4] = nullptr) {
    constexpr int EPV = 16 / (int)sizeof(T), LPR = 64 / EPV, RPI = 64 / LPR, ITERS = WROWS / RPI;
    const int lane = lane_id();
    const int chunk = lane % LPR, rsub = lane / LPR;
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(out, (uint32_t)M * (uint32_t)Nout * (uint32_t)sizeof(T));
    const char* src = mine + rsub * P + chunk * 16;
    auto put = [&](int it, const Vec16<T>& v) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v.v), ro, ops.voff, it * ops.vstep, EPI_NT_STORE ? 2 : 0);
    };
    if (stats && br.y) {
        // BatchNorm-backward partials { sum d, sum d * (y - mean) * invstd }, d = the stored gradient behind the ReLU mask, on the
        // matrix pipe: the masked gradient tile stays in the wave's staging area, the saved BatchNorm input rows (already in
        // registers) go to a second one, and  sum d (y - mean) invstd = invstd * (sum d y - mean * sum d)  per channel from
        // ones x D and the diagonal of D^T Y.  What is left on the VALU is the mask (5 instructions per element; none without ReLU).
        const bool masked = br.mscale != nullptr, rs = br.rowkeep != nullptr;
        const int n = ops.n;
        char* ydst = const_cast<char*>(src) + yoff;
        // stochastic depth: sample of this lane's first row and the row's position inside it (rows advance by RPI per iteration)
        int rs_sample = 0, rs_pos = 0;
        if (rs) { const int mrow = ops.m0 + ops.rsub; rs_sample = mrow / br.rows_per; rs_pos = mrow - rs_sample * br.rows_per; }
        float ms[EPV], mb[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) { ms[e] = masked ? br.mscale[n + e] : 0.f; mb[e] = masked ? br.mshift[n + e] : 1.f; }
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(src + it * RPI * P);
            if (has_res) {
#pragma unroll
                for (int e = 0; e < EPV; ++e) v.set(e, v.get(e) + ops.rv[it].get(e));
            }
            put(it, v);
            const Vec16<T> yy = ops.yv[it];
            uint32_t keep_bits = 0xffffffffu;               // all ones: the row's sample was kept (or no stochastic depth)
            if (rs) {
                while (rs_pos >= br.rows_per) { rs_pos -= br.rows_per; ++rs_sample; }
                keep_bits = br.rowkeep[rs_sample] != 0.f ? 0xffffffffu : 0u;
                rs_pos += RPI;
            }
            if (masked) {
                u32x4_t vb = __builtin_bit_cast(u32x4_t, v.v);
                const u32x4_t yb = __builtin_bit_cast(u32x4_t, yy.v);
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float ylo = __uint_as_float(yb[w] << 16), yhi = __uint_as_float(yb[w] & 0xffff0000u);
                    const uint32_t klo = (ylo * ms[2 * w] + mb[2 * w] > 0.f) ? 0x0000ffffu : 0u;
                    const uint32_t khi = (yhi * ms[2 * w + 1] + mb[2 * w + 1] > 0.f) ? 0xffff0000u : 0u;
                    vb[w] &= (klo | khi) & keep_bits;
                }
                *reinterpret_cast<u32x4_t*>(const_cast<char*>(src) + it * RPI * P) = vb;
            } else if (rs) {
                u32x4_t vb = __builtin_bit_cast(u32x4_t, v.v);
#pragma unroll
                for (int w = 0; w < 4; ++w) vb[w] &= keep_bits;
                *reinterpret_cast<u32x4_t*>(const_cast<char*>(src) + it * RPI * P) = vb;
            } else if (has_res) {
                *reinterpret_cast<Vec16<T>*>(const_cast<char*>(src) + it * RPI * P) = v;
            }
            *reinterpret_cast<Vec16<T>*>(ydst + it * RPI * P) = yy;
        }
        f32x4_t S[4], G[4];
        epi_mfma_sums<WROWS>(mine, mine + yoff, P, S, G);
        float a1[4], a2[4];
        const int n0 = ntile * BN + (wave_id() % WN) * 64, j = lane & 15;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const int c = n0 + 16 * cb + j;
            a1[cb] = S[cb][0];
            a2[cb] = br.invstd[c] * (epi_diag(G[cb], lane) - br.mean[c] * a1[cb]);
            if (rs) { a1[cb] *= br.keep_scale; a2[cb] *= br.keep_scale; }
        }
        if (acc) {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) { acc[0][cb] += a1[cb]; acc[1][cb] += a2[cb]; }
        } else {
            epi_stats_tail_mfma<WM, WN, THREADS, BN>(a1, a2, smem, Nout, stats, mtile, ntile);
        }
        return;
    } else if (br.aff_scale) {
        // eval-mode BatchNorm folded into the store (see EpiBnRed): no statistics
        float asc[EPV], ash[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) { asc[e] = br.aff_scale[ops.n + e]; ash[e] = br.aff_shift[ops.n + e]; }
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(src + it * RPI * P);
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
                float o = v.get(e) * asc[e] + ash[e];
                if (has_res) o += ops.rv[it].get(e);
                v.set(e, br.aff_relu ? fmaxf(o, 0.f) : o);
            }
            put(it, v);
        }
    } else {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(src + it * RPI * P);
            if (has_res) {
#pragma unroll
                for (int e = 0; e < EPV; ++e) v.set(e, v.get(e) + ops.rv[it].get(e));
            }
            put(it, v);
            if (has_res && stats)        // the statistics describe the STORED tensor: put the sum back for the matrix-pipe pass
                *reinterpret_cast<Vec16<T>*>(const_cast<char*>(src) + it * RPI * P) = v;
        }
        if (stats) {
            // forward statistics { sum x, sum x^2 } of the stored tile on the matrix pipe
            f32x4_t S[4], G[4];
            epi_mfma_sums<WROWS>(mine, mine, P, S, G);
            float a1[4], a2[4];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) { a1[cb] = S[cb][0]; a2[cb] = epi_diag(G[cb], lane); }
            epi_stats_tail_mfma<WM, WN, THREADS, BN>(a1, a2, smem, Nout, stats, mtile, ntile);
        }
    }
}

// Lean store epilogue of the LINEAR layers (nets/SwinV2.py qkv / proj / fc1 / fc2 and their data-gradients on the 256 x 256 tile: K = 256 ...
// 2048, where the general epilogue's ~2 000 VALU instructions per wave and tile outweigh the 256 ... 2 048 MFMAs): bf16, every tile whole,
// dense rows.  out = [gelu'(pre) *] (gemm [+ res]) [+ bias], rounded to T where the general epilogue rounds; act = gelu(out); the forward-style
// statistics { sum out, sum out^2 } (or the column sums of a GELU data-gradient) on the matrix pipe.  Rows by 32-bit buffer offsets; the
// residual / pre-activation rows of eight row groups are in flight at a time.  The fused BatchNorm-backward partials (br.y without gelu_bwd)
// need a second staging area for the saved BatchNorm-input rows, which the 256 x 256 tile's LDS does not have beside eight 128-row tiles: the
// kernel stages such a tile in two 64-row halves (NtMainloop::stage_half) and runs the 64-row lean epilogue on each, with the free upper half
// of the wave's area as the second tile.
template <int WM, int WN, int WROWS, int THREADS, int BN>
__device__ __forceinline__ void nt_epilogue_store_lean(const char* mine, int P, char* smem, int M, int Nout, void* __restrict__ out,
                                                       const void* __restrict__ res, float* __restrict__ stats, const EpiBnRed& br,
                                                       int mtile, int ntile, int m0, int n0) {
    typedef bf16_t T;
    constexpr int EPV = 8, RPI = 8, ITERS = WROWS / RPI, GRP = 8;
    static_assert(ITERS % GRP == 0, "row groups of eight");
    const int lane = lane_id();
    const int chunk = lane & 7, rsub = lane >> 3;
    const int n = n0 + chunk * EPV;
    const uint32_t bytes = (uint32_t)M * (uint32_t)Nout * 2u;
    const uint32_t voff = ((uint32_t)(m0 + rsub) * (uint32_t)Nout + (uint32_t)n) * 2u, vstep = (uint32_t)RPI * (uint32_t)Nout * 2u;
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(out, bytes);
    const bool has_res = res != nullptr, has_bias = br.bias != nullptr, gbw = br.gelu_bwd != 0, has_act = br.act != nullptr;
    const bool rewrite = stats != nullptr && (has_res || has_bias || gbw);     // the sums describe the STORED tile: put it back for the matrix-pipe pass
    float bb[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) bb[e] = has_bias ? br.bias[n + e] : 0.f;
    char* src = const_cast<char*>(mine) + rsub * P + chunk * 16;
    for (int g0 = 0; g0 < ITERS; g0 += GRP) {
        Vec16<T> rv[GRP], yv[GRP];
        if (has_res) {
            const __amdgpu_buffer_rsrc_t rr = make_rsrc(res, bytes);
#pragma unroll
            for (int j = 0; j < GRP; ++j) rv[j].v = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rr, voff, (g0 + j) * vstep, 0));
        }
        if (gbw) {
            const __amdgpu_buffer_rsrc_t ry = make_rsrc(br.y, bytes);
#pragma unroll
            for (int j = 0; j < GRP; ++j) yv[j].v = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(ry, voff, (g0 + j) * vstep, 0));
        }
#pragma unroll
        for (int j = 0; j < GRP; ++j) {
            const int it = g0 + j;
            Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(src + it * RPI * P);
            if (has_res) {
#pragma unroll
                for (int e = 0; e < EPV; ++e) v.set(e, v.get(e) + rv[j].get(e));
            }
            if (has_bias) {
#pragma unroll
                for (int e = 0; e < EPV; ++e) v.set(e, v.get(e) + bb[e]);
            }
            if (gbw) {
#pragma unroll
                for (int e = 0; e < EPV; ++e) {
                    v.set(e, v.get(e) * gelu_slope<T>(yv[j].get(e)));
                }
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v.v), ro, voff, it * vstep, EPI_NT_STORE ? 2 : 0);
            if (has_act) {
                Vec16<T> ga;
#pragma unroll
                for (int e = 0; e < EPV; ++e) {
                    ga.set(e, gelu_value<T>(v.get(e)));
                }
                const __amdgpu_buffer_rsrc_t ra = make_rsrc(br.act, bytes);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, ga.v), ra, voff, it * vstep, 0);
            }
            if (rewrite) *reinterpret_cast<Vec16<T>*>(src + it * RPI * P) = v;
        }
    }
    if (stats) {
        f32x4_t S[4], G[4];
        epi_mfma_sums<WROWS>(mine, mine, P, S, G);
        float a1[4], a2[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) { a1[cb] = S[cb][0]; a2[cb] = epi_diag(G[cb], lane); }
        epi_stats_tail_mfma<WM, WN, THREADS, BN>(a1, a2, smem, Nout, stats, mtile, ntile);
    }
}

// host side: may a launch on the 256 x 256 tile use the lean linear epilogue?
inline bool nt_lean_ok(bool bf16, long long M, int Nout, const EpiBnRed& br, const float* stats) {
    return bf16 && M % 256 == 0 && Nout % 256 == 0 && br.map.wc == 0 && br.res_w == 0 && !br.aff_scale &&
           !(stats && br.y && !br.gelu_bwd && (br.bias || br.act)) && M * Nout * 2 < 0x7fffffffLL;
}

// Shared store epilogue: the wave's (WROWS x 64) tile sits in its LDS staging area `mine` as T (row = pixel).
// Rows are written back as whole 128/256-byte lines (+ optional residual); the same read-back accumulates the
// per-channel sum / sum of squares of the STORED values -> BN batch-statistic partials [mtile][2][Nout]
// (or the BN-backward partials, see EpiBnRed).
template <typename T, int WM, int WN, int WROWS, int THREADS, int BN>
__device__ __forceinline__ void nt_epilogue_store(const char* mine, int P, char* smem, int M, int Nout,
                                                  void* __restrict__ out, bool has_res, float* __restrict__ stats,
                                                  const EpiBnRed& br, const EpiOperands<T, WROWS>& ops, int mtile,
                                                  int ntile, int m0, int n0) {
    constexpr int EPV = 16 / (int)sizeof(T);          // elements per 16-byte vector
    constexpr int LPR = 64 / EPV;                     // lanes per 64-channel row
    constexpr int RPI = 64 / LPR;                     // rows per wave instruction
    const int lane = lane_id(), wave = wave_id();
    const int chunk = lane % LPR, rsub = lane / LPR;
    const int n = n0 + chunk * EPV;
    float s1[EPV], s2[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    T* o = reinterpret_cast<T*>(out);
    if (stats && br.y && !br.gelu_bwd) {
        // BN-backward partials: per-channel constants of this lane's EPV channels
        float mu[EPV], is[EPV], ms[EPV], mb[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
            const bool ok = n + e < Nout;
            mu[e] = ok ? br.mean[n + e] : 0.f; is[e] = ok ? br.invstd[n + e] : 0.f;
            ms[e] = (ok && br.mscale) ? br.mscale[n + e] : 0.f; mb[e] = (ok && br.mscale) ? br.mshift[n + e] : 1.f;
        }
#pragma unroll
        for (int it = 0; it < WROWS / RPI; ++it) {
            const int row = it * RPI + rsub;
            const int m = m0 + row;
            Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(mine + row * P + chunk * 16);
            if (m < M && n < Nout) {
                if (has_res) {
                    const Vec16<T> rr = ops.res_row(it);
#pragma unroll
                    for (int e = 0; e < EPV; ++e) v.set(e, v.get(e) + rr.get(e));
                }
                epi_store_row(o + (size_t)br.map.row(m) * Nout + n, v);
                const Vec16<T> yr = ops.y_row(it);
                // stochastic depth: rows of a dropped sample contribute nothing, the kept ones keep_scale times their gradient
                const float kf = br.rowkeep ? (br.rowkeep[m / br.rows_per] != 0.f ? br.keep_scale : 0.f) : 1.f;
#pragma unroll
                for (int e = 0; e < EPV; ++e) {
                    const float yy = yr.get(e);
                    const float d = (yy * ms[e] + mb[e] > 0.f) ? v.get(e) * kf : 0.f;      // the value as stored (rounded to T)
                    s1[e] += d; s2[e] += d * (yy - mu[e]) * is[e];
                }
            }
        }
    } else {
        float bb[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) bb[e] = (br.bias && n + e < Nout) ? br.bias[n + e] : 0.f;
        T* ao = reinterpret_cast<T*>(br.act);
#pragma unroll
        for (int it = 0; it < WROWS / RPI; ++it) {
            const int row = it * RPI + rsub;
            const int m = m0 + row;
            Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(mine + row * P + chunk * 16);
            if (m < M && n < Nout) {
                if (has_res) {
                    const Vec16<T> rr = ops.res_row(it);
#pragma unroll
                    for (int e = 0; e < EPV; ++e) v.set(e, v.get(e) + rr.get(e));
                }
                if (br.bias) {
#pragma unroll
                    for (int e = 0; e < EPV; ++e) v.set(e, v.get(e) + bb[e]);
                }
                if (br.gelu_bwd) {
                    const Vec16<T> hv = ops.y_row(it);
#pragma unroll
                    for (int e = 0; e < EPV; ++e) {
                        v.set(e, v.get(e) * gelu_slope<T>(hv.get(e)));
                    }
                }
                const size_t orow = (size_t)br.map.row(m);
                epi_store_row(o + orow * Nout + n, v);
                if (ao) {
                    Vec16<T> ga;
#pragma unroll
                    for (int e = 0; e < EPV; ++e) {
                        ga.set(e, gelu_value<T>(v.get(e)));
                    }
                    *reinterpret_cast<Vec16<T>*>(ao + orow * Nout + n) = ga;
                }
            }
#pragma unroll
            for (int e = 0; e < EPV; ++e) { const float x = v.get(e); s1[e] += x; s2[e] += x * x; }
        }
    }
    if (stats) {
        // rows beyond M were gathered as zeros -> contribute 0.  Reduce over the lanes that share `chunk`.
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
            if constexpr (LPR <= 8) { s1[e] = lane_sum_bit3(s1[e]); s2[e] = lane_sum_bit3(s2[e]); }
            if constexpr (LPR <= 16) { s1[e] = lane_sum_bit4(s1[e]); s2[e] = lane_sum_bit4(s2[e]); }
            s1[e] = lane_sum_bit5(s1[e]); s2[e] = lane_sum_bit5(s2[e]);
        }
        static_assert(LPR == 8 || LPR == 16, "lanes per 64-channel row");
        __syncthreads();                                // staging area is free again
        float* red = reinterpret_cast<float*>(smem);   // [wave][2][64]
        if (lane < LPR) {
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
                red[(wave * 2 + 0) * 64 + chunk * EPV + e] = s1[e];
                red[(wave * 2 + 1) * 64 + chunk * EPV + e] = s2[e];
            }
        }
        __syncthreads();
        // one thread per (wn, stat, channel): sum over the WM waves that share wn
        for (int t = threadIdx.x; t < WN * 2 * 64; t += THREADS) {
            const int c = t & 63, st = (t >> 6) & 1, w_n = t >> 7;
            float a = 0.f;
#pragma unroll
            for (int w_m = 0; w_m < WM; ++w_m) a += red[((w_m * WN + w_n) * 2 + st) * 64 + c];
            const int nn = ntile * BN + w_n * 64 + c;
            if (nn < Nout) stats[((size_t)mtile * 2 + st) * Nout + nn] = a;
        }
    }
}

}  // namespace frhip
