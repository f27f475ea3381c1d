// "TN" implicit GEMM for gfx950: OUT[co][tap][ci] += sum_m P[m][co] * Q[pix(m, tap)][ci]
//   P = output-side gradient rows [M][ldp] (M = n*ho*wo GEMM-K), Q = NHWC activations gathered by conv geometry.
//   Covers: conv weight-gradient, fc weight-gradient, and the two head gradients (dW = dT^T E, dE = (dT^T)^T W).
// Both operands are K-strided in memory (K = pixel index, channels contiguous), so the MFMA fragments are
// read with the gfx950 transposing LDS read ds_read_b64_tr_b16 (bf16) / strided ds_read_b32 (f32 validation mode).
// Tile: 4 waves (2x2); LDS rows are RB = 256 or 128 bytes of channels; 64 pixels per K step; two stages filled
// by LDS-DMA with the 16-byte chunks XOR-swizzled on the source side so both the DMA image and the transposed
// reads are bank-conflict free.  Split-K over pixel ranges; results are added with fp32 atomics in whole
// 256-byte rows staged through LDS.
// Reference counterpart: autograd of nn.Conv2d / F.linear (cuDNN wgrad), nets/resnet.py:23-46, nets/PartialFC.py:201.
#include <cstdlib>
#include "common.h"
#include "frhip.h"

// TN_SLAB_NT (build-time experiment switch): bit 0 non-temporal slab stores in the nine-tap kernel, bit 1 non-temporal slab loads
#ifndef TN_SLAB_NT
#define TN_SLAB_NT 0
#endif
#ifndef T9_AUX
#define T9_AUX 0          // cache policy of the nine-tap kernel's operand loads (experiment switch)
#endif

namespace frhip {

// Workgroup -> (output tile, K split).  All output tiles of ONE K split read the same pixel range of both operands (each
// tile its own channel slice), so they should run at the same time on the same XCD and share its L2.  A (tiles, splits)
// grid does the opposite: linear id % 8 picks the XCD, so with 8 tiles per split every XCD sees ONE tile of every split
// and re-reads its operand slices from HBM (PMC: 280 MB fetched per 256-channel weight-gradient launch against 103 MB of
// operands).  One-dimensional grid, XCD-remapped, tile fastest: consecutive ids of one XCD = the tiles of one split.
struct TnSlot { int tile, split; };
__device__ __forceinline__ TnSlot tn_slot(int ntiles) {
    const uint32_t lin = xcd_remap(blockIdx.x, gridDim.x);
    TnSlot s; s.split = (int)(lin / (uint32_t)ntiles); s.tile = (int)(lin - (uint32_t)s.split * (uint32_t)ntiles);
    return s;
}

struct TnGeom {
    int H, W, C;             // Q tensor
    int Ho, Wo, R, S, stride, pad;
    int M, Kc, ldp;          // GEMM-K rows, valid P columns (= output rows), P row pitch (elements)
    int ksteps, ksteps_per_split;
    FastDiv d_howo, d_wo;
    int slab_stride;             // 0: add into `out` with fp32 atomics; > 0: each K split stores its tiles into out + split*slab_stride
    int adv_wo, adv_ho, adv_n;   // 64 pixels = adv_n images + adv_ho rows + adv_wo columns (mixed-radix step per K step)
    uint32_t p_bytes, q_bytes;
    // operand transform of the nine-tap kernel (XF instantiations): Q is the INPUT of a BatchNorm + ReLU and the weight
    // gradient wants their output; relu(q * xf_scale[ci] + xf_shift[ci]) is formed in LDS after every window stage has landed
    const float* xf_scale;
    const float* xf_shift;
    // nine-tap kernel, TAB instantiations: padding predicates of the four pixel rows a lane addresses, as 64-bit LANE masks per K step,
    // [mask_period][4 rows][up, dn, lf, rt] (t9_mask_table below); K step ks uses row ks % mask_period
    const unsigned long long* mask_tab;
    int mask_period;
};

constexpr int TN_THREADS = 256;
constexpr int TN_KP = 64;   // pixels per K step

template <int RB> __device__ __forceinline__ int tn_swz(int row);
template <> __device__ __forceinline__ int tn_swz<256>(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
template <> __device__ __forceinline__ int tn_swz<128>(int row) { return (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1; }

template <typename T, int RB>
struct TnTile {
    static constexpr int BC = RB / (int)sizeof(T);        // channels per tile side
    static constexpr int WC = BC / 2;                     // per wave
    static constexpr int NT = WC / 16;                    // MFMA tiles per wave side
    static constexpr int TILE_BYTES = TN_KP * RB;
    static constexpr int STAGE_BYTES = 2 * TILE_BYTES;
    static constexpr int ROWS_PER_PIECE = 1024 / RB;      // 4 or 8
    static constexpr int CHUNKS = RB / 16;                // 16 or 8
    static constexpr int PIECES = TN_KP / ROWS_PER_PIECE / 4;   // per wave per tile: 4 or 2
    static constexpr int OUT_PITCH = WC * 4 + 16;
    static constexpr int LDS_BYTES = (2 * STAGE_BYTES > 4 * WC * OUT_PITCH) ? 2 * STAGE_BYTES : 4 * WC * OUT_PITCH;
};

// fragment for one MFMA K group from a [pixel][channel] LDS tile, channels c0..c0+15, pixel rows r0 + (k slots)
template <typename T, int RB> struct TnFrag;
template <int RB> struct TnFrag<bf16_t, RB> {
    // 32 pixels per MFMA: lane group g covers pixels r0 + 8g .. 8g+7 via two transposed 4x16 block reads
    static constexpr int KROWS = 32;
    __device__ static __forceinline__ bf16x8_t load(const char* tile, int r0, int c0, int lane) {
        const int g = lane >> 4, j = lane & 15, q = j >> 2, p = j & 3;
        const int chunk = (c0 >> 3) + (p >> 1);
        const int row_a = r0 + 8 * g + q, row_b = row_a + 4;
        const char* pa = tile + row_a * RB + ((chunk ^ tn_swz<RB>(row_a)) << 4) + 8 * (p & 1);
        const char* pb = tile + row_b * RB + ((chunk ^ tn_swz<RB>(row_b)) << 4) + 8 * (p & 1);
        i16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4_t*)LDS_ADDR(pa));
        i16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4_t*)LDS_ADDR(pb));
        typedef __attribute__((ext_vector_type(8))) short i16x8_t;
        i16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8_t, v);
    }
};
template <int RB> struct TnFrag<float, RB> {
    // 16 pixels per "K group" (4 MFMA 16x16x4): element e of lane group g is pixel r0 + 4e + g
    static constexpr int KROWS = 16;
    __device__ static __forceinline__ f32x4_t load(const char* tile, int r0, int c0, int lane) {
        const int g = lane >> 4, i = lane & 15;
        const int col = c0 + i, chunk = col >> 2, within = (col & 3) * 4;
        f32x4_t v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = r0 + 4 * e + g;
            v[e] = *reinterpret_cast<const float*>(tile + row * RB + ((chunk ^ tn_swz<RB>(row)) << 4) + within);
        }
        return v;
    }
};

template <typename T, int RB, int NT>
__device__ __forceinline__ void tn_epilogue(f32x4_t (&acc)[NT][NT], char* smem, float* __restrict__ out,
                                            const TnGeom& g, int co0, int ci0, int tap, int taps, int split) {
    typedef TnTile<T, RB> Tile;
    static_assert(NT == Tile::NT, "accumulator shape");
    const int lane = lane_id(), wave = wave_id();
    const int wco = wave >> 1, wci = wave & 1;
    // ---- epilogue: D[row = co (4g+reg)][col = ci (lane&15)] -> LDS [co][ci] fp32 -> row-wise atomic adds
    __syncthreads();
    constexpr int P = Tile::OUT_PITCH;
    char* mine = smem + wave * Tile::WC * P;
    const int fi = lane & 15, fg = lane >> 4;
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                *reinterpret_cast<float*>(mine + (a * 16 + 4 * fg + e) * P + (b * 16 + fi) * 4) = acc[a][b][e];
    __syncthreads();
    constexpr int LPR = Tile::WC;            // lanes per row (64, 32 or 16)
    constexpr int RPI = 64 / LPR;
    const int col = lane % LPR, rsub = lane / LPR;
    const int ci = ci0 + wci * Tile::WC + col;
    for (int it = 0; it < Tile::WC / RPI; ++it) {
        const int row = it * RPI + rsub;
        const int co = co0 + wco * Tile::WC + row;
        if (co < g.Kc && ci < g.C) {
            const float v = *reinterpret_cast<const float*>(mine + row * P + col * 4);
            const size_t idx = ((size_t)co * taps + tap) * g.C + ci;
            if (g.slab_stride) out[(size_t)split * g.slab_stride + idx] = v;     // private slab: plain coalesced store
            else atomicAdd(out + idx, v);
        }
    }
}

template <typename T, int RB>
__global__ __launch_bounds__(TN_THREADS, 2) void tn_kernel(TnGeom g, const void* __restrict__ p_ptr,
                                                           const void* __restrict__ q_ptr, float* __restrict__ out,
                                                           int co_tiles, int ci_tiles, int taps) {
    typedef TnTile<T, RB> Tile;
    typedef typename Mma<T>::Frag Frag;
    constexpr int NT = Tile::NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = lane_id(), wave = wave_id();
    const TnSlot slot = tn_slot(co_tiles * ci_tiles * taps);
    uint32_t lin = (uint32_t)slot.tile;
    const int tap = (int)(lin % (uint32_t)taps); lin /= (uint32_t)taps;
    const int ci_tile = (int)(lin % (uint32_t)ci_tiles), co_tile = (int)(lin / (uint32_t)ci_tiles);
    const int fr = tap / g.S, fs = tap - fr * g.S;
    const int ks_begin = slot.split * g.ksteps_per_split;
    const int ks_end = min(g.ksteps, ks_begin + g.ksteps_per_split);
    const int wco = wave >> 1, wci = wave & 1;

    f32x4_t acc[NT][NT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // inline-assembly loads (glds16_asm, common.h): the builtin made the compiler wait for the NEXT step's loads (vmcnt(0)) before the
    // first transposed read of the current step -- load and compute never overlapped
    const u32x4_t rp = make_rsrc_words(p_ptr, g.p_bytes);
    const u32x4_t rq = make_rsrc_words(q_ptr, g.q_bytes);
    const uint32_t lds_base = (uint32_t)(uintptr_t)LDS_ADDR(smem);

    // DMA lane roles: piece covers ROWS_PER_PIECE rows; lane -> row sub, physical chunk
    const int sub = lane / Tile::CHUNKS, phys = lane % Tile::CHUNKS;
    const int co0 = co_tile * Tile::BC, ci0 = ci_tile * Tile::BC;

    // Row bookkeeping: the pixel (image base, ho, wo) of every row this thread stages, decoded ONCE at the first
    // K step and then advanced by 64 pixels per step with a mixed-radix add (two compares) instead of two
    // divisions per row per step.
    int row_of[Tile::PIECES], pixbase[Tile::PIECES], rho[Tile::PIECES], rwo[Tile::PIECES];
    uint32_t chunk_el[Tile::PIECES];
    const int HWin = g.H * g.W;
#pragma unroll
    for (int j = 0; j < Tile::PIECES; ++j) {
        const int piece = wave * Tile::PIECES + j;
        const int row = piece * Tile::ROWS_PER_PIECE + sub;
        row_of[j] = row;
        chunk_el[j] = (uint32_t)((phys ^ tn_swz<RB>(row)) * (16 / (int)sizeof(T)));   // first channel of this lane's chunk
        const uint32_t m = (uint32_t)(ks_begin * TN_KP + row);
        const uint32_t n = fdiv(m, g.d_howo);
        const uint32_t rem = m - n * (uint32_t)(g.Ho * g.Wo);
        const uint32_t ho = fdiv(rem, g.d_wo);
        pixbase[j] = (int)n * HWin; rho[j] = (int)ho; rwo[j] = (int)(rem - ho * (uint32_t)g.Wo);
    }

    auto stage = [&](int buf, int ks) {
        const uint32_t sp = lds_base + (uint32_t)(buf * Tile::STAGE_BYTES);
        const uint32_t sq = sp + (uint32_t)Tile::TILE_BYTES;
#pragma unroll
        for (int j = 0; j < Tile::PIECES; ++j) {
            const int piece = wave * Tile::PIECES + j;
            const int m = ks * TN_KP + row_of[j];
            const int ce = (int)chunk_el[j];
            uint32_t offp = OOB_OFFSET, offq = OOB_OFFSET;
            if (m < g.M) {
                if (co0 + ce < g.ldp) offp = ((uint32_t)m * (uint32_t)g.ldp + (uint32_t)(co0 + ce)) * (uint32_t)sizeof(T);
                const int hi = rho[j] * g.stride - g.pad + fr, wi = rwo[j] * g.stride - g.pad + fs;
                if ((unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W && ci0 + ce < g.C)
                    offq = ((uint32_t)(pixbase[j] + hi * g.W + wi) * (uint32_t)g.C + (uint32_t)(ci0 + ce)) * (uint32_t)sizeof(T);
            }
            glds16_asm(rp, sp + (uint32_t)(piece * 1024), offp);
            glds16_asm(rq, sq + (uint32_t)(piece * 1024), offq);
            // advance this row by one K step (64 pixels)
            int wo = rwo[j] + g.adv_wo, ho = rho[j] + g.adv_ho, nb = g.adv_n;
            if (wo >= g.Wo) { wo -= g.Wo; ++ho; }
            if (ho >= g.Ho) { ho -= g.Ho; ++nb; }
            rwo[j] = wo; rho[j] = ho; pixbase[j] += nb * HWin;
        }
    };

    auto compute = [&](int buf) {
        const char* tp = smem + buf * Tile::STAGE_BYTES;
        const char* tq = tp + Tile::TILE_BYTES;
        constexpr int KR = TnFrag<T, RB>::KROWS;
#pragma unroll
        for (int kk = 0; kk < TN_KP / KR; ++kk) {
            Frag pf[NT], qf[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                pf[t] = TnFrag<T, RB>::load(tp, kk * KR, wco * Tile::WC + t * 16, lane);
                qf[t] = TnFrag<T, RB>::load(tq, kk * KR, wci * Tile::WC + t * 16, lane);
            }
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b) Mma<T>::run(pf[a], qf[b], acc[a][b]);
        }
    };

    if (ks_begin < ks_end) {
        stage(0, ks_begin);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int cur = 0;
        for (int ks = ks_begin; ks < ks_end - 1; ++ks) {
            stage(cur ^ 1, ks + 1);
            compute(cur);
            // lgkmcnt(0): this wave's LDS reads of stage `cur` have RETURNED before the barrier behind which that stage is overwritten.  With
            // the builtin the compiler placed this wait itself; it does not know that the inline-assembly loads write LDS, so the source has
            // to say it (the fp32 mode's K step is hundreds of 4-byte LDS reads: a deep queue for the next load to overtake)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            cur ^= 1;
        }
        compute(cur);
    }

    tn_epilogue<T, RB, NT>(acc, smem, out, g, co0, ci0, tap, taps, slot.split);
}


// ------------------------------------------------------------------------------------------------------------------
// "All nine taps" weight-gradient for 3x3 / stride-1 / pad-1 convolutions, bf16.
// The per-tap kernel above moves 32 KB of LDS-DMA per 1 M MACs (32 MAC/B): at ~70 GB/s of L2 -> LDS per CU it is
// DMA-bound at less than half the MFMA rate.  Here one workgroup owns a (CO_T co x 64 ci) tile for ALL nine taps.
// Per K step (64 output pixels) it stages the dy rows once and ONE input window of 64 + 2W + 2 pixel rows (stride 1:
// the nine taps read the same rows at offsets dy*W + dx): 150-240 MAC/B, and the dy fragments are transposed-read once
// and reused by all taps.  Padding is resolved at the transposed read (lanes whose row is padding point at a zero
// block).  Every LDS address is computed once before the K loop.
//   <2,4,4,1>: 128 co x 64 ci, eight waves, wave tile 64 x 16, 144 accumulator registers
//              (a 64 x 32 wave tile needs 288 accumulators: more than the 256 AGPRs, the compiler then shuffles them)
//   <1,4,4,1>:  64 co x 64 ci, four waves, same wave tile, two workgroups per CU
// The inner loop is VALU-issue-bound before it is MFMA-bound (an MFMA shadows only ~3 VALU issues), so per fragment
// read it does ONE select: offsets are stage-relative registers, the stage base is an instruction immediate.
// Timing-only ablation switches (tools/ablate.py; results are WRONG with any bit set): 1 no per-step barrier, 2 no MFMA,
// 4 no LDS fragment reads, 8 no in-loop DMA, 16 no epilogue.
#ifndef FRHIP_ABL
#define FRHIP_ABL 0
#endif
#ifndef T9_DEPTH
#define T9_DEPTH 6        // window-fragment look-ahead (tools/ablate.py ABL_DEFS sweep, round 2: 4 -> 6 with the stagger below: -3.5 %)
#endif
#ifndef T9_PIN
#define T9_PIN 0
#endif
#ifndef T9_STAGGER
#define T9_STAGGER 8      // experiment: waves 4-7 of the 8-wave tile start every K step T9_STAGGER x 64 clocks late (MI355X_MICROARCH "two waves per SIMD", item 9)
#endif
constexpr int T9_MAXW = 56;
constexpr int T9_QROWS = 192;                               // 24 pieces of 8 rows >= 64 + 2*56 + 2

// NST = LDS stages (operand DMA runs NST - 1 K steps ahead), QROWS = window rows a stage holds (>= 64 + 2 W + 2)
template <int WCO, int WCI, int COF, int CIF, int NST = 2, int QROWS = T9_QROWS>
struct T9Cfg {
    static_assert(WCI * CIF * 16 == 64 && COF == 4, "64 input channels per tile");
    static constexpr int NW = WCO * WCI;                     // waves: 4 or 8
    static constexpr int QPW_MAX = QROWS / 8 / NW;           // window pieces per wave (every wave issues all of them: fixed DMA count)
    static constexpr int CO_T = WCO * COF * 16;              // 64 or 128 output channels per tile
    static constexpr int P_RB = CO_T * 2;                    // bytes per dy row: 128 or 256
    static constexpr int P_BYTES = TN_KP * P_RB;
    static constexpr int P_PIECES = P_BYTES / 1024 / NW;     // 1-KiB DMA pieces per wave
    static constexpr int P_CHUNKS = P_RB / 16, P_RPP = 1024 / P_RB;
    static constexpr int Q_BYTES = QROWS * 128;
    static constexpr int ZERO = P_BYTES + Q_BYTES;           // 64 zero bytes at the end of EACH stage
    static constexpr int STAGE = ZERO + 64;                  // two stages; stage-relative offsets + an immediate
    static constexpr int EP = CIF * 16 * 4 + 16;             // epilogue staging pitch (bytes)
    static constexpr int EPI_BYTES = NW * COF * 16 * EP;
    static constexpr int LDS = (NST * STAGE > EPI_BYTES) ? NST * STAGE : EPI_BYTES;
    static constexpr int DMA_PER_STAGE = P_PIECES + QPW_MAX; // DMA instructions a wave issues per stage
    static_assert((NST - 1) * STAGE < 65536 && (QROWS / 8) % NW == 0, "stage base must fit the 16-bit DS offset field");
};

// TAB: the padding predicates come from a table of lane masks (scalar loads, s_and_b64, v_cndmask on the SGPR pair) instead of per-lane
// position tracking and compares: 22 v_cmp + 20 position updates per K step (72 MFMAs) leave the vector instruction stream.
template <int WCO, int WCI, int COF, int CIF, bool XF = false, int NST = 2, int QROWS = T9_QROWS, bool TAB = false>
__global__ __launch_bounds__(64 * WCO * WCI, (WCO * WCI > 4 ? 1 : 2))
void tn_taps9_kernel(TnGeom g, const void* __restrict__ p_ptr, const void* __restrict__ q_ptr, float* __restrict__ out,
                     int co_tiles, int ci_tiles) {
    typedef bf16_t T;
    typedef T9Cfg<WCO, WCI, COF, CIF, NST, QROWS> Cfg;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = lane_id(), wave = wave_id();
    const TnSlot slot = tn_slot(co_tiles * ci_tiles);
    const uint32_t lin = (uint32_t)slot.tile;
    const int ci_tile = (int)(lin % (uint32_t)ci_tiles), co_tile = (int)(lin / (uint32_t)ci_tiles);
    const int ks_begin = slot.split * g.ksteps_per_split;
    const int ks_end = min(g.ksteps, ks_begin + g.ksteps_per_split);
    const int wco = wave / WCI, wci = wave % WCI;
    const int co0 = co_tile * Cfg::CO_T, ci0 = ci_tile * 64;
    const int qrows = TN_KP + 2 * g.W + 2, qpieces = (qrows + 7) >> 3;
    const int qpw = (qpieces + Cfg::NW - 1) / Cfg::NW;      // window pieces per wave

    f32x4_t acc[9][COF][CIF];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int a = 0; a < COF; ++a)
#pragma unroll
            for (int b = 0; b < CIF; ++b) acc[t][a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (threadIdx.x < 4 * NST)
        *reinterpret_cast<f32x4_t*>(smem + (threadIdx.x >> 2) * Cfg::STAGE + Cfg::ZERO + (threadIdx.x & 3) * 16) = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // operand loads as inline assembly (glds16_asm, common.h): with the builtin the compiler put an s_waitcnt vmcnt(0) in front of the first
    // transposed read of every K step -- right behind the step's own DMA issue, so the NST-stage pipeline below never had more than the
    // current step's loads in flight
    const u32x4_t rp = make_rsrc_words(p_ptr, g.p_bytes);
    const u32x4_t rq = make_rsrc_words(q_ptr, g.q_bytes);
    const uint32_t lds_base = (uint32_t)(uintptr_t)LDS_ADDR(smem);

    // ---- loader: 1-KiB pieces.  dy: P_RPP rows of P_RB bytes per piece; window: 8 rows of 128 B per piece.
    uint32_t offp[Cfg::P_PIECES], offq[Cfg::QPW_MAX];
    bool okp[Cfg::P_PIECES], okq[Cfg::QPW_MAX];
    const uint32_t incp = (uint32_t)(TN_KP * g.ldp) * 2u, incq = (uint32_t)(TN_KP * g.C) * 2u;
#pragma unroll
    for (int j = 0; j < Cfg::P_PIECES; ++j) {
        const int row = (wave * Cfg::P_PIECES + j) * Cfg::P_RPP + lane / Cfg::P_CHUNKS;
        const int ce = ((lane % Cfg::P_CHUNKS) ^ tn_swz<Cfg::P_RB>(row)) * 8;
        okp[j] = co0 + ce < g.ldp;
        offp[j] = (uint32_t)(((ks_begin * TN_KP + row) * g.ldp + co0 + ce) * 2);     // rows past M lie beyond p_bytes -> zero fill
    }
#pragma unroll
    for (int j = 0; j < Cfg::QPW_MAX; ++j) {
        const int row = (wave + Cfg::NW * j) * 8 + (lane >> 3);
        const int ce = ((lane & 7) ^ tn_swz<128>(row)) * 8;
        okq[j] = ci0 + ce < g.C && j < qpw;          // pieces past the window: out-of-range offset, zero fill, no traffic
        offq[j] = (uint32_t)(((ks_begin * TN_KP - g.W - 1 + row) * g.C + ci0 + ce) * 2);   // negative pixel -> out of range
    }
    auto stage = [&](int buf) {
        const uint32_t sp = lds_base + (uint32_t)(buf * Cfg::STAGE);
        const uint32_t sq = sp + (uint32_t)Cfg::P_BYTES;
#pragma unroll
        for (int j = 0; j < Cfg::P_PIECES; ++j) {
            glds16_asm(rp, sp + (uint32_t)((wave * Cfg::P_PIECES + j) * 1024), okp[j] ? offp[j] : OOB_OFFSET);
            offp[j] += incp;
        }
#pragma unroll
        for (int j = 0; j < Cfg::QPW_MAX; ++j) {
            glds16_asm(rq, sq + (uint32_t)((wave + Cfg::NW * j) * 1024), okq[j] ? offq[j] : OOB_OFFSET);
            offq[j] += incq;
        }
    };

    // XF: BatchNorm + ReLU of the window pieces THIS wave loaded into stage `buf` for K step `ks` (called after the wave's own
    // vmcnt(0), before the barrier that publishes the stage).  A lane's logical 16-byte chunk of a window row depends on the
    // row's swizzle bits 1 and 3 only -- bit 1 of (lane >> 3) and bit 0 of the wave -- so it is the same for every piece and the
    // lane's eight (scale, shift) pairs are loaded once.  Rows outside the tensor stay zero (the conv pads the ACTIVATED tensor).
    float xsc[8], xsh[8];
    if constexpr (XF) {
        const int row0 = wave * 8 + (lane >> 3);
        const int ce = ((lane & 7) ^ tn_swz<128>(row0)) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) { xsc[e] = g.xf_scale[ci0 + ce + e]; xsh[e] = g.xf_shift[ci0 + ce + e]; }
    }
    auto xform_stage = [&](int buf, int ks) {
        if constexpr (XF) {
            char* sq = smem + buf * Cfg::STAGE + Cfg::P_BYTES;
#pragma unroll
            for (int j = 0; j < Cfg::QPW_MAX; ++j) {
                if (j < qpw) {
                    const int row = (wave + Cfg::NW * j) * 8 + (lane >> 3);
                    const long long pix = (long long)ks * TN_KP - g.W - 1 + row;
                    if (pix >= 0 && pix < (long long)g.M) {
                        bf16x8_t* a = reinterpret_cast<bf16x8_t*>(sq + (wave + Cfg::NW * j) * 1024 + lane * 16);
                        bf16x8_t v = *a;
#pragma unroll
                        for (int e = 0; e < 8; ++e) { const float o = (float)v[e] * xsc[e] + xsh[e]; v[e] = (bf16_t)fmaxf(o, 0.f); }
                        *a = v;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    };

    // ---- consumer: the four pixel rows (of the 64) this lane addresses in the transposed reads.  All offsets are
    //      stage-relative; the stage base is a compile-time constant of the unrolled K loop and lands in the DS
    //      instruction's immediate, so a read costs one select (padding -> zero block) and nothing else.
    const int fg = lane >> 4, fj = lane & 15, fq = fj >> 2, fp = fj & 3;
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_ADDR(smem);      // LDS byte address of stage 0
    const uint32_t zero_a = lds0 + (uint32_t)Cfg::ZERO;
    int pho[4], pwo[4];
    uint32_t pa[4];        // dy slot of row i, channel tile 0 (tile t = ^ (t << 5))
    uint32_t qa[9][4];     // window slot of row i at tap t
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int prow = (i >> 1) * 32 + 8 * fg + fq + 4 * (i & 1);          // i = 2*kk + half
        const uint32_t m = (uint32_t)(ks_begin * TN_KP + prow);
        const uint32_t n = fdiv(m, g.d_howo);
        const uint32_t rem = m - n * (uint32_t)(g.Ho * g.Wo);
        const uint32_t ho = fdiv(rem, g.d_wo);
        pho[i] = (int)ho; pwo[i] = (int)(rem - ho * (uint32_t)g.Wo);
        const int cp = wco * COF * 2 + (fp >> 1), cq = wci * CIF * 2 + (fp >> 1);
        pa[i] = lds0 + (uint32_t)(prow * Cfg::P_RB + ((cp ^ tn_swz<Cfg::P_RB>(prow)) << 4) + 8 * (fp & 1));
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int qrow = prow + g.W + 1 + (t / 3 - 1) * g.W + (t % 3 - 1);
            qa[t][i] = lds0 + (uint32_t)(Cfg::P_BYTES + qrow * 128 + ((cq ^ tn_swz<128>(qrow)) << 4) + 8 * (fp & 1));
        }
    }
    int tstep = 0;
    unsigned long long mk[16];
    // the table is read-only for the lifetime of the process: constant address space -> the loads are scalar (s_load_dwordx16), the
    // masks live in SGPR pairs and feed s_and_b64 / v_cndmask directly
    typedef const __attribute__((address_space(4))) unsigned long long* mask_cp;
    const mask_cp mask_base = (mask_cp)(uintptr_t)g.mask_tab;
    if constexpr (TAB) {
        tstep = __builtin_amdgcn_readfirstlane(ks_begin % g.mask_period);
#pragma unroll
        for (int k = 0; k < 16; ++k) mk[k] = mask_base[tstep * 16 + k];
    }
    typedef __attribute__((ext_vector_type(8))) short i16x8_t;
    // lo_a / hi_a: LDS byte addresses in stage 0; OFF (compile time) selects the stage through the DS immediate
    auto tr8 = [&](auto off_c, uint32_t lo_a, uint32_t hi_a) {
        constexpr int OFF = decltype(off_c)::value;
        typedef __attribute__((address_space(3))) i16x4_t* lds_p;
        if constexpr (FRHIP_ABL & 4) {
            bf16x8_t f; asm volatile("" : "=v"(f) : "v"(lo_a), "v"(hi_a)); return f;
        } else {
            i16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)((__attribute__((address_space(3))) char*)(uintptr_t)lo_a + OFF));
            i16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)((__attribute__((address_space(3))) char*)(uintptr_t)hi_a + OFF));
            return __builtin_bit_cast(bf16x8_t, (i16x8_t)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
    };

    auto compute = [&](auto buf_c) {
        typedef std::integral_constant<int, decltype(buf_c)::value * Cfg::STAGE> Off;
        if constexpr (T9_STAGGER > 0 && Cfg::NW == 8) {
            if (wave >= 4) {
#pragma unroll
                for (int z = 0; z < T9_STAGGER; ++z) __builtin_amdgcn_s_sleep(1);
            }
        }
        // rows past M need no mask: their dy rows were zero-filled and every window row read is finite data or zero
        bool up[4], dn[4], lf[4], rt[4];
        if constexpr (TAB) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                up[i] = __builtin_amdgcn_inverse_ballot_w64(mk[4 * i + 0]); dn[i] = __builtin_amdgcn_inverse_ballot_w64(mk[4 * i + 1]);
                lf[i] = __builtin_amdgcn_inverse_ballot_w64(mk[4 * i + 2]); rt[i] = __builtin_amdgcn_inverse_ballot_w64(mk[4 * i + 3]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) { up[i] = pho[i] > 0; dn[i] = pho[i] < g.H - 1; lf[i] = pwo[i] > 0; rt[i] = pwo[i] < g.W - 1; }
        }
        // software pipeline over the 18 (kk, tap) groups: the transposed reads of group n+1 (and the dy fragments of
        // the second K half) are issued before the MFMAs of group n, so LDS latency hides behind the matrix pipe
        bf16x8_t pf[2][COF];
        auto loadp = [&](int kk) {
#pragma unroll
            for (int a = 0; a < COF; ++a)
                pf[kk][a] = tr8(Off{}, pa[2 * kk] ^ (uint32_t)(a << 5), pa[2 * kk + 1] ^ (uint32_t)(a << 5));
        };
        auto loadq = [&](int n, bf16x8_t (&q)[CIF]) {
            const int kk = n / 9, t = n - kk * 9;
            const int i0 = 2 * kk, i1 = 2 * kk + 1;
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            const bool v0 = (dy < 0 ? up[i0] : (dy > 0 ? dn[i0] : true)) && (dx < 0 ? lf[i0] : (dx > 0 ? rt[i0] : true));
            const bool v1 = (dy < 0 ? up[i1] : (dy > 0 ? dn[i1] : true)) && (dx < 0 ? lf[i1] : (dx > 0 ? rt[i1] : true));
            const uint32_t o0 = v0 ? qa[t][i0] : zero_a, o1 = v1 ? qa[t][i1] : zero_a;
#pragma unroll
            for (int b = 0; b < CIF; ++b)       // the zero block is 64 bytes, so ^32 stays inside it
                q[b] = tr8(Off{}, o0 ^ (uint32_t)(b << 5), o1 ^ (uint32_t)(b << 5));
        };
        auto mma = [&](int n, const bf16x8_t (&q)[CIF]) {
            const int kk = n / 9, t = n - kk * 9;
#pragma unroll
            for (int a = 0; a < COF; ++a)
#pragma unroll
                for (int b = 0; b < CIF; ++b) {
                    if constexpr (FRHIP_ABL & 2) { bf16x8_t fa = pf[kk][a], fb = q[b]; asm volatile("" :: "v"(fa), "v"(fb)); }
                    else Mma<T>::run(pf[kk][a], q[b], acc[t][a][b]);
                }
        };
        // ring of DEPTH window-fragment sets: group n + DEPTH - 1 is read while group n multiplies (a group is only
        // COF*CIF MFMAs = 64 matrix-pipe cycles, LDS latency under load is several times that)
        constexpr int DEPTH = T9_DEPTH;
        bf16x8_t qr[DEPTH][CIF];
        loadp(0);
#pragma unroll
        for (int n = 0; n < DEPTH - 1; ++n) loadq(n, qr[n]);
#pragma unroll
        for (int n = 0; n < 18; ++n) {
            if (n + DEPTH - 1 < 18) loadq(n + DEPTH - 1, qr[(n + DEPTH - 1) % DEPTH]);
            if (n == 3) loadp(1);
            mma(n, qr[n % DEPTH]);
#if T9_PIN == 1
            // issue-slot packing: the address selects and the transposed reads of the look-ahead group go BETWEEN this
            // group's MFMAs (an MFMA keeps the matrix pipe busy for 16 cycles; what is issued in its shadow is free)
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#elif T9_PIN == 2
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#endif
        }
        if constexpr (TAB) {
            // the next K step's masks: scalar loads, issued behind this step's last fragment reads and consumed after the barrier
            tstep = tstep + 1 == g.mask_period ? 0 : tstep + 1;
#pragma unroll
            for (int k = 0; k < 16; ++k) mk[k] = mask_base[tstep * 16 + k];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int wo = pwo[i] + g.adv_wo, ho = pho[i] + g.adv_ho;
                if (wo >= g.Wo) { wo -= g.Wo; ++ho; }
                if (ho >= g.Ho) ho -= g.Ho;
                pwo[i] = wo; pho[i] = ho;
            }
        }
    };

    // ---- K loop: NST LDS stages (unrolled by NST so the stage is a compile-time constant), DMA NST - 1 steps ahead.
    //      Two stages leave a load ONE K step (72 MFMAs per wave, ~0.5 us) to arrive; with one wave per SIMD (the 4-wave tile, one
    //      workgroup per CU) every step then ends waiting for memory.  Three stages (W <= 28: 24 KB each) give it two steps.
    //      Every wave issues DMA_PER_STAGE loads per stage, so "the stage before the newest has landed" is a counted wait.
    if (ks_begin < ks_end) {
        const int nks = ks_end - ks_begin;
#pragma unroll
        for (int pre = 0; pre < NST - 1; ++pre)
            if (pre < nks) stage(pre);
        // stages 0 .. NST-2 are in flight; stage 0 must have landed: all but the min(nks, NST-1) - 1 newest
        if (NST >= 4 && nks > 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * Cfg::DMA_PER_STAGE) : "memory");
        else if (NST >= 3 && nks > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(Cfg::DMA_PER_STAGE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        xform_stage(0, ks_begin);
        __syncthreads();
        auto step = [&](auto buf_c, int it) {
            constexpr int BUF = decltype(buf_c)::value;
            // refill the stage read one step ago (everyone passed the barrier that ended that step)
            if (it + NST - 1 < nks && !(FRHIP_ABL & 8)) stage((BUF + NST - 1) % NST);
            compute(buf_c);
            if (it + 1 < nks) {
                // the next stage (it + 1) must have landed; younger stages issued so far: it + 2 .. min(it + NST - 1, nks - 1)
                // lgkmcnt(0): this wave's reads of the stage refilled behind the barrier have returned (see the per-tap kernel)
                if (NST >= 4 && it + 3 < nks) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(2 * Cfg::DMA_PER_STAGE) : "memory");
                else if (NST >= 3 && it + 2 < nks) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(Cfg::DMA_PER_STAGE) : "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                xform_stage((BUF + 1) % NST, ks_begin + it + 1);
                if constexpr (!(FRHIP_ABL & 1)) __builtin_amdgcn_s_barrier();
            }
        };
        for (int it = 0; it < nks; it += NST) {
            step(std::integral_constant<int, 0>{}, it);
            if (it + 1 < nks) step(std::integral_constant<int, 1>{}, it + 1);
            if constexpr (NST >= 3) { if (it + 2 < nks) step(std::integral_constant<int, 2>{}, it + 2); }
            if constexpr (NST >= 4) { if (it + 3 < nks) step(std::integral_constant<int, 3>{}, it + 3); }
        }
    }
    // ---- epilogue: accumulators straight from registers.  D layout: lane holds rows co = a*16 + 4*(lane>>4) + e, column
    //      ci = b*16 + (lane & 15): 16 lanes write one 64-byte run of dw[co][tap][ci..] -- exactly what staging the wave's
    //      16-ci-wide tile through LDS produced, minus eighteen barriers per workgroup.
    const int fi = lane & 15;
    if constexpr (FRHIP_ABL & 16) {
        if (g.M >= 0) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int a = 0; a < COF; ++a)
#pragma unroll
                    for (int b = 0; b < CIF; ++b) asm volatile("" :: "v"(acc[t][a][b]));
            return;
        }
    }
    if (g.slab_stride) {
        // Private slab of this K split, in a layout made for the WRITER: [co / 4][j = 0..8][ci][4 floats], where the 36 values a
        // lane holds for four consecutive output channels (e) and the nine taps (t) are numbered k = 9 e + t = 4 j + r.  A lane
        // then issues nine 16-byte stores per accumulator group instead of thirty-six 4-byte ones, sixteen lanes cover 256
        // contiguous bytes, and the final reduce pass (slab9_final_kernel) converts to dw[co][tap][ci] while it sums the splits.
        float* dst = out + (size_t)slot.split * g.slab_stride;
#pragma unroll
        for (int a = 0; a < COF; ++a) {
            const int q = (co0 + wco * COF * 16 + a * 16 + 4 * fg) >> 2;           // co / 4 of this lane's four rows
            if (q * 4 >= g.Kc) continue;
#pragma unroll
            for (int b = 0; b < CIF; ++b) {
                const int ci = ci0 + wci * CIF * 16 + b * 16 + fi;
                if (ci >= g.C) continue;
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    f32x4_t v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const int k = 4 * j + r; v[r] = acc[k % 9][a][b][k / 9]; }
                    float* p = dst + (((size_t)q * 9 + j) * g.C + ci) * 4;
                    if constexpr (TN_SLAB_NT & 1) __builtin_nontemporal_store(v, reinterpret_cast<f32x4_t*>(p));
                    else *reinterpret_cast<f32x4_t*>(p) = v;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int a = 0; a < COF; ++a)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int co = co0 + wco * COF * 16 + a * 16 + 4 * fg + e;
            if (co >= g.Kc) continue;
#pragma unroll
            for (int b = 0; b < CIF; ++b) {
                const int ci = ci0 + wci * CIF * 16 + b * 16 + fi;
                if (ci >= g.C) continue;
#pragma unroll
                for (int t = 0; t < 9; ++t) atomicAdd(out + ((size_t)co * 9 + t) * g.C + ci, acc[t][a][b][e]);
            }
        }
}


// ------------------------------------------------------------------------------------------------------------------
// "Rows" nine-tap weight gradient for 14 x 14 maps (3x3 / stride 1 / pad 1, bf16; the 256-channel layers of the ResNet body: 28 of the
// 47 nine-tap launches of a step).  The kernel above walks the pixel stream in K steps of 64 consecutive pixels, so every tap's window
// fragment is a different shifted read: 52 transposing reads and 36 padding selects per 72 MFMAs, which the ablation names as the binding
// term.  Here a K step is ONE IMAGE ROW on v_mfma_f32_32x32x16_bf16 (K = 16 slots: 14 pixels + 2 pad), and that turns the
// three vertical taps into REGISTER reuse: the fragment of input row r at column shift dx serves tap (+1, dx) at output row r - 1, tap
// (0, dx) at row r and tap (-1, dx) at row r + 1.  Per row and wave: 2 transposed reads for the dy fragment + 6 for the three shifts of ONE
// new input row feed 9 MFMAs of 32 x 32 x 16 (0.22 reads per 16x16x32-equivalent MFMA instead of 0.72), and no select at all:
//   * LDS holds an image row at a pitch of 16 pixel slots; slots 14, 15 are zero-filled by the DMA itself (out-of-range lanes), so the pad
//     K slots of the dy operand, the right neighbour of column 13 and the left neighbour of column 0 (slot 15 of the row before, or the
//     zero tail of the dy tile in front of the window tile) are ordinary addresses that happen to hold zeros;
//   * rows above / below the image are MFMAs that are not issued (the image loop is unrolled, the row index is a compile-time constant).
// Workgroup = 64 co x 64 ci for all nine taps, four waves of 32 x 32 (144 accumulator registers); K split over whole images.
// LDS: a ring of seven 8-KB slots, slot c = rows 2c, 2c + 1 of the current image (dy tile 4 KB | window tile 4 KB); a slot is refilled
// with the next image's rows as soon as every wave has consumed it (one barrier per two rows), which leaves a load ten row steps
// (~2 900 cycles) to land.  16-byte chunks are XOR-swizzled by bit 1 of the pixel slot (x 4) on the source
// side: the 4 pixel x 64 byte footprint of one LDS cycle of a transposed read then covers all 64 banks once.
#ifndef R14_PIN
#define R14_PIN 0         // 1: one transposed read pinned behind each MFMA (left alone the compiler issues them in bursts in front of a fence): 103 -> 96 us
                          // stand-alone, but +0.1 ms on the step -- the steadier stream takes more from the co-resident main-stream workgroup
#endif
struct R14 {
    static constexpr int SLOT = 8192, QOFF = 4096, NSLOT = 7, LDS = NSLOT * SLOT + 128, ROWB = 2048;
};
// slabs of the previous launch of a chain (count = 0: none): `count` slabs `step` floats apart, writer's layout, to be ADDED to dw[kc4 * 4][9][C]
struct R14Prev { const float* slabs; float* dw; int count; size_t step; int kc4, C; };
__device__ __forceinline__ f32x4_t slab_load(const float* p);
__device__ __forceinline__ int r14_swz(int p) { return ((p >> 1) & 1) << 2; }
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// (256, 2): at most 256 registers per lane, arch + accumulator, so that a forward / data-gradient wave fits the same SIMD
__global__ __launch_bounds__(256, 2)
void tn_rows14_kernel(TnGeom g, const void* __restrict__ p_ptr, const void* __restrict__ q_ptr, float* __restrict__ out,
                      int co_tiles, int ci_tiles, R14Prev prev) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = lane_id(), wave = wave_id();
    // ---- chained launches: the K-split slabs of the PREVIOUS weight gradient of this stream are summed here, 1 / gridDim of them per
    //      workgroup, before this launch's own work -- instead of a reduce launch between the two.  Inside the training step that launch
    //      (576 small workgroups, 8 us alone) took 50 - 65 us: the main stream's workgroups keep every CU's register file full and it
    //      got on only as they retired, while the weight gradients -- one resident workgroup per CU -- waited behind it.
    if (prev.count > 0) {
        const uint32_t n4 = (uint32_t)prev.kc4 * 9u * (uint32_t)prev.C;
        const uint32_t per = (n4 + gridDim.x - 1) / gridDim.x;
        const uint32_t lo = blockIdx.x * per, hi = min(n4, lo + per);
        for (uint32_t i = lo + threadIdx.x; i < hi; i += 256) {
            f32x4_t a0 = f32x4_t{0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
            int sp = 0;
            for (; sp + 4 <= prev.count; sp += 4) {         // same order of additions as slab9_final_kernel
                const f32x4_t v0 = slab_load(prev.slabs + (size_t)(sp + 0) * prev.step + (size_t)i * 4);
                const f32x4_t v1 = slab_load(prev.slabs + (size_t)(sp + 1) * prev.step + (size_t)i * 4);
                const f32x4_t v2 = slab_load(prev.slabs + (size_t)(sp + 2) * prev.step + (size_t)i * 4);
                const f32x4_t v3 = slab_load(prev.slabs + (size_t)(sp + 3) * prev.step + (size_t)i * 4);
                a0 += v0; a1 += v1; a2 += v2; a3 += v3;
            }
            for (; sp < prev.count; ++sp) a0 += slab_load(prev.slabs + (size_t)sp * prev.step + (size_t)i * 4);
            const f32x4_t sum = (a0 + a1) + (a2 + a3);
            const uint32_t ci = i % (uint32_t)prev.C, qj = i / (uint32_t)prev.C;
            const uint32_t j = qj % 9u, q = qj / 9u;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t k = 4 * j + r, e = k / 9, t = k - 9 * e;
                float* o = prev.dw + ((size_t)(4 * q + e) * 9 + t) * prev.C + ci;
                *o += sum[r];
            }
        }
    }
    const TnSlot slot = tn_slot(co_tiles * ci_tiles);
    const uint32_t lin = (uint32_t)slot.tile;
    const int ci_tile = (int)(lin % (uint32_t)ci_tiles), co_tile = (int)(lin / (uint32_t)ci_tiles);
    const int img_begin = slot.split * g.ksteps_per_split;                     // here a "K step" of the geometry is one image
    const int nimg = min(g.ksteps, img_begin + g.ksteps_per_split) - img_begin;
    const int wco = wave >> 1, wci = wave & 1;
    const int co0 = co_tile * 64, ci0 = ci_tile * 64;

    f32x16_t acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    // pixel slot 0 of every dy tile and the 128 bytes behind the last slot are read (as pad K slots, against zeros of the other operand)
    // before the first load into them may have landed: they must not hold NaN patterns left by an earlier workgroup
    if (wave == 0) *reinterpret_cast<f32x4_t*>(smem + (lane >> 3) * R14::SLOT + (lane & 7) * 16) = f32x4_t{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();

    const u32x4_t rp = make_rsrc_words(p_ptr, g.p_bytes);
    const u32x4_t rq = make_rsrc_words(q_ptr, g.q_bytes);
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_ADDR(smem);

    // ---- loader: a chunk = 28 consecutive pixels (two image rows) of both operands; wave w moves the 1-KiB piece w (pixel slots 8w .. 8w+7,
    //      row j = slot >> 4, column = slot & 15; columns 14, 15: out-of-range offset = zero fill) of the dy tile and of the window tile.
    //      Chunks follow one another in the pixel stream, so the running offsets advance by a constant.
    uint32_t offp, offq;
    {
        const int prow = wave * 8 + (lane >> 3), j = prow >> 4, col = prow & 15;
        const int ce = ((lane & 7) ^ r14_swz(prow)) * 8;
        const uint32_t pix = (uint32_t)(img_begin * 196 + 14 * j + col);
        offp = (col < 14 && co0 + ce < g.ldp) ? (pix * (uint32_t)g.ldp + (uint32_t)(co0 + ce)) * 2u : OOB_OFFSET;
        offq = (col < 14 && ci0 + ce < g.C) ? (pix * (uint32_t)g.C + (uint32_t)(ci0 + ce)) * 2u : OOB_OFFSET;
    }
    const uint32_t incp = (uint32_t)(28 * g.ldp) * 2u, incq = (uint32_t)(28 * g.C) * 2u;
    int chunks_left = 7 * nimg;
    auto dma = [&](auto slot_c) {               // next chunk of the stream into ring slot SLOT
        constexpr int S = decltype(slot_c)::value;
        const bool on = chunks_left > 0;        // past the split's last image: zero fill, no traffic (the DMA count per wave stays fixed)
        if constexpr (!(FRHIP_ABL & 8)) {
            glds16_asm(rp, lds0 + (uint32_t)(S * R14::SLOT + wave * 1024), on ? offp : OOB_OFFSET);
            glds16_asm(rq, lds0 + (uint32_t)(S * R14::SLOT + R14::QOFF + wave * 1024), on ? offq : OOB_OFFSET);
        }
        offp += incp; offq += incq; --chunks_left;        // an out-of-range lane offset stays out of range: tensors are < 2 GiB
    };

    // ---- fragment addresses (slot 0, row 0 of the chunk; slot and row are instruction immediates).  16-lane group G reads a 4 pixel x 16
    //      channel block: channels 16 (G & 1) .. of the wave's 32, pixels 8 (G >> 1) + 4 half + (0..3)
    const int G = lane >> 4, fj = lane & 15, fq = fj >> 2, fp = fj & 3;
    uint32_t a_ad[2], b_ad[3][2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int px = 8 * (G >> 1) + 4 * half + fq;
        const int ca = wco * 4 + 2 * (G & 1) + (fp >> 1), cb = wci * 4 + 2 * (G & 1) + (fp >> 1);
        a_ad[half] = lds0 + (uint32_t)(px * 128 + ((ca ^ r14_swz(px)) << 4) + 8 * (fp & 1));
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int pq = px + d - 1;          // -1: the last (zero) pixel slot of the dy tile in front of the window tile
            b_ad[d][half] = lds0 + (uint32_t)(R14::QOFF + pq * 128 + ((cb ^ r14_swz(pq)) << 4) + 8 * (fp & 1));
        }
    }
    typedef __attribute__((ext_vector_type(8))) short i16x8_t;
    auto tr8 = [&](auto off_c, const uint32_t (&ad)[2]) {
        constexpr int OFF = decltype(off_c)::value;
        typedef __attribute__((address_space(3))) i16x4_t* lds_p;
        if constexpr (FRHIP_ABL & 4) {
            bf16x8_t f; asm volatile("" : "=v"(f) : "v"(ad[0]), "v"(ad[1])); return f;
        } else {
            i16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)((__attribute__((address_space(3))) char*)(uintptr_t)ad[0] + OFF));
            i16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)((__attribute__((address_space(3))) char*)(uintptr_t)ad[1] + OFF));
            return __builtin_bit_cast(bf16x8_t, (i16x8_t)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
    };

    bf16x8_t af[2];            // dy fragment of row h in af[h & 1]
    bf16x8_t rf[4][3];         // window fragments: stream row s (= row + 14 * image) in rf[s & 3][dx + 1]
    auto load_a = [&](auto row_c) {             // row 0..15 (14, 15: rows 0, 1 of the next image)
        constexpr int R = decltype(row_c)::value % 14;
        af[decltype(row_c)::value & 1] = tr8(std::integral_constant<int, (R >> 1) * R14::SLOT + (R & 1) * R14::ROWB>{}, a_ad);
    };
    auto load_r = [&](auto row_c, auto ph_c) {
        constexpr int R = decltype(row_c)::value % 14, SL = (decltype(row_c)::value + decltype(ph_c)::value) & 3;
        typedef std::integral_constant<int, (R >> 1) * R14::SLOT + (R & 1) * R14::ROWB> Off;
#pragma unroll
        for (int d = 0; d < 3; ++d) rf[SL][d] = tr8(Off{}, b_ad[d]);
    };
    auto row_step = [&](auto h_c, auto ph_c) {
        constexpr int H = decltype(h_c)::value, PH = decltype(ph_c)::value;
        // operands of the NEXT row first: they land behind this row's MFMAs
        load_a(std::integral_constant<int, H + 1>{});
        load_r(std::integral_constant<int, H + 2>{}, ph_c);
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            if (H + dy < 0 || H + dy > 13) continue;        // row outside the image: zero padding = no MFMA
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                if constexpr (FRHIP_ABL & 2) { bf16x8_t fa = af[H & 1], fb = rf[(H + dy + PH) & 3][d]; asm volatile("" :: "v"(fa), "v"(fb)); }
                else acc[(dy + 1) * 3 + d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[H & 1], rf[(H + dy + PH) & 3][d], acc[(dy + 1) * 3 + d], 0, 0, 0);
            }
        }
#if R14_PIN == 1
        // one transposed read behind each of the first eight MFMAs
#pragma unroll
        for (int z = 0; z < 8; ++z) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
#endif
    };
    auto fence = [&](auto n_c) {                // all but the n newest loads of this wave have landed; then everyone's
        // lgkmcnt(0): this wave's reads of the slot refilled behind the barrier have returned, not merely been issued
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(decltype(n_c)::value) : "memory");
        if constexpr (!(FRHIP_ABL & 1)) __builtin_amdgcn_s_barrier();
    };
    auto chunk = [&](auto c_c, auto ph_c) {
        constexpr int C = decltype(c_c)::value;
        row_step(std::integral_constant<int, 2 * C>{}, ph_c);
        row_step(std::integral_constant<int, 2 * C + 1>{}, ph_c);
        // rows 2C + 2, 2C + 3 read chunk C + 2 (issued ten row steps ago; the four chunks issued after it may still be in flight);
        // every wave has consumed chunk C: its slot takes the next image's rows (first read ten row steps from now)
        fence(std::integral_constant<int, 8>{});
        dma(c_c);
    };
    auto image = [&](auto ph_c) {
        chunk(std::integral_constant<int, 0>{}, ph_c); chunk(std::integral_constant<int, 1>{}, ph_c);
        chunk(std::integral_constant<int, 2>{}, ph_c); chunk(std::integral_constant<int, 3>{}, ph_c);
        chunk(std::integral_constant<int, 4>{}, ph_c); chunk(std::integral_constant<int, 5>{}, ph_c);
        chunk(std::integral_constant<int, 6>{}, ph_c);
    };

    // ---- prologue: the first image's seven chunks; chunks 0, 1 must have landed before row 0 starts (rows 0, 1 read rows 1 .. 3)
    dma(std::integral_constant<int, 0>{}); dma(std::integral_constant<int, 1>{}); dma(std::integral_constant<int, 2>{});
    dma(std::integral_constant<int, 3>{}); dma(std::integral_constant<int, 4>{}); dma(std::integral_constant<int, 5>{});
    dma(std::integral_constant<int, 6>{});
    fence(std::integral_constant<int, 10>{});
    load_a(std::integral_constant<int, 0>{});
    load_r(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    load_r(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
    // two images per trip (the fragment ring is back in phase after 28 rows).  An odd count runs one phantom image at the end: its loads are
    // switched off (zero fill), its MFMAs add zeros -- no second exit from the loop, whose register state would differ from the first
    for (int im = 0; im < nimg; im += 2) {
        image(std::integral_constant<int, 0>{});
        image(std::integral_constant<int, 2>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // zero-fill loads of the ring's tail: nothing may land after the workgroup has left

    // ---- epilogue: the slab layout of tn_taps9_kernel ([co / 4][j][ci][4 floats], k = 9 e + t = 4 j + r).  32 x 32 D layout: register v of
    //      lane l is row 8 (v >> 2) + 4 (l >> 5) + (v & 3), column l & 31 -- four consecutive output channels per register quad
    if constexpr (FRHIP_ABL & 16) {
        if (g.M >= 0) {
#pragma unroll
            for (int t = 0; t < 9; ++t) asm volatile("" :: "v"(acc[t]));
            return;
        }
    }
    float* dst = out + (size_t)slot.split * g.slab_stride;
    const int ci = ci0 + wci * 32 + (lane & 31);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = (co0 + wco * 32 + 8 * i + 4 * (lane >> 5)) >> 2;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            f32x4_t v;
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int k = 4 * j + r; v[r] = acc[k % 9][4 * i + k / 9]; }
            *reinterpret_cast<f32x4_t*>(dst + (((size_t)q * 9 + j) * g.C + ci) * 4) = v;
        }
    }
}

static int g_tn_taps9 = 1;
// Which tile the nine-tap weight gradients run on.  The weight gradients live on the side stream next to the main stream's
// forward / data-gradient / BatchNorm kernels, and what counts is what the two streams can do on one CU AT THE SAME TIME:
//   * the 8-wave 128 x 64 tile is the faster kernel on an empty chip (-10 %), but its 8 x 256 registers fill the CU's register
//     file: nothing of the other stream fits beside it, the streams alternate workgroup by workgroup, and the HBM-bound
//     BatchNorm passes of the main stream never overlap a weight gradient;
//   * the 4-wave 64 x 64 tile takes one wave slot per SIMD and half the registers.  With its dynamic-LDS request raised to 82 KB
//     at most ONE of its workgroups fits a CU, and beside it fit one 73-KB forward / data-gradient workgroup (4 waves) or the
//     BatchNorm passes' waves: the weight gradient then runs in the shadow of whatever the main stream is doing.
// Measured on the ResNet50 step (B = 512, two A/B rounds on one box): 29.0 ms (8-wave) -> 28.5 (4-wave) -> 28.2 ms (4-wave, one
// per CU); the gain from having a side stream at all grows from 0.9 to 1.7 ms.  FRHIP_T9_NARROW=0 / FRHIP_T9_LDS_PAD=0 restore
// the stand-alone-fastest choice (kernel micro-benchmarks use it).
static int g_t9_narrow = getenv("FRHIP_T9_NARROW") ? atoi(getenv("FRHIP_T9_NARROW")) : 1;
static int g_t9_stages4 = getenv("FRHIP_T9_STAGES4") ? atoi(getenv("FRHIP_T9_STAGES4")) : 1;     // A/B switch of the four-stage variant
static int g_t9_lds_pad = getenv("FRHIP_T9_LDS_PAD") ? atoi(getenv("FRHIP_T9_LDS_PAD")) : 83968;
// ---- lane-mask table of the TAB instantiations.  Lane l addresses the pixel rows prow_i(l) = (i >> 1) * 32 + 8 (l >> 4) + ((l & 15) >> 2)
// + 4 (i & 1) of a K step; K step t covers pixels 64 t .., and a pixel's padding predicates depend on its position inside its image
// only, so the masks repeat with period H W / gcd(H W, 64) K steps (49 for 7, 14, 28 and 56-wide square maps).  One table per geometry,
// in module memory, filled on first use (blocking copy: never inside a stream capture -- a warm-up step comes first).
constexpr int T9_TAB_GEOMS = 8, T9_TAB_MAXP = 64;
__device__ unsigned long long g_t9_masks[T9_TAB_GEOMS][T9_TAB_MAXP * 16];
static int g_t9_tab = getenv("FRHIP_T9_MASK_TABLE") ? atoi(getenv("FRHIP_T9_MASK_TABLE")) : 1;
static bool t9_mask_table(TnGeom& g, hipStream_t stream) {
    // keyed by DEVICE too: g_t9_masks is module memory, every device has its own copy at its own address
    static struct { int device, h, w, period; const unsigned long long* dev; } cache[T9_TAB_GEOMS];
    static int used = 0;
    g.mask_tab = nullptr; g.mask_period = 0;
    if (!g_t9_tab) return false;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) { (void)hipGetLastError(); return false; }
    for (int i = 0; i < used; ++i)
        if (cache[i].device == device && cache[i].h == g.H && cache[i].w == g.W) { g.mask_tab = cache[i].dev; g.mask_period = cache[i].period; return true; }
    const int hw = g.H * g.W;
    int a = hw, b = 64;
    while (b) { const int t = a % b; a = b; b = t; }
    const int period = hw / a;
    if (period > T9_TAB_MAXP || used == T9_TAB_GEOMS) return false;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {     // no blocking copy inside a capture:
        (void)hipGetLastError();                                                                  // this launch takes the VALU predicates
        return false;
    }
    static unsigned long long host[T9_TAB_MAXP * 16];
    for (int t = 0; t < period; ++t)
        for (int i = 0; i < 4; ++i) {
            unsigned long long m[4] = {0, 0, 0, 0};
            for (int l = 0; l < 64; ++l) {
                const int prow = (i >> 1) * 32 + 8 * (l >> 4) + ((l & 15) >> 2) + 4 * (i & 1);
                const int rem = (t * 64 + prow) % hw, y = rem / g.W, x = rem % g.W;
                if (y > 0) m[0] |= 1ULL << l;
                if (y < g.H - 1) m[1] |= 1ULL << l;
                if (x > 0) m[2] |= 1ULL << l;
                if (x < g.W - 1) m[3] |= 1ULL << l;
            }
            for (int d = 0; d < 4; ++d) host[(t * 4 + i) * 4 + d] = m[d];
        }
    void* sym = nullptr;
    if (hipGetSymbolAddress(&sym, HIP_SYMBOL(g_t9_masks)) != hipSuccess) return false;
    unsigned long long* dev = reinterpret_cast<unsigned long long*>(sym) + (size_t)used * T9_TAB_MAXP * 16;
    if (hipMemcpy(dev, host, sizeof(unsigned long long) * period * 16, hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); return false; }
    cache[used].device = device; cache[used].h = g.H; cache[used].w = g.W; cache[used].period = period; cache[used].dev = dev;
    ++used;
    g.mask_tab = dev; g.mask_period = period;
    return true;
}

template <int WCO, int WCI, int COF, int CIF, bool XF = false, int NST = 2, int QROWS = T9_QROWS, bool TAB = false>
static int tn_taps9_launch(const TnGeom& g, const void* p, const void* q, float* out, int splits, hipStream_t stream) {
    typedef T9Cfg<WCO, WCI, COF, CIF, NST, QROWS> Cfg;
    const int co_tiles = (g.Kc + Cfg::CO_T - 1) / Cfg::CO_T, ci_tiles = (g.C + 63) / 64;
    auto kern = tn_taps9_kernel<WCO, WCI, COF, CIF, XF, NST, QROWS, TAB>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (Cfg::NW == 4 && g_t9_lds_pad > Cfg::LDS) ? g_t9_lds_pad : Cfg::LDS) != hipSuccess) {
            set_error("igemm_tn(taps9): cannot raise dynamic LDS to %d bytes", Cfg::LDS);
            return FRHIP_ELAUNCH;
        }
        attr_done = true;
    }
    const int lds = (Cfg::NW == 4 && g_t9_lds_pad > Cfg::LDS) ? g_t9_lds_pad : Cfg::LDS;
    hipLaunchKernelGGL(kern, dim3(co_tiles * ci_tiles * splits), dim3(64 * Cfg::NW), lds, stream, g, p, q, out, co_tiles, ci_tiles);
    return check_launch("igemm_tn(taps9)");
}

// rows kernel (14 x 14 maps): K split over whole images, one workgroup per CU beside a forward / data-gradient workgroup (the 82-KB request)
static int g_t9_rows = getenv("FRHIP_T9_ROWS") ? atoi(getenv("FRHIP_T9_ROWS")) : 1;
static int tn_rows14_launch(const TnGeom& g, const void* p, const void* q, float* out, int splits, hipStream_t stream,
                            const R14Prev& prev = R14Prev{nullptr, nullptr, 0, 0, 0, 0}) {
    const int co_tiles = g.Kc / 64, ci_tiles = g.C / 64;
    const int lds = g_t9_lds_pad > R14::LDS ? g_t9_lds_pad : R14::LDS;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(tn_rows14_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            set_error("igemm_tn(rows14): cannot raise dynamic LDS to %d bytes", lds);
            return FRHIP_ELAUNCH;
        }
        attr_done = true;
    }
    hipLaunchKernelGGL(tn_rows14_kernel, dim3(co_tiles * ci_tiles * splits), dim3(256), lds, stream, g, p, q, out, co_tiles, ci_tiles, prev);
    return check_launch("igemm_tn(rows14)");
}

// Sum of K-split slabs, float4 per thread, no atomics (deterministic).  blockIdx.y = g owns slabs g*per_group ..:
//   final == 0: their sum overwrites the group's first slab (first level of a two-level tree)
//   final == 1: out[i] += sum (gridDim.y must be 1)
__device__ __forceinline__ f32x4_t slab_load(const float* p) {
    if constexpr (TN_SLAB_NT & 2) return __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p));
    else return *reinterpret_cast<const f32x4_t*>(p);
}

// At most 32 VGPRs per lane (two accumulators, two loads in flight): the persistent 8-wave linear kernels of the main stream (240 VGPRs, two
// waves per SIMD) leave exactly 32 registers per lane free on every CU, and with 46 this reduce -- 54 launches per Swin34 step on the side
// stream -- waited for a whole linear launch to retire every time (47.9 us per launch inside the step against 7.0 us alone).
#ifndef SLAB_WIDE
#define SLAB_WIDE 0
#endif
#ifndef SLAB_VGPR
#define SLAB_VGPR 32
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(SLAB_VGPR)))
void slab_reduce_kernel(float* __restrict__ slabs, int count, int per_group, size_t step, float* __restrict__ out, size_t n4, int final) {
    const int s0 = blockIdx.y * per_group, s1 = min(count, s0 + per_group);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4_t a0 = f32x4_t{0.f, 0.f, 0.f, 0.f}, a1 = a0;
        const float* src = slabs + (size_t)s0 * step + i * 4;
        int sp = s0;
#if SLAB_WIDE        // A/B build (-DSLAB_WIDE=1 -DSLAB_VGPR=64): four loads in flight, 46 registers -- the round-3 kernel
        f32x4_t a2 = a0, a3 = a0;
        for (; sp + 4 <= s1; sp += 4) {
            const f32x4_t v0 = slab_load(src), v1 = slab_load(src + step), v2 = slab_load(src + 2 * step), v3 = slab_load(src + 3 * step);
            src += 4 * step;
            a0 += v0; a1 += v1; a2 += v2; a3 += v3;
        }
        a0 += a2; a1 += a3;
#endif
#pragma unroll 1
        for (; sp + 2 <= s1; sp += 2) {
            const f32x4_t v0 = slab_load(src);
            const f32x4_t v1 = slab_load(src + step);
            src += 2 * step;
            a0 += v0; a1 += v1;
        }
        if (sp < s1) a0 += *reinterpret_cast<const f32x4_t*>(src);
        f32x4_t acc = a0 + a1;
        if (final) {
            acc += reinterpret_cast<const f32x4_t*>(out)[i];
            reinterpret_cast<f32x4_t*>(out)[i] = acc;
        } else {
            *reinterpret_cast<f32x4_t*>(slabs + (size_t)s0 * step + i * 4) = acc;
        }
    }
}

// Final level of the nine-tap split-K reduction: sums `count` slabs in the writer's layout ([co/4][j][ci][4], k = 4 j + r = 9 e + t)
// and ADDS the result to dw[co][tap][ci].  One thread per (co/4, j, ci) float4; lanes = consecutive ci, so slab reads are
// 16 bytes per lane contiguous and the four output writes of a lane are each coalesced over the wave.
__global__ __launch_bounds__(256) void slab9_final_kernel(const float* __restrict__ slabs, int count, size_t step,
                                                          float* __restrict__ out, int kc4, int C) {
    const size_t n4 = (size_t)kc4 * 9 * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4_t a0 = f32x4_t{0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
        int sp = 0;
        for (; sp + 4 <= count; sp += 4) {
            const f32x4_t v0 = slab_load(slabs + (size_t)(sp + 0) * step + i * 4);
            const f32x4_t v1 = slab_load(slabs + (size_t)(sp + 1) * step + i * 4);
            const f32x4_t v2 = slab_load(slabs + (size_t)(sp + 2) * step + i * 4);
            const f32x4_t v3 = slab_load(slabs + (size_t)(sp + 3) * step + i * 4);
            a0 += v0; a1 += v1; a2 += v2; a3 += v3;
        }
        for (; sp < count; ++sp) a0 += *reinterpret_cast<const f32x4_t*>(slabs + (size_t)sp * step + i * 4);
        const f32x4_t acc = (a0 + a1) + (a2 + a3);
        const int ci = (int)(i % (size_t)C);
        const size_t qj = i / (size_t)C;
        const int j = (int)(qj % 9), q = (int)(qj / 9);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 4 * j + r, e = k / 9, t = k - 9 * e;
            float* o = out + ((size_t)(4 * q + e) * 9 + t) * C + ci;
            *o += acc[r];
        }
    }
}

template <typename T, int RB>
static int tn_launch(const TnGeom& g, const void* p, const void* q, float* out, int taps, int splits, hipStream_t stream) {
    typedef TnTile<T, RB> Tile;
    const int co_tiles = (g.Kc + Tile::BC - 1) / Tile::BC, ci_tiles = (g.C + Tile::BC - 1) / Tile::BC;
    auto kern = tn_kernel<T, RB>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Tile::LDS_BYTES) != hipSuccess) {
            set_error("igemm_tn: cannot raise dynamic LDS to %d bytes", Tile::LDS_BYTES);
            return FRHIP_ELAUNCH;
        }
        attr_done = true;
    }
    dim3 grid(co_tiles * ci_tiles * taps * splits);
    hipLaunchKernelGGL(kern, grid, dim3(TN_THREADS), Tile::LDS_BYTES, stream, g, p, q, out, co_tiles, ci_tiles, taps);
    return check_launch("igemm_tn");
}

// Decide where the K splits put their results: private slabs in the caller's workspace (plain stores + one
// deterministic reduce pass) when it is large enough, else fp32 atomics straight into `out`.
static float* tn_pick_dst(TnGeom& g, float* out, int splits, size_t out_elems, float* ws, size_t ws_bytes) {
    g.slab_stride = 0;
    // same-address fp32 atomics from hundreds of K splits serialise in L2 (75 us for 500 splits of a 147 KB tile);
    // slabs + a grouped reduce pass cost two streaming passes over splits * out bytes instead
    if (splits > 1 && ws && (out_elems % 4) == 0 && out_elems >= 16384 &&
        (size_t)splits * out_elems * sizeof(float) <= ws_bytes) {
        g.slab_stride = (int)out_elems;
        return ws;
    }
    return out;
}

static int tn_finish(const TnGeom& g, float* out, int splits, size_t out_elems, float* ws, hipStream_t stream) {
    if (!g.slab_stride) return FRHIP_OK;
    const size_t n4 = out_elems / 4;
    int blocks = (int)((n4 + 255) / 256); if (blocks > 2048) blocks = 2048;
    // one pass while it has enough blocks to stream (or few slabs); else a two-level tree: groups of slabs first
    int groups = 1;
    while (splits > 32 * groups && blocks * groups < 512) groups *= 2;
    if (groups > 1) {
        const int per_group = (splits + groups - 1) / groups;
        groups = (splits + per_group - 1) / per_group;
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks, groups), dim3(256), 0, stream, ws, splits, per_group,
                           out_elems, (float*)nullptr, n4, 0);
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks, 1), dim3(256), 0, stream, ws, groups, groups,
                           (size_t)per_group * out_elems, out, n4, 1);
    } else {
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks, 1), dim3(256), 0, stream, ws, splits, splits, out_elems, out, n4, 1);
    }
    return check_launch("igemm_tn(slab reduce)");
}

// nine-tap kernel: its slabs are in the writer's layout (see tn_taps9_kernel); groups of slabs are pre-summed in that layout
// by the generic kernel, the final level converts to dw[co][tap][ci]
static int t9_finish(const TnGeom& g, float* out, int splits, size_t out_elems, float* ws, hipStream_t stream) {
    if (!g.slab_stride) return FRHIP_OK;
    const size_t n4 = out_elems / 4;
    int blocks = (int)((n4 + 255) / 256); if (blocks > 2048) blocks = 2048;
    int groups = 1;
    while (splits > 32 * groups && blocks * groups < 512) groups *= 2;
    int count = splits; size_t step = out_elems;
    if (groups > 1) {
        const int per_group = (splits + groups - 1) / groups;
        groups = (splits + per_group - 1) / per_group;
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks, groups), dim3(256), 0, stream, ws, splits, per_group,
                           out_elems, (float*)nullptr, n4, 0);
        count = groups; step = (size_t)per_group * out_elems;
    }
    hipLaunchKernelGGL(slab9_final_kernel, dim3(blocks), dim3(256), 0, stream, ws, count, step, out, g.Kc / 4, g.C);
    return check_launch("igemm_tn(nine-tap slab reduce)");
}

static bool t9_applicable(int dtype, int w, int c, int r, int s, int stride, int pad, long long M, int ldp) {
    const int es = 2;
    return g_tn_taps9 && dtype == FRHIP_DT_BF16 && r == 3 && s == 3 && stride == 1 && pad == 1 && w <= T9_MAXW &&
           1LL * (M + 64 + 2LL * w + 2) * (ldp > c ? ldp : c) * es < 0x7fffffffLL;
}

static bool r14_applicable(int dtype, int h, int w, int c, int kc, int r, int s, int stride, int pad, long long M, int ldp) {
    return g_t9_rows && g_t9_narrow && t9_applicable(dtype, w, c, r, s, stride, pad, M, ldp) && h == 14 && w == 14 && (c % 64) == 0 && (kc % 64) == 0;
}
// K split of the rows kernel over images: whole rounds of one workgroup per CU; the epilogue (nine store passes of 36 KB per wave) costs
// about two images.  Sets g.ksteps (= images) / ksteps_per_split and returns the split count.
static int r14_plan(TnGeom& g, int n, int splits) {
    const long long tiles = 1LL * (g.Kc / 64) * (g.C / 64);
    int best = 1; double best_t = 1e30;
    for (int sp = 1; sp <= n && sp <= 1024; ++sp) {
        const long long rounds = (tiles * sp + 255) / 256;
        const double t = (double)rounds * ((double)((n + sp - 1) / sp) + 2.0);
        if (t < best_t * 0.98) { best_t = t; best = sp; }
    }
    int sp = splits > 0 ? splits : best;
    if (sp > n) sp = n;
    const int per = (n + sp - 1) / sp;
    g.ksteps = n; g.ksteps_per_split = per;
    return (n + per - 1) / per;
}

static int tn_run(int dtype, const void* p, const void* q, float* out, int n, int h, int w, int c, int kc, int ldp,
                  int r, int s, int stride, int pad, int splits, float* ws, size_t ws_bytes, hipStream_t stream,
                  const char* who, bool overwrite = false, const float* xf_scale = nullptr, const float* xf_shift = nullptr) {
    const int es = dtype == FRHIP_DT_BF16 ? 2 : 4;
    if (dtype != FRHIP_DT_BF16 && dtype != FRHIP_DT_F32) { set_error("%s: bad dtype %d", who, dtype); return FRHIP_EINVAL; }
    const int epv = 16 / es;
    if (n <= 0 || c <= 0 || kc <= 0 || (c % epv) || (ldp % epv) || ldp < kc) {
        set_error("%s: unsupported shape c=%d kc=%d ldp=%d (c and ldp must be multiples of %d, ldp >= kc)", who, c, kc, ldp, epv);
        return FRHIP_EINVAL;
    }
    TnGeom g;
    g.H = h; g.W = w; g.C = c; g.R = r; g.S = s; g.stride = stride; g.pad = pad;
    g.Ho = (h + 2 * pad - r) / stride + 1; g.Wo = (w + 2 * pad - s) / stride + 1;
    const long long M = 1LL * n * g.Ho * g.Wo;
    const long long pb = M * ldp * es, qb = 1LL * n * h * w * c * es;
    if (pb > 0x7fffffffLL || qb > 0x7fffffffLL) { set_error("%s: tensor exceeds the 2 GiB buffer window", who); return FRHIP_EINVAL; }
    g.M = (int)M; g.Kc = kc; g.ldp = ldp;
    g.p_bytes = (uint32_t)pb; g.q_bytes = (uint32_t)qb;
    g.d_howo = make_fastdiv((uint32_t)(g.Ho * g.Wo)); g.d_wo = make_fastdiv((uint32_t)g.Wo);
    g.adv_n = TN_KP / (g.Ho * g.Wo);
    g.adv_ho = (TN_KP % (g.Ho * g.Wo)) / g.Wo;
    g.adv_wo = TN_KP % g.Wo;
    g.ksteps = (g.M + TN_KP - 1) / TN_KP;
    g.slab_stride = 0;
    g.xf_scale = xf_scale; g.xf_shift = xf_shift;
    g.mask_tab = nullptr; g.mask_period = 0;
    if (xf_scale && !(t9_applicable(dtype, w, c, r, s, stride, pad, M, ldp) && (c % 64) == 0)) {
        set_error("%s: operand transform needs the nine-tap kernel (bf16 3x3 stride 1, c %% 64 == 0)", who);
        return FRHIP_EINVAL;
    }
    const int taps = r * s;
    const size_t out_elems = (size_t)kc * taps * c;
    if (out_elems > 0x7fffffffULL) { set_error("%s: output too large", who); return FRHIP_EINVAL; }
    int rc;
    // The nine-tap kernel covers every 3x3/s1/p1 bf16 layer (g_tn_taps9: 0 off, 1/2 on); wide = 128-co tiles.
    if (!xf_scale && ws && r14_applicable(dtype, h, w, c, kc, r, s, stride, pad, M, ldp)) {
        TnGeom gi = g;
        int sp = r14_plan(gi, n, splits);
        float* dst = tn_pick_dst(gi, out, sp, out_elems, ws, ws_bytes);
        if (gi.slab_stride) {
            rc = tn_rows14_launch(gi, p, q, dst, sp, stream);
            return rc ? rc : t9_finish(gi, out, sp, out_elems, ws, stream);
        }
    }
    if (t9_applicable(dtype, w, c, r, s, stride, pad, M, ldp)) {
        const bool wide = kc > 64 && !g_t9_narrow;
        const int co_t = wide ? 128 : 64;
        const long long tiles = 1LL * ((kc + co_t - 1) / co_t) * ((c + 63) / 64);
        const int slots = 256 * ((wide || g_t9_lds_pad > 81920) ? 1 : 2);
        int best = 1; double best_t = 1e30;
        const int max_splits = g.ksteps / 8 > 0 ? g.ksteps / 8 : 1;
        for (int sp = 1; sp <= max_splits && sp <= 1024; ++sp) {
            const long long rounds = (tiles * sp + slots - 1) / slots;
            const double t = (double)rounds * ((double)g.ksteps / sp + 8.0);       // epilogue ~ 8 K steps (nine store passes)
            if (t < best_t * 0.98) { best_t = t; best = sp; }
        }
        if (splits <= 0) splits = best;
        if (splits > g.ksteps) splits = g.ksteps;
        g.ksteps_per_split = (g.ksteps + splits - 1) / splits;
        splits = (g.ksteps + g.ksteps_per_split - 1) / g.ksteps_per_split;
        float* dst = (kc % 4 == 0) ? tn_pick_dst(g, out, splits, out_elems, ws, ws_bytes) : out;      // slab layout packs co in fours
        const bool deep = !wide && 64 + 2 * w + 2 <= 128;      // three stages of a 128-row window
        // W <= 14: a 96-row window, 20-KB stages -- FOUR fit the co-resident tile's 82-KB request, so an operand load has three K steps to land
        const bool deep4 = deep && g_t9_stages4 && 64 + 2 * w + 2 <= 96 && !xf_scale;
        if (xf_scale) rc = wide ? tn_taps9_launch<2, 4, 4, 1, true>(g, p, q, dst, splits, stream)
                         : deep ? tn_taps9_launch<1, 4, 4, 1, true, 3, 128>(g, p, q, dst, splits, stream)
                                : tn_taps9_launch<1, 4, 4, 1, true>(g, p, q, dst, splits, stream);
        else if (!wide && g.Ho == g.H && g.Wo == g.W && t9_mask_table(g, stream))
            rc = deep4 ? tn_taps9_launch<1, 4, 4, 1, false, 4, 96, true>(g, p, q, dst, splits, stream)
               : deep ? tn_taps9_launch<1, 4, 4, 1, false, 3, 128, true>(g, p, q, dst, splits, stream)
                      : tn_taps9_launch<1, 4, 4, 1, false, 2, T9_QROWS, true>(g, p, q, dst, splits, stream);
        else rc = wide ? tn_taps9_launch<2, 4, 4, 1>(g, p, q, dst, splits, stream)
                : deep4 ? tn_taps9_launch<1, 4, 4, 1, false, 4, 96>(g, p, q, dst, splits, stream)
                : deep ? tn_taps9_launch<1, 4, 4, 1, false, 3, 128>(g, p, q, dst, splits, stream)
                       : tn_taps9_launch<1, 4, 4, 1>(g, p, q, dst, splits, stream);
        return rc ? rc : t9_finish(g, out, splits, out_elems, ws, stream);
    }
    const bool big = (c * es >= 256) && (kc * es >= 256);
    if (splits <= 0) {
        // Split-K heuristic: the grid should fill whole "rounds" of the chip (256 CUs x resident workgroups) with as
        // few splits as possible -- every extra split adds a full fp32 atomic pass over the output tile and a
        // partially filled last round wastes up to a third of the launch.
        const int bc = (big ? 256 : 128) / es;
        const long long tiles = 1LL * ((kc + bc - 1) / bc) * ((c + bc - 1) / bc) * taps;
        const int slots = 256 * (big ? 2 : 4);                 // LDS-limited residency: 2 (64 KB) / 4 (32 KB) per CU
        const int max_splits = (g.ksteps + 7) / 8 > 0 ? (g.ksteps + 7) / 8 : 1;     // >= 8 K steps per workgroup
        int best = 1; double best_score = -1.0;
        for (int sp = 1; sp <= max_splits && sp <= 512; ++sp) {
            const long long wgs = tiles * sp;
            const long long rounds = (wgs + slots - 1) / slots;
            const double fill = (double)wgs / (double)(rounds * slots);
            // time ~ rounds * (ksteps/sp + epilogue cost in K-step units)
            const double t = (double)rounds * ((double)g.ksteps / sp + 6.0);
            const double score = 1.0 / t;
            if (score > best_score * 1.02 || best_score < 0) { best_score = score; best = sp; }
            (void)fill;
        }
        splits = best;
    }
    if (splits > g.ksteps) splits = g.ksteps;
    g.ksteps_per_split = (g.ksteps + splits - 1) / splits;
    splits = (g.ksteps + g.ksteps_per_split - 1) / g.ksteps_per_split;
    if (overwrite && splits == 1) {
        // out = result (not +=) and a single K split: the "slab" store path with slab 0 = out itself -- plain coalesced stores,
        // no zero fill by the caller and no fp32 atomic read-modify-write pass over the output (the head's 250-MB dW)
        g.slab_stride = (int)out_elems;
        if (dtype == FRHIP_DT_BF16) return big ? tn_launch<bf16_t, 256>(g, p, q, out, taps, 1, stream) : tn_launch<bf16_t, 128>(g, p, q, out, taps, 1, stream);
        return big ? tn_launch<float, 256>(g, p, q, out, taps, 1, stream) : tn_launch<float, 128>(g, p, q, out, taps, 1, stream);
    }
    if (overwrite && hipMemsetAsync(out, 0, out_elems * sizeof(float), stream) != hipSuccess) {
        set_error("%s: cannot clear the output", who);
        return FRHIP_ELAUNCH;
    }
    float* dst = tn_pick_dst(g, out, splits, out_elems, ws, ws_bytes);
    if (dtype == FRHIP_DT_BF16) rc = big ? tn_launch<bf16_t, 256>(g, p, q, dst, taps, splits, stream)
                                              : tn_launch<bf16_t, 128>(g, p, q, dst, taps, splits, stream);
    else rc = big ? tn_launch<float, 256>(g, p, q, dst, taps, splits, stream)
                  : tn_launch<float, 128>(g, p, q, dst, taps, splits, stream);
    return rc ? rc : tn_finish(g, out, splits, out_elems, ws, stream);
}


// ------------------------------------------------------------------------------------------------------------------
// Head: gradient of the class centres with the normalise-backward fused (nets/PartialFC.py:464-484 autograd of F.normalize(weight) and
// F.linear): d_w[c][:] = (g[c][:] - what[c][:] <g[c], what[c]>) * out_scale / ||w_c||, g = dT^T E (contraction over the n samples).
// The per-tap kernel + frhip_l2norm_bwd wrote g (250 MB at 122 000 classes) and read it back with what: 163 + 104 us on the main stream in
// front of the backbone's backward pass.  Here a workgroup owns 64 classes x all 512 dimensions (4 waves x (64 x 128), 128 accumulator
// registers), so every class row is complete in one workgroup: g never leaves the chip.  K step = 32 samples; a stage = nine 4-KB blocks
// [32 samples][128 B] (block 0: the 64 classes of dT, blocks 1..8: 64 dimensions of E each), both operands K-strided as in tn_kernel
// (transposed LDS reads, source-side XOR swizzle); two stages, inline-assembly loads.  bf16, D = 512 only.
struct Hdw {
    static constexpr int CT = 64, KS = 32, BLK = KS * 128, NBLK = 9, STAGE = NBLK * BLK, LDS = 2 * STAGE;      // 72 KB: two workgroups per CU
    static constexpr int EP = 132 * 4;                                                                          // epilogue row pitch (bytes): 128 floats + pad
};
__global__ __launch_bounds__(256, 2)
void head_dw_kernel(const void* __restrict__ dt_ptr, int ldt, const void* __restrict__ e_ptr, const bf16_t* __restrict__ what,
                    const float* __restrict__ wnorm, float* __restrict__ dw, int n, int classes, uint32_t dt_bytes, uint32_t e_bytes,
                    float out_scale) {
    typedef bf16_t T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = lane_id(), wave = wave_id();
    const int c0 = blockIdx.x * Hdw::CT;
    const u32x4_t rp = make_rsrc_words(dt_ptr, dt_bytes);
    const u32x4_t rq = make_rsrc_words(e_ptr, e_bytes);
    const uint32_t lds_base = (uint32_t)(uintptr_t)LDS_ADDR(smem);

    f32x4_t acc[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // loader: wave w moves piece w (sample rows 8w .. 8w+7 of the step) of every block
    const int prow = wave * 8 + (lane >> 3);
    const int ce = ((lane & 7) ^ tn_swz<128>(prow)) * 8;                  // first element of this lane's (logical) 16-byte chunk
    const uint32_t offp0 = (c0 + ce < ldt) ? (uint32_t)((prow * ldt + c0 + ce) * 2) : OOB_OFFSET;
    const uint32_t offq0 = (uint32_t)((prow * 512 + ce) * 2);
    const uint32_t incp = (uint32_t)(Hdw::KS * ldt * 2), incq = (uint32_t)(Hdw::KS * 512 * 2);
    const int nks = (n + Hdw::KS - 1) / Hdw::KS;
    auto stage = [&](int buf, int ks) {
        const uint32_t sb = lds_base + (uint32_t)(buf * Hdw::STAGE + wave * 1024);
        // rows past n lie beyond the tensors' bytes -> zero fill
        glds16_asm(rp, sb, offp0 == OOB_OFFSET ? OOB_OFFSET : offp0 + (uint32_t)ks * incp);
#pragma unroll
        for (int b = 0; b < 8; ++b) glds16_asm(rq, sb + (uint32_t)((1 + b) * Hdw::BLK), offq0 + (uint32_t)ks * incq + (uint32_t)(b * 128));
    };
    auto compute = [&](int buf) {
        const char* st = smem + buf * Hdw::STAGE;
        bf16x8_t pf[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) pf[a] = TnFrag<T, 128>::load(st, 0, a * 16, lane);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bf16x8_t qf = TnFrag<T, 128>::load(st + (1 + 2 * wave + (b >> 2)) * Hdw::BLK, 0, (b & 3) * 16, lane);
#pragma unroll
            for (int a = 0; a < 4; ++a) Mma<T>::run(pf[a], qf, acc[a][b]);
        }
    };
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int cur = 0;
    for (int ks = 0; ks < nks - 1; ++ks) {
        stage(cur ^ 1, ks + 1);
        compute(cur);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // next stage landed; this stage's reads returned before it is refilled
        __builtin_amdgcn_s_barrier();
        cur ^= 1;
    }
    compute(cur);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- epilogue, sixteen classes at a time: the wave's 16 x 128 slice through LDS (row = class), per-row dot product with what over the
    //      wave's 128 dimensions, summed over the four waves, then d_w = (g - what * dot) * out_scale / ||w||
    float* red = reinterpret_cast<float*>(smem + 4 * 16 * Hdw::EP);       // [4 waves][16 rows]
    char* mine = smem + wave * 16 * Hdw::EP;
    const int fi = lane & 15, fg = lane >> 4;
    const int r = lane >> 2, seg = lane & 3;                               // read phase: row r of the sixteen, 32-dimension segment seg
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) *reinterpret_cast<float*>(mine + (4 * fg + e) * Hdw::EP + (b * 16 + fi) * 4) = acc[a][b][e];
        const int c = c0 + a * 16 + r;
        const bool live = c < classes;
        f32x4_t g[8];
        bf16x8_t h[4];
#pragma unroll
        for (int v = 0; v < 8; ++v) g[v] = *reinterpret_cast<const f32x4_t*>(mine + r * Hdw::EP + (seg * 32 + v * 4) * 4);
        const size_t off = (size_t)(live ? c : 0) * 512 + wave * 128 + seg * 32;
#pragma unroll
        for (int v = 0; v < 4; ++v) h[v] = *reinterpret_cast<const bf16x8_t*>(what + off + v * 8);
        float dot = 0.f;
#pragma unroll
        for (int v = 0; v < 8; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e) dot += g[v][e] * (float)h[v >> 1][(v & 1) * 4 + e];
        dot += __shfl_xor(dot, 1);
        dot += __shfl_xor(dot, 2);
        if (seg == 0) red[wave * 16 + r] = dot;
        __syncthreads();
        const float total = (red[r] + red[16 + r]) + (red[32 + r] + red[48 + r]);
        if (live) {
            const float inv = out_scale / wnorm[c];
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                f32x4_t o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (g[v][e] - (float)h[v >> 1][(v & 1) * 4 + e] * total) * inv;
                *reinterpret_cast<f32x4_t*>(dw + off + v * 4) = o;
            }
        }
        __syncthreads();                                                   // red and the staging rows are rewritten by the next sixteen classes
    }
}

static int head_dw_launch(const void* dt, int ldt, const void* ehat, const void* what, const float* wnorm, float* dw, int n, int classes,
                          float out_scale, hipStream_t stream) {
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(head_dw_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, Hdw::LDS) != hipSuccess) {
            set_error("frhip_head_dw: cannot raise dynamic LDS to %d bytes", Hdw::LDS);
            return FRHIP_ELAUNCH;
        }
        attr_done = true;
    }
    hipLaunchKernelGGL(head_dw_kernel, dim3((classes + Hdw::CT - 1) / Hdw::CT), dim3(256), Hdw::LDS, stream, dt, ldt, ehat, (const bf16_t*)what,
                       wnorm, dw, n, classes, (uint32_t)((size_t)n * ldt * 2), (uint32_t)((size_t)n * 512 * 2), out_scale);
    return check_launch("frhip_head_dw");
}

}  // namespace frhip

using namespace frhip;

extern "C" int frhip_set_wgrad_taps9(int enabled) {
    // 1 (default): 3x3 / stride-1 bf16 weight gradients on the nine-tap kernel; 0: on the per-tap gather kernel; < 0 queries
    const int old = g_tn_taps9;
    if (enabled >= 0) g_tn_taps9 = enabled ? 1 : 0;
    return old;
}

extern "C" int frhip_conv_wgrad(int dtype, const void* dy, const void* x, float* dw, int n, int h, int w, int c,
                                int k, int r, int s, int stride, int pad, int splits, float* workspace,
                                size_t workspace_bytes, hipStream_t stream) {
    // dw[k][r][s][c] (fp32, caller-zeroed) += sum over output pixels of dy[m][k] * x[pix(m,r,s)][c]
    return tn_run(dtype, dy, x, dw, n, h, w, c, k, k, r, s, stride, pad, splits, workspace, workspace_bytes, stream,
                  "frhip_conv_wgrad");
}

// ---- chained weight gradients (rows kernel): launch i sums the K-split slabs of launch i - 1 in its prologue ----------------------
extern "C" int frhip_conv_wgrad_chain_ok(int dtype, int n, int h, int w, int c, int k, int r, int s, int stride, int pad) {
    return (n >= 2 && frhip::r14_applicable(dtype, h, w, c, k, r, s, stride, pad, 1LL * n * h * w, k)) ? 1 : 0;
}

extern "C" int frhip_conv_wgrad_chain(int dtype, const void* dy, const void* x, int n, int h, int w, int c, int k,
                                      float* slabs, size_t slab_bytes, float* prev_dw, const float* prev_slabs, int prev_k, int prev_c,
                                      int prev_splits, int* splits_out, hipStream_t stream) {
    if (!frhip_conv_wgrad_chain_ok(dtype, n, h, w, c, k, 3, 3, 1, 1) || !slabs || !splits_out) {
        set_error("frhip_conv_wgrad_chain: shape not served by the rows kernel (bf16 3x3 stride 1 on 14 x 14 maps, c and k multiples of 64, n >= 2)");
        return FRHIP_EINVAL;
    }
    if (prev_splits > 0 && (!prev_dw || !prev_slabs || prev_k <= 0 || prev_c <= 0 || (prev_k % 4) || prev_slabs == slabs)) {
        set_error("frhip_conv_wgrad_chain: bad previous-launch descriptor (its slabs must not be this launch's)");
        return FRHIP_EINVAL;
    }
    TnGeom g;
    g.H = h; g.W = w; g.C = c; g.R = 3; g.S = 3; g.stride = 1; g.pad = 1; g.Ho = h; g.Wo = w;
    const long long M = 1LL * n * h * w;
    g.M = (int)M; g.Kc = k; g.ldp = k;
    g.p_bytes = (uint32_t)(M * k * 2); g.q_bytes = (uint32_t)(M * c * 2);
    g.d_howo = make_fastdiv((uint32_t)(h * w)); g.d_wo = make_fastdiv((uint32_t)w);
    g.adv_n = g.adv_ho = g.adv_wo = 0;
    g.xf_scale = g.xf_shift = nullptr; g.mask_tab = nullptr; g.mask_period = 0;
    const int sp = r14_plan(g, n, 0);
    const size_t out_elems = (size_t)k * 9 * c;
    if (sp < 2 || (size_t)sp * out_elems * sizeof(float) > slab_bytes) {
        set_error("frhip_conv_wgrad_chain: %d K splits of %zu bytes do not fit the slab buffer", sp, out_elems * sizeof(float));
        return FRHIP_EINVAL;
    }
    g.slab_stride = (int)out_elems;
    R14Prev prev{prev_slabs, prev_dw, prev_splits > 0 ? prev_splits : 0, (size_t)prev_k * 9 * prev_c, prev_k / 4, prev_c};
    *splits_out = sp;
    return tn_rows14_launch(g, dy, x, slabs, sp, stream, prev);
}

extern "C" int frhip_conv_wgrad_chain_finish(float* dw, float* slabs, int k, int c, int splits, hipStream_t stream) {
    if (!dw || !slabs || k <= 0 || c <= 0 || (k % 4) || splits <= 0) { set_error("frhip_conv_wgrad_chain_finish: bad descriptor"); return FRHIP_EINVAL; }
    TnGeom g; g.Kc = k; g.C = c; g.slab_stride = k * 9 * c;
    return t9_finish(g, dw, splits, (size_t)k * 9 * c, slabs, stream);
}

extern "C" int frhip_head_dw_ok(int dtype, int n, int classes, int d) {
    return (dtype == FRHIP_DT_BF16 && d == 512 && n > 0 && classes > 0 && (long long)n * ((classes + 7) / 8 * 8) * 2 < 0x7fffffffLL) ? 1 : 0;
}

extern "C" int frhip_head_dw(int dtype, const void* dt, int ldt, const void* ehat, const void* what, const float* wnorm, float* dw,
                             int n, int classes, int d, float out_scale, hipStream_t stream) {
    if (!frhip_head_dw_ok(dtype, n, classes, d) || !dt || !ehat || !what || !wnorm || !dw || ldt < classes || (ldt % 8) ||
        (long long)n * ldt * 2 >= 0x7fffffffLL) {
        set_error("frhip_head_dw: bf16, d == 512, ldt >= classes in whole eights and dT below 2 GiB are required");
        return FRHIP_EINVAL;
    }
    return head_dw_launch(dt, ldt, ehat, what, wnorm, dw, n, classes, out_scale, stream);
}

extern "C" int frhip_gemm_tn_overwrite(int dtype, const void* p, const void* q, float* out, int m, int kc, int ldp, int c,
                                       float* workspace, size_t workspace_bytes, hipStream_t stream) {
    // out[kc][c] (fp32, need not be initialised) = sum_m p[m][0..kc) (pitch ldp) * q[m][0..c)
    return tn_run(dtype, p, q, out, m, 1, 1, c, kc, ldp, 1, 1, 1, 0, 0, workspace, workspace_bytes, stream,
                  "frhip_gemm_tn_overwrite", true);
}

extern "C" int frhip_gemm_tn(int dtype, const void* p, const void* q, float* out, int m, int kc, int ldp, int c,
                             int splits, float* workspace, size_t workspace_bytes, hipStream_t stream) {
    // out[kc][c] (fp32, caller-zeroed) += sum_m p[m][0..kc) (pitch ldp) * q[m][0..c)
    return tn_run(dtype, p, q, out, m, 1, 1, c, kc, ldp, 1, 1, 1, 0, splits, workspace, workspace_bytes, stream,
                  "frhip_gemm_tn");
}

extern "C" int frhip_conv_wgrad_bnrelu_fusable(int dtype, int n, int h, int w, int c, int k, int r, int s, int stride, int pad) {
    return (frhip::t9_applicable(dtype, w, c, r, s, stride, pad, 1LL * n * h * w, k) && (c % 64) == 0) ? 1 : 0;
}

extern "C" int frhip_conv_wgrad_bnrelu(int dtype, const void* dy, const void* x, const float* in_scale, const float* in_shift,
                                       float* dw, int n, int h, int w, int c, int k, int r, int s, int stride, int pad, int splits,
                                       float* workspace, size_t workspace_bytes, hipStream_t stream) {
    // dw += wgrad(dy, relu(x * in_scale[c] + in_shift[c])): weight gradient of the convolution behind a BatchNorm + ReLU whose
    // output was never materialised (frhip_conv_fwd_bnrelu); the activation is re-formed in LDS from the saved BatchNorm input
    if (!in_scale || !in_shift) { frhip::set_error("frhip_conv_wgrad_bnrelu: scale / shift required"); return FRHIP_EINVAL; }
    return tn_run(dtype, dy, x, dw, n, h, w, c, k, k, r, s, stride, pad, splits, workspace, workspace_bytes, stream,
                  "frhip_conv_wgrad_bnrelu", false, in_scale, in_shift);
}
