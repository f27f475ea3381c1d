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
#include "common.h"
#include "frhip.h"

namespace frhip {

struct TnGeom {
    int H, W, C;             // Q tensor
    int Ho, Wo, R, S, stride, pad;
    int M, Kc, ldp;          // GEMM-K rows, valid P columns (= output rows), P row pitch (elements)
    int ksteps, ksteps_per_split;
    FastDiv d_howo, d_wo;
    int adv_wo, adv_ho, adv_n;   // 64 pixels = adv_n images + adv_ho rows + adv_wo columns (mixed-radix step per K step)
    uint32_t p_bytes, q_bytes;
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

template <typename T, int RB>
__global__ __launch_bounds__(TN_THREADS, 2) void tn_kernel(TnGeom g, const void* __restrict__ p_ptr,
                                                           const void* __restrict__ q_ptr, float* __restrict__ out,
                                                           int co_tiles, int ci_tiles, int taps) {
    typedef TnTile<T, RB> Tile;
    typedef typename Mma<T>::Frag Frag;
    constexpr int NT = Tile::NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = lane_id(), wave = wave_id();
    uint32_t lin = xcd_remap(blockIdx.x, gridDim.x);
    const int tap = (int)(lin % (uint32_t)taps); lin /= (uint32_t)taps;
    const int ci_tile = (int)(lin % (uint32_t)ci_tiles), co_tile = (int)(lin / (uint32_t)ci_tiles);
    const int fr = tap / g.S, fs = tap - fr * g.S;
    const int ks_begin = blockIdx.y * g.ksteps_per_split;
    const int ks_end = min(g.ksteps, ks_begin + g.ksteps_per_split);
    const int wco = wave >> 1, wci = wave & 1;

    f32x4_t acc[NT][NT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const __amdgpu_buffer_rsrc_t rp = make_rsrc(p_ptr, g.p_bytes);
    const __amdgpu_buffer_rsrc_t rq = make_rsrc(q_ptr, g.q_bytes);

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
        char* sp = smem + buf * Tile::STAGE_BYTES;
        char* sq = sp + Tile::TILE_BYTES;
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
            glds16(rp, sp + piece * 1024, offp);
            glds16(rq, sq + piece * 1024, offq);
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
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            cur ^= 1;
        }
        compute(cur);
    }

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
        if (co < g.Kc && ci < g.C)
            atomicAdd(out + ((size_t)co * taps + tap) * g.C + ci, *reinterpret_cast<const float*>(mine + row * P + col * 4));
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
    dim3 grid(co_tiles * ci_tiles * taps, splits);
    hipLaunchKernelGGL(kern, grid, dim3(TN_THREADS), Tile::LDS_BYTES, stream, g, p, q, out, co_tiles, ci_tiles, taps);
    return check_launch("igemm_tn");
}

static int tn_run(int dtype, const void* p, const void* q, float* out, int n, int h, int w, int c, int kc, int ldp,
                  int r, int s, int stride, int pad, int splits, hipStream_t stream, const char* who) {
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
    const int taps = r * s;
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
    if (dtype == FRHIP_DT_BF16) return big ? tn_launch<bf16_t, 256>(g, p, q, out, taps, splits, stream)
                                           : tn_launch<bf16_t, 128>(g, p, q, out, taps, splits, stream);
    return big ? tn_launch<float, 256>(g, p, q, out, taps, splits, stream)
               : tn_launch<float, 128>(g, p, q, out, taps, splits, stream);
}

}  // namespace frhip

using namespace frhip;

extern "C" int frhip_conv_wgrad(int dtype, const void* dy, const void* x, float* dw, int n, int h, int w, int c,
                                int k, int r, int s, int stride, int pad, int splits, hipStream_t stream) {
    // dw[k][r][s][c] (fp32, caller-zeroed) += sum over output pixels of dy[m][k] * x[pix(m,r,s)][c]
    return tn_run(dtype, dy, x, dw, n, h, w, c, k, k, r, s, stride, pad, splits, stream, "frhip_conv_wgrad");
}

extern "C" int frhip_gemm_tn(int dtype, const void* p, const void* q, float* out, int m, int kc, int ldp, int c,
                             int splits, hipStream_t stream) {
    // out[kc][c] (fp32, caller-zeroed) += sum_m p[m][0..kc) (pitch ldp) * q[m][0..c)
    return tn_run(dtype, p, q, out, m, 1, 1, c, kc, ldp, 1, 1, 1, 0, splits, stream, "frhip_gemm_tn");
}
