// 3x3 / stride-1 / pad-1 convolution (forward and data-gradient) with an LDS-resident halo tile, gfx950.
//
// The generic NT kernel re-fetches the activation rows of a tile once per filter tap (9x through L2 -> LDS-DMA),
// which makes it LDS-DMA bound.  Here a workgroup keeps, per 64-channel chunk, the whole input window of its BM
// output pixels in LDS once -- the contiguous pixel range [m0 - W - 1, m0 + BM + W] (stride 1: input and output
// share the linear pixel index) -- and all nine taps read their MFMA fragments from it at row offset
// dy*W + dx.  Pixels whose neighbour falls outside the image (or in another image) are redirected, per lane, to a
// zero row.  Only the weights (BN x 128 B per tap) stream per K step.
//   DMA bytes per K step (BM 256, BN 128): 16 KB weights + 1/9 of a ~40 KB halo  vs  48 KB in the generic kernel.
// Loop: iteration = (chunk, tap); weights in a ring of three buffers fetched TWO iterations ahead, halo
// double-buffered per chunk with the next chunk's halo fetched in nine slices behind the taps of the current one.
// Every wave issues the same number of LDS-DMA pieces per iteration (dummies fill the gaps), so one counted
// `s_waitcnt vmcnt(PIECES)` + one raw s_barrier per iteration retires exactly the previous iteration's loads and
// leaves the current iteration's in flight across the barrier.
#pragma once
#include <type_traits>
#include "igemm_nt.h"

namespace frhip {

struct HaloGeom {
    int H, W, C;            // activation tensor (same spatial size in and out)
    int M, Nout, Ktot;      // N*H*W, output channels, 9*C
    int sign;               // +1 forward (offset = (r-1, s-1)), -1 data-gradient (offset = (1-r, 1-s))
    uint32_t a_bytes, b_bytes;
};

template <typename T, int WM, int WN, int MT, int HBUFS>
struct HaloTile {
    static constexpr int WAVES = WM * WN, THREADS = 64 * WAVES;
    static constexpr int WROWS = MT * 16, BM = WM * WROWS, BN = WN * 64;
    static constexpr int BKE = NT_ROWB / (int)sizeof(T);
    static constexpr int MAXW = 56;                                     // widest map this instantiation accepts
    static constexpr int HROWS = ((BM + 2 * MAXW + 2 + 7) / 8) * 8;     // halo rows (upper bound), multiple of 8
    static constexpr int ZROW = HROWS * NT_ROWB;                       // zero row, relative to a halo buffer
    static constexpr int HALO_BYTES = ZROW + NT_ROWB;                  // halo rows + its own zero row
    static constexpr int WBUF_BYTES = BN * NT_ROWB;
    static constexpr int DUMP_OFF = HBUFS * HALO_BYTES;                // 1 KiB sink for dummy DMA pieces
    static constexpr int W_OFF = DUMP_OFF + 1024;
    static constexpr int WRING = 3;
    static constexpr int B_PIECES = (BN / 8 + WAVES - 1) / WAVES;
    static constexpr int ITER_PIECES = B_PIECES + (HBUFS == 2 ? 1 : 0);   // DMA pieces every wave issues per iteration
    template <typename TS> static constexpr int stage_pitch() { return 64 * (int)sizeof(TS) + 16; }
    template <typename TS> static constexpr int lds_bytes() {
        constexpr int loop = W_OFF + WRING * WBUF_BYTES;
        constexpr int epi = WAVES * WROWS * stage_pitch<TS>();
        return loop > epi ? loop : epi;
    }
};

template <typename T, int WM, int WN, int MT, int HBUFS>
struct HaloMainloop {
    typedef HaloTile<T, WM, WN, MT, HBUFS> Tile;
    typedef typename Mma<T>::Frag Frag;
    static constexpr int BM = Tile::BM, BN = Tile::BN, BKE = Tile::BKE;

    f32x4_t acc[4][MT];

    __device__ __forceinline__ void run(const HaloGeom& g, const void* __restrict__ a_ptr,
                                        const void* __restrict__ b_ptr, char* smem, int mtile, int ntile) {
        const int lane = lane_id(), wave = wave_id();
        const int wm = wave / WN, wn = wave % WN;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

        const __amdgpu_buffer_rsrc_t ra = make_rsrc(a_ptr, g.a_bytes);
        const __amdgpu_buffer_rsrc_t rb = make_rsrc(b_ptr, g.b_bytes);
        const int m0 = mtile * BM;
        const int hrows = BM + 2 * g.W + 2;                 // rows actually needed
        const int npieces = (hrows + 7) >> 3;
        const int p_lo = m0 - g.W - 1;                      // linear pixel of halo row 0
        const int sub = lane >> 3;
        const uint32_t chunk_bytes = (uint32_t)(((lane & 7) ^ sub) * 16);
        const int nchunks = g.C / BKE;

        // one 1-KiB halo piece: rows 8*piece .. 8*piece+7 of chunk c0 into buffer hb
        auto halo_piece = [&](int hb, int piece, int c0) {
            const int r = piece * 8 + sub;
            const int p = p_lo + r;
            const uint32_t off = (p >= 0 && p < g.M) ? (uint32_t)(p * g.C + c0) * (uint32_t)sizeof(T) + chunk_bytes : OOB_OFFSET;
            glds16(ra, smem + hb * Tile::HALO_BYTES + piece * 1024, off);
        };
        uint32_t brow_off[Tile::B_PIECES];
#pragma unroll
        for (int j = 0; j < Tile::B_PIECES; ++j) {
            const int piece = wave * Tile::B_PIECES + j;
            const int n = ntile * BN + piece * 8 + sub;
            brow_off[j] = (piece * 8 < BN && n < g.Nout) ? (uint32_t)n * (uint32_t)g.Ktot * (uint32_t)sizeof(T) + chunk_bytes
                                                         : OOB_OFFSET;
        }
        // weights of global iteration `wit` (= chunk*9 + tap) into ring slot wit % 3; past the end: zeros, never read.
        // Always exactly B_PIECES pieces per wave (a piece beyond BN goes to the dump area).
        auto weights = [&](int wit) {
            const int ch = wit / 9, tap = wit - ch * 9;
            const bool live = ch < nchunks;
            const uint32_t kb = (uint32_t)(tap * g.C + ch * BKE) * (uint32_t)sizeof(T);
            char* wb = smem + Tile::W_OFF + (wit % Tile::WRING) * Tile::WBUF_BYTES;
#pragma unroll
            for (int j = 0; j < Tile::B_PIECES; ++j) {
                const int piece = wave * Tile::B_PIECES + j;
                const bool in_tile = piece * 8 < BN;
                glds16(rb, in_tile ? wb + piece * 1024 : smem + Tile::DUMP_OFF,
                       (live && in_tile && brow_off[j] != OOB_OFFSET) ? brow_off[j] + kb : OOB_OFFSET);
            }
        };

        // ---- fragment read addresses, computed ONCE: for every tap and pixel sub-tile the byte offset (inside a
        //      halo buffer) of the 16-byte chunk this lane feeds to the MFMA, or the buffer's zero row when the
        //      neighbour lies outside the image.  The K loop then only adds compile-time constants to them.
        const int fi = lane & 15, fg = lane >> 4;
        int xa[9][MT];
        {
            const int HW = g.H * g.W;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int q = wm * Tile::WROWS + mt * 16 + fi;
                const int m = m0 + q;
                int y = 0, x = 0;
                const bool live = m < g.M;
                if (live) { const int rem = m % HW; y = rem / g.W; x = rem - y * g.W; }
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int dy = g.sign * (tap / 3 - 1), dx = g.sign * (tap % 3 - 1);
                    const bool ok = live && (unsigned)(y + dy) < (unsigned)g.H && (unsigned)(x + dx) < (unsigned)g.W;
                    const int row = q + g.W + 1 + dy * g.W + dx;
                    xa[tap][mt] = ok ? row * NT_ROWB + ((fg ^ (row & 7)) << 4) : Tile::ZROW + (fg << 4);
                }
            }
        }
        const int wa = (wn * 64 + fi) * NT_ROWB + ((fg ^ (fi & 7)) << 4);     // weight rows: chunk fg of row (wn*64 + t*16 + fi)

        // MFMAs of one (halo buffer, weight ring slot, tap) -- all three are compile-time after unrolling
        auto compute = [&](auto hb_c, auto slot_c, auto tap_c) {
            constexpr int HB = decltype(hb_c)::value, SLOT = decltype(slot_c)::value, TAP = decltype(tap_c)::value;
            const char* hbase = smem + HB * Tile::HALO_BYTES;
            const char* wbase = smem + Tile::W_OFF + SLOT * Tile::WBUF_BYTES;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                Frag xf[MT], wf[4];
                // second K half = chunk index ^ 4  <=>  byte offset ^ 64 (the zero row is 128 B, so ^64 stays inside it)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    wf[t] = *reinterpret_cast<const Frag*>(wbase + ((wa + t * 16 * NT_ROWB) ^ (h << 6)));
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) xf[mt] = *reinterpret_cast<const Frag*>(hbase + (xa[TAP][mt] ^ (h << 6)));
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) Mma<T>::run(wf[nt], xf[mt], acc[nt][mt]);
            }
        };

        // ---- prologue: zero rows, halo of chunk 0, weights of iterations 0 and 1
        if (threadIdx.x < 8 * HBUFS) {
            const int hbz = threadIdx.x >> 3;
            *reinterpret_cast<f32x4_t*>(smem + hbz * Tile::HALO_BYTES + Tile::ZROW + (threadIdx.x & 7) * 16) = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
        for (int piece = wave; piece < npieces; piece += Tile::WAVES) halo_piece(0, piece, 0);
        weights(0);
        weights(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        // one chunk = nine taps; hb (halo buffer) is a compile-time constant per instantiation of this lambda
        auto chunk_body = [&](auto hb_c, int ch) {
            constexpr int HB = decltype(hb_c)::value;
            const int c0 = ch * BKE;
            const int it0 = ch * 9;
            auto tap_body = [&](auto tap_c) {
                constexpr int TAP = decltype(tap_c)::value;
                // DMA two iterations ahead (weights) + one slice of the next chunk's halo.  The ring slot / halo
                // buffer written here was last read one iteration / one chunk ago, i.e. before the previous barrier.
                weights(it0 + TAP + 2);
                if (HBUFS == 2) {
                    const int piece = TAP * Tile::WAVES + wave;
                    if (ch + 1 < nchunks && piece < npieces) halo_piece(HB ^ 1, piece, c0 + BKE);
                    else glds16(ra, smem + Tile::DUMP_OFF, OOB_OFFSET);
                }
                compute(hb_c, std::integral_constant<int, TAP % Tile::WRING>{}, tap_c);     // (9*ch + TAP) % 3 == TAP % 3
                // all but this iteration's pieces have landed -> everything the NEXT iteration reads is in LDS
                if constexpr (Tile::ITER_PIECES == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else if constexpr (Tile::ITER_PIECES == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else if constexpr (Tile::ITER_PIECES == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else if constexpr (Tile::ITER_PIECES == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                static_assert(Tile::ITER_PIECES <= 5, "add a vmcnt case");
                __builtin_amdgcn_s_barrier();
            };
            tap_body(std::integral_constant<int, 0>{}); tap_body(std::integral_constant<int, 1>{});
            tap_body(std::integral_constant<int, 2>{}); tap_body(std::integral_constant<int, 3>{});
            tap_body(std::integral_constant<int, 4>{}); tap_body(std::integral_constant<int, 5>{});
            tap_body(std::integral_constant<int, 6>{}); tap_body(std::integral_constant<int, 7>{});
            tap_body(std::integral_constant<int, 8>{});
            if (ch + 1 < nchunks) {
                if (HBUFS == 1) {              // single halo buffer: reload it now (exposed latency)
                    for (int piece = wave; piece < npieces; piece += Tile::WAVES) halo_piece(0, piece, c0 + BKE);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                } else if (npieces > 9 * Tile::WAVES) {   // halo larger than the nine slices cover: fetch the rest
                    for (int piece = 9 * Tile::WAVES + wave; piece < npieces; piece += Tile::WAVES) halo_piece(HB ^ 1, piece, c0 + BKE);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
            }
        };
        for (int ch = 0; ch < nchunks; ch += HBUFS) {
            chunk_body(std::integral_constant<int, 0>{}, ch);
            if (HBUFS == 2 && ch + 1 < nchunks) chunk_body(std::integral_constant<int, HBUFS - 1>{}, ch + 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // drain the look-ahead pieces before LDS is reused
    }

    template <typename TS>
    __device__ __forceinline__ char* stage_out(char* smem) {
        constexpr int P = Tile::template stage_pitch<TS>();
        const int lane = lane_id();
        const int fi = lane & 15, fg = lane >> 4;
        char* mine = smem + wave_id() * Tile::WROWS * P;
        __syncthreads();
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

}  // namespace frhip
