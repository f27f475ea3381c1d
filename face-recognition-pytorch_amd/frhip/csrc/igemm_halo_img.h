// LDS-halo 3x3 / stride-1 / pad-1 convolution (forward and data-gradient), gfx950 bf16, with a tile HEIGHT CHOSEN PER LAUNCH
// so that a launch fills whole rounds of the resident workgroups.
//
// The 256-row tiles of igemm_halo.h leave 3.06 / 1.53 / 0.77 rounds at the ResNet shapes (B = 512: M = 100 352 rows at 14 x 14 is
// 392 tiles per 64- or 128-channel column), i.e. a last round that is mostly empty.  Here a workgroup owns `tpx` <= 208 consecutive
// pixels x 128 output channels, and the launcher picks tpx = ceil(M / T) for a tile count T that is a multiple of the resident
// workgroups per column: 196 rows (one 14 x 14 image, a quarter of a 28 x 28 one, four 7 x 7 ones) give 2048 / 1024 / 512
// workgroups for the 128- / 256- / 512-channel layers -- 4 / 2 / 1 whole rounds of two per CU, 8 / 4 / 2 beside a weight-gradient
// workgroup in the backward pass.
//
// Four waves as 2 (pixels) x 2 (channels).  The 13 pixel sub-tiles of 16 do not split evenly over two waves, so the split is by
// MFMA tiles: wave (wm, wn) owns sub-tiles 6 wm .. 6 wm + 5 for its four 16-channel columns and HALF of sub-tile 12 (two of the
// four columns): 26 MFMAs per K half for every wave, from 4 weight + 7 pixel fragments (0.42 ds_read_b128 per MFMA; the 64 x 64
// wave tile of igemm_halo.h needs 0.5).  The wave's weight fragments are fetched in the column order (0 1 2 3) ^ 2 wm, so that "the
// first two" are compile-time registers for both wave rows.
// LDS: two weight buffers of 16 KB at 0 and 16 KB (the slot toggles by XOR), the window (<= 272 rows) at 32 KB, its zero row
// behind it: 66 KB, two workgroups per CU -- or one beside an 82-KB weight-gradient workgroup.
// Loop: iteration = (64-channel chunk, tap), two K halves.  Fragments of a half are read while the MFMAs of the previous half
// run (weights: two register sets; pixels: ONE set, each fragment replaced right behind the MFMAs that consumed it).  The one
// barrier per iteration sits between the halves, behind `s_waitcnt vmcnt(0) lgkmcnt(0)`: by then every wave has finished reading
// this iteration's weight buffer (so the weights of iteration t + 2 are DMA'd into it right behind the barrier and have a whole
// iteration to land) and the weights of iteration t + 1 are visible for the prefetch that follows.  At the last tap of a chunk the
// window is free behind that barrier as well: the next chunk's window loads under the second K half.
#pragma once
#include <type_traits>
#include "igemm_halo.h"

namespace frhip {

struct HaloImgTile {
    static constexpr int WAVES = 4, THREADS = 256, BN = 128;
    static constexpr int NSUB = 13, TPX_MAX = NSUB * 16;                 // 208 rows of MFMA tiles, tpx of them live
    static constexpr int XS = 7;                                         // pixel fragments per wave and K half
    static constexpr int MAXW = 28;
    static constexpr int WBUF_BYTES = BN * NT_ROWB;                      // 16 KB
    static constexpr int HALO_OFF = 2 * WBUF_BYTES;                      // 32 KB
    static constexpr int HROWS = ((TPX_MAX + 2 * MAXW + 2 + 7) / 8) * 8; // 272
    static constexpr int ZROW = HROWS * NT_ROWB;                         // zero row, relative to the window
    static constexpr int TAB_OFF = HALO_OFF + ZROW + NT_ROWB;            // fragment-offset table: [wave][tap][pixel-fragment pair][fi] u32
    static constexpr int TAB_WAVE = 9 * 4 * 16 * 4;                      // 2304 B per wave
    static constexpr int DUMP_OFF = TAB_OFF + WAVES * TAB_WAVE;          // 1 KiB sink of the epilogue-operand prefetch
    static constexpr int LOOP_BYTES = DUMP_OFF + 1024;
    static constexpr int stage_pitch = 64 * 2 + 16;
    static constexpr int SROWS = TPX_MAX / 2;                            // rows a wave stores: 104
    static constexpr int EPI_BYTES = 2 * TPX_MAX * stage_pitch;
    static constexpr int LDS = LOOP_BYTES > EPI_BYTES ? LOOP_BYTES : EPI_BYTES;
    static_assert(ZROW + NT_ROWB < 65536, "packed fragment offsets");
    static_assert((WBUF_BYTES & (WBUF_BYTES - 1)) == 0, "slot toggle by XOR");
};

struct HaloImgMainloop {
    typedef HaloImgTile Tile;
    typedef bf16_t T;
    typedef Mma<T>::Frag Frag;
    f32x4_t acc[4][6];        // [column t (channel sub-tile t ^ 2 wm)][pixel sub-tile 6 wm + j]
    f32x4_t acc12[2];         // pixel sub-tile 12, columns t = 0, 1

    // m0: first pixel of the tile, tpx: its live rows
    __device__ __forceinline__ void run(const HaloGeom& g, const void* __restrict__ a_ptr, const void* __restrict__ b_ptr,
                                        char* smem, int m0, int tpx, int ntile) {
        constexpr int XS = Tile::XS, BKE = 64;
        const int lane = lane_id(), wave = wave_id();
        const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        acc12[0] = acc12[1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const __amdgpu_buffer_rsrc_t ra = make_rsrc(a_ptr, g.a_bytes);
        const __amdgpu_buffer_rsrc_t rb = make_rsrc(b_ptr, g.b_bytes);
        const int npieces = (tpx + 2 * g.W + 2 + 7) >> 3;
        const int p_lo = m0 - g.W - 1;
        const int sub = lane >> 3;
        const uint32_t chunk_bytes = (uint32_t)(((lane & 7) ^ sub) * 16);
        const int nchunks = g.C / BKE;
        const int niter = nchunks * 9;

        auto halo_load = [&](int c0) {
            for (int piece = wave; piece < npieces; piece += Tile::WAVES) {
                const int p = p_lo + piece * 8 + sub;
                const uint32_t off = (p >= 0 && p < g.M) ? (uint32_t)(p * g.C + c0) * 2u + chunk_bytes : OOB_OFFSET;
                glds16(ra, smem + Tile::HALO_OFF + piece * 1024, off);
            }
        };
        uint32_t brow_off[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = ntile * Tile::BN + (wave * 4 + j) * 8 + sub;
            brow_off[j] = n < g.Nout ? (uint32_t)n * (uint32_t)g.Ktot * 2u + chunk_bytes : OOB_OFFSET;
        }
        // weights of iteration `it` (= chunk * 9 + tap) into buffer `slot_bytes` (0 or WBUF_BYTES)
        auto weights = [&](int it, uint32_t slot_bytes) {
            const int ch = it / 9, tap = it - ch * 9;
            const uint32_t kb = (uint32_t)(tap * g.C + ch * BKE) * 2u;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                glds16(rb, smem + slot_bytes + (wave * 4 + j) * 1024, brow_off[j] != OOB_OFFSET ? brow_off[j] + kb : OOB_OFFSET);
        };

        // Fragment read offsets (relative to the window) of every (tap, pixel fragment): 63 per lane, too many to keep in registers
        // beside 104 accumulators.  Lanes that differ only in fg read the same row at chunk fg ^ (row & 7), i.e. their offsets
        // differ by XOR (fg << 4): the fg = 0 offsets go to an LDS table once (two 16-bit offsets per word, [tap][pair][fi]) and a
        // tap's four words are fetched one tap ahead.
        const int fi = lane & 15, fg = lane >> 4;
        {
            const int HW = g.H * g.W;
            uint32_t* tab = reinterpret_cast<uint32_t*>(smem + Tile::TAB_OFF + wave * Tile::TAB_WAVE);
            uint32_t pk[9];
#pragma unroll
            for (int j = 0; j < XS; ++j) {
                const int q = (j < 6 ? wm * 6 + j : 12) * 16 + fi;
                const int m = m0 + q;
                int y = 0, x = 0;
                const bool live = q < tpx && m < g.M;
                if (live) { const int rem = m - (int)fdiv((uint32_t)m, g.d_hw) * HW; y = (int)fdiv((uint32_t)rem, g.d_w); x = rem - y * g.W; }
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int dy = g.sign * (tap / 3 - 1), dx = g.sign * (tap % 3 - 1);
                    const bool ok = live && (unsigned)(y + dy) < (unsigned)g.H && (unsigned)(x + dx) < (unsigned)g.W;
                    const int row = q + g.W + 1 + dy * g.W + dx;
                    const uint32_t off = ok ? (uint32_t)(row * NT_ROWB + ((row & 7) << 4)) : (uint32_t)Tile::ZROW;
                    if (j & 1) pk[tap] |= off << 16;
                    else pk[tap] = off;
                    if ((j & 1) || j == XS - 1) { if (fg == 0) tab[(tap * 4 + (j >> 1)) * 16 + fi] = pk[tap]; }
                }
            }
        }
        const uint32_t tab_addr = (uint32_t)(Tile::TAB_OFF + wave * Tile::TAB_WAVE + fi * 4);
        const uint32_t fgx = (uint32_t)((fg << 4) | (fg << 20));
        // weight fragment addresses: row wn * 64 + (t ^ 2 wm) * 16 + fi, chunk fg ^ (fi & 7); [K half][t >> 1], + (t & 1) * 2 KB
        const uint32_t wrow = (uint32_t)((wn * 64 + fi) * NT_ROWB + ((fg ^ (fi & 7)) << 4));
        uint32_t wa[2][2];
        wa[0][0] = wrow + (wm ? 4096u : 0u); wa[0][1] = wrow + (wm ? 0u : 4096u);
        wa[1][0] = wa[0][0] ^ 64u;           wa[1][1] = wa[0][1] ^ 64u;
        typedef const __attribute__((address_space(3))) char* lds_cp;
        typedef const __attribute__((address_space(3))) Frag* lds_fp;
        if ((uint32_t)(uintptr_t)LDS_ADDR(smem) != 0u) __builtin_trap();      // offsets ARE LDS addresses (no static __shared__)

        Frag xf[XS], wf[2][4];
        uint32_t xcur[XS];
        uint32_t tabw[4];                   // the table words of the tap whose K half 0 is read next, XORed with fgx
        typedef const __attribute__((address_space(3))) uint32_t* lds_up;
        auto load_tab = [&](auto tap_c) {
            constexpr int TAP = decltype(tap_c)::value;
#pragma unroll
            for (int p = 0; p < 4; ++p) tabw[p] = *(lds_up)((lds_cp)(uintptr_t)tab_addr + (TAP * 4 + p) * 64);
        };
        auto load_w = [&](auto set_c, int t) {
            constexpr int SET = decltype(set_c)::value;
            wf[SET][t] = *(lds_fp)((lds_cp)(uintptr_t)wa[SET][t >> 1] + (t & 1) * 16 * NT_ROWB);
        };
        // pixel fragment j of K half HH of tap TAP (half 0 unpacks the offset, half 1 reuses it)
        auto load_x = [&](auto hh_c, auto tap_c, int j) {
            constexpr int HH = decltype(hh_c)::value, TAP = decltype(tap_c)::value;
            if constexpr (HH == 0) {
                const uint32_t w = tabw[j >> 1] ^ fgx;
                xcur[j] = (j & 1) ? (w >> 16) : (w & 0xffffu);
                xf[j] = *(lds_fp)((lds_cp)(uintptr_t)xcur[j] + Tile::HALO_OFF);
            } else {
                xf[j] = *(lds_fp)((lds_cp)(uintptr_t)(xcur[j] ^ 64u) + Tile::HALO_OFF);
            }
        };
        auto toggle_slot = [&]() {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int p = 0; p < 2; ++p) wa[h][p] ^= (uint32_t)Tile::WBUF_BYTES;
        };
        typedef std::integral_constant<int, 0> I0;
        typedef std::integral_constant<int, 1> I1;
        // the 26 MFMAs of one K half on weight set SET; behind the MFMAs of pixel fragment j the fragment of the NEXT half is read
        // into the same registers (next_x(j)), and the other weight set is filled along the way (next_w(t), t = 0..3).  mid() runs
        // behind the first four MFMAs (K half 1: the iteration's barrier, see tap_body).
        auto half = [&](auto set_c, auto mid_c, auto&& next_x, auto&& next_w, auto&& mid) {
            constexpr int SET = decltype(set_c)::value;
            constexpr bool MID = decltype(mid_c)::value != 0;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
#pragma unroll
                for (int t = 0; t < 4; ++t) Mma<T>::run(wf[SET][t], xf[j], acc[t][j]);
                if (j == 0 && MID) mid();
                if (j < 4) next_w(j);
                next_x(j);
            }
            Mma<T>::run(wf[SET][0], xf[6], acc12[0]);
            Mma<T>::run(wf[SET][1], xf[6], acc12[1]);
            next_x(6);
            // pin the order written above: 4 MFMAs, then the reads placed behind them (with MID the region starts behind mid())
#ifndef IMG_NOSCHED
            if constexpr (MID) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
            for (int j = MID ? 1 : 0; j < 4; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
#pragma unroll
            for (int j = 4; j < 6; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#endif
        };

        // ---- prologue
        if (threadIdx.x < 8) *reinterpret_cast<f32x4_t*>(smem + Tile::HALO_OFF + Tile::ZROW + threadIdx.x * 16) = f32x4_t{0.f, 0.f, 0.f, 0.f};
        halo_load(0);
        weights(0, 0);
        if (niter > 1) weights(1, Tile::WBUF_BYTES);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        load_tab(I0{});
#pragma unroll
        for (int t = 0; t < 4; ++t) load_w(I0{}, t);
#pragma unroll
        for (int j = 0; j < XS; ++j) load_x(I0{}, I0{}, j);

        uint32_t slot_bytes = 0;            // buffer of the current iteration
        int it = 0;
        auto tap_body = [&](auto tap_c, int ch) {
            constexpr int TAP = decltype(tap_c)::value;
            typedef std::integral_constant<int, (TAP + 1) % 9> NextTap;
            // K half 0; reads of K half 1 (same buffer, same window); the next tap's table words
            load_tab(NextTap{});
            half(I0{}, I0{}, [&](int j) { load_x(I1{}, tap_c, j); }, [&](int t) { load_w(I1{}, t); }, [] {});
            const bool reload = TAP == 8 && ch + 1 < nchunks;
            // The iteration's barrier, four MFMAs into K half 1 (every read issued during K half 0 has had that long to return,
            // so the waits cost nothing).  Behind it every wave is done with this iteration's weight buffer (and, at tap 8, with the
            // window), and the weights of iteration it + 1 are in LDS.
            auto sync = [&]() {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (it + 2 < niter) weights(it + 2, slot_bytes);
                if (TAP == 8 && reload) halo_load((ch + 1) * BKE);
                slot_bytes ^= (uint32_t)Tile::WBUF_BYTES;
                toggle_slot();
                ++it;
            };
            if constexpr (TAP < 8) {
                // K half 1; reads of K half 0 of the next tap (other buffer)
                half(I1{}, I1{}, [&](int j) { load_x(I0{}, NextTap{}, j); }, [&](int t) { load_w(I0{}, t); }, sync);
            } else {
                // the next chunk's window is still loading: nothing to prefetch from it
                half(I1{}, I1{}, [&](int) {}, [&](int t) { if (it < niter) load_w(I0{}, t); }, sync);
                if (reload) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < XS; ++j) load_x(I0{}, I0{}, j);
                }
            }
        };
        for (int ch = 0; ch < nchunks; ++ch) {
            tap_body(std::integral_constant<int, 0>{}, ch); tap_body(std::integral_constant<int, 1>{}, ch);
            tap_body(std::integral_constant<int, 2>{}, ch); tap_body(std::integral_constant<int, 3>{}, ch);
            tap_body(std::integral_constant<int, 4>{}, ch); tap_body(std::integral_constant<int, 5>{}, ch);
            tap_body(std::integral_constant<int, 6>{}, ch); tap_body(std::integral_constant<int, 7>{}, ch);
            tap_body(std::integral_constant<int, 8>{}, ch);
        }
    }

    // Touch the rows of a [M][Nout] bf16 tensor that this tile's store epilogue will read (residual, saved BatchNorm input): one
    // lane per 128-byte line, data into an LDS sink.  With whole rounds the workgroups of a launch reach their epilogues together,
    // and 100 KB of cold operand rows per workgroup then arrive as one HBM burst with the matrix pipes idle; fetched at the start
    // of the tile they come in under the main loop and wait in the last-level cache.
    __device__ static __forceinline__ void touch_rows(const void* __restrict__ t, int M, int Nout, char* smem, int m0, int tpx, int ntile) {
        const __amdgpu_buffer_rsrc_t r = make_rsrc(t, (uint32_t)((size_t)M * Nout * 2));
        const int lines = tpx * 2;                                      // 128 channels x 2 B = two lines per row
        for (int l = (int)threadIdx.x; l < lines; l += Tile::THREADS) {
            const int m = m0 + (l >> 1);
            const uint32_t off = m < M ? (uint32_t)((size_t)m * Nout + ntile * Tile::BN + (l & 1) * 64) * 2u : OOB_OFFSET;
            glds16(r, smem + Tile::DUMP_OFF, off);
        }
    }

    // The wave's tiles as bf16 into the staging region of its channel half (wn): row = pixel of the tile (0..207), 64 channels.
    // Both wave rows write into one region (sub-tile 12 is shared), so the caller needs a workgroup barrier before reading it.
    __device__ __forceinline__ char* stage_out(char* smem) {
        constexpr int P = Tile::stage_pitch;
        const int lane = lane_id(), wave = wave_id();
        const int fi = lane & 15, fg = lane >> 4;
        const int wm = wave >> 1, wn = wave & 1;
        char* region = smem + wn * Tile::TPX_MAX * P;
        __syncthreads();                  // every wave is done with the loop buffers
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int col = ((t ^ (2 * wm)) * 16 + 4 * fg) * 2;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                bf16x4_t v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (bf16_t)acc[t][j][e];
                *reinterpret_cast<bf16x4_t*>(region + ((wm * 6 + j) * 16 + fi) * P + col) = v;
            }
            if (t < 2) {
                bf16x4_t v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (bf16_t)acc12[t][e];
                *reinterpret_cast<bf16x4_t*>(region + (12 * 16 + fi) * P + col) = v;
            }
        }
        __syncthreads();
        return region + wm * Tile::SROWS * P;
    }
};

}  // namespace frhip
