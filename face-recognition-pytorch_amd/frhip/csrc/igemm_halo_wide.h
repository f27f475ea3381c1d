// LDS-halo 3x3 / stride-1 / pad-1 convolution (forward and data-gradient) with a 64 x 128 output tile PER WAVE, gfx950 bf16.
//
// The 4-wave halo kernel of igemm_halo.h gives a wave 64 pixels x 64 channels: per K half it reads 4 + 4 fragments for 16 MFMAs,
// 0.5 ds_read_b128 per MFMA, and it sits at the rate the chip sustains for that mix (taking the DMA, the
// barriers or the epilogue out does not make it faster).  What raises that rate is less LDS traffic per MFMA.  Here a wave owns
// 64 pixels x 128 channels (4 x 8 MFMA tiles, 128 accumulator registers): 4 + 8 fragments for 32 MFMAs = 0.375 reads per MFMA,
// and a workgroup (4 waves, 256 pixels x 128 channels) streams its weights once for twice the channels: 16 KB of weights +
// 1/9 of a 40-KB window per 18.9 M MACs = 20 B/clk/CU of L2 -> LDS DMA at full MFMA rate instead of 24.7.
// LDS: window (<= 320 rows for W <= 28) 40 KB + two weight buffers of 16 KB = 73 KB, so two workgroups share a CU -- or one
// shares it with an 82-KB weight-gradient workgroup of the side stream.
// Loop: iteration = (64-channel chunk, tap) = 64 MFMAs per wave in four steps (K half x channel half); the weights of the NEXT
// iteration are DMA'd into the other buffer at the top of the iteration, one s_waitcnt vmcnt(0) + one barrier per iteration.
// Fragment registers: two sets of four weight fragments (the next step's are read while this step multiplies), two sets of four
// pixel fragments (one per K half).  The window is reloaded between chunks (single buffer; the co-resident workgroup covers it).
#pragma once
#include <type_traits>
#include "igemm_halo.h"

namespace frhip {

struct HaloWideTile {
    static constexpr int WAVES = 4, THREADS = 256, MT = 4, NTW = 8;      // per wave: 4 x 16 pixels, 8 x 16 channels
    static constexpr int BM = 64 * MT, BN = 128, WROWS = 16 * MT;
    static constexpr int MAXW = 28;
    static constexpr int HROWS = ((256 + 2 * MAXW + 2 + 7) / 8) * 8;     // 320
    static constexpr int ZROW = HROWS * NT_ROWB;
    static constexpr int HALO_BYTES = ZROW + NT_ROWB;
    static constexpr int DUMP_OFF = HALO_BYTES;
    static constexpr int W_OFF = DUMP_OFF + 1024;
    static constexpr int WBUF_BYTES = BN * NT_ROWB;                      // 16 KB
    static constexpr int B_PIECES = BN / 8 / WAVES;                      // 4 one-KiB pieces per wave and tap
    static constexpr int stage_pitch = 64 * 2 + 16;
    static constexpr int LOOP_BYTES = W_OFF + 2 * WBUF_BYTES;
    static constexpr int EPI_BYTES = WAVES * 64 * stage_pitch;
    static constexpr int LDS = LOOP_BYTES > EPI_BYTES ? LOOP_BYTES : EPI_BYTES;
};

struct HaloWideMainloop {
    typedef HaloWideTile Tile;
    typedef bf16_t T;
    typedef Mma<T>::Frag Frag;
    f32x4_t acc[Tile::NTW][Tile::MT];

    __device__ __forceinline__ void run(const HaloGeom& g, const void* __restrict__ a_ptr, const void* __restrict__ b_ptr,
                                        char* smem, int m0, int ntile) {
        constexpr int MT = Tile::MT, BKE = 64;
        const int lane = lane_id(), wave = wave_id();
        // no zero fill of the 128 accumulator registers: the first K half of (chunk 0, tap 0) multiplies into a literal-zero C operand
        const __amdgpu_buffer_rsrc_t ra = make_rsrc(a_ptr, g.a_bytes);
        const __amdgpu_buffer_rsrc_t rb = make_rsrc(b_ptr, g.b_bytes);
        const int hrows = Tile::BM + 2 * g.W + 2;
        const int npieces = (hrows + 7) >> 3;
        const int p_lo = m0 - g.W - 1;
        const int sub = lane >> 3;
        const uint32_t chunk_bytes = (uint32_t)(((lane & 7) ^ sub) * 16);
        const int nchunks = g.C / BKE;

        const uint32_t row0_off = (uint32_t)((p_lo + sub) * g.C) * 2u + chunk_bytes;       // this lane's row of piece 0
        auto halo_load = [&](int c0) {
            // rows outside the tensor: a negative pixel wraps beyond 2 GiB, p >= M lies beyond a_bytes -- the buffer range check zero-fills
            for (int piece = wave; piece < npieces; piece += Tile::WAVES)
                glds16(ra, smem + piece * 1024, row0_off + (uint32_t)(piece * 8 * g.C + c0) * 2u);
        };
        uint32_t brow_off[Tile::B_PIECES];
#pragma unroll
        for (int j = 0; j < Tile::B_PIECES; ++j) {
            const int n = ntile * Tile::BN + (wave * Tile::B_PIECES + j) * 8 + sub;
            brow_off[j] = n < g.Nout ? (uint32_t)n * (uint32_t)g.Ktot * 2u + chunk_bytes : OOB_OFFSET;
        }
        // weights of (chunk ch, tap) into buffer `slot`
        auto weights = [&](int ch, int tap, int slot) {
            const uint32_t kb = (uint32_t)(tap * g.C + ch * BKE) * 2u;
            char* wb = smem + Tile::W_OFF + slot * Tile::WBUF_BYTES;
#pragma unroll
            for (int j = 0; j < Tile::B_PIECES; ++j)
                glds16(rb, wb + (wave * Tile::B_PIECES + j) * 1024, brow_off[j] != OOB_OFFSET ? brow_off[j] + kb : OOB_OFFSET);
        };

        // fragment read offsets, computed once (see igemm_halo.h): per tap and pixel sub-tile the byte offset inside the window
        // of the 16-byte chunk this lane feeds to the MFMA, or the zero row; two 16-bit offsets per register
        const int fi = lane & 15, fg = lane >> 4;
        static_assert(Tile::HALO_BYTES < 65536, "packed fragment offsets");
        uint32_t xa[9][(MT + 1) / 2];
        {
            // offset(mt, tap) = 128 q_mt + [128 tc_tap + swizzle(tap)], validity = AND of per-row predicates: see igemm_halo.h
            const int HW = g.H * g.W;
            uint32_t tt[9];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int tc = g.W + 1 + g.sign * ((tap / 3 - 1) * g.W + (tap % 3 - 1));
                tt[tap] = (uint32_t)(tc * NT_ROWB) + (uint32_t)((fg ^ ((fi + tc) & 7)) << 4);
            }
            const uint32_t zoff = (uint32_t)(Tile::ZROW + (fg << 4));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int q = wave * Tile::WROWS + mt * 16 + fi;
                const int m = m0 + q;
                int y = 0, x = 0;
                const bool live = m < g.M;
                if (live) { const int rem = m - (int)fdiv((uint32_t)m, g.d_hw) * HW; y = (int)fdiv((uint32_t)rem, g.d_w); x = rem - y * g.W; }
                const bool up = live && y > 0, dn = live && y < g.H - 1, lf = live && x > 0, rt = live && x < g.W - 1;
                const bool vy[3] = {g.sign > 0 ? up : dn, live, g.sign > 0 ? dn : up};
                const bool vx[3] = {g.sign > 0 ? lf : rt, live, g.sign > 0 ? rt : lf};
                const uint32_t aq = (uint32_t)(q * NT_ROWB);
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const bool ok = vy[tap / 3] && vx[tap % 3];
                    const uint32_t off = ok ? aq + tt[tap] : zoff;
                    if (mt & 1) xa[tap][mt >> 1] |= off << 16;
                    else xa[tap][mt >> 1] = off;
                }
            }
        }
        const uint32_t wa0 = (uint32_t)(fi * NT_ROWB + ((fg ^ (fi & 7)) << 4));
        typedef const __attribute__((address_space(3))) char* lds_cp;
        typedef const __attribute__((address_space(3))) Frag* lds_fp;
        if ((uint32_t)(uintptr_t)LDS_ADDR(smem) != 0u) __builtin_trap();      // offsets ARE LDS addresses (no static __shared__)

        Frag xf[2][MT], wf[2][4];
        uint32_t xcur[MT];
        // pixel fragments of K half HH for tap TAP
        auto load_x = [&](auto hh_c, auto tap_c) {
            constexpr int HH = decltype(hh_c)::value, TAP = decltype(tap_c)::value;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if constexpr (HH == 0) {
                    if (mt & 1) asm volatile("v_lshrrev_b32 %0, 16, %1" : "=v"(xcur[mt]) : "v"(xa[TAP][mt >> 1]));
                    else asm volatile("v_and_b32 %0, 0xffff, %1" : "=v"(xcur[mt]) : "v"(xa[TAP][mt >> 1]));
                    xf[0][mt] = *(lds_fp)((lds_cp)(uintptr_t)xcur[mt]);
                } else {
                    xf[1][mt] = *(lds_fp)((lds_cp)(uintptr_t)(xcur[mt] ^ 64u));
                }
            }
        };
        // weight fragments of (K half HH, channel half Q) of the current buffer into register set SET
        uint32_t wcur = wa0 + Tile::W_OFF;                  // this lane's K-half-0 fragment address in the current weight buffer
        auto load_w = [&](auto set_c, auto hh_c, auto q_c) {
            constexpr int SET = decltype(set_c)::value, HH = decltype(hh_c)::value, Q = decltype(q_c)::value;
            const uint32_t base = HH ? (wcur ^ 64u) : wcur;                    // K half 1 = chunk index ^ 4
#pragma unroll
            for (int t = 0; t < 4; ++t) wf[SET][t] = *(lds_fp)((lds_cp)(uintptr_t)base + (Q * 4 + t) * 16 * NT_ROWB);
        };
        auto mfma16 = [&](auto set_c, auto hh_c, auto q_c) {
            constexpr int SET = decltype(set_c)::value, HH = decltype(hh_c)::value, Q = decltype(q_c)::value;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int t = 0; t < 4; ++t) Mma<T>::run(wf[SET][t], xf[HH][mt], acc[Q * 4 + t][mt]);
        };
        // K half 0 of (chunk 0, tap 0): C = 0 as an inline constant of the instruction
        auto mfma16_first = [&](auto set_c, auto q_c) {
            constexpr int SET = decltype(set_c)::value, Q = decltype(q_c)::value;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    f32x4_t z = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    Mma<T>::run(wf[SET][t], xf[0][mt], z);
                    acc[Q * 4 + t][mt] = z;
                }
        };
        typedef std::integral_constant<int, 0> I0;
        typedef std::integral_constant<int, 1> I1;

        // ---- prologue
        if (threadIdx.x < 8) *reinterpret_cast<f32x4_t*>(smem + Tile::ZROW + threadIdx.x * 16) = f32x4_t{0.f, 0.f, 0.f, 0.f};
        halo_load(0);
        weights(0, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        // one tap.  Four steps of 16 MFMAs: (K half 0, channels 0-63) (0, 64-127) (1, 0-63) (1, 64-127); the next step's weight
        // fragments (and the second K half's pixel fragments) are read while the current step multiplies.  The weight buffer
        // alternates per iteration at run time (9 taps per chunk is odd), so there is ONE copy of the loop body.
        int slot = 0;
        auto tap_body = [&](auto tap_c, int ch) {
            constexpr int TAP = decltype(tap_c)::value;
            // the next iteration's weights into the other buffer: last read one iteration ago, a barrier since
            if constexpr (TAP < 8) weights(ch, TAP + 1, slot ^ 1);
            else if (ch + 1 < nchunks) weights(ch + 1, 0, slot ^ 1);
            load_x(I0{}, tap_c);
            load_w(I0{}, I0{}, I0{});
            load_w(I1{}, I0{}, I1{});
            const bool first = TAP == 0 && ch == 0;
            if (first) mfma16_first(I0{}, I0{}); else mfma16(I0{}, I0{}, I0{});
            load_x(I1{}, tap_c);
            load_w(I0{}, I1{}, I0{});
            if (first) mfma16_first(I1{}, I1{}); else mfma16(I1{}, I0{}, I1{});
            load_w(I1{}, I1{}, I1{});
            mfma16(I0{}, I1{}, I0{});
            mfma16(I1{}, I1{}, I1{});
            slot ^= 1;
            wcur = wa0 + Tile::W_OFF + (uint32_t)slot * Tile::WBUF_BYTES;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the next iteration's weights have landed (this wave's pieces)
            __builtin_amdgcn_s_barrier();                              // ... everyone's; and nobody still reads the buffer refilled next
        };
        for (int ch = 0; ch < nchunks; ++ch) {
            tap_body(std::integral_constant<int, 0>{}, ch); tap_body(std::integral_constant<int, 1>{}, ch);
            tap_body(std::integral_constant<int, 2>{}, ch); tap_body(std::integral_constant<int, 3>{}, ch);
            tap_body(std::integral_constant<int, 4>{}, ch); tap_body(std::integral_constant<int, 5>{}, ch);
            tap_body(std::integral_constant<int, 6>{}, ch); tap_body(std::integral_constant<int, 7>{}, ch);
            tap_body(std::integral_constant<int, 8>{}, ch);
            if (ch + 1 < nchunks) {            // reload the window for the next 64 channels (the barrier above ordered all reads of it)
                halo_load((ch + 1) * BKE);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
    }

    // Write channels [half*64, half*64 + 64) of this wave's tile to its LDS staging area (row = pixel, 64 channels), as bf16
    __device__ __forceinline__ char* stage_out(char* smem, int half) {
        constexpr int P = Tile::stage_pitch;
        const int lane = lane_id();
        const int fi = lane & 15, fg = lane >> 4;
        char* mine = smem + wave_id() * Tile::WROWS * P;
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < Tile::MT; ++mt) {
                char* p = mine + (mt * 16 + fi) * P + (nt * 16 + 4 * fg) * 2;
                bf16x4_t v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (bf16_t)(half ? acc[4 + nt][mt][e] : acc[nt][mt][e]);
                *reinterpret_cast<bf16x4_t*>(p) = v;
            }
        __syncthreads();
        return mine;
    }
};

}  // namespace frhip
