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
#ifndef HALO_A_AUX
#define HALO_A_AUX 0      // cache policy of the activation-window loads (experiment switch)
#endif
#include <type_traits>
#include "igemm_nt.h"

// Timing-only ablation switches (tools/ablate.py builds variants with -DFRHIP_ABL=bits; results are WRONG with any bit
// set): 1 no per-iteration barrier, 2 no MFMA, 4 no LDS fragment reads, 8 no in-loop DMA, 16 no epilogue.
#ifndef FRHIP_ABL
#define FRHIP_ABL 0
#endif

namespace frhip {

struct HaloGeom {
    int H, W, C;            // activation tensor (same spatial size in and out)
    int M, Nout, Ktot;      // N*H*W, output channels, 9*C
    int sign;               // +1 forward (offset = (r-1, s-1)), -1 data-gradient (offset = (1-r, 1-s))
    uint32_t a_bytes, b_bytes;
    // operand transform (XF kernels only): the gathered tensor is the INPUT of a BatchNorm + ReLU and the convolution wants
    // their output -- a = relu(x * xf_scale[c] + xf_shift[c]) is formed in LDS, per 64-channel chunk, right after the window
    // has landed; the activated tensor never exists in HBM (nets/resnet.py:91-93: bn1 -> relu -> conv2)
    const float* xf_scale;
    const float* xf_shift;
    void* xf_out;           // XF kernels: if not NULL, the activated tensor is ALSO written here (by the workgroups of channel column 0,
                            // each its own BM rows, straight from the registers that hold the transformed pieces): the backward pass
                            // wants it (conv2's weight gradient), and this way no separate BatchNorm-apply pass reads y1 to produce it
    FastDiv d_hw, d_w;      // pixel decode of the per-tile prologue without integer division instructions
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

template <typename T, int WM, int WN, int MT, int HBUFS, bool XF = false>
struct HaloMainloop {
    static_assert(!XF || (HBUFS == 1 && std::is_same<T, bf16_t>::value), "operand transform: single-buffer bf16 tile only");
    typedef HaloTile<T, WM, WN, MT, HBUFS> Tile;
    typedef typename Mma<T>::Frag Frag;
    static constexpr int BM = Tile::BM, BN = Tile::BN, BKE = Tile::BKE;

    f32x4_t acc[4][MT];

    __device__ __forceinline__ void run(const HaloGeom& g, const void* __restrict__ a_ptr,
                                        const void* __restrict__ b_ptr, char* smem, int mtile, int ntile) {
        const int lane = lane_id(), wave = wave_id();
        const int wm = wave / WN, wn = wave % WN;
        // no zero fill of the accumulators: the first K half of the first tap writes them with a literal-zero C operand
        const __amdgpu_buffer_rsrc_t ra = make_rsrc(a_ptr, g.a_bytes);
        const __amdgpu_buffer_rsrc_t rb = make_rsrc(b_ptr, g.b_bytes);
        const int m0 = mtile * BM;
        const int hrows = BM + 2 * g.W + 2;                 // rows actually needed
        const int npieces = (hrows + 7) >> 3;
        const int p_lo = m0 - g.W - 1;                      // linear pixel of halo row 0
        const int sub = lane >> 3;
        const uint32_t chunk_bytes = (uint32_t)(((lane & 7) ^ sub) * 16);
        const int nchunks = g.C / BKE;

        const uint32_t row0_off = (uint32_t)((p_lo + sub) * g.C) * (uint32_t)sizeof(T) + chunk_bytes;    // this lane's row of piece 0
        // one 1-KiB halo piece: rows 8*piece .. 8*piece+7 of chunk c0 into buffer hb
        auto halo_piece = [&](int hb, int piece, int c0) {
            // rows outside the tensor need no test: p < 0 wraps to an offset beyond 2 GiB, p >= M lies beyond a_bytes -- the buffer
            // range check zero-fills both (a_bytes = M * C * sizeof(T) < 2^31)
            const uint32_t off = row0_off + (uint32_t)(piece * 8 * g.C + c0) * (uint32_t)sizeof(T);
            glds16<HALO_A_AUX>(ra, smem + hb * Tile::HALO_BYTES + piece * 1024, off);
        };
        // XF: BatchNorm + ReLU of the pieces THIS wave loaded (after its own vmcnt(0), before the barrier that publishes them).
        // Rows outside the tensor were zero-filled by the DMA and stay zero: the convolution pads the ACTIVATED tensor.
        // Same expression and rounding as bn_apply_kernel, so the result is bit-identical to the materialised tensor.
        auto xform_chunk = [&](int c0) {
            if constexpr (XF) {
                const int lc = (lane & 7) ^ sub;                     // logical 16-byte chunk this lane holds of its row
                float sc[8], sh[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { sc[e] = g.xf_scale[c0 + 8 * lc + e]; sh[e] = g.xf_shift[c0 + 8 * lc + e]; }
                const bool emit = g.xf_out != nullptr && ntile == 0;
                for (int piece = wave; piece < npieces; piece += Tile::WAVES) {
                    const int p = p_lo + piece * 8 + sub;
                    if (p >= 0 && p < g.M) {
                        bf16x8_t* a = reinterpret_cast<bf16x8_t*>(smem + piece * 1024 + lane * 16);
                        bf16x8_t v = *a;
#pragma unroll
                        for (int e = 0; e < 8; ++e) { const float o = (float)v[e] * sc[e] + sh[e]; v[e] = (bf16_t)fmaxf(o, 0.f); }
                        *a = v;
                        if (emit && p >= m0 && p < m0 + BM)          // this tile's own rows (the halo rows belong to the neighbours)
                            *reinterpret_cast<bf16x8_t*>(reinterpret_cast<bf16_t*>(g.xf_out) + (size_t)p * g.C + c0 + 8 * lc) = v;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
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
        // two 16-bit offsets per register (a halo buffer is < 64 KiB): low half = even mt, high half = odd mt
        static_assert(Tile::HALO_BYTES < 65536, "packed fragment offsets");
        uint32_t xa[9][(MT + 1) / 2];
        {
            // row(mt, tap) = q_mt + tc_tap with tc_tap = W + 1 + dy W + dx wave-uniform, and q_mt & 7 == fi & 7 (tile and wave bases are
            // multiples of 16), so the byte offset splits into a per-mt and a per-tap term: 128 q_mt + [128 tc + ((fg ^ ((fi + tc) & 7)) << 4)];
            // the validity of a tap is an AND of four per-row predicates (scalar mask arithmetic).  4 + 9 address terms, then one
            // add and one select per (mt, tap) instead of the whole expression 36 times.
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
                const int q = wm * Tile::WROWS + mt * 16 + fi;
                const int m = m0 + q;
                int y = 0, x = 0;
                const bool live = m < g.M;
                if (live) { const int rem = m - (int)fdiv((uint32_t)m, g.d_hw) * HW; y = (int)fdiv((uint32_t)rem, g.d_w); x = rem - y * g.W; }
                // predicates in the direction of the gather: "dn" = the row y + 1 exists, ...; sign = -1 swaps the roles
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
        const int wa = (wn * 64 + fi) * NT_ROWB + ((fg ^ (fi & 7)) << 4);     // weight rows: chunk fg of row (wn*64 + t*16 + fi)

        // ---- software pipeline.  An iteration (= one tap of one 64-channel chunk) has two K halves of 32.  The
        //      fragments of a half are read from LDS while the MFMAs of the PREVIOUS half run (two register sets), so
        //      LDS latency and bandwidth hide behind the matrix pipe.  The one barrier per iteration sits BETWEEN
        //      the halves: it publishes the weights of iteration t+1 (issued at the top of t-1) just before the
        //      second half of t starts prefetching them, and it orders the last reads of ring slot (t+2)%3 / of the
        //      other halo buffer (first half of t-1 at the latest) before the DMA that overwrites them (top of t+1).
        Frag xf[2][MT], wf[2][4];
        // LDS byte addresses (not pointers: the stage / buffer base then folds into the DS immediate).  Address arithmetic is
        // kept off the VALU: it competes with the MFMAs for issue slots (PMC: 2.25 VALU per MFMA before, MFMA pipe 63 %
        // busy).  Weights: both K-half variants live in registers.  Pixels: the packed offset is unpacked by ONE opaque
        // instruction when the first K half of a tap is read and kept for the second half (address ^ 64).
        typedef const __attribute__((address_space(3))) char* lds_cp;
        typedef const __attribute__((address_space(3))) Frag* lds_fp;
        // The kernel has no static __shared__ data, so its dynamic LDS starts at byte 0 and offsets ARE addresses (adding the
        // symbolic base would cost one VALU add per read); checked once.
        if ((uint32_t)(uintptr_t)LDS_ADDR(smem) != 0u) __builtin_trap();
        const uint32_t wa0 = (uint32_t)wa, wa1 = (uint32_t)(wa ^ 64);
        uint32_t xcur[MT];
        auto load_frags = [&](auto set_c, auto h_c, auto hb_c, auto slot_c, auto tap_c) {
            constexpr int SET = decltype(set_c)::value, HH = decltype(h_c)::value, HB = decltype(hb_c)::value,
                          SLOT = decltype(slot_c)::value, TAP = decltype(tap_c)::value;
            (void)xf; (void)wf; (void)xcur;      // named outside the constexpr branches so that the generic lambda captures them
            // second K half = chunk index ^ 4  <=>  byte offset ^ 64 (the zero row is 128 B, so ^64 stays inside it)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if constexpr (FRHIP_ABL & 4) asm volatile("" : "=v"(wf[SET][t]));
                else wf[SET][t] = *(lds_fp)((lds_cp)(uintptr_t)(HH ? wa1 : wa0) + (Tile::W_OFF + SLOT * Tile::WBUF_BYTES + t * 16 * NT_ROWB));
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if constexpr (FRHIP_ABL & 4) asm volatile("" : "=v"(xf[SET][mt]));
                else {
                    if constexpr (HH == 0) {
                        // one instruction, opaque to the optimiser (it would otherwise hoist 72 unpacked copies out of the loop)
                        if (mt & 1) asm volatile("v_lshrrev_b32 %0, 16, %1" : "=v"(xcur[mt]) : "v"(xa[TAP][mt >> 1]));
                        else asm volatile("v_and_b32 %0, 0xffff, %1" : "=v"(xcur[mt]) : "v"(xa[TAP][mt >> 1]));
                        xf[SET][mt] = *(lds_fp)((lds_cp)(uintptr_t)xcur[mt] + HB * Tile::HALO_BYTES);
                    } else {
                        xf[SET][mt] = *(lds_fp)((lds_cp)(uintptr_t)(xcur[mt] ^ 64u) + HB * Tile::HALO_BYTES);
                    }
                }
            }
        };
        auto mfma_set = [&](auto set_c) {
            constexpr int SET = decltype(set_c)::value;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    if constexpr (FRHIP_ABL & 2) { Frag fa = wf[SET][nt], fb = xf[SET][mt]; asm volatile("" :: "v"(fa), "v"(fb)); }
                    else Mma<T>::run(wf[SET][nt], xf[SET][mt], acc[nt][mt]);
                }
        };
        // first K half of (chunk 0, tap 0): C = 0 (an inline constant of the instruction) instead of 16 * MT zeroed registers
        auto mfma_set_first = [&](auto set_c) {
            constexpr int SET = decltype(set_c)::value;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    f32x4_t z = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    if constexpr (FRHIP_ABL & 2) { Frag fa = wf[SET][nt], fb = xf[SET][mt]; asm volatile("" :: "v"(fa), "v"(fb)); }
                    else Mma<T>::run(wf[SET][nt], xf[SET][mt], z);
                    acc[nt][mt] = z;
                }
        };
        // ask the scheduler to spread the (4 + MT) fragment reads of a half between its 4*MT MFMAs
        auto interleave = [&]() {
            constexpr int NREAD = (4 + MT) * (int)(sizeof(Frag) / 16), NMFMA = 4 * MT * (sizeof(T) == 2 ? 1 : (sizeof(T) == 1 ? 2 : 4));
            constexpr int PER = NMFMA / NREAD > 0 ? NMFMA / NREAD : 1;
#pragma unroll
            for (int i = 0; i < NREAD; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);     // MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // DS read
            }
        };
        typedef std::integral_constant<int, 0> I0;
        typedef std::integral_constant<int, 1> I1;

        // ---- prologue: zero rows, halo of chunk 0, weights of iterations 0 and 1, first fragment set
        if (threadIdx.x < 8 * HBUFS) {
            const int hbz = threadIdx.x >> 3;
            *reinterpret_cast<f32x4_t*>(smem + hbz * Tile::HALO_BYTES + Tile::ZROW + (threadIdx.x & 7) * 16) = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
        for (int piece = wave; piece < npieces; piece += Tile::WAVES) halo_piece(0, piece, 0);
        weights(0);
        weights(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        xform_chunk(0);
        __syncthreads();
        load_frags(I0{}, I0{}, I0{}, I0{}, I0{});

        // one chunk = nine taps; hb (halo buffer) is a compile-time constant per instantiation of this lambda
        auto chunk_body = [&](auto hb_c, int ch) {
            constexpr int HB = decltype(hb_c)::value;
            constexpr int HBN = HBUFS == 2 ? (HB ^ 1) : 0;        // halo buffer of the next chunk
            const int c0 = ch * BKE;
            const int it0 = ch * 9;
            auto tap_body = [&](auto tap_c) {
                constexpr int TAP = decltype(tap_c)::value;
                // DMA: weights two iterations ahead + (taps 0..7) one slice of the next chunk's halo
                if constexpr (!(FRHIP_ABL & 8)) {
                    weights(it0 + TAP + 2);
                    if (HBUFS == 2) {
                        const int piece = TAP * Tile::WAVES + wave;
                        if (TAP < 8 && ch + 1 < nchunks && piece < npieces) halo_piece(HBN, piece, c0 + BKE);
                        else glds16(ra, smem + Tile::DUMP_OFF, OOB_OFFSET);
                    }
                }
                load_frags(I1{}, I1{}, hb_c, std::integral_constant<int, TAP % Tile::WRING>{}, tap_c);   // (9*ch + TAP) % 3 == TAP % 3
                if (TAP == 0 && ch == 0) mfma_set_first(I0{});
                else mfma_set(I0{});
                interleave();
                // everything but this iteration's pieces has landed -> weights of the next iteration are in LDS
                if constexpr (FRHIP_ABL & 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if constexpr (Tile::ITER_PIECES == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else if constexpr (Tile::ITER_PIECES == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else if constexpr (Tile::ITER_PIECES == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else if constexpr (Tile::ITER_PIECES == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                static_assert(Tile::ITER_PIECES <= 5, "add a vmcnt case");
                if constexpr (!(FRHIP_ABL & 1)) __builtin_amdgcn_s_barrier();
                if constexpr (TAP < 8)
                    load_frags(I0{}, I0{}, hb_c, std::integral_constant<int, (TAP + 1) % Tile::WRING>{}, std::integral_constant<int, (TAP + 1) % 9>{});
                else
                    load_frags(I0{}, I0{}, std::integral_constant<int, HBN>{}, I0{}, I0{});
                mfma_set(I1{});
                interleave();
            };
            tap_body(std::integral_constant<int, 0>{}); tap_body(std::integral_constant<int, 1>{});
            tap_body(std::integral_constant<int, 2>{}); tap_body(std::integral_constant<int, 3>{});
            tap_body(std::integral_constant<int, 4>{}); tap_body(std::integral_constant<int, 5>{});
            tap_body(std::integral_constant<int, 6>{}); tap_body(std::integral_constant<int, 7>{});
            tap_body(std::integral_constant<int, 8>{});
            if (HBUFS == 1 && ch + 1 < nchunks) {       // single halo buffer: reload it now (exposed latency)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                for (int piece = wave; piece < npieces; piece += Tile::WAVES) halo_piece(0, piece, c0 + BKE);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                xform_chunk(c0 + BKE);
                __builtin_amdgcn_s_barrier();
                load_frags(I0{}, I0{}, I0{}, I0{}, I0{});
            }
        };
        static_assert(HBUFS == 1 || Tile::HROWS / 8 <= 8 * Tile::WAVES, "eight halo slices must cover the halo");
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
