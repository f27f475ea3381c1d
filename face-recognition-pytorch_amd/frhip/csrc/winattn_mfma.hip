// SwinV2 cosine window attention on the gfx950 matrix cores (bf16 operands, fp32 accumulation and softmax).
// Reference: /root/reference/nets/SwinV2.py:139-179.  Under the reference's autocast both GEMMs run on bf16 operands
// (torch autocast casts the `@` inputs -- the fp32 F.normalize outputs and the fp32 softmax output -- to bf16), which
// is exactly what this kernel does; the scores themselves stay fp32 here (the reference rounds them to bf16).
//
// One wave per (window, head); a wave walks the windows of ONE head so the bias tile lives in registers.  Tokens are
// padded to 64, the head dim is 32.  Everything is computed TRANSPOSED so that no score ever leaves the registers:
//   St[j][i] = <k^_j, q^_i>            A = K^ rows, B = Q^ rows (both plain 16-byte row reads), 4x4 tiles of 16x16x32
//   D layout: lane (g, c) holds keys j = 16*tj + 4g + r (r = 0..3) of query i = 16*ti + c  -> the softmax over keys is
//   16 in-lane values + two cross-lane steps (xor 16, xor 32).
//   Ot[e][i] = sum_j Vt[e][j] P[i][j]  B = P straight from those registers: MFMA sums over its K slots in any order,
//   so K slot (g, 0..7) is DEFINED as keys {32s+4g+0..3, 32s+16+4g+0..3}, which is what the lane already holds; the
//   A side (V, stored [token][e]) is fetched with the transposing LDS read at the same permuted rows.
// Backward (recompute): dPt = V dO^T in the same layout, dS = P o (dP - rowsum(P o dP)) in registers, d(bias) accumulated
// in registers across the wave's windows, dq^t = K^t dSt from registers; P and dS go through one [64][64] bf16 LDS tile
// (row-major) to be read back transposed as the B operands of dVt = dOt P and dk^t = Q^t dS.
#include "winattn.h"
#include "frhip.h"

namespace frhip {

constexpr int WM_ROW = 80;                  // bytes per LDS row of a [64 tokens][32] bf16 tile (64 + 16: conflict-free row reads)
constexpr int WM_TILE = 64 * WM_ROW;
constexpr int WM_PROW = 144;                // bytes per row of the [64][64] bf16 P / dS tile
constexpr int WM_PTILE = 64 * WM_PROW;
constexpr float WM_NEG = -1e30f;
typedef __attribute__((ext_vector_type(8))) short i16x8_t;
typedef __attribute__((address_space(3))) i16x4_t* lds_i16x4_p;

// MFMA operand whose K runs along the 32 head dims of token row 16*t + (lane & 15)
__device__ __forceinline__ bf16x8_t wm_row_frag(const char* tile, int t, int lane) {
    return *reinterpret_cast<const bf16x8_t*>(tile + (16 * t + (lane & 15)) * WM_ROW + (lane >> 4) * 16);
}
// MFMA operand whose K runs along the ROWS of a [row][column] bf16 tile (pitch PITCH bytes): columns c0..c0+15 on the
// lanes, K slots of lane group g = rows {32s+4g+0..3, 32s+16+4g+0..3}
template <int PITCH>
__device__ __forceinline__ bf16x8_t wm_tr_frag(const char* tile, int s, int c0, int lane) {
    const int g = lane >> 4, j = lane & 15;
    const char* pa = tile + (32 * s + 4 * g + (j >> 2)) * PITCH + c0 * 2 + (j & 3) * 8;
    const i16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4_p)LDS_ADDR(pa));
    const i16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4_p)LDS_ADDR(pa + 16 * PITCH));
    return __builtin_bit_cast(bf16x8_t, (i16x8_t)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ bf16x8_t wm_pack(const f32x4_t& a, const f32x4_t& b) {
    bf16x8_t v;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = (bf16_t)a[e]; v[4 + e] = (bf16_t)b[e]; }
    return v;
}
// K slots of key-tile pair s for query tile ti: tiles 2s and 2s + 1 of t[.][ti]; a tile beyond the NT the window needs is zero (its keys are padding)
template <int NT>
__device__ __forceinline__ bf16x8_t wm_pack2(const f32x4_t (&t)[NT][NT], int s, int ti) {
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    return wm_pack(t[2 * s][ti], 2 * s + 1 < NT ? t[2 * s + 1 < NT ? 2 * s + 1 : 0][ti] : zero);
}
__device__ __forceinline__ bf16x4_t wm_pack4(const f32x4_t& a) {
    bf16x4_t v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (bf16_t)a[e];
    return v;
}
__device__ __forceinline__ float wm_sum4g(float x) { return lane_sum_bit5(lane_sum_bit4(x)); }      // over the four lane groups
__device__ __forceinline__ float wm_max4g(float x) { return lane_max_bit5(lane_max_bit4(x)); }      // (same lane & 15)

// One token row (32 bf16) per lane: load (wm_load_row), then optionally l2-normalise and park in the LDS tile (wm_park_row; returns
// 1 / max(|x|, eps)).  The two halves are separate so that the NEXT window's rows can be in flight while this one is computed: one wave
// per SIMD (backward) has nothing else to cover the HBM latency with.
struct WmRow { bf16x8_t v[4]; };
__device__ __forceinline__ void wm_load_row(const bf16_t* src, WmRow& r) {
#pragma unroll
    for (int c = 0; c < 4; ++c) r.v[c] = *reinterpret_cast<const bf16x8_t*>(src + 8 * c);
}
__device__ __forceinline__ float wm_park_row(const WmRow& r, char* tile, int lane, bool active, bool normalise) {
    float inv = 1.f;
    if (normalise) {
        float nn = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float x = (float)r.v[c][e]; nn += x * x; }
        inv = 1.f / fmaxf(sqrtf(nn), 1e-12f);
    }
    const float mul = active ? inv : 0.f;                        // padding tokens become zero rows
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        bf16x8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((float)r.v[c][e] * mul);
        *reinterpret_cast<bf16x8_t*>(tile + lane * WM_ROW + 16 * c) = o;
    }
    return inv;
}

// bias tile in the score layout: [tj][ti][r] = bias_h[i = 16ti + c][j = 16tj + 4g + r]; padded keys get -1e30
template <int NT>
__device__ __forceinline__ void wm_load_bias(const float* bias_h, int n, int lane, f32x4_t (&bs)[NT][NT]) {
    const int g = lane >> 4, c = lane & 15;
#pragma unroll
    for (int tj = 0; tj < NT; ++tj)
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * ti + c, j = 16 * tj + 4 * g + r;
                bs[tj][ti][r] = j >= n ? WM_NEG : (i < n ? bias_h[i * n + j] : 0.f);
            }
}

// St = K^ Q^t (cosines) for the staged window
template <int NT>
__device__ __forceinline__ void wm_scores(const char* kh, const char* qh, int lane, f32x4_t (&st)[NT][NT]) {
    bf16x8_t a[NT], b[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { a[t] = wm_row_frag(kh, t, lane); b[t] = wm_row_frag(qh, t, lane); }
#pragma unroll
    for (int tj = 0; tj < NT; ++tj)
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
            st[tj][ti] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            st[tj][ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[tj], b[ti], st[tj][ti], 0, 0, 0);
        }
}

// in place: cosines -> probabilities (softmax over the keys of every query column)
template <int NT>
__device__ __forceinline__ void wm_softmax(f32x4_t (&st)[NT][NT], const f32x4_t (&bs)[NT][NT], float sc, const int* sreg,
                                           bool masked, int lane) {
    const int g = lane >> 4, c = lane & 15;
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
        float mx = -INFINITY;
        const int rq = masked ? sreg[16 * ti + c] : 0;
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = st[tj][ti][r] * sc + bs[tj][ti][r];
                if (masked) v += sreg[16 * tj + 4 * g + r] != rq ? -100.f : 0.f;
                st[tj][ti][r] = v;
                mx = fmaxf(mx, v);
            }
        }
        mx = wm_max4g(mx);
        float sum = 0.f;
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float e = __expf(st[tj][ti][r] - mx); st[tj][ti][r] = e; sum += e; }
        const float inv = 1.f / wm_sum4g(sum);
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) st[tj][ti][r] *= inv;
    }
}

// grid = (heads, chunks), 4 waves; LDS per wave: Q^, K^, V tiles + pixel / region tables
constexpr int WM_FWD_WAVE = 3 * WM_TILE + 512;

template <int NT>
__global__ __launch_bounds__(256, 2) void wm_fwd_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ bias,
                                                        const float* __restrict__ scale, bf16_t* __restrict__ out, int nwin,
                                                        WaGeom g, int C, int win_per_block, int heads) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = wave_id(), lane = lane_id();
    char* qh = smem + wave * WM_FWD_WAVE;
    char* kh = qh + WM_TILE;
    char* vt = kh + WM_TILE;
    int* spix = reinterpret_cast<int*>(vt + WM_TILE);
    int* sreg = spix + 64;
    // (head, window chunk) from ONE XCD-remapped grid dimension with the head fastest: a head's slice of a token row is 64 bytes -- half a
    // cache line -- so the workgroups of adjacent heads of one window chunk should run on the same XCD at the same time (with a (heads,
    // chunks) grid and eight heads, head h ALWAYS landed on XCD h and every XCD's L2 fetched every line of qkv for half of it)
    const uint32_t lin = xcd_remap(blockIdx.x, gridDim.x);
    const int h = (int)(lin % (uint32_t)heads), chunk = (int)(lin / (uint32_t)heads), n = g.n, grp = lane >> 4, c = lane & 15;
    const float sc = scale[h];
    const bool masked = g.shift > 0;
    f32x4_t bs[NT][NT];
    wm_load_bias(bias + (size_t)h * n * n, n, lane, bs);
    const bool active = lane < n;
    const int tok = active ? lane : 0;
    const int w_begin = chunk * win_per_block, w_end = min(nwin, w_begin + win_per_block);
    WmRow nq, nk, nv;                                              // the next window's rows, in flight during this window's compute
    int nregion = 0;
    size_t npix = 0;
    auto fetch = [&](int win) {
        npix = wa_pixel(win, tok, g, &nregion);
        const bf16_t* row = qkv + npix * 3 * C + h * WA_D;
        wm_load_row(row, nq); wm_load_row(row + C, nk); wm_load_row(row + 2 * C, nv);
    };
    if (w_begin + wave < w_end) fetch(w_begin + wave);
    for (int win = w_begin + wave; win < w_end; win += 4) {
        __builtin_amdgcn_wave_barrier();                           // the previous window's LDS reads are done
        wm_park_row(nq, qh, lane, active, true);
        wm_park_row(nk, kh, lane, active, true);
        wm_park_row(nv, vt, lane, active, false);
        spix[lane] = (int)npix;
        sreg[lane] = nregion;
        if (win + 4 < w_end) fetch(win + 4);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        f32x4_t st[NT][NT];
        wm_scores(kh, qh, lane, st);
        wm_softmax(st, bs, sc, sreg, masked, lane);
        // Ot[e][i] = sum_j Vt[e][j] P[i][j]
        f32x4_t ot[2][NT];
#pragma unroll
        for (int te = 0; te < 2; ++te)
#pragma unroll
            for (int ti = 0; ti < NT; ++ti) ot[te][ti] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < (NT + 1) / 2; ++s) {
            bf16x8_t va[2];
#pragma unroll
            for (int te = 0; te < 2; ++te) va[te] = wm_tr_frag<WM_ROW>(vt, s, 16 * te, lane);
#pragma unroll
            for (int ti = 0; ti < NT; ++ti) {
                const bf16x8_t pf = wm_pack2<NT>(st, s, ti);
#pragma unroll
                for (int te = 0; te < 2; ++te) ot[te][ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va[te], pf, ot[te][ti], 0, 0, 0);
            }
        }
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
            const int i = 16 * ti + c;
            if (i < n) {
                bf16_t* dst = out + (size_t)spix[i] * C + h * WA_D + 4 * grp;
#pragma unroll
                for (int te = 0; te < 2; ++te) *reinterpret_cast<bf16x4_t*>(dst + 16 * te) = wm_pack4(ot[te][ti]);
            }
        }
    }
}

// grid = (heads, chunks), 4 waves; LDS per wave: Q^, K^, V, dO tiles, the P / dS tile, 1/|q|, 1/|k|, pixel, region tables
constexpr int WM_BWD_WAVE = 4 * WM_TILE + WM_PTILE + 1024;

// xt[te][t][r] = d(x^)[token 16t + c][e = 16te + 4g + r] -> d(x) through x^ = x / |x|, stored as bf16 (8-byte pieces)
template <int NT>
__device__ __forceinline__ void wm_store_unnormalised(const f32x4_t (&xt)[2][NT], const char* xh, const float* sinv, const int* spix,
                                                      float sc, bf16_t* dst_base, int C, int n, int lane, f32x4_t (&colsum)[2]) {
    const int grp = lane >> 4, c = lane & 15;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int i = 16 * t + c;
        f32x4_t xv[2], dv[2];
        float dot = 0.f;
#pragma unroll
        for (int te = 0; te < 2; ++te) {
            const bf16x4_t x4 = *reinterpret_cast<const bf16x4_t*>(xh + i * WM_ROW + (16 * te + 4 * grp) * 2);
#pragma unroll
            for (int r = 0; r < 4; ++r) { xv[te][r] = (float)x4[r]; dv[te][r] = xt[te][t][r] * sc; dot += dv[te][r] * xv[te][r]; }
        }
        dot = wm_sum4g(dot);
        const float inv = sinv[i];
        if (i < n) {
            bf16_t* dst = dst_base + (size_t)spix[i] * 3 * C + 4 * grp;
#pragma unroll
            for (int te = 0; te < 2; ++te) {
                f32x4_t o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (dv[te][r] - xv[te][r] * dot) * inv;
                const bf16x4_t ob = wm_pack4(o);
                *reinterpret_cast<bf16x4_t*>(dst + 16 * te) = ob;
#pragma unroll
                for (int r = 0; r < 4; ++r) colsum[te][r] += (float)ob[r];           // of the values as stored
            }
        }
    }
}

template <int NT>
__global__ __launch_bounds__(256, 1) void wm_bwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                        const float* __restrict__ bias, const float* __restrict__ scale,
                                                        bf16_t* __restrict__ dqkv, float* __restrict__ dbias,
                                                        float* __restrict__ dscale, WaColsum colsum, int nwin,
                                                        WaGeom g, int C, int win_per_block, int heads) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = wave_id(), lane = lane_id();
    char* qh = smem + wave * WM_BWD_WAVE;
    char* kh = qh + WM_TILE;
    char* vt = kh + WM_TILE;
    char* gt = vt + WM_TILE;                                       // dO
    char* pt = gt + WM_TILE;                                       // P, then dS, row-major [query][key] bf16
    float* siq = reinterpret_cast<float*>(pt + WM_PTILE);
    float* sik = siq + 64;
    int* spix = reinterpret_cast<int*>(sik + 64);
    int* sreg = spix + 64;
    // (head, window chunk) from ONE XCD-remapped grid dimension with the head fastest: a head's slice of a token row is 64 bytes -- half a
    // cache line -- so the workgroups of adjacent heads of one window chunk should run on the same XCD at the same time (with a (heads,
    // chunks) grid and eight heads, head h ALWAYS landed on XCD h and every XCD's L2 fetched every line of qkv for half of it)
    const uint32_t lin = xcd_remap(blockIdx.x, gridDim.x);
    const int h = (int)(lin % (uint32_t)heads), chunk = (int)(lin / (uint32_t)heads), n = g.n, grp = lane >> 4, c = lane & 15;
    const float sc = scale[h];
    const bool masked = g.shift > 0;
    // the head's bias tile in the score layout sits in LDS ([tile][lane] float4, one copy for the four waves) and is read per window:
    // as 64 registers held across the loop it left no room for the next window's rows
    f32x4_t* bs_lds = reinterpret_cast<f32x4_t*>(smem + 4 * WM_BWD_WAVE);
    {
        f32x4_t bs0[NT][NT];
        wm_load_bias(bias + (size_t)h * n * n, n, lane, bs0);
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
                if (((NT * tj + ti) & 3) == wave) bs_lds[(NT * tj + ti) * 64 + lane] = bs0[tj][ti];     // the four waves hold the same tiles: each parks a quarter
    }
    // rows / columns of the P / dS tile beyond the NT x NT tiles are read (as K slots against zero operand rows) and never written: keep them finite
    for (int i = lane; i < WM_PTILE / 16; i += 64) reinterpret_cast<f32x4_t*>(pt)[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    f32x4_t db[NT][NT];
#pragma unroll
    for (int tj = 0; tj < NT; ++tj)
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) db[tj][ti] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float dsc = 0.f;
    f32x4_t csum[3][2];                                            // column sums of the stored dq, dk, dv (bias gradients)
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int te = 0; te < 2; ++te) csum[a][te] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const bool active = lane < n;
    const int tok = active ? lane : 0;
    const int w_begin = chunk * win_per_block, w_end = min(nwin, w_begin + win_per_block);
    WmRow nq, nk, nv, ng;                                          // the next window's rows, in flight during this window's compute
    int nregion = 0;
    size_t npix = 0;
    auto fetch = [&](int win) {
        npix = wa_pixel(win, tok, g, &nregion);
        const bf16_t* row = qkv + npix * 3 * C + h * WA_D;
        wm_load_row(row, nq); wm_load_row(row + C, nk); wm_load_row(row + 2 * C, nv);
        wm_load_row(dout + npix * C + h * WA_D, ng);
    };
    if (w_begin + wave < w_end) fetch(w_begin + wave);
    for (int win = w_begin + wave; win < w_end; win += 4) {
        __builtin_amdgcn_wave_barrier();
        siq[lane] = wm_park_row(nq, qh, lane, active, true);
        sik[lane] = wm_park_row(nk, kh, lane, active, true);
        wm_park_row(nv, vt, lane, active, false);
        wm_park_row(ng, gt, lane, active, false);
        spix[lane] = (int)npix;
        sreg[lane] = nregion;
        if (win + 4 < w_end) fetch(win + 4);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        f32x4_t st[NT][NT];
        wm_scores(kh, qh, lane, st);
        {
            f32x4_t bs[NT][NT];
#pragma unroll
            for (int tj = 0; tj < NT; ++tj)
#pragma unroll
                for (int ti = 0; ti < NT; ++ti) bs[tj][ti] = bs_lds[(NT * tj + ti) * 64 + lane];
            wm_softmax(st, bs, sc, sreg, masked, lane);
        }
        // dPt[j][i] = <v_j, dO_i>
        f32x4_t dp[NT][NT];
        {
            bf16x8_t a[NT], b[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) { a[t] = wm_row_frag(vt, t, lane); b[t] = wm_row_frag(gt, t, lane); }
#pragma unroll
            for (int tj = 0; tj < NT; ++tj)
#pragma unroll
                for (int ti = 0; ti < NT; ++ti) {
                    dp[tj][ti] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    dp[tj][ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[tj], b[ti], dp[tj][ti], 0, 0, 0);
                }
        }
        // P -> LDS (row-major [i][j]) for the dV product; dS = P o (dP - rowsum(P o dP)) in place of dP
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
            float rd = 0.f;
#pragma unroll
            for (int tj = 0; tj < NT; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) rd += st[tj][ti][r] * dp[tj][ti][r];
            rd = wm_sum4g(rd);
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) {
                *reinterpret_cast<bf16x4_t*>(pt + (16 * ti + c) * WM_PROW + (16 * tj + 4 * grp) * 2) = wm_pack4(st[tj][ti]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float ds = st[tj][ti][r] * (dp[tj][ti][r] - rd);
                    dp[tj][ti][r] = ds;
                    db[tj][ti][r] += ds;
                }
            }
        }
        // d(scale) += sum dS o cosines: the cosines are formed again (16 MFMAs) into the registers P has just left, instead of a copy
        // kept across the softmax (64 registers that now hold the next window's rows)
        wm_scores(kh, qh, lane, st);
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int r = 0; r < 4; ++r) dsc += dp[tj][ti][r] * st[tj][ti][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        f32x4_t xt[2][NT];
        // ---- dq^t[e][i] = sum_j K^t[e][j] dSt[j][i]   (B from registers)
#pragma unroll
        for (int te = 0; te < 2; ++te)
#pragma unroll
            for (int t = 0; t < NT; ++t) xt[te][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < (NT + 1) / 2; ++s) {
            bf16x8_t ka[2];
#pragma unroll
            for (int te = 0; te < 2; ++te) ka[te] = wm_tr_frag<WM_ROW>(kh, s, 16 * te, lane);
#pragma unroll
            for (int ti = 0; ti < NT; ++ti) {
                const bf16x8_t df = wm_pack2<NT>(dp, s, ti);
#pragma unroll
                for (int te = 0; te < 2; ++te) xt[te][ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[te], df, xt[te][ti], 0, 0, 0);
            }
        }
        wm_store_unnormalised(xt, qh, siq, spix, sc, dqkv + h * WA_D, C, n, lane, csum[0]);
        // ---- dVt[e][j] = sum_i dOt[e][i] P[i][j]   (both operands transposed reads: K runs along the query rows)
#pragma unroll
        for (int te = 0; te < 2; ++te)
#pragma unroll
            for (int t = 0; t < NT; ++t) xt[te][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < (NT + 1) / 2; ++s) {
            bf16x8_t ga[2];
#pragma unroll
            for (int te = 0; te < 2; ++te) ga[te] = wm_tr_frag<WM_ROW>(gt, s, 16 * te, lane);
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) {
                const bf16x8_t pf = wm_tr_frag<WM_PROW>(pt, s, 16 * tj, lane);
#pragma unroll
                for (int te = 0; te < 2; ++te) xt[te][tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga[te], pf, xt[te][tj], 0, 0, 0);
            }
        }
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
            const int j = 16 * tj + c;
            if (j < n) {
                bf16_t* dst = dqkv + (size_t)spix[j] * 3 * C + 2 * C + h * WA_D + 4 * grp;
#pragma unroll
                for (int te = 0; te < 2; ++te) {
                    const bf16x4_t ob = wm_pack4(xt[te][tj]);
                    *reinterpret_cast<bf16x4_t*>(dst + 16 * te) = ob;
#pragma unroll
                    for (int r = 0; r < 4; ++r) csum[2][te][r] += (float)ob[r];
                }
            }
        }
        // ---- dS -> LDS over P, then dk^t[e][j] = sum_i Q^t[e][i] dS[i][j]
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int tj = 0; tj < NT; ++tj)
                *reinterpret_cast<bf16x4_t*>(pt + (16 * ti + c) * WM_PROW + (16 * tj + 4 * grp) * 2) = wm_pack4(dp[tj][ti]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int te = 0; te < 2; ++te)
#pragma unroll
            for (int t = 0; t < NT; ++t) xt[te][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < (NT + 1) / 2; ++s) {
            bf16x8_t qa[2];
#pragma unroll
            for (int te = 0; te < 2; ++te) qa[te] = wm_tr_frag<WM_ROW>(qh, s, 16 * te, lane);
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) {
                const bf16x8_t df = wm_tr_frag<WM_PROW>(pt, s, 16 * tj, lane);
#pragma unroll
                for (int te = 0; te < 2; ++te) xt[te][tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[te], df, xt[te][tj], 0, 0, 0);
            }
        }
        wm_store_unnormalised(xt, kh, sik, spix, sc, dqkv + C + h * WA_D, C, n, lane, csum[1]);
    }
    // ---- combine the four waves' d(bias) tiles and d(scale): one atomic pass per workgroup
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);                   // [4][64][64]
#pragma unroll
    for (int tj = 0; tj < NT; ++tj)
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
            *reinterpret_cast<f32x4_t*>(red + wave * 4096 + (16 * ti + c) * 64 + 16 * tj + 4 * grp) = db[tj][ti];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) dsc += __shfl_xor(dsc, d);
    __shared__ float red_s[4];
    if (lane == 0) red_s[wave] = dsc;
    float* red_c = red + 4 * 4096;                                 // [4 waves][3][32]
    const bool want_cs = colsum.p[0] || colsum.p[1] || colsum.p[2];
    if (want_cs) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int te = 0; te < 2; ++te)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = csum[a][te][r];
#pragma unroll
                    for (int d = 1; d < 16; d <<= 1) v += __shfl_xor(v, d);       // over the 16 tokens of a lane group
                    if (c == 0) red_c[(wave * 3 + a) * 32 + 16 * te + 4 * grp + r] = v;
                }
    }
    __syncthreads();
    if (want_cs && threadIdx.x < 96) {
        const int a = threadIdx.x >> 5, e = threadIdx.x & 31;
        float* dst = a == 0 ? colsum.p[0] : (a == 1 ? colsum.p[1] : colsum.p[2]);
        if (dst) atomicAdd(dst + h * WA_D + e, red_c[a * 32 + e] + red_c[96 + a * 32 + e] + red_c[192 + a * 32 + e] + red_c[288 + a * 32 + e]);
    }
    for (int idx = threadIdx.x; idx < n * n; idx += 256) {
        const int i = idx / n, j = idx - i * n, o = i * 64 + j;
        atomicAdd(dbias + (size_t)h * n * n + idx, red[o] + red[4096 + o] + red[8192 + o] + red[12288 + o]);
    }
    if (threadIdx.x == 0) atomicAdd(dscale + h, red_s[0] + red_s[1] + red_s[2] + red_s[3]);
}

// Workgroups per launch.  Every workgroup pays a prologue (bias tile) and, in the backward kernel, an epilogue (four d(bias) tiles folded
// through LDS, n*n + 96 atomics) that does not depend on how many windows it walked: with 1024 backward workgroups a wave saw four
// windows and that fixed cost was a third of the launch.  One workgroup per CU (the backward kernel's LDS allows no second one anyway):
// Swin34 15.93 -> 15.44 ms, AlterNet50 13.11 -> 12.68 ms (same-box A/B over 1024 / 512 / 384 / 256 / 192); forward 4 per CU (-0.05 ms).
static int wm_cus() {
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

static int wm_chunks(int nwin, int heads, int target_wgs, int* wpb_out) {
    int chunks = (target_wgs + heads - 1) / heads;
    int wpb = (nwin + chunks - 1) / chunks;
    if (wpb < 4) wpb = 4;
    *wpb_out = wpb;
    return (nwin + wpb - 1) / wpb;
}

// The kernels are instantiated per NT = ceil(tokens / 16) 16-token tiles a window needs (7x7: 4, 6x6: 3, 3x3: 1): scores, softmax and the
// five products walk NT x NT tiles instead of the padded 4 x 4 (a 6x6 window did 16 / 9 of the matrix and softmax work it needed).
template <int NT>
static int wm_fwd_launch(const void* qkv, const float* bias, const float* scale, void* out, int nwin, const WaGeom& g, int C,
                         int heads, hipStream_t stream) {
    const int lds = 4 * WM_FWD_WAVE;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(wm_fwd_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            set_error("frhip_winattn_fwd: cannot raise dynamic LDS to %d bytes", lds);
            return FRHIP_ELAUNCH;
        }
        attr_done = true;
    }
    int wpb;
    static const int env_fwd = getenv("FRHIP_WA_FWD_WGS") ? atoi(getenv("FRHIP_WA_FWD_WGS")) : 0;
    const int target = env_fwd ? env_fwd : 4 * wm_cus();
    const int chunks = wm_chunks(nwin, heads, target, &wpb);
    hipLaunchKernelGGL(wm_fwd_kernel<NT>, dim3(heads * chunks), dim3(256), lds, stream, (const bf16_t*)qkv, bias, scale, (bf16_t*)out,
                       nwin, g, C, wpb, heads);
    return check_launch("frhip_winattn_fwd");
}

int winattn_mfma_fwd(const void* qkv, const float* bias, const float* scale, void* out, int nwin, const WaGeom& g, int C,
                     int heads, hipStream_t stream) {
    switch ((g.n + 15) / 16) {
        case 1: return wm_fwd_launch<1>(qkv, bias, scale, out, nwin, g, C, heads, stream);
        case 2: return wm_fwd_launch<2>(qkv, bias, scale, out, nwin, g, C, heads, stream);
        case 3: return wm_fwd_launch<3>(qkv, bias, scale, out, nwin, g, C, heads, stream);
        default: return wm_fwd_launch<4>(qkv, bias, scale, out, nwin, g, C, heads, stream);
    }
}

template <int NT>
static int wm_bwd_launch(const void* qkv, const void* dout, const float* bias, const float* scale, void* dqkv, float* dbias,
                         float* dscale, const WaColsum& colsum, int nwin, const WaGeom& g, int C, int heads, hipStream_t stream) {
    const int lds = 4 * WM_BWD_WAVE + 16 * 64 * 16;                  // + the head's bias tile in the score layout
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(wm_bwd_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            set_error("frhip_winattn_bwd: cannot raise dynamic LDS to %d bytes", lds);
            return FRHIP_ELAUNCH;
        }
        attr_done = true;
    }
    int wpb;
    static const int env_bwd = getenv("FRHIP_WA_BWD_WGS") ? atoi(getenv("FRHIP_WA_BWD_WGS")) : 0;
    const int target = env_bwd ? env_bwd : wm_cus();
    const int chunks = wm_chunks(nwin, heads, target, &wpb);
    hipLaunchKernelGGL(wm_bwd_kernel<NT>, dim3(heads * chunks), dim3(256), lds, stream, (const bf16_t*)qkv, (const bf16_t*)dout, bias,
                       scale, (bf16_t*)dqkv, dbias, dscale, colsum, nwin, g, C, wpb, heads);
    return check_launch("frhip_winattn_bwd");
}

int winattn_mfma_bwd(const void* qkv, const void* dout, const float* bias, const float* scale, void* dqkv, float* dbias,
                     float* dscale, const WaColsum& colsum, int nwin, const WaGeom& g, int C, int heads, hipStream_t stream) {
    switch ((g.n + 15) / 16) {
        case 1: return wm_bwd_launch<1>(qkv, dout, bias, scale, dqkv, dbias, dscale, colsum, nwin, g, C, heads, stream);
        case 2: return wm_bwd_launch<2>(qkv, dout, bias, scale, dqkv, dbias, dscale, colsum, nwin, g, C, heads, stream);
        case 3: return wm_bwd_launch<3>(qkv, dout, bias, scale, dqkv, dbias, dscale, colsum, nwin, g, C, heads, stream);
        default: return wm_bwd_launch<4>(qkv, dout, bias, scale, dqkv, dbias, dscale, colsum, nwin, g, C, heads, stream);
    }
}

}  // namespace frhip
