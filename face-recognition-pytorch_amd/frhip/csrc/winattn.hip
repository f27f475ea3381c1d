// SwinV2 cosine window attention (7x7 windows, head dim 32) for gfx950, forward and backward.
// Reference: /root/reference/nets/SwinV2.py:139-179 (WindowAttention.forward) with window_partition / window_reverse
// (:35-62) folded into index arithmetic: q/k/v are read straight from the [B*H*W][3C] projection of the NHWC
// activation and the output is written in pixel order, so no partition / reverse copies exist.
//   attn = softmax( normalize(q) normalize(k)^T * scale_h + bias_h ),  out = attn v
// One wave per (window, head); lane i < 49 owns query row i (q-hat, its score row, its output row live in VGPRs),
// K-hat / V sit in LDS and are read as broadcasts.  49x49x32 problems are far too small for a wave-level MFMA tile to
// pay (N = 49 pads to 64, K = 32 is one MFMA step); this first version is a VALU/LDS kernel -- it is <5 % of the
// Swin34 step FLOPs.
// Backward recomputes the probabilities, keeps dS / P rows in registers, transposes through LDS for dK / dV, and
// accumulates d(bias) and d(scale) across all windows of a workgroup's head before one atomic pass.
#include "winattn.h"
#include "frhip.h"

namespace frhip {

constexpr int WA_LD = WA_D + 4;        // row pitch of the [49][32] LDS tiles: 16-byte aligned rows -> b128 broadcast reads
constexpr int WA_LM = WA_N;            // row pitch of the [50][49] LDS tiles: odd, so lane-per-row accesses spread over banks
// floats of LDS per wave (forward: one [50][49] tile, backward: two), kept a multiple of four for the b128 rows
constexpr int wa_per_wave(int mats) { return (2 * WA_N * WA_LD + mats * (WA_N + 1) * WA_LM + 64 + 3) & ~3; }

template <typename T> __device__ __forceinline__ void load32(const T* p, float* v);
template <> __device__ __forceinline__ void load32<float>(const float* p, float* v) {
#pragma unroll
    for (int c = 0; c < 8; ++c) { const f32x4_t t = *reinterpret_cast<const f32x4_t*>(p + 4 * c); v[4*c] = t[0]; v[4*c+1] = t[1]; v[4*c+2] = t[2]; v[4*c+3] = t[3]; }
}
template <> __device__ __forceinline__ void load32<bf16_t>(const bf16_t* p, float* v) {
#pragma unroll
    for (int c = 0; c < 4; ++c) { const bf16x8_t t = *reinterpret_cast<const bf16x8_t*>(p + 8 * c);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[8 * c + e] = (float)t[e]; }
}
template <typename T> __device__ __forceinline__ void store32(T* p, const float* v);
template <> __device__ __forceinline__ void store32<float>(float* p, const float* v) {
#pragma unroll
    for (int c = 0; c < 8; ++c) *reinterpret_cast<f32x4_t*>(p + 4 * c) = f32x4_t{v[4*c], v[4*c+1], v[4*c+2], v[4*c+3]};
}
template <> __device__ __forceinline__ void store32<bf16_t>(bf16_t* p, const float* v) {
#pragma unroll
    for (int c = 0; c < 4; ++c) { bf16x8_t t;
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = (bf16_t)v[8 * c + e];
        *reinterpret_cast<bf16x8_t*>(p + 8 * c) = t; }
}

// <a, row> and acc += s * row against one 32-float LDS row that every lane reads (broadcast, eight ds_read_b128)
__device__ __forceinline__ float dot32(const float* a, const float* row) {
    float d = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const f32x4_t t = *reinterpret_cast<const f32x4_t*>(row + 4 * c);
        d += a[4 * c] * t[0]; d += a[4 * c + 1] * t[1]; d += a[4 * c + 2] * t[2]; d += a[4 * c + 3] * t[3];
    }
    return d;
}
__device__ __forceinline__ void axpy32(float* acc, float s, const float* row) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const f32x4_t t = *reinterpret_cast<const f32x4_t*>(row + 4 * c);
        acc[4 * c] += s * t[0]; acc[4 * c + 1] += s * t[1]; acc[4 * c + 2] += s * t[2]; acc[4 * c + 3] += s * t[3];
    }
}
__device__ __forceinline__ void put32(float* row, const float* v, float s) {
#pragma unroll
    for (int c = 0; c < 8; ++c)
        *reinterpret_cast<f32x4_t*>(row + 4 * c) = f32x4_t{v[4 * c] * s, v[4 * c + 1] * s, v[4 * c + 2] * s, v[4 * c + 3] * s};
}

// Score row of this lane's query against all n keys, softmax'ed, kept in the lane's own LDS row `srow`
// (loops over keys stay rolled: a 49-element register array per lane makes hipcc unroll 49x32 FMAs and spill).
//   srow[j] <- softmax_j( scale * <qh, kh_j> + bias[j] + (region_j != region_i ? -100 : 0) )
__device__ __forceinline__ void wa_softmax_row(const float* qh, const float* sk, const float* bias_row, float scale,
                                               float* srow, int n, const int* sreg, int my_region) {
    float mx = -INFINITY;
#pragma unroll 1
    for (int j = 0; j < n; ++j) {
        const float d = dot32(qh, sk + j * WA_LD);
        const float sv = d * scale + bias_row[j] + (sreg[j] != my_region ? -100.f : 0.f);
        srow[j] = sv;
        mx = fmaxf(mx, sv);
    }
    float sum = 0.f;
#pragma unroll 1
    for (int j = 0; j < n; ++j) { const float ev = __expf(srow[j] - mx); srow[j] = ev; sum += ev; }
    const float inv = 1.f / sum;
#pragma unroll 1
    for (int j = 0; j < n; ++j) srow[j] *= inv;
}

template <typename T>
__global__ __launch_bounds__(256) void winattn_fwd_kernel(const T* __restrict__ qkv, const float* __restrict__ bias,
                                                          const float* __restrict__ scale, T* __restrict__ out,
                                                          int nwin, WaGeom g, int C, int heads) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int PER_WAVE = wa_per_wave(1);
    float* sk = reinterpret_cast<float*>(smem_raw) + wave * PER_WAVE;
    float* sv = sk + WA_N * WA_LD;
    float* ss = sv + WA_N * WA_LD;                    // [50][50]: score / probability rows (row 49 = idle lanes)
    int* sreg = reinterpret_cast<int*>(ss + (WA_N + 1) * WA_LM);      // mask region of every token
    const int pair = blockIdx.x * 4 + wave;
    if (pair >= nwin * heads) return;                 // whole wave exits together
    const int win = pair / heads, h = pair - win * heads;
    const int n = g.n;
    const bool active = lane < n;
    const int tok = active ? lane : 0;
    int region;
    const size_t pix = wa_pixel(win, tok, g, &region);
    if (active) sreg[tok] = region;
    const T* row = qkv + pix * 3 * C + h * WA_D;
    float q[WA_D], t[WA_D];
    load32<T>(row, q);
    load32<T>(row + C, t);                            // k
    float nq = 0.f, nk = 0.f;
#pragma unroll
    for (int e = 0; e < WA_D; ++e) { nq += q[e] * q[e]; nk += t[e] * t[e]; }
    const float iq = 1.f / fmaxf(sqrtf(nq), 1e-12f), ik = 1.f / fmaxf(sqrtf(nk), 1e-12f);
    if (active) put32(sk + tok * WA_LD, t, ik);
#pragma unroll
    for (int e = 0; e < WA_D; ++e) q[e] *= iq;
    load32<T>(row + 2 * C, t);                        // v
    if (active) put32(sv + tok * WA_LD, t, 1.f);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float* srow = ss + (active ? tok : WA_N) * WA_LM;
    wa_softmax_row(q, sk, bias + ((size_t)h * n + tok) * n, scale[h], srow, n, sreg, region);
    float o[WA_D];
#pragma unroll
    for (int e = 0; e < WA_D; ++e) o[e] = 0.f;
#pragma unroll 1
    for (int j = 0; j < n; ++j) {
        axpy32(o, srow[j], sv + j * WA_LD);
    }
    if (active) store32<T>(out + pix * C + h * WA_D, o);
}

// Backward.  grid = (heads, chunks); each wave walks windows of ONE head, so d(bias) / d(scale) accumulate in
// registers across windows and are combined once per workgroup.
template <typename T>
__global__ __launch_bounds__(256, 1) void winattn_bwd_kernel(const T* __restrict__ qkv, const T* __restrict__ dout,
                                                          const float* __restrict__ bias, const float* __restrict__ scale,
                                                          T* __restrict__ dqkv, float* __restrict__ dbias,
                                                          float* __restrict__ dscale, int nwin, WaGeom g, int C,
                                                          int heads, int win_per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int PER_WAVE = wa_per_wave(2);
    float* sa = reinterpret_cast<float*>(smem_raw) + wave * PER_WAVE;      // k-hat, later q-hat
    float* sb = sa + WA_N * WA_LD;                                          // v, later dO
    float* sp = sb + WA_N * WA_LD;                                          // P   [50][50]
    float* sd = sp + (WA_N + 1) * WA_LM;                                    // dP, then dS [50][50]
    int* sreg = reinterpret_cast<int*>(sd + (WA_N + 1) * WA_LM);
    const int h = blockIdx.x;
    const int n = g.n;
    const bool active = lane < n;
    const int tok = active ? lane : 0;
    const int myrow = (active ? tok : WA_N) * WA_LM;
    const float sc = scale[h];
    const float* bias_row = bias + ((size_t)h * n + tok) * n;
    float db[WA_N];
#pragma unroll
    for (int j = 0; j < WA_N; ++j) db[j] = 0.f;
    float dsc = 0.f;
    const int w_begin = blockIdx.y * win_per_block, w_end = min(nwin, w_begin + win_per_block);
    for (int win = w_begin + wave; win < w_end; win += 4) {
        int region;
        const size_t pix = wa_pixel(win, tok, g, &region);
        const T* row = qkv + pix * 3 * C + h * WA_D;
        float q[WA_D], kh[WA_D], go[WA_D];
        load32<T>(row, q);
        load32<T>(row + C, kh);
        float nq = 0.f, nk = 0.f;
#pragma unroll
        for (int e = 0; e < WA_D; ++e) { nq += q[e] * q[e]; nk += kh[e] * kh[e]; }
        const float iq = 1.f / fmaxf(sqrtf(nq), 1e-12f), ik = 1.f / fmaxf(sqrtf(nk), 1e-12f);
#pragma unroll
        for (int e = 0; e < WA_D; ++e) { q[e] *= iq; kh[e] *= ik; }          // q-hat, k-hat of this lane's token
        load32<T>(row + 2 * C, go);                                           // v (staged through `go`)
        __builtin_amdgcn_wave_barrier();                                      // previous window's LDS reads are done
        if (active) {
            sreg[tok] = region;
            put32(sa + tok * WA_LD, kh, 1.f); put32(sb + tok * WA_LD, go, 1.f);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        wa_softmax_row(q, sa, bias_row, sc, sp + myrow, n, sreg, region);
        load32<T>(dout + pix * C + h * WA_D, go);
        // dP_ij = <dO_i, v_j>; dS = P o (dP - rowsum(P o dP))
        float rd = 0.f;
#pragma unroll 1
        for (int j = 0; j < n; ++j) {
            const float d = dot32(go, sb + j * WA_LD);
            sd[myrow + j] = d; rd += d * sp[myrow + j];
        }
        // dq-hat_i = scale * sum_j dS_ij k-hat_j ; d(scale) += sum_j dS_ij cos_ij
        float acc[WA_D];
#pragma unroll
        for (int e = 0; e < WA_D; ++e) acc[e] = 0.f;
        float dsc_w = 0.f;
#pragma unroll 1
        for (int j = 0; j < n; ++j) {
            const float dsj = sp[myrow + j] * (sd[myrow + j] - rd);
            sd[myrow + j] = dsj;
            float c = 0.f;
#pragma unroll
            for (int e4 = 0; e4 < 8; ++e4) {
                const f32x4_t kv = *reinterpret_cast<const f32x4_t*>(sa + j * WA_LD + 4 * e4);
#pragma unroll
                for (int u = 0; u < 4; ++u) { acc[4 * e4 + u] += dsj * kv[u]; c += q[4 * e4 + u] * kv[u]; }
            }
            dsc_w += dsj * c;
        }
        if (active) {
            dsc += dsc_w;
#pragma unroll
            for (int j = 0; j < WA_N; ++j) if (j < n) db[j] += sd[myrow + j];  // static register indices
        }
        float dotq = 0.f;
#pragma unroll
        for (int e = 0; e < WA_D; ++e) { acc[e] *= sc; dotq += acc[e] * q[e]; }
#pragma unroll
        for (int e = 0; e < WA_D; ++e) acc[e] = (acc[e] - q[e] * dotq) * iq;      // normalise-backward
        T* drow = dqkv + pix * 3 * C + h * WA_D;
        if (active) store32<T>(drow, acc);
        // ---- dV_j = sum_i P_ij dO_i : dO -> LDS (over v); q-hat -> LDS (over k-hat) for the dK pass
        __builtin_amdgcn_wave_barrier();
        if (active) { put32(sb + tok * WA_LD, go, 1.f); put32(sa + tok * WA_LD, q, 1.f); }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int e = 0; e < WA_D; ++e) acc[e] = 0.f;
#pragma unroll 1
        for (int i = 0; i < n; ++i) {
            axpy32(acc, sp[i * WA_LM + tok], sb + i * WA_LD);
        }
        if (active) store32<T>(drow + 2 * C, acc);
        // ---- dk-hat_j = scale * sum_i dS_ij q-hat_i
#pragma unroll
        for (int e = 0; e < WA_D; ++e) acc[e] = 0.f;
#pragma unroll 1
        for (int i = 0; i < n; ++i) {
            axpy32(acc, sd[i * WA_LM + tok], sa + i * WA_LD);
        }
        float dotk = 0.f;
#pragma unroll
        for (int e = 0; e < WA_D; ++e) { acc[e] *= sc; dotk += acc[e] * kh[e]; }
#pragma unroll
        for (int e = 0; e < WA_D; ++e) acc[e] = (acc[e] - kh[e] * dotk) * ik;
        if (active) store32<T>(drow + C, acc);
    }
    // ---- combine the four waves' d(bias) rows and d(scale), one atomic pass per workgroup
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem_raw);          // [4][49][50]
    if (active) {
#pragma unroll
        for (int j = 0; j < WA_N; ++j) red[(wave * WA_N + tok) * WA_LM + j] = db[j];
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) dsc += __shfl_xor(dsc, d);
    __shared__ float red_s[4];
    if (lane == 0) red_s[wave] = dsc;
    __syncthreads();
    for (int idx = threadIdx.x; idx < n * n; idx += 256) {
        const int i = idx / n, j = idx - i * n;
        const float v = red[(0 * WA_N + i) * WA_LM + j] + red[(1 * WA_N + i) * WA_LM + j] + red[(2 * WA_N + i) * WA_LM + j] +
                        red[(3 * WA_N + i) * WA_LM + j];
        atomicAdd(dbias + (size_t)h * n * n + idx, v);
    }
    if (threadIdx.x == 0) atomicAdd(dscale + h, red_s[0] + red_s[1] + red_s[2] + red_s[3]);
}

// y[rows][C] += bias[C]; optionally a = gelu(y) (exact erf form, nn.GELU default).  In place on y.
template <typename T>
__global__ __launch_bounds__(256) void bias_gelu_fwd_kernel(T* __restrict__ y, const float* __restrict__ bias, T* __restrict__ a,
                                                            size_t nvec, int C) {
    constexpr int EPV = 16 / (int)sizeof(T);
    const int vpr = C / EPV;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (size_t)gridDim.x * 256) {
        const int c0 = (int)(v % vpr) * EPV;
        Vec16<T> x = *reinterpret_cast<const Vec16<T>*>(y + v * EPV), g;
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
            const float hh = x.get(e) + bias[c0 + e];
            x.set(e, hh);
            g.set(e, gelu_value<T>(x.get(e)));            // GELU of the STORED (rounded) pre-activation
        }
        *reinterpret_cast<Vec16<T>*>(y + v * EPV) = x;
        if (a) *reinterpret_cast<Vec16<T>*>(a + v * EPV) = g;
    }
}

// dh = da * gelu'(h)
template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const T* __restrict__ da, const T* __restrict__ h, T* __restrict__ dh, size_t nvec) {
    constexpr int EPV = 16 / (int)sizeof(T);
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (size_t)gridDim.x * 256) {
        const Vec16<T> g = *reinterpret_cast<const Vec16<T>*>(da + v * EPV);
        Vec16<T> x = *reinterpret_cast<const Vec16<T>*>(h + v * EPV);
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
            x.set(e, g.get(e) * gelu_slope<T>(x.get(e)));
        }
        *reinterpret_cast<Vec16<T>*>(dh + v * EPV) = x;
    }
}

static int ew_blocks(size_t nvec) { size_t b = (nvec + 255) / 256; if (b > 4096) b = 4096; return (int)(b < 1 ? 1 : b); }

}  // namespace frhip

using namespace frhip;

static bool wa_shape_ok(int dtype, int b, int h, int w, int c, int heads, int ws, int shift, const char* who) {
    if ((dtype != FRHIP_DT_BF16 && dtype != FRHIP_DT_F32) || b <= 0 || ws < 1 || ws > 7 || (h % ws) || (w % ws) || heads <= 0 ||
        c != heads * WA_D || shift < 0 || shift >= ws) {
        set_error("%s: needs ws <= 7 dividing h and w, 0 <= shift < ws, head dim 32 (c=%d heads=%d h=%d w=%d ws=%d shift=%d)",
                  who, c, heads, h, w, ws, shift);
        return false;
    }
    return true;
}

static int g_wa_mfma = 1;
extern "C" int frhip_set_winattn_mfma(int enabled) {
    const int old = g_wa_mfma;
    if (enabled >= 0) g_wa_mfma = enabled ? 1 : 0;              // negative: query only
    return old;
}

extern "C" int frhip_winattn_fwd(int dtype, const void* qkv, const float* bias, const float* scale, void* out, int b, int h,
                                 int w, int c, int heads, int ws, int shift, hipStream_t stream) {
    if (!wa_shape_ok(dtype, b, h, w, c, heads, ws, shift, "frhip_winattn_fwd")) return FRHIP_EINVAL;
    const int nwin = b * (h / ws) * (w / ws);
    WaGeom g; g.H = h; g.W = w; g.ws = ws; g.shift = shift; g.n = ws * ws;
    if (dtype == FRHIP_DT_BF16 && g_wa_mfma) return winattn_mfma_fwd(qkv, bias, scale, out, nwin, g, c, heads, stream);
    const int blocks = (nwin * heads + 3) / 4, lds = 4 * wa_per_wave(1) * 4;
    static bool fattr[2] = {false, false};
    if (!fattr[dtype]) {
        const void* fn = dtype == FRHIP_DT_BF16 ? reinterpret_cast<const void*>(winattn_fwd_kernel<bf16_t>) : reinterpret_cast<const void*>(winattn_fwd_kernel<float>);
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) { set_error("frhip_winattn_fwd: LDS %d", lds); return FRHIP_ELAUNCH; }
        fattr[dtype] = true;
    }
    if (dtype == FRHIP_DT_BF16)
        hipLaunchKernelGGL(winattn_fwd_kernel<bf16_t>, dim3(blocks), dim3(256), lds, stream, (const bf16_t*)qkv, bias, scale, (bf16_t*)out, nwin, g, c, heads);
    else
        hipLaunchKernelGGL(winattn_fwd_kernel<float>, dim3(blocks), dim3(256), lds, stream, (const float*)qkv, bias, scale, (float*)out, nwin, g, c, heads);
    return check_launch("frhip_winattn_fwd");
}

extern "C" int frhip_winattn_bwd(int dtype, const void* qkv, const void* dout, const float* bias, const float* scale,
                                 void* dqkv, float* dbias, float* dscale, int b, int h, int w, int c, int heads,
                                 int ws, int shift, hipStream_t stream) {
    if (!wa_shape_ok(dtype, b, h, w, c, heads, ws, shift, "frhip_winattn_bwd")) return FRHIP_EINVAL;
    const int nwin = b * (h / ws) * (w / ws);
    WaGeom g; g.H = h; g.W = w; g.ws = ws; g.shift = shift; g.n = ws * ws;
    if (dtype == FRHIP_DT_BF16 && g_wa_mfma) return winattn_mfma_bwd(qkv, dout, bias, scale, dqkv, dbias, dscale, WaColsum{{nullptr, nullptr, nullptr}}, nwin, g, c, heads, stream);
    int chunks = (1024 + heads - 1) / heads;                 // ~1024 workgroups
    int wpb = (nwin + chunks - 1) / chunks; if (wpb < 4) wpb = 4;
    chunks = (nwin + wpb - 1) / wpb;
    const int lds = 4 * wa_per_wave(2) * 4;
    static bool attr_done[2] = {false, false};
    const void* fn = dtype == FRHIP_DT_BF16 ? reinterpret_cast<const void*>(winattn_bwd_kernel<bf16_t>) : reinterpret_cast<const void*>(winattn_bwd_kernel<float>);
    if (!attr_done[dtype]) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) { set_error("frhip_winattn_bwd: LDS %d", lds); return FRHIP_ELAUNCH; }
        attr_done[dtype] = true;
    }
    if (dtype == FRHIP_DT_BF16)
        hipLaunchKernelGGL(winattn_bwd_kernel<bf16_t>, dim3(heads, chunks), dim3(256), lds, stream, (const bf16_t*)qkv, (const bf16_t*)dout, bias, scale, (bf16_t*)dqkv, dbias, dscale, nwin, g, c, heads, wpb);
    else
        hipLaunchKernelGGL(winattn_bwd_kernel<float>, dim3(heads, chunks), dim3(256), lds, stream, (const float*)qkv, (const float*)dout, bias, scale, (float*)dqkv, dbias, dscale, nwin, g, c, heads, wpb);
    return check_launch("frhip_winattn_bwd");
}

extern "C" int frhip_bias_gelu_fwd(int dtype, void* y, const float* bias, void* act_out, int rows, int c, hipStream_t stream) {
    const int epv = dtype == FRHIP_DT_BF16 ? 8 : 4;
    if ((dtype != FRHIP_DT_BF16 && dtype != FRHIP_DT_F32) || (c % epv)) { set_error("frhip_bias_gelu_fwd: bad dtype/channels"); return FRHIP_EINVAL; }
    const size_t nvec = (size_t)rows * c / epv;
    if (dtype == FRHIP_DT_BF16) hipLaunchKernelGGL(bias_gelu_fwd_kernel<bf16_t>, dim3(ew_blocks(nvec)), dim3(256), 0, stream, (bf16_t*)y, bias, (bf16_t*)act_out, nvec, c);
    else hipLaunchKernelGGL(bias_gelu_fwd_kernel<float>, dim3(ew_blocks(nvec)), dim3(256), 0, stream, (float*)y, bias, (float*)act_out, nvec, c);
    return check_launch("frhip_bias_gelu_fwd");
}

extern "C" int frhip_gelu_bwd(int dtype, const void* da, const void* h, void* dh, size_t n, hipStream_t stream) {
    const int epv = dtype == FRHIP_DT_BF16 ? 8 : 4;
    if ((dtype != FRHIP_DT_BF16 && dtype != FRHIP_DT_F32) || (n % epv)) { set_error("frhip_gelu_bwd: bad dtype/size"); return FRHIP_EINVAL; }
    const size_t nvec = n / epv;
    if (dtype == FRHIP_DT_BF16) hipLaunchKernelGGL(gelu_bwd_kernel<bf16_t>, dim3(ew_blocks(nvec)), dim3(256), 0, stream, (const bf16_t*)da, (const bf16_t*)h, (bf16_t*)dh, nvec);
    else hipLaunchKernelGGL(gelu_bwd_kernel<float>, dim3(ew_blocks(nvec)), dim3(256), 0, stream, (const float*)da, (const float*)h, (float*)dh, nvec);
    return check_launch("frhip_gelu_bwd");
}

extern "C" int frhip_winattn_bwd_colsum(int dtype, const void* qkv, const void* dout, const float* bias, const float* scale,
                                        void* dqkv, float* dbias, float* dscale, float* dqkv_colsum, int b, int h, int w,
                                        int c, int heads, int ws, int shift, hipStream_t stream) {
    if (!wa_shape_ok(dtype, b, h, w, c, heads, ws, shift, "frhip_winattn_bwd_colsum")) return FRHIP_EINVAL;
    if (dtype != FRHIP_DT_BF16 || !g_wa_mfma) {
        set_error("frhip_winattn_bwd_colsum: only the bf16 MFMA kernels produce the column sums (frhip_set_winattn_mfma)");
        return FRHIP_EINVAL;
    }
    const int nwin = b * (h / ws) * (w / ws);
    WaGeom g; g.H = h; g.W = w; g.ws = ws; g.shift = shift; g.n = ws * ws;
    const WaColsum cs = {{dqkv_colsum, dqkv_colsum ? dqkv_colsum + c : nullptr, dqkv_colsum ? dqkv_colsum + 2 * c : nullptr}};
    return winattn_mfma_bwd(qkv, dout, bias, scale, dqkv, dbias, dscale, cs, nwin, g, c, heads, stream);
}

extern "C" int frhip_winattn_bwd_qvbias(int dtype, const void* qkv, const void* dout, const float* bias, const float* scale,
                                        void* dqkv, float* dbias, float* dscale, float* dq_bias, float* dv_bias, int b, int h,
                                        int w, int c, int heads, int ws, int shift, hipStream_t stream) {
    if (!wa_shape_ok(dtype, b, h, w, c, heads, ws, shift, "frhip_winattn_bwd_qvbias")) return FRHIP_EINVAL;
    if (dtype != FRHIP_DT_BF16 || !g_wa_mfma) {
        set_error("frhip_winattn_bwd_qvbias: only the bf16 MFMA kernels produce the column sums (frhip_set_winattn_mfma)");
        return FRHIP_EINVAL;
    }
    const int nwin = b * (h / ws) * (w / ws);
    WaGeom g; g.H = h; g.W = w; g.ws = ws; g.shift = shift; g.n = ws * ws;
    const WaColsum cs = {{dq_bias, nullptr, dv_bias}};
    return winattn_mfma_bwd(qkv, dout, bias, scale, dqkv, dbias, dscale, cs, nwin, g, c, heads, stream);
}
