// frhip -- shared device helpers for the gfx950 (MI355X / CDNA4) kernels.
// One target only: wave = 64 lanes, MFMA 16x16x32 bf16 / 16x16x4 f32, LDS-DMA via buffer_load ... lds.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(4))) short i16x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

#define FRHIP_OK 0
#define FRHIP_EINVAL (-1)
#define FRHIP_ELAUNCH (-2)
#define FRHIP_DT_BF16 0
#define FRHIP_DT_F32 1

#define LDS_ADDR(p) ((__attribute__((address_space(3))) void*)(p))

namespace frhip {

void set_error(const char* fmt, ...);
int check_launch(const char* what);

// ---- buffer resource (raw, stride 0).  OOB reads return 0 -- used for conv zero padding.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
constexpr uint32_t OOB_OFFSET = 0x80000000u;   // >= any num_records we use (< 2 GiB per tensor)

// 16 bytes per lane, global -> LDS without touching VGPRs.  LDS dest = lds_base + lane*16.
// AUX = cache policy bits of the buffer instruction (gfx950: 1 = sc0, 2 = nt, 16 = sc1)
template <int AUX = 0>
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, void* lds_base, uint32_t voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_ADDR(lds_base), 16, voff, 0, 0, AUX);
}

// The same load issued as inline assembly.  The builtin above is known to the compiler's wait-count pass as an LDS write it cannot
// disambiguate: every later LDS read whose address it cannot prove distinct -- all ds_read_b64_tr_b16 reads -- gets an
// s_waitcnt vmcnt(0) in front, which also waits for loads issued to land SEVERAL steps later and so serialises a multi-stage
// operand pipeline (seen in the disassembly of the nine-tap weight-gradient kernel: one such wait per K step, right behind the
// step's DMA issue).  A kernel that uses this form owns its vmcnt bookkeeping entirely (counted s_waitcnt + barrier before any
// read of the landed data) and must not mix it with the builtin (the compiler does not know that M0 changed).
// lds_addr: wave-uniform LDS byte address of the 1-KiB piece; descriptor: raw buffer, stride 0, out-of-range reads return 0.
__device__ __forceinline__ u32x4_t make_rsrc_words(const void* p, uint32_t bytes) {
    const uint64_t a = (uint64_t)(uintptr_t)p;
    return u32x4_t{(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, bytes, 0x00020000u};
}
#ifndef FRHIP_DMA_ASM
#define FRHIP_DMA_ASM 1      // 0: A/B build -- the same call sites through the builtin (and the compiler's waits)
#endif
__device__ __forceinline__ void glds16_asm(u32x4_t rsrc, uint32_t lds_addr, uint32_t voff) {
#if FRHIP_DMA_ASM
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :: "s"(__builtin_amdgcn_readfirstlane(lds_addr)), "v"(voff), "s"(rsrc) : "memory");
#else
    const void* p = reinterpret_cast<const void*>((uintptr_t)rsrc[0] | ((uintptr_t)rsrc[1] << 32));
    __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, rsrc[2], 0x00020000),
                                             (__attribute__((address_space(3))) void*)(uintptr_t)__builtin_amdgcn_readfirstlane(lds_addr), 16, voff, 0, 0, 0);
#endif
}

// ---- element <-> float
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// 16-byte vector of T viewed as floats
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    f32x4_t v;
    __device__ __forceinline__ float get(int i) const { return v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = x; }
};
template <> struct Vec16<bf16_t> {
    static constexpr int N = 8;
    bf16x8_t v;
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = (bf16_t)x; }
};

// ---- MFMA wrappers: one "K group" = one 16-byte fragment per lane for each operand.
//  bf16: 8 k-values / lane, one v_mfma_f32_16x16x32_bf16.
//  f32 : 4 k-values / lane, four v_mfma_f32_16x16x4_f32 (exact f32, validation mode).
// Operand map (both): lane l supplies A[row l&15][k-slot (l>>4)] and B[k-slot (l>>4)][col l&15];
// D: col = l&15, row = 4*(l>>4) + reg.
template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    typedef bf16x8_t Frag;
    __device__ static __forceinline__ void run(const Frag& a, const Frag& b, f32x4_t& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    typedef f32x4_t Frag;
    __device__ static __forceinline__ void run(const Frag& a, const Frag& b, f32x4_t& c) {
#pragma unroll
        for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], c, 0, 0, 0);
    }
};

// fp8 (OCP e4m3fn) operands on the block-scaled MFMA: one v_mfma_scale_f32_16x16x128_f8f6f4 consumes a whole 128-byte LDS
// row per operand row (32 k-values per lane = two 16-byte reads) at twice the bf16 rate.  Block scales are the unit scale
// (E8M0 127): real scales are per tensor / per output channel and applied to the fp32 accumulators in the epilogue.
// Which k-values a lane holds does not matter as long as both operands are read by the same rule (the MFMA sums over k).
struct fp8_t { uint8_t bits; };
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
template <> struct Mma<fp8_t> {
    typedef i32x8_t Frag;
    __device__ static __forceinline__ void run(const Frag& a, const Frag& b, f32x4_t& c) {
        c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0 /* A: e4m3 */, 0 /* B: e4m3 */, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
};

// fp8 operands on the NON-scaled MFMA (v_mfma_f32_16x16x32_fp8_fp8, bf16 rate): a 16-byte fragment feeds two instructions,
// so a 128-byte row is two K sub-steps of 64 channels like a bf16 row is of 32 -- the software pipeline of the LDS-halo
// kernel (two K halves per tap) applies unchanged, at half the bytes per MAC.
struct fp8n_t { uint8_t bits; };
template <> struct Mma<fp8n_t> {
    typedef __attribute__((ext_vector_type(4))) int Frag;
    __device__ static __forceinline__ void run(const Frag& a, const Frag& b, f32x4_t& c) {
        const long a0 = ((long)(uint32_t)a[1] << 32) | (uint32_t)a[0], a1 = ((long)(uint32_t)a[3] << 32) | (uint32_t)a[2];
        const long b0 = ((long)(uint32_t)b[1] << 32) | (uint32_t)b[0], b1 = ((long)(uint32_t)b[3] << 32) | (uint32_t)b[2];
        c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a0, b0, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a1, b1, c, 0, 0, 0);
    }
};

// MFMA fragments of one 128-byte LDS row (16-byte chunks XOR-swizzled by sw = row & 7): bf16 / f32 rows hold two K sub-steps
// of one 16-byte fragment per lane group fg (chunk fg + 4 s), an fp8 row is ONE sub-step of two chunks (fg and fg + 4).
template <typename T> struct RowFrag {
    static constexpr int KSUB = 2;
    __device__ static __forceinline__ typename Mma<T>::Frag load(const char* row, int fg, int sw, int s) {
        return *reinterpret_cast<const typename Mma<T>::Frag*>(row + (((fg + 4 * s) ^ sw) << 4));
    }
};
template <> struct RowFrag<fp8_t> {
    static constexpr int KSUB = 1;
    __device__ static __forceinline__ i32x8_t load(const char* row, int fg, int sw, int) {
        typedef __attribute__((ext_vector_type(4))) int i32x4_t;
        const i32x4_t lo = *reinterpret_cast<const i32x4_t*>(row + ((fg ^ sw) << 4));
        const i32x4_t hi = *reinterpret_cast<const i32x4_t*>(row + (((fg + 4) ^ sw) << 4));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
};

// GELU (exact erf form, nn.GELU default) pieces for a pre-activation h: cdf = Phi(h), pdf = phi(h).
// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, below fp32 round-off of the products it feeds) sharing ONE
// exponential with the density: erf(h/sqrt2) = 1 - poly(t) * exp(-h^2/2), t = 1/(1 + p*|h|/sqrt2).  About 18 VALU
// instructions, branch-free; libm's erff costs twice that plus a divergent branch, which a GEMM epilogue cannot hide.
__device__ __forceinline__ void gelu_parts(float h, float& cdf, float& pdf) {
    const float ax = fabsf(h) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * ax);
    const float ex = __expf(-0.5f * h * h);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float er = copysignf(1.f - poly * ex, h);
    cdf = 0.5f * (1.f + er);
    pdf = 0.3989422804014327f * ex;
}

// bf16 mode: GELU in the logistic ("tanh") form  gelu(h) ~ h * sigma(2u),  u = h (c1 + c3 h^2), with (c1, c3) the minimax fit to the exact
// erf form over every bf16 input (tools: /tmp-free numpy search, recorded in DESIGN 4.7): |gelu error| <= 3.4e-4, |gelu' error| <= 6.7e-4,
// both under 2^-10 -- the size of the bf16 rounding of the stored activation from |a| = 0.2 up -- for 4 plain + 2 transcendental
// instructions (32 issue cycles per element) against 12 + 2 (64) of the A&S 7.1.26 form, which the fp32 validation mode keeps.
// The Swin MLP epilogues are bound by exactly this arithmetic (a 256 x 256 x 256 tile: 3.4 us of MFMA, 9 us of exact GELU).
// gelu'(h) is the derivative of the SAME approximation (one shared exponential, no second transcendental).
#ifndef FRHIP_GELU_EXACT
#define FRHIP_GELU_EXACT 0            // A/B switch: 1 = the exact erf form in bf16 mode too
#endif
constexpr float GELU_C1 = 0.7999131f, GELU_C3 = 0.03497319f, GELU_L2E2 = 2.f * 1.4426950408889634f;
__device__ __forceinline__ float gelu_fast_sigma(float h, float h2) {
    const float w = h * __builtin_fmaf(h2, GELU_C3 * GELU_L2E2, GELU_C1 * GELU_L2E2);      // 2u log2(e)
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-w));
}
template <typename T> __device__ __forceinline__ float gelu_value(float h) {
    if constexpr (sizeof(T) == 2 && !FRHIP_GELU_EXACT) {
        return h * gelu_fast_sigma(h, h * h);
    } else {
        float cdf, pdf;
        gelu_parts(h, cdf, pdf);
        return h * cdf;
    }
}
template <typename T> __device__ __forceinline__ float gelu_slope(float h) {
    if constexpr (sizeof(T) == 2 && !FRHIP_GELU_EXACT) {
        const float h2 = h * h, r = gelu_fast_sigma(h, h2);
        const float q = __builtin_fmaf(h2, 6.f * GELU_C3, 2.f * GELU_C1);                   // d(2u)/dh
        return __builtin_fmaf(h * __builtin_fmaf(-r, r, r), q, r);                          // r + h r (1 - r) q
    } else {
        float cdf, pdf;
        gelu_parts(h, cdf, pdf);
        return cdf + h * pdf;
    }
}

// Sum over the lanes whose id differs from this one in bit 3 / 4 / 5; every lane gets the result.  DPP row rotate and the
// gfx950 v_permlane{16,32}_swap (rows of 16 / halves of 32 lanes exchanged between two registers) are plain VALU operations;
// __shfl_xor compiles to ds_bpermute_b32, an LDS-unit instruction with LDS latency.
__device__ __forceinline__ float lane_sum_bit3(float x) {
    return x + __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(x), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_sum_bit4(float x) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float lane_sum_bit5(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float lane_max_bit4(float x) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float lane_max_bit5(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// sum over the 16 lanes of a DPP row (lanes with equal lane >> 4): four row rotates
__device__ __forceinline__ float lane_sum_row16(float x) {
    x += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(x), 0x121 /* row_ror:1 */, 0xf, 0xf, false));
    x += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(x), 0x122 /* row_ror:2 */, 0xf, 0xf, false));
    x += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(x), 0x124 /* row_ror:4 */, 0xf, 0xf, false));
    x += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(x), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
    return x;
}

__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// XCD-aware remap of a linear workgroup id (bijective for any nwg): consecutive logical ids land on one XCD
// so neighbouring tiles share that XCD's L2 (blocks b and b+8 share an XCD under round-robin dispatch).
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nwg) {
    const uint32_t q = nwg >> 3, r = nwg & 7u, xcd = bid & 7u, k = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

struct FastDiv {      // n / d for 0 <= n < 2^31, d >= 1
    uint32_t mul, shr, d;
};
inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f; f.d = d;
    if (d == 1) { f.mul = 0; f.shr = 0; return f; }
    uint32_t l = 0; while ((1u << l) < d) ++l;          // ceil(log2 d)
    uint64_t m = ((uint64_t(1) << (31 + l)) + d - 1) / d;   // ceil(2^(31+l)/d) fits in 32 bits for n < 2^31
    f.mul = (uint32_t)m; f.shr = l - 1 + 0; f.shr = 31 + l - 32;
    return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
    return f.d == 1 ? n : (__umulhi(n, f.mul) >> f.shr);
}

}  // namespace frhip
