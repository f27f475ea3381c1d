// Recompute-style stem for gfx950: conv3x3(3->64, s1, p1) -> BN -> ReLU -> MaxPool(3,2,1), forward and backward, WITHOUT
// ever writing the 112x112x64 conv output (822 MB in bf16 at B = 512) or its im2col matrix (another 822 MB).
// Reference: /root/reference/nets/resnet.py:186-189 (layers), :232-235 (forward); autograd of the same.
//
// The conv is cheap (27 MACs per output, 22 GFLOP per 512 images) and its input tiny (77 MB), the map it produces
// is what costs: the im2col + GEMM + BN/pool + backward chain streamed that map ~10 times (~3 ms of a 32 ms step).
// Here every pass that needs conv outputs recomputes them from x on the matrix pipe:
//   stats   : y -> per-workgroup { sum y, sum y^2 }                                   (BN batch statistics)
//   forward : y -> relu(y*scale+shift) -> LDS tile -> 3x3/s2 max -> pooled + argmax   (writes 1/4 map + 1 B/elt)
//   reduce  : y, d = pool_grad(dpool, argmax) * relu-mask -> { sum d, sum d*xhat }    (BN backward sums)
//   wgrad   : dy = ca*d + cb*y + cc;  dW[co][k] += sum_pixels dy[co] * col[k]         (weight gradient: dy and the im2col rows go
//             through wave-private LDS tiles and come back as K = pixel MFMA fragments via transposing reads)
// Recomputed values are bit-identical between passes (same instruction sequence), so statistics, ReLU masks and
// arg-max positions agree.
//
// MFMA mapping (Mma<T>, common.h): B operand = im2col fragment gathered straight from global x (L2-resident) in the
// lane layout the MFMA wants -- lane (i = lane & 15, g = lane >> 4) holds k = run*4*EPL + g*EPL + e of pixel i --
// A operand = packed weights [64][32]; D: lane holds channels t*16 + 4g + reg of pixel i.
#include "common.h"
#include "frhip.h"

namespace frhip {

constexpr int SF_PH = 4, SF_PW = 28;                 // pooled tile of the forward / backward kernels
constexpr int SF_RH = 2 * SF_PH + 1, SF_RW = 2 * SF_PW + 1;   // activation region a pooled tile needs (9 x 57)

// The 3-channel input window of a tile is staged in LDS as T ([3][XR][XC], zeros outside the image, coalesced dword loads):
// gathering the im2col fragments straight from global memory (eight predicated scalar loads per lane per 16 pixels) made
// every pass gather-bound (stats alone took 269 us); from LDS the gather is eight ds_read and no bounds checks.
template <typename T>
struct StemConv {
    static constexpr int EPL = 16 / (int)sizeof(T);  // k-values per lane per MFMA group: 8 (bf16) or 4 (f32)
    static constexpr int RUNS = 32 / (4 * EPL);      // MFMA groups that cover the 32 (27 used) k-values: 1 or 2
    typedef typename Mma<T>::Frag Frag;
    Frag wf[RUNS][4];
    int koff[RUNS][EPL];     // element offset of k inside the window relative to the pixel's top-left tap; k >= 27: 0 (any
                             // finite value does: the packed weights are zero there and the gradient columns are dropped)
    const T* xs;
    int XR, XC;

    __device__ __forceinline__ void init(const T* __restrict__ wp, const T* xs_, int XR_, int XC_) {
        xs = xs_; XR = XR_; XC = XC_;
        const int lane = lane_id(), i = lane & 15, g = lane >> 4;
#pragma unroll
        for (int r = 0; r < RUNS; ++r) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                wf[r][t] = *reinterpret_cast<const Frag*>(wp + (t * 16 + i) * 32 + r * 4 * EPL + g * EPL);
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const int k = r * 4 * EPL + g * EPL + e;
                const int tap = k / 3, ci = k - tap * 3, fr = tap / 3, fs = tap - fr * 3;
                koff[r][e] = k < 27 ? (ci * XR + fr) * XC + fs : 0;
            }
        }
    }
    // stage the window whose top-left element is image pixel (h_org, w_org) (may be negative: zero padding)
    __device__ __forceinline__ void stage(T* xs_w, const float* __restrict__ ximg, int h_org, int w_org, int H, int W) const {
        const int total = 3 * XR * XC;
        for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
            const int cc = idx % XC, rest = idx / XC;
            const int rr = rest % XR, ci = rest / XR;
            const int h = h_org + rr, w = w_org + cc;
            const float v = ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) ? ximg[((size_t)ci * H + h) * W + w] : 0.f;
            xs_w[idx] = from_f32<T>(v);
        }
    }
    // im2col fragment of the pixel whose top-left tap sits at window element (hl, wl); dead lanes read the window's first
    // pixel (finite values; their results are discarded by the callers)
    __device__ __forceinline__ void gather(int hl, int wl, bool live, Frag (&b)[RUNS]) const {
        const int base = live ? hl * XC + wl : 0;
#pragma unroll
        for (int r = 0; r < RUNS; ++r) {
            Vec16<T> v;
#pragma unroll
            for (int e = 0; e < EPL; ++e) v.v[e] = xs[base + koff[r][e]];
            b[r] = v.v;
        }
    }
    __device__ __forceinline__ void conv(const Frag (&b)[RUNS], f32x4_t (&acc)[4]) const {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < RUNS; ++r) Mma<T>::run(wf[r][t], b[r], acc[t]);
        }
    }
};

// [pixel][channel] LDS tiles read as K = pixel MFMA fragments (the layout rules of igemm_tn.hip: 16-byte chunks XOR-swizzled
// per row so that the transposing read ds_read_b64_tr_b16 / the strided f32 reads are bank-conflict free)
template <int RB> __device__ __forceinline__ int sf_swz(int row);
template <> __device__ __forceinline__ int sf_swz<256>(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
template <> __device__ __forceinline__ int sf_swz<128>(int row) { return (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1; }

template <typename T, int RB> struct SfFrag;
template <int RB> struct SfFrag<bf16_t, RB> {
    static constexpr int KROWS = 32;     // pixels per MFMA group
    __device__ static __forceinline__ bf16x8_t load(const char* tile, int c0, int lane) {
        const int g = lane >> 4, j = lane & 15, q = j >> 2, p = j & 3;
        const int chunk = (c0 >> 3) + (p >> 1);
        const int row_a = 8 * g + q, row_b = row_a + 4;
        const char* pa = tile + row_a * RB + ((chunk ^ sf_swz<RB>(row_a)) << 4) + 8 * (p & 1);
        const char* pb = tile + row_b * RB + ((chunk ^ sf_swz<RB>(row_b)) << 4) + 8 * (p & 1);
        i16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4_t*)LDS_ADDR(pa));
        i16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4_t*)LDS_ADDR(pb));
        typedef __attribute__((ext_vector_type(8))) short i16x8_t;
        return __builtin_bit_cast(bf16x8_t, (i16x8_t)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
};
template <int RB> struct SfFrag<float, RB> {
    static constexpr int KROWS = 16;
    __device__ static __forceinline__ f32x4_t load(const char* tile, int c0, int lane) {
        const int g = lane >> 4, i = lane & 15;
        const int col = c0 + i, chunk = col >> 2, within = (col & 3) * 4;
        f32x4_t v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = 4 * e + g;
            v[e] = *reinterpret_cast<const float*>(tile + row * RB + ((chunk ^ sf_swz<RB>(row)) << 4) + within);
        }
        return v;
    }
};

// channel of accumulator element (t, reg) of this lane
__device__ __forceinline__ int sf_chan(int t, int reg) { return t * 16 + 4 * (lane_id() >> 4) + reg; }

// sum v over the 16 pixel lanes that share a channel group (lanes with equal lane >> 4)
__device__ __forceinline__ float sf_sum16(float v) { return lane_sum_row16(v); }

// ---------------------------------------------------------------------------------------------------------------------
// stats: grid-stride over (image, 8-row band); wave wv takes rows h0 + 2wv, h0 + 2wv + 1
template <typename T>
__global__ __launch_bounds__(256) void stem_stats_kernel(const float* __restrict__ x, const T* __restrict__ wp,
                                                         float* __restrict__ partial, int B, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* xs = reinterpret_cast<T*>(smem);                       // [3][10][W + 2]
    const int XR = 10, XC = W + 2;
    StemConv<T> sc;
    sc.init(wp, xs, XR, XC);
    const int lane = lane_id(), wave = wave_id(), i = lane & 15;
    const int bands = (H + 7) / 8;
    float s1[4][4], s2[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[t][r] = 0.f; s2[t][r] = 0.f; }
    for (int tile = blockIdx.x; tile < B * bands; tile += gridDim.x) {
        const int n = tile / bands, h0 = (tile - n * bands) * 8;
        __syncthreads();
        sc.stage(xs, x + (size_t)n * 3 * H * W, h0 - 1, -1, H, W);
        __syncthreads();
        for (int rr = 0; rr < 2; ++rr) {
            const int hl = wave * 2 + rr;
            if (h0 + hl >= H) break;
            for (int w0 = 0; w0 < W; w0 += 16) {
                const int w = w0 + i;
                typename StemConv<T>::Frag b[StemConv<T>::RUNS];
                f32x4_t acc[4];
                sc.gather(hl, w, w < W, b);
                sc.conv(b, acc);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const float y = w < W ? acc[t][r] : 0.f; s1[t][r] += y; s2[t][r] += y * y; }
            }
        }
    }
    __syncthreads();
    float (*red)[2][64] = reinterpret_cast<float (*)[2][64]>(smem);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = sf_sum16(s1[t][r]), b2 = sf_sum16(s2[t][r]);
            if (i == 0) { red[wave][0][sf_chan(t, r)] = a; red[wave][1][sf_chan(t, r)] = b2; }
        }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int st = threadIdx.x >> 6, c = threadIdx.x & 63;
        partial[((size_t)blockIdx.x * 2 + st) * 64 + c] = red[0][st][c] + red[1][st][c] + red[2][st][c] + red[3][st][c];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// forward: one workgroup = one pooled tile (SF_PH x SF_PW) of one image
template <typename T>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ x, const T* __restrict__ wp,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       T* __restrict__ pooled, uint8_t* __restrict__ argmax, int B, int H, int W) {
    constexpr int EPV = 16 / (int)sizeof(T), VPR = 64 / EPV;
    constexpr int PITCH = 64 * (int)sizeof(T) + 16;          // bytes per region pixel (+16: the 2-pixel stride of the pool reads)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int XR = SF_RH + 2, XC = SF_RW + 2;              // input window of the activation region
    T* xs = reinterpret_cast<T*>(smem + SF_RH * SF_RW * PITCH);
    StemConv<T> sc;
    sc.init(wp, xs, XR, XC);
    const int lane = lane_id(), wave = wave_id(), i = lane & 15, g = lane >> 4;
    const int Hp = (H - 1) / 2 + 1, Wp = (W - 1) / 2 + 1;
    const int tw = (Wp + SF_PW - 1) / SF_PW, th = (Hp + SF_PH - 1) / SF_PH;
    int tile = blockIdx.x;
    const int n = tile / (tw * th); tile -= n * tw * th;
    const int ph0 = (tile / tw) * SF_PH, pw0 = (tile % tw) * SF_PW;
    sc.stage(xs, x + (size_t)n * 3 * H * W, 2 * ph0 - 2, 2 * pw0 - 2, H, W);
    __syncthreads();
    float sc_[4][4], sh_[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { sc_[t][r] = scale[sf_chan(t, r)]; sh_[t][r] = shift[sf_chan(t, r)]; }
    // ---- activation region -> LDS as T; pixels outside the image hold -inf (MaxPool pads with -inf)
    for (int it = wave; it * 16 < SF_RH * SF_RW; it += 4) {
        const int rp = it * 16 + i;
        const int rr = rp / SF_RW, rc = rp - rr * SF_RW;
        const int h = 2 * ph0 - 1 + rr, w = 2 * pw0 - 1 + rc;
        const bool inside = rp < SF_RH * SF_RW && (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
        typename StemConv<T>::Frag b[StemConv<T>::RUNS];
        f32x4_t acc[4];
        sc.gather(rr, rc, rp < SF_RH * SF_RW, b);            // region pixel (rr, rc): its top-left tap is window element (rr, rc)
        sc.conv(b, acc);
        if (rp < SF_RH * SF_RW) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                T* dst = reinterpret_cast<T*>(smem + rp * PITCH) + t * 16 + 4 * g;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = fmaxf(acc[t][r] * sc_[t][r] + sh_[t][r], 0.f);
                    dst[r] = from_f32<T>(inside ? a : -INFINITY);
                }
            }
        }
    }
    __syncthreads();
    // ---- 3x3 / stride 2 max over the LDS tile: one task = one pooled pixel x one 16-byte channel group
    for (int task = threadIdx.x; task < SF_PH * SF_PW * VPR; task += 256) {
        const int cv = task % VPR, pp = task / VPR;
        const int pl = pp / SF_PW, pc = pp - pl * SF_PW;
        const int ph = ph0 + pl, pw = pw0 + pc;
        if (ph >= Hp || pw >= Wp) continue;
        float best[EPV]; int bi[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) { best[e] = -INFINITY; bi[e] = 0; }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(smem + ((2 * pl + r) * SF_RW + 2 * pc + s) * PITCH + cv * 16);
#pragma unroll
                for (int e = 0; e < EPV; ++e) {
                    const float a = v.get(e);
                    if (a > best[e]) { best[e] = a; bi[e] = r * 3 + s; }
                }
            }
        const size_t o = ((((size_t)n * Hp + ph) * Wp + pw) * VPR + cv) * EPV;
        Vec16<T> ov;
#pragma unroll
        for (int e = 0; e < EPV; ++e) ov.set(e, best[e]);
        *reinterpret_cast<Vec16<T>*>(pooled + o) = ov;
        // the EPV arg-max bytes of this lane as ONE store (o is a multiple of EPV): 64 lanes x 8 B = 512 contiguous bytes
        // instead of eight instructions of stride-8 single bytes
        if constexpr (EPV == 8) {
            uint64_t pk = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) pk |= (uint64_t)(uint32_t)bi[e] << (8 * e);
            *reinterpret_cast<uint64_t*>(argmax + o) = pk;
        } else {
            uint32_t pk = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk |= (uint32_t)bi[e] << (8 * e);
            *reinterpret_cast<uint32_t*>(argmax + o) = pk;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward (both passes): grid-stride over pooled tiles; a tile owns the 2*SF_PH x 2*SF_PW activation pixels
// [2ph0, 2ph0 + 2PH) x [2pw0, 2pw0 + 2PW) and stages the (PH+1) x (PW+1) pooled gradients / arg-max bytes they can
// receive from in LDS.
//   WGRAD = false: partial[block] = { sum d, sum d*(y-mean)*invstd }            (p0 = mean, p1 = invstd)
//   WGRAD = true : dy = p0*d + p1*y + p2;  slab[block][co][k] = sum dy[co]*col[k]
#ifndef SF_BWD_OCC
#define SF_BWD_OCC 2
#endif
template <typename T, bool WGRAD>
__global__ __launch_bounds__(256, SF_BWD_OCC) void stem_bwd_kernel(const float* __restrict__ x, const T* __restrict__ wp,
                                                       const T* __restrict__ dpool, const uint8_t* __restrict__ argmax,
                                                       const float* __restrict__ p0, const float* __restrict__ p1,
                                                       const float* __restrict__ p2, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, float* __restrict__ outbuf,
                                                       int B, int H, int W) {
    constexpr int EPL = StemConv<T>::EPL, RUNS = StemConv<T>::RUNS;
    constexpr int TH = SF_PH + 1, TW = SF_PW + 1;
    constexpr int DROW = 64 * (int)sizeof(T) + 16;           // bytes per pooled pixel in the dpool tile (+16: bank spread)
    constexpr int AROW = 64 + 16;                            // bytes per pooled pixel in the arg-max tile
    constexpr int DY_RB = 64 * (int)sizeof(T), COL_RB = 128; // wgrad staging rows: 64 channels / 32 k-values (bf16: half used)
    constexpr int KR = SfFrag<T, 128>::KROWS;                // pixels per weight-gradient MFMA group: 32 (bf16) / 16 (f32)
    constexpr int SUB = KR / 16;                             // 16-pixel conv iterations per weight-gradient group
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_dp = smem;                                        // [TH*TW][DROW]
    uint8_t* s_arg = reinterpret_cast<uint8_t*>(smem + TH * TW * DROW);     // [TH*TW][AROW]
    char* s_wg = smem + TH * TW * (DROW + AROW);              // per wave: dy tile [KR][DY_RB] + col tile [KR][COL_RB]
    constexpr int XR = 2 * SF_PH + 2, XC = 2 * SF_PW + 2;     // input window of the tile's 2PH x 2PW activation pixels
    T* xs = reinterpret_cast<T*>(s_wg + (WGRAD ? 4 * KR * (DY_RB + COL_RB) : 0) + 5 * 64 * 4);
    StemConv<T> sc;
    sc.init(wp, xs, XR, XC);
    const int lane = lane_id(), wave = wave_id(), i = lane & 15, g = lane >> 4;
    char* my_dy = s_wg + wave * KR * (DY_RB + COL_RB);
    char* my_col = my_dy + KR * DY_RB;
    const int Hp = (H - 1) / 2 + 1, Wp = (W - 1) / 2 + 1;
    const int tw = (Wp + SF_PW - 1) / SF_PW, th = (Hp + SF_PH - 1) / SF_PH;
    // per-channel constants live in LDS (80 registers otherwise): [0] mask scale, [1] mask shift, [2..4] ca, cb, cc (WGRAD)
    float* s_const = reinterpret_cast<float*>(s_wg + (WGRAD ? 4 * KR * (DY_RB + COL_RB) : 0));
    for (int v = threadIdx.x; v < 64; v += 256) {
        s_const[v] = scale[v]; s_const[64 + v] = shift[v];
        if (WGRAD) { s_const[128 + v] = p0[v]; s_const[192 + v] = p1[v]; s_const[256 + v] = p2[v]; }
    }
    auto cvec = [&](int which, int t) { return *reinterpret_cast<const f32x4_t*>(s_const + which * 64 + t * 16 + 4 * g); };
    float s1[4][4], s2[4][4];
    f32x4_t dw[4][2];                                         // D[co = t*16 + 4g + reg][k = kt*16 + i]
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[t][r] = 0.f; s2[t][r] = 0.f; }
        dw[t][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dw[t][1] = dw[t][0];
    }
    if (WGRAD && sizeof(T) == 2) {                            // the unused half of the bf16 col rows must read as zeros
        for (int v = lane; v < KR * COL_RB / 16; v += 64) *reinterpret_cast<f32x4_t*>(my_col + v * 16) = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }

    for (int tile = blockIdx.x; tile < B * th * tw; tile += gridDim.x) {
        int rem = tile;
        const int n = rem / (tw * th); rem -= n * tw * th;
        const int ph0 = (rem / tw) * SF_PH, pw0 = (rem % tw) * SF_PW;
        __syncthreads();                                      // previous tile's readers are done
        sc.stage(xs, x + (size_t)n * 3 * H * W, 2 * ph0 - 1, 2 * pw0 - 1, H, W);
        // ---- stage pooled gradients and arg-max bytes of windows [ph0, ph0+PH] x [pw0, pw0+PW] (zeros outside the map)
        constexpr int VPT = 64 * (int)sizeof(T) / 16;         // 16-byte vectors per pooled pixel (8 bf16 / 16 f32)
        for (int v = threadIdx.x; v < TH * TW * VPT; v += 256) {
            const int pix = v / VPT, part = v - pix * VPT;
            const int ph = ph0 + pix / TW, pw = pw0 + pix % TW;
            f32x4_t val = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (ph < Hp && pw < Wp)
                val = *reinterpret_cast<const f32x4_t*>(reinterpret_cast<const char*>(dpool) +
                                                        ((((size_t)n * Hp + ph) * Wp + pw) * 64) * sizeof(T) + part * 16);
            *reinterpret_cast<f32x4_t*>(s_dp + pix * DROW + part * 16) = val;
        }
        for (int v = threadIdx.x; v < TH * TW * 4; v += 256) {
            const int pix = v >> 2, part = v & 3;
            const int ph = ph0 + pix / TW, pw = pw0 + pix % TW;
            f32x4_t val = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (ph < Hp && pw < Wp)
                val = *reinterpret_cast<const f32x4_t*>(argmax + (((size_t)n * Hp + ph) * Wp + pw) * 64 + part * 16);
            *reinterpret_cast<f32x4_t*>(s_arg + pix * AROW + part * 16) = val;
        }
        __syncthreads();
        // ---- the tile's activation pixels: a wave takes KR consecutive pixels per step (SUB conv iterations of 16)
        for (int grp = wave; grp * KR < 4 * SF_PH * SF_PW; grp += 4) {
#pragma unroll
            for (int sub = 0; sub < SUB; ++sub) {
                const int bp = grp * KR + sub * 16 + i;
                const int br = bp / (2 * SF_PW), bc = bp - br * (2 * SF_PW);
                const int h = 2 * ph0 + br, w = 2 * pw0 + bc;
                const bool live = bp < 4 * SF_PH * SF_PW && h < H && w < W;
                typename StemConv<T>::Frag b[RUNS];
                f32x4_t acc[4];
                sc.gather(br, bc, bp < 4 * SF_PH * SF_PW, b);
                sc.conv(b, acc);
                // pool gradient reaching this pixel: windows (ph, pw), ph in {h>>1, (h+1)>>1}, pw likewise, whose arg-max is here
                float d[4][4];
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) d[t][r] = 0.f;
                const int plo = (h >> 1) - ph0, phi = ((h + 1) >> 1) - ph0, qlo = (w >> 1) - pw0, qhi = ((w + 1) >> 1) - pw0;
#pragma unroll 1
                for (int pp = 0; pp < 2; ++pp) {
                    const int pl = pp == 0 ? plo : phi;
                    const bool rowuse = live && (pp == 0 || phi != plo) && ph0 + pl < Hp;
                    const int rtap = (h - (2 * (ph0 + pl) - 1)) * 3;
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq) {
                        const int pc = qq == 0 ? qlo : qhi;
                        const bool use = rowuse && (qq == 0 || qhi != qlo) && pw0 + pc < Wp;
                        const int tap = rtap + (w - (2 * (pw0 + pc) - 1));
                        const int pix = use ? pl * TW + pc : 0;
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int ch0 = t * 16 + 4 * g;
                            const uint32_t a4 = *reinterpret_cast<const uint32_t*>(s_arg + pix * AROW + ch0);
                            float dv[4];
                            if constexpr (sizeof(T) == 2) {
                                const bf16x4_t q = *reinterpret_cast<const bf16x4_t*>(s_dp + pix * DROW + ch0 * 2);
#pragma unroll
                                for (int r = 0; r < 4; ++r) dv[r] = (float)q[r];
                            } else {
                                const f32x4_t q = *reinterpret_cast<const f32x4_t*>(s_dp + pix * DROW + ch0 * 4);
#pragma unroll
                                for (int r = 0; r < 4; ++r) dv[r] = q[r];
                            }
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (use && (int)((a4 >> (8 * r)) & 0xffu) == tap) d[t][r] += dv[r];
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const f32x4_t ms = cvec(0, t), mb = cvec(1, t);
                    f32x4_t ca, cb, cc;
                    if (WGRAD) { ca = cvec(2, t); cb = cvec(3, t); cc = cvec(4, t); }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float y = acc[t][r];
                        const float de = (live && y * ms[r] + mb[r] > 0.f) ? d[t][r] : 0.f;
                        if (WGRAD) d[t][r] = live ? ca[r] * de + cb[r] * y + cc[r] : 0.f;      // dy
                        else { s1[t][r] += de; s2[t][r] += de * y; }      // sum d*(y-mean)*invstd = invstd*(sum d*y - mean*sum d), below
                    }
                }
                if (WGRAD) {
                    // stage dy [pixel][64 co] and the im2col row [pixel][32 k] of this pixel in the wave's private tiles
                    const int row = sub * 16 + i;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        constexpr int CPV = 16 / (int)sizeof(T);                   // channels per 16-byte chunk
                        const int ch0 = t * 16 + 4 * g;
                        char* dst = my_dy + row * DY_RB + (((ch0 / CPV) ^ sf_swz<DY_RB>(row)) << 4) + (ch0 % CPV) * (int)sizeof(T);
                        if constexpr (sizeof(T) == 2) {
                            bf16x4_t q;
#pragma unroll
                            for (int r = 0; r < 4; ++r) q[r] = (bf16_t)d[t][r];
                            *reinterpret_cast<bf16x4_t*>(dst) = q;
                        } else {
                            *reinterpret_cast<f32x4_t*>(dst) = f32x4_t{d[t][0], d[t][1], d[t][2], d[t][3]};
                        }
                    }
#pragma unroll
                    for (int rn = 0; rn < RUNS; ++rn) {
                        // k-values rn*4*EPL + g*EPL .. + EPL of this pixel = one 16-byte chunk
                        const int chunk = rn * 4 + g;
                        *reinterpret_cast<typename StemConv<T>::Frag*>(my_col + row * COL_RB + ((chunk ^ sf_swz<COL_RB>(row)) << 4)) = b[rn];
                    }
                }
            }
            if (WGRAD) {
                // dW[co][k] += sum over the KR staged pixels of dy[pix][co] * col[pix][k]  (K = pixel MFMA, fragments transposed-read)
                typename StemConv<T>::Frag af[4], bf[2];
#pragma unroll
                for (int t = 0; t < 4; ++t) af[t] = SfFrag<T, DY_RB>::load(my_dy, t * 16, lane);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) bf[kt] = SfFrag<T, COL_RB>::load(my_col, kt * 16, lane);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt) Mma<T>::run(af[t], bf[kt], dw[t][kt]);
            }
        }
    }
    // ---- workgroup result
    __syncthreads();
    if (!WGRAD) {
        float (*red)[2][64] = reinterpret_cast<float (*)[2][64]>(smem);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a = sf_sum16(s1[t][r]), b2 = sf_sum16(s2[t][r]);
                if (i == 0) { red[wave][0][sf_chan(t, r)] = a; red[wave][1][sf_chan(t, r)] = b2; }
            }
        __syncthreads();
        if (threadIdx.x < 64) {
            const int c = threadIdx.x;
            const float sd = red[0][0][c] + red[1][0][c] + red[2][0][c] + red[3][0][c];
            const float sdy = red[0][1][c] + red[1][1][c] + red[2][1][c] + red[3][1][c];
            outbuf[((size_t)blockIdx.x * 2 + 0) * 64 + c] = sd;
            outbuf[((size_t)blockIdx.x * 2 + 1) * 64 + c] = p1[c] * (sdy - p0[c] * sd);      // p0 = mean, p1 = invstd
        }
    } else {
        float* red = reinterpret_cast<float*>(smem);          // [4 waves][64 co][32 k]
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[(wave * 64 + t * 16 + 4 * g + r) * 32 + kt * 16 + i] = dw[t][kt][r];
        __syncthreads();
        for (int o = threadIdx.x; o < 64 * 32; o += 256)
            outbuf[(size_t)blockIdx.x * 2048 + o] = red[o] + red[2048 + o] + red[4096 + o] + red[6144 + o];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward, bf16, scatter form.  The gather form above asks, for each of the 448 x 64 (pixel, channel) pairs of a tile,
// "which of my (up to 4) windows chose me?" -- ~200 VALU ops per pixel-lane and 1.5 ms for the two passes.  Here the
// tile's conv outputs go to LDS once ([448 pixels][64 channels] bf16 in the transposed-read layout) and the 145 x 64
// pooled elements of the tile scatter instead: 7x fewer elements, one select each.
//   WGRAD = false: Y tile = y;   pooled loop: d = dpool * (y*ms+mb > 0) -> { sum d, sum d*y }
//   WGRAD = true : DY tile = cb*y + cc (+ a ReLU mask bit per element); pooled loop in four parity phases (windows of one
//                  phase are disjoint, so the bf16 read-modify-writes need no atomics): DY[argmax pixel] += ca*dpool;
//                  then dW += DY^T x im2col on the matrix pipe, DY fragments transposed-read straight from the tile.
constexpr int SB_THREADS = 512, SB_WAVES = 8;
constexpr int SB_NPIX = 4 * SF_PH * SF_PW;                     // 448 activation pixels per tile
constexpr int SB_TH = SF_PH + 1, SB_TW = SF_PW + 1;            // 5 x 29 pooled windows can reach them
constexpr int SB_DROW = 128 + 16, SB_AROW = 64 + 16;
constexpr int SB_OFF_ARG = SB_TH * SB_TW * SB_DROW;
constexpr int SB_OFF_XS = SB_OFF_ARG + SB_TH * SB_TW * SB_AROW;
constexpr int SB_XR = 2 * SF_PH + 2, SB_XC = 2 * SF_PW + 2;
constexpr int SB_OFF_CONST = SB_OFF_XS + ((3 * SB_XR * SB_XC * 2 + 15) / 16) * 16;
constexpr int SB_OFF_TILE = SB_OFF_CONST + 5 * 64 * 4;
constexpr int SB_OFF_MASK = SB_OFF_TILE + SB_NPIX * 128;
constexpr int SB_OFF_COL = SB_OFF_MASK + SB_NPIX * 8;
constexpr int SB_LDS_REDUCE = SB_OFF_MASK > SB_THREADS * 16 * 4 ? SB_OFF_MASK : SB_THREADS * 16 * 4;
constexpr int SB_LDS_WGRAD = SB_OFF_COL + SB_WAVES * 32 * 128;

// byte address of channel c of pixel row `row` in the [pixel][64 ch] bf16 tile (16-byte chunks swizzled per row)
__device__ __forceinline__ int sb_addr(int row, int c) { return row * 128 + (((c >> 3) ^ sf_swz<128>(row)) << 4) + (c & 7) * 2; }

template <bool WGRAD>
__global__ __launch_bounds__(SB_THREADS, 1) void stem_bwd2_kernel(const float* __restrict__ x, const bf16_t* __restrict__ wp,
                                                                 const bf16_t* __restrict__ dpool, const uint8_t* __restrict__ argmax,
                                                                 const float* __restrict__ p0, const float* __restrict__ p1,
                                                                 const float* __restrict__ p2, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, float* __restrict__ outbuf,
                                                                 int B, int H, int W) {
    typedef bf16_t T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_dp = smem;
    uint8_t* s_arg = reinterpret_cast<uint8_t*>(smem + SB_OFF_ARG);
    T* xs = reinterpret_cast<T*>(smem + SB_OFF_XS);
    float* s_const = reinterpret_cast<float*>(smem + SB_OFF_CONST);     // [0] ms [1] mb [2] ca [3] cb [4] cc
    char* s_tile = smem + SB_OFF_TILE;
    uint32_t* s_mask = reinterpret_cast<uint32_t*>(smem + SB_OFF_MASK);  // [pixel][2 words]
    StemConv<T> sc;
    sc.init(wp, xs, SB_XR, SB_XC);
    const int lane = lane_id(), wave = wave_id(), i = lane & 15, g = lane >> 4;
    char* my_col = smem + SB_OFF_COL + wave * 32 * 128;
    const int Hp = (H - 1) / 2 + 1, Wp = (W - 1) / 2 + 1;
    const int tw = (Wp + SF_PW - 1) / SF_PW, th = (Hp + SF_PH - 1) / SF_PH;
    for (int v = threadIdx.x; v < 64; v += SB_THREADS) {
        s_const[v] = scale[v]; s_const[64 + v] = shift[v];
        if (WGRAD) { s_const[128 + v] = p0[v]; s_const[192 + v] = p1[v]; s_const[256 + v] = p2[v]; }
    }
    auto cvec = [&](int which, int t) { return *reinterpret_cast<const f32x4_t*>(s_const + which * 64 + t * 16 + 4 * g); };
    // the pooled loop gives every thread a fixed 8-channel group (cv) of varying windows
    const int cv = threadIdx.x & 7;
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    f32x4_t dw[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) { dw[t][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dw[t][1] = dw[t][0]; }
    if (WGRAD)
        for (int v = lane; v < 32 * 128 / 16; v += 64) *reinterpret_cast<f32x4_t*>(my_col + v * 16) = f32x4_t{0.f, 0.f, 0.f, 0.f};

    for (int tile = blockIdx.x; tile < B * th * tw; tile += gridDim.x) {
        int rem = tile;
        const int n = rem / (tw * th); rem -= n * tw * th;
        const int ph0 = (rem / tw) * SF_PH, pw0 = (rem % tw) * SF_PW;
        __syncthreads();
        sc.stage(xs, x + (size_t)n * 3 * H * W, 2 * ph0 - 1, 2 * pw0 - 1, H, W);
        for (int v = threadIdx.x; v < SB_TH * SB_TW * 8; v += SB_THREADS) {
            const int pix = v >> 3, part = v & 7;
            const int ph = ph0 + pix / SB_TW, pw = pw0 + pix % SB_TW;
            f32x4_t val = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (ph < Hp && pw < Wp)
                val = *reinterpret_cast<const f32x4_t*>(reinterpret_cast<const char*>(dpool) + ((((size_t)n * Hp + ph) * Wp + pw) * 64) * 2 + part * 16);
            *reinterpret_cast<f32x4_t*>(s_dp + pix * SB_DROW + part * 16) = val;
        }
        for (int v = threadIdx.x; v < SB_TH * SB_TW * 4; v += SB_THREADS) {
            const int pix = v >> 2, part = v & 3;
            const int ph = ph0 + pix / SB_TW, pw = pw0 + pix % SB_TW;
            f32x4_t val = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (ph < Hp && pw < Wp)
                val = *reinterpret_cast<const f32x4_t*>(argmax + (((size_t)n * Hp + ph) * Wp + pw) * 64 + part * 16);
            *reinterpret_cast<f32x4_t*>(s_arg + pix * SB_AROW + part * 16) = val;
        }
        if (WGRAD) for (int v = threadIdx.x; v < SB_NPIX * 2; v += SB_THREADS) s_mask[v] = 0u;
        __syncthreads();
        // ---- A: conv outputs of the tile's pixels -> LDS tile
        for (int it = wave; it * 16 < SB_NPIX; it += SB_WAVES) {
            const int bp = it * 16 + i;
            const int br = bp / (2 * SF_PW), bc = bp - br * (2 * SF_PW);
            const bool live = 2 * ph0 + br < H && 2 * pw0 + bc < W;
            typename StemConv<T>::Frag b[1];
            f32x4_t acc[4];
            sc.gather(br, bc, true, b);
            sc.conv(b, acc);
            uint32_t mbits[2] = {0u, 0u};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int ch0 = t * 16 + 4 * g;
                bf16x4_t q;
                if (WGRAD) {
                    const f32x4_t ms = cvec(0, t), mb = cvec(1, t), cb = cvec(3, t), cc = cvec(4, t);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float y = acc[t][r];
                        q[r] = (bf16_t)(live ? cb[r] * y + cc[r] : 0.f);
                        if (live && y * ms[r] + mb[r] > 0.f) mbits[ch0 >> 5] |= 1u << ((ch0 & 31) + r);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) q[r] = (bf16_t)acc[t][r];
                }
                *reinterpret_cast<bf16x4_t*>(s_tile + sb_addr(bp, ch0)) = q;
            }
            if (WGRAD) {
                if (mbits[0]) atomicOr(&s_mask[bp * 2 + 0], mbits[0]);
                if (mbits[1]) atomicOr(&s_mask[bp * 2 + 1], mbits[1]);
            }
        }
        __syncthreads();
        // ---- B: pooled elements scatter
        const f32x4_t* cms = reinterpret_cast<const f32x4_t*>(s_const + cv * 8);
        const f32x4_t* cmb = reinterpret_cast<const f32x4_t*>(s_const + 64 + cv * 8);
        const f32x4_t* cca = reinterpret_cast<const f32x4_t*>(s_const + 128 + cv * 8);
        auto element = [&](int pl, int pc) {
            const int pix = pl * SB_TW + pc;
            const bf16x8_t dv = *reinterpret_cast<const bf16x8_t*>(s_dp + pix * SB_DROW + cv * 16);
            const uint64_t a8 = *reinterpret_cast<const uint64_t*>(s_arg + pix * SB_AROW + cv * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int tap = (int)((a8 >> (8 * e)) & 0xffu);
                const int br = 2 * pl - 1 + tap / 3, bc = 2 * pc - 1 + (tap - (tap / 3) * 3);
                if ((unsigned)br >= (unsigned)(2 * SF_PH) || (unsigned)bc >= (unsigned)(2 * SF_PW)) continue;   // a neighbour tile's pixel
                const int bp = br * (2 * SF_PW) + bc, c = cv * 8 + e;
                bf16_t* slot = reinterpret_cast<bf16_t*>(s_tile + sb_addr(bp, c));
                const float d = (float)dv[e];
                if (WGRAD) {
                    if ((s_mask[bp * 2 + (c >> 5)] >> (c & 31)) & 1u) *slot = (bf16_t)((float)*slot + cca[e >> 2][e & 3] * d);
                } else {
                    const float y = (float)*slot;
                    const float de = (y * cms[e >> 2][e & 3] + cmb[e >> 2][e & 3] > 0.f) ? d : 0.f;
                    s1[e] += de; s2[e] += de * y;
                }
            }
        };
        if (WGRAD) {
            for (int phase = 0; phase < 4; ++phase) {
                const int a = phase >> 1, b2 = phase & 1;
                const int nr = (SB_TH - a + 1) / 2, nc = (SB_TW - b2 + 1) / 2;      // windows of this parity
                for (int task = threadIdx.x; task < nr * nc * 8; task += SB_THREADS) {
                    const int widx = task >> 3;
                    element(a + 2 * (widx / nc), b2 + 2 * (widx % nc));
                }
                __syncthreads();
            }
        } else {
            for (int task = threadIdx.x; task < SB_TH * SB_TW * 8; task += SB_THREADS) {
                const int widx = task >> 3;
                element(widx / SB_TW, widx % SB_TW);
            }
        }
        // ---- C (WGRAD): dW += DY^T x im2col, 32 pixels per MFMA group
        if (WGRAD) {
            for (int grp = wave; grp * 32 < SB_NPIX; grp += SB_WAVES) {
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {
                    const int bp = grp * 32 + sub * 16 + i;
                    const int br = bp / (2 * SF_PW), bc = bp - br * (2 * SF_PW);
                    typename StemConv<T>::Frag b[1];
                    sc.gather(br, bc, true, b);
                    const int row = sub * 16 + i;
                    *reinterpret_cast<bf16x8_t*>(my_col + row * 128 + ((g ^ sf_swz<128>(row)) << 4)) = b[0];
                }
                bf16x8_t af[4], bf[2];
#pragma unroll
                for (int t = 0; t < 4; ++t) af[t] = SfFrag<T, 128>::load(s_tile + grp * 32 * 128, t * 16, lane);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) bf[kt] = SfFrag<T, 128>::load(my_col, kt * 16, lane);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt) Mma<T>::run(af[t], bf[kt], dw[t][kt]);
            }
        }
    }
    __syncthreads();
    if (!WGRAD) {
        float* red = reinterpret_cast<float*>(smem);              // [thread][16]
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[threadIdx.x * 16 + e] = s1[e]; red[threadIdx.x * 16 + 8 + e] = s2[e]; }
        __syncthreads();
        if (threadIdx.x < 64) {
            const int c = threadIdx.x, cvv = c >> 3, e = c & 7;
            float sd = 0.f, sdy = 0.f;
            for (int tt = cvv; tt < SB_THREADS; tt += 8) { sd += red[tt * 16 + e]; sdy += red[tt * 16 + 8 + e]; }
            outbuf[((size_t)blockIdx.x * 2 + 0) * 64 + c] = sd;
            outbuf[((size_t)blockIdx.x * 2 + 1) * 64 + c] = p1[c] * (sdy - p0[c] * sd);      // p0 = mean, p1 = invstd
        }
    } else {
        float* red = reinterpret_cast<float*>(smem);              // [8 waves][64 co][32 k]
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[(wave * 64 + t * 16 + 4 * g + r) * 32 + kt * 16 + i] = dw[t][kt][r];
        __syncthreads();
        for (int o = threadIdx.x; o < 64 * 32; o += SB_THREADS) {
            float a = 0.f;
#pragma unroll
            for (int wv = 0; wv < SB_WAVES; ++wv) a += red[wv * 2048 + o];
            outbuf[(size_t)blockIdx.x * 2048 + o] = a;
        }
    }
}

// dw[co][27] += sum over slabs of slab[co][32] (k < 27)
__global__ __launch_bounds__(256) void stem_dw_reduce_kernel(const float* __restrict__ slabs, int nslabs, float* __restrict__ dw) {
    __shared__ float red[256];
    const int o = blockIdx.x;                                 // one output (co, k) per block
    const int co = o / 27, k = o - co * 27;
    float acc = 0.f;
    for (int s = threadIdx.x; s < nslabs; s += 256) acc += slabs[(size_t)s * 2048 + co * 32 + k];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) dw[o] += red[0];
}

static int sf_tiles(int b, int h, int w) {
    const int hp = (h - 1) / 2 + 1, wp = (w - 1) / 2 + 1;
    return b * ((hp + SF_PH - 1) / SF_PH) * ((wp + SF_PW - 1) / SF_PW);
}

}  // namespace frhip

using namespace frhip;

extern "C" int frhip_stem_blocks(int b, int h, int w) {
    // workgroups (= partial rows / weight-gradient slabs) of the grid-stride stem kernels
    const int t = sf_tiles(b, h, w);
    return t < 1024 ? t : 1024;
}

static bool sf_ok(int dtype, int b, int h, int w, const char* who) {
    if ((dtype != FRHIP_DT_BF16 && dtype != FRHIP_DT_F32) || b <= 0 || h <= 0 || w <= 0 ||
        3LL * b * h * w > 0x7fffffffLL) {
        set_error("%s: unsupported dtype / shape (dtype=%d b=%d h=%d w=%d)", who, dtype, b, h, w);
        return false;
    }
    return true;
}

extern "C" int frhip_stem_stats(int dtype, const float* x, const void* wp, int b, int h, int w, float* partial,
                                hipStream_t stream) {
    if (!sf_ok(dtype, b, h, w, "frhip_stem_stats")) return FRHIP_EINVAL;
    const int blocks = frhip_stem_blocks(b, h, w);
    const int es = dtype == FRHIP_DT_BF16 ? 2 : 4;
    int lds = 3 * 10 * (w + 2) * es;
    if (lds < 4 * 2 * 64 * 4) lds = 4 * 2 * 64 * 4;
    if (lds > 64 * 1024) { set_error("frhip_stem_stats: image too wide (%d)", w); return FRHIP_EINVAL; }
    if (dtype == FRHIP_DT_BF16)
        hipLaunchKernelGGL(stem_stats_kernel<bf16_t>, dim3(blocks), dim3(256), lds, stream, x, (const bf16_t*)wp, partial, b, h, w);
    else
        hipLaunchKernelGGL(stem_stats_kernel<float>, dim3(blocks), dim3(256), lds, stream, x, (const float*)wp, partial, b, h, w);
    return check_launch("frhip_stem_stats");
}

template <typename T>
static int sf_fwd(const float* x, const void* wp, const float* scale, const float* shift, void* pooled, uint8_t* argmax,
                  int b, int h, int w, hipStream_t stream) {
    const int lds = SF_RH * SF_RW * (64 * (int)sizeof(T) + 16) + 3 * (SF_RH + 2) * (SF_RW + 2) * (int)sizeof(T);
    auto kern = stem_fwd_kernel<T>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            set_error("frhip_stem_fwd: cannot raise dynamic LDS to %d bytes", lds);
            return FRHIP_ELAUNCH;
        }
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(sf_tiles(b, h, w)), dim3(256), lds, stream, x, (const T*)wp, scale, shift, (T*)pooled, argmax, b, h, w);
    return check_launch("frhip_stem_fwd");
}

extern "C" int frhip_stem_fwd(int dtype, const float* x, const void* wp, const float* scale, const float* shift,
                              void* pooled, uint8_t* argmax, int b, int h, int w, hipStream_t stream) {
    if (!sf_ok(dtype, b, h, w, "frhip_stem_fwd")) return FRHIP_EINVAL;
    return dtype == FRHIP_DT_BF16 ? sf_fwd<bf16_t>(x, wp, scale, shift, pooled, argmax, b, h, w, stream)
                                  : sf_fwd<float>(x, wp, scale, shift, pooled, argmax, b, h, w, stream);
}

template <typename T, bool WGRAD>
static int sf_bwd(const float* x, const void* wp, const void* dpool, const uint8_t* argmax, const float* p0, const float* p1,
                  const float* p2, const float* scale, const float* shift, float* outbuf, int b, int h, int w,
                  hipStream_t stream, const char* who) {
    constexpr int KR = sizeof(T) == 2 ? 32 : 16;
    const int tile_bytes = (SF_PH + 1) * (SF_PW + 1) * (64 * (int)sizeof(T) + 16 + 80) +
                           (WGRAD ? 4 * KR * (64 * (int)sizeof(T) + 128) : 0) + 5 * 64 * 4 +
                           3 * (2 * SF_PH + 2) * (2 * SF_PW + 2) * (int)sizeof(T);
    const int red_bytes = WGRAD ? 4 * 64 * 32 * 4 : 4 * 2 * 64 * 4;
    const int lds = tile_bytes > red_bytes ? tile_bytes : red_bytes;
    auto kern = stem_bwd_kernel<T, WGRAD>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            set_error("%s: cannot raise dynamic LDS to %d bytes", who, lds);
            return FRHIP_ELAUNCH;
        }
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(frhip_stem_blocks(b, h, w)), dim3(256), lds, stream, x, (const T*)wp, (const T*)dpool, argmax,
                       p0, p1, p2, scale, shift, outbuf, b, h, w);
    return check_launch(who);
}

template <bool WGRAD>
static int sf_bwd2(const float* x, const void* wp, const void* dpool, const uint8_t* argmax, const float* p0, const float* p1,
                   const float* p2, const float* scale, const float* shift, float* outbuf, int b, int h, int w,
                   hipStream_t stream, const char* who) {
    const int lds = WGRAD ? SB_LDS_WGRAD : SB_LDS_REDUCE;
    static_assert(SB_LDS_WGRAD >= SB_WAVES * 64 * 32 * 4 && SB_LDS_WGRAD <= 160 * 1024, "LDS budget");
    auto kern = stem_bwd2_kernel<WGRAD>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            set_error("%s: cannot raise dynamic LDS to %d bytes", who, lds);
            return FRHIP_ELAUNCH;
        }
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(frhip_stem_blocks(b, h, w)), dim3(SB_THREADS), lds, stream, x, (const bf16_t*)wp,
                       (const bf16_t*)dpool, argmax, p0, p1, p2, scale, shift, outbuf, b, h, w);
    return check_launch(who);
}

static int g_stem_scatter = 1;       // test hook: 0 = gather-form backward for bf16 too

extern "C" int frhip_stem_bwd_reduce(int dtype, const float* x, const void* wp, const void* dpool, const uint8_t* argmax,
                                     const float* mean, const float* invstd, const float* scale, const float* shift,
                                     int b, int h, int w, float* partial, hipStream_t stream) {
    if (!sf_ok(dtype, b, h, w, "frhip_stem_bwd_reduce")) return FRHIP_EINVAL;
    if (dtype == FRHIP_DT_BF16 && g_stem_scatter)
        return sf_bwd2<false>(x, wp, dpool, argmax, mean, invstd, nullptr, scale, shift, partial, b, h, w, stream, "frhip_stem_bwd_reduce");
    return dtype == FRHIP_DT_BF16
        ? sf_bwd<bf16_t, false>(x, wp, dpool, argmax, mean, invstd, nullptr, scale, shift, partial, b, h, w, stream, "frhip_stem_bwd_reduce")
        : sf_bwd<float, false>(x, wp, dpool, argmax, mean, invstd, nullptr, scale, shift, partial, b, h, w, stream, "frhip_stem_bwd_reduce");
}

extern "C" int frhip_stem_bwd_wgrad(int dtype, const float* x, const void* wp, const void* dpool, const uint8_t* argmax,
                                    const float* ca, const float* cb, const float* cc, const float* scale, const float* shift,
                                    int b, int h, int w, float* slabs, float* dw, hipStream_t stream) {
    if (!sf_ok(dtype, b, h, w, "frhip_stem_bwd_wgrad")) return FRHIP_EINVAL;
    int rc = (dtype == FRHIP_DT_BF16 && g_stem_scatter)
        ? sf_bwd2<true>(x, wp, dpool, argmax, ca, cb, cc, scale, shift, slabs, b, h, w, stream, "frhip_stem_bwd_wgrad")
        : dtype == FRHIP_DT_BF16
        ? sf_bwd<bf16_t, true>(x, wp, dpool, argmax, ca, cb, cc, scale, shift, slabs, b, h, w, stream, "frhip_stem_bwd_wgrad")
        : sf_bwd<float, true>(x, wp, dpool, argmax, ca, cb, cc, scale, shift, slabs, b, h, w, stream, "frhip_stem_bwd_wgrad");
    if (rc) return rc;
    hipLaunchKernelGGL(stem_dw_reduce_kernel, dim3(64 * 27), dim3(256), 0, stream, slabs, frhip_stem_blocks(b, h, w), dw);
    return check_launch("frhip_stem_bwd_wgrad(reduce)");
}

extern "C" int frhip_set_stem_scatter(int enabled) { const int old = g_stem_scatter; g_stem_scatter = enabled; return old; }
