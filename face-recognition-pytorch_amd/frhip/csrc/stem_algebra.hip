// Weight gradient of the recompute-style stem WITHOUT recomputing the convolution and without a scatter, gfx950.
// Reference: autograd of /root/reference/nets/resnet.py:186-189, :232-235 (conv3x3(3->64, s1) -> BN -> ReLU -> MaxPool(3,2,1)).
//
// The stem's BatchNorm backward gives, for every pixel p of the 112 x 112 conv map,  dy[k][p] = ca[k] d[k][p] + cb[k] y[k][p] + cc[k]
// (d = pooled gradient routed to the arg-max pixels behind the ReLU mask, y = conv output), and
//     dW[k][j] = sum_p dy[k][p] col[p][j]                         (col[p] = the 27 inputs under pixel p, zero padded)
//              = ca[k] D[k][j] + cb[k] (W G)[k][j] + cc[k] s[j]
//   G[j][j'] = sum_p col[p][j] col[p][j']   (27 x 27, a function of the INPUT BATCH only),   s[j] = sum_p col[p][j],
//   D[k][j]  = sum over POOLED elements (q, k) with pooled > 0 of  dpool[q][k] col[argmax pixel of (q, k)][j].
// because y[k][p] = sum_j' W[k][j'] col[p][j'].  So:
//   * stem_gram_kernel  -- G and s, a pass over the 77-MB input with VALU work only.  It depends on nothing the step computes:
//     the caller runs it on the side stream during the forward pass;
//   * stem_dgather_kernel -- D: one lane per (pooled pixel, channel), 27 LDS reads + 27 FMAs from the staged input window at the
//     arg-max position; 4x fewer elements than the conv map, no read-modify-write, no matrix pipe;
//   * stem_dw_final_kernel -- sums the slabs and applies the formula (64 x 27 outputs).
// It replaces stem_bwd2_kernel<true> (0.78 ms at B = 512: conv recompute on the matrix pipe, cb y + cc over 411 M elements, LDS
// read-modify-write scatter in four parity phases, transposed-read weight-gradient GEMM).
// Inputs are rounded to T exactly as the recompute kernels round them (x and W feed the MFMA as T there), so y is the same sum.
#include "common.h"
#include "frhip.h"

namespace frhip {

constexpr int SG_ROWS = 8;                        // image rows per Gram tile
constexpr int SG_ROLES = 8;                       // 6 channel pairs (c <= c') + column sums + idle
constexpr int SG_VALS = 81;                       // a role's accumulators: [a][b], a, b = window offsets 0..8
constexpr int SG_OUT = 7 * SG_VALS;               // floats a workgroup emits (6 pair blocks + the sums block)

// ---- G and s.  Tile = SG_ROWS image rows of one image; thread = (role, pixel lane): role r < 6 accumulates the 9 x 9 block
//      B[a][b] = sum_p x_c(p + a) x_c'(p + b) of the channel pair (c, c'), role 6 the 27 window sums.
template <typename T>
__global__ __launch_bounds__(256) void stem_gram_kernel(const float* __restrict__ x, float* __restrict__ partial, int B, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xs = reinterpret_cast<float*>(smem);                 // [3][SG_ROWS + 2][W + 2], rounded through T
    const int XR = SG_ROWS + 2, XC = W + 2;
    const int role = threadIdx.x & 7, pl = threadIdx.x >> 3;    // 32 pixel lanes
    const int ca_ = role < 3 ? 0 : (role < 5 ? 1 : 2);          // pairs: (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
    const int cb_ = role < 3 ? role : (role < 5 ? role - 2 : 2);
    float acc[SG_VALS];
#pragma unroll
    for (int i = 0; i < SG_VALS; ++i) acc[i] = 0.f;
    const int bands = (H + SG_ROWS - 1) / SG_ROWS;
    for (int tile = blockIdx.x; tile < B * bands; tile += gridDim.x) {
        const int n = tile / bands, h0 = (tile - n * bands) * SG_ROWS;
        const float* ximg = x + (size_t)n * 3 * H * W;
        __syncthreads();
        for (int idx = threadIdx.x; idx < 3 * XR * XC; idx += 256) {
            const int cc = idx % XC, rest = idx / XC, rr = rest % XR, ci = rest / XR;
            const int h = h0 - 1 + rr, w = cc - 1;
            const float v = ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) ? ximg[((size_t)ci * H + h) * W + w] : 0.f;
            xs[idx] = to_f32<T>(from_f32<T>(v));
        }
        __syncthreads();
        const int rows = min(SG_ROWS, H - h0);
        for (int p = pl; p < rows * W; p += 32) {
            const int py = p / W, px = p - py * W;
            const float* base = xs + py * XC + px;               // top-left tap of pixel p
            if (role < 6) {
                float va[9], vb[9];
#pragma unroll
                for (int a = 0; a < 9; ++a) {
                    va[a] = base[(ca_ * XR + a / 3) * XC + a % 3];
                    vb[a] = base[(cb_ * XR + a / 3) * XC + a % 3];
                }
#pragma unroll
                for (int a = 0; a < 9; ++a)
#pragma unroll
                    for (int b = 0; b < 9; ++b) acc[a * 9 + b] = fmaf(va[a], vb[b], acc[a * 9 + b]);
            } else if (role == 6) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int a = 0; a < 9; ++a) acc[c * 9 + a] += base[(c * XR + a / 3) * XC + a % 3];
            }
        }
    }
    // reduce over the 8 lanes of a wave that share a role (lane bits 3..5), then over the 4 waves through LDS
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);                // [4 waves][8 roles][81]
    const int lane = lane_id(), wave = wave_id();
#pragma unroll
    for (int i = 0; i < SG_VALS; ++i) {
        float v = acc[i];
        v = lane_sum_bit3(v); v = lane_sum_bit4(v); v = lane_sum_bit5(v);
        if (lane < 8) red[(wave * 8 + lane) * SG_VALS + i] = v;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < SG_OUT; o += 256)
        partial[(size_t)blockIdx.x * SG_OUT + o] = red[o] + red[8 * SG_VALS + o] + red[16 * SG_VALS + o] + red[24 * SG_VALS + o];
}

// blk[7][81] = sum of the workgroups' partial blocks: one workgroup per role block, three lanes per output
__global__ __launch_bounds__(256) void stem_gram_fold_kernel(const float* __restrict__ partial, int nparts, float* __restrict__ blk) {
    __shared__ float red[3][SG_VALS];
    const int o = threadIdx.x % SG_VALS, part = threadIdx.x / SG_VALS;
    if (part < 3) {
        float a = 0.f;
        for (int p = part; p < nparts; p += 3) a += partial[(size_t)p * SG_OUT + blockIdx.x * SG_VALS + o];
        red[part][o] = a;
    }
    __syncthreads();
    if (threadIdx.x < SG_VALS) blk[blockIdx.x * SG_VALS + o] = red[0][o] + red[1][o] + red[2][o];
}

// gram[0..728] = G[j][j'] (j = (fr * 3 + fs) * 3 + ci, the k order of the packed stem weights), gram[729..755] = s[j]
__global__ __launch_bounds__(256) void stem_gram_assemble_kernel(const float* __restrict__ blk, float* __restrict__ gram) {
    for (int o = threadIdx.x; o < 27 * 27 + 27; o += 256) {
        if (o < 729) {
            const int j = o / 27, j2 = o - j * 27;
            const int a = j / 3, c = j - a * 3, b = j2 / 3, c2 = j2 - b * 3;
            // block of the pair (min, max): role index 0..5 for (0,0) (0,1) (0,2) (1,1) (1,2) (2,2); B_{c c'}[a][b], transposed for c > c'
            const int lo = c < c2 ? c : c2, hi = c < c2 ? c2 : c;
            const int role = lo == 0 ? hi : (lo == 1 ? 2 + hi : 5);
            gram[o] = c <= c2 ? blk[role * SG_VALS + a * 9 + b] : blk[role * SG_VALS + b * 9 + a];
        } else {
            const int j = o - 729, a = j / 3, c = j - a * 3;
            gram[o] = blk[6 * SG_VALS + c * 9 + a];
        }
    }
}

// ---- D.  Tile = DG_P x DG_P pooled pixels of one image; wave = pooled pixel, lane = channel.
constexpr int DG_P = 8;
constexpr int DG_X = 2 * DG_P + 3;                 // input window edge: activation rows 2 ph0 - 1 .. 2 (ph0 + 7) + 1, one tap each side

template <typename T>
__global__ __launch_bounds__(256) void stem_dgather_kernel(const float* __restrict__ x, const T* __restrict__ dpool,
                                                           const T* __restrict__ pooled, const uint8_t* __restrict__ argmax,
                                                           float* __restrict__ slabs, int B, int H, int W) {
    __shared__ float xs[3 * DG_X * DG_X];
    __shared__ float gs[4 * 64 * 27];               // masked pooled gradient of the tile [64 pixels][64 ch]; at the end the waves' sums
    __shared__ uint8_t ab8[DG_P * DG_P * 64];
    const int Hp = (H - 1) / 2 + 1, Wp = (W - 1) / 2 + 1;
    const int th = (Hp + DG_P - 1) / DG_P, tw = (Wp + DG_P - 1) / DG_P;
    const int lane = lane_id(), wave = wave_id();
    float acc[27];
#pragma unroll
    for (int j = 0; j < 27; ++j) acc[j] = 0.f;
    constexpr int EPV = 16 / (int)sizeof(T), VPR = 64 / EPV;
    constexpr int XN = (3 * DG_X * DG_X + 255) / 256;           // window elements per thread
    constexpr int VN = DG_P * DG_P * VPR / 256;                 // 16-byte channel groups per thread
    static_assert(DG_P * DG_P * VPR % 256 == 0, "tile vectors split evenly over the threads");
    // The next tile's global operands are fetched into registers while this tile is being gathered: with one buffer and the loads
    // in front of the barrier every tile paid two dependent HBM round trips with nothing else in flight.
    float xr[XN];
    Vec16<T> gr[VN], pr[VN];
    uint64_t ar[VN];
    bool inr[VN];
    auto fetch = [&](int tile) {
        const int n = tile / (th * tw), rem = tile - n * th * tw;
        const int ph0 = (rem / tw) * DG_P, pw0 = (rem % tw) * DG_P;
        const float* ximg = x + (size_t)n * 3 * H * W;
#pragma unroll
        for (int i = 0; i < XN; ++i) {
            const int idx = threadIdx.x + i * 256;
            const int cc = idx % DG_X, rest = idx / DG_X, rr = rest % DG_X, ci = rest / DG_X;
            const int h = 2 * ph0 - 2 + rr, w = 2 * pw0 - 2 + cc;
            xr[i] = (idx < 3 * DG_X * DG_X && (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) ? ximg[((size_t)ci * H + h) * W + w] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < VN; ++i) {
            const int idx = threadIdx.x + i * 256;
            const int cv = idx % VPR, pp = idx / VPR, pl = pp / DG_P, pc = pp - pl * DG_P;
            const int ph = ph0 + pl, pw = pw0 + pc;
            inr[i] = ph < Hp && pw < Wp;
            const size_t o = ((((size_t)n * Hp + ph) * Wp + pw) * VPR + cv) * EPV;
            ar[i] = 0;
            if (inr[i]) {
                gr[i] = *reinterpret_cast<const Vec16<T>*>(dpool + o);
                pr[i] = *reinterpret_cast<const Vec16<T>*>(pooled + o);
                if constexpr (EPV == 8) ar[i] = *reinterpret_cast<const uint64_t*>(argmax + o);
                else ar[i] = *reinterpret_cast<const uint32_t*>(argmax + o);
            }
        }
    };
    const int ntiles = B * th * tw;
    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads();                                          // the previous tile's gather is done with xs / gs / ab8
#pragma unroll
        for (int i = 0; i < XN; ++i) {
            const int idx = threadIdx.x + i * 256;
            if (idx < 3 * DG_X * DG_X) xs[idx] = to_f32<T>(from_f32<T>(xr[i]));
        }
#pragma unroll
        for (int i = 0; i < VN; ++i) {
            const int idx = threadIdx.x + i * 256;
            const int cv = idx % VPR, pp = idx / VPR;
#pragma unroll
            for (int e = 0; e < EPV; ++e) gs[pp * 64 + cv * EPV + e] = (inr[i] && pr[i].get(e) > 0.f) ? gr[i].get(e) : 0.f;
            if constexpr (EPV == 8) *reinterpret_cast<uint64_t*>(ab8 + pp * 64 + cv * 8) = ar[i];
            else *reinterpret_cast<uint32_t*>(ab8 + pp * 64 + cv * 4) = (uint32_t)ar[i];
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
        for (int pp = wave; pp < DG_P * DG_P; pp += 4) {
            const int pl = pp / DG_P, pc = pp - pl * DG_P;
            const float g = gs[pp * 64 + lane];
            const int a = ab8[pp * 64 + lane];
            const int r = (a * 11) >> 5, s = a - 3 * r;           // arg-max window position: activation pixel (2 pl + r, 2 pc + s) of the region
            const float* base = xs + (2 * pl + r) * DG_X + 2 * pc + s;
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) acc[t * 3 + ci] = fmaf(g, base[(ci * DG_X + t / 3) * DG_X + t % 3], acc[t * 3 + ci]);
        }
    }
    __syncthreads();
    float* red = gs;                                              // [4 waves][64][27]
#pragma unroll
    for (int j = 0; j < 27; ++j) red[(wave * 64 + lane) * 27 + j] = acc[j];
    __syncthreads();
    for (int o = threadIdx.x; o < 64 * 32; o += 256) {
        const int k = o >> 5, j = o & 31;
        float a = 0.f;
        if (j < 27)
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) a += red[(wv * 64 + k) * 27 + j];
        slabs[(size_t)blockIdx.x * 2048 + o] = a;
    }
}

// dw[k][j] += ca[k] * sum_slabs D[k][j] + cb[k] * sum_j' W[k][j'] G[j'][j] + cc[k] * s[j];  one workgroup per channel k
template <typename T>
__global__ __launch_bounds__(1024) void stem_dw_final_kernel(const float* __restrict__ slabs, int nslabs, const float* __restrict__ gram,
                                                            const T* __restrict__ wp, const float* __restrict__ ca,
                                                            const float* __restrict__ cb, const float* __restrict__ cc,
                                                            float* __restrict__ dw) {
    __shared__ float red[32][32];
    const int k = blockIdx.x, j = threadIdx.x & 31, part = threadIdx.x >> 5;
    float a = 0.f;
    for (int s = part; s < nslabs; s += 32) a += slabs[(size_t)s * 2048 + k * 32 + j];
    red[part][j] = a;
    __syncthreads();
    if (threadIdx.x < 27) {
        float d = 0.f;
#pragma unroll
        for (int p = 0; p < 32; ++p) d += red[p][j];
        float wg = 0.f;
        for (int j2 = 0; j2 < 27; ++j2) wg = fmaf(to_f32<T>(wp[k * 32 + j2]), gram[j2 * 27 + j], wg);
        dw[k * 27 + j] += ca[k] * d + cb[k] * wg + cc[k] * gram[729 + j];
    }
}

}  // namespace frhip

using namespace frhip;

static bool sa_ok(int dtype, int b, int h, int w, const char* who) {
    if ((dtype != FRHIP_DT_BF16 && dtype != FRHIP_DT_F32) || b <= 0 || h <= 0 || w <= 0 || 3LL * b * h * w > 0x7fffffffLL ||
        3 * (SG_ROWS + 2) * (w + 2) * 4 > 60 * 1024) {
        set_error("%s: unsupported dtype / shape (dtype=%d b=%d h=%d w=%d)", who, dtype, b, h, w);
        return false;
    }
    return true;
}

extern "C" int frhip_stem_gram_floats(void) { return 27 * 27 + 27; }

extern "C" int frhip_stem_gram_blocks(int b, int h, int w) {
    // workgroups of the Gram pass; the scratch it needs is (this + 1) x 567 floats
    const int t = b * ((h + SG_ROWS - 1) / SG_ROWS);
    return t < 512 ? t : 512;
}

extern "C" int frhip_stem_gram(int dtype, const float* x, int b, int h, int w, float* partial, float* gram, hipStream_t stream) {
    if (!sa_ok(dtype, b, h, w, "frhip_stem_gram")) return FRHIP_EINVAL;
    const int blocks = frhip_stem_gram_blocks(b, h, w);
    int lds = 3 * (SG_ROWS + 2) * (w + 2) * 4;
    if (lds < 4 * 8 * SG_VALS * 4) lds = 4 * 8 * SG_VALS * 4;
    if (dtype == FRHIP_DT_BF16) hipLaunchKernelGGL(stem_gram_kernel<bf16_t>, dim3(blocks), dim3(256), lds, stream, x, partial, b, h, w);
    else hipLaunchKernelGGL(stem_gram_kernel<float>, dim3(blocks), dim3(256), lds, stream, x, partial, b, h, w);
    int rc = check_launch("frhip_stem_gram");
    if (rc) return rc;
    float* blk = partial + (size_t)blocks * SG_OUT;            // 567 more floats of the caller's scratch
    hipLaunchKernelGGL(stem_gram_fold_kernel, dim3(7), dim3(256), 0, stream, partial, blocks, blk);
    hipLaunchKernelGGL(stem_gram_assemble_kernel, dim3(1), dim3(256), 0, stream, blk, gram);
    return check_launch("frhip_stem_gram(reduce)");
}

extern "C" int frhip_stem_bwd_wgrad_gram(int dtype, const float* x, const void* wp, const void* dpool, const void* pooled,
                                         const uint8_t* argmax, const float* gram, const float* ca, const float* cb, const float* cc,
                                         int b, int h, int w, float* slabs, float* dw, hipStream_t stream) {
    if (!sa_ok(dtype, b, h, w, "frhip_stem_bwd_wgrad_gram")) return FRHIP_EINVAL;
    const int blocks = frhip_stem_blocks(b, h, w);
    if (dtype == FRHIP_DT_BF16)
        hipLaunchKernelGGL(stem_dgather_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, x, (const bf16_t*)dpool, (const bf16_t*)pooled,
                           argmax, slabs, b, h, w);
    else
        hipLaunchKernelGGL(stem_dgather_kernel<float>, dim3(blocks), dim3(256), 0, stream, x, (const float*)dpool, (const float*)pooled,
                           argmax, slabs, b, h, w);
    int rc = check_launch("frhip_stem_bwd_wgrad_gram(gather)");
    if (rc) return rc;
    if (dtype == FRHIP_DT_BF16)
        hipLaunchKernelGGL(stem_dw_final_kernel<bf16_t>, dim3(64), dim3(1024), 0, stream, slabs, blocks, gram, (const bf16_t*)wp, ca, cb, cc, dw);
    else
        hipLaunchKernelGGL(stem_dw_final_kernel<float>, dim3(64), dim3(1024), 0, stream, slabs, blocks, gram, (const float*)wp, ca, cb, cc, dw);
    return check_launch("frhip_stem_bwd_wgrad_gram(final)");
}
