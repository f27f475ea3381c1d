// Fused margin-softmax head for gfx950 (PartialFC / ArcFace / distributed softmax cross-entropy).
//   forward : l2-normalise -> cos(theta) GEMM (MFMA) -> clamp -> additive angular margin on the target column ->
//             x s -> per-row online max / sum-exp partials.  Logits are never written to HBM.
//   backward: the same GEMM is recomputed, the tile is turned into d(loss)/d(cos) in registers and stored once
//             (compute dtype); dW = dT^T E and dE = dT W then run on the TN GEMM.
// Reference chain this replaces (file:line in /root/reference):
//   nets/PartialFC.py:198-204 normalize + linear + clamp, nets/ArcFace.py:76-91 margin,
//   nets/PartialFC.py:441-484 DistCrossEntropyFunc forward/backward.
// Cross-rank steps (all-reduce MAX / SUM of the per-row scalars) are done by the host between these kernels.
#include "igemm_nt.h"
#include "frhip.h"

namespace frhip {

struct MarginConst { float s, cos_m, sin_m, theta, sinmm; };

// one wave per row: xhat = x / max(|x|, eps) (T), norm (fp32)
template <typename T>
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float* __restrict__ x, T* __restrict__ xhat,
                                                          float* __restrict__ norms, int rows, int D, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (size_t)row * D;
    float ss = 0.f;
    for (int j = lane * 4; j < D; j += 256) {
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(xr + j);
        ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) ss += __shfl_xor(ss, d);
    const float nrm = fmaxf(sqrtf(ss), eps), inv = 1.f / nrm;
    if (lane == 0) norms[row] = nrm;
    T* o = xhat + (size_t)row * D;
    for (int j = lane * 4; j < D; j += 256) {
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(xr + j);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[j + e] = from_f32<T>(v[e] * inv);
    }
}

// D == 512 (the embedding width of every configuration): a lane owns eight consecutive elements of its row -- 16-byte loads and stores,
// the row read ONCE and kept in registers, the row sum by DPP / permlane (no LDS permutes).  The generic kernels above / below issue
// 4- and 2-byte accesses and read the row twice (156 us for the 122 000 x 512 classifier in the backward pass).
__device__ __forceinline__ float wave_sum(float x) { return lane_sum_bit5(lane_sum_bit4(lane_sum_row16(x))); }

__global__ __launch_bounds__(256) void l2norm_rows512_kernel(const float* __restrict__ x, bf16_t* __restrict__ xhat,
                                                             float* __restrict__ norms, int rows, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (size_t)row * 512 + lane * 8;
    const f32x4_t a = *reinterpret_cast<const f32x4_t*>(xr), b = *reinterpret_cast<const f32x4_t*>(xr + 4);
    float ss = a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3] + b[0] * b[0] + b[1] * b[1] + b[2] * b[2] + b[3] * b[3];
    ss = wave_sum(ss);
    const float nrm = fmaxf(sqrtf(ss), eps), inv = 1.f / nrm;
    if (lane == 0) norms[row] = nrm;
    bf16x8_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = (bf16_t)(a[e] * inv); o[4 + e] = (bf16_t)(b[e] * inv); }
    *reinterpret_cast<bf16x8_t*>(xhat + (size_t)row * 512 + lane * 8) = o;
}

__global__ __launch_bounds__(256) void l2norm_bwd512_kernel(const float* __restrict__ dxhat, const bf16_t* __restrict__ xhat,
                                                            const float* __restrict__ norms, float* __restrict__ dx, int rows,
                                                            float out_scale) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const size_t off = (size_t)row * 512 + lane * 8;
    const f32x4_t a = *reinterpret_cast<const f32x4_t*>(dxhat + off), b = *reinterpret_cast<const f32x4_t*>(dxhat + off + 4);
    const bf16x8_t h = *reinterpret_cast<const bf16x8_t*>(xhat + off);
    float hv[8], dot = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) hv[e] = (float)h[e];
#pragma unroll
    for (int e = 0; e < 4; ++e) dot += a[e] * hv[e] + b[e] * hv[4 + e];
    dot = wave_sum(dot);
    const float inv = out_scale / norms[row];
    f32x4_t oa, ob;
#pragma unroll
    for (int e = 0; e < 4; ++e) { oa[e] = (a[e] - hv[e] * dot) * inv; ob[e] = (b[e] - hv[4 + e] * dot) * inv; }
    *reinterpret_cast<f32x4_t*>(dx + off) = oa;
    *reinterpret_cast<f32x4_t*>(dx + off + 4) = ob;
}

// dx = (dxhat - xhat * <dxhat, xhat>) / norm     (all fp32 except xhat which is T)
template <typename T>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dxhat, const T* __restrict__ xhat,
                                                         const float* __restrict__ norms, float* __restrict__ dx,
                                                         int rows, int D, float out_scale) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* dr = dxhat + (size_t)row * D;
    const T* hr = xhat + (size_t)row * D;
    float dot = 0.f;
    for (int j = lane; j < D; j += 64) dot += dr[j] * to_f32<T>(hr[j]);
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) dot += __shfl_xor(dot, d);
    const float inv = out_scale / norms[row];
    for (int j = lane; j < D; j += 64) dx[(size_t)row * D + j] = (dr[j] - to_f32<T>(hr[j]) * dot) * inv;
}

// ---------------------------------------------------------------------------------------------------------
// FWD = true : partial row max / sum-exp per 64-class column group, target logit.
// FWD = false: dT tile (compute dtype) from the recomputed cosines and the global row max / sum.
template <typename T, bool FWD>
__global__ __launch_bounds__(256, 2) void head_kernel(NtGeom g, const void* __restrict__ ehat,
                                                             const void* __restrict__ what, const int* __restrict__ labels,
                                                             MarginConst mc, float* __restrict__ part_max,
                                                             float* __restrict__ part_sum, float* __restrict__ ztarget,
                                                             const float* __restrict__ rowmax, const float* __restrict__ rowsum,
                                                             float gscale, const float* __restrict__ upstream,
                                                             void* __restrict__ dt, int ldt, void* __restrict__ dtt, int ldtt,
                                                             int mtiles, int ntiles) {
    typedef NtTile<T, 2, 2> Tile;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lin = xcd_remap(blockIdx.x, gridDim.x);
    // mtile fastest: the (few) row tiles that share one weight tile run together on one XCD
    const int mtile = (int)(lin % (uint32_t)mtiles), ntile = (int)(lin / (uint32_t)mtiles);
    NtMainloop<T, 2, 2> ml;
    ml.run(g, ehat, what, smem, mtile, ntile, 0, g.ksteps);

    const int lane = lane_id(), wave = wave_id();
    const int wm = wave >> 1, wn = wave & 1;
    const int fi = lane & 15, fg = lane >> 4;
    const int m0 = mtile * Tile::BM + wm * 64, n0 = ntile * Tile::BN + wn * 64;
    const int group = ntile * 2 + wn;                   // 64-class column group id
    if (!FWD && upstream) gscale *= upstream[0];        // d(loss)/d(loss) stays on the device: no host sync

#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = m0 + mt * 16 + fi;
        const bool mrow = m < g.M;
        const int lab = mrow ? labels[m] : -1;
        float gm = 0.f, gs = 1.f;
        if (!FWD && mrow) { gm = rowmax[m]; gs = 1.f / rowsum[m]; }
        float z[4][4];
        float vmax = -INFINITY;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int cls = n0 + nt * 16 + 4 * fg + e;
                const float raw = ml.acc[nt][mt][e];
                float t = fminf(fmaxf(raw, -1.f), 1.f);
                float slope = 1.f;
                if (cls == lab) {
                    const float sin_t = sqrtf(1.f - t * t);
                    if (t > mc.theta) { slope = mc.cos_m + t * mc.sin_m / sin_t; t = t * mc.cos_m - sin_t * mc.sin_m; }
                    else t = t - mc.sinmm;
                }
                const float zz = t * mc.s;
                if (FWD) {
                    z[nt][e] = (cls < g.Nout) ? zz : -INFINITY;
                    vmax = fmaxf(vmax, z[nt][e]);
                    if (cls == lab && mrow) ztarget[m] = zz;
                } else {
                    float d = 0.f;
                    if (cls < g.Nout && mrow) {
                        const float p = __expf(zz - gm) * gs;
                        const bool inside = raw >= -1.f && raw <= 1.f;
                        d = inside ? (p - (cls == lab ? 1.f : 0.f)) * gscale * mc.s * slope : 0.f;
                    }
                    ml.acc[nt][mt][e] = d;
                }
            }
        if (FWD) {
            vmax = lane_max_bit5(lane_max_bit4(vmax));
            float vs = 0.f;
            const float vref = vmax == -INFINITY ? 0.f : vmax;                  // a fully masked group: sum 0, no NaN
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int e = 0; e < 4; ++e) vs += __expf(z[nt][e] - vref);      // exp(-inf) = 0 for masked classes
            vs = lane_sum_bit5(lane_sum_bit4(vs));
            if (fg == 0 && mrow) {
                part_max[(size_t)group * g.M + m] = vmax;
                part_sum[(size_t)group * g.M + m] = vs;
            }
        }
    }
    if (!FWD) {
        constexpr int P = Tile::template stage_pitch<T>();
        constexpr int EPV = 16 / (int)sizeof(T), LPR = 64 / EPV, RPI = 64 / LPR;
        const char* mine = ml.template stage_out<T>(smem);
        const int chunk = lane % LPR, rsub = lane / LPR;
        const int n = n0 + chunk * EPV;
        T* o = reinterpret_cast<T*>(dt);
        for (int it = 0; it < 64 / RPI; ++it) {
            const int row = it * RPI + rsub, m = m0 + row;
            if (m < g.M && n < ldt)         // columns in [Nout, ldt) hold zeros (dT pitch padding)
                *reinterpret_cast<Vec16<T>*>(o + (size_t)m * ldt + n) = *reinterpret_cast<const Vec16<T>*>(mine + row * P + chunk * 16);
        }
        if (dtt) {
            // the same tile transposed, dTt[class][sample] (pitch ldtt, multiple of the 16-byte vector): the embedding-gradient GEMM
            // contracts over classes and wants class-major rows -- written here, a separate transpose pass (125 MB read + written at
            // 122 000 classes) is not needed.  Lane = class row of the tile, eight 16-byte vectors of samples each.
            T* ot = reinterpret_cast<T*>(dtt);
            const int cls = n0 + lane;
            if (cls < g.Nout) {
#pragma unroll
                for (int v = 0; v < 64 / EPV; ++v) {
                    Vec16<T> tv;
#pragma unroll
                    for (int e = 0; e < EPV; ++e) tv.v[e] = *reinterpret_cast<const T*>(mine + (v * EPV + e) * P + lane * (int)sizeof(T));
                    const int mcol = m0 + v * EPV;
                    if (mcol < ldtt) *reinterpret_cast<Vec16<T>*>(ot + (size_t)cls * ldtt + mcol) = tv;      // rows past M hold zeros (masked above)
                }
            }
        }
    }
}

// rowmax[m] = max_g part_max[g][m]; rowsum[m] = sum_g part_sum[g][m] * exp(part_max[g][m] - rowmax[m])
// block = 16 rows x 16 group-lanes; each lane folds its share of the column groups with an online max/sum merge,
// the 16 partial (max, sum) pairs of a row are merged through LDS.
__global__ __launch_bounds__(256) void head_rowreduce_kernel(const float* __restrict__ part_max, const float* __restrict__ part_sum,
                                                             int ngroups, int N, float* __restrict__ rowmax, float* __restrict__ rowsum) {
    __shared__ float smax[16][17], ssum[16][17];
    const int ml = threadIdx.x & 15, gl = threadIdx.x >> 4;
    const int m = blockIdx.x * 16 + ml;
    float mx = -INFINITY, s = 0.f;
    if (m < N) {
        for (int gq = gl; gq < ngroups; gq += 16) {
            const float pm = part_max[(size_t)gq * N + m], ps = part_sum[(size_t)gq * N + m];
            if (pm > mx) { s = s * __expf(mx - pm) + ps; mx = pm; }
            else if (pm > -INFINITY) s += ps * __expf(pm - mx);
        }
    }
    smax[gl][ml] = mx; ssum[gl][ml] = s;
    __syncthreads();
    if (gl == 0 && m < N) {
        float M = -INFINITY;
#pragma unroll
        for (int k = 0; k < 16; ++k) M = fmaxf(M, smax[k][ml]);
        float S = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) if (smax[k][ml] > -INFINITY) S += ssum[k][ml] * __expf(smax[k][ml] - M);
        rowmax[m] = M; rowsum[m] = S;
    }
}

// rowsum *= exp(local_max - global_max)      (before the cross-rank SUM)
__global__ void head_rescale_kernel(float* __restrict__ rowsum, const float* __restrict__ local_max,
                                    const float* __restrict__ global_max, int N) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < N) rowsum[m] *= __expf(local_max[m] - global_max[m]);
}

// q[m] = owned ? exp(z_target - M)/S : 0
__global__ void head_target_prob_kernel(const float* __restrict__ ztarget, const int* __restrict__ labels,
                                        const float* __restrict__ rowmax, const float* __restrict__ rowsum,
                                        float* __restrict__ q, int N) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < N) q[m] = labels[m] >= 0 ? __expf(ztarget[m] - rowmax[m]) / rowsum[m] : 0.f;
}

// Cross-shard merge of the per-row softmax statistics in ONE exchange (reference nets/PartialFC.py:448, :453, :459 issues an
// all-reduce MAX and two all-reduce SUMs).  Every rank packs {local max, local sum-exp, target logit or -inf} per row; the
// packed [N][3] blocks of all ranks are all-gathered and every rank folds them in rank order (same result on every rank).
__global__ void head_pack_stats_kernel(const float* __restrict__ ztarget, const int* __restrict__ labels,
                                       const float* __restrict__ rowmax, const float* __restrict__ rowsum,
                                       float* __restrict__ packed, int N) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < N) {
        packed[3 * m + 0] = rowmax[m];
        packed[3 * m + 1] = rowsum[m];
        packed[3 * m + 2] = labels[m] >= 0 ? ztarget[m] : -INFINITY;
    }
}

// gathered [ws][N][3] -> global max M, global sum S = sum_r s_r exp(m_r - M), q = exp(z_owner - M) / S (0: nobody owns the row)
__global__ void head_merge_stats_kernel(const float* __restrict__ gathered, int ws, int N, float* __restrict__ gmax,
                                        float* __restrict__ gsum, float* __restrict__ q) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= N) return;
    float M = -INFINITY;
    for (int r = 0; r < ws; ++r) M = fmaxf(M, gathered[((size_t)r * N + m) * 3]);
    float S = 0.f, num = 0.f;
    for (int r = 0; r < ws; ++r) {
        const float* g = gathered + ((size_t)r * N + m) * 3;
        S += g[1] * __expf(g[0] - M);
        if (g[2] > -INFINITY) num += __expf(g[2] - M);
    }
    gmax[m] = M; gsum[m] = S; q[m] = num / S;
}

// loss = -mean(log(max(q, 1e-30)))          single block
__global__ void head_loss_kernel(const float* __restrict__ q, int N, float* __restrict__ loss) {
    __shared__ float red[256];
    float acc = 0.f;
    for (int m = threadIdx.x; m < N; m += 256) acc += logf(fmaxf(q[m], 1e-30f));
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) { if (threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d]; __syncthreads(); }
    if (threadIdx.x == 0) loss[0] = -red[0] / (float)N;
}

template <typename T, bool FWD>
static int head_launch(const NtGeom& g, const void* ehat, const void* what, const int* labels, const MarginConst& mc,
                       float* pmax, float* psum, float* zt, const float* rmax, const float* rsum, float gscale,
                       const float* upstream, void* dt, int ldt, void* dtt, int ldtt, hipStream_t stream) {
    typedef NtTile<T, 2, 2> Tile;
    const int mtiles = (g.M + Tile::BM - 1) / Tile::BM, ntiles = (g.Nout + Tile::BN - 1) / Tile::BN;
    const int lds = Tile::template lds_bytes<T>();
    auto kern = head_kernel<T, FWD>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            set_error("head: cannot raise dynamic LDS to %d bytes", lds);
            return FRHIP_ELAUNCH;
        }
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(mtiles * ntiles), dim3(256), lds, stream, g, ehat, what, labels, mc, pmax, psum,
                       zt, rmax, rsum, gscale, upstream, dt, ldt, dtt, ldtt, mtiles, ntiles);
    return check_launch("head");
}

static int head_geom(NtGeom& g, int dtype, int n, int cl, int d, const char* who) {
    const int es = dtype == FRHIP_DT_BF16 ? 2 : 4, bke = NT_ROWB / es;
    if ((dtype != FRHIP_DT_BF16 && dtype != FRHIP_DT_F32) || n <= 0 || cl <= 0 || d <= 0 || (d % bke)) {
        set_error("%s: unsupported shape/dtype (n=%d classes=%d d=%d dtype=%d; d must be a multiple of %d)", who, n, cl, d, dtype, bke);
        return FRHIP_EINVAL;
    }
    if (1LL * cl * d * es > 0x7fffffffLL || 1LL * n * d * es > 0x7fffffffLL) { set_error("%s: operand exceeds 2 GiB", who); return FRHIP_EINVAL; }
    g.H = 1; g.W = 1; g.C = d; g.Ho = 1; g.Wo = 1; g.R = 1; g.S = 1; g.stride = 1; g.pad = 0; g.mode = 0;
    g.M = n; g.Nout = cl; g.Ktot = d; g.ksteps = d / bke; g.ksteps_per_split = g.ksteps;
    g.a_bytes = (uint32_t)(1LL * n * d * es); g.b_bytes = (uint32_t)(1LL * cl * d * es);
    g.par_a = -1; g.par_b = -1; g.hc = 0; g.wc = 0; g.par_r0 = 0; g.par_s0 = 0;
    return FRHIP_OK;
}

static MarginConst margin_const(float s, float m) {
    MarginConst mc;
    const double pi = 3.14159265358979323846;
    mc.s = s; mc.cos_m = (float)cos((double)m); mc.sin_m = (float)sin((double)m);
    mc.theta = (float)cos(pi - (double)m); mc.sinmm = (float)(sin(pi - (double)m) * (double)m);
    return mc;
}

}  // namespace frhip

using namespace frhip;

extern "C" int frhip_l2norm_rows(int dtype, const float* x, void* xhat, float* norms, int rows, int d, float eps,
                                 hipStream_t stream) {
    if (d % 4) { set_error("frhip_l2norm_rows: d must be a multiple of 4"); return FRHIP_EINVAL; }
    if (dtype == FRHIP_DT_BF16 && d == 512) hipLaunchKernelGGL(l2norm_rows512_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, x, (bf16_t*)xhat, norms, rows, eps);
    else if (dtype == FRHIP_DT_BF16) hipLaunchKernelGGL(l2norm_rows_kernel<bf16_t>, dim3((rows + 3) / 4), dim3(256), 0, stream, x, (bf16_t*)xhat, norms, rows, d, eps);
    else if (dtype == FRHIP_DT_F32) hipLaunchKernelGGL(l2norm_rows_kernel<float>, dim3((rows + 3) / 4), dim3(256), 0, stream, x, (float*)xhat, norms, rows, d, eps);
    else { set_error("frhip_l2norm_rows: bad dtype %d", dtype); return FRHIP_EINVAL; }
    return check_launch("frhip_l2norm_rows");
}

extern "C" int frhip_l2norm_bwd(int dtype, const float* dxhat, const void* xhat, const float* norms, float* dx,
                                int rows, int d, float out_scale, hipStream_t stream) {
    if (dtype == FRHIP_DT_BF16 && d == 512) hipLaunchKernelGGL(l2norm_bwd512_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, dxhat, (const bf16_t*)xhat, norms, dx, rows, out_scale);
    else if (dtype == FRHIP_DT_BF16) hipLaunchKernelGGL(l2norm_bwd_kernel<bf16_t>, dim3((rows + 3) / 4), dim3(256), 0, stream, dxhat, (const bf16_t*)xhat, norms, dx, rows, d, out_scale);
    else if (dtype == FRHIP_DT_F32) hipLaunchKernelGGL(l2norm_bwd_kernel<float>, dim3((rows + 3) / 4), dim3(256), 0, stream, dxhat, (const float*)xhat, norms, dx, rows, d, out_scale);
    else { set_error("frhip_l2norm_bwd: bad dtype %d", dtype); return FRHIP_EINVAL; }
    return check_launch("frhip_l2norm_bwd");
}

extern "C" int frhip_head_groups(int num_classes) { return ((num_classes + 127) / 128) * 2; }

extern "C" int frhip_head_fwd(int dtype, const void* ehat, const void* what, const int* labels, int n, int classes,
                              int d, float s, float m, float* part_max, float* part_sum, float* ztarget,
                              float* rowmax, float* rowsum, hipStream_t stream) {
    NtGeom g;
    int rc = head_geom(g, dtype, n, classes, d, "frhip_head_fwd");
    if (rc) return rc;
    const MarginConst mc = margin_const(s, m);
    if (dtype == FRHIP_DT_BF16) rc = head_launch<bf16_t, true>(g, ehat, what, labels, mc, part_max, part_sum, ztarget, nullptr, nullptr, 0.f, nullptr, nullptr, 0, nullptr, 0, stream);
    else rc = head_launch<float, true>(g, ehat, what, labels, mc, part_max, part_sum, ztarget, nullptr, nullptr, 0.f, nullptr, nullptr, 0, nullptr, 0, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(head_rowreduce_kernel, dim3((n + 15) / 16), dim3(256), 0, stream, part_max, part_sum,
                       frhip_head_groups(classes), n, rowmax, rowsum);
    return check_launch("frhip_head_fwd/rowreduce");
}

extern "C" int frhip_head_rescale(float* rowsum, const float* local_max, const float* global_max, int n, hipStream_t stream) {
    hipLaunchKernelGGL(head_rescale_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, rowsum, local_max, global_max, n);
    return check_launch("frhip_head_rescale");
}

extern "C" int frhip_head_target_prob(const float* ztarget, const int* labels, const float* rowmax, const float* rowsum,
                                      float* q, int n, hipStream_t stream) {
    hipLaunchKernelGGL(head_target_prob_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, ztarget, labels, rowmax, rowsum, q, n);
    return check_launch("frhip_head_target_prob");
}

extern "C" int frhip_head_pack_stats(const float* ztarget, const int* labels, const float* rowmax, const float* rowsum,
                                     float* packed, int n, hipStream_t stream) {
    hipLaunchKernelGGL(head_pack_stats_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, ztarget, labels, rowmax, rowsum, packed, n);
    return check_launch("frhip_head_pack_stats");
}

extern "C" int frhip_head_merge_stats(const float* gathered, int world_size, int n, float* rowmax, float* rowsum, float* q,
                                      hipStream_t stream) {
    if (world_size <= 0 || n <= 0) { set_error("frhip_head_merge_stats: bad shape (ws=%d n=%d)", world_size, n); return FRHIP_EINVAL; }
    hipLaunchKernelGGL(head_merge_stats_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, gathered, world_size, n, rowmax, rowsum, q);
    return check_launch("frhip_head_merge_stats");
}

extern "C" int frhip_head_loss(const float* q, int n, float* loss, hipStream_t stream) {
    hipLaunchKernelGGL(head_loss_kernel, dim3(1), dim3(256), 0, stream, q, n, loss);
    return check_launch("frhip_head_loss");
}

extern "C" int frhip_head_bwd_dt(int dtype, const void* ehat, const void* what, const int* labels, int n, int classes,
                                 int d, float s, float m, const float* rowmax, const float* rowsum, float gscale,
                                 const float* upstream, void* dt, int ldt, void* dtt, int ldtt, hipStream_t stream) {
    NtGeom g;
    int rc = head_geom(g, dtype, n, classes, d, "frhip_head_bwd_dt");
    if (rc) return rc;
    const int epv = dtype == FRHIP_DT_BF16 ? 8 : 4;
    if (ldt < classes || (ldt % epv)) { set_error("frhip_head_bwd_dt: bad dT pitch %d", ldt); return FRHIP_EINVAL; }
    if (dtt && (ldtt < n || (ldtt % epv))) { set_error("frhip_head_bwd_dt: bad transposed pitch %d", ldtt); return FRHIP_EINVAL; }
    const MarginConst mc = margin_const(s, m);
    if (dtype == FRHIP_DT_BF16) return head_launch<bf16_t, false>(g, ehat, what, labels, mc, nullptr, nullptr, nullptr, rowmax, rowsum, gscale, upstream, dt, ldt, dtt, ldtt, stream);
    return head_launch<float, false>(g, ehat, what, labels, mc, nullptr, nullptr, nullptr, rowmax, rowsum, gscale, upstream, dt, ldt, dtt, ldtt, stream);
}
