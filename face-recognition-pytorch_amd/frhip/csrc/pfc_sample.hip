// PartialFC negative sampling in ONE launch (reference /root/reference/nets/PartialFC.py:108-121, :188-193):
//   positive = unique(labels owned by this shard);  perm = rand(num_local);  perm[positive] = 2;
//   index = sort(topk(perm, num_sample).indices);   labels[owned] = searchsorted(index, labels[owned])
// i.e. the sampled rows are all positives plus the R = num_sample - #positives non-positive rows with the largest draws, and
// a label becomes the position of its class in the sorted row list.  The torch formulation is ~25 small launches (mask, scatter,
// top-k, sort, searchsorted, ...); here one workgroup does it:
//   1. positives -> LDS bitset;  P = popcount
//   2. the R-th largest draw among the non-positives by a 4 x 8-bit radix select on the float bits (draws are in [0, 1): the bit
//      pattern orders like the value) -> threshold T and how many rows with draw == T are still needed (ties: lowest row ids)
//   3. selected-row bitset, per-word exclusive prefix -> index[] ascending, label positions by popcount
// P > num_sample (the reference's `index = positive` branch, a different output length) is only REPORTED (n_positive): the
// caller takes its other route.  num_local <= 393 216 rows per rank (LDS); the reference's 86 690 / 122 000 identities fit on one rank.
#include "common.h"
#include "frhip.h"

namespace frhip {

constexpr int PS_THREADS = 1024;
constexpr int PS_MAX_LOCAL = 393216;              // 3 bitset / prefix words of LDS per 32 rows: 147 KB at this size

__device__ __forceinline__ int ps_block_scan_excl(int v, int* scratch, int& total) {      // exclusive scan over the 1024 threads
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
    __syncthreads();
    if (lane == 63) scratch[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < PS_THREADS / 64; ++w) { const int s = scratch[w]; if (w < wave) base += s; tot += s; }
    total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(PS_THREADS) void pfc_sample_kernel(const int64_t* __restrict__ labels, int n, long long class_start,
                                                                int num_local, const float* __restrict__ u, int num_sample,
                                                                int64_t* __restrict__ index_out, int* __restrict__ rel_out,
                                                                int64_t* __restrict__ n_positive) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int words = (num_local + 31) >> 5;
    uint32_t* pos = reinterpret_cast<uint32_t*>(smem);           // [words] positives
    uint32_t* sel = pos + words;                                  // [words] selected rows
    int* wpre = reinterpret_cast<int*>(sel + words);              // [words] selected rows in front of the word
    int* hist = wpre + words;                                     // [256]
    int* scratch = hist + 256;                                    // [32]
    const int tid = threadIdx.x;
    for (int w = tid; w < words; w += PS_THREADS) pos[w] = 0u;
    __syncthreads();
    for (int i = tid; i < n; i += PS_THREADS) {
        const long long l = labels[i] - class_start;
        if (l >= 0 && l < num_local) atomicOr(&pos[l >> 5], 1u << (l & 31));
    }
    __syncthreads();
    int cnt = 0;
    for (int w = tid; w < words; w += PS_THREADS) cnt += __popc(pos[w]);
    int P;
    ps_block_scan_excl(cnt, scratch, P);
    if (tid == 0) *n_positive = P;
    if (P > num_sample) {                        // the other branch of the reference: only reported.  index_out still gets VALID rows
        for (int i = tid; i < n; i += PS_THREADS) rel_out[i] = -1;              // (0 .. num_sample-1): an optimistic caller gathers them
        for (int i = tid; i < num_sample; i += PS_THREADS) index_out[i] = i;    // before it has seen the count, then throws them away
        return;
    }
    // ---- radix select: the R largest draws among the non-positives
    int need = num_sample - P;                   // rows still to take from the non-positives
    uint32_t prefix = 0, pmask = 0;              // key bits fixed so far
    for (int shift = 24; shift >= 0 && need > 0; shift -= 8) {
        for (int b = tid; b < 256; b += PS_THREADS) hist[b] = 0;
        __syncthreads();
        for (int i = tid; i < num_local; i += PS_THREADS) {
            if ((pos[i >> 5] >> (i & 31)) & 1u) continue;
            const uint32_t key = __float_as_uint(u[i]);
            if ((key & pmask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1);
        }
        __syncthreads();
        // the bin (from the top) in which the cumulative count reaches `need`
        int bin = 0, above = 0;
        if (tid == 0) {
            int c = 0, b = 255;
            for (; b > 0; --b) { if (c + hist[b] >= need) break; c += hist[b]; }
            scratch[0] = b; scratch[1] = c;
        }
        __syncthreads();
        bin = scratch[0]; above = scratch[1];
        __syncthreads();
        need -= above;                           // rows with a larger digit are all taken
        prefix |= (uint32_t)bin << shift; pmask |= 255u << shift;
    }
    // now: take every non-positive with key > prefix (all 32 bits fixed), and `need` of those with key == prefix (lowest ids first)
    const uint32_t T = prefix;
    const bool none = (num_sample - P) == 0;
    const int wpt = (words + PS_THREADS - 1) / PS_THREADS;        // consecutive words per thread
    const int w0 = tid * wpt, w1 = min(words, w0 + wpt);
    int eq_cnt = 0;
    for (int w = w0; w < w1; ++w) {
        uint32_t gt = 0, eq = 0;
        for (int b = 0; b < 32; ++b) {
            const int i = (w << 5) + b;
            if (i >= num_local) break;
            if ((pos[w] >> b) & 1u) continue;
            const uint32_t key = __float_as_uint(u[i]);
            if (!none && key > T) gt |= 1u << b;
            else if (!none && key == T) eq |= 1u << b;
        }
        sel[w] = pos[w] | gt;
        wpre[w] = (int)eq;                       // parked: this word's tie candidates
        eq_cnt += __popc(eq);
    }
    int eq_total;
    int eq_rank = ps_block_scan_excl(eq_cnt, scratch, eq_total);
    int sel_cnt = 0;
    for (int w = w0; w < w1; ++w) {
        uint32_t eq = (uint32_t)wpre[w], take = 0;
        while (eq && eq_rank < need) { const int b = __ffs(eq) - 1; take |= 1u << b; eq &= eq - 1; ++eq_rank; }
        eq_rank += __popc(eq);                   // candidates not taken still advance the rank
        sel[w] |= take;
        sel_cnt += __popc(sel[w]);
    }
    int sel_total;
    int base = ps_block_scan_excl(sel_cnt, scratch, sel_total);
    for (int w = w0; w < w1; ++w) {
        wpre[w] = base;
        uint32_t s = sel[w];
        while (s) { const int b = __ffs(s) - 1; index_out[base++] = ((int64_t)w << 5) + b; s &= s - 1; }
    }
    __syncthreads();
    for (int i = tid; i < n; i += PS_THREADS) {
        const long long l = labels[i] - class_start;
        int r = -1;
        if (l >= 0 && l < num_local) r = wpre[l >> 5] + __popc(sel[l >> 5] & ((1u << (l & 31)) - 1u));
        rel_out[i] = r;
    }
}

}  // namespace frhip

using namespace frhip;

extern "C" int frhip_pfc_sample_max_local(void) { return PS_MAX_LOCAL; }

extern "C" int frhip_pfc_sample(const int64_t* labels, int n, long long class_start, int num_local, const float* u, int num_sample,
                                int64_t* index_out, int* rel_out, int64_t* n_positive, hipStream_t stream) {
    if (!labels || !u || !index_out || !rel_out || !n_positive || n <= 0 || num_local <= 0 || num_local > PS_MAX_LOCAL ||
        num_sample <= 0 || num_sample > num_local) {
        set_error("frhip_pfc_sample: bad arguments (n=%d num_local=%d num_sample=%d; num_local <= %d)", n, num_local, num_sample, PS_MAX_LOCAL);
        return FRHIP_EINVAL;
    }
    const int words = (num_local + 31) / 32;
    const int lds = words * 12 + 256 * 4 + 32 * 4;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(pfc_sample_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            set_error("frhip_pfc_sample: cannot raise dynamic LDS");
            return FRHIP_ELAUNCH;
        }
        attr_done = true;
    }
    hipLaunchKernelGGL(pfc_sample_kernel, dim3(1), dim3(PS_THREADS), lds, stream, labels, n, class_start, num_local, u, num_sample,
                       index_out, rel_out, n_positive);
    return check_launch("frhip_pfc_sample");
}
