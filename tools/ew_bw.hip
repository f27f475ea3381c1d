// Stand-alone bandwidth study of the BatchNorm-backward-apply access pattern (two bf16 tensors in, one out, per-channel coefficients):
//   hipcc -O3 --offload-arch=gfx950 tools/ew_bw.hip -o /tmp/ew_bw && /tmp/ew_bw
// Rotates through enough distinct buffer sets that nothing stays in the 256-MB Infinity Cache between launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <bool NTL, bool NTS> __device__ __forceinline__ void body(const bf16x8* __restrict__ d, const bf16x8* __restrict__ y, bf16x8* __restrict__ o,
                                                                 size_t i, const float* a, const float* b, const float* c) {
    bf16x8 dv, yv;
    if (NTL) { dv = __builtin_nontemporal_load(d + i); yv = __builtin_nontemporal_load(y + i); } else { dv = d[i]; yv = y[i]; }
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float yy = (float)yv[e]; const float de = yy * a[e] + 1.f > 0.f ? (float)dv[e] : 0.f; r[e] = (__bf16)(a[e] * de + b[e] * yy + c[e]); }
    if (NTS) __builtin_nontemporal_store(r, o + i); else o[i] = r;
}

// V0: the product's shape -- thread owns one 16-B channel group, rows strided by the grid
template <bool NTL, bool NTS, int UNROLL>
__global__ __launch_bounds__(256) void v0(const bf16x8* __restrict__ d, const bf16x8* __restrict__ y, bf16x8* __restrict__ o, const float* __restrict__ coef, int rows, int C) {
    const int vpr = C / 8, cg = threadIdx.x % vpr, rl = threadIdx.x / vpr, rlanes = 256 / vpr;
    float a[8], b[8], c[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = coef[cg * 8 + e]; b[e] = coef[C + cg * 8 + e]; c[e] = coef[2 * C + cg * 8 + e]; }
#pragma unroll UNROLL
    for (int r = blockIdx.x * rlanes + rl; r < rows; r += gridDim.x * rlanes) body<NTL, NTS>(d, y, o, (size_t)r * vpr + cg, a, b, c);
}
// V1: contiguous chunk per block (block b streams rows [b*R, (b+1)*R)), U independent rows in flight per thread
template <bool NTL, bool NTS, int U>
__global__ __launch_bounds__(256) void v1(const bf16x8* __restrict__ d, const bf16x8* __restrict__ y, bf16x8* __restrict__ o, const float* __restrict__ coef, int rows, int C, int rows_per_block) {
    const int vpr = C / 8, cg = threadIdx.x % vpr, rl = threadIdx.x / vpr, rlanes = 256 / vpr;
    float a[8], b[8], c[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = coef[cg * 8 + e]; b[e] = coef[C + cg * 8 + e]; c[e] = coef[2 * C + cg * 8 + e]; }
    const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    int r = r0 + rl;
    for (; r + (U - 1) * rlanes < r1; r += U * rlanes) {
        bf16x8 dv[U], yv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = (size_t)(r + u * rlanes) * vpr + cg;
            if (NTL) { dv[u] = __builtin_nontemporal_load(d + i); yv[u] = __builtin_nontemporal_load(y + i); } else { dv[u] = d[i]; yv[u] = y[i]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            bf16x8 rr;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float yy = (float)yv[u][e]; const float de = yy * a[e] + 1.f > 0.f ? (float)dv[u][e] : 0.f; rr[e] = (__bf16)(a[e] * de + b[e] * yy + c[e]); }
            const size_t i = (size_t)(r + u * rlanes) * vpr + cg;
            if (NTS) __builtin_nontemporal_store(rr, o + i); else o[i] = rr;
        }
    }
    for (; r < r1; r += rlanes) body<NTL, NTS>(d, y, o, (size_t)r * vpr + cg, a, b, c);
}

int main() {
    const int B = 512;
    const int shapes[4][2] = {{56, 64}, {28, 128}, {14, 256}, {7, 512}};
    float* coef; CK(hipMalloc(&coef, 3 * 512 * 4));
    std::vector<float> h(3 * 512, 0.5f); CK(hipMemcpy(coef, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (auto& s : shapes) {
        const int rows = B * s[0] * s[0], C = s[1];
        const size_t bytes = (size_t)rows * C * 2;
        const int NSET = (int)(((size_t)1200 << 20) / (3 * bytes)) + 2;          // > 1.2 GB in rotation
        std::vector<void*> D(NSET), Y(NSET), O(NSET);
        for (int i = 0; i < NSET; ++i) { CK(hipMalloc(&D[i], bytes)); CK(hipMalloc(&Y[i], bytes)); CK(hipMalloc(&O[i], bytes)); CK(hipMemset(D[i], 0x3c, bytes)); CK(hipMemset(Y[i], 0x3d, bytes)); }
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto run = [&](const char* name, auto launch) {
            for (int i = 0; i < NSET; ++i) launch(i);
            CK(hipDeviceSynchronize());
            const int reps = 3 * NSET;
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) launch(i % NSET);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("  %-34s %7.1f us  %5.2f TB/s\n", name, ms / reps * 1e3, 3.0 * bytes / (ms / reps * 1e-3) / 1e12);
        };
        printf("h=%d C=%d  tensor %.1f MB x3, %d buffer sets\n", s[0], C, bytes / 1e6, NSET);
        const int vpr = C / 8, rlanes = 256 / vpr;
        int blocks0 = (rows + rlanes * 16 - 1) / (rlanes * 16); if (blocks0 > 2048) blocks0 = 2048;
#define ARGS(i) (const bf16x8*)D[i], (const bf16x8*)Y[i], (bf16x8*)O[i], coef, rows, C
        run("v0 product (nt loads, u2, 2048)", [&](int i) { hipLaunchKernelGGL((v0<true, false, 2>), dim3(blocks0), dim3(256), 0, 0, ARGS(i)); });
        run("v0 plain loads", [&](int i) { hipLaunchKernelGGL((v0<false, false, 2>), dim3(blocks0), dim3(256), 0, 0, ARGS(i)); });
        run("v0 nt loads + nt stores", [&](int i) { hipLaunchKernelGGL((v0<true, true, 2>), dim3(blocks0), dim3(256), 0, 0, ARGS(i)); });
        run("v0 unroll 4", [&](int i) { hipLaunchKernelGGL((v0<true, false, 4>), dim3(blocks0), dim3(256), 0, 0, ARGS(i)); });
        for (int nb : {512, 1024, 4096, 8192}) {
            char nm[64]; snprintf(nm, 64, "v0 nt loads, %d blocks", nb);
            run(nm, [&](int i) { hipLaunchKernelGGL((v0<true, false, 2>), dim3(nb), dim3(256), 0, 0, ARGS(i)); });
        }
        for (int nb : {1024, 2048, 4096}) {
            const int rpb = ((rows + nb - 1) / nb + rlanes - 1) / rlanes * rlanes;
            char nm[64];
            snprintf(nm, 64, "v1 contiguous U2, %d blocks", nb);
            run(nm, [&](int i) { hipLaunchKernelGGL((v1<true, false, 2>), dim3(nb), dim3(256), 0, 0, ARGS(i), rpb); });
            snprintf(nm, 64, "v1 contiguous U4, %d blocks", nb);
            run(nm, [&](int i) { hipLaunchKernelGGL((v1<true, false, 4>), dim3(nb), dim3(256), 0, 0, ARGS(i), rpb); });
            snprintf(nm, 64, "v1 contiguous U4 nt-store, %d blocks", nb);
            run(nm, [&](int i) { hipLaunchKernelGGL((v1<true, true, 4>), dim3(nb), dim3(256), 0, 0, ARGS(i), rpb); });
            snprintf(nm, 64, "v1 contiguous U4 plain, %d blocks", nb);
            run(nm, [&](int i) { hipLaunchKernelGGL((v1<false, false, 4>), dim3(nb), dim3(256), 0, 0, ARGS(i), rpb); });
        }
        for (int i = 0; i < NSET; ++i) { CK(hipFree(D[i])); CK(hipFree(Y[i])); CK(hipFree(O[i])); }
    }
    return 0;
}
