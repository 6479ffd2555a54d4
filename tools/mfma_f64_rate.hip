// fp64 issue-rate microbenchmark for gfx950 (MI355X): dependent-free chains of v_fma_f64 and of v_mfma_f64_16x16x4_f64, every CU busy.
// The MI355X guide (MI355X_MICROARCH.md) has no fp64 row; the datasheet figure is 78.6 TFLOP/s for both the vector and the matrix pipe.  bench.py prices the
// optimisers' kernels against the datasheet figure and reports what this tool measures beside it (tools/mfma_f64_rate.py prints the JSON).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ __launch_bounds__(256) void k_fma_f64(double* out, int iters, double b, double c) {
#pragma clang fp contract(fast)
    double a[CHAINS];
#pragma unroll
    for (int i = 0; i < CHAINS; i++) a[i] = (double)(threadIdx.x + i) * 1e-9;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i++) a[i] = __builtin_fma(a[i], b, c);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; i++) s += a[i];
    if (s == 12345.678) out[0] = s;   // never true: keeps the chains alive
}

template <int CHAINS>
__global__ __launch_bounds__(256) void k_mfma_f64(double* out, int iters, double av, double bv) {
    v4f64 acc[CHAINS];
#pragma unroll
    for (int i = 0; i < CHAINS; i++) acc[i] = (v4f64){0.0, 0.0, 0.0, 0.0};
    const double a = av + threadIdx.x * 1e-12, b = bv;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[0] = s;
}

// kind 0: v_fma_f64, 1: v_mfma_f64_16x16x4_f64.  blocks_per_cu x 256 threads = blocks_per_cu waves per SIMD.  Returns TFLOP/s (best of `reps` launches), < 0 on error.
extern "C" double oslam_tool_f64_rate(int kind, int blocks_per_cu, int iters, int reps, int device) {
    if (hipSetDevice(device) != hipSuccess) return -1;
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, device) != hipSuccess) return -1;
    const int cus = pr.multiProcessorCount;
    double* d = nullptr;
    if (hipMalloc((void**)&d, 64) != hipSuccess) return -1;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    constexpr int CH = 8;
    const dim3 grid(cus * blocks_per_cu), block(256);
    double best = 0;
    for (int r = 0; r < reps + 1; r++) {
        (void)hipEventRecord(e0, 0);
        if (kind == 0) hipLaunchKernelGGL(k_fma_f64<CH>, grid, block, 0, 0, d, iters, 0.999999, 1e-7);
        else hipLaunchKernelGGL(k_mfma_f64<CH>, grid, block, 0, 0, d, iters, 1e-3, 1e-3);
        (void)hipEventRecord(e1, 0);
        if (hipEventSynchronize(e1) != hipSuccess) return -1;
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (r == 0) continue;   // warm-up launch
        const double waves = (double)cus * blocks_per_cu * 4;
        const double flop = kind == 0 ? waves * 64.0 * 2.0 * CH * (double)iters : waves * (16.0 * 16.0 * 4.0 * 2.0) * CH * (double)iters;
        const double tf = flop / (ms * 1e-3) / 1e12;
        if (tf > best) best = tf;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(d);
    return best;
}
