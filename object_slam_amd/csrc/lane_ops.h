// Lane exchanges of a 64-wide wavefront in the vector ALU (gfx950), for the reductions of the optimiser kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace oslam {

// ---- wavefront reductions without the LDS crossbar (round 5, second pass) ----
// __shfl_xor of a double is two ds_bpermute_b32: the 27 sums of a keyframe block of k_w_lin were 324 of them per wavefront, sixteen wavefronts of a CU queueing
// at one LDS pipe.  gfx950 can exchange lanes in the vector ALU: v_permlane32_swap / v_permlane16_swap (halves / odd-even rows of 16 between two registers),
// DPP row_ror:8 and quad_perm; only lane ^ 4 still goes through ds_swizzle.  Every function below forms the sums of the xor butterfly d = 32, 16, 8, 4, 2, 1
// with the same operand pairs (a + b against b + a at most): bit-identical results.
typedef unsigned int oslam_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double f64_from(unsigned lo, unsigned hi) { return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)); }
__device__ __forceinline__ unsigned f64_lo(double v) { return (unsigned)(unsigned long long)__double_as_longlong(v); }
__device__ __forceinline__ unsigned f64_hi(double v) { return (unsigned)((unsigned long long)__double_as_longlong(v) >> 32); }
// a's upper 32 lanes <-> b's lower 32 lanes
__device__ __forceinline__ void swap32_f64(double& a, double& b) {
    const oslam_u2 l = __builtin_amdgcn_permlane32_swap(f64_lo(a), f64_lo(b), false, false);
    const oslam_u2 h = __builtin_amdgcn_permlane32_swap(f64_hi(a), f64_hi(b), false, false);
    a = f64_from(l[0], h[0]); b = f64_from(l[1], h[1]);
}
// a's rows 1, 3 (of 16 lanes) <-> b's rows 0, 2
__device__ __forceinline__ void swap16_f64(double& a, double& b) {
    const oslam_u2 l = __builtin_amdgcn_permlane16_swap(f64_lo(a), f64_lo(b), false, false);
    const oslam_u2 h = __builtin_amdgcn_permlane16_swap(f64_hi(a), f64_hi(b), false, false);
    a = f64_from(l[0], h[0]); b = f64_from(l[1], h[1]);
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {   // (all lanes valid for the controls used here: quad_perm, row_ror)
    return f64_from((unsigned)__builtin_amdgcn_mov_dpp((int)f64_lo(v), CTRL, 0xf, 0xf, true), (unsigned)__builtin_amdgcn_mov_dpp((int)f64_hi(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ double lane_xor8(double v) { return dpp_mov_f64<0x128>(v); }   // row_ror:8
__device__ __forceinline__ double lane_xor4(double v) {                                     // swizzle(SWAP, 4)
    return f64_from((unsigned)__builtin_amdgcn_ds_swizzle((int)f64_lo(v), 0x101F), (unsigned)__builtin_amdgcn_ds_swizzle((int)f64_hi(v), 0x101F));
}
__device__ __forceinline__ double lane_xor2(double v) { return dpp_mov_f64<0x4E>(v); }    // quad_perm [2,3,0,1]
__device__ __forceinline__ double lane_xor1(double v) { return dpp_mov_f64<0xB1>(v); }    // quad_perm [1,0,3,2]
// the last four butterfly steps (inside a row of 16 lanes)
__device__ __forceinline__ double row16_sum(double v) { v += lane_xor8(v); v += lane_xor4(v); v += lane_xor2(v); v += lane_xor1(v); return v; }

// ---- 32-bit integers (sums are exact: any order) ----
__device__ __forceinline__ void swap32_i32(int& a, int& b) {
    const oslam_u2 r = __builtin_amdgcn_permlane32_swap((unsigned)a, (unsigned)b, false, false);
    a = (int)r[0]; b = (int)r[1];
}
__device__ __forceinline__ void swap16_i32(int& a, int& b) {
    const oslam_u2 r = __builtin_amdgcn_permlane16_swap((unsigned)a, (unsigned)b, false, false);
    a = (int)r[0]; b = (int)r[1];
}
__device__ __forceinline__ int row16_sum_i32(int v) {
    v += __builtin_amdgcn_mov_dpp(v, 0x128, 0xf, 0xf, true);    // row_ror:8
    v += __builtin_amdgcn_ds_swizzle(v, 0x101F);                // lane ^ 4
    v += __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true);     // lane ^ 2
    v += __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true);     // lane ^ 1
    return v;
}
// Sums over the wavefront of N ints at once (the paired first steps of wsum_rows): out[q], in every lane of row (b5, b4), = the sum of value 4 q + 2 b4 + b5;
// wave_sums_get(out, k) hands value k to every lane.
template <int N>
__device__ __forceinline__ void wave_sums_i32(const int (&in)[N], int (&out)[(N + 3) / 4]) {
    constexpr int N2 = (N + 1) / 2, N4 = (N + 3) / 4;
    int r1[2 * N4];
#pragma unroll
    for (int q = 0; q < 2 * N4; q++) {
        int a = 2 * q < N ? in[2 * q] : 0, b = 2 * q + 1 < N ? in[2 * q + 1] : 0;
        if (q < N2) { swap32_i32(a, b); r1[q] = a + b; } else r1[q] = 0;
    }
#pragma unroll
    for (int q = 0; q < N4; q++) {
        int a = r1[2 * q], b = r1[2 * q + 1];
        swap16_i32(a, b);
        out[q] = row16_sum_i32(a + b);
    }
}
template <int N4>
__device__ __forceinline__ int wave_sums_get(const int (&out)[N4], int k) {   // k: compile-time constant after unrolling
    return __builtin_amdgcn_readlane(out[k >> 2], 32 * (k & 1) + 16 * ((k >> 1) & 1));
}

// sum over the wavefront, every lane gets it: the xor butterfly d = 32, 16, 8, 4, 2, 1 (same operand pairs as the __shfl_xor loop it replaces)
__device__ __forceinline__ double wave_sum_xor(double v) {
    double a = v, b = v;
    swap32_f64(a, b);   // a = [lower half | lower half], b = [upper half | upper half]
    v = a + b;
    a = v; b = v;
    swap16_f64(a, b);   // a = [row 0, row 0, row 2, row 2], b = [row 1, row 1, row 3, row 3]
    v = a + b;
    return row16_sum(v);
}

}   // namespace oslam
