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
