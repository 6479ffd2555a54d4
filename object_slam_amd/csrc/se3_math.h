// fp64 SE3 / edge math shared by the pose-optimisation and local-BA kernels (product code).
// Formulas follow g2o's SE3Quat on Eigen::Quaterniond and the ORB_SLAM2 g2o fork's
// types_six_dof_expmap edges; the fork is not in the reference tree.
#pragma once
#include <hip/hip_runtime.h>

namespace oslam {

struct SE3 {
    double q[4];   // x y z w
    double t[3];
};

struct Cam {
    double fx, fy, cx, cy, bf;
};

__host__ __device__ inline void se3_normalize(SE3& s) {   // SE3Quat::normalizeRotation
    if (s.q[3] < 0) { s.q[0] = -s.q[0]; s.q[1] = -s.q[1]; s.q[2] = -s.q[2]; s.q[3] = -s.q[3]; }
    const double n = sqrt(s.q[0] * s.q[0] + s.q[1] * s.q[1] + s.q[2] * s.q[2] + s.q[3] * s.q[3]);
    s.q[0] /= n; s.q[1] /= n; s.q[2] /= n; s.q[3] /= n;
}

__host__ __device__ inline void quat_from_R(const double m[9], double q[4]) {   // Eigen matrix -> quaternion
    double t = m[0] + m[4] + m[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m[7] - m[5]) * t;
        q[1] = (m[2] - m[6]) * t;
        q[2] = (m[3] - m[1]) * t;
    } else if (!(m[4] > m[0]) && !(m[8] > m[0])) {   // i = 0, j = 1, k = 2 (explicit cases: no dynamic indexing)
        t = sqrt(m[0] - m[4] - m[8] + 1.0);
        q[0] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (m[7] - m[5]) * t;
        q[1] = (m[3] + m[1]) * t;
        q[2] = (m[6] + m[2]) * t;
    } else if ((m[4] > m[0]) && !(m[8] > m[4])) {    // i = 1, j = 2, k = 0
        t = sqrt(m[4] - m[8] - m[0] + 1.0);
        q[1] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (m[2] - m[6]) * t;
        q[2] = (m[7] + m[5]) * t;
        q[0] = (m[1] + m[3]) * t;
    } else {                                         // i = 2, j = 0, k = 1
        t = sqrt(m[8] - m[0] - m[4] + 1.0);
        q[2] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (m[3] - m[1]) * t;
        q[0] = (m[2] + m[6]) * t;
        q[1] = (m[5] + m[7]) * t;
    }
}

__host__ __device__ inline void se3_R(const SE3& s, double R[9]) {   // Eigen toRotationMatrix
    const double x = s.q[0], y = s.q[1], z = s.q[2], w = s.q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

__host__ __device__ inline void quat_rot(const double q[4], const double v[3], double o[3]) {
    double uv0 = q[1] * v[2] - q[2] * v[1], uv1 = q[2] * v[0] - q[0] * v[2], uv2 = q[0] * v[1] - q[1] * v[0];
    uv0 += uv0; uv1 += uv1; uv2 += uv2;
    o[0] = v[0] + q[3] * uv0 + (q[1] * uv2 - q[2] * uv1);
    o[1] = v[1] + q[3] * uv1 + (q[2] * uv0 - q[0] * uv2);
    o[2] = v[2] + q[3] * uv2 + (q[0] * uv1 - q[1] * uv0);
}

__host__ __device__ inline void se3_map(const SE3& s, const double X[3], double o[3]) {
    quat_rot(s.q, X, o);
    o[0] += s.t[0]; o[1] += s.t[1]; o[2] += s.t[2];
}

__host__ __device__ inline SE3 se3_mul(const SE3& a, const SE3& b) {   // SE3Quat::operator*
    SE3 r = a;
    double rt[3];
    quat_rot(a.q, b.t, rt);
    r.t[0] += rt[0]; r.t[1] += rt[1]; r.t[2] += rt[2];
    const double ax = a.q[0], ay = a.q[1], az = a.q[2], aw = a.q[3];
    const double bx = b.q[0], by = b.q[1], bz = b.q[2], bw = b.q[3];
    r.q[3] = aw * bw - ax * bx - ay * by - az * bz;
    r.q[0] = aw * bx + ax * bw + ay * bz - az * by;
    r.q[1] = aw * by + ay * bw + az * bx - ax * bz;
    r.q[2] = aw * bz + az * bw + ax * by - ay * bx;
    se3_normalize(r);
    return r;
}

__host__ __device__ inline SE3 se3_exp(const double u[6]) {   // SE3Quat::exp incl. its small-angle branch
    const double wx = u[0], wy = u[1], wz = u[2];
    const double theta = sqrt(wx * wx + wy * wy + wz * wz);
    const double W[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double W2[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) W2[i * 3 + j] = W[i * 3] * W[j] + W[i * 3 + 1] * W[3 + j] + W[i * 3 + 2] * W[6 + j];
    double R[9], V[9];
    if (theta < 0.00001) {
        for (int i = 0; i < 9; i++) { R[i] = ((i % 4) == 0 ? 1.0 : 0.0) + W[i] + W2[i]; V[i] = R[i]; }
    } else {
        const double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta);
        const double c = (theta - sin(theta)) / (theta * theta * theta);
        for (int i = 0; i < 9; i++) {
            const double I = (i % 4) == 0 ? 1.0 : 0.0;
            R[i] = I + a * W[i] + b * W2[i];
            V[i] = I + b * W[i] + c * W2[i];
        }
    }
    SE3 s;
    quat_from_R(R, s.q);
    for (int i = 0; i < 3; i++) s.t[i] = V[i * 3] * u[3] + V[i * 3 + 1] * u[4] + V[i * 3 + 2] * u[5];
    se3_normalize(s);
    return s;
}

__host__ __device__ inline SE3 se3_from_T(const float* T) {   // Converter::toSE3Quat (float -> double)
    double R[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) R[r * 3 + c] = (double)T[r * 4 + c];
    SE3 s;
    quat_from_R(R, s.q);
    s.t[0] = (double)T[3]; s.t[1] = (double)T[7]; s.t[2] = (double)T[11];
    se3_normalize(s);
    return s;
}

__host__ __device__ inline void se3_to_T(const SE3& s, float* T) {   // Converter::toCvMat (double -> float)
    double R[9];
    se3_R(s, R);
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) T[r * 4 + c] = (float)R[r * 3 + c];
        T[r * 4 + 3] = (float)s.t[r];
    }
    T[12] = 0.f; T[13] = 0.f; T[14] = 0.f; T[15] = 1.f;
}

// residual of a (mono | stereo) reprojection edge at camera-frame point p; returns chi2 = info*|e|^2
__host__ __device__ inline double edge_error(const Cam& c, const double p[3], const double obs[3], bool stereo, double info,
                                             double e[3]) {
    if (!stereo) {
        e[0] = obs[0] - (p[0] / p[2] * c.fx + c.cx);
        e[1] = obs[1] - (p[1] / p[2] * c.fy + c.cy);
        e[2] = 0;
        return e[0] * (info * e[0]) + e[1] * (info * e[1]);
    }
    const float invz = (float)(1.0 / p[2]);   // the fork's `const float invz = 1.0f/trans_xyz[2];`
    const double r0 = p[0] * invz * c.fx + c.cx;
    const double r1 = p[1] * invz * c.fy + c.cy;
    const double r2 = r0 - c.bf * invz;
    e[0] = obs[0] - r0; e[1] = obs[1] - r1; e[2] = obs[2] - r2;
    return e[0] * (info * e[0]) + e[1] * (info * e[1]) + e[2] * (info * e[2]);
}

// RobustKernelHuber::robustify: rho0 (robust chi2) and rho1 (weight)
__host__ __device__ inline void huber(double e2, double delta, double& rho0, double& rho1) {
    const double dsqr = delta * delta;
    if (e2 <= dsqr) { rho0 = e2; rho1 = 1.0; }
    else {
        const double sq = sqrt(e2);
        rho0 = 2 * sq * delta - dsqr;
        rho1 = delta / sq;
    }
}

// pose Jacobian (rows u, v, [ur]; columns w1 w2 w3 v1 v2 v3) of the OnlyPose edges (invz products)
__host__ __device__ inline void jac_pose_onlypose(const Cam& c, const double p[3], bool stereo, double J[18]) {
    const double x = p[0], y = p[1], invz = 1.0 / p[2], invz_2 = invz * invz;
    J[0] = x * y * invz_2 * c.fx; J[1] = -(1 + (x * x * invz_2)) * c.fx; J[2] = y * invz * c.fx;
    J[3] = -invz * c.fx; J[4] = 0; J[5] = x * invz_2 * c.fx;
    J[6] = (1 + y * y * invz_2) * c.fy; J[7] = -x * y * invz_2 * c.fy; J[8] = -x * invz * c.fy;
    J[9] = 0; J[10] = -invz * c.fy; J[11] = y * invz_2 * c.fy;
    if (stereo) {
        J[12] = J[0] - c.bf * y * invz_2; J[13] = J[1] + c.bf * x * invz_2; J[14] = J[2];
        J[15] = J[3]; J[16] = 0; J[17] = J[5] - c.bf * invz_2;
    } else {
        J[12] = J[13] = J[14] = J[15] = J[16] = J[17] = 0;
    }
}

// pose (Jp) and point (Jx) Jacobians of the binary edges (divisions by z, z^2)
__host__ __device__ inline void jac_binary(const Cam& c, const double p[3], const double R[9], bool stereo, double Jp[18],
                                           double Jx[9]) {
    const double x = p[0], y = p[1], z = p[2], z_2 = z * z;
    Jp[0] = x * y / z_2 * c.fx; Jp[1] = -(1 + (x * x / z_2)) * c.fx; Jp[2] = y / z * c.fx;
    Jp[3] = -1. / z * c.fx; Jp[4] = 0; Jp[5] = x / z_2 * c.fx;
    Jp[6] = (1 + y * y / z_2) * c.fy; Jp[7] = -x * y / z_2 * c.fy; Jp[8] = -x / z * c.fy;
    Jp[9] = 0; Jp[10] = -1. / z * c.fy; Jp[11] = y / z_2 * c.fy;
    if (!stereo) {
        const double t0 = -1. / z * c.fx, t2 = -1. / z * (-x / z * c.fx);
        const double t4 = -1. / z * c.fy, t5 = -1. / z * (-y / z * c.fy);
        for (int k = 0; k < 3; k++) {
            Jx[k] = t0 * R[k] + t2 * R[6 + k];
            Jx[3 + k] = t4 * R[3 + k] + t5 * R[6 + k];
            Jx[6 + k] = 0;
        }
        Jp[12] = Jp[13] = Jp[14] = Jp[15] = Jp[16] = Jp[17] = 0;
    } else {
        for (int k = 0; k < 3; k++) {
            Jx[k] = -c.fx * R[k] / z + c.fx * x * R[6 + k] / z_2;
            Jx[3 + k] = -c.fy * R[3 + k] / z + c.fy * y * R[6 + k] / z_2;
            Jx[6 + k] = Jx[k] - c.bf * R[6 + k] / z_2;
        }
        Jp[12] = Jp[0] - c.bf * y / z_2; Jp[13] = Jp[1] + c.bf * x / z_2; Jp[14] = Jp[2];
        Jp[15] = Jp[3]; Jp[16] = 0; Jp[17] = Jp[5] - c.bf / z_2;
    }
}

// in-register Cholesky solve of a 6x6 SPD system (stands in for Eigen::LDLT on a 6x6).
// A: row-major full symmetric (destroyed), b: rhs -> solution.  Returns false if not positive.
__host__ __device__ inline bool solve6(double A[36], double b[6]) {
    for (int j = 0; j < 6; j++) {
        double d = A[j * 6 + j];
        for (int k = 0; k < j; k++) d -= A[j * 6 + k] * A[j * 6 + k];
        if (!(d > 0) || !(d < 1.7e308)) return false;
        d = sqrt(d);
        A[j * 6 + j] = d;
        for (int i = j + 1; i < 6; i++) {
            double s = A[i * 6 + j];
            for (int k = 0; k < j; k++) s -= A[i * 6 + k] * A[j * 6 + k];
            A[i * 6 + j] = s / d;
        }
    }
    for (int i = 0; i < 6; i++) {
        double s = b[i];
        for (int k = 0; k < i; k++) s -= A[i * 6 + k] * b[k];
        b[i] = s / A[i * 6 + i];
    }
    for (int i = 5; i >= 0; i--) {
        double s = b[i];
        for (int k = i + 1; k < 6; k++) s -= A[k * 6 + i] * b[k];
        b[i] = s / A[i * 6 + i];
    }
    return true;
}

__host__ __device__ inline bool inv3(const double m[9], double o[9]) {   // Eigen 3x3 inverse by cofactors
    const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    const double id = 1.0 / det;
    o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    return id == id && (id - id) == 0;
}


// ---- device-only fast forms of the per-edge arithmetic (round 4) ------------------------------------------------------------------------------------------
// The optimiser kernels are bound by fp64 instruction issue, so their passes over the edges are written for instruction count: one reciprocal per edge by
// v_rcp_f64 + two Newton steps instead of IEEE divisions (~14 instructions each), the Huber square root and its quotient from one v_rsq_f64 + two Newton steps,
// fused multiply-adds, the pose as rotation matrix + translation (9 fused multiply-adds per point instead of the quaternion sandwich's ~24 instructions), and the
// normal equations accumulated over the NON-ZERO entries of the pose-Jacobian rows only (g2o's rows have J[4] = J[9] = J[16] = 0).  All of it is a few ulp away
// from the divided / unfused form above: far inside the 1e-4 bar the parity tests hold (the float `invz` of the fork's stereo projection is kept).
#ifdef __HIPCC__
__device__ __forceinline__ double rcp_nr(double z) {
    double r = __builtin_amdgcn_rcp(z);
    r = __builtin_fma(__builtin_fma(-z, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-z, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double rsq_nr(double d) {
    double r = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
    r = r * __builtin_fma(-h * r, r, 1.5);
    r = r * __builtin_fma(-h * r, r, 1.5);
    return r;
}
struct PoseRt { double R[9], t[3]; };
__device__ __forceinline__ PoseRt pose_rt(const SE3& P) { PoseRt o; se3_R(P, o.R); o.t[0] = P.t[0]; o.t[1] = P.t[1]; o.t[2] = P.t[2]; return o; }
__device__ __forceinline__ void map_rt(const double* R, const double* t, const double X[3], double p[3]) {
#pragma clang fp contract(fast)
#pragma unroll
    for (int r = 0; r < 3; r++) p[r] = R[r * 3] * X[0] + R[r * 3 + 1] * X[1] + R[r * 3 + 2] * X[2] + t[r];
}
// residual and chi2 = info * |e|^2 of a (mono | stereo) reprojection edge at camera-frame point p; iz = 1 / z is returned for the Jacobians
__device__ __forceinline__ double residual_fast(const Cam& c, const double p[3], const double ob[3], bool stereo, double info, double e[3], double& iz) {
#pragma clang fp contract(fast)
    iz = rcp_nr(p[2]);
    if (!stereo) {
        e[0] = ob[0] - (p[0] * iz * c.fx + c.cx);
        e[1] = ob[1] - (p[1] * iz * c.fy + c.cy);
        e[2] = 0;
        return info * (e[0] * e[0] + e[1] * e[1]);
    }
    const double izf = (double)(float)iz;   // the fork's `const float invz = 1.0f/trans_xyz[2];`
    const double r0 = p[0] * izf * c.fx + c.cx;
    const double r1 = p[1] * izf * c.fy + c.cy;
    const double r2 = r0 - c.bf * izf;
    e[0] = ob[0] - r0; e[1] = ob[1] - r1; e[2] = ob[2] - r2;
    return info * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
}
__device__ __forceinline__ double edge_residual_fast(const Cam& c, const PoseRt& P, const double X[3], const double ob[3], bool stereo, double info, double p[3], double e[3], double& iz) {
    map_rt(P.R, P.t, X, p);
    return residual_fast(c, p, ob, stereo, info, e, iz);
}
__device__ __forceinline__ void huber_fast(double e2, double delta, double& rho0, double& rho1) {
#pragma clang fp contract(fast)
    const double dsqr = delta * delta;
    if (e2 <= dsqr) { rho0 = e2; rho1 = 1.0; }
    else {
        const double rs = rsq_nr(e2), sq = e2 * rs;
        rho0 = 2 * sq * delta - dsqr;
        rho1 = delta * rs;
    }
}
constexpr int h_idx(int a, int cc) { return a * 6 - a * (a - 1) / 2 + (cc - a); }   // position of H(a, cc), a <= cc, among the 21 accumulators of the upper triangle
// acc[0..20] += (w info) Jr^T Jr (upper triangle) and acc[21..26] -= (w info) Jr^T e_r for pose-Jacobian row R of one edge, over the row's non-zero columns
template <int R, int NA>
__device__ __forceinline__ void accumulate_row(const double (&Jr)[6], double er, double wi, double (&acc)[NA]) {
#pragma clang fp contract(fast)
    constexpr int zero = R == 1 ? 3 : 4;   // rows u and ur do not depend on v2 (column 4), row v not on v1 (column 3)
#pragma unroll
    for (int a = 0; a < 6; a++) {
        if (a == zero) continue;
        const double wJ = wi * Jr[a];
        acc[21 + a] -= wJ * er;
#pragma unroll
        for (int cc = a; cc < 6; cc++)
            if (cc != zero) acc[h_idx(a, cc)] += wJ * Jr[cc];
    }
}
// rows of the pose Jacobian (u, v, ur; columns w1 w2 w3 v1 v2 v3) from iz = 1 / z: the formulas of jac_pose_onlypose / jac_binary as products with iz
__device__ __forceinline__ void pose_jac_rows(const Cam& c, const double p[3], double iz, bool stereo, double (&Ju)[6], double (&Jv)[6], double (&Jr)[6]) {
#pragma clang fp contract(fast)
    const double x = p[0], y = p[1], iz2 = iz * iz, fxiz = c.fx * iz, fyiz = c.fy * iz, xiz2 = x * iz2, yiz2 = y * iz2;
    Ju[0] = x * yiz2 * c.fx; Ju[1] = -(1 + x * xiz2) * c.fx; Ju[2] = y * fxiz; Ju[3] = -fxiz; Ju[4] = 0; Ju[5] = xiz2 * c.fx;
    Jv[0] = (1 + y * yiz2) * c.fy; Jv[1] = -x * yiz2 * c.fy; Jv[2] = -x * fyiz; Jv[3] = 0; Jv[4] = -fyiz; Jv[5] = yiz2 * c.fy;
    if (stereo) {
        const double tb = c.bf * iz2;
        Jr[0] = Ju[0] - tb * y; Jr[1] = Ju[1] + tb * x; Jr[2] = Ju[2]; Jr[3] = Ju[3]; Jr[4] = 0; Jr[5] = Ju[5] - tb;
    } else {
#pragma unroll
        for (int k = 0; k < 6; k++) Jr[k] = 0;
    }
}
// rows of the point Jacobian of a binary edge (jac_binary's Jx = -d proj / d p * R) from iz
__device__ __forceinline__ void point_jac_rows(const Cam& c, const double p[3], double iz, const double R[9], bool stereo, double (&Jx)[9]) {
#pragma clang fp contract(fast)
    const double iz2 = iz * iz, fxiz = c.fx * iz, fyiz = c.fy * iz, t2 = p[0] * iz2 * c.fx, t5 = p[1] * iz2 * c.fy, tb = c.bf * iz2;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        Jx[k] = -fxiz * R[k] + t2 * R[6 + k];
        Jx[3 + k] = -fyiz * R[3 + k] + t5 * R[6 + k];
        Jx[6 + k] = stereo ? Jx[k] - tb * R[6 + k] : 0.0;
    }
}
#endif

}  // namespace oslam
