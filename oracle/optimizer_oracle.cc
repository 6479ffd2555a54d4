// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_common.h).  PARITY UNPINNED.
// See optimizer_oracle.h.  g2o arithmetic restated from the published ORB_SLAM2 Thirdparty/g2o
// sources (types_six_dof_expmap.{h,cpp}, se3quat.h, robust_kernel_impl.cpp,
// optimization_algorithm_levenberg.cpp, block_solver.hpp, base_{unary,binary}_edge.hpp) and
// Eigen's Quaternion / 3x3 inverse formulas.
#include "optimizer_oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <limits>
#include <map>

namespace oracle {

// ------------------------------------------------------------------------------------------
// SE3Quat (g2o/types/se3quat.h) on Eigen::Quaterniond semantics
// ------------------------------------------------------------------------------------------
static void quat_normalize_rotation(SE3Quat& s) {   // SE3Quat::normalizeRotation
    if (s.q[3] < 0) for (int i = 0; i < 4; i++) s.q[i] *= -1;
    const double n = std::sqrt(s.q[0] * s.q[0] + s.q[1] * s.q[1] + s.q[2] * s.q[2] + s.q[3] * s.q[3]);
    for (int i = 0; i < 4; i++) s.q[i] /= n;
}

static void quat_from_matrix(const double m[9], double q[4]) {   // Eigen quaternionbase_assign_impl<3,3>
    double t = m[0] + m[4] + m[8];
    if (t > 0) {
        t = std::sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m[2 * 3 + 1] - m[1 * 3 + 2]) * t;
        q[1] = (m[0 * 3 + 2] - m[2 * 3 + 0]) * t;
        q[2] = (m[1 * 3 + 0] - m[0 * 3 + 1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 3 + i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (m[k * 3 + j] - m[j * 3 + k]) * t;
        q[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
        q[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
    }
}

void se3_rotation(const SE3Quat& s, double R[9]) {   // Eigen QuaternionBase::toRotationMatrix
    const double x = s.q[0], y = s.q[1], z = s.q[2], w = s.q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

static void quat_rotate(const double q[4], const double v[3], double out[3]) {   // Eigen _transformVector
    double uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
    for (int i = 0; i < 3; i++) uv[i] += uv[i];
    const double c[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]};
    for (int i = 0; i < 3; i++) out[i] = v[i] + q[3] * uv[i] + c[i];
}

void se3_map(const SE3Quat& s, const double* X, double* out) {
    quat_rotate(s.q, X, out);
    for (int i = 0; i < 3; i++) out[i] += s.t[i];
}

SE3Quat se3_mul(const SE3Quat& a, const SE3Quat& b) {   // SE3Quat::operator*
    SE3Quat r = a;
    double rt[3];
    quat_rotate(a.q, b.t, rt);
    for (int i = 0; i < 3; i++) r.t[i] += rt[i];
    const double ax = a.q[0], ay = a.q[1], az = a.q[2], aw = a.q[3];
    const double bx = b.q[0], by = b.q[1], bz = b.q[2], bw = b.q[3];
    r.q[3] = aw * bw - ax * bx - ay * by - az * bz;
    r.q[0] = aw * bx + ax * bw + ay * bz - az * by;
    r.q[1] = aw * by + ay * bw + az * bx - ax * bz;
    r.q[2] = aw * bz + az * bw + ax * by - ay * bx;
    quat_normalize_rotation(r);
    return r;
}

SE3Quat se3_exp(const double* u) {   // SE3Quat::exp (note the small-angle branch: R = I + W + W*W, V = R)
    const double omega[3] = {u[0], u[1], u[2]}, upsilon[3] = {u[3], u[4], u[5]};
    const double theta = std::sqrt(omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2]);
    const double W[9] = {0, -omega[2], omega[1], omega[2], 0, -omega[0], -omega[1], omega[0], 0};
    double W2[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += W[i * 3 + k] * W[k * 3 + j];
            W2[i * 3 + j] = s;
        }
    double R[9], V[9];
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (theta < 0.00001) {
        for (int i = 0; i < 9; i++) { R[i] = I[i] + W[i] + W2[i]; V[i] = R[i]; }
    } else {
        const double a = std::sin(theta) / theta, b = (1 - std::cos(theta)) / (theta * theta);
        const double c = (theta - std::sin(theta)) / std::pow(theta, 3);
        for (int i = 0; i < 9; i++) { R[i] = I[i] + a * W[i] + b * W2[i]; V[i] = I[i] + b * W[i] + c * W2[i]; }
    }
    SE3Quat s;
    quat_from_matrix(R, s.q);
    for (int i = 0; i < 3; i++) s.t[i] = V[i * 3] * upsilon[0] + V[i * 3 + 1] * upsilon[1] + V[i * 3 + 2] * upsilon[2];
    quat_normalize_rotation(s);
    return s;
}

SE3Quat se3_from_cvmat(const float* T) {
    double R[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) R[r * 3 + c] = T[r * 4 + c];
    SE3Quat s;
    quat_from_matrix(R, s.q);
    for (int r = 0; r < 3; r++) s.t[r] = T[r * 4 + 3];
    quat_normalize_rotation(s);
    return s;
}

void se3_to_cvmat(const SE3Quat& s, float* T) {
    double R[9];
    se3_rotation(s, R);
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) T[r * 4 + c] = (float)R[r * 3 + c];
        T[r * 4 + 3] = (float)s.t[r];
    }
    T[12] = T[13] = T[14] = 0.f;
    T[15] = 1.f;
}

// ------------------------------------------------------------------------------------------
// edges (types_six_dof_expmap)
// ------------------------------------------------------------------------------------------
static void edge_point(const Graph& g, const GraphEdge& e, double X[3]) {
    if (e.point >= 0) for (int i = 0; i < 3; i++) X[i] = g.points[e.point][i];
    else for (int i = 0; i < 3; i++) X[i] = e.Xw[i];
}

void edge_compute_error(const Graph& g, GraphEdge& e) {
    double X[3], p[3];
    edge_point(g, e, X);
    se3_map(g.poses[e.pose], X, p);
    const Camera& c = g.cam;
    if (!e.stereo) {   // cam_project: project2d then *f + c, all double
        const double px = p[0] / p[2], py = p[1] / p[2];
        e.err[0] = e.obs[0] - (px * c.fx + c.cx);
        e.err[1] = e.obs[1] - (py * c.fy + c.cy);
        e.err[2] = 0;
    } else {           // Edge(Stereo)...::cam_project: `const float invz = 1.0f/trans_xyz[2];`
        const float invz = 1.0f / p[2];
        const double r0 = p[0] * invz * c.fx + c.cx;
        const double r1 = p[1] * invz * c.fy + c.cy;
        const double r2 = r0 - c.bf * invz;
        e.err[0] = e.obs[0] - r0;
        e.err[1] = e.obs[1] - r1;
        e.err[2] = e.obs[2] - r2;
    }
}

double edge_chi2(const GraphEdge& e) {   // _error.dot(information()*_error)
    const int D = e.stereo ? 3 : 2;
    double s = 0;
    for (int i = 0; i < D; i++) s += e.err[i] * (e.info * e.err[i]);
    return s;
}

bool edge_depth_positive(const Graph& g, const GraphEdge& e) {
    double X[3], p[3];
    edge_point(g, e, X);
    se3_map(g.poses[e.pose], X, p);
    return p[2] > 0.0;
}

static void huber(double e2, double delta, double rho[3]) {   // RobustKernelHuber::robustify
    const double dsqr = delta * delta;
    if (e2 <= dsqr) { rho[0] = e2; rho[1] = 1.; rho[2] = 0.; }
    else {
        const double sqrte = std::sqrt(e2);
        rho[0] = 2 * sqrte * delta - dsqr;
        rho[1] = delta / sqrte;
        rho[2] = -0.5 * rho[1] / e2;
    }
}

// Jacobians: Jp (D x 6, pose, columns w1 w2 w3 v1 v2 v3) and Jx (D x 3, point).
void edge_jacobians(const Graph& g, const GraphEdge& e, double Jp[18], double Jx[9]) {
    double X[3], p[3], R[9];
    edge_point(g, e, X);
    const SE3Quat& T = g.poses[e.pose];
    se3_map(T, X, p);
    se3_rotation(T, R);
    const Camera& c = g.cam;
    const double x = p[0], y = p[1], z = p[2];
    if (e.point < 0) {   // OnlyPose edges use invz products
        const double invz = 1.0 / z, invz_2 = invz * invz;
        Jp[0] = x * y * invz_2 * c.fx; Jp[1] = -(1 + (x * x * invz_2)) * c.fx; Jp[2] = y * invz * c.fx;
        Jp[3] = -invz * c.fx; Jp[4] = 0; Jp[5] = x * invz_2 * c.fx;
        Jp[6] = (1 + y * y * invz_2) * c.fy; Jp[7] = -x * y * invz_2 * c.fy; Jp[8] = -x * invz * c.fy;
        Jp[9] = 0; Jp[10] = -invz * c.fy; Jp[11] = y * invz_2 * c.fy;
        if (e.stereo) {
            Jp[12] = Jp[0] - c.bf * y * invz_2; Jp[13] = Jp[1] + c.bf * x * invz_2; Jp[14] = Jp[2];
            Jp[15] = Jp[3]; Jp[16] = 0; Jp[17] = Jp[5] - c.bf * invz_2;
        }
        return;
    }
    const double z_2 = z * z;
    Jp[0] = x * y / z_2 * c.fx; Jp[1] = -(1 + (x * x / z_2)) * c.fx; Jp[2] = y / z * c.fx;
    Jp[3] = -1. / z * c.fx; Jp[4] = 0; Jp[5] = x / z_2 * c.fx;
    Jp[6] = (1 + y * y / z_2) * c.fy; Jp[7] = -x * y / z_2 * c.fy; Jp[8] = -x / z * c.fy;
    Jp[9] = 0; Jp[10] = -1. / z * c.fy; Jp[11] = y / z_2 * c.fy;
    if (!e.stereo) {
        // -1./z * tmp * R, tmp = [[fx,0,-x/z*fx],[0,fy,-y/z*fy]]
        const double tmp[6] = {c.fx, 0, -x / z * c.fx, 0, c.fy, -y / z * c.fy};
        for (int r = 0; r < 2; r++)
            for (int k = 0; k < 3; k++) {
                double s = 0;
                for (int m = 0; m < 3; m++) s += (-1. / z * tmp[r * 3 + m]) * R[m * 3 + k];
                Jx[r * 3 + k] = s;
            }
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

// ------------------------------------------------------------------------------------------
// dense symmetric positive-definite solve (stands in for Eigen LDLT / SimplicialLDLT)
// ------------------------------------------------------------------------------------------
static bool spd_solve(std::vector<double>& A, int n, std::vector<double>& b) {
    // in-place Cholesky A = L L^T (lower), then two triangular solves
    for (int j = 0; j < n; j++) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; k++) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
        if (!(d > 0) || !std::isfinite(d)) return false;
        d = std::sqrt(d);
        A[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[(size_t)i * n + j];
            for (int k = 0; k < j; k++) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
            A[(size_t)i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; i++) {
        double s = b[i];
        for (int k = 0; k < i; k++) s -= A[(size_t)i * n + k] * b[k];
        b[i] = s / A[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = b[i];
        for (int k = i + 1; k < n; k++) s -= A[(size_t)k * n + i] * b[k];
        b[i] = s / A[(size_t)i * n + i];
    }
    return true;
}

static bool inv3(const double m[9], double o[9]) {   // Eigen 3x3 inverse (cofactors / determinant)
    const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    const double id = 1.0 / det;
    o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    return std::isfinite(id);
}

// ------------------------------------------------------------------------------------------
// SparseOptimizer::optimize + OptimizationAlgorithmLevenberg + BlockSolver_6_3
// ------------------------------------------------------------------------------------------
namespace {
struct System {
    std::vector<int> act;                 // active edge indices (insertion order)
    std::vector<int> pose_idx, point_idx; // vertex -> block index or -1
    std::vector<int> poses, points;       // block index -> vertex
    int nP = 0, nL = 0;
    std::vector<double> Hpp;              // nP x 36 (block diagonal)
    std::vector<double> Hll;              // nL x 9
    std::map<std::pair<int, int>, std::array<double, 18>> Hpl;   // (pose block, point block) -> 6x3
    std::vector<double> b;                // 6 nP + 3 nL
    std::vector<double> x;
};

double active_robust_chi2(Graph& g, const System& s, bool compute_errors) {
    double F = 0;
    for (int ei : s.act) {
        GraphEdge& e = g.edges[ei];
        if (compute_errors) edge_compute_error(g, e);
        const double c2 = edge_chi2(e);
        if (e.robust) {
            double rho[3];
            huber(c2, e.delta, rho);
            F += rho[0];
        } else
            F += c2;
    }
    return F;
}

void build_system(Graph& g, System& s) {
    std::fill(s.Hpp.begin(), s.Hpp.end(), 0.0);
    std::fill(s.Hll.begin(), s.Hll.end(), 0.0);
    for (auto& kv : s.Hpl) kv.second.fill(0.0);
    std::fill(s.b.begin(), s.b.end(), 0.0);
    for (int ei : s.act) {
        GraphEdge& e = g.edges[ei];
        const int D = e.stereo ? 3 : 2;
        double Jp[18] = {0}, Jx[9] = {0};
        edge_jacobians(g, e, Jp, Jx);
        double w = 1.0;
        if (e.robust) {
            double rho[3];
            huber(edge_chi2(e), e.delta, rho);
            w = rho[1];
        }
        const double wi = w * e.info;          // weightedOmega = rho[1] * information
        const int pb = s.pose_idx[e.pose];
        const int lb = e.point >= 0 ? s.point_idx[e.point] : -1;
        if (pb >= 0) {
            double* H = &s.Hpp[(size_t)pb * 36];
            double* bp = &s.b[(size_t)pb * 6];
            for (int a = 0; a < 6; a++) {
                double sb = 0;
                for (int d = 0; d < D; d++) sb += Jp[d * 6 + a] * (e.info * e.err[d]);
                bp[a] -= w * sb;   // b -= rho[1] * J^T * omega * error
                for (int c = 0; c < 6; c++) {
                    double sh = 0;
                    for (int d = 0; d < D; d++) sh += Jp[d * 6 + a] * wi * Jp[d * 6 + c];
                    H[a * 6 + c] += sh;
                }
            }
        }
        if (lb >= 0) {
            double* H = &s.Hll[(size_t)lb * 9];
            double* bl = &s.b[(size_t)s.nP * 6 + (size_t)lb * 3];
            for (int a = 0; a < 3; a++) {
                double sb = 0;
                for (int d = 0; d < D; d++) sb += Jx[d * 3 + a] * (e.info * e.err[d]);
                bl[a] -= w * sb;
                for (int c = 0; c < 3; c++) {
                    double sh = 0;
                    for (int d = 0; d < D; d++) sh += Jx[d * 3 + a] * wi * Jx[d * 3 + c];
                    H[a * 3 + c] += sh;
                }
            }
            if (pb >= 0) {
                auto& B = s.Hpl[std::make_pair(pb, lb)];
                for (int a = 0; a < 6; a++)
                    for (int c = 0; c < 3; c++) {
                        double sh = 0;
                        for (int d = 0; d < D; d++) sh += Jp[d * 6 + a] * wi * Jx[d * 3 + c];
                        B[a * 3 + c] += sh;
                    }
            }
        }
    }
}

bool solve_system(System& s, double lambda) {
    const int nP = s.nP, nL = s.nL, n = 6 * nP;
    std::fill(s.x.begin(), s.x.end(), 0.0);
    if (nL == 0) {
        // pose-only: Hpp is block diagonal; each 6x6 solved on its own (dense LDLT in g2o)
        for (int p = 0; p < nP; p++) {
            std::vector<double> A(s.Hpp.begin() + (size_t)p * 36, s.Hpp.begin() + (size_t)(p + 1) * 36);
            for (int i = 0; i < 6; i++) A[i * 6 + i] += lambda;
            std::vector<double> rhs(s.b.begin() + p * 6, s.b.begin() + p * 6 + 6);
            if (!spd_solve(A, 6, rhs)) return false;
            for (int i = 0; i < 6; i++) s.x[p * 6 + i] = rhs[i];
        }
        return true;
    }
    std::vector<double> Hs((size_t)n * n, 0.0), bs(s.b.begin(), s.b.begin() + n);
    for (int p = 0; p < nP; p++)
        for (int a = 0; a < 6; a++)
            for (int c = 0; c < 6; c++) Hs[(size_t)(p * 6 + a) * n + p * 6 + c] = s.Hpp[(size_t)p * 36 + a * 6 + c] + (a == c ? lambda : 0.0);
    // group Hpl by landmark
    std::vector<std::vector<std::pair<int, const double*>>> cols(nL);
    for (auto& kv : s.Hpl) cols[kv.first.second].push_back(std::make_pair(kv.first.first, kv.second.data()));
    std::vector<double> Dinv((size_t)nL * 9);
    for (int l = 0; l < nL; l++) {
        double D[9];
        for (int i = 0; i < 9; i++) D[i] = s.Hll[(size_t)l * 9 + i];
        D[0] += lambda; D[4] += lambda; D[8] += lambda;
        double* Di = &Dinv[(size_t)l * 9];
        inv3(D, Di);
        const double* bl = &s.b[(size_t)n + (size_t)l * 3];
        double db[3];
        for (int i = 0; i < 3; i++) db[i] = Di[i * 3] * bl[0] + Di[i * 3 + 1] * bl[1] + Di[i * 3 + 2] * bl[2];
        auto& col = cols[l];
        std::sort(col.begin(), col.end(), [](const std::pair<int, const double*>& a, const std::pair<int, const double*>& b2) { return a.first < b2.first; });
        for (size_t i1 = 0; i1 < col.size(); i1++) {
            const int pa = col[i1].first;
            const double* Bi = col[i1].second;
            double BD[18];
            for (int a = 0; a < 6; a++)
                for (int c = 0; c < 3; c++) BD[a * 3 + c] = Bi[a * 3] * Di[c] + Bi[a * 3 + 1] * Di[3 + c] + Bi[a * 3 + 2] * Di[6 + c];
            for (int a = 0; a < 6; a++) bs[pa * 6 + a] -= Bi[a * 3] * db[0] + Bi[a * 3 + 1] * db[1] + Bi[a * 3 + 2] * db[2];
            for (size_t i2 = i1; i2 < col.size(); i2++) {
                const int pb2 = col[i2].first;
                const double* Bj = col[i2].second;
                for (int a = 0; a < 6; a++)
                    for (int c = 0; c < 6; c++) {
                        const double v = BD[a * 3] * Bj[c * 3] + BD[a * 3 + 1] * Bj[c * 3 + 1] + BD[a * 3 + 2] * Bj[c * 3 + 2];
                        Hs[(size_t)(pa * 6 + a) * n + pb2 * 6 + c] -= v;
                        if (pa != pb2) Hs[(size_t)(pb2 * 6 + c) * n + pa * 6 + a] -= v;
                    }
            }
        }
    }
    if (!spd_solve(Hs, n, bs)) return false;
    for (int i = 0; i < n; i++) s.x[i] = bs[i];
    // landmarks: x_l = Dinv (b_l - sum_a B_a^T x_a)
    for (int l = 0; l < nL; l++) {
        double cl[3] = {s.b[(size_t)n + l * 3], s.b[(size_t)n + l * 3 + 1], s.b[(size_t)n + l * 3 + 2]};
        for (auto& pr : cols[l]) {
            const double* B = pr.second;
            const double* xp = &s.x[(size_t)pr.first * 6];
            for (int c = 0; c < 3; c++) {
                double sv = 0;
                for (int a = 0; a < 6; a++) sv += B[a * 3 + c] * xp[a];
                cl[c] -= sv;
            }
        }
        const double* Di = &Dinv[(size_t)l * 9];
        for (int i = 0; i < 3; i++) s.x[(size_t)n + l * 3 + i] = Di[i * 3] * cl[0] + Di[i * 3 + 1] * cl[1] + Di[i * 3 + 2] * cl[2];
    }
    return true;
}
}  // namespace

// Test hook: the Levenberg-Marquardt trials of every graph_optimize call while set (F before, F of the trial, rho, lambda of the trial, accepted, first trial of the call).
static double* g_trace = nullptr;
static int g_trace_cap = 0, g_trace_n = 0;
void lm_trace_set(double* buf, int cap) { g_trace = buf; g_trace_cap = cap; g_trace_n = 0; }
int lm_trace_count() { return g_trace_n; }

int graph_optimize(Graph& g, int iterations, int level, const volatile int* stop) {
    System s;
    s.pose_idx.assign(g.poses.size(), -1);
    s.point_idx.assign(g.points.size(), -1);
    std::vector<uint8_t> pose_used(g.poses.size(), 0), point_used(g.points.size(), 0);
    for (size_t i = 0; i < g.edges.size(); i++)
        if (g.edges[i].level == level) {
            s.act.push_back((int)i);
            pose_used[g.edges[i].pose] = 1;
            if (g.edges[i].point >= 0) point_used[g.edges[i].point] = 1;
        }
    for (size_t p = 0; p < g.poses.size(); p++)
        if (pose_used[p] && !g.pose_fixed[p]) { s.pose_idx[p] = s.nP++; s.poses.push_back((int)p); }
    for (size_t p = 0; p < g.points.size(); p++)
        if (point_used[p]) { s.point_idx[p] = s.nL++; s.points.push_back((int)p); }
    g.lm_iterations = 0;
    g.lm_trials = 0;
    if (s.nP + s.nL == 0) return -1;   // "0 vertices to optimize"
    s.Hpp.assign((size_t)s.nP * 36, 0.0);
    s.Hll.assign((size_t)s.nL * 9, 0.0);
    s.b.assign((size_t)s.nP * 6 + (size_t)s.nL * 3, 0.0);
    s.x.assign(s.b.size(), 0.0);
    for (int ei : s.act) {
        const GraphEdge& e = g.edges[ei];
        if (e.point >= 0 && s.pose_idx[e.pose] >= 0)
            s.Hpl[std::make_pair(s.pose_idx[e.pose], s.point_idx[e.point])].fill(0.0);
    }
    auto terminate = [&]() { return stop ? (*stop != 0) : false; };

    double lambda = 0, ni = 2;
    const int maxTrials = 10;
    const double goodLow = 1. / 3., goodUp = 2. / 3.;
    int done = 0;
    bool ok = true;
    for (int it = 0; it < iterations && !terminate() && ok; it++) {
        double currentChi = active_robust_chi2(g, s, true);   // computeActiveErrors + activeRobustChi2
        double tempChi = currentChi;
        build_system(g, s);
        if (it == 0) {   // computeLambdaInit: tau * max |H_jj| over all non-fixed active vertices
            double maxDiag = 0;
            for (int p = 0; p < s.nP; p++)
                for (int k = 0; k < 6; k++) maxDiag = std::max(std::fabs(s.Hpp[(size_t)p * 36 + k * 7]), maxDiag);
            for (int l = 0; l < s.nL; l++)
                for (int k = 0; k < 3; k++) maxDiag = std::max(std::fabs(s.Hll[(size_t)l * 9 + k * 4]), maxDiag);
            lambda = 1e-5 * maxDiag;
            ni = 2;
        }
        double rho = 0;
        int qmax = 0;
        do {
            // push
            std::vector<SE3Quat> bp;
            std::vector<std::array<double, 3>> bl;
            for (int p : s.poses) bp.push_back(g.poses[p]);
            for (int l : s.points) bl.push_back(g.points[l]);
            const bool ok2 = solve_system(s, lambda);
            // update (x stays 0 on a failed solve)
            for (int p = 0; p < s.nP; p++) g.poses[s.poses[p]] = se3_mul(se3_exp(&s.x[(size_t)p * 6]), g.poses[s.poses[p]]);
            for (int l = 0; l < s.nL; l++)
                for (int k = 0; k < 3; k++) g.points[s.points[l]][k] += s.x[(size_t)s.nP * 6 + l * 3 + k];
            tempChi = active_robust_chi2(g, s, true);
            if (!ok2) tempChi = std::numeric_limits<double>::max();
            rho = (currentChi - tempChi);
            double scale = 0;   // computeScale
            for (size_t j = 0; j < s.x.size(); j++) scale += s.x[j] * (lambda * s.x[j] + s.b[j]);
            scale += 1e-3;
            rho /= scale;
            if (g_trace) {
                if (g_trace_n < g_trace_cap) {
                    double* t = g_trace + 6 * (size_t)g_trace_n;
                    t[0] = currentChi; t[1] = tempChi; t[2] = rho; t[3] = lambda; t[4] = (rho > 0 && std::isfinite(tempChi)) ? 1.0 : 0.0; t[5] = (it == 0 && qmax == 0) ? 1.0 : 0.0;
                }
                g_trace_n++;
            }
            if (rho > 0 && std::isfinite(tempChi)) {
                double alpha = 1. - std::pow((2 * rho - 1), 3);
                alpha = std::min(alpha, goodUp);
                const double scaleFactor = std::max(goodLow, alpha);
                lambda *= scaleFactor;
                ni = 2;
                currentChi = tempChi;
            } else {
                lambda *= ni;
                ni *= 2;
                for (int p = 0; p < s.nP; p++) g.poses[s.poses[p]] = bp[p];   // pop (edge errors stay stale)
                for (int l = 0; l < s.nL; l++) g.points[s.points[l]] = bl[l];
            }
            qmax++;
            g.lm_trials++;
        } while (rho < 0 && qmax < maxTrials && !terminate());
        done++;
        g.lm_iterations = done;
        if (qmax == maxTrials || rho == 0) ok = false;   // Terminate
    }
    return done;
}

// ------------------------------------------------------------------------------------------
// Optimizer::PoseOptimization, reference src/Optimizer.cc:239-451
// ------------------------------------------------------------------------------------------
int PoseOptimization(int N, const float* Tcw_in, const float* Xw, const float* obs, const float* invSigma2,
                     const uint8_t* has_mp, const float* K5, float* Tcw_out, uint8_t* outlier, int* stats) {
    Graph g;
    g.cam = Camera{K5[0], K5[1], K5[2], K5[3], K5[4]};
    g.poses.push_back(se3_from_cvmat(Tcw_in));
    g.pose_fixed.push_back(0);
    const float deltaMono = sqrt(5.991);
    const float deltaStereo = sqrt(7.815);
    std::vector<int> idx;
    int nInitialCorrespondences = 0;
    for (int i = 0; i < N; i++) {
        if (!has_mp[i]) continue;
        nInitialCorrespondences++;
        outlier[i] = 0;
        GraphEdge e;
        memset(&e, 0, sizeof(e));
        e.pose = 0;
        e.point = -1;
        e.stereo = !(obs[i * 3 + 2] < 0);
        e.obs[0] = obs[i * 3]; e.obs[1] = obs[i * 3 + 1]; e.obs[2] = e.stereo ? obs[i * 3 + 2] : 0;
        e.info = invSigma2[i];
        e.robust = true;
        e.delta = e.stereo ? deltaStereo : deltaMono;
        e.level = 0;
        for (int k = 0; k < 3; k++) e.Xw[k] = Xw[i * 3 + k];
        g.edges.push_back(e);
        idx.push_back(i);
    }
    if (stats) stats[0] = stats[1] = 0;
    if (nInitialCorrespondences < 3) {
        // reference returns 0 and leaves the pose untouched
        memcpy(Tcw_out, Tcw_in, 16 * sizeof(float));
        return 0;
    }
    const float chi2Mono[4] = {5.991, 5.991, 5.991, 5.991};
    const float chi2Stereo[4] = {7.815, 7.815, 7.815, 7.815};
    const int its[4] = {10, 10, 10, 10};
    int nBad = 0;
    for (size_t it = 0; it < 4; it++) {
        g.poses[0] = se3_from_cvmat(Tcw_in);
        graph_optimize(g, its[it], 0, nullptr);
        if (stats) { stats[0] += g.lm_iterations; stats[1] += g.lm_trials; }
        nBad = 0;
        // the reference walks mono edges then stereo edges; the per-edge logic is identical and independent
        for (size_t k = 0; k < g.edges.size(); k++) {
            GraphEdge& e = g.edges[k];
            const int i = idx[k];
            if (outlier[i]) edge_compute_error(g, e);
            const float chi2 = edge_chi2(e);
            const float th = e.stereo ? chi2Stereo[it] : chi2Mono[it];
            if (chi2 > th) { outlier[i] = 1; e.level = 1; nBad++; }
            else { outlier[i] = 0; e.level = 0; }
            if (it == 2) e.robust = false;
        }
        if (g.edges.size() < 10) break;
    }
    se3_to_cvmat(g.poses[0], Tcw_out);
    return nInitialCorrespondences - nBad;
}

// ------------------------------------------------------------------------------------------
// Optimizer::LocalBundleAdjustment, reference src/Optimizer.cc:453-778 (graph already gathered)
// ------------------------------------------------------------------------------------------
void LocalBundleAdjustment(int nKF, const float* poses, const uint8_t* fixed, int nP, const float* points, int nE,
                           const int32_t* edge_kf, const int32_t* edge_pt, const float* edge_obs,
                           const float* edge_invSigma2, const float* K5, const volatile int* stop, float* poses_out,
                           float* points_out, uint8_t* erase, int* stats) {
    Graph g;
    g.cam = Camera{K5[0], K5[1], K5[2], K5[3], K5[4]};
    for (int k = 0; k < nKF; k++) { g.poses.push_back(se3_from_cvmat(poses + 16 * k)); g.pose_fixed.push_back(fixed[k]); }
    for (int p = 0; p < nP; p++) g.points.push_back({(double)points[3 * p], (double)points[3 * p + 1], (double)points[3 * p + 2]});
    const float thHuberMono = sqrt(5.991);
    const float thHuberStereo = sqrt(7.815);
    for (int i = 0; i < nE; i++) {
        GraphEdge e;
        memset(&e, 0, sizeof(e));
        e.pose = edge_kf[i];
        e.point = edge_pt[i];
        e.stereo = !(edge_obs[3 * i + 2] < 0);
        e.obs[0] = edge_obs[3 * i]; e.obs[1] = edge_obs[3 * i + 1]; e.obs[2] = e.stereo ? edge_obs[3 * i + 2] : 0;
        e.info = edge_invSigma2[i];
        e.robust = true;
        e.delta = e.stereo ? thHuberStereo : thHuberMono;
        g.edges.push_back(e);
    }
    memcpy(poses_out, poses, (size_t)nKF * 16 * sizeof(float));
    memcpy(points_out, points, (size_t)nP * 3 * sizeof(float));
    memset(erase, 0, nE);
    if (stats) stats[0] = stats[1] = stats[2] = stats[3] = 0;
    if (stop && *stop) return;   // :655-657

    graph_optimize(g, 5, 0, stop);
    if (stats) { stats[0] = g.lm_iterations; stats[1] = g.lm_trials; }
    bool bDoMore = true;
    if (stop && *stop) bDoMore = false;
    if (bDoMore) {
        for (auto& e : g.edges) {
            const double th = e.stereo ? 7.815 : 5.991;
            if (edge_chi2(e) > th || !edge_depth_positive(g, e)) e.level = 1;
            e.robust = false;
        }
        graph_optimize(g, 10, 0, stop);
        if (stats) { stats[2] = g.lm_iterations; stats[3] = g.lm_trials; }
    }
    for (int i = 0; i < nE; i++) {
        const GraphEdge& e = g.edges[i];
        const double th = e.stereo ? 7.815 : 5.991;
        if (edge_chi2(e) > th || !edge_depth_positive(g, e)) erase[i] = 1;
    }
    // write back local keyframes (:762-768) and points (:771-777); fixed keyframes are not written
    for (int k = 0; k < nKF; k++)
        if (fixed[k] != 1) se3_to_cvmat(g.poses[k], poses_out + 16 * k);   // 2 = local KF with mnId==0: fixed but written back
    for (int p = 0; p < nP; p++)
        for (int c = 0; c < 3; c++) points_out[3 * p + c] = (float)g.points[p][c];
}

}  // namespace oracle

// test hook (see oracle_capi.cc)
extern "C" void oo_internal_edge_jac(const oracle::Graph& g, const oracle::GraphEdge& e, double* Jp, double* Jx) {
    double a[18] = {0}, b[9] = {0};
    oracle::edge_jacobians(g, e, a, b);
    for (int i = 0; i < 18; i++) Jp[i] = a[i];
    for (int i = 0; i < 9; i++) Jx[i] = b[i];
}

namespace oracle {
namespace {
struct MaskArea {
    std::vector<float> x, y;   // pcl::PointXY(col, row) in row-major scan order (:699-710)
    // nearestKSearch(p, 1): exact NN, squared float L2 (FLANN L2_Simple: ((0 + dx*dx) + dy*dy))
    bool nearest(float u, float v, int& idx, float& d2) const {
        if (x.empty()) return false;
        float best = 0;
        int bi = -1;
        for (size_t i = 0; i < x.size(); i++) {
            const float dx = x[i] - u, dy = y[i] - v;
            float d = 0;
            d += dx * dx;
            d += dy * dy;
            if (bi < 0 || d < best) { best = d; bi = (int)i; }
        }
        idx = bi;
        d2 = best;
        return true;
    }
};
// cv::Mat  R(3x3 float) * P(3x1 float) + t  as one cv::gemm (flags==0, len==3): OpenCV 3.2's small-matrix
// branch accumulates the row product in float, then d = (float)(t0*alpha + c*beta), alpha = beta = 1.0 (double)
inline void project_f32(const float* T, const float* P, float Pc[3]) {
    for (int r = 0; r < 3; r++) {
        const float t0 = T[r * 4] * P[0] + T[r * 4 + 1] * P[1] + T[r * 4 + 2] * P[2];
        Pc[r] = (float)((double)t0 * 1.0 + (double)T[r * 4 + 3] * 1.0);
    }
}
}  // namespace

int PoseOptimization2(int N, const float* Tcw_in, const float* Xw, const float* obs, const float* invSigma2,
                      const uint8_t* has_mp, const float* K5, int nObj, int H, int W, const uint8_t* masks,
                      int nObjMp, const float* objmp_Xw, const int32_t* objmp_obj, int nJoint, const int32_t* joint_kp,
                      const int32_t* joint_obj, const float* kp_uv, const float* bounds, float invSigma2_0,
                      float* Tcw_out, uint8_t* outlier, int* nSemNumOut) {
    Graph g;
    g.cam = Camera{K5[0], K5[1], K5[2], K5[3], K5[4]};
    const float fx = K5[0], fy = K5[1], cx = K5[2], cy = K5[3];
    g.poses.push_back(se3_from_cvmat(Tcw_in));
    g.pose_fixed.push_back(0);
    const float deltaMono = sqrt(5.991);
    const float deltaStereo = sqrt(7.815);
    int nSemNum = 0;

    std::vector<MaskArea> areas(nObj);
    for (int o = 0; o < nObj; o++)
        for (int row = 0; row < H; row++)
            for (int col = 0; col < W; col++)
                if (masks[((size_t)o * H + row) * W + col] == 255) { areas[o].x.push_back((float)col); areas[o].y.push_back((float)row); }

    auto sem_edge = [&](const float* X, float ox, float oy) {
        GraphEdge e;
        memset(&e, 0, sizeof(e));
        e.pose = 0; e.point = -1; e.stereo = false;
        e.obs[0] = ox; e.obs[1] = oy;
        e.info = invSigma2_0;
        e.robust = true; e.delta = deltaMono; e.level = 0; e.semantic = true;
        for (int k = 0; k < 3; k++) e.Xw[k] = X[k];
        return e;
    };
    // M_joint constraints for the initial optimisation (:719-767)
    std::vector<int> initEdge, initObj;
    std::vector<uint8_t> initOutlier;
    for (int j = 0; j < nJoint; j++) {
        const int kp = joint_kp[j], o = joint_obj[j];
        int idx; float d2;
        if (areas[o].nearest(kp_uv[kp * 2], kp_uv[kp * 2 + 1], idx, d2)) {
            if (d2 < 1.0) continue;
            g.edges.push_back(sem_edge(Xw + 3 * kp, areas[o].x[idx], areas[o].y[idx]));
            initEdge.push_back((int)g.edges.size() - 1);
            initObj.push_back(o);
            initOutlier.push_back(0);
            nSemNum++;
        }
    }
    // regular edges (:800-905), identical to PoseOptimization
    std::vector<int> idx, regEdge;
    int nInitialCorrespondences = 0;
    for (int i = 0; i < N; i++) {
        if (!has_mp[i]) continue;
        nInitialCorrespondences++;
        outlier[i] = 0;
        GraphEdge e;
        memset(&e, 0, sizeof(e));
        e.pose = 0; e.point = -1;
        e.stereo = !(obs[i * 3 + 2] < 0);
        e.obs[0] = obs[i * 3]; e.obs[1] = obs[i * 3 + 1]; e.obs[2] = e.stereo ? obs[i * 3 + 2] : 0;
        e.info = invSigma2[i];
        e.robust = true;
        e.delta = e.stereo ? deltaStereo : deltaMono;
        for (int k = 0; k < 3; k++) e.Xw[k] = Xw[i * 3 + k];
        g.edges.push_back(e);
        regEdge.push_back((int)g.edges.size() - 1);
        idx.push_back(i);
    }
    *nSemNumOut = 0;
    if (nInitialCorrespondences < 3) { memcpy(Tcw_out, Tcw_in, 64); return 0; }

    const float chi2Mono[4] = {5.991, 5.991, 5.991, 5.991};
    const float chi2Stereo[4] = {7.815, 7.815, 7.815, 7.815};
    int nBad = 0;
    for (size_t it = 0; it < 4; it++) {
        g.poses[0] = se3_from_cvmat(Tcw_in);
        graph_optimize(g, 10, 0, nullptr);
        float Pose[16];
        se3_to_cvmat(g.poses[0], Pose);
        // re-gate the M_joint edges (:928-973 for it==0, :1042-1098 for it>=1)
        for (size_t i = 0; i < initEdge.size(); i++) {
            GraphEdge& e = g.edges[initEdge[i]];
            const float Pw[3] = {(float)e.Xw[0], (float)e.Xw[1], (float)e.Xw[2]};
            float Pc[3];
            project_f32(Pose, Pw, Pc);
            const float x = Pc[0] / Pc[2], y = Pc[1] / Pc[2];
            const float u = fx * x + cx, v = fy * y + cy;
            bool out;
            if (u < bounds[0] || v < bounds[1] || u > bounds[2] || v > bounds[3]) out = true;
            else {
                int ni; float d2;
                if (!areas[initObj[i]].nearest(u, v, ni, d2)) continue;   // nearestKSearch returned 0: edge untouched
                out = d2 > 10;
                if (!out) { e.obs[0] = u; e.obs[1] = v; }   // measurement := the projection itself (:969-971)
            }
            if (out) {
                e.level = 1;
                if (!initOutlier[i]) { initOutlier[i] = 1; nSemNum--; }
            } else {
                e.level = 0;
                if (initOutlier[i]) { initOutlier[i] = 0; nSemNum++; }
            }
        }
        if (it == 0) {
            // M_semantic constraints (:978-1032): every object map point whose projection is near the mask
            for (int m = 0; m < nObjMp; m++) {
                const int o = objmp_obj[m];
                float Pc[3];
                project_f32(Pose, objmp_Xw + 3 * m, Pc);
                const float x = Pc[0] / Pc[2], y = Pc[1] / Pc[2];
                const float u = fx * x + cx, v = fy * y + cy;
                int ni; float d2;
                if (areas[o].nearest(u, v, ni, d2) && d2 < 10) {
                    g.edges.push_back(sem_edge(objmp_Xw + 3 * m, areas[o].x[ni], areas[o].y[ni]));
                    nSemNum++;
                }
            }
        }
        nBad = 0;
        for (size_t k = 0; k < regEdge.size(); k++) {
            GraphEdge& e = g.edges[regEdge[k]];
            const int i = idx[k];
            if (outlier[i]) edge_compute_error(g, e);
            const float chi2 = edge_chi2(e);
            const float th = e.stereo ? chi2Stereo[it] : chi2Mono[it];
            if (chi2 > th) { outlier[i] = 1; e.level = 1; nBad++; }
            else { outlier[i] = 0; e.level = 0; }
            if (it == 2) e.robust = false;
        }
        if (g.edges.size() < 10) break;
    }
    se3_to_cvmat(g.poses[0], Tcw_out);
    *nSemNumOut = nSemNum;
    return nInitialCorrespondences - nBad;
}
}  // namespace oracle

namespace oracle {
// reference src/Optimizer.cc:49-237
void BundleAdjustment(int nKF, const float* poses, const uint8_t* fixed, int nP, const float* points, int nE,
                      const int32_t* edge_kf, const int32_t* edge_pt, const float* edge_obs, const float* edge_invSigma2,
                      const float* K5, int nIterations, int bRobust, float* poses_out, float* points_out) {
    Graph g;
    g.cam = Camera{K5[0], K5[1], K5[2], K5[3], K5[4]};
    for (int k = 0; k < nKF; k++) { g.poses.push_back(se3_from_cvmat(poses + 16 * k)); g.pose_fixed.push_back(fixed[k]); }
    for (int p = 0; p < nP; p++) g.points.push_back({(double)points[3 * p], (double)points[3 * p + 1], (double)points[3 * p + 2]});
    const float thHuber2D = sqrt(5.99);
    const float thHuber3D = sqrt(7.815);
    for (int i = 0; i < nE; i++) {
        GraphEdge e;
        memset(&e, 0, sizeof(e));
        e.pose = edge_kf[i];
        e.point = edge_pt[i];
        e.stereo = !(edge_obs[3 * i + 2] < 0);
        e.obs[0] = edge_obs[3 * i]; e.obs[1] = edge_obs[3 * i + 1]; e.obs[2] = e.stereo ? edge_obs[3 * i + 2] : 0;
        e.info = edge_invSigma2[i];
        e.robust = bRobust != 0;
        e.delta = e.stereo ? thHuber3D : thHuber2D;
        g.edges.push_back(e);
    }
    graph_optimize(g, nIterations, 0, nullptr);
    for (int k = 0; k < nKF; k++) se3_to_cvmat(g.poses[k], poses_out + 16 * k);
    for (int p = 0; p < nP; p++)
        for (int c = 0; c < 3; c++) points_out[3 * p + c] = (float)g.points[p][c];
}
}  // namespace oracle
