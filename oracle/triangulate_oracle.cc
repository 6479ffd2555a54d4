// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_common.h).  PARITY UNPINNED: cv::Mat expression arithmetic and
// cv::SVD (OpenCV 3.2, absent from /root/reference) are restated from the published implementation.
// LocalMapping::CreateNewMapPoints, per-match numeric core (reference src/LocalMapping.cc:291-432) and
// KeyFrame::UnprojectStereo (src/KeyFrame.cc:615-631), on flat arrays.  SURVEY.md §8(f)-3.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>

#include "oracle_common.h"

namespace oracle {

// One keyframe as CreateNewMapPoints reads it.
struct TriKF {
    const float* Tcw;          // 4x4 row-major (GetRotation / GetTranslation)
    const float* Twc;          // 4x4 row-major (Rwc = Rcw.t(), Ow = GetCameraCenter)
    float fx, fy, cx, cy, invfx, invfy, mbf, mb;
    const KeyPoint* keysUn;    // mvKeysUn
    const KeyPoint* keys;      // mvKeys (UnprojectStereo uses the raw keypoint, src/KeyFrame.cc:620-621)
    const float* uRight;       // mvuRight
    const float* depth;        // mvDepth
};

// cv::SVD::compute on a 4x4 CV_32F matrix -> vt (OpenCV 3.2 modules/core/src/lapack.cpp JacobiSVDImpl_<float>:
// one-sided Jacobi on the rows of A^T, fp64 norms W, float rotations, eps = FLT_EPSILON*2, max_iter = max(m,30),
// then selection sort by decreasing singular value with row swaps of Vt).  A is row-major; Vt row-major out.
static void JacobiSVD4_vt(const float A[16], float Vt[16]) {
    const int m = 4, n = 4;
    float At[16];
    for (int i = 0; i < 4; i++)
        for (int k = 0; k < 4; k++) At[i * 4 + k] = A[k * 4 + i];   // transpose(src, temp_a)
    double W[4];
    const float eps = FLT_EPSILON * 2;
    for (int i = 0; i < n; i++) {
        double sd = 0;
        for (int k = 0; k < m; k++) { const float t = At[i * 4 + k]; sd += (double)t * t; }
        W[i] = sd;
        for (int k = 0; k < n; k++) Vt[i * 4 + k] = 0;
        Vt[i * 4 + i] = 1;
    }
    const int max_iter = std::max(m, 30);
    for (int iter = 0; iter < max_iter; iter++) {
        bool changed = false;
        for (int i = 0; i < n - 1; i++)
            for (int j = i + 1; j < n; j++) {
                float *Ai = At + i * 4, *Aj = At + j * 4;
                double a = W[i], p = 0, b = W[j];
                for (int k = 0; k < m; k++) p += (double)Ai[k] * Aj[k];
                if (std::abs(p) <= eps * std::sqrt((double)a * b)) continue;
                p *= 2;
                const double beta = a - b, gamma = hypot((double)p, beta);
                float c, s;
                if (beta < 0) {
                    const double delta = (gamma - beta) * 0.5;
                    s = (float)std::sqrt(delta / gamma);
                    c = (float)(p / (gamma * s * 2));
                } else {
                    c = (float)std::sqrt((gamma + beta) / (gamma * 2));
                    s = (float)(p / (gamma * c * 2));
                }
                a = b = 0;
                for (int k = 0; k < m; k++) {
                    const float t0 = c * Ai[k] + s * Aj[k];
                    const float t1 = -s * Ai[k] + c * Aj[k];
                    Ai[k] = t0; Aj[k] = t1;
                    a += (double)t0 * t0; b += (double)t1 * t1;
                }
                W[i] = a; W[j] = b;
                changed = true;
                float *Vi = Vt + i * 4, *Vj = Vt + j * 4;
                for (int k = 0; k < n; k++) {
                    const float t0 = c * Vi[k] + s * Vj[k];
                    const float t1 = -s * Vi[k] + c * Vj[k];
                    Vi[k] = t0; Vj[k] = t1;
                }
            }
        if (!changed) break;
    }
    for (int i = 0; i < n; i++) {
        double sd = 0;
        for (int k = 0; k < m; k++) { const float t = At[i * 4 + k]; sd += (double)t * t; }
        W[i] = std::sqrt(sd);
    }
    for (int i = 0; i < n - 1; i++) {
        int j = i;
        for (int k = i + 1; k < n; k++)
            if (W[j] < W[k]) j = k;
        if (i != j) {
            std::swap(W[i], W[j]);
            for (int k = 0; k < m; k++) std::swap(At[i * 4 + k], At[j * 4 + k]);
            for (int k = 0; k < n; k++) std::swap(Vt[i * 4 + k], Vt[j * 4 + k]);
        }
    }
}

// Mat(1x3 or 3x1 float).dot -> double (cv::dotProd_32f accumulates in double)
static inline double dot3(const float* a, const float* b) {
    double r = 0;
    for (int k = 0; k < 3; k++) r += (double)a[k] * b[k];
    return r;
}
static inline double norm3(const float* v) { return std::sqrt(dot3(v, v)); }
// R(3x3) * x(3x1) [+ c]: cv::gemm small-matrix branch, float accumulation (see matcher_oracle.cc gemm_row)
static inline float gemm3(const float* row, const float* x, float c, bool has_c) {
    const float t0 = row[0] * x[0] + row[1] * x[1] + row[2] * x[2];
    return has_c ? (float)((double)t0 * 1.0 + (double)c * 1.0) : (float)((double)t0 * 1.0 + 0.0 * 0.0);
}

// Reprojection gate of one view (:352-378 / :380-406). mbf is ALWAYS the current keyframe's (reference :399).
static bool reproj_ok(const TriKF& kf, float mbf, const KeyPoint& kp, float kp_ur, bool bStereo, float x, float y, float z, float sigmaSquare) {
    const float invz = 1.0 / z;
    if (!bStereo) {
        const float u = kf.fx * x * invz + kf.cx;
        const float v = kf.fy * y * invz + kf.cy;
        const float errX = u - kp.x, errY = v - kp.y;
        if ((errX * errX + errY * errY) > 5.991 * sigmaSquare) return false;
    } else {
        const float u = kf.fx * x * invz + kf.cx;
        const float u_r = u - mbf * invz;
        const float v = kf.fy * y * invz + kf.cy;
        const float errX = u - kp.x, errY = v - kp.y, errX_r = u_r - kp_ur;
        if ((errX * errX + errY * errY + errX_r * errX_r) > 7.8 * sigmaSquare) return false;
    }
    return true;
}

// KeyFrame::UnprojectStereo, src/KeyFrame.cc:615-631 (z <= 0 returns an empty Mat there; the caller below
// would then throw inside Mat::dot — unreachable because mvuRight >= 0 implies mvDepth > 0; we reject).
static bool UnprojectStereo(const TriKF& kf, int i, float out[3]) {
    const float z = kf.depth[i];
    if (!(z > 0)) return false;
    const float u = kf.keys[i].x, v = kf.keys[i].y;
    const float x = (u - kf.cx) * z * kf.invfx;
    const float y = (v - kf.cy) * z * kf.invfy;
    const float xc[3] = {x, y, z};
    for (int r = 0; r < 3; r++) out[r] = gemm3(kf.Twc + r * 4, xc, kf.Twc[r * 4 + 3], true);
    return true;
}

// src/LocalMapping.cc:291-432 for M matches (idx1 in kf1 = mpCurrentKeyFrame, idx2 in kf2 = pKF2).
// ok[m] = 1 and x3D[m] = the new map point position when every gate passes.
int TriangulateMatches(const TriKF& kf1, const TriKF& kf2, int M, const int32_t* idx1, const int32_t* idx2, const float* scaleFactors,
                       const float* levelSigma2, float ratioFactor, uint8_t* ok, float* x3D_out) {
    const float* T1 = kf1.Tcw; const float* T2 = kf2.Tcw;
    const float Ow1[3] = {kf1.Twc[3], kf1.Twc[7], kf1.Twc[11]}, Ow2[3] = {kf2.Twc[3], kf2.Twc[7], kf2.Twc[11]};
    int nnew = 0;
    for (int ikp = 0; ikp < M; ikp++) {
        ok[ikp] = 0;
        x3D_out[3 * ikp] = x3D_out[3 * ikp + 1] = x3D_out[3 * ikp + 2] = 0;
        const int i1 = idx1[ikp], i2 = idx2[ikp];
        const KeyPoint& kp1 = kf1.keysUn[i1];
        const float kp1_ur = kf1.uRight[i1];
        const bool bStereo1 = kp1_ur >= 0;
        const KeyPoint& kp2 = kf2.keysUn[i2];
        const float kp2_ur = kf2.uRight[i2];
        const bool bStereo2 = kp2_ur >= 0;

        const float xn1[3] = {(kp1.x - kf1.cx) * kf1.invfx, (kp1.y - kf1.cy) * kf1.invfy, 1.0f};
        const float xn2[3] = {(kp2.x - kf2.cx) * kf2.invfx, (kp2.y - kf2.cy) * kf2.invfy, 1.0f};
        float ray1[3], ray2[3];
        for (int r = 0; r < 3; r++) { ray1[r] = gemm3(kf1.Twc + r * 4, xn1, 0, false); ray2[r] = gemm3(kf2.Twc + r * 4, xn2, 0, false); }
        const float cosParallaxRays = dot3(ray1, ray2) / (norm3(ray1) * norm3(ray2));

        float cosParallaxStereo = cosParallaxRays + 1;
        float cosParallaxStereo1 = cosParallaxStereo;
        float cosParallaxStereo2 = cosParallaxStereo;
        // float overloads: the reference TU is under `using namespace std` (atan2f, cosf)
        if (bStereo1) cosParallaxStereo1 = std::cos(2 * std::atan2(kf1.mb / 2, kf1.depth[i1]));
        else if (bStereo2) cosParallaxStereo2 = std::cos(2 * std::atan2(kf2.mb / 2, kf2.depth[i2]));
        cosParallaxStereo = std::min(cosParallaxStereo1, cosParallaxStereo2);

        float x3D[3];
        if (cosParallaxRays < cosParallaxStereo && cosParallaxRays > 0 && (bStereo1 || bStereo2 || cosParallaxRays < 0.9998)) {
            // A.row(r) = xn*Tcw.row(2) - Tcw.row(k): MatExpr AddEx(alpha = xn, beta = -1) -> cv::addWeighted,
            // 32f works in double: (float)(a*alpha + b*beta + 0)
            float A[16];
            for (int k = 0; k < 4; k++) {
                A[0 * 4 + k] = (float)((double)T1[8 + k] * (double)xn1[0] + (double)T1[0 + k] * -1.0 + 0.0);
                A[1 * 4 + k] = (float)((double)T1[8 + k] * (double)xn1[1] + (double)T1[4 + k] * -1.0 + 0.0);
                A[2 * 4 + k] = (float)((double)T2[8 + k] * (double)xn2[0] + (double)T2[0 + k] * -1.0 + 0.0);
                A[3 * 4 + k] = (float)((double)T2[8 + k] * (double)xn2[1] + (double)T2[4 + k] * -1.0 + 0.0);
            }
            float Vt[16];
            JacobiSVD4_vt(A, Vt);
            const float w = Vt[15];
            if (w == 0) continue;
            // x3D.rowRange(0,3)/w -> convertTo(alpha = 1/w): cvtScale 32f->32f in float
            const float inv = (float)(1.0 / (double)w);
            for (int k = 0; k < 3; k++) x3D[k] = Vt[12 + k] * inv + 0.0f;
        } else if (bStereo1 && cosParallaxStereo1 < cosParallaxStereo2) {
            if (!UnprojectStereo(kf1, i1, x3D)) continue;
        } else if (bStereo2 && cosParallaxStereo2 < cosParallaxStereo1) {
            if (!UnprojectStereo(kf2, i2, x3D)) continue;
        } else
            continue;

        const float z1 = dot3(T1 + 8, x3D) + T1[11];
        if (z1 <= 0) continue;
        const float z2 = dot3(T2 + 8, x3D) + T2[11];
        if (z2 <= 0) continue;

        const float sigmaSquare1 = levelSigma2[kp1.octave];
        const float x1 = dot3(T1, x3D) + T1[3];
        const float y1 = dot3(T1 + 4, x3D) + T1[7];
        if (!reproj_ok(kf1, kf1.mbf, kp1, kp1_ur, bStereo1, x1, y1, z1, sigmaSquare1)) continue;

        const float sigmaSquare2 = levelSigma2[kp2.octave];
        const float x2 = dot3(T2, x3D) + T2[3];
        const float y2 = dot3(T2 + 4, x3D) + T2[7];
        if (!reproj_ok(kf2, kf1.mbf, kp2, kp2_ur, bStereo2, x2, y2, z2, sigmaSquare2)) continue;

        const float n1[3] = {x3D[0] - Ow1[0], x3D[1] - Ow1[1], x3D[2] - Ow1[2]};
        const float dist1 = norm3(n1);
        const float n2[3] = {x3D[0] - Ow2[0], x3D[1] - Ow2[1], x3D[2] - Ow2[2]};
        const float dist2 = norm3(n2);
        if (dist1 == 0 || dist2 == 0) continue;
        const float ratioDist = dist2 / dist1;
        const float ratioOctave = scaleFactors[kp1.octave] / scaleFactors[kp2.octave];
        if (ratioDist * ratioFactor < ratioOctave || ratioDist > ratioOctave * ratioFactor) continue;

        ok[ikp] = 1;
        for (int k = 0; k < 3; k++) x3D_out[3 * ikp + k] = x3D[k];
        nnew++;
    }
    return nnew;
}

}  // namespace oracle

extern "C" {
// kfN: {Tcw[16], Twc[16], fx, fy, cx, cy, invfx, invfy, mbf, mb} = 40 floats
int oo_triangulate(const float* kf1p, const oracle::KeyPoint* keysUn1, const oracle::KeyPoint* keys1, const float* uR1, const float* depth1,
                   const float* kf2p, const oracle::KeyPoint* keysUn2, const oracle::KeyPoint* keys2, const float* uR2, const float* depth2, int M,
                   const int32_t* idx1, const int32_t* idx2, const float* scaleFactors, const float* levelSigma2, float ratioFactor, uint8_t* ok,
                   float* x3D) {
    auto mk = [](const float* p, const oracle::KeyPoint* ku, const oracle::KeyPoint* k, const float* ur, const float* d) {
        oracle::TriKF kf;
        kf.Tcw = p; kf.Twc = p + 16;
        kf.fx = p[32]; kf.fy = p[33]; kf.cx = p[34]; kf.cy = p[35]; kf.invfx = p[36]; kf.invfy = p[37]; kf.mbf = p[38]; kf.mb = p[39];
        kf.keysUn = ku; kf.keys = k; kf.uRight = ur; kf.depth = d;
        return kf;
    };
    return oracle::TriangulateMatches(mk(kf1p, keysUn1, keys1, uR1, depth1), mk(kf2p, keysUn2, keys2, uR2, depth2), M, idx1, idx2, scaleFactors,
                                      levelSigma2, ratioFactor, ok, x3D);
}
void oo_svd4_vt(const float* A, float* Vt) { oracle::JacobiSVD4_vt(A, Vt); }
}
