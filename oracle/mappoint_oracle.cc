// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_common.h).  PARITY UNPINNED for the cv::Mat arithmetic.
// MapPoint maintenance (reference src/MapPoint.cc:345-521) and Frame::isInFrustum (src/Frame.cc:509-565)
// on flat arrays — the rows SURVEY.md §8(f)-2 places right before SearchByProjection and after local BA.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <vector>

#include "matcher_oracle.h"

namespace oracle {

// MapPoint::ComputeDistinctiveDescriptors, src/MapPoint.cc:345-410. desc: N x 32 in observation order.
int ComputeDistinctiveDescriptor(int N, const uint8_t* desc) {
    if (N <= 0) return -1;
    std::vector<float> D((size_t)N * N);
    for (int i = 0; i < N; i++) {
        D[(size_t)i * N + i] = 0;
        for (int j = i + 1; j < N; j++) {
            const int d = DescriptorDistance(desc + (size_t)i * 32, desc + (size_t)j * 32);
            D[(size_t)i * N + j] = d;
            D[(size_t)j * N + i] = d;
        }
    }
    int BestMedian = INT_MAX, BestIdx = 0;
    for (int i = 0; i < N; i++) {
        std::vector<int> v(D.begin() + (size_t)i * N, D.begin() + (size_t)(i + 1) * N);
        std::sort(v.begin(), v.end());
        const int median = v[0.5 * (N - 1)];
        if (median < BestMedian) { BestMedian = median; BestIdx = i; }
    }
    return BestIdx;
}

static inline double norm3(const float* v) { return std::sqrt((double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2]); }

// MapPoint::UpdateNormalAndDepth, src/MapPoint.cc:433-474.  Ow: camera centres of the observing keyframes
// (n x 3, observation order), OwRef: centre of mpRefKF.  out = {normal[3], mfMaxDistance, mfMinDistance}.
void UpdateNormalAndDepth(const float* Pos, int n, const float* Ow, const float* OwRef, float levelScaleFactor, float lastScaleFactor,
                          float* out) {
    float normal[3] = {0, 0, 0};
    for (int i = 0; i < n; i++) {
        float ni[3] = {Pos[0] - Ow[3 * i], Pos[1] - Ow[3 * i + 1], Pos[2] - Ow[3 * i + 2]};
        // normal + normali/norm: the MatExpr folds to AddEx(a=normali, alpha=1/norm, b=normal, beta=1), evaluated by
        // cv::scaleAdd, which for CV_32F narrows alpha to float: dst = src1*falpha + src2
        const float inv = (float)(1.0 / norm3(ni));
        for (int k = 0; k < 3; k++) normal[k] = ni[k] * inv + normal[k];
    }
    float PC[3] = {Pos[0] - OwRef[0], Pos[1] - OwRef[1], Pos[2] - OwRef[2]};
    const float dist = (float)norm3(PC);
    const float maxD = dist * levelScaleFactor;
    // normal/n -> convertTo(alpha = 1/n): cvtScale 32f->32f works in float: src*(float)alpha + 0.f
    const float invn = (float)(1.0 / (double)n);
    for (int k = 0; k < 3; k++) out[k] = normal[k] * invn + 0.0f;
    out[3] = maxD;
    out[4] = maxD / lastScaleFactor;
}

// Frame::isInFrustum (src/Frame.cc:509-565) + the query fields SearchByProjection reads (src/ORBmatcher.cc:57-67).
void IsInFrustum(int M, const float* Pw, const float* Pn, const float* maxDist, const float* minDist, const uint8_t* obs_gt0,
                 const uint8_t* mp_desc, const float* Tcw, const float* K5, const float* bounds, float viewingCosLimit,
                 float logScaleFactor, const float* scaleFactors, int nLevels, float th, ProjQuery* out) {
    float Ow[3];
    for (int r = 0; r < 3; r++) {   // mOw = -Rcw^T tcw (src/Frame.cc:447-453: one gemm, alpha = -1)
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)Tcw[k * 4 + r] * (double)Tcw[k * 4 + 3];
        Ow[r] = (float)(-1.0 * s);
    }
    const float fx = K5[0], fy = K5[1], cx = K5[2], cy = K5[3], bf = K5[4];
    for (int i = 0; i < M; i++) {
        ProjQuery& q = out[i];
        memset(&q, 0, sizeof(q));
        q.minLevel = q.maxLevel = -1;
        const float* P = Pw + 3 * i;
        float Pc[3];
        for (int r = 0; r < 3; r++) {   // Rcw*P+tcw: cv::gemm small-matrix branch (float accumulate), see matcher_oracle.cc gemm_row
            const float t0 = Tcw[r * 4] * P[0] + Tcw[r * 4 + 1] * P[1] + Tcw[r * 4 + 2] * P[2];
            Pc[r] = (float)((double)t0 * 1.0 + (double)Tcw[r * 4 + 3] * 1.0);
        }
        if (Pc[2] < 0.0f) continue;
        const float invz = 1.0f / Pc[2];
        const float u = fx * Pc[0] * invz + cx;
        const float v = fy * Pc[1] * invz + cy;
        if (u < bounds[0] || u > bounds[2]) continue;
        if (v < bounds[1] || v > bounds[3]) continue;
        const float maxDistance = 1.2f * maxDist[i];   // GetMaxDistanceInvariance
        const float minDistance = 0.8f * minDist[i];
        const float PO[3] = {P[0] - Ow[0], P[1] - Ow[1], P[2] - Ow[2]};
        const float dist = (float)norm3(PO);
        if (dist < minDistance || dist > maxDistance) continue;
        const float* n = Pn + 3 * i;
        const float viewCos = ((double)PO[0] * n[0] + (double)PO[1] * n[1] + (double)PO[2] * n[2]) / dist;
        if (viewCos < viewingCosLimit) continue;
        const float ratio = maxDist[i] / dist;          // PredictScale, src/MapPoint.cc:505-521
        int nScale = std::ceil(std::log(ratio) / logScaleFactor);   // float overloads (the reference TU has `using namespace std`)
        if (nScale < 0) nScale = 0;
        else if (nScale >= nLevels) nScale = nLevels - 1;
        float r = viewCos > 0.998 ? 2.5 : 4.0;         // RadiusByViewingCos
        if (th != 1.0) r *= th;
        q.u = u; q.v = v; q.ur = u - bf * invz;
        q.radius = r * scaleFactors[nScale];
        q.minLevel = nScale - 1; q.maxLevel = nScale;
        q.flags = 1 | (obs_gt0[i] ? 2 : 0);
        q.angle = viewCos;   // mTrackViewCos kept for inspection (SearchByProjection(F,...) does not use the angle)
        memcpy(q.desc, mp_desc + (size_t)i * 32, 32);
    }
}

}  // namespace oracle

extern "C" {
int oo_distinctive_descriptor(int N, const uint8_t* desc) { return oracle::ComputeDistinctiveDescriptor(N, desc); }
void oo_update_normal_depth(const float* Pos, int n, const float* Ow, const float* OwRef, float lsf, float last, float* out) {
    oracle::UpdateNormalAndDepth(Pos, n, Ow, OwRef, lsf, last, out);
}
void oo_is_in_frustum(int M, const float* Pw, const float* Pn, const float* maxDist, const float* minDist, const uint8_t* obs_gt0,
                      const uint8_t* mp_desc, const float* Tcw, const float* K5, const float* bounds, float viewingCosLimit,
                      float logScaleFactor, const float* scaleFactors, int nLevels, float th, oracle::ProjQuery* out) {
    oracle::IsInFrustum(M, Pw, Pn, maxDist, minDist, obs_gt0, mp_desc, Tcw, K5, bounds, viewingCosLimit, logScaleFactor, scaleFactors, nLevels, th, out);
}
}
