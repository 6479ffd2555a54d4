// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_common.h).  PARITY UNPINNED: cv::undistortPoints (OpenCV 3.2
// modules/imgproc/src/undistort.cpp, cvUndistortPoints) is absent from /root/reference and restated here.
// Frame::UndistortKeyPoints (reference src/Frame.cc:644-675), Frame::ComputeImageBounds (:677-704) and
// Frame::ComputeStereoFromRGBD (:883-904) on flat arrays.
#include <algorithm>
#include <cstring>
#include <vector>

#include "oracle_common.h"

namespace oracle {

// cvUndistortPoints(src, dst, K, distCoeffs, R = I, P = K) for 2-channel float points: fp64 inside, 5 fixed-point
// iterations when distortion coefficients are given.  K: fx, fy, cx, cy (CV_32F in the reference, widened to double);
// dist: k1, k2, p1, p2[, k3] (CV_32F, 4 or 5 entries).
void UndistortPoints(int n, const float* src_xy, const float K4[4], const float* dist, int ndist, float* dst_xy) {
    double k[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < ndist && i < 12; i++) k[i] = (double)dist[i];
    const int iters = ndist > 0 ? 5 : 1;
    const double fx = (double)K4[0], fy = (double)K4[1], cx = (double)K4[2], cy = (double)K4[3];
    const double ifx = 1. / fx, ify = 1. / fy;
    // RR = PP * I with PP = K
    const double RR[3][3] = {{fx, 0, cx}, {0, fy, cy}, {0, 0, 1}};
    for (int i = 0; i < n; i++) {
        double x = (double)src_xy[2 * i], y = (double)src_xy[2 * i + 1], x0, y0;
        x0 = x = (x - cx) * ifx;
        y0 = y = (y - cy) * ify;
        for (int j = 0; j < iters; j++) {
            const double r2 = x * x + y * y;
            const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
            const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        const double xx = RR[0][0] * x + RR[0][1] * y + RR[0][2];
        const double yy = RR[1][0] * x + RR[1][1] * y + RR[1][2];
        const double ww = 1. / (RR[2][0] * x + RR[2][1] * y + RR[2][2]);
        x = xx * ww;
        y = yy * ww;
        dst_xy[2 * i] = (float)x;
        dst_xy[2 * i + 1] = (float)y;
    }
}

// Frame::UndistortKeyPoints, src/Frame.cc:644-675
void UndistortKeyPoints(int n, const KeyPoint* keys, const float K4[4], const float* dist, int ndist, KeyPoint* keysUn) {
    if (n > 0) memcpy(keysUn, keys, (size_t)n * sizeof(KeyPoint));
    if (ndist <= 0 || dist[0] == 0.0f) return;   // mDistCoef.at<float>(0)==0.0
    std::vector<float> xy((size_t)2 * n), out((size_t)2 * n);
    for (int i = 0; i < n; i++) { xy[2 * i] = keys[i].x; xy[2 * i + 1] = keys[i].y; }
    UndistortPoints(n, xy.data(), K4, dist, ndist, out.data());
    for (int i = 0; i < n; i++) { keysUn[i].x = out[2 * i]; keysUn[i].y = out[2 * i + 1]; }
}

// Frame::ComputeImageBounds, src/Frame.cc:677-704 -> mnMinX, mnMinY, mnMaxX, mnMaxY
void ComputeImageBounds(int cols, int rows, const float K4[4], const float* dist, int ndist, float bounds[4]) {
    if (ndist > 0 && dist[0] != 0.0f) {
        const float c[8] = {0.f, 0.f, (float)cols, 0.f, 0.f, (float)rows, (float)cols, (float)rows};
        float u[8];
        UndistortPoints(4, c, K4, dist, ndist, u);
        bounds[0] = std::min(u[0], u[4]);
        bounds[2] = std::max(u[2], u[6]);
        bounds[1] = std::min(u[1], u[3]);
        bounds[3] = std::max(u[5], u[7]);
    } else {
        bounds[0] = 0.0f; bounds[2] = (float)cols; bounds[1] = 0.0f; bounds[3] = (float)rows;
    }
}

// Frame::ComputeStereoFromRGBD, src/Frame.cc:883-904.  depth: rows x pitch floats (already scaled by mDepthMapFactor).
void ComputeStereoFromRGBD(int n, const KeyPoint* keys, const KeyPoint* keysUn, const float* depth, int pitch, float mbf, float* uRight,
                           float* mvDepth) {
    for (int i = 0; i < n; i++) {
        uRight[i] = -1;
        mvDepth[i] = -1;
        const float v = keys[i].y, u = keys[i].x;
        const float d = depth[(size_t)(int)v * pitch + (int)u];   // Mat::at<float>(int row, int col): floats truncate
        if (d > 0) {
            mvDepth[i] = d;
            uRight[i] = keysUn[i].x - mbf / d;
        }
    }
}

}  // namespace oracle

extern "C" {
void oo_undistort_keypoints(int n, const oracle::KeyPoint* keys, const float* K4, const float* dist, int ndist, oracle::KeyPoint* out) {
    oracle::UndistortKeyPoints(n, keys, K4, dist, ndist, out);
}
void oo_image_bounds(int cols, int rows, const float* K4, const float* dist, int ndist, float* bounds) { oracle::ComputeImageBounds(cols, rows, K4, dist, ndist, bounds); }
void oo_stereo_from_rgbd(int n, const oracle::KeyPoint* keys, const oracle::KeyPoint* keysUn, const float* depth, int pitch, float mbf, float* uRight,
                         float* mvDepth) {
    oracle::ComputeStereoFromRGBD(n, keys, keysUn, depth, pitch, mbf, uRight, mvDepth);
}
}
