// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_common.h).  PARITY UNPINNED.
// CPU restatement of ORB_SLAM2::ORBextractor (reference src/ORBextractor.cc).
#pragma once
#include "oracle_common.h"

namespace oracle {

class OrbExtractor {
public:
    // reference src/ORBextractor.cc:410-470
    OrbExtractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);

    // reference src/ORBextractor.cc:1043-1105 (operator()).  Returns 0, or -1 on empty image.
    int extract(const uint8_t* img, int w, int h, int stride, std::vector<KeyPoint>& kps,
                std::vector<uint8_t>& desc);

    int nfeatures;
    double scaleFactor;  // the reference stores the ctor's float in a double member (ORBextractor.h:96)
    int nlevels, iniThFAST, minThFAST;
    // Model cv::GaussianBlur's x86 SSE2 column pass (float accumulate + cvtps2dq, half-even)
    // for columns x < (w & ~3) and the scalar fixed-point tail ((v + 2^15) >> 16, half-up)
    // for the rest.  false = scalar rule everywhere.
    bool blur_sse2_rounding = true;

    std::vector<int> mnFeaturesPerLevel, umax;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    std::vector<Image> mvImagePyramid;  // un-padded level images (the ROI the reference exposes)

    // intermediates kept for stage-by-stage parity tests
    std::vector<Image> blurred;                      // per level (only levels with keypoints, like the reference)
    std::vector<std::vector<KeyPoint>> candidates;   // FAST output per level, region coords (before octree)
    std::vector<std::vector<KeyPoint>> levelKeys;    // after octree + orientation, level coords (before *scale)

    void ComputePyramid(const uint8_t* img, int w, int h, int stride);  // :1107-1132
    void ComputeKeyPointsOctTree();                                    // :765-853
    std::vector<KeyPoint> DistributeOctTree(const std::vector<KeyPoint>& vToDistributeKeys, int minX,
                                            int maxX, int minY, int maxY, int N);  // :539-763
};

// ---- restated OpenCV 3.2 primitives (exposed for unit tests) ----
// cv::FAST(img, kps, threshold, nonmaxSuppression=true, TYPE_9_16) on a ROI.
void fast_9_16(const uint8_t* roi, int stride, int cols, int rows, int threshold, bool nms,
               std::vector<KeyPoint>& out);
// cornerScore<16>: threshold-independent for pixels that pass (max arc strength - 1).
int fast_corner_score(const uint8_t* p, int stride, int threshold);
// cv::resize(..., INTER_LINEAR) CV_8UC1.
void resize_linear_u8(const Image& src, Image& dst);
// cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) CV_8UC1.
void gaussian_blur_7x7_s2(const Image& src, Image& dst, bool sse2_rounding);
// cv::fastAtan2 (degrees).
float fastAtan2(float y, float x);
// IC_Angle, reference src/ORBextractor.cc:77-104.
float IC_Angle(const Image& image, float ptx, float pty, const std::vector<int>& u_max);
// computeOrbDescriptor, reference src/ORBextractor.cc:108-147.
void computeOrbDescriptor(const KeyPoint& kpt, const Image& img, uint8_t* desc);
const int8_t* brief_pattern();  // 1024 int8 (x0,y0,x1,y1)*256

}  // namespace oracle
