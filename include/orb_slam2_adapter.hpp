// orb_slam2_adapter.hpp — header-only C++ adapter that re-exposes the reference's class API
// (ORB_SLAM2::ORBextractor, include/ORBextractor.h:45-110; the projection searches of
// ORB_SLAM2::ORBmatcher, include/ORBmatcher.h:41-83) on top of the C ABI in oslam_hip.h.
//
// It is written against POD mirrors of cv::KeyPoint / cv::Mat so it compiles without OpenCV
// (OpenCV is not installed in the build image).  In the reference tree, define
// OSLAM_ADAPTER_USE_OPENCV before including it: oslam::KeyPoint becomes cv::KeyPoint (identical
// 28-byte layout) and descriptors are cv::Mat CV_8U N x 32.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "oslam_hip.h"

#ifdef OSLAM_ADAPTER_USE_OPENCV
#include <opencv2/core/core.hpp>
#endif

namespace oslam {

#ifdef OSLAM_ADAPTER_USE_OPENCV
typedef cv::KeyPoint KeyPoint;
static_assert(sizeof(cv::KeyPoint) == sizeof(oslam_keypoint_t), "cv::KeyPoint layout");
#else
typedef oslam_keypoint_t KeyPoint;
#endif

struct Image8 {   // CV_8UC1 view
    const uint8_t* data; int cols, rows, step;
    bool empty() const { return !data || cols == 0 || rows == 0; }
};

inline void throw_on(int rc) {
    if (rc != OSLAM_OK) throw std::runtime_error(std::string("oslam: ") + oslam_last_error());
}

}  // namespace oslam

namespace ORB_SLAM2 {

// Same constructor arguments and getters as the reference class; the image geometry is bound at
// the first call (the reference sizes its pyramid per call, src/ORBextractor.cc:1111-1116).
class ORBextractor {
public:
    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int device = 0)
        : nfeatures_(nfeatures), scaleFactor_(scaleFactor), nlevels_(nlevels), iniTh_(iniThFAST), minTh_(minThFAST),
          device_(device) {}
    ~ORBextractor() { oslam_orb_destroy(h_); }
    ORBextractor(const ORBextractor&) = delete;
    ORBextractor& operator=(const ORBextractor&) = delete;

    // void operator()(InputArray image, InputArray mask /*ignored*/, vector<KeyPoint>&, OutputArray descriptors)
    void operator()(const oslam::Image8& image, std::vector<oslam::KeyPoint>& keypoints,
                    std::vector<uint8_t>& descriptors /* N x 32 row-major */) {
        keypoints.clear();
        descriptors.clear();
        if (image.empty()) return;   // reference: silent return, src/ORBextractor.cc:1046
        ensure(image.cols, image.rows);
        const int cap = oslam_orb_max_keypoints(h_);
        keypoints.resize(cap);
        descriptors.resize((size_t)cap * 32);
        int n = 0;
        oslam::throw_on(oslam_orb_extract(h_, image.data, image.cols, image.rows, image.step,
                                          reinterpret_cast<oslam_keypoint_t*>(keypoints.data()), descriptors.data(), cap, &n));
        keypoints.resize(n);
        descriptors.resize((size_t)n * 32);
    }

    int GetLevels() { return nlevels_; }
    float GetScaleFactor() { return scaleFactor_; }
    std::vector<float> GetScaleFactors() { return table(0); }
    std::vector<float> GetInverseScaleFactors() { return table(1); }
    std::vector<float> GetScaleSigmaSquares() { return table(2); }
    std::vector<float> GetInverseScaleSigmaSquares() { return table(3); }

    // mvImagePyramid[level] (public member in the reference, include/ORBextractor.h:85)
    std::vector<uint8_t> ImagePyramidLevel(int level, int& cols, int& rows) {
        oslam::throw_on(oslam_orb_level_size(h_, level, &cols, &rows));
        std::vector<uint8_t> out((size_t)cols * rows);
        oslam::throw_on(oslam_orb_get_pyramid_level(h_, 0, level, out.data()));
        return out;
    }
    oslam_orb_t* handle() { return h_; }

private:
    void ensure(int w, int hgt) {
        if (h_ && w == w_ && hgt == hgt_) return;
        oslam_orb_destroy(h_);
        h_ = nullptr;
        oslam::throw_on(oslam_orb_create(&h_, nfeatures_, scaleFactor_, nlevels_, iniTh_, minTh_, w, hgt, 1, device_));
        w_ = w; hgt_ = hgt;
    }
    std::vector<float> table(int which) {
        // tables depend only on the ctor arguments; a 64x64 probe handle is enough before the first image
        oslam_orb_t* h = h_;
        bool tmp = false;
        if (!h) { oslam::throw_on(oslam_orb_create(&h, nfeatures_, scaleFactor_, 1, iniTh_, minTh_, 64, 64, 1, device_)); tmp = true; }
        std::vector<float> t[4];
        for (auto& v : t) v.resize(OSLAM_MAX_LEVELS);
        if (tmp) {   // recompute the float chain for all levels (src/ORBextractor.cc:415-432)
            oslam_orb_destroy(h);
            std::vector<float> s(nlevels_), s2(nlevels_), is(nlevels_), is2(nlevels_);
            s[0] = 1.f; s2[0] = 1.f;
            const double sf = scaleFactor_;
            for (int i = 1; i < nlevels_; i++) { s[i] = (float)(s[i - 1] * sf); s2[i] = s[i] * s[i]; }
            for (int i = 0; i < nlevels_; i++) { is[i] = 1.0f / s[i]; is2[i] = 1.0f / s2[i]; }
            return which == 0 ? s : which == 1 ? is : which == 2 ? s2 : is2;
        }
        oslam::throw_on(oslam_orb_get_scale_tables(h, t[0].data(), t[1].data(), t[2].data(), t[3].data(), nullptr));
        t[which].resize(nlevels_);
        return t[which];
    }
    int nfeatures_; float scaleFactor_; int nlevels_, iniTh_, minTh_, device_;
    oslam_orb_t* h_ = nullptr;
    int w_ = 0, hgt_ = 0;
};

// Flat view of the Frame members the projection searches read/write (include/Frame.h).
struct FrameView {
    int N;
    const oslam::KeyPoint* mvKeysUn;
    const float* mvuRight;
    const uint8_t* mDescriptors;          // N x 32
    const uint8_t* blocked;               // mvpMapPoints[i] && mvpMapPoints[i]->Observations()>0
    float mnMinX, mnMinY, mnMaxX, mnMaxY;
};

class ORBmatcher {
public:
    static const int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30;   // src/ORBmatcher.cc:37-39
    ORBmatcher(float nnratio = 0.6f, bool checkOri = true, int max_keypoints = 2400, int max_queries = 8192, int device = 0)
        : mfNNratio(nnratio), mbCheckOrientation(checkOri) {
        oslam::throw_on(oslam_matcher_create(&h_, 1, max_keypoints, max_queries, device));
    }
    ~ORBmatcher() { oslam_matcher_destroy(h_); }
    ORBmatcher(const ORBmatcher&) = delete;

    // int SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, const float th):
    // the caller gathers one oslam_proj_query_t per map point (INTEGRATION.md shows the loop) and
    // applies kp_match[k] >= 0  =>  F.mvpMapPoints[k] = vpMapPoints[kp_match[k]].
    int SearchByProjection(const FrameView& F, const std::vector<oslam_proj_query_t>& queries, std::vector<int32_t>& kp_match,
                           std::vector<int32_t>* q_match = nullptr) {
        const float bounds[4] = {F.mnMinX, F.mnMinY, F.mnMaxX, F.mnMaxY};
        kp_match.assign(F.N > 0 ? F.N : 1, -1);
        std::vector<int32_t> qm(queries.size() + 1), qd(queries.size() + 1);
        int32_t nm = 0;
        oslam::throw_on(oslam_match_search_by_projection(h_, F.N, reinterpret_cast<const oslam_keypoint_t*>(F.mvKeysUn), F.mvuRight,
                                                         F.mDescriptors, F.blocked, bounds, queries.data(), (int)queries.size(),
                                                         mfNNratio, 1, 0, qm.data(), qd.data(), kp_match.data(), &nm));
        kp_match.resize(F.N);
        if (q_match) { qm.resize(queries.size()); *q_match = qm; }
        return nm;
    }

    // int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono)
    int SearchByProjection(const FrameView& Cur, int Nlast, const float* lastXw, const uint8_t* last_has_mp,
                           const oslam::KeyPoint* lastKeysUn, const uint8_t* lastMpDesc, const float Tcw[16], const float Tlw[16],
                           const oslam_camera_t& cam, const std::vector<float>& scaleFactors, float th, bool bMono,
                           std::vector<int32_t>& kp_match) {
        const float bounds[4] = {Cur.mnMinX, Cur.mnMinY, Cur.mnMaxX, Cur.mnMaxY};
        kp_match.assign(Cur.N > 0 ? Cur.N : 1, -1);
        std::vector<int32_t> qm(Nlast + 1), qd(Nlast + 1);
        int32_t nm = 0;
        oslam::throw_on(oslam_match_project_last_frame(h_, Cur.N, reinterpret_cast<const oslam_keypoint_t*>(Cur.mvKeysUn), Cur.mvuRight,
                                                       Cur.mDescriptors, Cur.blocked, bounds, Nlast, lastXw, last_has_mp,
                                                       reinterpret_cast<const oslam_keypoint_t*>(lastKeysUn), lastMpDesc, Tcw, Tlw, &cam,
                                                       scaleFactors.data(), (int)scaleFactors.size(), th, bMono ? 1 : 0,
                                                       mbCheckOrientation ? 1 : 0, qm.data(), qd.data(), kp_match.data(), &nm));
        kp_match.resize(Cur.N);
        return nm;
    }

    float mfNNratio;
    bool mbCheckOrientation;

private:
    oslam_matcher_t* h_ = nullptr;
};

}  // namespace ORB_SLAM2
