// orb_slam2_adapter.hpp — header-only C++ adapter that re-exposes the reference's class API on top of the C ABI in oslam_hip.h:
// ORB_SLAM2::ORBextractor (include/ORBextractor.h:45-110), ORB_SLAM2::ORBmatcher (include/ORBmatcher.h:41-83: both projection searches,
// SearchByBoW, SearchForTriangulation, Fuse, DescriptorDistance), Frame::ComputeStereoMatches (src/Frame.cc:706), ORB_SLAM2::Optimizer
// (include/Optimizer.h:38-46: PoseOptimization, LocalBundleAdjustment, BundleAdjustment) and ObjectOptimizer::PoseOptimization2
// (include/ObjectOptimizer.h:23).  The reference methods walk Frame / KeyFrame / MapPoint pointer graphs; here every method takes a flat
// "view" of exactly the members it reads and writes (the gather loops are in INTEGRATION.md).  tests/adapter_program.cc uses nothing but
// these classes; tests/test_adapter_gpu.py builds it, runs it and compares its outputs with the ctypes path.
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

    // the reference signature carries a mask argument that it ignores (src/ORBextractor.cc:1043-1044)
    void operator()(const oslam::Image8& image, const oslam::Image8& /*mask: ignored like the reference*/, std::vector<oslam::KeyPoint>& keypoints,
                    std::vector<uint8_t>& descriptors) {
        (*this)(image, keypoints, descriptors);
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
        // the tables depend only on the constructor arguments: the float chain of src/ORBextractor.cc:415-432, computed on the host (no GPU handle)
        std::vector<float> sc(nlevels_), s2(nlevels_), is(nlevels_), is2(nlevels_);
        sc[0] = 1.f; s2[0] = 1.f;
        const double sf = scaleFactor_;
        for (int i = 1; i < nlevels_; i++) { sc[i] = (float)(sc[i - 1] * sf); s2[i] = sc[i] * sc[i]; }
        for (int i = 0; i < nlevels_; i++) { is[i] = 1.0f / sc[i]; is2[i] = 1.0f / s2[i]; }
        return which == 0 ? sc : which == 1 ? is : which == 2 ? s2 : is2;
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
    ~ORBmatcher() { oslam_matcher_destroy(h_); oslam_bow_destroy(bow_); }
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

    // static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b) (src/ORBmatcher.cc:1647-1663)
    static int DescriptorDistance(const uint8_t* a, const uint8_t* b) {
        int d = 0;
        for (int i = 0; i < 8; i++) {
            uint32_t x, y;
            memcpy(&x, a + 4 * i, 4); memcpy(&y, b + 4 * i, 4);
            d += __builtin_popcount(x ^ y);
        }
        return d;
    }

    // int Fuse(KeyFrame* pKF, const vector<MapPoint*> &vpMapPoints, const float th) (src/ORBmatcher.cc:825-975), search half: the caller gathers
    // one query per candidate point that passes the projection gates (:840-890) and applies the surgery (:950-970) in query order
    // for q_match[i] >= 0.  Returns the number of fused points.
    int Fuse(const FrameView& KF, const std::vector<oslam_proj_query_t>& queries, const std::vector<float>& invLevelSigma2, std::vector<int32_t>& q_match) {
        const float bounds[4] = {KF.mnMinX, KF.mnMinY, KF.mnMaxX, KF.mnMaxY};
        q_match.assign(queries.size() + 1, -1);
        std::vector<int32_t> qd(queries.size() + 1);
        int32_t nf = 0;
        oslam::throw_on(oslam_match_fuse_search(h_, KF.N, reinterpret_cast<const oslam_keypoint_t*>(KF.mvKeysUn), KF.mvuRight, KF.mDescriptors, bounds,
                                                queries.data(), (int)queries.size(), invLevelSigma2.data(), (int)invLevelSigma2.size(), q_match.data(),
                                                qd.data(), &nf));
        q_match.resize(queries.size());
        return nf;
    }

    // DBoW2::FeatureVector of one side (node id -> keypoint indices), flattened as oslam_hip.h describes: the vocabulary is not in the reference tree
    struct FeatureVector {
        std::vector<int32_t> q_idx; std::vector<uint32_t> q_node;               // side 1: (node ascending, index order)
        std::vector<uint32_t> nodes; std::vector<int32_t> start, items;        // side 2: CSR over the sorted unique node ids
    };

    // int SearchByBoW(KeyFrame* pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches) (:159-288): match_f[k] = keyframe keypoint whose map point goes to
    // vpMapPointMatches[k]; kf_has_good_mp[i] = pKF's map point i exists and is not bad
    int SearchByBoW(const FrameView& KF, const FeatureVector& fvKF, const uint8_t* kf_has_good_mp, const FrameView& F, const FeatureVector& fvF,
                    std::vector<int32_t>& match_f) {
        ensure_bow();
        oslam_bow_side1_t s1 = {KF.N, reinterpret_cast<const oslam_keypoint_t*>(KF.mvKeysUn), KF.mDescriptors, nullptr, kf_has_good_mp,
                                (int32_t)fvKF.q_idx.size(), fvKF.q_idx.data(), fvKF.q_node.data()};
        oslam_bow_side2_t s2 = {F.N, reinterpret_cast<const oslam_keypoint_t*>(F.mvKeysUn), F.mDescriptors, nullptr, nullptr,
                                (int32_t)fvF.nodes.size(), fvF.nodes.data(), fvF.start.data(), fvF.items.data()};
        match_f.assign(F.N + 1, -1);
        int32_t nm = 0;
        oslam::throw_on(oslam_match_search_by_bow(bow_, &s1, &s2, mfNNratio, mbCheckOrientation ? 1 : 0, match_f.data(), &nm));
        match_f.resize(F.N);
        return nm;
    }

    // int SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, vector<pair<size_t,size_t>> &vMatchedPairs, const bool bOnlyStereo) (:657-823):
    // match12[i] = keypoint of KF2 matched to keypoint i of KF1 or -1; (ex, ey) = epipole of camera 1 in image 2 (:663-670)
    int SearchForTriangulation(const FrameView& KF1, const FeatureVector& fv1, const uint8_t* kf1_has_mp, const FrameView& KF2, const FeatureVector& fv2,
                               const uint8_t* kf2_has_mp, const float F12[9], float ex, float ey, const std::vector<float>& scaleFactors,
                               const std::vector<float>& levelSigma2, bool bOnlyStereo, std::vector<int32_t>& match12) {
        ensure_bow();
        oslam_bow_side1_t s1 = {KF1.N, reinterpret_cast<const oslam_keypoint_t*>(KF1.mvKeysUn), KF1.mDescriptors, KF1.mvuRight, kf1_has_mp,
                                (int32_t)fv1.q_idx.size(), fv1.q_idx.data(), fv1.q_node.data()};
        oslam_bow_side2_t s2 = {KF2.N, reinterpret_cast<const oslam_keypoint_t*>(KF2.mvKeysUn), KF2.mDescriptors, KF2.mvuRight, kf2_has_mp,
                                (int32_t)fv2.nodes.size(), fv2.nodes.data(), fv2.start.data(), fv2.items.data()};
        match12.assign(KF1.N + 1, -1);
        int32_t nm = 0;
        oslam::throw_on(oslam_match_search_for_triangulation(bow_, &s1, &s2, F12, ex, ey, scaleFactors.data(), levelSigma2.data(), (int)scaleFactors.size(),
                                                             bOnlyStereo ? 1 : 0, mbCheckOrientation ? 1 : 0, match12.data(), &nm));
        match12.resize(KF1.N);
        return nm;
    }

    float mfNNratio;
    bool mbCheckOrientation;

private:
    void ensure_bow() { if (!bow_) oslam::throw_on(oslam_bow_create(&bow_, 2400, 0)); }
    oslam_matcher_t* h_ = nullptr;
    oslam_bow_t* bow_ = nullptr;
};

// void Frame::ComputeStereoMatches() (src/Frame.cc:706-880): reads the two extractors' pyramids (mvImagePyramid) where they are, in HBM
inline void ComputeStereoMatches(ORBextractor& left, ORBextractor& right, const std::vector<oslam::KeyPoint>& mvKeys, const std::vector<uint8_t>& mDescriptors,
                                 const std::vector<oslam::KeyPoint>& mvKeysRight, const std::vector<uint8_t>& mDescriptorsRight, float mbf, float mb,
                                 std::vector<float>& mvuRight, std::vector<float>& mvDepth) {
    oslam_stereo_t* st = nullptr;
    oslam::throw_on(oslam_stereo_create(&st, 1, 2400, 0));
    mvuRight.assign(mvKeys.size() + 1, -1.f); mvDepth.assign(mvKeys.size() + 1, -1.f);
    const int rc = oslam_stereo_match(st, left.handle(), right.handle(), (int)mvKeys.size(), reinterpret_cast<const oslam_keypoint_t*>(mvKeys.data()),
                                      mDescriptors.data(), (int)mvKeysRight.size(), reinterpret_cast<const oslam_keypoint_t*>(mvKeysRight.data()),
                                      mDescriptorsRight.data(), left.GetLevels(), mbf, mb, mvuRight.data(), mvDepth.data());
    oslam_stereo_destroy(st);
    oslam::throw_on(rc);
    mvuRight.resize(mvKeys.size()); mvDepth.resize(mvKeys.size());
}

// Flat view of what Optimizer::PoseOptimization reads from / writes to a Frame (src/Optimizer.cc:239-451)
struct PoseFrameView {
    int N;
    float* mTcw;                          // 4x4 row-major CV_32F: input estimate, overwritten by SetPose
    const float* Xw;                      // [N][3] pMP->GetWorldPos() where has_mp[i]
    const uint8_t* has_mp;                // mvpMapPoints[i] != NULL
    const oslam::KeyPoint* mvKeysUn;
    const float* mvuRight;
    const float* mvInvLevelSigma2;        // [nLevels]
    uint8_t* mvbOutlier;                  // [N] out
    float fx, fy, cx, cy, mbf;
};

struct SemanticView {                     // what ObjectOptimizer::PoseOptimization2 adds (src/ObjectOptimizer.cc:685-767), see oslam_semantic_t
    int nObj, H, W; const uint8_t* masks;                               // [nObj][H][W] Object2D masks of the matched objects
    int nObjMp; const float* objmp_Xw; const int32_t* objmp_obj;        // map points of the matched Object3Ds
    int nJoint; const int32_t* joint_kp; const int32_t* joint_obj;      // keypoints with a map point of object joint_obj lying outside its Object2D
    float mnMinX, mnMinY, mnMaxX, mnMaxY;
};

// Flat local-BA graph: what Optimizer::LocalBundleAdjustment gathers at src/Optimizer.cc:456-650 and writes back at :746-777
struct BAGraph {
    int nKF; float* poses /*[nKF][16] in/out*/; const uint8_t* fixed /* 0 local, 1 fixed camera, 2 local keyframe with mnId == 0 */;
    int nP; float* points /*[nP][3] in/out*/;
    int nE; const int32_t* edge_kf; const int32_t* edge_pt; const float* edge_obs /*[nE][3] u, v, uR (<0 mono)*/; const float* edge_invSigma2;
    uint8_t* erase;                       // [nE] out: observations the reference erases (:711-757); may be NULL for BundleAdjustment
    float fx, fy, cx, cy, mbf;
};

class Optimizer {
public:
    // int static PoseOptimization(Frame* pFrame) (include/Optimizer.h:46)
    static int PoseOptimization(PoseFrameView& F) {
        std::vector<float> obs, inv;
        gather(F, obs, inv);
        const float K5[5] = {F.fx, F.fy, F.cx, F.cy, F.mbf};
        float Tout[16];
        int32_t n = 0;
        oslam::throw_on(oslam_pose_optimize(pose_handle(F.N), F.N, F.mTcw, F.Xw, obs.data(), inv.data(), F.has_mp, K5, Tout, F.mvbOutlier, &n, nullptr));
        memcpy(F.mTcw, Tout, sizeof(Tout));
        return n;
    }
    // void static LocalBundleAdjustment(KeyFrame* pKF, bool *pbStopFlag, Map* pMap) (include/Optimizer.h:45)
    static void LocalBundleAdjustment(BAGraph& g, bool* pbStopFlag = nullptr) {
        oslam_lba_t* h = lba_handle();
        volatile int32_t* flag = oslam_lba_stop_flag(h);
        *flag = (pbStopFlag && *pbStopFlag) ? 1 : 0;   // LocalMapping::InterruptBA sets the flag from another thread through StopFlag()
        const float K5[5] = {g.fx, g.fy, g.cx, g.cy, g.mbf};
        std::vector<float> po((size_t)g.nKF * 16), xo((size_t)g.nP * 3 + 3);
        oslam::throw_on(oslam_lba_optimize(h, g.nKF, g.poses, g.fixed, g.nP, g.points, g.nE, g.edge_kf, g.edge_pt, g.edge_obs, g.edge_invSigma2, K5, 1, po.data(),
                                           xo.data(), g.erase, nullptr));
        memcpy(g.poses, po.data(), po.size() * 4);
        if (g.nP) memcpy(g.points, xo.data(), (size_t)g.nP * 12);
    }
    static volatile int32_t* StopFlag() { return oslam_lba_stop_flag(lba_handle()); }
    // void static BundleAdjustment(const vector<KeyFrame*>&, const vector<MapPoint*>&, int nIterations, bool* pbStopFlag, const unsigned long nLoopKF, const bool bRobust)
    static void BundleAdjustment(BAGraph& g, int nIterations = 5, bool* pbStopFlag = nullptr, bool bRobust = true) {
        oslam_lba_t* h = lba_handle();
        *oslam_lba_stop_flag(h) = (pbStopFlag && *pbStopFlag) ? 1 : 0;
        const float K5[5] = {g.fx, g.fy, g.cx, g.cy, g.mbf};
        std::vector<float> po((size_t)g.nKF * 16), xo((size_t)g.nP * 3 + 3);
        oslam::throw_on(oslam_ba_optimize(h, g.nKF, g.poses, g.fixed, g.nP, g.points, g.nE, g.edge_kf, g.edge_pt, g.edge_obs, g.edge_invSigma2, K5, nIterations,
                                          bRobust ? 1 : 0, pbStopFlag ? 1 : 0, po.data(), xo.data()));
        memcpy(g.poses, po.data(), po.size() * 4);
        if (g.nP) memcpy(g.points, xo.data(), (size_t)g.nP * 12);
    }

    static void gather(const PoseFrameView& F, std::vector<float>& obs, std::vector<float>& inv) {
        obs.resize((size_t)F.N * 3 + 3); inv.resize(F.N + 1);
        for (int i = 0; i < F.N; i++) {
            obs[(size_t)i * 3] = F.mvKeysUn[i].x; obs[(size_t)i * 3 + 1] = F.mvKeysUn[i].y; obs[(size_t)i * 3 + 2] = F.mvuRight[i];
            inv[i] = F.mvInvLevelSigma2[F.mvKeysUn[i].octave];
        }
    }
    static oslam_poseopt_t* pose_handle(int N) {   // one handle per thread (the reference's static methods keep no state)
        static thread_local oslam_poseopt_t* h = nullptr;
        static thread_local int cap = 0;
        if (!h || N > cap) {
            oslam_poseopt_destroy(h);
            h = nullptr;
            cap = N < 2400 ? 2400 : N;
            oslam::throw_on(oslam_poseopt_create(&h, 1, cap, 0));
        }
        return h;
    }
    static oslam_lba_t* lba_handle() {
        static thread_local oslam_lba_t* h = nullptr;
        if (!h) oslam::throw_on(oslam_lba_create(&h, 1, 128, 4096, 32768, 0));
        return h;
    }
};

class ObjectOptimizer {
public:
    // static int PoseOptimization2(Frame* pFrame) (include/ObjectOptimizer.h:23); *nSemNum receives what the reference adds to N_AllSemanticConstraintNum
    static int PoseOptimization2(PoseFrameView& F, const SemanticView& S, int* nSemNum = nullptr) {
        std::vector<float> obs, inv, kp_uv((size_t)F.N * 2 + 2);
        Optimizer::gather(F, obs, inv);
        for (int i = 0; i < F.N; i++) { kp_uv[(size_t)i * 2] = F.mvKeysUn[i].x; kp_uv[(size_t)i * 2 + 1] = F.mvKeysUn[i].y; }
        oslam_semantic_t sem;
        sem.nObj = S.nObj; sem.H = S.H; sem.W = S.W; sem.masks = S.masks; sem.nObjMp = S.nObjMp; sem.objmp_Xw = S.objmp_Xw; sem.objmp_obj = S.objmp_obj;
        sem.nJoint = S.nJoint; sem.joint_kp = S.joint_kp; sem.joint_obj = S.joint_obj; sem.kp_uv = kp_uv.data();
        sem.bounds[0] = S.mnMinX; sem.bounds[1] = S.mnMinY; sem.bounds[2] = S.mnMaxX; sem.bounds[3] = S.mnMaxY; sem.invSigma2_0 = F.mvInvLevelSigma2[0];
        const float K5[5] = {F.fx, F.fy, F.cx, F.cy, F.mbf};
        float Tout[16];
        int32_t n = 0, ns = 0;
        oslam::throw_on(oslam_pose_optimize2(Optimizer::pose_handle(F.N), F.N, F.mTcw, F.Xw, obs.data(), inv.data(), F.has_mp, K5, &sem, Tout, F.mvbOutlier, &n, &ns));
        memcpy(F.mTcw, Tout, sizeof(Tout));
        if (nSemNum) *nSemNum = ns;
        return n;
    }
};

}  // namespace ORB_SLAM2
