// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_common.h).
// CPU restatement of the Frame grid (reference src/Frame.cc:455-470,567-632) and of
// ORBmatcher::SearchByProjection (reference src/ORBmatcher.cc:45-137, :1328-1470, :1601-1663).
// All arithmetic of these functions is in-tree reference code, so this restatement is exact.
#pragma once
#include "oracle_common.h"

namespace oracle {

constexpr int FRAME_GRID_ROWS = 48;  // reference include/Frame.h:43
constexpr int FRAME_GRID_COLS = 64;  // :44
constexpr int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30;  // src/ORBmatcher.cc:37-39

// One projected map point (what SearchByProjection reads from MapPoint / computes per point).
struct ProjQuery {
    float u, v;        // mTrackProjX/Y (:67) or projected u,v (:1370-1371)
    float ur;          // mTrackProjXR (:93) or u - mbf*invzc (:1411)
    float radius;      // r*mvScaleFactors[level] (:67) or th*mvScaleFactors[octave] (:1381)
    int32_t minLevel, maxLevel;  // level gate handed to GetFeaturesInArea
    int32_t flags;     // bit0 valid (mbTrackInView && !isBad, or projected inside the image);
                       // bit1 the map point has Observations()>0 (blocks later queries, :87-89)
    float angle;       // LastFrame.mvKeysUn[i].angle (:1432)
    uint8_t desc[32];
};
static_assert(sizeof(ProjQuery) == 64, "ProjQuery layout");

struct FrameView {
    int N;
    const KeyPoint* keysUn;   // mvKeysUn
    const float* uRight;      // mvuRight
    const uint8_t* desc;      // mDescriptors, N x 32
    const uint8_t* blocked;   // mvpMapPoints[i] && Observations()>0 before the call
    float minX, minY, maxX, maxY;  // mnMinX ... (src/Frame.cc:691-702)
};

int DescriptorDistance(const uint8_t* a, const uint8_t* b);  // src/ORBmatcher.cc:1647-1663

struct Grid {
    float minX, minY, invW, invH;
    std::vector<int> cells[FRAME_GRID_COLS][FRAME_GRID_ROWS];
    void build(const FrameView& f);  // AssignFeaturesToGrid + PosInGrid
    // GetFeaturesInArea (src/Frame.cc:567-620)
    std::vector<int> area(const FrameView& f, float x, float y, float r, int minLevel, int maxLevel) const;
};

// Windowed search shared by SearchByProjection(F, vpMapPoints) [use_ratio=1, check_ori=0] and
// SearchByProjection(Cur, Last) [use_ratio=0, check_ori=mbCheckOrientation].
// q_match[i] = keypoint claimed by query i (or -1), q_dist[i] = its Hamming distance,
// kp_match[k] = query whose map point ends up in mvpMapPoints[k]; -1 untouched; -2 set to NULL by
// the rotation-consistency pass.  Returns nmatches.
int SearchByProjection(const FrameView& f, const ProjQuery* q, int M, float nnratio, int use_ratio, int check_ori,
                       int* q_match, int* q_dist, int* kp_match);

// Projection front half of SearchByProjection(Cur, Last) (src/ORBmatcher.cc:1338-1392): builds the
// queries from the last frame's map points.  Tcw / Tlw are row-major 4x4 float (cv::Mat CV_32F).
struct LastFrameView {
    int N;
    const float* Xw;          // [N][3] world position of mvpMapPoints[i] (ignored if !has_mp[i])
    const uint8_t* has_mp;    // bit0: mvpMapPoints[i] && !mvbOutlier[i]; bit1: Observations()>0
    const KeyPoint* keys;     // mvKeys (octave) == mvKeysUn octave/angle
    const uint8_t* mp_desc;   // [N][32] pMP->GetDescriptor()
};
void ProjectLastFrame(const LastFrameView& last, const float* Tcw, const float* Tlw, float fx, float fy, float cx,
                      float cy, float bf, float b, const FrameView& cur, const float* scaleFactors, float th,
                      int bMono, ProjQuery* out);

void ComputeThreeMaxima(const int* histo_sizes, int L, int& ind1, int& ind2, int& ind3);  // :1601-1642

// Search half of ORBmatcher::Fuse(KeyFrame*, vpMapPoints, th) (src/ORBmatcher.cc:888-947): window query
// KeyFrame::GetFeaturesInArea (src/KeyFrame.cc:569-608, no level filter), level gate [minLevel, maxLevel],
// chi2 reprojection gate (7.8 stereo if mvuRight>=0, 5.99 mono), best Hamming, accepted if <= TH_LOW.
// The projection / distance / viewing-angle gates (:851-886) and the map surgery (:950-970) stay with the
// caller.  q_match[i] = keypoint or -1.
int FuseSearch(const FrameView& f, const ProjQuery* q, int M, const float* invLevelSigma2, int* q_match, int* q_dist);

}  // namespace oracle

namespace oracle {
// Frame::ComputeStereoMatches, reference src/Frame.cc:706-880.  pyrL/pyrR: un-padded pyramid levels
// (mvImagePyramid of the left / right extractor).  Out-of-image row indices of the row table
// (:723-733, unchecked in the reference) are skipped.  Empty match set: nothing to filter (the
// reference would index an empty vector, :867).
void ComputeStereoMatches(int N, const KeyPoint* keysL, const uint8_t* descL, int Nr, const KeyPoint* keysR,
                          const uint8_t* descR, const std::vector<Image>& pyrL, const std::vector<Image>& pyrR,
                          const float* scaleFactors, const float* invScaleFactors, float bf, float b, float* uRight,
                          float* depth);
}  // namespace oracle

namespace oracle {
// BoW-guided matchers.  The DBoW2 vocabulary is not in the reference tree, so the FeatureVectors
// (node id -> keypoint indices, std::map order) are inputs: side 1 as a flat list in iteration order
// (node ascending, indices in vector order), side 2 as CSR over its sorted node ids.
struct BowSide2 {
    int nNodes; const uint32_t* nodes; const int32_t* start; const int32_t* items;
};
// ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches), reference src/ORBmatcher.cc:159-288.
// q_idx1/q_node: KF entries in FeatureVector order; valid1[idx1] = map point exists && !isBad().
// match_f[k] = KF keypoint whose map point lands in vpMapPointMatches[k] (-1 none, -2 nulled by rotation check).
int SearchByBoW(int nq, const int32_t* q_idx1, const uint32_t* q_node, const KeyPoint* keys1, const uint8_t* desc1,
                const uint8_t* valid1, int N2, const KeyPoint* keys2, const uint8_t* desc2, const BowSide2& s2, float nnratio,
                int checkOri, int* match_f);
// ORBmatcher::SearchForTriangulation, reference src/ORBmatcher.cc:657-823 (vbMatched2 is never set there, so
// queries are independent).  skip1[idx1] = pKF1 map point exists; has_mp2[idx2] likewise; F12 row-major 3x3 float.
int SearchForTriangulation(int nq, const int32_t* q_idx1, const uint32_t* q_node, int N1, const KeyPoint* keys1,
                           const uint8_t* desc1, const float* uRight1, const uint8_t* skip1, int N2, const KeyPoint* keys2,
                           const uint8_t* desc2, const float* uRight2, const uint8_t* has_mp2, const BowSide2& s2,
                           const float* F12, float ex, float ey, const float* scaleFactors, const float* levelSigma2,
                           int bOnlyStereo, int checkOri, int* match12);
}  // namespace oracle
