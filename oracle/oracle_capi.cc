// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_common.h).  PARITY UNPINNED.
// C entry points (ctypes) over the CPU restatement.
#include "orb_extractor_oracle.h"

using namespace oracle;

extern "C" {

void* oo_orb_create(int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh) {
    return new OrbExtractor(nfeatures, scaleFactor, nlevels, iniTh, minTh);
}
void oo_orb_destroy(void* h) { delete (OrbExtractor*)h; }
void oo_orb_set_blur_sse2(void* h, int on) { ((OrbExtractor*)h)->blur_sse2_rounding = on != 0; }

int oo_orb_extract(void* h, const uint8_t* img, int w, int hgt, int stride, KeyPoint* kps, uint8_t* desc,
                   int cap, int* n_out) {
    OrbExtractor* e = (OrbExtractor*)h;
    std::vector<KeyPoint> k;
    std::vector<uint8_t> d;
    int rc = e->extract(img, w, hgt, stride, k, d);
    if (rc) { *n_out = 0; return rc; }
    *n_out = (int)k.size();
    if ((int)k.size() > cap) return -2;
    if (!k.empty()) {
        memcpy(kps, k.data(), k.size() * sizeof(KeyPoint));
        memcpy(desc, d.data(), d.size());
    }
    return 0;
}

void oo_orb_tables(void* h, float* scale, float* inv, float* sigma2, float* invsigma2, int* nfeat, int* umax) {
    OrbExtractor* e = (OrbExtractor*)h;
    for (int i = 0; i < e->nlevels; i++) {
        scale[i] = e->mvScaleFactor[i];
        inv[i] = e->mvInvScaleFactor[i];
        sigma2[i] = e->mvLevelSigma2[i];
        invsigma2[i] = e->mvInvLevelSigma2[i];
        nfeat[i] = e->mnFeaturesPerLevel[i];
    }
    for (int i = 0; i < 16; i++) umax[i] = e->umax[i];
}

int oo_orb_level_size(void* h, int level, int* w, int* hgt) {
    OrbExtractor* e = (OrbExtractor*)h;
    *w = e->mvImagePyramid[level].w;
    *hgt = e->mvImagePyramid[level].h;
    return 0;
}
int oo_orb_get_level(void* h, int level, uint8_t* out) {
    const Image& im = ((OrbExtractor*)h)->mvImagePyramid[level];
    memcpy(out, im.d.data(), im.d.size());
    return 0;
}
int oo_orb_get_blurred(void* h, int level, uint8_t* out) {
    const Image& im = ((OrbExtractor*)h)->blurred[level];
    if (im.d.empty()) return -1;
    memcpy(out, im.d.data(), im.d.size());
    return 0;
}
int oo_orb_num_candidates(void* h, int level) { return (int)((OrbExtractor*)h)->candidates[level].size(); }
int oo_orb_get_candidates(void* h, int level, KeyPoint* out) {
    auto& v = ((OrbExtractor*)h)->candidates[level];
    if (!v.empty()) memcpy(out, v.data(), v.size() * sizeof(KeyPoint));
    return (int)v.size();
}
int oo_orb_num_level_keys(void* h, int level) { return (int)((OrbExtractor*)h)->levelKeys[level].size(); }
int oo_orb_get_level_keys(void* h, int level, KeyPoint* out) {
    auto& v = ((OrbExtractor*)h)->levelKeys[level];
    if (!v.empty()) memcpy(out, v.data(), v.size() * sizeof(KeyPoint));
    return (int)v.size();
}
// run only the quad-tree on caller-supplied candidates
int oo_distribute_octree(void* h, const KeyPoint* cand, int n, int minX, int maxX, int minY, int maxY, int N,
                         KeyPoint* out, int cap) {
    std::vector<KeyPoint> v(cand, cand + n);
    auto r = ((OrbExtractor*)h)->DistributeOctTree(v, minX, maxX, minY, maxY, N);
    if ((int)r.size() > cap) return -2;
    if (!r.empty()) memcpy(out, r.data(), r.size() * sizeof(KeyPoint));
    return (int)r.size();
}

// primitives
int oo_fast_9_16(const uint8_t* roi, int stride, int cols, int rows, int threshold, int nms, KeyPoint* out, int cap) {
    std::vector<KeyPoint> v;
    fast_9_16(roi, stride, cols, rows, threshold, nms != 0, v);
    if ((int)v.size() > cap) return -2;
    if (!v.empty()) memcpy(out, v.data(), v.size() * sizeof(KeyPoint));
    return (int)v.size();
}
int oo_fast_corner_score(const uint8_t* p, int stride, int threshold) { return fast_corner_score(p, stride, threshold); }
void oo_resize_linear_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh) {
    Image s(sw, sh), d(dw, dh);
    memcpy(s.d.data(), src, s.d.size());
    resize_linear_u8(s, d);
    memcpy(dst, d.d.data(), d.d.size());
}
void oo_gaussian_blur(const uint8_t* src, int w, int h, uint8_t* dst, int sse2) {
    Image s(w, h), d;
    memcpy(s.d.data(), src, s.d.size());
    gaussian_blur_7x7_s2(s, d, sse2 != 0);
    memcpy(dst, d.d.data(), d.d.size());
}
float oo_fast_atan2(float y, float x) { return fastAtan2(y, x); }
const int8_t* oo_brief_pattern() { return brief_pattern(); }

}  // extern "C"

// ---------------------------------------------------------------------------------------------
#include "matcher_oracle.h"
extern "C" {
int oo_descriptor_distance(const uint8_t* a, const uint8_t* b) { return DescriptorDistance(a, b); }

int oo_search_by_projection(int N, const KeyPoint* keysUn, const float* uRight, const uint8_t* desc,
                            const uint8_t* blocked, const float* bounds /*minX,minY,maxX,maxY*/, const ProjQuery* q,
                            int M, float nnratio, int use_ratio, int check_ori, int* q_match, int* q_dist,
                            int* kp_match) {
    FrameView f{N, keysUn, uRight, desc, blocked, bounds[0], bounds[1], bounds[2], bounds[3]};
    return SearchByProjection(f, q, M, nnratio, use_ratio, check_ori, q_match, q_dist, kp_match);
}

void oo_project_last_frame(int N, const float* Xw, const uint8_t* has_mp, const KeyPoint* keys, const uint8_t* mp_desc,
                           const float* Tcw, const float* Tlw, const float* K /*fx,fy,cx,cy,bf,b*/,
                           const float* bounds, const float* scaleFactors, float th, int bMono, ProjQuery* out) {
    LastFrameView last{N, Xw, has_mp, keys, mp_desc};
    FrameView cur{0, nullptr, nullptr, nullptr, nullptr, bounds[0], bounds[1], bounds[2], bounds[3]};
    ProjectLastFrame(last, Tcw, Tlw, K[0], K[1], K[2], K[3], K[4], K[5], cur, scaleFactors, th, bMono, out);
}

int oo_features_in_area(int N, const KeyPoint* keysUn, const float* bounds, float x, float y, float r, int minLevel,
                        int maxLevel, int* out, int cap) {
    FrameView f{N, keysUn, nullptr, nullptr, nullptr, bounds[0], bounds[1], bounds[2], bounds[3]};
    Grid g;
    g.build(f);
    auto v = g.area(f, x, y, r, minLevel, maxLevel);
    for (size_t i = 0; i < v.size() && (int)i < cap; i++) out[i] = v[i];
    return (int)v.size();
}
}

// ---------------------------------------------------------------------------------------------
#include "optimizer_oracle.h"
extern "C" {
int oo_pose_optimization(int N, const float* Tcw_in, const float* Xw, const float* obs, const float* invSigma2,
                         const uint8_t* has_mp, const float* K5, float* Tcw_out, uint8_t* outlier, int* stats) {
    return PoseOptimization(N, Tcw_in, Xw, obs, invSigma2, has_mp, K5, Tcw_out, outlier, stats);
}
void oo_local_bundle_adjustment(int nKF, const float* poses, const uint8_t* fixed, int nP, const float* points, int nE,
                                const int32_t* edge_kf, const int32_t* edge_pt, const float* edge_obs,
                                const float* edge_invSigma2, const float* K5, const int* stop, float* poses_out,
                                float* points_out, uint8_t* erase, int* stats) {
    LocalBundleAdjustment(nKF, poses, fixed, nP, points, nE, edge_kf, edge_pt, edge_obs, edge_invSigma2, K5, stop, poses_out,
                          points_out, erase, stats);
}
// LM trace of the optimisations that follow (buf[cap][6]: F before, F of the trial, rho, lambda, accepted, first trial of a graph_optimize call); buf == NULL stops. Returns trials seen so far.
int oo_lm_trace(double* buf, int cap) {
    const int n = lm_trace_count();
    lm_trace_set(buf, cap);
    return n;
}
// SE3 helpers for unit tests
void oo_se3_exp_mul(const double* update6, const float* T_in, float* T_out) {
    SE3Quat T = se3_from_cvmat(T_in);
    SE3Quat r = se3_mul(se3_exp(update6), T);
    se3_to_cvmat(r, T_out);
}
// error + Jacobians of one edge at a pose (pose-only if point == NULL)
void oo_edge_eval(const float* T, const double* X, const double* obs, int stereo, int binary, const double* K5, double* err,
                  double* Jp, double* Jx) {
    Graph g;
    g.cam = Camera{K5[0], K5[1], K5[2], K5[3], K5[4]};
    g.poses.push_back(se3_from_cvmat(T));
    g.pose_fixed.push_back(0);
    GraphEdge e;
    memset(&e, 0, sizeof(e));
    e.pose = 0; e.stereo = stereo != 0; e.info = 1; e.point = -1;
    for (int i = 0; i < 3; i++) { e.Xw[i] = X[i]; e.obs[i] = obs[i]; }
    if (binary) { g.points.push_back({X[0], X[1], X[2]}); e.point = 0; }
    edge_compute_error(g, e);
    for (int i = 0; i < 3; i++) err[i] = e.err[i];
    // Jacobians through the (file-static) routine: run one build via graph internals is overkill; expose directly
    extern void oo_internal_edge_jac(const Graph&, const GraphEdge&, double*, double*);
    oo_internal_edge_jac(g, e, Jp, Jx);
}
}

extern "C" {
// stereo: uses the pyramids held by two oracle extractors after extract()
void oo_stereo_matches(void* orbL, void* orbR, int N, const KeyPoint* keysL, const uint8_t* descL, int Nr,
                       const KeyPoint* keysR, const uint8_t* descR, float bf, float b, float* uRight, float* depth) {
    OrbExtractor* L = (OrbExtractor*)orbL;
    OrbExtractor* R = (OrbExtractor*)orbR;
    ComputeStereoMatches(N, keysL, descL, Nr, keysR, descR, L->mvImagePyramid, R->mvImagePyramid, L->mvScaleFactor.data(),
                         L->mvInvScaleFactor.data(), bf, b, uRight, depth);
}
}

extern "C" {
int oo_pose_optimization2(int N, const float* Tcw_in, const float* Xw, const float* obs, const float* invSigma2,
                          const uint8_t* has_mp, const float* K5, int nObj, int H, int W, const uint8_t* masks, int nObjMp,
                          const float* objmp_Xw, const int32_t* objmp_obj, int nJoint, const int32_t* joint_kp,
                          const int32_t* joint_obj, const float* kp_uv, const float* bounds, float invSigma2_0, float* Tcw_out,
                          uint8_t* outlier, int* nSemNum) {
    return PoseOptimization2(N, Tcw_in, Xw, obs, invSigma2, has_mp, K5, nObj, H, W, masks, nObjMp, objmp_Xw, objmp_obj, nJoint,
                             joint_kp, joint_obj, kp_uv, bounds, invSigma2_0, Tcw_out, outlier, nSemNum);
}
}

extern "C" {
int oo_fuse_search(int N, const KeyPoint* keysUn, const float* uRight, const uint8_t* desc, const float* bounds, const ProjQuery* q,
                   int M, const float* invLevelSigma2, int* q_match, int* q_dist) {
    FrameView f{N, keysUn, uRight, desc, nullptr, bounds[0], bounds[1], bounds[2], bounds[3]};
    return FuseSearch(f, q, M, invLevelSigma2, q_match, q_dist);
}
}

extern "C" {
void oo_bundle_adjustment(int nKF, const float* poses, const uint8_t* fixed, int nP, const float* points, int nE, const int32_t* edge_kf,
                          const int32_t* edge_pt, const float* edge_obs, const float* edge_invSigma2, const float* K5, int nIterations,
                          int bRobust, float* poses_out, float* points_out) {
    BundleAdjustment(nKF, poses, fixed, nP, points, nE, edge_kf, edge_pt, edge_obs, edge_invSigma2, K5, nIterations, bRobust, poses_out, points_out);
}
}

extern "C" {
int oo_search_by_bow(int nq, const int32_t* q_idx1, const uint32_t* q_node, const KeyPoint* keys1, const uint8_t* desc1,
                     const uint8_t* valid1, int N2, const KeyPoint* keys2, const uint8_t* desc2, int nNodes, const uint32_t* nodes,
                     const int32_t* start, const int32_t* items, float nnratio, int checkOri, int* match_f) {
    BowSide2 s2{nNodes, nodes, start, items};
    return SearchByBoW(nq, q_idx1, q_node, keys1, desc1, valid1, N2, keys2, desc2, s2, nnratio, checkOri, match_f);
}
int oo_search_for_triangulation(int nq, const int32_t* q_idx1, const uint32_t* q_node, int N1, const KeyPoint* keys1, const uint8_t* desc1,
                                const float* uRight1, const uint8_t* skip1, int N2, const KeyPoint* keys2, const uint8_t* desc2,
                                const float* uRight2, const uint8_t* has_mp2, int nNodes, const uint32_t* nodes, const int32_t* start,
                                const int32_t* items, const float* F12, float ex, float ey, const float* scaleFactors,
                                const float* levelSigma2, int bOnlyStereo, int checkOri, int* match12) {
    BowSide2 s2{nNodes, nodes, start, items};
    return SearchForTriangulation(nq, q_idx1, q_node, N1, keys1, desc1, uRight1, skip1, N2, keys2, desc2, uRight2, has_mp2, s2, F12, ex, ey,
                                  scaleFactors, levelSigma2, bOnlyStereo, checkOri, match12);
}
}
