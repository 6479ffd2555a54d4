// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_common.h).  PARITY UNPINNED.
// CPU restatement of Optimizer::PoseOptimization (reference src/Optimizer.cc:239-451),
// Optimizer::LocalBundleAdjustment (:453-778) and ObjectOptimizer::PoseOptimization2
// (src/ObjectOptimizer.cc:624-1240) together with the arithmetic they delegate to the ORB_SLAM2
// fork of g2o (Thirdparty/g2o, unversioned, NOT in /root/reference): SE3Quat / VertexSE3Expmap,
// EdgeSE3ProjectXYZ[OnlyPose], EdgeStereoSE3ProjectXYZ[OnlyPose], RobustKernelHuber,
// OptimizationAlgorithmLevenberg, BlockSolver_6_3 (Schur complement), restated from the published
// g2o sources.
#pragma once
#include <array>
#include <cstdint>
#include <vector>

namespace oracle {

struct SE3Quat {          // g2o::SE3Quat: unit quaternion (x,y,z,w) + translation, camera-from-world
    double q[4];
    double t[3];
};

SE3Quat se3_from_cvmat(const float* T16);          // Converter::toSE3Quat, src/Converter.cc:38-48
void se3_to_cvmat(const SE3Quat& s, float* T16);   // Converter::toCvMat(SE3Quat), :64-72
void se3_map(const SE3Quat& s, const double* X, double* out);
SE3Quat se3_exp(const double* update6);            // SE3Quat::exp
SE3Quat se3_mul(const SE3Quat& a, const SE3Quat& b);
void se3_rotation(const SE3Quat& s, double R[9]);

struct Camera { double fx, fy, cx, cy, bf; };

struct GraphEdge {
    int pose;             // pose vertex index
    int point;            // point vertex index, or -1 for a pose-only edge (Xw fixed in the edge)
    double Xw[3];         // pose-only edges
    double obs[3];        // u, v, (ur)
    bool stereo;
    double info;          // invSigma2 (information = info * I)
    bool robust;
    double delta;         // Huber delta (double of the reference's float sqrt(5.991) / sqrt(7.815))
    int level;            // g2o edge level: 0 active, 1 excluded
    double err[3];        // _error buffer (as last computed; may be stale, like g2o's)
    bool semantic;        // PoseOptimization2 extra edges (bookkeeping only)
};

struct Graph {
    Camera cam;
    std::vector<SE3Quat> poses;
    std::vector<uint8_t> pose_fixed;
    std::vector<std::array<double, 3>> points;
    std::vector<GraphEdge> edges;
    // statistics of the last optimize() call
    int lm_iterations = 0, lm_trials = 0;
};

void edge_compute_error(const Graph& g, GraphEdge& e);
double edge_chi2(const GraphEdge& e);
bool edge_depth_positive(const Graph& g, const GraphEdge& e);

// SparseOptimizer::initializeOptimization(level) + optimize(iterations) with
// OptimizationAlgorithmLevenberg.  stop (may be NULL) is the force-stop flag polled like g2o does.
// Returns the number of iterations performed.
int graph_optimize(Graph& g, int iterations, int level, const volatile int* stop);
// test hook: record the LM trials of the graph_optimize calls that follow into buf[cap][6] (nullptr: stop); not thread safe
void lm_trace_set(double* buf, int cap);
int lm_trace_count();
void edge_jacobians(const Graph& g, const GraphEdge& e, double Jp[18], double Jx[9]);

// reference src/Optimizer.cc:239-451.  has_mp[i] != 0 <=> pFrame->mvpMapPoints[i] != NULL.
// obs[i] = (kpUn.pt.x, kpUn.pt.y, mvuRight[i]) ; mono iff mvuRight[i] < 0.
int PoseOptimization(int N, const float* Tcw_in, const float* Xw, const float* obs, const float* invSigma2,
                     const uint8_t* has_mp, const float* K5 /*fx,fy,cx,cy,bf*/, float* Tcw_out, uint8_t* outlier,
                     int* stats /* [2] iterations, trials; may be NULL */);

// reference src/Optimizer.cc:453-778 after the graph has been gathered:
// poses [nKF][16] float (Tcw), fixed[k] = 1 for lFixedCameras, 2 for a local keyframe with mnId==0
// (setFixed(true) at :529 but still written back at :762-768), 0 for free local keyframes; points [nP][3];
// edges in insertion order: (kf, pt, u, v, ur, invSigma2).  Outputs poses/points rounded to float
// like Converter::toCvMat, erase[e] = 1 for observations the reference erases (:711-743).
void LocalBundleAdjustment(int nKF, const float* poses, const uint8_t* fixed, int nP, const float* points, int nE,
                           const int32_t* edge_kf, const int32_t* edge_pt, const float* edge_obs,
                           const float* edge_invSigma2, const float* K5, const volatile int* stop, float* poses_out,
                           float* points_out, uint8_t* erase, int* stats /* [4] it1, trials1, it2, trials2 */);

// Optimizer::BundleAdjustment, reference src/Optimizer.cc:49-237 (graph already gathered).
void BundleAdjustment(int nKF, const float* poses, const uint8_t* fixed, int nP, const float* points, int nE,
                      const int32_t* edge_kf, const int32_t* edge_pt, const float* edge_obs, const float* edge_invSigma2,
                      const float* K5, int nIterations, int bRobust, float* poses_out, float* points_out);
}  // namespace oracle

namespace oracle {
// ObjectOptimizer::PoseOptimization2, reference src/ObjectOptimizer.cc:624-1240, on flat inputs.
// Regular inputs as PoseOptimization.  Semantic inputs (the object layer's outputs):
//   masks    [nObj][H][W] uint8 {0,255}: Object2D.mask of every matched object (mvpObject3Ds[i] != NULL),
//   objmp_Xw [nObjMp][3], objmp_obj [nObjMp]: world positions of pObj3D->mvpMapPoints, object-major,
//   joint_kp / joint_obj [nJoint]: keypoints whose map point belongs to object joint_obj but whose
//            mvObjectKpIndices[idx].first differs (the M_joint set, :721-726), in creation order,
//   kp_uv [N][2]: mvKeysUn[i].pt, bounds = {mnMinX, mnMinY, mnMaxX, mnMaxY}, invSigma2_0 = mvInvLevelSigma2[0].
// Nearest mask pixel = exact NN under FLANN's float L2 (squared); ties -> first pixel in row-major order
// (PCL's order is unspecified).  Returns nInitialCorrespondences - nBad; *nSemNum = semantic constraints used.
int PoseOptimization2(int N, const float* Tcw_in, const float* Xw, const float* obs, const float* invSigma2,
                      const uint8_t* has_mp, const float* K5, int nObj, int H, int W, const uint8_t* masks,
                      int nObjMp, const float* objmp_Xw, const int32_t* objmp_obj, int nJoint, const int32_t* joint_kp,
                      const int32_t* joint_obj, const float* kp_uv, const float* bounds, float invSigma2_0,
                      float* Tcw_out, uint8_t* outlier, int* nSemNum);
}  // namespace oracle
