/*
 * oslam_hip.h — C ABI of the MI355X-native (gfx950) front-end + local-BA hot path of
 * yangliu9527/Object_SLAM (an ORB_SLAM2 fork).  Plain pointers and sizes only; no C++ / torch
 * types.  Every entry point returns 0 on success or a negative OSLAM_E_* code;
 * oslam_last_error() gives a thread-local message.  One handle per caller thread / HIP stream;
 * handles are not re-entrant (same contract as the reference classes, which own their buffers).
 *
 * Each declaration cites the reference interface (file:line under the reference tree) it
 * replaces.  INTEGRATION.md shows the adapter a maintainer adds on the reference side.
 */
#ifndef OSLAM_HIP_H
#define OSLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OSLAM_OK 0
#define OSLAM_E_INVALID (-1)   /* bad argument / unsupported shape */
#define OSLAM_E_HIP (-2)       /* HIP runtime error (no device, OOM, launch failure) */
#define OSLAM_E_CAPACITY (-3)  /* caller buffer or internal arena too small (never truncates silently) */
#define OSLAM_E_NUMERIC (-4)   /* solver failure */

#define OSLAM_MAX_LEVELS 16

/* cv::KeyPoint POD mirror (OpenCV 3.2 core/types.hpp; used at include/ORBextractor.h:59-61). */
typedef struct oslam_keypoint {
    float x, y;       /* pt */
    float size;
    float angle;      /* degrees [0,360) */
    float response;   /* FAST score */
    int32_t octave;
    int32_t class_id; /* -1 */
} oslam_keypoint_t;

const char* oslam_last_error(void);
/* Number of visible HIP devices (0 if none); never throws. */
int oslam_device_count(void);

/* ------------------------------------------------------------------------------------------
 * ORBextractor — replaces ORB_SLAM2::ORBextractor (include/ORBextractor.h:45-110,
 * src/ORBextractor.cc:410-470 ctor, :1043-1105 operator()).
 * The handle is created for one image geometry and a maximum batch of images per call;
 * batch > 1 is the batch-of-sequences mode (independent images, same arithmetic per image).
 * ---------------------------------------------------------------------------------------- */
typedef struct oslam_orb oslam_orb_t;

int oslam_orb_create(oslam_orb_t** out, int nfeatures, float scaleFactor, int nlevels, int iniThFAST,
                     int minThFAST, int width, int height, int max_batch, int device);
void oslam_orb_destroy(oslam_orb_t* h);

/* Getters of include/ORBextractor.h:63-83; each array has nlevels entries (any may be NULL). */
int oslam_orb_get_scale_tables(const oslam_orb_t* h, float* scaleFactors, float* invScaleFactors,
                               float* levelSigma2, float* invLevelSigma2, int* nFeaturesPerLevel);
/* cv::GaussianBlur column-pass rounding model (DESIGN.md "blur rounding"): 1 (default) = OpenCV 3.2
 * on x86 SSE2 (half-to-even for x < (w & ~3), half-up tail); 0 = scalar fixed point everywhere. */
int oslam_orb_set_blur_rounding(oslam_orb_t* h, int sse2);
/* Per image capacity of the keypoint / descriptor outputs (sum of per-level node caps). */
int oslam_orb_max_keypoints(const oslam_orb_t* h);

/* operator() drop-in (src/ORBextractor.cc:1043): host gray image in, host keypoints +
 * descriptors (n x 32 bytes, row-major) out.  Empty image -> *n_out = 0, returns 0 (the
 * reference returns silently, :1046).  cap < n -> OSLAM_E_CAPACITY with *n_out = n. */
int oslam_orb_extract(oslam_orb_t* h, const uint8_t* gray, int width, int height, int stride,
                      oslam_keypoint_t* keypoints, uint8_t* descriptors, int cap, int* n_out);

/* Batch mode, everything resident in HBM: d_gray holds `batch` images (row pitch `stride`
 * bytes, `image_stride` bytes between images).  Asynchronous on `stream` (a hipStream_t, may be
 * NULL = default stream); results stay on the device, see oslam_orb_results_device(). */
int oslam_orb_extract_batch_device(oslam_orb_t* h, const uint8_t* d_gray, int batch, int stride,
                                   size_t image_stride, void* stream);
/* Device pointers of the last batch: keypoints [batch][cap] (oslam_keypoint_t), descriptors
 * [batch][cap][32], counts [batch] (int32), cap = oslam_orb_max_keypoints().  status[0] != 0
 * after the stream has drained means an internal arena overflowed (OSLAM_E_CAPACITY). */
int oslam_orb_results_device(const oslam_orb_t* h, const oslam_keypoint_t** d_keypoints,
                             const uint8_t** d_descriptors, const int32_t** d_counts,
                             const int32_t** d_status);
/* Synchronise the stream and copy image b's results of the last batch to the host. */
int oslam_orb_fetch(oslam_orb_t* h, int b, oslam_keypoint_t* keypoints, uint8_t* descriptors, int cap,
                    int* n_out);

/* mvImagePyramid (public member, include/ORBextractor.h:85; read by Frame::ComputeStereoMatches,
 * src/Frame.cc:713,803,815): level image b of the last batch, host copy (w*h bytes, tight) or
 * device view. */
int oslam_orb_level_size(const oslam_orb_t* h, int level, int* width, int* height);
int oslam_orb_get_pyramid_level(oslam_orb_t* h, int b, int level, uint8_t* out);
int oslam_orb_pyramid_level_device(const oslam_orb_t* h, int b, int level, const uint8_t** d_ptr,
                                   int* pitch);

/* Stage outputs for parity tests (host copies; call after a batch has run). */
int oslam_orb_debug_get_blurred(oslam_orb_t* h, int b, int level, uint8_t* out);
/* FAST candidates of a level in reference order (x, y in region coords, response):
 * src/ORBextractor.cc:789-829.  out is [cap][3] int32; returns count in *n_out. */
int oslam_orb_debug_get_candidates(oslam_orb_t* h, int b, int level, int32_t* out, int cap, int* n_out);
/* Quad-tree survivors of a level in reference order (level coords): src/ORBextractor.cc:834-847. */
int oslam_orb_debug_get_level_keys(oslam_orb_t* h, int b, int level, int32_t* out, int cap, int* n_out);

/* Kernel-group timing with HIP events recorded on the launch stream (bench.py's roofline leg).
 * Groups: 0 pyramid (K1 x (nlevels-1)), 1 FAST cells (K2/K3), 2 blur (K6 x nlevels), 3 quad-tree (K4),
 * 4 orientation + descriptors (K5/K7).  get_profile returns accumulated milliseconds since
 * set_profiling(h, 1) and the number of batches / images they cover. */
int oslam_orb_set_profiling(oslam_orb_t* h, int on);
int oslam_orb_get_profile(oslam_orb_t* h, double ms[5], long long* batches, long long* images);

/* Phase cycle counters of profiling builds (-DOSLAM_FAST_PROFILE); zeros otherwise. */
int oslam_orb_debug_counters(oslam_orb_t* h, unsigned long long out[16], int reset);

/* Work model of one extract call: algorithmic bytes per image. */
/* Synchronous device-to-host copy (tests and tools read the *_results_device arrays with it). */
int oslam_memcpy_from_device(void* dst, const void* d_src, size_t bytes);
int64_t oslam_orb_algorithmic_bytes(const oslam_orb_t* h, int n_keypoints);

/* ------------------------------------------------------------------------------------------
 * ORBmatcher — replaces the projection searches of ORB_SLAM2::ORBmatcher (include/ORBmatcher.h:41-83)
 * and the Frame grid they query (src/Frame.cc:455-470 AssignFeaturesToGrid, :567-620
 * GetFeaturesInArea, :622-632 PosInGrid).  Constants TH_HIGH=100, TH_LOW=50, HISTO_LENGTH=30
 * (src/ORBmatcher.cc:37-39).
 * ---------------------------------------------------------------------------------------- */
typedef struct oslam_matcher oslam_matcher_t;

/* One projected map point: what SearchByProjection reads from the MapPoint (mTrackProjX/Y/XR,
 * mnTrackScaleLevel, GetDescriptor(), Observations()) or computes from the last frame. 64 bytes. */
typedef struct oslam_proj_query {
    float u, v;            /* mTrackProjX, mTrackProjY (src/ORBmatcher.cc:67) / projection (:1370-1371) */
    float ur;              /* mTrackProjXR (:93) / u - mbf*invzc (:1411) */
    float radius;          /* r*mvScaleFactors[level] (:67) / th*mvScaleFactors[octave] (:1381) */
    int32_t minLevel, maxLevel; /* GetFeaturesInArea level gate (src/Frame.cc:567) */
    int32_t flags;         /* bit0: searched (mbTrackInView && !isBad()); bit1: Observations()>0 (blocks
                              later candidates, src/ORBmatcher.cc:87-89) */
    float angle;           /* LastFrame.mvKeysUn[i].angle for the rotation histogram (:1432) */
    uint8_t desc[32];      /* pMP->GetDescriptor() */
} oslam_proj_query_t;

/* Frame side of a search, batch of frames in HBM (device pointers), fixed per-frame stride. */
typedef struct oslam_match_frames {
    const oslam_keypoint_t* keysUn; /* mvKeysUn [batch][kp_stride] */
    int kp_stride;
    const float* uRight;            /* mvuRight [batch][kp_stride], NULL = monocular (-1) */
    const uint8_t* desc;            /* mDescriptors [batch][kp_stride][32] */
    const uint8_t* blocked;         /* [batch][kp_stride]: mvpMapPoints[i] && Observations()>0 before the call; NULL = none */
    const int32_t* n_kps;           /* [batch] (device) or NULL -> n_kps_const */
    int n_kps_const;
    float minX, minY, maxX, maxY;   /* mnMinX.. (src/Frame.cc:691-702) */
} oslam_match_frames_t;

/* Last-frame side of SearchByProjection(CurrentFrame, LastFrame, th, bMono) (src/ORBmatcher.cc:1328). */
typedef struct oslam_match_last {
    const float* Xw;                /* [batch][kp_stride][3] pMP->GetWorldPos() */
    const uint8_t* has_mp;          /* bit0: mvpMapPoints[i] && !mvbOutlier[i]; bit1: Observations()>0 */
    const oslam_keypoint_t* keys;   /* LastFrame.mvKeysUn (octave, angle) */
    const uint8_t* mp_desc;         /* [batch][kp_stride][32] pMP->GetDescriptor() */
    int kp_stride;
    const int32_t* n_kps;           /* [batch] (device) or NULL -> n_kps_const */
    int n_kps_const;
} oslam_match_last_t;

typedef struct oslam_camera { float fx, fy, cx, cy, bf, b; } oslam_camera_t;

/* max_keypoints <= 2400: one frame's keypoints, descriptors and 64x48 grid live in 160 KiB of LDS. */
int oslam_matcher_create(oslam_matcher_t** out, int max_batch, int max_keypoints, int max_queries, int device);
void oslam_matcher_destroy(oslam_matcher_t* h);

/* Windowed Hamming search with the reference's sequential claim order (a later query skips a keypoint
 * claimed by an earlier observed map point).  use_ratio=1, check_ori=0: SearchByProjection(Frame&,
 * vector<MapPoint*>&, th) (src/ORBmatcher.cc:45-129).  use_ratio=0: the search half of
 * SearchByProjection(Cur, Last) (:1394-1467).  d_queries NULL = the handle's internal query buffer
 * (filled by oslam_match_project_last_batch_device).  Asynchronous on `stream`. */
int oslam_match_search_batch_device(oslam_matcher_t* h, const oslam_match_frames_t* frames,
                                    const oslam_proj_query_t* d_queries, int q_stride, const int32_t* d_n_queries,
                                    int n_queries_const, int batch, float nnratio, int use_ratio, int check_ori,
                                    int th_high, void* stream);
/* Projection half of SearchByProjection(Cur, Last) (src/ORBmatcher.cc:1338-1392): fills the internal
 * query buffer (stride last->kp_stride) from the last frame's map points. d_Tcw/d_Tlw: [batch][16]
 * row-major float (cv::Mat CV_32F mTcw). */
int oslam_match_project_last_batch_device(oslam_matcher_t* h, const oslam_match_last_t* last, const float* d_Tcw,
                                          const float* d_Tlw, const oslam_camera_t* cam,
                                          const oslam_match_frames_t* cur, const float* scaleFactors, int nlevels,
                                          float th, int bMono, int batch, void* stream);
/* Device results of the last search: q_match/q_dist [batch][q_stride] (keypoint index or -1, Hamming
 * distance), kp_match [batch][max_keypoints] (query index now held in mvpMapPoints[k]; -1 untouched;
 * -2 set to NULL by the rotation check), nmatches [batch] (return value of the reference call). */
int oslam_match_results_device(const oslam_matcher_t* h, const int32_t** q_match, const int32_t** q_dist,
                               const int32_t** kp_match, const int32_t** nmatches,
                               const oslam_proj_query_t** queries, const int32_t** n_queries);
int oslam_match_fetch(oslam_matcher_t* h, int b, int q_stride, int n_queries, int kp_stride, int n_kps,
                      int32_t* q_match, int32_t* q_dist, int32_t* kp_match, int32_t* nmatches,
                      int32_t* iterations, void* stream);

/* Host drop-ins (one frame, host buffers in and out). bounds = {mnMinX, mnMinY, mnMaxX, mnMaxY}. */
int oslam_match_search_by_projection(oslam_matcher_t* h, int N, const oslam_keypoint_t* keysUn,
                                     const float* uRight, const uint8_t* desc, const uint8_t* blocked,
                                     const float bounds[4], const oslam_proj_query_t* queries, int M,
                                     float nnratio, int use_ratio, int check_ori, int32_t* q_match,
                                     int32_t* q_dist, int32_t* kp_match, int32_t* nmatches);
int oslam_match_project_last_frame(oslam_matcher_t* h, int N, const oslam_keypoint_t* keysUn, const float* uRight,
                                   const uint8_t* desc, const uint8_t* blocked, const float bounds[4], int Nlast,
                                   const float* Xw, const uint8_t* has_mp, const oslam_keypoint_t* last_keys,
                                   const uint8_t* mp_desc, const float Tcw[16], const float Tlw[16],
                                   const oslam_camera_t* cam, const float* scaleFactors, int nlevels, float th,
                                   int bMono, int check_ori, int32_t* q_match, int32_t* q_dist,
                                   int32_t* kp_match, int32_t* nmatches);
/* Search half of ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th) (src/ORBmatcher.cc:888-947):
 * KeyFrame::GetFeaturesInArea window (src/KeyFrame.cc:569-608), level gate [minLevel, maxLevel] =
 * [nPredictedLevel-1, nPredictedLevel], chi2 reprojection gate (7.8 if mvuRight>=0 else 5.99) with
 * mvInvLevelSigma2, best Hamming <= TH_LOW.  q_match[i] = keypoint to fuse map point i with (or -1).
 * The per-point projection / distance / viewing-angle gates (:851-886) fill the queries and the
 * Replace / AddObservation surgery (:950-970) is applied by the caller in query order. */
int oslam_match_fuse_search(oslam_matcher_t* h, int N, const oslam_keypoint_t* keysUn, const float* uRight,
                            const uint8_t* desc, const float bounds[4], const oslam_proj_query_t* queries, int M,
                            const float* invLevelSigma2, int nlevels, int32_t* q_match, int32_t* q_dist,
                            int32_t* n_fused);
int oslam_match_fuse_batch_device(oslam_matcher_t* h, const oslam_match_frames_t* frames,
                                  const oslam_proj_query_t* d_queries, int q_stride, const int32_t* d_n_queries,
                                  int n_queries_const, int batch, const float* invLevelSigma2, int nlevels,
                                  void* stream);
/* Descriptor pairs compared (256-bit XOR + popcount) by the first pass of the last search over `batch` frames — the work unit of the
 * matching roofline; synchronises the device. */
int oslam_match_hamming_pairs(oslam_matcher_t* h, int batch, int64_t* total);
/* Phase wall-clock counters (100 MHz ticks) of frame 0, profiling builds (-DOSLAM_MATCH_PROFILE); zeros otherwise. */
int oslam_match_debug_counters(oslam_matcher_t* h, long long out[8], int reset);
int oslam_match_debug_get_queries(oslam_matcher_t* h, int b, int q_stride, int n, oslam_proj_query_t* out);

/* ------------------------------------------------------------------------------------------
 * BoW-guided matchers: ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&) (src/ORBmatcher.cc:159-288)
 * and ORBmatcher::SearchForTriangulation (:657-823).  DBoW2 and its vocabulary are not in the reference tree:
 * the DBoW2::FeatureVector of each side (node id -> keypoint indices) is an INPUT.  Side 1 is the flat list the
 * reference iterates (std::map order: node ascending, indices in vector order); side 2 is CSR over its sorted,
 * unique node ids.  Host pointers, one pair per call.
 * SearchByBoW: side1.flag[i] = pKF map point exists && !isBad(); match_f[k] = keyframe keypoint whose map point is
 *   written to vpMapPointMatches[k] (-1 none, -2 removed by the rotation check); nmatches = return value.
 *   The claim order (:211-212: a frame keypoint matched by an earlier keyframe keypoint is skipped) is reproduced.
 * SearchForTriangulation: side1.flag[i] = pKF1->GetMapPoint(i) != NULL (skipped); side2.has_mp likewise;
 *   F12 row-major 3x3 (cv::Mat CV_32F), (ex, ey) = epipole in image 2 (:663-670); match12[i] = keypoint of KF2 or -1.
 *   (vbMatched2 is never set in the reference, :722 — queries are independent and so they are here.)
 * ---------------------------------------------------------------------------------------- */
typedef struct oslam_bow oslam_bow_t;
typedef struct oslam_bow_side1 {
    int32_t N; const oslam_keypoint_t* keys; const uint8_t* desc; const float* uRight /* NULL = all -1 */; const uint8_t* flag;
    int32_t nq; const int32_t* q_idx; const uint32_t* q_node;
} oslam_bow_side1_t;
typedef struct oslam_bow_side2 {
    int32_t N; const oslam_keypoint_t* keys; const uint8_t* desc; const float* uRight /* NULL = all -1 */; const uint8_t* has_mp /* NULL = none */;
    int32_t nNodes; const uint32_t* nodes; const int32_t* start; const int32_t* items;
} oslam_bow_side2_t;
int oslam_bow_create(oslam_bow_t** out, int max_keypoints /* <= 2400 */, int device);
void oslam_bow_destroy(oslam_bow_t* h);
int oslam_match_search_by_bow(oslam_bow_t* h, const oslam_bow_side1_t* kf, const oslam_bow_side2_t* frame, float nnratio,
                              int checkOri, int32_t* match_f, int32_t* nmatches);
int oslam_match_search_for_triangulation(oslam_bow_t* h, const oslam_bow_side1_t* kf1, const oslam_bow_side2_t* kf2,
                                         const float F12[9], float ex, float ey, const float* scaleFactors,
                                         const float* levelSigma2, int nlevels, int bOnlyStereo, int checkOri,
                                         int32_t* match12, int32_t* nmatches);

/* Batch of independent pairs, one workgroup each in one launch (the batch-of-sequences layout).  triangulation == 0: SearchByBoW, match
 * [s2.N]; != 0: SearchForTriangulation (bOnlyStereo = false), match [s1.N].  nmatches is written per job. */
typedef struct oslam_bow_job {
    oslam_bow_side1_t s1; oslam_bow_side2_t s2;
    int32_t triangulation;
    float nnratio; int32_t checkOri;
    float F12[9]; float ex, ey;
    int32_t* match;
    int32_t nmatches;
} oslam_bow_job_t;
int oslam_match_bow_batch(oslam_bow_t* h, int n, oslam_bow_job_t* jobs, const float* scaleFactors, const float* levelSigma2, int nlevels);
/* The same with device-resident copies of the keypoint (mvKeysUn), descriptor and mvuRight arrays of either side: res[i] members that are NULL fall back to
 * the host arrays of jobs[i] (which are then packed and uploaded as above).  The resident arrays must be complete when the call is made. */
typedef struct oslam_bow_resident {
    const oslam_keypoint_t* d_keys1; const uint8_t* d_desc1; const float* d_uRight1;
    const oslam_keypoint_t* d_keys2; const uint8_t* d_desc2; const float* d_uRight2;
} oslam_bow_resident_t;
int oslam_match_bow_batch_resident(oslam_bow_t* h, int n, oslam_bow_job_t* jobs, const oslam_bow_resident_t* res, const float* scaleFactors,
                                   const float* levelSigma2, int nlevels);

/* ------------------------------------------------------------------------------------------
 * Frame::ComputeStereoMatches (src/Frame.cc:706-880): row-band Hamming search of left keypoints in the
 * right image (levels +-1, u in [uL - bf/b, uL], best < (TH_HIGH+TH_LOW)/2), 11-shift 11x11 SAD on the
 * left keypoint's pyramid level, parabola sub-pixel fit, depth = bf/disparity, rejection of matches
 * with SAD >= 1.5*1.4*median.  Reads the pyramids of the two extractor handles' last batch
 * (mvImagePyramid, include/ORBextractor.h:85).  Outputs mvuRight / mvDepth (-1 = no match).
 * Window reads that would leave the image (unchecked in the reference) are skipped.
 * ---------------------------------------------------------------------------------------- */
typedef struct oslam_stereo oslam_stereo_t;
int oslam_stereo_create(oslam_stereo_t** out, int max_batch, int max_keypoints /* <= 2400; level-0 image height <= 4096 */, int device);
void oslam_stereo_destroy(oslam_stereo_t* h);
int oslam_stereo_match(oslam_stereo_t* h, oslam_orb_t* orbL, oslam_orb_t* orbR, int N, const oslam_keypoint_t* keysL,
                       const uint8_t* descL, int Nr, const oslam_keypoint_t* keysR, const uint8_t* descR, int nlevels,
                       float bf, float b, float* uRight, float* depth);
int oslam_stereo_match_batch_device(oslam_stereo_t* h, oslam_orb_t* orbL, oslam_orb_t* orbR, int batch, int kp_stride,
                                    const oslam_keypoint_t* d_kpL, const uint8_t* d_descL, const int32_t* d_nL,
                                    int nL_const, const oslam_keypoint_t* d_kpR, const uint8_t* d_descR,
                                    const int32_t* d_nR, int nR_const, int nlevels, float bf, float b, void* stream);
int oslam_stereo_results_device(const oslam_stereo_t* h, const float** d_uRight, const float** d_depth,
                                const int32_t** d_n_matched);

/* ------------------------------------------------------------------------------------------
 * Optimizer::PoseOptimization — motion-only BA (include/Optimizer.h:46, src/Optimizer.cc:239-451):
 * 4 rounds x optimize(10) of g2o Levenberg-Marquardt on one SE3 vertex, Huber(sqrt(5.991) mono,
 * sqrt(7.815) stereo) dropped after round index 2, chi2 re-classification after every round, every
 * round restarted from the input pose.  State is float32 at the boundary (cv::Mat CV_32F), fp64
 * inside (Converter::toSE3Quat / toCvMat, src/Converter.cc:38-72).
 * Per keypoint i: has_mp[i] = pFrame->mvpMapPoints[i] != NULL, Xw = pMP->GetWorldPos(),
 * obs = (mvKeysUn[i].pt.x, .pt.y, mvuRight[i]) with mvuRight < 0 => monocular edge,
 * invSigma2 = mvInvLevelSigma2[mvKeysUn[i].octave].  K5 = {fx, fy, cx, cy, mbf}.
 * Outputs: pose (pFrame->SetPose), outlier[i] = mvbOutlier[i], n_inliers = return value
 * (nInitialCorrespondences - nBad; 0 and pose untouched if < 3 correspondences).
 * ---------------------------------------------------------------------------------------- */
typedef struct oslam_poseopt oslam_poseopt_t;
int oslam_poseopt_create(oslam_poseopt_t** out, int max_batch, int max_points /* <= 11000 (chi2 + level per edge slot live in LDS) */, int device);
void oslam_poseopt_destroy(oslam_poseopt_t* h);
int oslam_pose_optimize(oslam_poseopt_t* h, int N, const float Tcw_in[16], const float* Xw, const float* obs,
                        const float* invSigma2, const uint8_t* has_mp, const float K5[5], float Tcw_out[16],
                        uint8_t* outlier, int32_t* n_inliers, int32_t stats[2] /* LM iterations, trials; may be NULL */);
/* Test hook: record the Levenberg-Marquardt trials of frame 0 of the calls that follow (cap > 0; cap == 0 stops), and read them back as
 * out[cap][6] = (F before the trial, F of the trial, rho, lambda of the trial, accepted, first trial of a round); *n = trials seen since the last read.  The parity
 * tests compare this sequence with the oracle's (g2o OptimizationAlgorithmLevenberg::solve, replaced at src/Optimizer.cc:407-409). */
int oslam_poseopt_trace(oslam_poseopt_t* h, int cap);
int oslam_poseopt_trace_read(oslam_poseopt_t* h, double* out, int32_t* n);
/* Batch of independent frames in HBM (device pointers, per-frame stride in points). Asynchronous. */
int oslam_pose_optimize_batch_device(oslam_poseopt_t* h, int batch, int stride, const int32_t* d_n, int n_const,
                                     const float* d_Tcw, const float* d_Xw, const float* d_obs,
                                     const float* d_invSigma2, const uint8_t* d_has_mp, const float K5[5],
                                     void* stream);
int oslam_poseopt_results_device(const oslam_poseopt_t* h, const float** d_Tcw_out, const uint8_t** d_outlier,
                                 const int32_t** d_n_inliers, const int32_t** d_stats);

/* ObjectOptimizer::PoseOptimization2 (include/ObjectOptimizer.h:23, src/ObjectOptimizer.cc:624-1240), the
 * pose optimisation Tracking::TrackLocalMap calls (src/Tracking.cc:1022): PoseOptimization plus
 * semantic reprojection edges to the nearest object-mask pixel.  Host pointers.  The object layer
 * (association, out of scope) supplies: the Object2D masks of the matched objects (uint8 {0,255},
 * [nObj][H][W], tight), the world positions of every matched Object3D's map points (object-major),
 * and the M_joint set (:721-726): keypoints whose map point belongs to object joint_obj[j] while
 * mvObjectKpIndices[kp].first != that object.  Nearest pixel = exact NN under FLANN's float squared
 * L2, ties -> first pixel in row-major order.  n_semantic = the reference's nSemNum (:1232). */
typedef struct oslam_semantic {
    int32_t nObj, H, W;
    const uint8_t* masks;
    int32_t nObjMp; const float* objmp_Xw; const int32_t* objmp_obj;
    int32_t nJoint; const int32_t* joint_kp; const int32_t* joint_obj;
    const float* kp_uv;      /* [N][2] mvKeysUn[i].pt */
    float bounds[4];         /* mnMinX, mnMinY, mnMaxX, mnMaxY */
    float invSigma2_0;       /* mvInvLevelSigma2[0] */
} oslam_semantic_t;
int oslam_pose_optimize2(oslam_poseopt_t* h, int N, const float Tcw_in[16], const float* Xw, const float* obs,
                         const float* invSigma2, const uint8_t* has_mp, const float K5[5], const oslam_semantic_t* sem,
                         float Tcw_out[16], uint8_t* outlier, int32_t* n_inliers, int32_t* n_semantic);

/* Batch form (Tracking::TrackLocalMap of several sequences, src/Tracking.cc:1022): device pointers, asynchronous on `stream`.  The base arrays are those
 * of oslam_pose_optimize_batch_device.  Frame b uses masks d_mask_ptrs[obj0 .. obj0 + nObj) (device pointers to uint8 {0,255} images of H x W, rows
 * mask_pitch bytes apart: nothing is copied), object map points objmp0 .. objmp0 + nObjMp of the pools (objmp_obj = object index INSIDE the frame)
 * and M_joint entries joint0 .. joint0 + nJoint; mvKeysUn[kp].pt is read from the frame's obs rows.  n_semantic per frame through
 * oslam_poseopt_semantic_results_device; a frame with fewer than 3 correspondences reports 0. */
typedef struct oslam_sem_frame { int32_t nObj, obj0, nObjMp, objmp0, nJoint, joint0; } oslam_sem_frame_t;
int oslam_pose_optimize2_batch_device(oslam_poseopt_t* h, int batch, int stride, const int32_t* d_n, const float* d_Tcw, const float* d_Xw, const float* d_obs,
                                      const float* d_invSigma2, const uint8_t* d_has_mp, const float K5[5], const oslam_sem_frame_t* d_frames, int total_obj,
                                      const uint8_t* const* d_mask_ptrs, int H, int W, int mask_pitch, int total_objmp, const float* d_objmp_Xw,
                                      const int32_t* d_objmp_obj, int total_joint, const int32_t* d_joint_kp, const int32_t* d_joint_obj, const float bounds[4],
                                      float invSigma2_0, void* stream);
int oslam_poseopt_semantic_results_device(const oslam_poseopt_t* h, const int32_t** d_n_semantic);
/* One-bit-per-pixel form of n instance masks (device pointers, rows `pitch` bytes apart): d_bits [n][H][ceil(W / 64)] uint64, bit i of word w = pixel
 * 64 w + i == 255 (the test of src/Frame.cc:266 and of the pcl cloud at src/ObjectOptimizer.cc:699-710).  One pass over the mask bytes; the keypoint test
 * (oslam_frame_object_kp_test_bits_batch_device, same semantics as the byte form below) and the boundary lists of the next
 * oslam_pose_optimize2_batch_device call (oslam_poseopt_use_mask_bits: pooled object o = bitmap d_bits_index[o]) then read words. */
int oslam_mask_bits_device(const uint8_t* const* d_mask_ptrs, int n, int H, int W, int pitch, uint64_t* d_bits, void* stream);
int oslam_poseopt_use_mask_bits(oslam_poseopt_t* h, const uint64_t* d_bits, const int32_t* d_bits_index);
int oslam_frame_object_kp_test_bits_batch_device(const oslam_keypoint_t* d_keysUn, int kp_stride, const int32_t* d_n_kps, int batch, const uint64_t* d_bits,
                                                 const int32_t* d_mask0, const int32_t* d_n_masks, int H, int W, uint8_t* d_out, void* stream);
/* Frame::BuildObject2DsRGBD / BuildObject2DsStereo keypoint test (src/Frame.cc:262-272, :336-346): bit o of d_out[b][k] is set iff every pixel
 * (int)(kpUn.y + row), (int)(kpUn.x + col), row / col in [-10, 10), of mask d_mask_ptrs[d_mask0[b] + o] equals 255 (a pixel outside the image fails: the
 * reference reads out of bounds there); at most 8 masks per frame.  The depth gate and the sequential assignment (:273-301) stay with the caller. */
int oslam_frame_object_kp_test_batch_device(const oslam_keypoint_t* d_keysUn, int kp_stride, const int32_t* d_n_kps, int batch, const uint8_t* const* d_mask_ptrs,
                                            const int32_t* d_mask0, const int32_t* d_n_masks, int H, int W, int mask_pitch, uint8_t* d_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Optimizer::LocalBundleAdjustment (include/Optimizer.h:45, src/Optimizer.cc:453-778) after the
 * covisibility gather (:456-504, pointer-graph walk, stays with the caller):
 * g2o Levenberg-Marquardt + BlockSolver_6_3 Schur complement, optimize(5) with Huber, chi2 (5.991 /
 * 7.815) and depth gating (:672-702), optimize(10) without robust kernel, erase list (:711-743).
 * poses [nKF][16] Tcw float32; fixed[k]: 0 local keyframe (free), 1 fixed camera (lFixedCameras),
 * 2 local keyframe with mnId==0 (setFixed(true) at :529 yet written back at :762-768);
 * points [nP][3]; one edge per observation: (keyframe, point, kpUn.pt.x, kpUn.pt.y, mvuRight (<0 mono),
 * mvInvLevelSigma2[octave]).  K5 = {fx, fy, cx, cy, mbf}.
 * Outputs: poses / points rounded to float32 like Converter::toCvMat, erase[e] = 1 for observations
 * the reference erases.  pbStopFlag (:517-518): oslam_lba_stop_flag() returns a pinned, device-visible
 * int the caller may set from another thread (LocalMapping::InterruptBA, src/LocalMapping.cc:628-631);
 * it is polled at the points g2o polls its force-stop flag when use_stop_flag != 0.
 * ---------------------------------------------------------------------------------------- */
typedef struct oslam_lba oslam_lba_t;
int oslam_lba_create(oslam_lba_t** out, int max_batch, int max_keyframes /* per window, fixed cameras included; at most 128 of them free */, int max_points,
                     int max_edges, int device);
void oslam_lba_destroy(oslam_lba_t* h);
volatile int32_t* oslam_lba_stop_flag(oslam_lba_t* h);
/* 1 (default): a single problem is spread over the whole GPU (multi-kernel LM, device-side control);
 * 2: one workgroup per problem, the whole LM schedule in ONE launch with the reduced camera system resident in LDS (the batch-of-windows layout of the
 *    driver; a window whose reduced system and poses exceed the CU's LDS — roughly 30 free keyframes — runs in mode 1 beside the others);
 * 0: the round-1 one-workgroup-per-problem kernel (reduced system in global memory), kept as an A/B layout.  Same arithmetic in all three. */
int oslam_lba_set_mode(oslam_lba_t* h, int wide);
/* Schur complement of mode 1: 0 = gather of the (W_a, B_b) pairs from memory (one wavefront per 6x6 block), 1 = on chip by tiles of points staged in LDS (one
 * coalesced read of the per-edge blocks per trial), 2 = chosen per call from the mean window size.  Same results to rounding.  The pair lists of mode 0 (the
 * default) are built on the device once per call (k_w_pair_*); 3 = mode 0 with the lists built by the host (the round-2 path): bit-identical results. */
int oslam_lba_set_schur(oslam_lba_t* h, int mode);
/* Kernel timing for bench.py's roofline: HIP events on the handle's stream around the solve kernels of every later call.
 * Returns and clears the accumulated milliseconds / kernel launches, then sets the switch to `enable`. */
int oslam_lba_kernel_time(oslam_lba_t* h, int enable, double* ms_out, long long* launches_out);
/* Test hook, as oslam_poseopt_trace: the LM trials of window 0 of the wide-mode calls that follow (g2o's OptimizationAlgorithmLevenberg::solve as
 * driven from src/Optimizer.cc:659-660,706-707); field 5 = first trial of a stage.  Process-wide: one handle traces at a time. */
int oslam_lba_trace(oslam_lba_t* h, int cap);
int oslam_lba_trace_read(oslam_lba_t* h, double* out, int32_t* n);
/* Reduced-camera-system solver of the wide mode: 0 (default) = the LDS-resident scalar kernel when EVERY system of the call has at most 132 unknowns
 * (6 x free keyframes), otherwise matrix cores (v_mfma_f64_16x16x4_f64 trailing updates, 16-wide panels): with the system resident in LDS up to 186 unknowns
 * (k_w_chol_lds_mfma: the windows of the driver's steady state), in global memory beyond; 1 = the global-memory matrix-core kernel for every size; 2 = no
 * matrix cores; 3 = the round-3 choice (scalar packed LDS kernel up to 192 unknowns, global-memory matrix cores beyond), kept as an A/B. */
int oslam_lba_set_solver(oslam_lba_t* h, int mode);
int oslam_lba_debug_stats(oslam_lba_t* h, int32_t out[16]);   /* [0..3] stats, [8..15] per-phase kilo-cycles in profiling builds */
int oslam_lba_optimize(oslam_lba_t* h, int nKF, const float* poses, const uint8_t* fixed, int nP,
                       const float* points, int nE, const int32_t* edge_kf, const int32_t* edge_pt,
                       const float* edge_obs, const float* edge_invSigma2, const float K5[5], int use_stop_flag,
                       float* poses_out, float* points_out, uint8_t* erase,
                       int32_t stats[4] /* iterations/trials of stage 1 and 2; may be NULL */);

/* Batch of independent local-BA problems (one workgroup each, ONE launch): the batch-of-sequences layout. */
typedef struct oslam_lba_problem {
    int32_t nKF; const float* poses; const uint8_t* fixed;
    int32_t nP; const float* points;
    int32_t nE; const int32_t* edge_kf; const int32_t* edge_pt; const float* edge_obs; const float* edge_invSigma2;
    float* poses_out; float* points_out; uint8_t* erase; int32_t* stats;   /* stats may be NULL */
} oslam_lba_problem_t;
/* A window of the batch that the solver refuses (an index out of range, a duplicate observation, more free keyframes than the bound ...) fails ALONE when it has
 * a stats array: stats = {-1, OSLAM_E_* code, 0, 0}, poses_out / points_out = the inputs, erase = 0, and the other windows are solved (return value OSLAM_OK).
 * Without a stats array such a window fails the call, as in rounds 1-4. */
int oslam_lba_optimize_batch(oslam_lba_t* h, int n, const oslam_lba_problem_t* probs, const float K5[5]);

/* Optimizer::BundleAdjustment (include/Optimizer.h:38, src/Optimizer.cc:49-237) on the same flattened graph:
 * one optimize(nIterations), Huber sqrt(5.99) / sqrt(7.815) only if bRobust, no gating, no erase list.
 * fixed[k] = 1 for the keyframe with mnId == 0 (:79). */
int oslam_ba_optimize(oslam_lba_t* h, int nKF, const float* poses, const uint8_t* fixed, int nP, const float* points, int nE,
                      const int32_t* edge_kf, const int32_t* edge_pt, const float* edge_obs, const float* edge_invSigma2,
                      const float K5[5], int nIterations, int bRobust, int use_stop_flag, float* poses_out, float* points_out);

/* ---------------- MapPoint maintenance + Frame::isInFrustum ----------------
 * Batched over map points. Observations of point p are rows obs_start[p] .. obs_start[p+1]-1 of the obs_* tables,
 * in the iteration order of the reference's std::map<KeyFrame*,size_t> (the caller keeps that order). */
typedef struct oslam_mappoint oslam_mappoint_t;
int oslam_mappoint_create(oslam_mappoint_t** out, int device);
void oslam_mappoint_destroy(oslam_mappoint_t* h);
/* Device time of the batched SearchByBoW / SearchForTriangulation kernel (oslam_match_bow_batch) and of the batched triangulation kernel
 * (oslam_mp_triangulate_pairs), HIP events on the handles' own streams: what accumulated since the last call, then the switch is set to `enable`. */
int oslam_bow_kernel_time(oslam_bow_t* h, int enable, double* ms_out, long long* launches_out);
int oslam_mappoint_kernel_time(oslam_mappoint_t* h, int enable, double* ms_out, long long* launches_out);

/* MapPoint::ComputeDistinctiveDescriptors (reference src/MapPoint.cc:345-410): for each point the observation whose
 * median Hamming distance to the others (sorted row, element int(0.5*(N-1))) is the first strict minimum.
 * best_idx[p] = index inside the point's list (-1 if it has no observation; out_desc row then zero). */
int oslam_mp_distinctive_descriptors(oslam_mappoint_t* h, int P, const int32_t* obs_start, const uint8_t* obs_desc /*[total][32]*/,
                                     int32_t* best_idx /*[P]*/, uint8_t* out_desc /*[P][32]*/);

/* MapPoint::UpdateNormalAndDepth (reference src/MapPoint.cc:433-474). obs_Ow = camera centres of the observing
 * keyframes, OwRef = reference keyframe centre, levelScaleFactor[p] = mvScaleFactors[level of the point in the
 * reference keyframe], lastScaleFactor = mvScaleFactors[nLevels-1].
 * out[p] = {normal.x, normal.y, normal.z, mfMaxDistance, mfMinDistance}. */
int oslam_mp_update_normal_depth(oslam_mappoint_t* h, int P, const float* Pos /*[P][3]*/, const int32_t* obs_start, const float* obs_Ow /*[total][3]*/,
                                 const float* OwRef /*[P][3]*/, const float* levelScaleFactor /*[P]*/, float lastScaleFactor, float* out /*[P][5]*/);

/* Resident map-point table: d_tab[slot] = device array of 64-byte records indexed by map-point id — float pos[3], normal[3], minDistance, maxDistance, then the
 * 32 descriptor bytes (mWorldPos, mNormalVector, mfMinDistance, mfMaxDistance, mDescriptor of include/MapPoint.h:116-141).  After a MapPoint update of P points
 * (the two functions below) this writes its results into the records d_items[i] = (slot, id): position (d_Pos) + normal / distances (d_out5) when do_normal,
 * the descriptor when do_desc and the point's descriptor list (d_desc_start) is not empty; a point without observations (culled) keeps its record except
 * for the position, which the caller may have changed before culling it (local BA). */
/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:345-430) and / or UpdateNormalAndDepth (:432-474) for P points in ONE launch, for the driver's small
 * updates: the descriptors of the observations are read from resident keyframe records (d_rec = (record, keypoint) per entry of the descriptor list d_desc_start,
 * d_rec_desc[record] = that keyframe's descriptor array), the results go to o_best / o_desc / o_out5 — any device-accessible memory, e.g. pinned host memory — and,
 * with d_tab, into the points' resident records like oslam_mp_table_write_device.  Same results as the separate functions.  At most 128 observations per point:
 * a point with a longer descriptor list gets o_best = -2 and no descriptor (neither in o_desc nor in its record) — use oslam_mp_distinctive_descriptors for it. */
int oslam_mp_update_fused_device(int P, int do_desc, int do_normal, const int32_t* d_obs_start, const int32_t* d_desc_start, const int32_t* d_rec, const uint8_t* const* d_rec_desc,
                                 const float* d_obs_Ow, const float* d_Pos, const float* d_OwRef, const float* d_lsf, float lastScale, const int32_t* d_items, uint8_t* const* d_tab,
                                 int32_t* o_best, uint8_t* o_desc, float* o_out5, void* stream);
/* MapPoint::UpdateNormalAndDepth (src/MapPoint.cc:432-474) for P points of solved local-BA windows, as Optimizer::LocalBundleAdjustment's write-back calls it
 * per point (src/Optimizer.cc:769-776), from the windows' own arrays (include/oslam_slam.h oslam_job_mp_window_t, flattened over the windows of a call): point j
 * has the edges d_e0[j] .. d_e0[j] + d_ne[j] of d_edge_kf / d_erase (window keyframe index, erased flag), its window's camera centres start at row d_kbase[j] of
 * d_Ow, d_ref[j] = window index of its reference keyframe, d_lsf[j] = that observation's level scale factor, d_skip[j] != 0 = position only.  Results as
 * oslam_mp_update_normal_depth in d_out5; with d_tab the resident records (d_items[2 j], d_items[2 j + 1]) are updated in the same pass. */
int oslam_mp_update_windows_device(int P, const int32_t* d_items, uint8_t* const* d_tab, const int32_t* d_e0, const int32_t* d_ne, const int32_t* d_kbase, const int32_t* d_ref,
                                   const float* d_lsf, const uint8_t* d_skip, const float* d_Pos, const int32_t* d_edge_kf, const uint8_t* d_erase, const float* d_Ow, float lastScale,
                                   float* d_out5, void* stream);
int oslam_mp_table_write_device(int P, const int32_t* d_items, uint8_t* const* d_tab, const int32_t* d_obs_start, const int32_t* d_desc_start, const float* d_Pos,
                                const float* d_out5, const uint8_t* d_out_desc, int do_desc, int do_normal, void* stream);
/* Inputs of Optimizer::PoseOptimization (src/Optimizer.cc:258-340) for a batch of frames that are still on the device: frame b is frame d_slots[b] of the
 * keypoint arrays d_keysUn / d_uRight ([.][kp_stride]) with d_n[b] keypoints; d_ids[b][i] = map point of keypoint i (record of d_tab[d_slots[b]]) or -1.
 * Writes Xw, obs = (x, y, uRight), invSigma2 = invLevelSigma2[octave] and has_mp in the [batch][stride] layout of oslam_pose_optimize_batch_device. */
int oslam_pose_inputs_gather_device(int batch, int stride, const int32_t* d_slots, const int32_t* d_n, const int32_t* d_ids, uint8_t* const* d_tab,
                                    const oslam_keypoint_t* d_keysUn, const float* d_uRight, int kp_stride, const float* invLevelSigma2, int nLevels,
                                    float* d_Xw, float* d_obs, float* d_invSigma2, uint8_t* d_has_mp, void* stream);
/* Position and descriptor of the map points d_ids[b][i] (records of d_tab[b]; -1 = none: zeros) into the [batch][stride] arrays that
 * oslam_match_project_last_batch_device reads (the last frame's mvpMapPoints of ORBmatcher::SearchByProjection(Cur, Last), src/ORBmatcher.cc:1338-1366). */
/* The projection gates of ORBmatcher::Fuse (src/ORBmatcher.cc:840-890) for n (keyframe, candidate list) jobs from the resident map-point records: job b has
 * d_M[b] candidates d_ids[b][i] (records of d_tab[d_slots[b]]; excluded when d_ids < 0 or d_excl != 0), keyframe pose d_Tcw[b] / centre d_Ow[b]; writes the
 * query of every candidate (inactive = flags 0) in candidate order into d_q[b][.] for oslam_match_fuse_batch_device. */
int oslam_fuse_queries_device(int n, int stride, const int32_t* d_slots, const int32_t* d_M, const int32_t* d_ids, const uint8_t* d_excl, uint8_t* const* d_tab,
                              const float* d_Tcw, const float* d_Ow, const float K5[5], const float bounds[4], float th, float logScaleFactor,
                              const float* scaleFactors, int nLevels, oslam_proj_query_t* d_q, void* stream);
/* Fuse against RESIDENT keyframes.  KeyFrame::mGrid is fixed at construction (src/KeyFrame.cc:44-52), so the grid of a registered keyframe is built ONCE:
 * oslam_kf_grid_build_device sorts the keypoints of job i (the d_counts[slot] keypoints at `keys` / `uRight`) by cell — cells in the (x, y) nesting of
 * Frame::mGrid, index order inside a cell (src/Frame.cc:455-470) — into cell_end[3072] (end of every cell in the sorted list) and cand[N][4] =
 * (x, y, uRight, bit pattern of octave << 16 | keypoint index).  *d_status is set to 1 if a count exceeds max_keypoints (never truncated silently).
 * oslam_fuse_search_device = oslam_fuse_queries_device + the window search of ORBmatcher::Fuse (src/ORBmatcher.cc:890-947: GetFeaturesInArea, level gate,
 * chi2 gates, best Hamming distance <= TH_LOW) with one candidate per thread, straight from those arrays: d_q_match[b][i] = keypoint index or -1, the same
 * values oslam_match_fuse_batch_device returns for the queries of oslam_fuse_queries_device (tests/test_mp_table_gpu.py). */
typedef struct oslam_kf_grid_job { const oslam_keypoint_t* keys; const float* uRight; uint16_t* cell_end; float* cand; int32_t slot; int32_t pad_; } oslam_kf_grid_job_t;
typedef struct oslam_kf_grid_ref { const uint16_t* cell_end; const float* cand; const uint8_t* desc; } oslam_kf_grid_ref_t;
int oslam_kf_grid_build_device(int n, const oslam_kf_grid_job_t* d_jobs, const int32_t* d_counts, const float bounds[4], int max_keypoints, int32_t* d_status, void* stream);
int oslam_fuse_search_device(int n, int stride, const oslam_kf_grid_ref_t* d_kfs, const int32_t* d_slots, const int32_t* d_M, const int32_t* d_ids, const uint8_t* d_excl,
                             uint8_t* const* d_tab, const float* d_Tcw, const float* d_Ow, const float K5[5], const float bounds[4], float th, float logScaleFactor,
                             const float* scaleFactors, const float* invLevelSigma2, int nLevels, int32_t* d_q_match, void* stream);
/* The arrays of Tracking::SearchLocalPoints for n frames from the resident records: job i = {slot, M, byte offset of its ids (int32 [M]) and of its
 * Observations() > 0 flags (uint8 [M]) inside d_stage}; writes position, normal, distances, flags and descriptor of every point into the [.][stride] arrays
 * at row `slot` (the layout oslam_frame_is_in_frustum_batch_resident_device reads). */
typedef struct oslam_local_gather { int32_t slot, M; uint32_t ids_off, obs_off; } oslam_local_gather_t;
int oslam_mp_table_local_gather_device(int n, int maxM, const oslam_local_gather_t* d_jobs, const uint8_t* d_stage, uint8_t* const* d_tab, int stride, float* d_Pw,
                                       float* d_Pn, float* d_maxDist, float* d_minDist, uint8_t* d_obs_gt0, uint8_t* d_mp_desc, void* stream);
/* Positions of n map points named by (d_slots[i], d_ids[i]) into d_Xw[n][3] (the object map points of ObjectOptimizer::PoseOptimization2). */
int oslam_mp_table_positions_device(int n, const int32_t* d_slots, const int32_t* d_ids, uint8_t* const* d_tab, float* d_Xw, void* stream);
int oslam_mp_table_gather_device(int batch, int stride, const int32_t* d_n, const int32_t* d_ids, uint8_t* const* d_tab, float* d_Xw, uint8_t* d_desc, void* stream);
/* device-pointer forms (asynchronous on `stream`); d_out_desc rows of points without observations are left untouched (zero-fill them first) */
int oslam_mp_distinctive_descriptors_device(int P, const int32_t* d_obs_start, const uint8_t* d_obs_desc, int32_t* d_best_idx, uint8_t* d_out_desc, void* stream);
int oslam_mp_update_normal_depth_device(int P, const float* d_Pos, const int32_t* d_obs_start, const float* d_obs_Ow, const float* d_OwRef,
                                        const float* d_levelScaleFactor, float lastScaleFactor, float* d_out, void* stream);

/* Frame::isInFrustum (reference src/Frame.cc:509-565) + MapPoint::PredictScale (src/MapPoint.cc:505-521) + the window
 * radius of ORBmatcher::SearchByProjection(F, vpMapPoints, th) (src/ORBmatcher.cc:57-67, RadiusByViewingCos :93-99),
 * for M map points against one frame pose. Tcw row-major 4x4; K5 = fx,fy,cx,cy,bf; bounds = mnMinX,mnMinY,mnMaxX,mnMaxY;
 * maxDist/minDist = the raw mfMaxDistance/mfMinDistance (the 1.2/0.8 invariance factors are applied inside);
 * obs_gt0[i] = pMP->Observations() > 0. Writes one oslam_proj_query_t per point, ready for
 * oslam_match_search_by_projection: flags bit0 = mbTrackInView, bit1 = obs>0, angle = mTrackViewCos,
 * minLevel = nPredictedLevel-1, maxLevel = nPredictedLevel, desc = the point's descriptor. Points outside the frustum get flags=0. */
int oslam_frame_is_in_frustum(oslam_mappoint_t* h, int M, const float* Pw /*[M][3]*/, const float* Pn /*[M][3]*/, const float* maxDist, const float* minDist,
                              const uint8_t* obs_gt0, const uint8_t* mp_desc /*[M][32]*/, const float Tcw[16], const float K5[5], const float bounds[4],
                              float viewingCosLimit, float logScaleFactor, const float* scaleFactors, int nLevels, float th, oslam_proj_query_t* out);
/* same, all per-point arrays and the output in device memory, asynchronous on `stream` (hipStream_t) */
int oslam_frame_is_in_frustum_device(int M, const float* d_Pw, const float* d_Pn, const float* d_maxDist, const float* d_minDist, const uint8_t* d_obs_gt0,
                                     const uint8_t* d_mp_desc, const float Tcw[16], const float K5[5], const float bounds[4], float viewingCosLimit,
                                     float logScaleFactor, const float* scaleFactors, int nLevels, float th, oslam_proj_query_t* d_out, void* stream);

/* batch of frames (Tracking::SearchLocalPoints for several sequences): per-point arrays [batch][stride], d_M[b] points of frame b are
 * tested against pose d_Tcw[b] (16 floats) with radius factor d_th[b]; d_out [batch][stride] feeds oslam_match_search_batch_device
 * (q_stride = stride); d_in_view [batch][stride] = mbTrackInView (may be NULL). */
int oslam_frame_is_in_frustum_batch_device(int batch, int stride, const int32_t* d_M, const float* d_Pw, const float* d_Pn, const float* d_maxDist,
                                           const float* d_minDist, const uint8_t* d_obs_gt0, const uint8_t* d_mp_desc, const float* d_Tcw, const float* d_th,
                                           const float K5[5], const float bounds[4], float viewingCosLimit, float logScaleFactor, const float* scaleFactors,
                                           int nLevels, oslam_proj_query_t* d_out, uint8_t* d_in_view, void* stream);

/* same with the per-point arrays at their own stride ([batch][stride_in], a table that keeps the local points of every sequence resident) while the
 * queries, in_view and the optional skip flags are [batch][stride_out], stride_out <= stride_in.  d_skip[b][i] != 0: point i of frame b already carries
 * mnLastFrameSeen == the frame's id and is not projected (src/Tracking.cc:1413-1427): inactive query, in_view 0. */
int oslam_frame_is_in_frustum_batch_resident_device(int batch, int stride_in, int stride_out, const int32_t* d_M, const float* d_Pw, const float* d_Pn,
                                                    const float* d_maxDist, const float* d_minDist, const uint8_t* d_obs_gt0, const uint8_t* d_mp_desc,
                                                    const uint8_t* d_skip, const float* d_Tcw, const float* d_th, const float K5[5], const float bounds[4],
                                                    float viewingCosLimit, float logScaleFactor, const float* scaleFactors, int nLevels,
                                                    oslam_proj_query_t* d_out, uint8_t* d_in_view, void* stream);

/* LocalMapping::CreateNewMapPoints, per-match numeric core (reference src/LocalMapping.cc:291-432):
 * parallax test, 4x4 DLT by cv::SVD (one-sided Jacobi) or KeyFrame::UnprojectStereo (src/KeyFrame.cc:615-631),
 * cheirality, chi2 reprojection gates (5.991 / 7.8 x sigma2[octave]; the second view uses the CURRENT keyframe's mbf
 * like the reference, :399) and the scale-consistency gate.  kf1 = mpCurrentKeyFrame, kf2[p] = the neighbours that
 * passed the baseline test (:246-263), matches of pair p = rows pair_start[p] .. pair_start[p+1]-1 of idx1/idx2
 * (SearchForTriangulation output order).  Map insertion (:408-430) stays with the caller. */
typedef struct oslam_tri_kf {
    float Tcw[16];   /* row-major 4x4: GetRotation / GetTranslation */
    float Twc[16];   /* row-major 4x4: Rwc = Rcw.t(), column 3 = GetCameraCenter */
    float fx, fy, cx, cy, invfx, invfy, mbf, mb;
    const oslam_keypoint_t* keysUn;   /* mvKeysUn */
    const oslam_keypoint_t* keys;     /* mvKeys (UnprojectStereo reads the raw keypoint) */
    const float* uRight;              /* mvuRight */
    const float* depth;               /* mvDepth */
    int32_t n_kps;
} oslam_tri_kf_t;
int oslam_mp_triangulate(oslam_mappoint_t* h, const oslam_tri_kf_t* kf1, int nPairs, const oslam_tri_kf_t* kf2, const int32_t* pair_start,
                         const int32_t* idx1, const int32_t* idx2, const float* scaleFactors, const float* levelSigma2, int nLevels,
                         float ratioFactor /* 1.5f*mfScaleFactor */, uint8_t* ok /*[M]*/, float* x3D /*[M][3]*/, int32_t* nnew);

/* Same numeric core for nPairs independent (current keyframe, neighbour) pairs, e.g. one pair per sequence of a batch: pair p = (kf1[p], kf2[p]). */
int oslam_mp_triangulate_pairs(oslam_mappoint_t* h, int nPairs, const oslam_tri_kf_t* kf1, const oslam_tri_kf_t* kf2, const int32_t* pair_start,
                               const int32_t* idx1, const int32_t* idx2, const float* scaleFactors, const float* levelSigma2, int nLevels,
                               float ratioFactor, uint8_t* ok /*[M]*/, float* x3D /*[M][3]*/);

/* ---------------- Frame construction between the extractor and the grid / matchers ----------------
 * Frame::UndistortKeyPoints (reference src/Frame.cc:644-675: cv::undistortPoints(mat, mat, mK, mDistCoef, Mat(), mK), OpenCV 3.2
 * cvUndistortPoints: fp64, 5 fixed-point iterations), Frame::ComputeImageBounds (:677-704) and Frame::ComputeStereoFromRGBD
 * (:883-904).  K4 = fx, fy, cx, cy (mK, CV_32F); dist = mDistCoef (k1, k2, p1, p2[, k3]; ndist 0 or dist[0] == 0 -> mvKeysUn = mvKeys,
 * bounds = the image rectangle).  The *_batch_device forms work on the extractor's device arrays ([batch][stride] keypoints,
 * counts per frame or n_const) and are asynchronous on `stream`; `image_stride` of the depth batch is in floats. */
typedef struct oslam_frame oslam_frame_t;
int oslam_frame_create(oslam_frame_t** out, int device);
void oslam_frame_destroy(oslam_frame_t* h);
int oslam_frame_undistort_keypoints(oslam_frame_t* h, int n, const oslam_keypoint_t* keys, const float K4[4], const float* dist, int ndist,
                                    oslam_keypoint_t* keysUn);
int oslam_frame_undistort_batch_device(const oslam_keypoint_t* d_keys, oslam_keypoint_t* d_keysUn, const int32_t* d_counts, int n_const, int stride,
                                       int batch, const float K4[4], const float* dist, int ndist, void* stream);
/* bounds = mnMinX, mnMinY, mnMaxX, mnMaxY (the order every matcher entry point takes) */
int oslam_frame_image_bounds(oslam_frame_t* h, int cols, int rows, const float K4[4], const float* dist, int ndist, float bounds[4]);
/* depth = imDepth after the mDepthMapFactor scaling (CV_32F, `pitch` floats per row); keys = mvKeys (raw pixel, truncated to
 * the depth pixel like Mat::at<float>(v,u)), keysUn = mvKeysUn; uRight / mvDepth = -1 where depth <= 0.  A keypoint outside the
 * depth image is an error (the reference would read out of bounds). */
int oslam_frame_stereo_from_rgbd(oslam_frame_t* h, int n, const oslam_keypoint_t* keys, const oslam_keypoint_t* keysUn, const float* depth, int rows,
                                 int cols, int pitch, float mbf, float* uRight, float* mvDepth);
/* n images given by a table of device pointers (rows src_pitch bytes apart) gathered into one contiguous batch [n][rows][dst_pitch]: the extractor's input
 * layout, in one launch. */
int oslam_frame_gather_images_device(const void* const* d_src_ptrs, int n, int src_pitch, int row_bytes, int rows, void* d_dst, size_t dst_image_stride, int dst_pitch,
                                     void* stream);
/* Device-to-device helpers of the driver's resident keyframe store: n segments {const void* src; void* dst; uint32 bytes; uint32 pad} copied in one launch;
 * d_out[i] = the 32-byte descriptor number d_rec[i][1] of the array d_desc_base[d_rec[i][0]]. */
int oslam_copy_segments_device(const void* d_segs, int n, void* stream);
int oslam_gather_descriptors_device(const uint8_t* const* d_desc_base, const int32_t* d_rec, int n, uint8_t* d_out, void* stream);
int oslam_frame_stereo_from_rgbd_batch_device(const oslam_keypoint_t* d_keys, const oslam_keypoint_t* d_keysUn, const int32_t* d_counts, int n_const,
                                              int stride, int batch, const float* d_depth, int rows, int cols, int pitch, size_t image_stride,
                                              float mbf, float* d_uRight, float* d_mvDepth, int32_t* d_status, void* stream);

/* Frame::ComputeStereoFromRGBD with the depth images where they are: image b = d_depth_ptrs[b] (device table of device pointers), rows `pitch` floats apart. */
int oslam_frame_stereo_from_rgbd_batch_ptrs_device(const oslam_keypoint_t* d_keys, const oslam_keypoint_t* d_keysUn, const int32_t* d_counts, int n_const,
                                                   int stride, int batch, const float* const* d_depth_ptrs, int rows, int cols, int pitch, float mbf,
                                                   float* d_uRight, float* d_mvDepth, int32_t* d_status, void* stream);

/* The same on RAW 16-bit depth images (what System::TrackRGBD receives, include/System.h:75): a value is scaled on lookup by depth_factor = mDepthMapFactor
 * as imDepth.convertTo(CV_32F, mDepthMapFactor) scales the whole image (reference src/Tracking.cc:98-102,262): (float)d16 * depth_factor in float arithmetic. */
int oslam_frame_stereo_from_rgbd_batch_ptrs_u16_device(const oslam_keypoint_t* d_keys, const oslam_keypoint_t* d_keysUn, const int32_t* d_counts, int n_const,
                                                       int stride, int batch, const uint16_t* const* d_depth16_ptrs, int rows, int cols, int pitch, float depth_factor,
                                                       float mbf, float* d_uRight, float* d_mvDepth, int32_t* d_status, void* stream);

/* Two-level nearest-centre descent over 256-bit descriptors (the node assignment of a DBoW2-style vocabulary tree with branching 10, depth 2):
 * d_out[i][k] = 11 + 10 b1 + b2 for descriptor k of array d_desc_ptrs[i] (d_counts[i] of them, at most `stride`), first minimum on ties. */
int oslam_bow_nodes_device(const uint8_t* const* d_desc_ptrs, const int32_t* d_counts, int n, int stride, const uint64_t* d_top, const uint64_t* d_sub,
                           uint32_t* d_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OSLAM_HIP_H */
