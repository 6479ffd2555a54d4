/*
 * oslam_slam.h — C ABI of the batch-of-sequences tracking + local-mapping driver.
 *
 * S independent RGB-D sequences advance in lockstep on one GPU: every stage of the reference's per-frame call
 * pattern (Tracking::GrabImageRGBD -> Track, src/Tracking.cc:241-587; LocalMapping::Run, src/LocalMapping.cc:48-113)
 * is executed for all sequences as ONE batch of the hot-path operators of oslam_hip.h.  The map bookkeeping between
 * the operator calls (MapPoint / KeyFrame / covisibility graph / spanning tree) is host C++ over flat, index-based
 * arrays.  The driver reproduces the reference's control flow for the STEREO/RGBD SLAM mode:
 *   StereoInitialization :590-642, CheckReplacedInLastFrame :820, TrackReferenceKeyFrame :838, UpdateLastFrame :882,
 *   TrackWithMotionModel :948, TrackLocalMap :1011 (UpdateLocalKeyFrames :1496, UpdateLocalPoints :1470,
 *   SearchLocalPoints :1408), NeedNewKeyFrame :1242, CreateNewKeyFrame :1328, and on the mapping side
 *   ProcessNewKeyFrame (src/LocalMapping.cc:129), MapPointCulling :171, CreateNewMapPoints :208, SearchInNeighbors :455,
 *   Optimizer::LocalBundleAdjustment's graph gather / write-back (src/Optimizer.cc:456-504, :746-777), KeyFrameCulling :633.
 * Normalisations (the reference is not deterministic there):
 *   - LocalMapping runs at fixed points of the step instead of on a second thread (src/System.cc:95): AcceptKeyFrames() is therefore always true and the BA
 *     is never interrupted.  Two deterministic schedules (oslam_slam_config_t::local_mapping bit 5, OSLAM_SLAM_LM_DEFERRED):
 *       synchronous — the whole pass (LocalMapping::Run, src/LocalMapping.cc:48-113) right after the frame that inserted the keyframe;
 *       deferred    — ProcessNewKeyFrame .. SearchInNeighbors and the local-BA graph gather (src/Optimizer.cc:456-504) right after the frame that inserted
 *                     the keyframe; Optimizer::LocalBundleAdjustment's solve runs while the NEXT frame is tracked (the reference's overlap of the two threads)
 *                     and its write-back (:746-777), the MapPoint updates and KeyFrameCulling (:633) are applied after that frame's tracking, before the next
 *                     local-mapping pass.  Tracking of frame t+1 thus reads the map as the reference's tracking thread does while the BA of keyframe t is
 *                     still running.  oslam_slam_finish applies what is pending (System::Shutdown waits for the local mapper, src/System.cc:303-320);
 *                     the trajectory getters call it.
 *   - containers ordered by pointer value (std::map<KeyFrame*,..>, std::set<KeyFrame*>, pair<int,KeyFrame*> sorts)
 *     are ordered by keyframe id;
 *   - DBoW2 and its vocabulary are not in the reference tree: ComputeBoW uses a substitute vocabulary (k = 10, two
 *     levels of seeded random 256-bit words, nearest child by Hamming distance) that yields the same FeatureVector
 *     structure (node id at the 4th level from the leaves -> keypoint indices);
 *   - out of scope: Relocalization (a sequence lost with more than 5 keyframes stays LOST; with <= 5 the
 *     system resets like src/Tracking.cc:553-561 and re-initialises on the next frame), loop closing, the object layer
 *     (ObjectMatcher, Object3D outlier rejection, ObjectMapRegularization), the viewer.
 * Semantic constraints (BASELINE.json configs[2]): oslam_slam_track_rgbd_objects / _stereo_objects take the frame's instance masks.  The driver
 * builds the Object2Ds (Frame::BuildObject2DsRGBD, src/Frame.cc:240-312: 20x20 mask window test, depth gate, > 5 keypoints) and calls
 * ObjectOptimizer::PoseOptimization2 in TrackLocalMap (src/Tracking.cc:1022) with the M_joint / M_semantic edges of src/ObjectOptimizer.cc:687-1100.
 * What the out-of-scope object layer would supply is replaced by a minimal substitute, stated here so that nobody mistakes it for the reference's:
 *   - association (Tracking::TrackObject: ObjectMatcher::MatchTwoFrame / MatchMapToFrame) = the caller's `track_id` per detection: mvpObject3Ds[o] is
 *     the Object3D created earlier for that id (PoseOptimization2 without matched objects is PoseOptimization);
 *   - Object3D bookkeeping (Tracking::UpdateCurrentObject :1079-1210, Object3D::Update src/ObjectTypes.cc:56-140) keeps its list logic — a new
 *     Object3D from the Object2D's keypoints with map points when there are more than MIN_OBJ3DMP_NUM = 5 of them, later frames append the
 *     non-bad, non-outlier map points not yet listed — without the PCL Euclidean clustering / RejectOutliers steps.
 */
/* Limits of the HIP operator table (the reference's containers are unbounded):
 *   - keypoints per frame: the extractor's capacity (sum of the level quotas + slack, oslam_orb_max_keypoints) — never exceeded by construction;
 *   - local map points searched per frame (Tracking::SearchLocalPoints): no fixed bound, the matcher's query buffers grow on demand;
 *   - local BA: keyframes, points and edges per window grow on demand (every fixed keyframe enters the window, src/Optimizer.cc:489-504); at most 128 LOCAL
 *     (free) keyframes per window, i.e. 768 unknowns of the reduced camera system — a window beyond that is solved DEGRADED: the current keyframe and its
 *     strongest covisible keyframes (KeyFrame::GetVectorCovisibleKeyFrames order, weight-descending) stay free up to the bound, the weaker local keyframes
 *     enter as fixed cameras with all their points and edges; such windows are counted (oslam_slam_lba_window_stats [5]);
 *   - detections per frame: OSLAM_SLAM_MAX_OBJECTS;
 *   - resident keyframe records (keypoints, descriptors, stereo coordinates, feature grid: ~86 B x capacity + 6 KB each): one per keyframe in the current map
 *     of a sequence; the records of culled keyframes (release_keyframes, after KeyFrameCulling) and of a map that was reset are reused; the store itself only
 *     grows and is released with the handle.
 * Per-sequence failure isolation: a local-BA window the operator refuses (malformed or beyond its bounds; oslam_lba_problem_t::stats[0] < 0, include/oslam_hip.h)
 * fails ITS sequence only — the sequence is counted (oslam_slam_lba_window_stats [7]), reports LOST for that frame and starts a new map with its next frame, the
 * way one System of the reference resets itself (src/Tracking.cc:553-560) — while the other sequences of the handle, and the other handles that shared the
 * local-BA batch, continue with unchanged results.  Device / runtime errors (OSLAM_E_HIP: no device, out of memory, a failed launch) are not per-sequence: they
 * abort the lockstep step of ALL sequences of the handle; the map bookkeeping of that step has then partly run, so the handle must be discarded. */
#ifndef OSLAM_SLAM_H
#define OSLAM_SLAM_H

#include "oslam_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oslam_slam oslam_slam_t;

/* Settings the reference reads from the YAML file (src/Tracking.cc:60-170). */
typedef struct oslam_slam_config {
    int32_t width, height;
    float fx, fy, cx, cy;        /* Camera.fx .. */
    float dist[5]; int32_t ndist; /* Camera.k1,k2,p1,p2[,k3] */
    float bf;                    /* Camera.bf */
    float thDepth;               /* ThDepth: mThDepth = bf * ThDepth / fx (:159) */
    float fps;                   /* Camera.fps: mMaxFrames = fps, mMinFrames = 0 (:109-110) */
    int32_t nFeatures; float scaleFactor; int32_t nLevels, iniThFAST, minThFAST;   /* ORBextractor.* */
    int32_t n_sequences;         /* S: sequences advanced per call */
    int32_t device;
    int32_t host_threads;        /* worker threads for the per-sequence bookkeeping (0 = 1) */
    int32_t local_mapping;       /* bit0 MapPointCulling, bit1 CreateNewMapPoints, bit2 SearchInNeighbors, bit3 LocalBundleAdjustment,
                                    bit4 KeyFrameCulling; 0x1F = the reference's LocalMapping::Run; bit5 (OSLAM_SLAM_LM_DEFERRED): the deferred schedule of
                                    the head comment (local BA solved during the next frame's tracking) */
    int32_t sensor;              /* System::eSensor (include/System.h:60-64): 1 STEREO, 2 RGBD; 0 = RGBD */
} oslam_slam_config_t;

#define OSLAM_SLAM_LM_DEFERRED 0x20
#define OSLAM_SLAM_NOT_INITIALIZED 1   /* Tracking::eTrackingState, include/Tracking.h:91-97 */
#define OSLAM_SLAM_OK 2
#define OSLAM_SLAM_LOST 3

/* ------------------------------------------------------------------------------------------------------------
 * Operator table: the batched hot-path calls the driver makes.  oslam_slam_create() binds the HIP implementation
 * (slam_ops_hip.hip: the kernels behind oslam_hip.h, chained on the device); the table is a public type only so that
 * tests can run the same driver over another implementation and compare trajectories.  All pointers are host
 * pointers unless a field says otherwise; every call is synchronous (results are in the job structs on return).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct oslam_slam_frame {          /* what Frame::Frame leaves behind (src/Frame.cc:117-172) */
    int32_t N;
    oslam_keypoint_t* keys;                /* mvKeys   [cap] (a table with oslam_slam_ops_t::keyframe_raw_keys may leave them untouched) */
    oslam_keypoint_t* keysUn;              /* mvKeysUn [cap] */
    uint8_t* desc;                         /* mDescriptors [cap][32] (a table with oslam_slam_ops_t::frame_descriptors may leave them untouched) */
    float* uRight;                         /* mvuRight [cap] */
    float* depth;                          /* mvDepth  [cap] */
} oslam_slam_frame_t;

typedef struct oslam_job_search_last {     /* ORBmatcher::SearchByProjection(Cur, Last, th, bMono), src/ORBmatcher.cc:1328 */
    int32_t slot;                          /* sequence index: cur is the frame built for this slot in this step */
    const oslam_slam_frame_t* cur;
    int32_t Nlast; const float* Xw; const uint8_t* has_mp; const oslam_keypoint_t* last_keysUn; const uint8_t* mp_desc;
    float Tcw[16], Tlw[16]; float th;
    int32_t* kp_match;                     /* out [cur->N]: last-frame keypoint whose map point is now in mvpMapPoints[k]; < 0 none */
    int32_t nmatches;                      /* out */
    const int32_t* mp_ids;                 /* [Nlast] map-point id of the last frame's keypoints or -1, or NULL.  A table that answers resident_points()
                                            * takes Xw and mp_desc from its records and last_keysUn from the previous step's frame of the slot, which it
                                            * still holds (the last frame of a tracked sequence IS the previous step's frame); those three may then be
                                            * NULL.  has_mp is always given (it carries Observations() > 0, which only the driver knows). */
} oslam_job_search_last_t;

typedef struct oslam_job_search_local {    /* Frame::isInFrustum(pMP, 0.5) over the local points + SearchByProjection(F, points, th), nnratio 0.8 */
    int32_t slot; const oslam_slam_frame_t* cur; const uint8_t* blocked /* [N] mvpMapPoints[i] && Observations()>0 */;
    int32_t M; const float* Pw; const float* Pn; const float* maxDist; const float* minDist; const uint8_t* obs_gt0; const uint8_t* mp_desc;
    float Tcw[16]; float th;
    const uint8_t* skip;                   /* [M] or NULL: != 0 = mnLastFrameSeen == this frame (already matched, or an outlier of the initial pose
                                            * optimisation): the reference does not project it (src/Tracking.cc:1413-1427) — in_view 0, never matched */
    int64_t content_id;                    /* identity of the CONTENT of Pw .. mp_desc (not of the pointers), 0 = none: a table that keeps the arrays of the
                                            * slot's previous job resident may skip the upload when the id is the same; the pointers stay valid either way */
    uint8_t* in_view;                      /* out [M]: mbTrackInView */
    int32_t* kp_match;                     /* out [N]: local point index now in mvpMapPoints[k]; < 0 none */
    int32_t nmatches;
    const int32_t* local_ids;              /* [M] map-point ids of the local points or NULL: with resident_points() Pw, Pn, maxDist, minDist and mp_desc
                                            * come from the table's records and may be NULL (obs_gt0 is always given: only the driver knows it) */
} oslam_job_search_local_t;

typedef struct oslam_job_pose {            /* Optimizer::PoseOptimization, src/Optimizer.cc:239 */
    int32_t slot; int32_t N; float Tcw_in[16];
    const float* Xw; const float* obs; const float* invSigma2; const uint8_t* has_mp;
    float Tcw_out[16]; uint8_t* outlier; int32_t n_inliers;   /* out */
    const int32_t* mp_ids;                 /* [N] map-point id of every keypoint of the slot's CURRENT frame or -1 (mvpMapPoints), or NULL.  A table that
                                            * answers resident_points() reads the positions from its records and obs / invSigma2 from the frame it still
                                            * holds; the four arrays above may then be NULL.  Other tables get the arrays and ignore this field. */
} oslam_job_pose_t;

typedef struct oslam_job_mp_update {       /* MapPoint::ComputeDistinctiveDescriptors + UpdateNormalAndDepth over P points (CSR observations) */
    int32_t P; const int32_t* obs_start; const uint8_t* obs_desc; const float* obs_Ow;
    const float* Pos; const float* OwRef; const float* levelScaleFactor;
    int32_t do_desc, do_normal;
    const int32_t* items;                  /* [P][2] or NULL: (slot, map-point id) of every point — a table that keeps the map points resident (position,
                                            * normal, distances, descriptor) writes the results into their records; every change of those fields goes
                                            * through this job (new points, local BA, fusions), so the records always equal the host's */
    const int32_t* desc_start;             /* [P + 1] or NULL (= obs_start): CSR of the observations whose keyframe is NOT bad, the only ones
                                            * ComputeDistinctiveDescriptors uses (src/MapPoint.cc:362-368; UpdateNormalAndDepth uses all of them, :441-453).
                                            * obs_desc — and the keys of mp_update_keyed — follow THIS layout, obs_Ow follows obs_start.  A point whose
                                            * list is empty keeps its descriptor (:370-371). */
    int32_t* best_idx; uint8_t* out_desc; float* out5;      /* out: see oslam_mp_distinctive_descriptors / oslam_mp_update_normal_depth */
} oslam_job_mp_update_t;

typedef struct oslam_job_mp_window {       /* MapPoint::UpdateNormalAndDepth (src/MapPoint.cc:432-474) for the points of ONE solved local-BA window, straight from the
                                            * window's own arrays — what Optimizer::LocalBundleAdjustment's write-back does per point (src/Optimizer.cc:769-776:
                                            * SetWorldPos, UpdateNormalAndDepth) after the outlier observations were erased (:752-757).  The observations of a point
                                            * are the window's edges of that point that were not erased: the driver lists a point here only when that holds (every
                                            * observing keyframe is in the window and alive); the others go through mp_update as before. */
    int32_t slot;                          /* sequence slot: the resident records (slot, pt_ids[j]) are updated like oslam_job_mp_update_t::items */
    int32_t nP, nE, nK;
    const int32_t* pt_ids;                 /* [nP] map point ids */
    const int32_t* pt_start;               /* [nP + 1] the edges of point j are pt_start[j] .. pt_start[j + 1], in the order of its observation list */
    const int32_t* edge_kf;                /* [nE] window index of the observing keyframe */
    const uint8_t* erase;                  /* [nE] != 0: the observation was erased by the write-back */
    const uint8_t* skip;                   /* [nP] != 0: position only (the point went bad in the write-back, or is updated through mp_update) */
    const int32_t* ref_kf;                 /* [nP] window index of the point's reference keyframe (after the write-back) */
    const float* lsf;                      /* [nP] mvScaleFactors[octave of the reference keyframe's observation] */
    const float* Ow;                       /* [nK][3] camera centres after the write-back */
    const float* pos;                      /* [nP][3] positions after the write-back */
    float* out5;                           /* out [nP][5]: normal, mfMaxDistance, mfMinDistance (oslam_mp_update_normal_depth); untouched rows for skipped points */
} oslam_job_mp_window_t;

typedef struct oslam_job_fuse {            /* search half of ORBmatcher::Fuse on one keyframe, src/ORBmatcher.cc:888-947 */
    int32_t N; const oslam_keypoint_t* keysUn; const float* uRight; const uint8_t* desc;
    int32_t M; const oslam_proj_query_t* queries;
    int32_t* q_match;                      /* out [M] */
} oslam_job_fuse_t;

typedef struct oslam_job_fuse_pts {        /* ORBmatcher::Fuse on one resident keyframe with the candidates named by map-point id (tables with resident_points):
                                            * the projection gates of src/ORBmatcher.cc:840-890 run on the device from the points' records */
    int32_t slot, kf, N;                   /* the target keyframe (registered with register_keyframes) and its keypoint count */
    int32_t M; const int32_t* ids;         /* candidates: map-point ids, -1 = none */
    const uint8_t* excl;                   /* [M] != 0: the point is bad or already observed by the keyframe (:849) — only the driver knows */
    float Tcw[16], Ow[3]; float th;        /* the keyframe's pose and camera centre, the radius factor (3.0) */
    int32_t* q_match;                      /* out [M]: keypoint of the keyframe for candidate i or -1 */
} oslam_job_fuse_pts_t;

#define OSLAM_SLAM_MAX_OBJECTS 8           /* detections per frame (one bit each in the keypoint test) */

typedef struct oslam_slam_objects {        /* the semantic detections of ONE frame (reference include/Semantic.h; src/Semantic.cc:14-96) */
    int32_t n;                             /* <= OSLAM_SLAM_MAX_OBJECTS, in the order of the semantic file */
    const uint8_t* const* masks;           /* n images of width x height, uint8 {0, 255}, rows mask_stride bytes apart (host or device like the frames);
                                            * mask_stride = 0: one-bit-per-pixel images, see oslam_slam_track_rgbd_raw16 */
    const int32_t* track_id;               /* n: identity of the physical object (substitute for ObjectMatcher, see the head comment); < 0 = unknown */
    const int32_t* label;                  /* n: class label (recorded, not used on this path); may be NULL */
} oslam_slam_objects_t;

typedef struct oslam_job_object_kps {      /* keypoint test of Frame::BuildObject2DsRGBD (src/Frame.cc:262-272) for the frame built for `slot` */
    int32_t slot; const oslam_slam_frame_t* cur;
    int32_t n_masks; const uint8_t* const* masks; int32_t mask_stride; int32_t on_device;
    uint8_t* in_mask;                      /* out [cur->N]: bit o = every pixel of the 20x20 window around mvKeysUn[k] equals 255 in mask o */
} oslam_job_object_kps_t;

typedef struct oslam_job_pose2 {           /* ObjectOptimizer::PoseOptimization2 (src/ObjectOptimizer.cc:624-1240), see oslam_semantic_t in oslam_hip.h */
    oslam_job_pose_t base;
    int32_t nObj; const uint8_t* const* masks; int32_t mask_stride; int32_t on_device;   /* Object2D masks of the matched objects, idx_obj order */
    int32_t nObjMp; const float* objmp_Xw; const int32_t* objmp_obj;
    int32_t nJoint; const int32_t* joint_kp; const int32_t* joint_obj;
    int32_t n_semantic;                    /* out: nSemNum (:1232) */
    const int32_t* objmp_ids;              /* [nObjMp] map-point ids of the objects' points or NULL: with resident_points() their positions come from the
                                            * table's records and objmp_Xw may be NULL */
} oslam_job_pose2_t;

typedef oslam_bow_job_t oslam_job_bow_t;
typedef struct oslam_kf_key { int32_t slot, kf1, kf2; } oslam_kf_key_t;   /* identity of the keyframes of a job (see register_keyframes) */   /* ORBmatcher::SearchByBoW(KF, F) (:159) or SearchForTriangulation (:657), see oslam_hip.h */

typedef struct oslam_job_triangulate {     /* oslam_mp_triangulate for one (current keyframe, neighbour) pair */
    oslam_tri_kf_t kf1, kf2; int32_t M; const int32_t* idx1; const int32_t* idx2;
    uint8_t* ok; float* x3D;
} oslam_job_triangulate_t;

#define OSLAM_SLAM_KT_GROUPS 8   /* kernel-time groups: 0 Frame::Frame (extraction .. stereo), 1 pose optimisation, 2 local BA, 3 window searches (SearchByProjection /
                                  * SearchLocalPoints), 4 ORBmatcher::Fuse (projection gates + window search), 5 SearchByBoW / SearchForTriangulation + triangulation,
                                  * 6 MapPoint updates (descriptor gather, ComputeDistinctiveDescriptors, UpdateNormalAndDepth, record writes), 7 other (keyframe
                                  * registration copies, vocabulary nodes, object keypoint tests / mask bitmaps, local-map gathers) */

/* Changes of one sequence's map for the table's device mirror (oslam_slam_ops_t::map_journal).  Everything that only needs its final state carries CURRENT values;
 * the observation events are in program order and for one (kf, idx) the LAST one decides (an erase clears the cell whoever held it).
 *   reset        != 0: the sequence started a new map before these changes (Tracking::Reset): ids restart at 0
 *   new_kfs      n_new keyframes created since the last call: id, keypoint count, KeyFrame::mvpMapPoints as it is now, and one bit per keypoint
 *                !(mvDepth[i] > mThDepth || mvDepth[i] < 0) (src/LocalMapping.cc:663-667; keypoints, depths and mThDepth never change)
 *   cells        n_cells x (kf, idx, p): KeyFrame::mvpMapPoints[idx] is now p (-1: none)                        (src/KeyFrame.cc:201-230)
 *   events       n_events x (kf, idx | set << 31, p): MapPoint p gained (set) / lost the observation (kf, idx)  (src/MapPoint.cc:196-318)
 *   points       n_points x (p, Observations(), isBad(), histogram lo, histogram hi): byte o of the 64-bit histogram = observations of p at octave o */
typedef struct oslam_map_new_kf { int32_t kf, N; const int32_t* mp; const uint32_t* good; } oslam_map_new_kf_t;
typedef struct oslam_map_changes {
    int32_t slot, reset;
    int32_t n_new; const oslam_map_new_kf_t* new_kfs;
    int32_t n_cells; const int32_t* cells;
    int32_t n_events; const uint32_t* events;
    int32_t n_points; const uint32_t* points;
} oslam_map_changes_t;
/* KeyFrameCulling candidates of one sequence (oslam_slam_ops_t::kf_culling_counts): n keyframe ids in the reference's order, out [n][4]. */
typedef struct oslam_job_cull { int32_t slot, n; const int32_t* kf_ids; int32_t* out; } oslam_job_cull_t;

/* LocalMapping::SearchInNeighbors, second direction (src/LocalMapping.cc:492-515), for one sequence: the map points of the target keyframes are fused into the
 * current keyframe.  The table builds vpFuseCandidates itself — the targets' point lists in the given order, bad points skipped, every point once
 * (mnFuseCandidateForKF) — from its mirror of the observation graph (map_journal), runs ORBmatcher::Fuse's gates and search for them and returns the matches:
 * pairs[2 q] = map point, pairs[2 q + 1] = keypoint of the current keyframe, in candidate order; the driver applies them (Replace / AddObservation) as it applies the
 * match table of fuse_points_keyed.  overflow != 0: more candidates or matches than the table's bounds — the driver then takes its own path for the whole call.
 * dbg_ids / dbg_excl (optional, tests): the candidate list and its "bad or already observed by the keyframe" flags, at most dbg_cap entries. */
typedef struct oslam_job_fuse_cur {
    int32_t slot, kf;
    int32_t n_targets; const int32_t* targets;
    float Tcw[16], Ow[3], th;
    int32_t max_pairs; int32_t* pairs;
    int32_t n_pairs, n_candidates, overflow;      /* out */
    int32_t dbg_cap; int32_t* dbg_ids; uint8_t* dbg_excl;
} oslam_job_fuse_cur_t;

typedef struct oslam_job_local_list { int32_t slot, n_kfs; const int32_t* kfs; int32_t cap; int32_t* ids; int32_t n_ids, overflow; } oslam_job_local_list_t;

typedef struct oslam_slam_ops {
    void* ctx;
    /* capacity of the per-frame arrays the driver must allocate */
    int (*max_keypoints)(void* ctx);
    /* scale tables of the extractor (include/ORBextractor.h:63-83) */
    int (*scale_tables)(void* ctx, float* scale, float* invScale, float* sigma2, float* invSigma2);
    /* Frame::ComputeImageBounds */
    int (*image_bounds)(void* ctx, float bounds[4]);
    /* Frame::Frame for n RGB-D frames: ExtractORB + UndistortKeyPoints + ComputeStereoFromRGBD.  gray[i]: width x height u8 with row
     * pitch gray_stride; depth[i]: CV_32F metres with row pitch depth_pitch floats; on_device != 0: both are device pointers. */
    int (*frames_rgbd)(void* ctx, int n, const int32_t* slots, const uint8_t* const* gray, int gray_stride, const float* const* depth,
                       int depth_pitch, int on_device, oslam_slam_frame_t* const* out);
    int (*search_last)(void* ctx, int n, oslam_job_search_last_t* jobs);
    int (*search_local)(void* ctx, int n, oslam_job_search_local_t* jobs);
    int (*pose_opt)(void* ctx, int n, oslam_job_pose_t* jobs);
    int (*mp_update)(void* ctx, oslam_job_mp_update_t* job);
    int (*lba)(void* ctx, int n, const oslam_lba_problem_t* probs);
    int (*fuse)(void* ctx, int n, oslam_job_fuse_t* jobs);
    int (*bow)(void* ctx, int n, oslam_job_bow_t* jobs);
    int (*triangulate)(void* ctx, int n, oslam_job_triangulate_t* jobs);
    void (*destroy)(void* ctx);
    /* Frame::Frame for n rectified stereo pairs (src/Frame.cc:61-115): two ExtractORB + ComputeStereoMatches (:706-880); NULL if unsupported */
    int (*frames_stereo)(void* ctx, int n, const int32_t* slots, const uint8_t* const* left, const uint8_t* const* right, int gray_stride,
                         int on_device, oslam_slam_frame_t* const* out);
    /* semantic constraints (NULL = the table cannot run frames with objects) */
    int (*object_kps)(void* ctx, int n, oslam_job_object_kps_t* jobs);
    int (*pose_opt2)(void* ctx, int n, oslam_job_pose2_t* jobs);
    /* Resident keyframes (optional, NULL in tables without a device).  register_keyframes: the frames built for `slots` in THIS step have become keyframes
     * kf_ids of their sequences: the table keeps a device copy of their keypoints / descriptors / stereo coordinates (a device-to-device copy while the
     * frame arrays are still in HBM; kf_id 0 restarts the sequence after a reset).  The *_keyed forms do what bow / fuse / mp_update do, with the identity
     * of the keyframes beside the jobs, so their arrays are read from the resident copies instead of being packed and uploaded again (kf2 = -2: side 2 is
     * the current frame of `slot`; obs_key [total][3] = (slot, keyframe id, keypoint index) of every observation). */
    int (*register_keyframes)(void* ctx, int n, const int32_t* slots, const int32_t* kf_ids);
    int (*bow_keyed)(void* ctx, int n, oslam_job_bow_t* jobs, const oslam_kf_key_t* keys);
    int (*fuse_keyed)(void* ctx, int n, oslam_job_fuse_t* jobs, const oslam_kf_key_t* keys);
    int (*mp_update_keyed)(void* ctx, oslam_job_mp_update_t* job, const int32_t* obs_key);
    /* optional (NULL in tables without a device): see oslam_slam_kernel_times */
    int (*kernel_times)(void* ctx, int enable, double out[OSLAM_SLAM_KT_GROUPS * 3]);
    /* optional, with register_keyframes: the FeatureVector node of every descriptor of registered keyframes (slot, kf_id), read from the resident copy.  The
     * driver's substitute for the DBoW2 vocabulary that is not in the reference tree (KeyFrame::ComputeBoW, src/KeyFrame.cc:66-76) is a two-level tree of
     * 10 x 10 256-bit centres: node = 11 + 10 b1 + b2 with b1 the nearest of top[10] and b2 the nearest of sub[b1][10] under the Hamming distance, first
     * on ties.  counts[i] descriptors of keyframe i -> out[i][0 .. counts[i]). */
    int (*bow_nodes_keyed)(void* ctx, int n, const int32_t* slots, const int32_t* kf_ids, const uint64_t* top /* [10][4] */, const uint64_t* sub /* [10][10][4] */,
                           const int32_t* counts, uint32_t* const* out);
    /* optional: != 0 if the table keeps the map points resident (oslam_job_mp_update_t::items) and serves pose jobs from mp_ids */
    int (*resident_points)(void* ctx);
    /* optional, with resident_points and register_keyframes: the search half of ORBmatcher::Fuse with candidates by id (oslam_job_fuse_pts_t) */
    int (*fuse_points_keyed)(void* ctx, int n, oslam_job_fuse_pts_t* jobs);
    /* optional test hook of tables with resident map points: the 64-byte record of point `id` of `slot` (see oslam_job_mp_update_t::items) */
    int (*point_record)(void* ctx, int slot, int id, uint8_t out[64]);
    /* optional: frames_rgbd on RAW 16-bit depth images (depth16[i]: width x height uint16, rows depth_pitch ELEMENTS apart), scaled by depth_factor =
     * mDepthMapFactor on lookup exactly as imDepth.convertTo(CV_32F, mDepthMapFactor) scales the image (src/Tracking.cc:262).  NULL = oslam_slam_track_rgbd_raw16
     * is refused on this table. */
    int (*frames_rgbd_raw16)(void* ctx, int n, const int32_t* slots, const uint8_t* const* gray, int gray_stride, const uint16_t* const* depth16,
                             int depth_pitch, float depth_factor, int on_device, oslam_slam_frame_t* const* out);
    /* optional, with register_keyframes: the keyframes (slots[i], kf_ids[i]) were culled (KeyFrame::SetBadFlag, src/KeyFrame.cc:441-509) — no later job names
     * them (the driver skips bad keyframes wherever the reference does, and ComputeDistinctiveDescriptors skips their observations, src/MapPoint.cc:366): a
     * table may give their resident records to later keyframes.  Called once per local-mapping pass, after KeyFrameCulling. */
    int (*release_keyframes)(void* ctx, int n, const int32_t* slots, const int32_t* kf_ids);
    /* optional pair, used by the deferred local-mapping schedule (OSLAM_SLAM_LM_DEFERRED): lba_submit starts the local BA of n windows and returns at once — the
     * arrays `probs` names (inputs AND outputs) stay valid and untouched until lba_wait, which returns when their results are in place (same results as `lba` on
     * the same windows: a window's result does not depend on what else is solved beside it).  At most one submission per table is in flight.  NULL: the driver
     * calls `lba` where it would have waited. */
    int (*lba_submit)(void* ctx, int n, const oslam_lba_problem_t* probs);
    int (*lba_wait)(void* ctx);
    /* optional: the MapPoint updates after a local BA from the solved windows themselves (oslam_job_mp_window_t).  NULL: the driver packs every point's
     * observations into an oslam_job_mp_update_t as for any other update. */
    int (*mp_update_windows)(void* ctx, int n, oslam_job_mp_window_t* wins);
    /* optional pair, with register_keyframes (round 5): a DEVICE MIRROR of the observation graph and its first consumer.
     * map_journal: what changed in the maps of n sequences since the driver's last call for them (oslam_map_changes_t).  The table applies the changes to device
     * copies kept beside the resident keyframe / map-point records: per keyframe KeyFrame::mvpMapPoints, "which point holds the observation (kf, idx)" and the
     * usable-depth bits; per point Observations(), isBad() and the octave histogram of its observations.
     * kf_culling_counts: LocalMapping::KeyFrameCulling's counting loop (src/LocalMapping.cc:649-690) for the candidate keyframes of every job from those copies:
     * out[4 q] = slots of keyframe q with a point at a usable depth, [4 q + 1] = nMPs, [4 q + 2] = nRedundantObservations, [4 q + 3] != 0: a point with exactly
     * three observations at a fine enough scale whose own observation the mirror cannot vouch for — the driver recounts that keyframe from its lists.  The driver
     * takes the verdicts in the reference's order and recounts on the host from the first culled keyframe of a pass on (SetBadFlag changes the counts of the
     * candidates behind it).  NULL: the driver counts on the host. */
    int (*map_journal)(void* ctx, int n, const oslam_map_changes_t* changes);
    int (*kf_culling_counts)(void* ctx, int n, const oslam_job_cull_t* jobs, float thDepth);
    /* optional, with kf_culling_counts: when set, kf_culling_counts may return before the counts are in the jobs' `out` arrays (which must stay valid); this call
     * returns when they are.  The driver asks for the counts right after the local-BA write-back — the last step of a pass that changes observations — and
     * collects them after the MapPoint updates, so the round trip to the device hides behind that stage. */
    int (*kf_culling_collect)(void* ctx);
    /* optional, with map_journal and fuse_points_keyed (round 5): oslam_job_fuse_cur_t above.  The driver flushes its change sets (map_journal) right before. */
    int (*fuse_into_current)(void* ctx, int n, oslam_job_fuse_cur_t* jobs);
    /* optional, with map_journal (round 5): Tracking::UpdateLocalPoints (src/Tracking.cc:1470-1493) from the mirror — mvpLocalMapPoints of a sequence = the point
     * lists of its local keyframes in the given order, every point once, bad points left out.  ids receives at most `cap` point ids; overflow != 0: more than the
     * table's bound or `cap` (the driver then walks the lists itself).  The driver flushes its change sets (map_journal) right before. */
    int (*local_points_list)(void* ctx, int n, oslam_job_local_list_t* jobs);
    /* optional pair (round 5): mp_update_keyed whose results may arrive later.  mp_update_keyed_async enqueues the job and may return before best_idx / out_desc /
     * out5 are written (the job and every array it names must stay valid and untouched); mp_update_collect returns when they are.  At most one job is in flight per
     * table; operators called in between see the job's effects on the resident records (same stream order).  The driver uses it for the descriptor updates that
     * follow every ORBmatcher::Fuse round of SearchInNeighbors (MapPoint::Replace -> ComputeDistinctiveDescriptors, src/MapPoint.cc:314): the next round's host
     * bookkeeping and search are issued behind the update instead of waiting for it. */
    /* optional, with register_keyframes (round 5): mvKeys on demand.  A table that offers it may leave oslam_slam_frame_t::keys of every frame untouched; the driver
     * then asks for the raw keypoints of the frames that BECAME keyframes — right after register_keyframes, for the same slots: out[q] receives counts[q] keypoints —
     * because only KeyFrame::UnprojectStereo reads them (src/KeyFrame.cc:620-621, called by CreateNewMapPoints).  One frame in ~15 becomes a keyframe: 28 of the 96
     * bytes per keypoint a frame sends back to the host stay on the device. */
    int (*keyframe_raw_keys)(void* ctx, int n, const int32_t* slots, const int32_t* counts, oslam_keypoint_t* const* out);
    /* optional pair, with keyframe_raw_keys (round 5): mDescriptors on demand.  A table that offers both may leave oslam_slam_frame_t::desc of every frame untouched.
     * keyframe_descriptors: like keyframe_raw_keys, for the descriptors of the frames that became keyframes (the host copy KeyFrame::mDescriptors).
     * frame_descriptors: the descriptors of the CURRENT frames of `slots` (valid until the next frames_* call) — the driver asks for them when a frame takes the
     * TrackReferenceKeyFrame path, whose Frame::ComputeBoW runs on the host (src/Tracking.cc:841); every other reader of a frame's descriptors is an operator. */
    int (*keyframe_descriptors)(void* ctx, int n, const int32_t* slots, const int32_t* counts, uint8_t* const* out);
    int (*frame_descriptors)(void* ctx, int n, const int32_t* slots, const int32_t* counts, uint8_t* const* out);
    int (*mp_update_keyed_async)(void* ctx, oslam_job_mp_update_t* job, const int32_t* obs_key);
    int (*mp_update_collect)(void* ctx);
} oslam_slam_ops_t;

/* System::System for S sequences of one camera model (src/System.cc:33-120, minus vocabulary / viewer / loop closer). */
int oslam_slam_create(oslam_slam_t** out, const oslam_slam_config_t* cfg);
/* Same driver over a caller-supplied operator table (ownership of ops->ctx passes to the handle). Test seam. */
int oslam_slam_create_with_ops(oslam_slam_t** out, const oslam_slam_config_t* cfg, const oslam_slam_ops_t* ops);
void oslam_slam_destroy(oslam_slam_t* h);

/* System::TrackRGBD (include/System.h:75) for every sequence: gray[s] / depth[s] = next frame of sequence s (depth in metres,
 * i.e. after the DepthMapFactor scaling of src/Tracking.cc:262).  Tcw_out [S][16] = mCurrentFrame.mTcw (zeros while the sequence has no
 * pose), state_out [S] = mState after the frame. */
int oslam_slam_track_rgbd(oslam_slam_t* h, const uint8_t* const* gray, int gray_stride, const float* const* depth, int depth_pitch,
                          int on_device, const double* timestamps, float* Tcw_out, int32_t* state_out);

/* System::TrackStereo (include/System.h:69) for every sequence (cfg.sensor = 1): rectified left / right images. */
int oslam_slam_track_stereo(oslam_slam_t* h, const uint8_t* const* left, const uint8_t* const* right, int gray_stride, int on_device,
                            const double* timestamps, float* Tcw_out, int32_t* state_out);

/* The same with the frame's semantic detections (objs [S]; objs[s].n = 0 or objs == NULL: no detections); on_device applies to the masks too. */
int oslam_slam_track_rgbd_objects(oslam_slam_t* h, const uint8_t* const* gray, int gray_stride, const float* const* depth, int depth_pitch, int on_device,
                                  const double* timestamps, const oslam_slam_objects_t* objs, int mask_stride, float* Tcw_out, int32_t* state_out);
int oslam_slam_track_stereo_objects(oslam_slam_t* h, const uint8_t* const* left, const uint8_t* const* right, int gray_stride, int on_device,
                                    const double* timestamps, const oslam_slam_objects_t* objs, int mask_stride, float* Tcw_out, int32_t* state_out);
/* System::TrackRGBD on the RAW inputs (include/System.h:75, src/Tracking.cc:241-275): gray as above, depth16[s] = the 16-bit depth image as read from the
 * dataset (rows depth_pitch elements apart) with depth_factor = mDepthMapFactor = 1 / DepthMapFactor (1.0f / 5000 for TUM); objs may be NULL.  mask_stride = 0
 * says that objs[s].masks are ONE-BIT-PER-PIXEL images in the layout of oslam_mask_bits_device ([height][ceil(width / 64)] uint64, bit i of word w = pixel
 * 64 w + i == 255) packed by the caller.  With on_device != 0 every pointer only has to be DEVICE-ACCESSIBLE: pinned host memory (hipHostMalloc) qualifies —
 * the Frame::Frame / object kernels then read the images over PCIe where they are (8-bit gray once, the depth values at the keypoints, the mask words),
 * no staging copy.  Needs a table with frames_rgbd_raw16. */
int oslam_slam_track_rgbd_raw16(oslam_slam_t* h, const uint8_t* const* gray, int gray_stride, const uint16_t* const* depth16, int depth_pitch, float depth_factor,
                                int on_device, const double* timestamps, const oslam_slam_objects_t* objs, int mask_stride, float* Tcw_out, int32_t* state_out);

/* Object layer counters of one sequence: [0] N_AllSemanticConstraintNum (src/ObjectOptimizer.cc:1233), [1] frames optimised with matched objects,
 * [2] frames whose nSemNum was > 0, [3] Object3Ds, [4] map points listed in Object3Ds, [5] Object2Ds built, [6] local-BA windows degraded for capacity (see
 * oslam_slam_lba_window_stats [5]). */
int oslam_slam_object_stats(oslam_slam_t* h, int seq, int64_t out[8]);

/* Sizes of the local-BA windows of one sequence since creation (Optimizer::LocalBundleAdjustment's graph gather, src/Optimizer.cc:456-504):
 * [0] windows, then sums over them: [1] local keyframes, [2] fixed keyframes, [3] map points, [4] edges; [5] windows DEGRADED because they had more than 128 free
 * keyframes (the local-BA operator's bound; the reference has none): their weakest covisible keyframes were held fixed. */
int oslam_slam_lba_window_stats(oslam_slam_t* h, int seq, int64_t out[8]);
/* Test hook of the per-sequence failure isolation: the next local-BA window of sequence `seq` is handed to the operator with an edge that names a keyframe outside
 * the window, which an operator table that validates its windows (the HIP table does) refuses with OSLAM_E_INVALID for that window alone.  out[7] of
 * oslam_slam_lba_window_stats counts the operator failures of a sequence. */
int oslam_slam_inject_failure(oslam_slam_t* h, int seq);

/* Deferred schedule: waits for the local BA in flight and applies its write-back, the MapPoint updates and KeyFrameCulling (what System::Shutdown's wait for the
 * local mapper does, src/System.cc:303-320).  No effect when nothing is pending.  Called by oslam_slam_trajectory / oslam_slam_keyframe_trajectory. */
int oslam_slam_finish(oslam_slam_t* h);

/* System::SaveTrajectoryTUM (src/System.cc:378-440): per tracked frame the pose re-anchored on its reference keyframe's final pose.
 * Twc [n][12] = rows of [Rwc | twc]; lost frames are skipped like the reference.  Returns the count in *n_out (cap < n -> OSLAM_E_CAPACITY). */
int oslam_slam_trajectory(oslam_slam_t* h, int seq, int cap, double* stamps, float* Twc, int32_t* n_out);
/* System::SaveKeyFrameTrajectoryTUM (:443-477): non-bad keyframes by id. */
int oslam_slam_keyframe_trajectory(oslam_slam_t* h, int seq, int cap, double* stamps, float* Twc, int32_t* n_out);

/* Counters of one sequence: [0] frames, [1] keyframes created, [2] keyframes in map, [3] map points created, [4] map points in map,
 * [5] local BAs, [6] frames tracked by the motion model, [7] by the reference keyframe, [8] lost frames, [9] points fused,
 * [10] points triangulated, [11] keyframes culled, [12] map points culled, [13] last mnMatchesInliers, [14] LBA edges total, [15] map-consistency violations (0). */
int oslam_slam_stats(oslam_slam_t* h, int seq, int64_t out[16]);
/* Wall-clock seconds spent per stage since creation: [0] frames, [1] search_last, [2] pose_opt, [3] search_local, [4] host tracking,
 * [5] mp_update, [6] lba, [7] host mapping, [8] fuse/bow/triangulate; sub-splits of [7]: [9] ProcessNewKeyFrame / UpdateConnections / MapPointCulling,
 * [10] CreateNewMapPoints, [11] SearchInNeighbors, [12] LocalBundleAdjustment gather + write-back, [13] KeyFrameCulling; of [4]: [14] motion model /
 * reference keyframe / after-tracking bookkeeping, [15] UpdateLocalMap + SearchLocalPoints + pose job. */
int oslam_slam_stage_seconds(oslam_slam_t* h, double out[16]);
/* Core-seconds of the same stages: CPU time of the stepping thread plus the time the shared workers spent on its parallel sections (for the device
 * stages [0]-[3], [5], [6], [8] this is the host side of the operator: packing, launches, polling). */
int oslam_slam_stage_cpu_seconds(oslam_slam_t* h, double out[16]);
/* sizeof of the structs that cross this boundary, as THIS library was compiled: [0] oslam_slam_config_t, [1] oslam_slam_ops_t, [2] oslam_slam_objects_t,
 * [3] oslam_map_changes_t.  A binding in another language mirrors these structs by hand (object_slam_amd/slam.py does): it compares its own sizes with these before
 * the first call — a mirror of oslam_slam_ops_t that misses the members added later is a buffer overflow on both sides of the table. */
int oslam_slam_struct_sizes(int32_t out[4]);
/* out[0] = tracked frames whose local map (mvpLocalMapPoints and the packed SearchLocalPoints arrays) was reused from the previous frame because neither
 * the ordered local keyframe list nor the sequence's map had changed, out[1] = all tracked frames; summed over the sequences of the handle. */
int oslam_slam_local_map_reuse(oslam_slam_t* h, int64_t out[2]);
/* Observations in culled keyframes that ComputeDistinctiveDescriptors left out (src/MapPoint.cc:366: `if(!pKF->isBad())`), summed over the handle's
 * sequences and calls: such observations exist when two new points triangulate against the same neighbour keypoint (the second AddMapPoint wins,
 * src/LocalMapping.cc:440-446) and the neighbour is culled later (its SetBadFlag only erases the observations of its own mvpMapPoints). */
int oslam_slam_bad_keyframe_observations(oslam_slam_t* h, int64_t* out);
/* Test hook: host[64] = position, normal, minimum / maximum distance and descriptor of map point `id` of sequence `seq` as the driver holds them, resident[64] =
 * the operator table's record of the same point (OSLAM_E_INVALID if the table keeps none); *bad = the point's mbBad.  The two must be equal for every
 * point that is not bad and has observations. */
int oslam_slam_debug_point(oslam_slam_t* h, int seq, int id, uint8_t host[64], uint8_t resident[64], int32_t* bad);
/* Device time of the kernel groups of the HIP operator table, measured with HIP events on the stream each group is launched on
 * (bench.py's roofline).  Returns what accumulated since the last call, then sets the switch to `enable`.  Per group g:
 * out[3g] = milliseconds, out[3g+1] = kernel launches, out[3g+2] = algorithmic work of those launches — bytes for group 0 (SURVEY.md
 * §8(d) extraction model), fp64 flop for groups 1 and 2 (§8(d): 700 flop per edge and linearisation, 90 per edge and trial evaluation,
 * Schur 324 k_p^2 per point, Cholesky (6K)^3/3, back-substitution 2(6K)^2 + 45P per trial), observations for group 6, 0 elsewhere.
 * OSLAM_E_INVALID on a table without device timing (the test seam). */
int oslam_slam_kernel_times(oslam_slam_t* h, int enable, double out[OSLAM_SLAM_KT_GROUPS * 3]);

#ifdef __cplusplus
}
#endif
#endif /* OSLAM_SLAM_H */
