// slam_ops_hip.hip — the HIP operator table of the batch-of-sequences driver (include/oslam_slam.h): every stage of the
// lockstep step becomes batch launches of the kernels behind oslam_hip.h.  Host glue only (the kernels live in the other
// translation units).  No CPU fallback: creation fails without a HIP device.  Never includes oracle/.
#include <cmath>
#include <vector>

#include "../../include/oslam_slam.h"
#include "common.h"

namespace {

struct HipOps {
    oslam_slam_config_t cfg;
    int S = 0, cap = 0;
    oslam_orb_t* orb = nullptr;
    oslam_matcher_t* m_last = nullptr;
    oslam_matcher_t* m_map = nullptr;
    oslam_poseopt_t* po = nullptr;
    oslam_lba_t* ba = nullptr;
    oslam_lba_t* ba1 = nullptr;
    oslam_mappoint_t* mp = nullptr;
    oslam_frame_t* fr = nullptr;
    oslam_bow_t* bow = nullptr;
    float scale[OSLAM_MAX_LEVELS], invScale[OSLAM_MAX_LEVELS], sigma2[OSLAM_MAX_LEVELS], invSigma2[OSLAM_MAX_LEVELS];
    float bounds[4], K4[4], K5[5];
    oslam_camera_t cam;
    float logScale;
    int max_local = 0;
    // device-resident batch state of the current step
    uint8_t* d_gray = nullptr; float* d_depth = nullptr; size_t gray_pitch = 0;
    oslam_keypoint_t* d_keysUn = nullptr; float* d_uRight = nullptr; float* d_mvDepth = nullptr; int32_t* d_status = nullptr;
    std::vector<oslam_proj_query_t> q;
    std::vector<int32_t> qm, qd;
};

#define OPS_CHECK(x) do { const int rc_ = (x); if (rc_) return rc_; } while (0)

int h_max_keypoints(void* p) { return ((HipOps*)p)->cap; }
int h_scale_tables(void* p, float* a, float* b, float* c, float* d) {
    HipOps* o = (HipOps*)p;
    const int n = o->cfg.nLevels;
    memcpy(a, o->scale, 4 * n); memcpy(b, o->invScale, 4 * n); memcpy(c, o->sigma2, 4 * n); memcpy(d, o->invSigma2, 4 * n);
    return OSLAM_OK;
}
int h_image_bounds(void* p, float* b) { memcpy(b, ((HipOps*)p)->bounds, 16); return OSLAM_OK; }

// Frame::Frame (src/Frame.cc:117-172) for n frames: one batched extraction, one undistort launch, one depth lookup launch; the
// keypoints / descriptors come back in one pass of copies after a single synchronisation.
int h_frames(void* p, int n, const int32_t* slots, const uint8_t* const* gray, int gray_stride, const float* const* depth, int depth_pitch, int on_device,
             oslam_slam_frame_t* const* out) {
    HipOps* o = (HipOps*)p;
    if (n > o->S) { oslam::set_error("frames_rgbd: n > n_sequences"); return OSLAM_E_INVALID; }
    const int W = o->cfg.width, H = o->cfg.height;
    const size_t gimg = o->gray_pitch * H, dimg = (size_t)W * H;
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    for (int i = 0; i < n; i++) {
        OSLAM_HIP_CHECK(hipMemcpy2DAsync(o->d_gray + gimg * i, o->gray_pitch, gray[i], gray_stride, W, H, kind, nullptr));
        OSLAM_HIP_CHECK(hipMemcpy2DAsync(o->d_depth + dimg * i, (size_t)W * 4, depth[i], (size_t)depth_pitch * 4, (size_t)W * 4, H, kind, nullptr));
    }
    OPS_CHECK(oslam_orb_extract_batch_device(o->orb, o->d_gray, n, (int)o->gray_pitch, gimg, nullptr));
    const oslam_keypoint_t* d_kp; const uint8_t* d_desc; const int32_t* d_cnt; const int32_t* d_st;
    OPS_CHECK(oslam_orb_results_device(o->orb, &d_kp, &d_desc, &d_cnt, &d_st));
    OPS_CHECK(oslam_frame_undistort_batch_device(d_kp, o->d_keysUn, d_cnt, 0, o->cap, n, o->K4, o->cfg.dist, o->cfg.ndist, nullptr));
    OPS_CHECK(oslam_frame_stereo_from_rgbd_batch_device(d_kp, o->d_keysUn, d_cnt, 0, o->cap, n, o->d_depth, H, W, W, dimg, o->cfg.bf, o->d_uRight,
                                                        o->d_mvDepth, o->d_status, nullptr));
    std::vector<int32_t> cnt(n);
    int32_t st[2] = {0, 0};
    OSLAM_HIP_CHECK(hipMemcpyAsync(cnt.data(), d_cnt, 4 * n, hipMemcpyDeviceToHost, nullptr));
    OSLAM_HIP_CHECK(hipMemcpyAsync(&st[0], d_st, 4, hipMemcpyDeviceToHost, nullptr));
    OSLAM_HIP_CHECK(hipMemcpyAsync(&st[1], o->d_status, 4, hipMemcpyDeviceToHost, nullptr));
    OSLAM_HIP_CHECK(hipStreamSynchronize(nullptr));
    if (st[0]) { oslam::set_error("extractor arena overflow"); return OSLAM_E_CAPACITY; }
    if (st[1]) { oslam::set_error("keypoint outside the depth image"); return OSLAM_E_INVALID; }
    for (int i = 0; i < n; i++) {
        oslam_slam_frame_t* f = out[i];
        const int N = cnt[i];
        f->N = N;
        const size_t at = (size_t)i * o->cap;
        OSLAM_HIP_CHECK(hipMemcpyAsync(f->keys, d_kp + at, sizeof(oslam_keypoint_t) * N, hipMemcpyDeviceToHost, nullptr));
        OSLAM_HIP_CHECK(hipMemcpyAsync(f->keysUn, o->d_keysUn + at, sizeof(oslam_keypoint_t) * N, hipMemcpyDeviceToHost, nullptr));
        OSLAM_HIP_CHECK(hipMemcpyAsync(f->desc, d_desc + at * 32, (size_t)N * 32, hipMemcpyDeviceToHost, nullptr));
        OSLAM_HIP_CHECK(hipMemcpyAsync(f->uRight, o->d_uRight + at, (size_t)N * 4, hipMemcpyDeviceToHost, nullptr));
        OSLAM_HIP_CHECK(hipMemcpyAsync(f->depth, o->d_mvDepth + at, (size_t)N * 4, hipMemcpyDeviceToHost, nullptr));
    }
    OSLAM_HIP_CHECK(hipStreamSynchronize(nullptr));
    (void)slots;
    return OSLAM_OK;
}

int h_search_last(void* p, int n, oslam_job_search_last_t* jobs) {
    HipOps* o = (HipOps*)p;
    for (int i = 0; i < n; i++) {
        oslam_job_search_last_t& j = jobs[i];
        o->qm.resize(j.Nlast + 1); o->qd.resize(j.Nlast + 1);
        OPS_CHECK(oslam_match_project_last_frame(o->m_last, j.cur->N, j.cur->keysUn, j.cur->uRight, j.cur->desc, nullptr, o->bounds, j.Nlast, j.Xw, j.has_mp,
                                                 j.last_keysUn, j.mp_desc, j.Tcw, j.Tlw, &o->cam, o->scale, o->cfg.nLevels, j.th, 0, 1, o->qm.data(),
                                                 o->qd.data(), j.kp_match, &j.nmatches));
    }
    return OSLAM_OK;
}

int h_search_local(void* p, int n, oslam_job_search_local_t* jobs) {
    HipOps* o = (HipOps*)p;
    for (int i = 0; i < n; i++) {
        oslam_job_search_local_t& j = jobs[i];
        for (int k = 0; k < j.cur->N; k++) j.kp_match[k] = -1;
        j.nmatches = 0;
        if (j.M == 0) continue;
        if (j.M > o->max_local) { oslam::set_error("search_local: %d local points > capacity %d", j.M, o->max_local); return OSLAM_E_CAPACITY; }
        o->q.resize(j.M); o->qm.resize(j.M); o->qd.resize(j.M);
        OPS_CHECK(oslam_frame_is_in_frustum(o->mp, j.M, j.Pw, j.Pn, j.maxDist, j.minDist, j.obs_gt0, j.mp_desc, j.Tcw, o->K5, o->bounds, 0.5f, o->logScale,
                                            o->scale, o->cfg.nLevels, j.th, o->q.data()));
        int nin = 0;
        for (int e = 0; e < j.M; e++) { j.in_view[e] = o->q[e].flags & 1; nin += j.in_view[e]; }
        if (nin == 0) continue;
        OPS_CHECK(oslam_match_search_by_projection(o->m_map, j.cur->N, j.cur->keysUn, j.cur->uRight, j.cur->desc, j.blocked, o->bounds, o->q.data(), j.M,
                                                   0.8f, 1, 0, o->qm.data(), o->qd.data(), j.kp_match, &j.nmatches));
    }
    return OSLAM_OK;
}

int h_pose_opt(void* p, int n, oslam_job_pose_t* jobs) {
    HipOps* o = (HipOps*)p;
    for (int i = 0; i < n; i++) {
        oslam_job_pose_t& j = jobs[i];
        OPS_CHECK(oslam_pose_optimize(o->po, j.N, j.Tcw_in, j.Xw, j.obs, j.invSigma2, j.has_mp, o->K5, j.Tcw_out, j.outlier, &j.n_inliers, nullptr));
    }
    return OSLAM_OK;
}

int h_mp_update(void* p, oslam_job_mp_update_t* j) {
    HipOps* o = (HipOps*)p;
    if (j->do_desc) OPS_CHECK(oslam_mp_distinctive_descriptors(o->mp, j->P, j->obs_start, j->obs_desc, j->best_idx, j->out_desc));
    if (j->do_normal)
        OPS_CHECK(oslam_mp_update_normal_depth(o->mp, j->P, j->Pos, j->obs_start, j->obs_Ow, j->OwRef, j->levelScaleFactor, o->scale[o->cfg.nLevels - 1], j->out5));
    return OSLAM_OK;
}

int h_lba(void* p, int n, const oslam_lba_problem_t* pr) {
    HipOps* o = (HipOps*)p;
    if (n == 1)   // one window: spread over the whole GPU
        return oslam_lba_optimize(o->ba1, pr[0].nKF, pr[0].poses, pr[0].fixed, pr[0].nP, pr[0].points, pr[0].nE, pr[0].edge_kf, pr[0].edge_pt, pr[0].edge_obs,
                                  pr[0].edge_invSigma2, o->K5, 0, pr[0].poses_out, pr[0].points_out, pr[0].erase, nullptr);
    return oslam_lba_optimize_batch(o->ba, n, pr, o->K5);   // one workgroup per window, one launch
}

int h_fuse(void* p, int n, oslam_job_fuse_t* jobs) {
    HipOps* o = (HipOps*)p;
    for (int i = 0; i < n; i++) {
        oslam_job_fuse_t& j = jobs[i];
        if (j.M > o->max_local) { oslam::set_error("fuse: %d queries > capacity %d", j.M, o->max_local); return OSLAM_E_CAPACITY; }
        o->qd.resize(j.M + 1);
        int32_t nf = 0;
        OPS_CHECK(oslam_match_fuse_search(o->m_map, j.N, j.keysUn, j.uRight, j.desc, o->bounds, j.queries, j.M, o->invSigma2, o->cfg.nLevels, j.q_match,
                                          o->qd.data(), &nf));
    }
    return OSLAM_OK;
}

int h_bow(void* p, int n, oslam_job_bow_t* jobs) {
    HipOps* o = (HipOps*)p;
    for (int i = 0; i < n; i++) {
        oslam_job_bow_t& j = jobs[i];
        if (!j.triangulation) OPS_CHECK(oslam_match_search_by_bow(o->bow, &j.s1, &j.s2, j.nnratio, j.checkOri, j.match, &j.nmatches));
        else
            OPS_CHECK(oslam_match_search_for_triangulation(o->bow, &j.s1, &j.s2, j.F12, j.ex, j.ey, o->scale, o->sigma2, o->cfg.nLevels, 0, j.checkOri, j.match,
                                                           &j.nmatches));
    }
    return OSLAM_OK;
}

int h_triangulate(void* p, int n, oslam_job_triangulate_t* jobs) {
    HipOps* o = (HipOps*)p;
    for (int i = 0; i < n; i++) {
        oslam_job_triangulate_t& j = jobs[i];
        if (j.M == 0) continue;
        const int32_t ps[2] = {0, j.M};
        int32_t nnew = 0;
        OPS_CHECK(oslam_mp_triangulate(o->mp, &j.kf1, 1, &j.kf2, ps, j.idx1, j.idx2, o->scale, o->sigma2, o->cfg.nLevels, 1.5f * o->cfg.scaleFactor, j.ok, j.x3D,
                                       &nnew));
    }
    return OSLAM_OK;
}

void h_destroy(void* p) {
    HipOps* o = (HipOps*)p;
    oslam_orb_destroy(o->orb); oslam_matcher_destroy(o->m_last); oslam_matcher_destroy(o->m_map); oslam_poseopt_destroy(o->po);
    oslam_lba_destroy(o->ba); oslam_lba_destroy(o->ba1); oslam_mappoint_destroy(o->mp); oslam_frame_destroy(o->fr); oslam_bow_destroy(o->bow);
    (void)hipFree(o->d_gray); (void)hipFree(o->d_depth); (void)hipFree(o->d_keysUn); (void)hipFree(o->d_uRight); (void)hipFree(o->d_mvDepth); (void)hipFree(o->d_status);
    delete o;
}

}  // namespace

int oslam_slam_make_hip_ops(const oslam_slam_config_t* cfg, oslam_slam_ops_t* ops) {
    if (oslam_device_count() <= 0 || hipSetDevice(cfg->device) != hipSuccess) {
        oslam::set_error("oslam_slam_create: no HIP device (the tracking driver has no CPU fallback)");
        return OSLAM_E_HIP;
    }
    HipOps* o = new HipOps;
    o->cfg = *cfg; o->S = cfg->n_sequences;
    const int dev = cfg->device;
    int rc = oslam_orb_create(&o->orb, cfg->nFeatures, cfg->scaleFactor, cfg->nLevels, cfg->iniThFAST, cfg->minThFAST, cfg->width, cfg->height, o->S, dev);
    if (!rc) {
        o->cap = oslam_orb_max_keypoints(o->orb);
        o->max_local = 16384;
        rc = oslam_matcher_create(&o->m_last, o->S, o->cap, o->cap, dev);
    }
    if (!rc) rc = oslam_matcher_create(&o->m_map, o->S, o->cap, o->max_local, dev);
    if (!rc) rc = oslam_poseopt_create(&o->po, o->S, o->cap, dev);
    if (!rc) rc = oslam_lba_create(&o->ba, o->S, 128, 32768, 262144, dev);
    if (!rc) rc = oslam_lba_create(&o->ba1, 1, 128, 32768, 262144, dev);
    if (!rc) rc = oslam_lba_set_mode(o->ba, 0);
    if (!rc) rc = oslam_mappoint_create(&o->mp, dev);
    if (!rc) rc = oslam_frame_create(&o->fr, dev);
    if (!rc) rc = oslam_bow_create(&o->bow, o->cap, dev);
    if (!rc) rc = oslam_orb_get_scale_tables(o->orb, o->scale, o->invScale, o->sigma2, o->invSigma2, nullptr);
    o->K4[0] = cfg->fx; o->K4[1] = cfg->fy; o->K4[2] = cfg->cx; o->K4[3] = cfg->cy;
    memcpy(o->K5, o->K4, 16); o->K5[4] = cfg->bf;
    o->cam.fx = cfg->fx; o->cam.fy = cfg->fy; o->cam.cx = cfg->cx; o->cam.cy = cfg->cy; o->cam.bf = cfg->bf; o->cam.b = cfg->bf / cfg->fx;
    o->logScale = std::log(cfg->scaleFactor);
    if (!rc) rc = oslam_frame_image_bounds(o->fr, cfg->width, cfg->height, o->K4, cfg->dist, cfg->ndist, o->bounds);
    if (!rc) {
        o->gray_pitch = oslam::align_up((size_t)cfg->width, 64);
        const size_t S = o->S;
        hipError_t e = hipMalloc((void**)&o->d_gray, o->gray_pitch * cfg->height * S);
        if (e == hipSuccess) e = hipMalloc((void**)&o->d_depth, (size_t)cfg->width * cfg->height * 4 * S);
        if (e == hipSuccess) e = hipMalloc((void**)&o->d_keysUn, sizeof(oslam_keypoint_t) * o->cap * S);
        if (e == hipSuccess) e = hipMalloc((void**)&o->d_uRight, 4 * (size_t)o->cap * S);
        if (e == hipSuccess) e = hipMalloc((void**)&o->d_mvDepth, 4 * (size_t)o->cap * S);
        if (e == hipSuccess) e = hipMalloc((void**)&o->d_status, 64);
        if (e == hipSuccess) e = hipMemset(o->d_status, 0, 64);
        if (e != hipSuccess) { oslam::set_error("slam ops: hipMalloc failed: %s", hipGetErrorString(e)); rc = OSLAM_E_HIP; }
    }
    if (rc) { h_destroy(o); return rc; }
    ops->ctx = o;
    ops->max_keypoints = h_max_keypoints; ops->scale_tables = h_scale_tables; ops->image_bounds = h_image_bounds; ops->frames_rgbd = h_frames;
    ops->search_last = h_search_last; ops->search_local = h_search_local; ops->pose_opt = h_pose_opt; ops->mp_update = h_mp_update; ops->lba = h_lba;
    ops->fuse = h_fuse; ops->bow = h_bow; ops->triangulate = h_triangulate; ops->destroy = h_destroy;
    return OSLAM_OK;
}
