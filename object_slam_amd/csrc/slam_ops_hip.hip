// slam_ops_hip.hip — the HIP operator table of the batch-of-sequences driver (include/oslam_slam.h): every stage of the
// lockstep step becomes batch launches of the kernels behind oslam_hip.h.  Host glue only (the kernels live in the other
// translation units).  No CPU fallback: creation fails without a HIP device.  Never includes oracle/.
#include <algorithm>
#include <cmath>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <functional>
#include <vector>

#include "../../include/oslam_slam.h"
#include "common.h"
#include "slam_pool.h"

namespace {

struct LbaService;
struct LbaJob {   // one handle's submission to the local-BA service (below)
    int n = 0; const oslam_lba_problem_t* probs = nullptr; float K5[5] = {0, 0, 0, 0, 0};
    bool timing = false, done = false; int rc = 0; char err[256] = "";
    double ms = 0, launches = 0, flop = 0;
};

struct HipOps {
    LbaService* svc = nullptr;            // deferred schedule: the process-wide local-BA service of this device
    LbaJob job; bool job_active = false;
    oslam_slam_config_t cfg;
    int S = 0, cap = 0;
    oslam_orb_t* orb = nullptr;
    oslam_orb_t* orbR = nullptr;          // STEREO: right image extractor (src/Tracking.cc:145)
    oslam_stereo_t* stereo = nullptr;
    uint8_t* d_grayR = nullptr;
    const float* cur_uRight = nullptr;    // mvuRight of the current batch on the device (RGB-D: d_uRight, STEREO: the stereo matcher's output)
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
    oslam_keypoint_t* d_keysUn_prev = nullptr;   // mvKeysUn of the PREVIOUS step's frames (the buffers swap at every Frame::Frame stage): the last frames of search_last
    oslam_keypoint_t* d_keysUn = nullptr; float* d_uRight = nullptr; float* d_mvDepth = nullptr; int32_t* d_status = nullptr;
    const oslam_keypoint_t* d_kp = nullptr; const uint8_t* d_desc = nullptr; const int32_t* d_cnt = nullptr;
    std::vector<oslam_proj_query_t> q;
    std::vector<int32_t> qm, qd;
    // staging: one pinned upload block mirrored on the device, one pinned download block; a stage of the lockstep step is
    // fill (parallel memcpy) -> ONE host-to-device copy -> batch kernels -> result copies -> ONE synchronisation -> scatter
    uint8_t* up_h = nullptr; uint8_t* up_d = nullptr; size_t up_cap = 0;
    uint8_t* dn_h = nullptr; size_t dn_cap = 0;
    // second staging set + event pair for the ONE deferred MapPoint update (mp_update_keyed_async): its job block is read in place by the kernel and its results
    // land in the pinned block while the next operator already uses the first set
    uint8_t* upB_h = nullptr; uint8_t* upB_d = nullptr; size_t upB_cap = 0;
    uint8_t* dnB_h = nullptr; size_t dnB_cap = 0;
    hipEvent_t tevB0 = nullptr, tevB1 = nullptr;
    bool lazy_desc = false;                          // the same for mDescriptors (keyframe_descriptors / frame_descriptors)
    bool lazy_keys = false;                          // frames do not send mvKeys back; register_keyframes stages the new keyframes' rows for keyframe_raw_keys
    uint8_t* kfk_h = nullptr; size_t kfk_cap = 0; std::vector<int32_t> kfk_slots;
    static constexpr int kFuseCurStride = 16384, kFuseCurPairs = 2048;   // candidates / matches per job of fuse_into_current (beyond: overflow, the driver's own path)
    uint8_t* fc_d = nullptr; size_t fc_cap = 0; uint32_t fc_stamp = 0;
    struct MpuPending { bool on = false; oslam_job_mp_update_t* j = nullptr; size_t P = 0, rBest = 0, rOut = 0, rOut5 = 0; double dtotal = 0; std::function<int()> launch; } mpu_pend;
    int mpu_launch_pending() { if (mpu_pend.on && mpu_pend.launch) { std::function<int()> f; f.swap(mpu_pend.launch); return f(); } return OSLAM_OK; }
    void swap_staging() { std::swap(up_h, upB_h); std::swap(up_d, upB_d); std::swap(up_cap, upB_cap); std::swap(dn_h, dnB_h); std::swap(dn_cap, dnB_cap); std::swap(tev0, tevB0); std::swap(tev1, tevB1); }
    oslam_proj_query_t* d_lq = nullptr; uint8_t* d_inview = nullptr; size_t lq_cap = 0;
    uint8_t* d_objbits = nullptr;                        // [S][cap] keypoint test bits (object_kps)
    // Resident local maps: the packed SearchLocalPoints arrays of every slot stay on the device ([S][loc_st] each); a job whose content id equals the
    // slot's is not uploaded again (the driver repacks only when the sequence's local keyframe list or map changed)
    uint8_t* d_loc = nullptr; size_t loc_st = 0;
    std::vector<long long> loc_id;
    float* loc_Pw() const { return (float*)d_loc; }
    float* loc_Pn() const { return (float*)(d_loc + 12 * loc_st * S); }
    float* loc_Max() const { return (float*)(d_loc + 24 * loc_st * S); }
    float* loc_Min() const { return (float*)(d_loc + 28 * loc_st * S); }
    uint8_t* loc_Obs() const { return d_loc + 32 * loc_st * S; }
    uint8_t* loc_Desc() const { return d_loc + 33 * loc_st * S; }
    int ensure_loc() {   // follows max_local
        const size_t want = oslam::align_up((size_t)std::max(max_local, 1), 64);
        if (d_loc && loc_st == want) return OSLAM_OK;
        OSLAM_HIP_CHECK(hipStreamSynchronize(strm));
        if (d_loc) (void)hipFree(d_loc);
        d_loc = nullptr;
        loc_st = want;
        OSLAM_HIP_CHECK(hipMalloc((void**)&d_loc, 65 * loc_st * S));
        loc_id.assign(S, 0);
        return OSLAM_OK;
    }
    // Resident keyframe store: one record per keyframe = device copies of mvKeysUn | mDescriptors | mvuRight (cap entries each), in chunks of kRecChunk records
    static constexpr int kRecChunk = 256;
    std::vector<uint8_t*> rec_chunks;
    std::vector<std::vector<int>> rec_of_kf;              // [slot][kf id] -> record index or -1
    long long n_released = 0;                             // records returned by release_keyframes (culled keyframes)
    std::vector<int32_t> mpu_rec;                         // scratch of mp_update_impl: (record, keypoint) of every observation
    std::vector<int> free_recs;                           // records of maps that were reset, reused before the store grows
    int n_rec = 0;
    uint8_t** d_rec_desc = nullptr; size_t rec_desc_cap = 0; int rec_desc_n = 0;   // device table: descriptor array of every record (for k_gather_desc)
    std::vector<uint8_t*> h_rec_desc;
    // a record = mvKeysUn, descriptors, mvuRight and — built once at registration (oslam_kf_grid_build_device) — the feature grid: cell ends + candidates sorted by cell
    // (+ the mirror of the observation graph, round 5: the keyframe's point list, "which point observes keypoint i" and the usable-depth bits: oslam_slam_ops_t::map_journal)
    size_t rec_core_bytes() const {
        return oslam::align_up((size_t)cap * sizeof(oslam_keypoint_t), 256) + oslam::align_up((size_t)cap * 32, 256) + oslam::align_up((size_t)cap * 4, 256) +
               oslam::align_up((size_t)3072 * 2, 256) + oslam::align_up((size_t)cap * 16, 256);
    }
    size_t rec_bytes() const { return rec_core_bytes() + oslam::align_up((size_t)cap * 4, 256) + oslam::align_up((size_t)cap * 8, 256) + oslam::align_up(((size_t)cap + 31) / 32 * 4, 256); }
    std::vector<uint32_t> okf_seq;                        // [S] event counter of the mirror's okf cells (a cell keeps the event with the largest number)
    size_t mir_used = 0;                                  // bytes of the mirror's upload block in use by the flush / count request in flight
    uint8_t* cull_h = nullptr; size_t cull_cap = 0;       // pinned result block of kf_culling_counts
    struct CullPending { int32_t* out; int n; };
    std::vector<CullPending> cull_pending;
    uint8_t* mir_h = nullptr; uint8_t* mir_d = nullptr; size_t mir_cap = 0;   // the mirror's own upload block (map_journal returns without waiting: the consumer that follows on the stream does)
    int ensure_mir(size_t bytes) {
        if (bytes <= mir_cap) return OSLAM_OK;
        OSLAM_HIP_CHECK(hipStreamSynchronize(strm));
        if (mir_h) (void)hipHostFree(mir_h);
        if (mir_d) (void)hipFree(mir_d);
        mir_h = nullptr; mir_d = nullptr; mir_cap = 0;
        const size_t want = bytes + bytes / 2 + (1 << 20);
        OSLAM_HIP_CHECK(hipHostMalloc((void**)&mir_h, want, 0));
        OSLAM_HIP_CHECK(hipMalloc((void**)&mir_d, want));
        mir_cap = want;
        return OSLAM_OK;
    }
    int32_t* rec_mp(int r) const { return (int32_t*)(rec_ptr(r) + rec_core_bytes()); }
    // per-point scalars of the mirror: 32-byte records per slot, grown like the map-point table: (Observations(), isBad(), octave histogram) in the first 16 bytes,
    // then the transient marks of SearchInNeighbors' second direction (k_fusecur_*: 64-bit first-occurrence key, stamp of the current keyframe's points)
    static constexpr size_t kPtAuxBytes = 32;
    std::vector<uint8_t*> pt_aux; std::vector<size_t> pt_aux_cap;
    uint8_t** d_pt_aux = nullptr; uint8_t** d_rec_chunk = nullptr; size_t rec_chunk_cap = 0, rec_chunk_n = 0;
    bool pt_aux_dirty = true;
    int ensure_pt_aux(int slot, size_t need) {
        if ((int)pt_aux.size() < S) { pt_aux.resize(S, nullptr); pt_aux_cap.resize(S, 0); }
        if (need <= pt_aux_cap[slot]) return OSLAM_OK;
        const size_t ncap = std::max<size_t>(need * 2, 16384);
        uint8_t* nb = nullptr;
        OSLAM_HIP_CHECK(hipMalloc((void**)&nb, ncap * kPtAuxBytes));
        OSLAM_HIP_CHECK(hipMemsetAsync(nb, 0, ncap * kPtAuxBytes, strm));
        if (pt_aux[slot]) {
            OSLAM_HIP_CHECK(hipMemcpyAsync(nb, pt_aux[slot], pt_aux_cap[slot] * kPtAuxBytes, hipMemcpyDeviceToDevice, strm));
            OSLAM_HIP_CHECK(hipStreamSynchronize(strm));
            (void)hipFree(pt_aux[slot]);
        }
        pt_aux[slot] = nb; pt_aux_cap[slot] = ncap; pt_aux_dirty = true;
        return OSLAM_OK;
    }
    int sync_mirror_tables() {   // device copies of the per-slot aux pointers and of the record chunk pointers
        if ((int)pt_aux.size() < S) { pt_aux.resize(S, nullptr); pt_aux_cap.resize(S, 0); pt_aux_dirty = true; }
        if (!d_pt_aux) { OSLAM_HIP_CHECK(hipMalloc((void**)&d_pt_aux, sizeof(uint8_t*) * S)); pt_aux_dirty = true; }
        if (pt_aux_dirty) { OSLAM_HIP_CHECK(hipMemcpyAsync(d_pt_aux, pt_aux.data(), sizeof(uint8_t*) * S, hipMemcpyHostToDevice, strm)); OSLAM_HIP_CHECK(hipStreamSynchronize(strm)); pt_aux_dirty = false; }
        if (rec_chunks.size() > rec_chunk_cap) {
            OSLAM_HIP_CHECK(hipStreamSynchronize(strm));
            if (d_rec_chunk) (void)hipFree(d_rec_chunk);
            rec_chunk_cap = rec_chunks.size() * 2 + 64; rec_chunk_n = 0;
            OSLAM_HIP_CHECK(hipMalloc((void**)&d_rec_chunk, sizeof(uint8_t*) * rec_chunk_cap));
        }
        if (rec_chunk_n < rec_chunks.size()) {
            OSLAM_HIP_CHECK(hipMemcpyAsync(d_rec_chunk + rec_chunk_n, rec_chunks.data() + rec_chunk_n, sizeof(uint8_t*) * (rec_chunks.size() - rec_chunk_n), hipMemcpyHostToDevice, strm));
            OSLAM_HIP_CHECK(hipStreamSynchronize(strm));
            rec_chunk_n = rec_chunks.size();
        }
        return OSLAM_OK;
    }
    uint8_t* rec_ptr(int r) const { return rec_chunks[r / kRecChunk] + (size_t)(r % kRecChunk) * rec_bytes(); }
    const oslam_keypoint_t* rec_keys(int r) const { return (const oslam_keypoint_t*)rec_ptr(r); }
    const uint8_t* rec_desc(int r) const { return rec_ptr(r) + oslam::align_up((size_t)cap * sizeof(oslam_keypoint_t), 256); }
    const float* rec_ur(int r) const { return (const float*)(rec_desc(r) + oslam::align_up((size_t)cap * 32, 256)); }
    uint16_t* rec_cell_end(int r) const { return (uint16_t*)((const uint8_t*)rec_ur(r) + oslam::align_up((size_t)cap * 4, 256)); }
    float* rec_cand(int r) const { return (float*)((uint8_t*)rec_cell_end(r) + oslam::align_up((size_t)3072 * 2, 256)); }
    int rec_lookup(int slot, int kf) const { return (slot >= 0 && slot < (int)rec_of_kf.size() && kf >= 0 && kf < (int)rec_of_kf[slot].size()) ? rec_of_kf[slot][kf] : -1; }
    // Resident map points: one growing array of 64-byte records per slot (position, normal, distances, descriptor), written by every MapPoint update
    std::vector<uint8_t*> mp_tab; std::vector<size_t> mp_cap;   // [S] device arrays and their capacity in records
    uint8_t** d_mp_tab = nullptr;                               // [S] device copy of the pointers
    bool mp_tab_on = true, mp_tab_dirty = true;
    int ensure_mp_records(int slot, size_t need) {
        if ((int)mp_tab.size() < S) { mp_tab.resize(S, nullptr); mp_cap.resize(S, 0); }
        if (need <= mp_cap[slot]) return OSLAM_OK;
        // 16 K records (1 MB) per sequence to start with, doubling: a growth costs an allocation, a device copy, a stream synchronisation and a free (measured:
        // 0.9 ms each, 22 s of the 64 s the MapPoint-update operator took in a 225-step run of 8 x 1024 sequences when the tables started at 4 K records x 1.5)
        const size_t ncap = std::max<size_t>(need * 2, 16384);
        uint8_t* nb = nullptr;
        OSLAM_HIP_CHECK(hipMalloc((void**)&nb, ncap * 64));
        if (mp_tab[slot]) {
            OSLAM_HIP_CHECK(hipMemcpyAsync(nb, mp_tab[slot], mp_cap[slot] * 64, hipMemcpyDeviceToDevice, strm));
            OSLAM_HIP_CHECK(hipStreamSynchronize(strm));
            (void)hipFree(mp_tab[slot]);
        }
        mp_tab[slot] = nb; mp_cap[slot] = ncap; mp_tab_dirty = true;
        return OSLAM_OK;
    }
    int sync_mp_table() {   // the device pointer table follows the host's
        if ((int)mp_tab.size() < S) { mp_tab.resize(S, nullptr); mp_cap.resize(S, 0); mp_tab_dirty = true; }
        if (!mp_tab_dirty) return OSLAM_OK;
        if (!d_mp_tab) OSLAM_HIP_CHECK(hipMalloc((void**)&d_mp_tab, sizeof(uint8_t*) * (size_t)S));
        OSLAM_HIP_CHECK(hipMemcpyAsync(d_mp_tab, mp_tab.data(), sizeof(uint8_t*) * (size_t)S, hipMemcpyHostToDevice, strm));
        OSLAM_HIP_CHECK(hipStreamSynchronize(strm));   // (mp_tab is pageable host memory: the copy must have read it before it can change again)
        mp_tab_dirty = false;
        return OSLAM_OK;
    }
    uint8_t* d_maskstage = nullptr; size_t mask_cap = 0;  // host masks of a stage, packed H x W
    // One-bit-per-pixel form of the step's masks (oslam_mask_bits_device), built by object_kps and reused by pose_opt2 of the SAME step: keyed by the
    // caller's mask pointer, valid while step_epoch (advanced by every Frame::Frame stage) equals bits_epoch
    uint64_t* d_maskbits = nullptr; size_t maskbits_cap = 0;
    std::unordered_map<const uint8_t*, int> bits_of_ptr;
    long long step_epoch = 0, bits_epoch = -1;
    int ensure_maskbits(size_t words) {
        if (words <= maskbits_cap) return OSLAM_OK;
        OSLAM_HIP_CHECK(hipStreamSynchronize(strm));
        if (d_maskbits) (void)hipFree(d_maskbits);
        d_maskbits = nullptr; maskbits_cap = 0;
        OSLAM_HIP_CHECK(hipMalloc((void**)&d_maskbits, (words + words / 4) * 8));
        maskbits_cap = words + words / 4;
        return OSLAM_OK;
    }
    int ensure_masks(size_t bytes) {
        if (bytes <= mask_cap) return OSLAM_OK;
        OSLAM_HIP_CHECK(hipDeviceSynchronize());
        if (d_maskstage) (void)hipFree(d_maskstage);
        d_maskstage = nullptr; mask_cap = 0;
        const size_t ncap = bytes + bytes / 2 + 4096;
        OSLAM_HIP_CHECK(hipMalloc((void**)&d_maskstage, ncap));
        mask_cap = ncap;
        return OSLAM_OK;
    }
    oslam_drv::Pool* pool = nullptr;
    hipStream_t strm = nullptr;   // this handle's stream (non-blocking): several handles on one GPU, each driven by its own host thread, overlap
    // kernel-time groups (oslam_slam_kernel_times): HIP events on `strm` around the launches of a group, read after the stage's synchronisation
    int timing = 0;
    hipEvent_t tev0 = nullptr, tev1 = nullptr;
    double kt[OSLAM_SLAM_KT_GROUPS * 3] = {0};
    void t_begin() { if (timing) (void)hipEventRecord(tev0, strm); }
    void t_end() { if (timing) (void)hipEventRecord(tev1, strm); }
    void t_collect(int g, double launches, double work) {   // after the stream has been synchronised
        if (!timing) return;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, tev0, tev1) == hipSuccess) { kt[3 * g] += ms; kt[3 * g + 1] += launches; kt[3 * g + 2] += work; }
    }
    int ensure_up(size_t bytes) {
        if (bytes <= up_cap) return OSLAM_OK;
        OSLAM_HIP_CHECK(hipDeviceSynchronize());
        if (up_h) (void)hipHostFree(up_h);
        if (up_d) (void)hipFree(up_d);
        up_h = nullptr; up_d = nullptr; up_cap = 0;
        const size_t ncap = bytes + bytes / 2 + 4096;
        OSLAM_HIP_CHECK(hipHostMalloc((void**)&up_h, ncap, 0));
        OSLAM_HIP_CHECK(hipMalloc((void**)&up_d, ncap));
        up_cap = ncap;
        return OSLAM_OK;
    }
    int ensure_dn(size_t bytes) {
        if (bytes <= dn_cap) return OSLAM_OK;
        OSLAM_HIP_CHECK(hipDeviceSynchronize());
        if (dn_h) (void)hipHostFree(dn_h);
        dn_h = nullptr; dn_cap = 0;
        const size_t ncap = bytes + bytes / 2 + 4096;
        OSLAM_HIP_CHECK(hipHostMalloc((void**)&dn_h, ncap, 0));
        dn_cap = ncap;
        return OSLAM_OK;
    }
    int ensure_lq(size_t entries) {
        if (entries <= lq_cap) return OSLAM_OK;
        OSLAM_HIP_CHECK(hipDeviceSynchronize());
        if (d_lq) (void)hipFree(d_lq);
        if (d_inview) (void)hipFree(d_inview);
        d_lq = nullptr; d_inview = nullptr; lq_cap = 0;
        const size_t ncap = entries + entries / 2 + 1024;
        OSLAM_HIP_CHECK(hipMalloc((void**)&d_lq, ncap * sizeof(oslam_proj_query_t)));
        OSLAM_HIP_CHECK(hipMalloc((void**)&d_inview, ncap));
        lq_cap = ncap;
        return OSLAM_OK;
    }
};

// bump layout of the staging blocks
struct Layout {
    size_t off = 0;
    size_t take(size_t bytes) { const size_t at = off; off += oslam::align_up(bytes ? bytes : 1, 256); return at; }
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

// keypoints / descriptors / stereo coordinates of the n frames just built: one pass of pinned copies, one synchronisation, parallel scatter
static int download_frames(HipOps* o, int n, const oslam_keypoint_t* d_kp, const uint8_t* d_desc, const int32_t* d_cnt, const int32_t* d_st, const float* d_uR,
                           const float* d_dp, oslam_slam_frame_t* const* out) {
    o->d_kp = d_kp; o->d_desc = d_desc; o->d_cnt = d_cnt; o->cur_uRight = d_uR;
    const size_t cap = o->cap;
    Layout L;
    const size_t oCnt = L.take(4 * (size_t)n), oSt = L.take(8), oKeys = L.take(sizeof(oslam_keypoint_t) * cap * n), oKeysUn = L.take(sizeof(oslam_keypoint_t) * cap * n),
                 oDesc = L.take(32 * cap * n), oUr = L.take(4 * cap * n), oDp = L.take(4 * cap * n);
    OPS_CHECK(o->ensure_dn(L.off));
    uint8_t* D = o->dn_h;
    OSLAM_HIP_CHECK(hipMemcpyAsync(D + oCnt, d_cnt, 4 * (size_t)n, hipMemcpyDeviceToHost, o->strm));
    OSLAM_HIP_CHECK(hipMemcpyAsync(D + oSt, d_st, 4, hipMemcpyDeviceToHost, o->strm));
    OSLAM_HIP_CHECK(hipMemcpyAsync(D + oSt + 4, o->d_status, 4, hipMemcpyDeviceToHost, o->strm));
    if (!o->lazy_keys) OSLAM_HIP_CHECK(hipMemcpyAsync(D + oKeys, d_kp, sizeof(oslam_keypoint_t) * cap * n, hipMemcpyDeviceToHost, o->strm));
    OSLAM_HIP_CHECK(hipMemcpyAsync(D + oKeysUn, o->d_keysUn, sizeof(oslam_keypoint_t) * cap * n, hipMemcpyDeviceToHost, o->strm));
    if (!o->lazy_desc) OSLAM_HIP_CHECK(hipMemcpyAsync(D + oDesc, d_desc, 32 * cap * n, hipMemcpyDeviceToHost, o->strm));
    OSLAM_HIP_CHECK(hipMemcpyAsync(D + oUr, d_uR, 4 * cap * n, hipMemcpyDeviceToHost, o->strm));
    OSLAM_HIP_CHECK(hipMemcpyAsync(D + oDp, d_dp, 4 * cap * n, hipMemcpyDeviceToHost, o->strm));
    OSLAM_HIP_CHECK(oslam::stream_wait(o->strm));
    const int32_t* st = (const int32_t*)(D + oSt);
    if (st[0]) { oslam::set_error("extractor arena overflow"); return OSLAM_E_CAPACITY; }
    if (st[1]) { oslam::set_error("keypoint outside the depth image / right extractor arena overflow"); return OSLAM_E_INVALID; }
    const int32_t* cnt = (const int32_t*)(D + oCnt);
    o->pool->parallel_for(n, [&](int i) {
        oslam_slam_frame_t* f = out[i];
        const size_t N = cnt[i], at = (size_t)i * cap;
        f->N = (int)N;
        if (!o->lazy_keys) memcpy(f->keys, D + oKeys + at * sizeof(oslam_keypoint_t), N * sizeof(oslam_keypoint_t));
        memcpy(f->keysUn, D + oKeysUn + at * sizeof(oslam_keypoint_t), N * sizeof(oslam_keypoint_t));
        if (!o->lazy_desc) memcpy(f->desc, D + oDesc + at * 32, N * 32);
        memcpy(f->uRight, D + oUr + at * 4, N * 4);
        memcpy(f->depth, D + oDp + at * 4, N * 4);
    });
    return OSLAM_OK;
}

// Frame::Frame (src/Frame.cc:117-172) for n frames: one batched extraction, one undistort launch, one depth lookup launch; the
// keypoints / descriptors come back in one pass of copies after a single synchronisation.
static int frames_impl(HipOps* o, int n, const uint8_t* const* gray, int gray_stride, const float* const* depth, const uint16_t* const* depth16, int depth_pitch,
                       float depth_factor, int on_device, oslam_slam_frame_t* const* out) {
    o->step_epoch++;   // a new step: the mask bitmaps of the previous one are stale
    std::swap(o->d_keysUn, o->d_keysUn_prev);   // what was the current frame of every slot is now its last frame
    OSLAM_HIP_CHECK(hipSetDevice(o->cfg.device));   // the HIP current device is per host thread: a handle may be stepped from any thread
    if (n > o->S) { oslam::set_error("frames_rgbd: n > n_sequences"); return OSLAM_E_INVALID; }
    const int W = o->cfg.width, H = o->cfg.height;
    const size_t gimg = o->gray_pitch * H, dimg = (size_t)W * H;
    const void* const* depth_table = nullptr;
    const void* const* dsrc = depth16 ? (const void* const*)depth16 : (const void* const*)depth;
    if (on_device) {   // two pointer tables up, ONE gather launch for the gray images (instead of 2n two-dimensional copies).  "Device" pointers only have to be
        // device-accessible: with pinned host images the gather and the depth lookup read them over PCIe where they are.
        OPS_CHECK(o->ensure_up(16 * (size_t)n + 512));
        memcpy(o->up_h, gray, 8 * (size_t)n); memcpy(o->up_h + 8 * (size_t)n + 256 - (8 * (size_t)n) % 256, dsrc, 8 * (size_t)n);
        const size_t oD = 8 * (size_t)n + 256 - (8 * (size_t)n) % 256;
        OSLAM_HIP_CHECK(hipMemcpyAsync(o->up_d, o->up_h, oD + 8 * (size_t)n, hipMemcpyHostToDevice, o->strm));
        OPS_CHECK(oslam_frame_gather_images_device((const void* const*)o->up_d, n, gray_stride, W, H, o->d_gray, gimg, (int)o->gray_pitch, o->strm));
        depth_table = (const void* const*)(o->up_d + oD);   // the depth images are read where they are: only the values at the keypoints are needed
    } else {
        // host images: rows packed into the pinned block in the device layout (parallel), then ONE copy per plane (a pageable 2-D copy is row-by-row);
        // raw 16-bit depth is scaled here the way convertTo scales it (one float multiplication per pixel)
        OPS_CHECK(o->ensure_up((gimg + dimg * 4) * n));
        uint8_t* U = o->up_h;
        o->pool->parallel_for(n, [&](int i) {
            for (int r = 0; r < H; r++) {
                memcpy(U + gimg * i + o->gray_pitch * r, gray[i] + (size_t)gray_stride * r, W);
                float* drow = (float*)(U + gimg * n + (dimg * i + (size_t)W * r) * 4);
                if (depth16) { const uint16_t* srow = depth16[i] + (size_t)depth_pitch * r; for (int x = 0; x < W; x++) drow[x] = (float)srow[x] * depth_factor; }
                else memcpy(drow, depth[i] + (size_t)depth_pitch * r, (size_t)W * 4);
            }
        });
        OSLAM_HIP_CHECK(hipMemcpyAsync(o->d_gray, U, gimg * n, hipMemcpyHostToDevice, o->strm));
        OSLAM_HIP_CHECK(hipMemcpyAsync(o->d_depth, U + gimg * n, dimg * 4 * n, hipMemcpyHostToDevice, o->strm));
    }
    o->t_begin();
    OPS_CHECK(oslam_orb_extract_batch_device(o->orb, o->d_gray, n, (int)o->gray_pitch, gimg, o->strm));
    const oslam_keypoint_t* d_kp; const uint8_t* d_desc; const int32_t* d_cnt; const int32_t* d_st;
    OPS_CHECK(oslam_orb_results_device(o->orb, &d_kp, &d_desc, &d_cnt, &d_st));
    OPS_CHECK(oslam_frame_undistort_batch_device(d_kp, o->d_keysUn, d_cnt, 0, o->cap, n, o->K4, o->cfg.dist, o->cfg.ndist, o->strm));
    if (depth_table && depth16) OPS_CHECK(oslam_frame_stereo_from_rgbd_batch_ptrs_u16_device(d_kp, o->d_keysUn, d_cnt, 0, o->cap, n, (const uint16_t* const*)depth_table, H, W, depth_pitch,
                                                                                              depth_factor, o->cfg.bf, o->d_uRight, o->d_mvDepth, o->d_status, o->strm));
    else if (depth_table) OPS_CHECK(oslam_frame_stereo_from_rgbd_batch_ptrs_device(d_kp, o->d_keysUn, d_cnt, 0, o->cap, n, (const float* const*)depth_table, H, W, depth_pitch, o->cfg.bf,
                                                                                   o->d_uRight, o->d_mvDepth, o->d_status, o->strm));
    else OPS_CHECK(oslam_frame_stereo_from_rgbd_batch_device(d_kp, o->d_keysUn, d_cnt, 0, o->cap, n, o->d_depth, H, W, W, dimg, o->cfg.bf, o->d_uRight,
                                                             o->d_mvDepth, o->d_status, o->strm));
    o->t_end();
    OPS_CHECK(download_frames(o, n, d_kp, d_desc, d_cnt, d_st, o->d_uRight, o->d_mvDepth, out));
    if (o->timing) { double bytes = 0; for (int i = 0; i < n; i++) bytes += (double)oslam_orb_algorithmic_bytes(o->orb, out[i]->N); o->t_collect(0, 14, bytes); }
    return OSLAM_OK;
}

int h_frames(void* p, int n, const int32_t* slots, const uint8_t* const* gray, int gray_stride, const float* const* depth, int depth_pitch, int on_device,
             oslam_slam_frame_t* const* out) {
    (void)slots;
    return frames_impl((HipOps*)p, n, gray, gray_stride, depth, nullptr, depth_pitch, 1.f, on_device, out);
}

// the same on raw 16-bit depth images (oslam_slam_track_rgbd_raw16)
int h_frames_raw16(void* p, int n, const int32_t* slots, const uint8_t* const* gray, int gray_stride, const uint16_t* const* depth16, int depth_pitch, float depth_factor,
                   int on_device, oslam_slam_frame_t* const* out) {
    (void)slots;
    return frames_impl((HipOps*)p, n, gray, gray_stride, nullptr, depth16, depth_pitch, depth_factor, on_device, out);
}

// Frame::Frame for n rectified stereo pairs (src/Frame.cc:61-115): both images extracted as two batches, UndistortKeyPoints, then ONE
// Frame::ComputeStereoMatches launch (one workgroup per pair) reading the two extractors' pyramids in HBM.
int h_frames_stereo(void* p, int n, const int32_t* slots, const uint8_t* const* left, const uint8_t* const* right, int gray_stride, int on_device,
                    oslam_slam_frame_t* const* out) {
    HipOps* o = (HipOps*)p;
    o->step_epoch++;
    std::swap(o->d_keysUn, o->d_keysUn_prev);
    OSLAM_HIP_CHECK(hipSetDevice(o->cfg.device));   // the HIP current device is per host thread: a handle may be stepped from any thread
    if (!o->orbR || !o->stereo) { oslam::set_error("frames_stereo: the handle was not created for the STEREO sensor"); return OSLAM_E_INVALID; }
    if (n > o->S) { oslam::set_error("frames_stereo: n > n_sequences"); return OSLAM_E_INVALID; }
    const int W = o->cfg.width, H = o->cfg.height;
    const size_t gimg = o->gray_pitch * H;
    if (on_device) {
        OPS_CHECK(o->ensure_up(16 * (size_t)n + 512));
        const size_t oD = 8 * (size_t)n + 256 - (8 * (size_t)n) % 256;
        memcpy(o->up_h, left, 8 * (size_t)n); memcpy(o->up_h + oD, right, 8 * (size_t)n);
        OSLAM_HIP_CHECK(hipMemcpyAsync(o->up_d, o->up_h, oD + 8 * (size_t)n, hipMemcpyHostToDevice, o->strm));
        OPS_CHECK(oslam_frame_gather_images_device((const void* const*)o->up_d, n, gray_stride, W, H, o->d_gray, gimg, (int)o->gray_pitch, o->strm));
        OPS_CHECK(oslam_frame_gather_images_device((const void* const*)(o->up_d + oD), n, gray_stride, W, H, o->d_grayR, gimg, (int)o->gray_pitch, o->strm));
    } else {
        OPS_CHECK(o->ensure_up(2 * gimg * n));
        uint8_t* U = o->up_h;
        o->pool->parallel_for(n, [&](int i) {
            for (int r = 0; r < H; r++) {
                memcpy(U + gimg * i + o->gray_pitch * r, left[i] + (size_t)gray_stride * r, W);
                memcpy(U + gimg * (n + i) + o->gray_pitch * r, right[i] + (size_t)gray_stride * r, W);
            }
        });
        OSLAM_HIP_CHECK(hipMemcpyAsync(o->d_gray, U, gimg * n, hipMemcpyHostToDevice, o->strm));
        OSLAM_HIP_CHECK(hipMemcpyAsync(o->d_grayR, U + gimg * n, gimg * n, hipMemcpyHostToDevice, o->strm));
    }
    o->t_begin();
    OPS_CHECK(oslam_orb_extract_batch_device(o->orb, o->d_gray, n, (int)o->gray_pitch, gimg, o->strm));
    OPS_CHECK(oslam_orb_extract_batch_device(o->orbR, o->d_grayR, n, (int)o->gray_pitch, gimg, o->strm));
    const oslam_keypoint_t* d_kp; const uint8_t* d_desc; const int32_t* d_cnt; const int32_t* d_st;
    const oslam_keypoint_t* d_kpR; const uint8_t* d_descR; const int32_t* d_cntR; const int32_t* d_stR;
    OPS_CHECK(oslam_orb_results_device(o->orb, &d_kp, &d_desc, &d_cnt, &d_st));
    OPS_CHECK(oslam_orb_results_device(o->orbR, &d_kpR, &d_descR, &d_cntR, &d_stR));
    OPS_CHECK(oslam_frame_undistort_batch_device(d_kp, o->d_keysUn, d_cnt, 0, o->cap, n, o->K4, o->cfg.dist, o->cfg.ndist, o->strm));
    OPS_CHECK(oslam_stereo_match_batch_device(o->stereo, o->orb, o->orbR, n, o->cap, d_kp, d_desc, d_cnt, 0, d_kpR, d_descR, d_cntR, 0, o->cfg.nLevels, o->cfg.bf,
                                              o->cfg.bf / o->cfg.fx, o->strm));
    const float* d_uR; const float* d_dp;
    OPS_CHECK(oslam_stereo_results_device(o->stereo, &d_uR, &d_dp, nullptr));
    o->t_end();
    OSLAM_HIP_CHECK(hipMemcpyAsync(o->d_status, d_stR, 4, hipMemcpyDeviceToDevice, o->strm));   // right extractor's overflow flag rides in the second status word
    (void)slots;
    OPS_CHECK(download_frames(o, n, d_kp, d_desc, d_cnt, d_st, d_uR, d_dp, out));
    if (o->timing) { double bytes = 0; for (int i = 0; i < n; i++) bytes += 2.0 * (double)oslam_orb_algorithmic_bytes(o->orb, out[i]->N); o->t_collect(0, 28, bytes); }
    return OSLAM_OK;
}

static void frames_view(HipOps* o, oslam_match_frames_t& fr, const uint8_t* d_blocked) {
    fr.keysUn = o->d_keysUn; fr.kp_stride = o->cap; fr.uRight = o->cur_uRight; fr.desc = o->d_desc; fr.blocked = d_blocked;
    fr.n_kps = o->d_cnt; fr.n_kps_const = 0;
    fr.minX = o->bounds[0]; fr.minY = o->bounds[1]; fr.maxX = o->bounds[2]; fr.maxY = o->bounds[3];
}

// ORBmatcher::SearchByProjection(Cur, Last) for the sequences in `jobs`: the current frames are still on the device (slot-major, from
// frames_rgbd); the last frames' map-point arrays are uploaded slot-major (slots without a job get zero queries), then one projection
// launch and one search launch cover all slots.
int h_search_last(void* p, int n, oslam_job_search_last_t* jobs) {
    HipOps* o = (HipOps*)p;
    OSLAM_HIP_CHECK(hipSetDevice(o->cfg.device));   // the HIP current device is per host thread: a handle may be stepped from any thread
    if (n == 0) return OSLAM_OK;
    const size_t S = o->S, cap = o->cap;
    for (int i = 0; i < n; i++)
        if (jobs[i].slot < 0 || jobs[i].slot >= (int)S || jobs[i].Nlast > (int)cap) { oslam::set_error("search_last: bad slot / size"); return OSLAM_E_INVALID; }
    // by id: positions and descriptors of the last frame's map points come from the resident records, its keypoints from the previous step's buffer
    bool by_id = o->mp_tab_on;
    for (int i = 0; i < n && by_id; i++) by_id = jobs[i].mp_ids != nullptr;
    if (by_id) OPS_CHECK(o->sync_mp_table());
    else for (int i = 0; i < n; i++) if (!jobs[i].Xw || !jobs[i].last_keysUn || !jobs[i].mp_desc) { oslam::set_error("search_last: job without arrays and without usable map-point ids"); return OSLAM_E_INVALID; }
    Layout L;
    const size_t oN = L.take(4 * S), oTc = L.take(64 * S), oTl = L.take(64 * S), oHas = L.take(cap * S), oIds = L.take(by_id ? 4 * cap * S : 0);
    const size_t head = L.off;
    const size_t oXw = L.take(12 * cap * S), oKeys = L.take(by_id ? 0 : sizeof(oslam_keypoint_t) * cap * S), oDesc = L.take(32 * cap * S);
    OPS_CHECK(o->ensure_up(L.off));
    uint8_t* U = o->up_h;
    memset(U + oN, 0, 4 * S);
    o->pool->parallel_for(n, [&](int i) {
        const oslam_job_search_last_t& j = jobs[i];
        const size_t b = j.slot, N = j.Nlast;
        ((int32_t*)(U + oN))[b] = j.Nlast;
        memcpy(U + oTc + 64 * b, j.Tcw, 64); memcpy(U + oTl + 64 * b, j.Tlw, 64);
        memcpy(U + oHas + cap * b, j.has_mp, N);
        if (by_id) { memcpy(U + oIds + 4 * cap * b, j.mp_ids, 4 * N); return; }
        memcpy(U + oXw + 12 * cap * b, j.Xw, 12 * N);
        memcpy(U + oKeys + sizeof(oslam_keypoint_t) * cap * b, j.last_keysUn, sizeof(oslam_keypoint_t) * N);
        memcpy(U + oDesc + 32 * cap * b, j.mp_desc, 32 * N);
    });
    OSLAM_HIP_CHECK(hipMemcpyAsync(o->up_d, U, by_id ? head : L.off, hipMemcpyHostToDevice, o->strm));
    uint8_t* Dv = o->up_d;
    if (by_id) OPS_CHECK(oslam_mp_table_gather_device((int)S, (int)cap, (const int32_t*)(Dv + oN), (const int32_t*)(Dv + oIds), o->d_mp_tab, (float*)(Dv + oXw), Dv + oDesc, o->strm));
    oslam_match_frames_t fr;
    frames_view(o, fr, nullptr);
    oslam_match_last_t la;
    la.Xw = (const float*)(Dv + oXw); la.has_mp = Dv + oHas; la.keys = by_id ? o->d_keysUn_prev : (const oslam_keypoint_t*)(Dv + oKeys); la.mp_desc = Dv + oDesc;
    la.kp_stride = (int)cap; la.n_kps = (const int32_t*)(Dv + oN); la.n_kps_const = 0;
    o->t_begin();
    OPS_CHECK(oslam_match_project_last_batch_device(o->m_last, &la, (const float*)(Dv + oTc), (const float*)(Dv + oTl), &o->cam, &fr, o->scale, o->cfg.nLevels,
                                                    jobs[0].th, 0, (int)S, o->strm));
    const int32_t* d_km; const int32_t* d_nm; const int32_t* d_nq;
    OPS_CHECK(oslam_match_results_device(o->m_last, nullptr, nullptr, &d_km, &d_nm, nullptr, &d_nq));
    OPS_CHECK(oslam_match_search_batch_device(o->m_last, &fr, nullptr, (int)cap, d_nq, 0, (int)S, 0.9f, 0, 1, 100, o->strm));
    o->t_end();
    Layout R;
    const size_t rKm = R.take(4 * cap * S), rNm = R.take(4 * S);
    OPS_CHECK(o->ensure_dn(R.off));
    OSLAM_HIP_CHECK(hipMemcpyAsync(o->dn_h + rKm, d_km, 4 * cap * S, hipMemcpyDeviceToHost, o->strm));
    OSLAM_HIP_CHECK(hipMemcpyAsync(o->dn_h + rNm, d_nm, 4 * S, hipMemcpyDeviceToHost, o->strm));
    OSLAM_HIP_CHECK(oslam::stream_wait(o->strm));
    o->t_collect(3, 2, 0);
    o->pool->parallel_for(n, [&](int i) {
        oslam_job_search_last_t& j = jobs[i];
        memcpy(j.kp_match, o->dn_h + rKm + 4 * cap * j.slot, 4 * (size_t)j.cur->N);
        j.nmatches = ((const int32_t*)(o->dn_h + rNm))[j.slot];
    });
    return OSLAM_OK;
}

// Tracking::SearchLocalPoints for the sequences in `jobs`: local points uploaded slot-major, ONE Frame::isInFrustum launch writes the
// projection queries on the device, ONE windowed search launch consumes them; only the match table and the in-view flags come back.
struct CopySegH { const uint8_t* src; uint8_t* dst; uint32_t bytes, pad; };   // oslam_copy_segments_device's record

int h_search_local(void* p, int n, oslam_job_search_local_t* jobs) {
    HipOps* o = (HipOps*)p;
    OSLAM_HIP_CHECK(hipSetDevice(o->cfg.device));   // the HIP current device is per host thread: a handle may be stepped from any thread
    if (n == 0) return OSLAM_OK;
    const size_t S = o->S, cap = o->cap;
    int maxM = 0;
    for (int i = 0; i < n; i++) {
        if (jobs[i].slot < 0 || jobs[i].slot >= (int)S) { oslam::set_error("search_local: bad slot"); return OSLAM_E_INVALID; }
        maxM = std::max(maxM, jobs[i].M);
    }
    if (maxM > o->max_local) {   // the reference's local map is unbounded: re-create the matcher with room to spare
        OSLAM_HIP_CHECK(oslam::stream_wait(o->strm));
        oslam_matcher_destroy(o->m_map);
        o->m_map = nullptr;
        o->max_local = (int)oslam::align_up((size_t)maxM + maxM / 2, 64);
        OPS_CHECK(oslam_matcher_create(&o->m_map, o->S, o->cap, o->max_local, o->cfg.device));
    }
    OPS_CHECK(o->ensure_loc());
    const size_t st = oslam::align_up((size_t)std::max(maxM, 1), 64), lst = o->loc_st;
    const bool resident = getenv("OSLAM_SLAM_NO_RESIDENT_LOCAL") == nullptr;
    // jobs whose packed arrays are not the ones the slot holds: their arrays travel (packed back to back) and one launch scatters them into the slots
    std::vector<int> fresh;
    std::vector<size_t> foff;
    size_t fbytes = 0;
    bool by_id = o->mp_tab_on;   // the fresh jobs name their points: the slot's arrays are gathered from the resident records (5 bytes per point travel, not 65)
    for (int i = 0; i < n; i++) {
        const oslam_job_search_local_t& j = jobs[i];
        if (resident && j.content_id != 0 && o->loc_id[j.slot] == j.content_id) continue;
        fresh.push_back(i);
        by_id = by_id && j.local_ids != nullptr;
    }
    if (by_id && !fresh.empty()) OPS_CHECK(o->sync_mp_table());
    int freshMaxM = 0;
    for (int i : fresh) {
        const oslam_job_search_local_t& j = jobs[i];
        if (!by_id && (!j.Pw || !j.Pn || !j.maxDist || !j.minDist || !j.mp_desc)) { oslam::set_error("search_local: job without arrays and without usable map-point ids"); return OSLAM_E_INVALID; }
        foff.push_back(fbytes);
        fbytes += by_id ? oslam::align_up(4 * (size_t)j.M, 16) + oslam::align_up((size_t)j.M, 64) : oslam::align_up(65 * (size_t)j.M + 6 * 16, 64);   // (six sub-arrays, each on a 16-byte boundary)
        freshMaxM = std::max(freshMaxM, j.M);
    }
    Layout L;
    const size_t oM = L.take(4 * S), oTc = L.take(64 * S), oTh = L.take(4 * S), oBl = L.take(cap * S), oSk = L.take(st * S),
                 oSeg = L.take(by_id ? sizeof(oslam_local_gather_t) * fresh.size() : sizeof(CopySegH) * 6 * fresh.size());
    const size_t small_bytes = L.off;
    const size_t oF = L.take(fbytes);
    OPS_CHECK(o->ensure_up(L.off));
    OPS_CHECK(o->ensure_lq(st * S));
    uint8_t* U = o->up_h;
    uint8_t* Dv = o->up_d;
    memset(U + oM, 0, 4 * S);
    memset(U + oTh, 0, 4 * S);
    o->pool->parallel_for(n, [&](int i) {
        const oslam_job_search_local_t& j = jobs[i];
        const size_t b = j.slot, M = j.M;
        ((int32_t*)(U + oM))[b] = j.M;
        ((float*)(U + oTh))[b] = j.th;
        memcpy(U + oTc + 64 * b, j.Tcw, 64);
        memcpy(U + oBl + cap * b, j.blocked, (size_t)j.cur->N);
        if (j.skip) memcpy(U + oSk + st * b, j.skip, M);
        else memset(U + oSk + st * b, 0, M);
    });
    o->pool->parallel_for((int)fresh.size(), [&](int q) {
        const oslam_job_search_local_t& j = jobs[fresh[q]];
        const size_t b = j.slot, M = j.M;
        uint8_t* at = U + oF + foff[q];
        if (by_id) {
            const size_t ooff = oslam::align_up(4 * M, 16);
            memcpy(at, j.local_ids, 4 * M); memcpy(at + ooff, j.obs_gt0, M);
            ((oslam_local_gather_t*)(U + oSeg))[q] = {(int32_t)b, (int32_t)M, (uint32_t)foff[q], (uint32_t)(foff[q] + ooff)};
            o->loc_id[b] = j.content_id;
            return;
        }
        const uint8_t* dv = Dv + oF + foff[q];
        CopySegH* sg = (CopySegH*)(U + oSeg) + 6 * (size_t)q;
        const void* src[6] = {j.Pw, j.Pn, j.maxDist, j.minDist, j.obs_gt0, j.mp_desc};
        const size_t bytes[6] = {12 * M, 12 * M, 4 * M, 4 * M, M, 32 * M};
        uint8_t* dst[6] = {(uint8_t*)(o->loc_Pw() + 3 * lst * b), (uint8_t*)(o->loc_Pn() + 3 * lst * b), (uint8_t*)(o->loc_Max() + lst * b),
                           (uint8_t*)(o->loc_Min() + lst * b), o->loc_Obs() + lst * b, o->loc_Desc() + 32 * lst * b};
        size_t off = 0;
        for (int k = 0; k < 6; k++) {   // sources and destinations are 16-byte aligned: the copy kernel moves 16 bytes per lane
            memcpy(at + off, src[k], bytes[k]);
            sg[k].src = dv + off; sg[k].dst = dst[k]; sg[k].bytes = (uint32_t)bytes[k]; sg[k].pad = 0;
            off += oslam::align_up(bytes[k], 16);
        }
        o->loc_id[b] = j.content_id;
    });
    OSLAM_HIP_CHECK(hipMemcpyAsync(Dv, U, fresh.empty() ? small_bytes : L.off, hipMemcpyHostToDevice, o->strm));
    o->t_begin();   // (the gather of repacked local maps counts with the searches it feeds)
    if (!fresh.empty() && by_id)
        OPS_CHECK(oslam_mp_table_local_gather_device((int)fresh.size(), freshMaxM, (const oslam_local_gather_t*)(Dv + oSeg), Dv + oF, o->d_mp_tab, (int)lst, o->loc_Pw(), o->loc_Pn(),
                                                     o->loc_Max(), o->loc_Min(), o->loc_Obs(), o->loc_Desc(), o->strm));
    else if (!fresh.empty()) OPS_CHECK(oslam_copy_segments_device(Dv + oSeg, 6 * (int)fresh.size(), o->strm));   // (a zero-byte segment's workgroup returns at once)
    OPS_CHECK(oslam_frame_is_in_frustum_batch_resident_device((int)S, (int)lst, (int)st, (const int32_t*)(Dv + oM), o->loc_Pw(), o->loc_Pn(), o->loc_Max(), o->loc_Min(),
                                                              o->loc_Obs(), o->loc_Desc(), Dv + oSk, (const float*)(Dv + oTc), (const float*)(Dv + oTh), o->K5,
                                                              o->bounds, 0.5f, o->logScale, o->scale, o->cfg.nLevels, o->d_lq, o->d_inview, o->strm));
    oslam_match_frames_t fr;
    frames_view(o, fr, Dv + oBl);
    OPS_CHECK(oslam_match_search_batch_device(o->m_map, &fr, o->d_lq, (int)st, (const int32_t*)(Dv + oM), 0, (int)S, 0.8f, 1, 0, 100, o->strm));
    o->t_end();
    const int32_t* d_km; const int32_t* d_nm;
    OPS_CHECK(oslam_match_results_device(o->m_map, nullptr, nullptr, &d_km, &d_nm, nullptr, nullptr));
    Layout R;
    const size_t rKm = R.take(4 * cap * S), rNm = R.take(4 * S), rIn = R.take(st * S);
    OPS_CHECK(o->ensure_dn(R.off));
    OSLAM_HIP_CHECK(hipMemcpyAsync(o->dn_h + rKm, d_km, 4 * cap * S, hipMemcpyDeviceToHost, o->strm));
    OSLAM_HIP_CHECK(hipMemcpyAsync(o->dn_h + rNm, d_nm, 4 * S, hipMemcpyDeviceToHost, o->strm));
    OSLAM_HIP_CHECK(hipMemcpyAsync(o->dn_h + rIn, o->d_inview, st * S, hipMemcpyDeviceToHost, o->strm));
    OSLAM_HIP_CHECK(oslam::stream_wait(o->strm));
    o->t_collect(3, 2, 0);
    o->pool->parallel_for(n, [&](int i) {
        oslam_job_search_local_t& j = jobs[i];
        memcpy(j.kp_match, o->dn_h + rKm + 4 * cap * j.slot, 4 * (size_t)j.cur->N);
        memcpy(j.in_view, o->dn_h + rIn + st * j.slot, (size_t)j.M);
        j.nmatches = ((const int32_t*)(o->dn_h + rNm))[j.slot];
    });
    return OSLAM_OK;
}

// Optimizer::PoseOptimization for n frames in one launch (one workgroup per frame).
int h_pose_opt(void* p, int n, oslam_job_pose_t* jobs) {
    HipOps* o = (HipOps*)p;
    OSLAM_HIP_CHECK(hipSetDevice(o->cfg.device));   // the HIP current device is per host thread: a handle may be stepped from any thread
    if (n == 0) return OSLAM_OK;
    if (n > o->S) { oslam::set_error("pose_opt: n > n_sequences"); return OSLAM_E_INVALID; }
    const size_t cap = o->cap, B = n;
    for (int i = 0; i < n; i++) if (jobs[i].N > (int)cap) { oslam::set_error("pose_opt: N > capacity"); return OSLAM_E_CAPACITY; }
    // by id: the positions come from the resident map-point records, obs / invSigma2 from the frames of this step that are still on the device
    bool by_id = o->mp_tab_on;
    for (int i = 0; i < n && by_id; i++) by_id = jobs[i].mp_ids != nullptr && jobs[i].slot >= 0 && jobs[i].slot < o->S;
    if (by_id) OPS_CHECK(o->sync_mp_table());
    else for (int i = 0; i < n; i++) if (!jobs[i].Xw || !jobs[i].obs || !jobs[i].invSigma2 || !jobs[i].has_mp) { oslam::set_error("pose_opt: job without arrays and without usable map-point ids"); return OSLAM_E_INVALID; }
    Layout L;
    const size_t oN = L.take(4 * B), oT = L.take(64 * B), oSl = L.take(by_id ? 4 * B : 0), oIds = L.take(by_id ? 4 * cap * B : 0);
    const size_t head = L.off;
    const size_t oXw = L.take(12 * cap * B), oObs = L.take(12 * cap * B), oInv = L.take(4 * cap * B), oHas = L.take(cap * B);
    OPS_CHECK(o->ensure_up(L.off));
    uint8_t* U = o->up_h;
    o->pool->parallel_for(n, [&](int i) {
        const oslam_job_pose_t& j = jobs[i];
        const size_t N = j.N;
        ((int32_t*)(U + oN))[i] = j.N;
        memcpy(U + oT + 64 * i, j.Tcw_in, 64);
        if (by_id) { ((int32_t*)(U + oSl))[i] = j.slot; memcpy(U + oIds + 4 * cap * i, j.mp_ids, 4 * N); return; }
        memcpy(U + oXw + 12 * cap * i, j.Xw, 12 * N); memcpy(U + oObs + 12 * cap * i, j.obs, 12 * N);
        memcpy(U + oInv + 4 * cap * i, j.invSigma2, 4 * N); memcpy(U + oHas + cap * i, j.has_mp, N);
    });
    OSLAM_HIP_CHECK(hipMemcpyAsync(o->up_d, U, by_id ? head : L.off, hipMemcpyHostToDevice, o->strm));
    uint8_t* Dv = o->up_d;
    if (by_id)
        OPS_CHECK(oslam_pose_inputs_gather_device(n, (int)cap, (const int32_t*)(Dv + oSl), (const int32_t*)(Dv + oN), (const int32_t*)(Dv + oIds), o->d_mp_tab, o->d_keysUn,
                                                  o->cur_uRight, (int)cap, o->invSigma2, o->cfg.nLevels, (float*)(Dv + oXw), (float*)(Dv + oObs), (float*)(Dv + oInv), Dv + oHas, o->strm));
    o->t_begin();
    OPS_CHECK(oslam_pose_optimize_batch_device(o->po, n, (int)cap, (const int32_t*)(Dv + oN), 0, (const float*)(Dv + oT), (const float*)(Dv + oXw),
                                               (const float*)(Dv + oObs), (const float*)(Dv + oInv), Dv + oHas, o->K5, o->strm));
    o->t_end();
    const float* d_T; const uint8_t* d_out; const int32_t* d_ni; const int32_t* d_stats;
    OPS_CHECK(oslam_poseopt_results_device(o->po, &d_T, &d_out, &d_ni, &d_stats));
    Layout R;
    const size_t rT = R.take(64 * B), rO = R.take(cap * B), rN = R.take(4 * B), rS = R.take(8 * B);
    OPS_CHECK(o->ensure_dn(R.off));
    if (o->timing) OSLAM_HIP_CHECK(hipMemcpyAsync(o->dn_h + rS, d_stats, 8 * B, hipMemcpyDeviceToHost, o->strm));
    OSLAM_HIP_CHECK(hipMemcpyAsync(o->dn_h + rT, d_T, 64 * B, hipMemcpyDeviceToHost, o->strm));
    OSLAM_HIP_CHECK(hipMemcpyAsync(o->dn_h + rO, d_out, cap * B, hipMemcpyDeviceToHost, o->strm));
    OSLAM_HIP_CHECK(hipMemcpyAsync(o->dn_h + rN, d_ni, 4 * B, hipMemcpyDeviceToHost, o->strm));
    OSLAM_HIP_CHECK(oslam::stream_wait(o->strm));
    if (o->timing) {   // 700 flop per edge and linearisation, 90 per edge and trial evaluation
        double flop = 0;
        const int32_t* stt = (const int32_t*)(o->dn_h + rS);
        for (int i = 0; i < n; i++) {
            int ne = 0;
            for (int k = 0; k < jobs[i].N; k++) ne += by_id ? jobs[i].mp_ids[k] >= 0 : jobs[i].has_mp[k] != 0;
            flop += (double)ne * (700.0 * stt[2 * i] + 90.0 * stt[2 * i + 1]);
        }
        o->t_collect(1, 1, flop);
    }
    o->pool->parallel_for(n, [&](int i) {
        oslam_job_pose_t& j = jobs[i];
        memcpy(j.Tcw_out, o->dn_h + rT + 64 * i, 64);
        memcpy(j.outlier, o->dn_h + rO + cap * i, (size_t)j.N);
        j.n_inliers = ((const int32_t*)(o->dn_h + rN))[i];
    });
    return OSLAM_OK;
}

// host masks of a stage -> d_maskstage (rows packed to W), returns the device pointer of mask m in ptrs[m]; device masks are used where they are
static int stage_masks(HipOps* o, int total, const std::vector<const uint8_t*>& src, int mask_stride, int on_device, std::vector<const uint8_t*>& ptrs, int& pitch) {
    const size_t W = o->cfg.width, H = o->cfg.height;
    ptrs.resize(total);
    if (on_device) { for (int m = 0; m < total; m++) ptrs[m] = src[m]; pitch = mask_stride; return OSLAM_OK; }
    pitch = (int)W;
    OPS_CHECK(o->ensure_masks(W * H * (size_t)total));
    for (int m = 0; m < total; m++) {
        OSLAM_HIP_CHECK(hipMemcpy2DAsync(o->d_maskstage + W * H * m, W, src[m], mask_stride, W, H, hipMemcpyHostToDevice, o->strm));
        ptrs[m] = o->d_maskstage + W * H * m;
    }
    return OSLAM_OK;
}

// Frame::BuildObject2DsRGBD keypoint test for the frames still on the device (slot-major): one launch, one byte per keypoint back
int h_object_kps(void* p, int n, oslam_job_object_kps_t* jobs) {
    HipOps* o = (HipOps*)p;
    OSLAM_HIP_CHECK(hipSetDevice(o->cfg.device));
    if (n == 0) return OSLAM_OK;
    const size_t S = o->S, cap = o->cap;
    std::vector<const uint8_t*> src;
    std::vector<int32_t> mask0(S, 0), nmask(S, 0);
    for (int i = 0; i < n; i++) {
        const oslam_job_object_kps_t& j = jobs[i];
        if (j.slot < 0 || j.slot >= (int)S || j.n_masks < 0 || j.n_masks > OSLAM_SLAM_MAX_OBJECTS || j.on_device != jobs[0].on_device || j.mask_stride != jobs[0].mask_stride) {
            oslam::set_error("object_kps: bad slot / mask count / mixed mask layouts"); return OSLAM_E_INVALID;
        }
        mask0[j.slot] = (int32_t)src.size(); nmask[j.slot] = j.n_masks;
        for (int m = 0; m < j.n_masks; m++) src.push_back(j.masks[m]);
    }
    const int total = (int)src.size();
    if (total == 0) { for (int i = 0; i < n; i++) memset(jobs[i].in_mask, 0, (size_t)jobs[i].cur->N); return OSLAM_OK; }
    std::vector<const uint8_t*> ptrs;
    int pitch = 0;
    const bool caller_bits = jobs[0].mask_stride == 0;   // one-bit-per-pixel images packed by the caller (oslam_slam_track_rgbd_raw16), device-accessible
    if (caller_bits && !jobs[0].on_device) { oslam::set_error("object_kps: one-bit masks must be device-accessible"); return OSLAM_E_INVALID; }
    if (!caller_bits) OPS_CHECK(stage_masks(o, total, src, jobs[0].mask_stride, jobs[0].on_device, ptrs, pitch));
    Layout L;
    const size_t oPtr = L.take(8 * (size_t)total), oM0 = L.take(4 * S), oNm = L.take(4 * S), oSeg = L.take(caller_bits ? sizeof(CopySegH) * (size_t)total : 0);
    OPS_CHECK(o->ensure_up(L.off));
    uint8_t* U = o->up_h;
    const size_t bits_bytes = (size_t)o->cfg.height * ((o->cfg.width + 63) / 64) * 8;
    if (caller_bits) {
        OPS_CHECK(o->ensure_maskbits((size_t)total * o->cfg.height * ((o->cfg.width + 63) / 64)));
        CopySegH* sg = (CopySegH*)(U + oSeg);
        for (int m = 0; m < total; m++) { sg[m].src = src[m]; sg[m].dst = (uint8_t*)o->d_maskbits + bits_bytes * m; sg[m].bytes = (uint32_t)bits_bytes; sg[m].pad = 0; }
    } else memcpy(U + oPtr, ptrs.data(), 8 * (size_t)total);
    memcpy(U + oM0, mask0.data(), 4 * S); memcpy(U + oNm, nmask.data(), 4 * S);
    OSLAM_HIP_CHECK(hipMemcpyAsync(o->up_d, U, L.off, hipMemcpyHostToDevice, o->strm));
    uint8_t* Dv = o->up_d;
    o->t_begin();
    if (caller_bits || !getenv("OSLAM_SLAM_NO_MASK_BITS")) {   // one pass over the mask bytes; the test (and pose_opt2's boundary lists later in this step) read bitmaps
        const int H = o->cfg.height, W = o->cfg.width;
        OPS_CHECK(o->ensure_maskbits((size_t)total * H * ((W + 63) / 64)));
        if (caller_bits) OPS_CHECK(oslam_copy_segments_device(Dv + oSeg, total, o->strm));   // the caller's bitmaps (pinned host memory: read over PCIe) into the step's bitmap array
        else OPS_CHECK(oslam_mask_bits_device((const uint8_t* const*)(Dv + oPtr), total, H, W, pitch, o->d_maskbits, o->strm));
        OPS_CHECK(oslam_frame_object_kp_test_bits_batch_device(o->d_keysUn, (int)cap, o->d_cnt, (int)S, o->d_maskbits, (const int32_t*)(Dv + oM0), (const int32_t*)(Dv + oNm), H, W,
                                                               o->d_objbits, o->strm));
        o->bits_of_ptr.clear();
        for (int m = 0; m < total; m++) o->bits_of_ptr[src[m]] = m;
        o->bits_epoch = o->step_epoch;
    } else
    OPS_CHECK(oslam_frame_object_kp_test_batch_device(o->d_keysUn, (int)cap, o->d_cnt, (int)S, (const uint8_t* const*)(Dv + oPtr), (const int32_t*)(Dv + oM0),
                                                      (const int32_t*)(Dv + oNm), o->cfg.height, o->cfg.width, pitch, o->d_objbits, o->strm));
    o->t_end();
    OPS_CHECK(o->ensure_dn(cap * S));
    OSLAM_HIP_CHECK(hipMemcpyAsync(o->dn_h, o->d_objbits, cap * S, hipMemcpyDeviceToHost, o->strm));
    OSLAM_HIP_CHECK(oslam::stream_wait(o->strm));
    o->t_collect(7, 2, 0);
    for (int i = 0; i < n; i++) memcpy(jobs[i].in_mask, o->dn_h + cap * jobs[i].slot, (size_t)jobs[i].cur->N);
    return OSLAM_OK;
}

// ObjectOptimizer::PoseOptimization2 for n frames in one launch (one workgroup per frame); the masks are read where they are (device) or staged (host)
int h_pose_opt2(void* p, int n, oslam_job_pose2_t* jobs) {
    HipOps* o = (HipOps*)p;
    OSLAM_HIP_CHECK(hipSetDevice(o->cfg.device));
    if (n == 0) return OSLAM_OK;
    if (n > o->S) { oslam::set_error("pose_opt2: n > n_sequences"); return OSLAM_E_INVALID; }
    const size_t cap = o->cap, B = n;
    std::vector<oslam_sem_frame_t> fr(n);
    std::vector<const uint8_t*> src;
    int tObj = 0, tMp = 0, tJ = 0;
    for (int i = 0; i < n; i++) {
        const oslam_job_pose2_t& j = jobs[i];
        if (j.base.N > (int)cap || j.nObj < 0 || j.nObj > OSLAM_SLAM_MAX_OBJECTS || j.on_device != jobs[0].on_device || j.mask_stride != jobs[0].mask_stride) {
            oslam::set_error("pose_opt2: N > capacity / bad object count / mixed mask layouts"); return OSLAM_E_INVALID;
        }
        fr[i].nObj = j.nObj; fr[i].obj0 = tObj; fr[i].nObjMp = j.nObjMp; fr[i].objmp0 = tMp; fr[i].nJoint = j.nJoint; fr[i].joint0 = tJ;
        for (int m = 0; m < j.nObj; m++) src.push_back(j.masks[m]);
        tObj += j.nObj; tMp += j.nObjMp; tJ += j.nJoint;
    }
    std::vector<const uint8_t*> ptrs;
    int pitch = o->cfg.width;
    // with this step's bitmaps (object_kps) the kernel reads no mask byte: nothing to stage
    bool use_bits = tObj > 0 && o->bits_epoch == o->step_epoch && o->d_maskbits;
    std::vector<int32_t> bidx(tObj);
    for (int m = 0; m < tObj && use_bits; m++) {
        const auto it = o->bits_of_ptr.find(src[m]);
        if (it == o->bits_of_ptr.end()) use_bits = false;
        else bidx[m] = it->second;
    }
    if (tObj && !use_bits && jobs[0].mask_stride == 0) { oslam::set_error("pose_opt2: one-bit masks without the step's bitmaps (object_kps must run first)"); return OSLAM_E_INVALID; }
    if (tObj && !use_bits) OPS_CHECK(stage_masks(o, tObj, src, jobs[0].mask_stride, jobs[0].on_device, ptrs, pitch));
    bool by_id = o->mp_tab_on;
    for (int i = 0; i < n && by_id; i++) by_id = jobs[i].base.mp_ids != nullptr && jobs[i].base.slot >= 0 && jobs[i].base.slot < o->S && (jobs[i].nObjMp == 0 || jobs[i].objmp_ids != nullptr);
    if (by_id) OPS_CHECK(o->sync_mp_table());
    else for (int i = 0; i < n; i++) if ((jobs[i].nObjMp > 0 && !jobs[i].objmp_Xw) || !jobs[i].base.Xw || !jobs[i].base.obs || !jobs[i].base.invSigma2 || !jobs[i].base.has_mp) { oslam::set_error("pose_opt2: job without arrays and without usable map-point ids"); return OSLAM_E_INVALID; }
    Layout L;
    // (the arrays the device builds itself when the frames are served by id sit behind everything that is uploaded)
    const size_t oN = L.take(4 * B), oT = L.take(64 * B), oSl = L.take(by_id ? 4 * B : 0), oIds = L.take(by_id ? 4 * cap * B : 0),
                 oFr = L.take(sizeof(oslam_sem_frame_t) * B), oPtr = L.take(8 * (size_t)tObj), oMx = L.take(by_id ? 0 : 12 * (size_t)tMp), oMo = L.take(4 * (size_t)tMp),
                 oJk = L.take(4 * (size_t)tJ), oJo = L.take(4 * (size_t)tJ), oBi = L.take(4 * (size_t)tObj), oMi = L.take(by_id ? 4 * (size_t)tMp : 0),
                 oMs = L.take(by_id ? 4 * (size_t)tMp : 0);
    const size_t head = L.off;
    const size_t oMxDev = by_id ? L.take(12 * (size_t)tMp) : oMx;   // object map-point positions: uploaded, or gathered from the records
    const size_t oXw = L.take(12 * cap * B), oObs = L.take(12 * cap * B), oInv = L.take(4 * cap * B), oHas = L.take(cap * B);
    OPS_CHECK(o->ensure_up(L.off));
    uint8_t* U = o->up_h;
    memcpy(U + oFr, fr.data(), sizeof(oslam_sem_frame_t) * B);
    if (use_bits) memcpy(U + oBi, bidx.data(), 4 * (size_t)tObj);
    if (tObj && !use_bits) memcpy(U + oPtr, ptrs.data(), 8 * (size_t)tObj);
    o->pool->parallel_for(n, [&](int i) {
        const oslam_job_pose2_t& j2 = jobs[i];
        const oslam_job_pose_t& j = j2.base;
        const size_t N = j.N;
        ((int32_t*)(U + oN))[i] = j.N;
        memcpy(U + oT + 64 * i, j.Tcw_in, 64);
        if (by_id) { ((int32_t*)(U + oSl))[i] = j.slot; memcpy(U + oIds + 4 * cap * i, j.mp_ids, 4 * N); }
        else {
            memcpy(U + oXw + 12 * cap * i, j.Xw, 12 * N); memcpy(U + oObs + 12 * cap * i, j.obs, 12 * N);
            memcpy(U + oInv + 4 * cap * i, j.invSigma2, 4 * N); memcpy(U + oHas + cap * i, j.has_mp, N);
        }
        if (j2.nObjMp) {
            memcpy(U + oMo + 4 * (size_t)fr[i].objmp0, j2.objmp_obj, 4 * (size_t)j2.nObjMp);
            if (by_id) {
                memcpy(U + oMi + 4 * (size_t)fr[i].objmp0, j2.objmp_ids, 4 * (size_t)j2.nObjMp);
                int32_t* sl = (int32_t*)(U + oMs) + fr[i].objmp0;
                for (int q = 0; q < j2.nObjMp; q++) sl[q] = j.slot;
            } else memcpy(U + oMx + 12 * (size_t)fr[i].objmp0, j2.objmp_Xw, 12 * (size_t)j2.nObjMp);
        }
        if (j2.nJoint) { memcpy(U + oJk + 4 * (size_t)fr[i].joint0, j2.joint_kp, 4 * (size_t)j2.nJoint); memcpy(U + oJo + 4 * (size_t)fr[i].joint0, j2.joint_obj, 4 * (size_t)j2.nJoint); }
    });
    OSLAM_HIP_CHECK(hipMemcpyAsync(o->up_d, U, by_id ? head : L.off, hipMemcpyHostToDevice, o->strm));
    uint8_t* Dv = o->up_d;
    if (by_id) {
        OPS_CHECK(oslam_pose_inputs_gather_device(n, (int)cap, (const int32_t*)(Dv + oSl), (const int32_t*)(Dv + oN), (const int32_t*)(Dv + oIds), o->d_mp_tab, o->d_keysUn,
                                                  o->cur_uRight, (int)cap, o->invSigma2, o->cfg.nLevels, (float*)(Dv + oXw), (float*)(Dv + oObs), (float*)(Dv + oInv), Dv + oHas, o->strm));
        OPS_CHECK(oslam_mp_table_positions_device(tMp, (const int32_t*)(Dv + oMs), (const int32_t*)(Dv + oMi), o->d_mp_tab, (float*)(Dv + oMxDev), o->strm));
    }
    o->t_begin();
    if (use_bits) OPS_CHECK(oslam_poseopt_use_mask_bits(o->po, o->d_maskbits, (const int32_t*)(Dv + oBi)));
    OPS_CHECK(oslam_pose_optimize2_batch_device(o->po, n, (int)cap, (const int32_t*)(Dv + oN), (const float*)(Dv + oT), (const float*)(Dv + oXw), (const float*)(Dv + oObs),
                                                (const float*)(Dv + oInv), Dv + oHas, o->K5, (const oslam_sem_frame_t*)(Dv + oFr), tObj, use_bits ? nullptr : (const uint8_t* const*)(Dv + oPtr),
                                                o->cfg.height, o->cfg.width, pitch, tMp, (const float*)(Dv + oMxDev), (const int32_t*)(Dv + oMo), tJ, (const int32_t*)(Dv + oJk),
                                                (const int32_t*)(Dv + oJo), o->bounds, o->invSigma2[0], o->strm));
    o->t_end();
    const float* d_T; const uint8_t* d_out; const int32_t* d_ni; const int32_t* d_stats; const int32_t* d_ns;
    OPS_CHECK(oslam_poseopt_results_device(o->po, &d_T, &d_out, &d_ni, &d_stats));
    OPS_CHECK(oslam_poseopt_semantic_results_device(o->po, &d_ns));
    Layout R;
    const size_t rT = R.take(64 * B), rO = R.take(cap * B), rN = R.take(4 * B), rS = R.take(8 * B), rNs = R.take(4 * B);
    OPS_CHECK(o->ensure_dn(R.off));
    OSLAM_HIP_CHECK(hipMemcpyAsync(o->dn_h + rT, d_T, 64 * B, hipMemcpyDeviceToHost, o->strm));
    OSLAM_HIP_CHECK(hipMemcpyAsync(o->dn_h + rO, d_out, cap * B, hipMemcpyDeviceToHost, o->strm));
    OSLAM_HIP_CHECK(hipMemcpyAsync(o->dn_h + rN, d_ni, 4 * B, hipMemcpyDeviceToHost, o->strm));
    OSLAM_HIP_CHECK(hipMemcpyAsync(o->dn_h + rS, d_stats, 8 * B, hipMemcpyDeviceToHost, o->strm));
    OSLAM_HIP_CHECK(hipMemcpyAsync(o->dn_h + rNs, d_ns, 4 * B, hipMemcpyDeviceToHost, o->strm));
    OSLAM_HIP_CHECK(oslam::stream_wait(o->strm));
    if (o->timing) {
        double flop = 0;
        const int32_t* stt = (const int32_t*)(o->dn_h + rS);
        for (int i = 0; i < n; i++) {
            int ne = jobs[i].nObjMp + jobs[i].nJoint;   // upper bound of the semantic edges
            for (int k = 0; k < jobs[i].base.N; k++) ne += by_id ? jobs[i].base.mp_ids[k] >= 0 : jobs[i].base.has_mp[k] != 0;
            flop += (double)ne * (700.0 * stt[2 * i] + 90.0 * stt[2 * i + 1]);
        }
        o->t_collect(1, 5, flop);
    }
    o->pool->parallel_for(n, [&](int i) {
        oslam_job_pose_t& j = jobs[i].base;
        memcpy(j.Tcw_out, o->dn_h + rT + 64 * i, 64);
        memcpy(j.outlier, o->dn_h + rO + cap * i, (size_t)j.N);
        j.n_inliers = ((const int32_t*)(o->dn_h + rN))[i];
        jobs[i].n_semantic = ((const int32_t*)(o->dn_h + rNs))[i];
    });
    return OSLAM_OK;
}

// MapPoint::ComputeDistinctiveDescriptors + UpdateNormalAndDepth over the touched points of all sequences: one block up, two launches, one block down
static int mp_update_impl(HipOps* o, oslam_job_mp_update_t* j, const int32_t* obs_key, bool defer = false);
int h_mp_update_collect(void* p);
int h_mp_update(void* p, oslam_job_mp_update_t* j) { int rc = h_mp_update_collect(p); return rc ? rc : mp_update_impl((HipOps*)p, j, nullptr); }
int h_mp_update_keyed(void* p, oslam_job_mp_update_t* j, const int32_t* obs_key) { int rc = h_mp_update_collect(p); return rc ? rc : mp_update_impl((HipOps*)p, j, obs_key); }
// mvKeys of the keyframes registered by the LAST register_keyframes call (include/oslam_slam.h): staged by that call's copy launch, handed out here
int h_keyframe_raw_keys(void* p, int n, const int32_t* slots, const int32_t* counts, oslam_keypoint_t* const* out) {
    HipOps* o = (HipOps*)p;
    if (!o->lazy_keys) { oslam::set_error("keyframe_raw_keys: the table sends mvKeys with every frame"); return OSLAM_E_INVALID; }
    if (n != (int)o->kfk_slots.size()) { oslam::set_error("keyframe_raw_keys: not the keyframes of the last register_keyframes call"); return OSLAM_E_INVALID; }
    const size_t cap = o->cap;
    for (int i = 0; i < n; i++) {
        if (slots[i] != o->kfk_slots[i] || counts[i] < 0 || (size_t)counts[i] > cap) { oslam::set_error("keyframe_raw_keys: not the keyframes of the last register_keyframes call"); return OSLAM_E_INVALID; }
        memcpy(out[i], o->kfk_h + (size_t)i * cap * sizeof(oslam_keypoint_t), (size_t)counts[i] * sizeof(oslam_keypoint_t));
    }
    return OSLAM_OK;
}
int h_keyframe_descriptors(void* p, int n, const int32_t* slots, const int32_t* counts, uint8_t* const* out) {
    HipOps* o = (HipOps*)p;
    if (!o->lazy_desc || n != (int)o->kfk_slots.size()) { oslam::set_error("keyframe_descriptors: not the keyframes of the last register_keyframes call"); return OSLAM_E_INVALID; }
    const size_t cap = o->cap;
    const uint8_t* base = o->kfk_h + (size_t)n * cap * sizeof(oslam_keypoint_t);
    for (int i = 0; i < n; i++) {
        if (slots[i] != o->kfk_slots[i] || counts[i] < 0 || (size_t)counts[i] > cap) { oslam::set_error("keyframe_descriptors: not the keyframes of the last register_keyframes call"); return OSLAM_E_INVALID; }
        memcpy(out[i], base + (size_t)i * cap * 32, (size_t)counts[i] * 32);
    }
    return OSLAM_OK;
}
// mDescriptors of the current frames of `slots`, from the extractor's batch arrays (still on the device until the next Frame::Frame stage)
int h_frame_descriptors(void* p, int n, const int32_t* slots, const int32_t* counts, uint8_t* const* out) {
    HipOps* o = (HipOps*)p;
    OSLAM_HIP_CHECK(hipSetDevice(o->cfg.device));
    if (n == 0) return OSLAM_OK;
    if (!o->d_desc) { oslam::set_error("frame_descriptors: no frame has been built yet"); return OSLAM_E_INVALID; }
    const size_t cap = o->cap;
    OPS_CHECK(o->ensure_dn((size_t)n * cap * 32));
    std::vector<CopySegH> segs(n);
    for (int i = 0; i < n; i++) {
        if (slots[i] < 0 || slots[i] >= o->S || counts[i] < 0 || (size_t)counts[i] > cap) { oslam::set_error("frame_descriptors: bad slot / count"); return OSLAM_E_INVALID; }
        segs[i] = {o->d_desc + 32 * cap * slots[i], o->dn_h + (size_t)i * cap * 32, (uint32_t)(cap * 32), 0};
    }
    OPS_CHECK(o->ensure_up(segs.size() * sizeof(CopySegH)));
    memcpy(o->up_h, segs.data(), segs.size() * sizeof(CopySegH));
    OSLAM_HIP_CHECK(hipMemcpyAsync(o->up_d, o->up_h, segs.size() * sizeof(CopySegH), hipMemcpyHostToDevice, o->strm));
    OPS_CHECK(oslam_copy_segments_device(o->up_d, (int)segs.size(), o->strm));
    OSLAM_HIP_CHECK(oslam::stream_wait(o->strm));
    for (int i = 0; i < n; i++) memcpy(out[i], o->dn_h + (size_t)i * cap * 32, (size_t)counts[i] * 32);
    return OSLAM_OK;
}
// The deferred form: only the one-launch path (k_mp_update_fused) is deferred — it needs no staging beyond the job block and writes its results into a pinned
// block by itself; any other job is run to completion here (mp_update_collect then has nothing to wait for).
int h_mp_update_keyed_async(void* p, oslam_job_mp_update_t* j, const int32_t* obs_key) {
    HipOps* o = (HipOps*)p;
    int rc = h_mp_update_collect(p);
    if (rc) return rc;
    o->swap_staging();
    rc = mp_update_impl(o, j, obs_key, true);
    o->swap_staging();
    return rc;
}
int h_mp_update_collect(void* p) {
    HipOps* o = (HipOps*)p;
    if (!o->mpu_pend.on) return OSLAM_OK;
    OSLAM_HIP_CHECK(hipSetDevice(o->cfg.device));
    OPS_CHECK(o->mpu_launch_pending());
    OSLAM_HIP_CHECK(oslam::stream_wait(o->strm));
    HipOps::MpuPending& q = o->mpu_pend;
    q.on = false;
    o->swap_staging();   // (the deferred job's event pair and result block)
    o->t_collect(6, 1, q.dtotal);
    if (q.j->do_desc) { memcpy(q.j->best_idx, o->dn_h + q.rBest, 4 * q.P); memcpy(q.j->out_desc, o->dn_h + q.rOut, 32 * q.P); }
    if (q.j->do_normal) memcpy(q.j->out5, o->dn_h + q.rOut5, 20 * q.P);
    o->swap_staging();
    return OSLAM_OK;
}

// OSLAM_MPU_PROF=1: wall-clock split of the operator summed over all calls of the process, printed at exit
struct MpuOpProf {
    std::atomic<long long> ns[6];
    MpuOpProf() { for (auto& x : ns) x = 0; }
    ~MpuOpProf() {
        fprintf(stderr, "[mp_update operator prof] record lookup %.1f ms, table growth / sync %.1f ms, staging %.1f ms, enqueue %.1f ms, wait %.1f ms, copy out %.1f ms\n", ns[0] * 1e-6, ns[1] * 1e-6,
                ns[2] * 1e-6, ns[3] * 1e-6, ns[4] * 1e-6, ns[5] * 1e-6);
    }
};
static MpuOpProf* mpu_op_prof() { static MpuOpProf* p = getenv("OSLAM_MPU_PROF") ? new MpuOpProf : nullptr; static struct D { ~D() { delete mpu_op_prof(); } } d; return p; }

static int mp_update_impl(HipOps* o, oslam_job_mp_update_t* j, const int32_t* obs_key, bool defer) {
    OSLAM_HIP_CHECK(hipSetDevice(o->cfg.device));   // the HIP current device is per host thread: a handle may be stepped from any thread
    const size_t P = j->P;
    if (P == 0) return OSLAM_OK;
    MpuOpProf* pf = mpu_op_prof();
    auto t0_ = std::chrono::steady_clock::now();
    auto lap_ = [&](int k) { if (pf) { auto t1_ = std::chrono::steady_clock::now(); pf->ns[k] += std::chrono::duration_cast<std::chrono::nanoseconds>(t1_ - t0_).count(); t0_ = t1_; } };
    const size_t total = (size_t)j->obs_start[P];
    // the descriptor selection has its own observation list when some observations sit in culled keyframes (oslam_job_mp_update_t::desc_start)
    const int32_t* dstart = j->desc_start ? j->desc_start : j->obs_start;
    const size_t dtotal = (size_t)dstart[P];
    Layout L;
    // with resident keyframes the observations' descriptors are gathered on the device from (record, keypoint) pairs: 8 bytes per observation travel
    // instead of 32, and the caller did not have to collect them
    std::vector<int32_t>& rec = o->mpu_rec;   // (keeps its capacity: 8 bytes per observation, up to ~10 MB per call after a local BA)
    bool keyed = obs_key && j->do_desc && dtotal > 0;
    if (keyed) {   // (hundreds of thousands of observations per call: the lookup runs on the shared workers)
        rec.resize(2 * dtotal);
        const size_t chunk = 16384, nch = (dtotal + chunk - 1) / chunk;
        std::atomic<int> missing{0};
        o->pool->parallel_for((int)nch, [&](int ch) {
            const size_t e0 = (size_t)ch * chunk, e1 = std::min(dtotal, e0 + chunk);
            bool ok = true;
            for (size_t e = e0; e < e1; e++) {
                const int r = o->rec_lookup(obs_key[3 * e], obs_key[3 * e + 1]);
                if (r < 0 || obs_key[3 * e + 2] < 0 || obs_key[3 * e + 2] >= o->cap) ok = false;
                rec[2 * e] = r; rec[2 * e + 1] = obs_key[3 * e + 2];
            }
            if (!ok) missing.fetch_add(1, std::memory_order_relaxed);
        });
        if (missing.load() > 0) keyed = false;
        if (!keyed && !j->obs_desc) { oslam::set_error("mp_update: observation of a keyframe that is not resident"); return OSLAM_E_INVALID; }
    }
    lap_(0);
    const bool table = o->mp_tab_on && j->items != nullptr;
    if (table) {   // room for the records of the points named (ids grow with the map)
        std::vector<int> mx(o->S, -1);
        for (size_t i = 0; i < P; i++) {
            const int sl = j->items[2 * i], id = j->items[2 * i + 1];
            if (sl < 0 || sl >= o->S || id < 0) { oslam::set_error("mp_update: bad item"); return OSLAM_E_INVALID; }
            mx[sl] = std::max(mx[sl], id);
        }
        for (int sl = 0; sl < o->S; sl++) if (mx[sl] >= 0) OPS_CHECK(o->ensure_mp_records(sl, (size_t)mx[sl] + 1));
        OPS_CHECK(o->sync_mp_table());
    }
    lap_(1);
    const size_t oItems = L.take(table ? 8 * P : 0);
    const size_t oStart = L.take(4 * (P + 1)), oDStart = L.take(j->desc_start ? 4 * (P + 1) : 0), oRec = L.take(keyed ? 8 * dtotal : 0), oOw = L.take(12 * total),
                 oPos = L.take(12 * P), oRef = L.take(12 * P), oLsf = L.take(4 * P), oDescUp = L.take(keyed ? 0 : 32 * dtotal);
    const size_t in_bytes = L.off;
    const size_t oDesc = keyed ? L.take(32 * dtotal) : oDescUp;
    const size_t oBest = L.take(4 * P), oOut = L.take(32 * P), oOut5 = L.take(20 * P);
    OPS_CHECK(o->ensure_up(L.off));
    uint8_t* U = o->up_h;
    uint8_t* Dv = o->up_d;
    memcpy(U + oStart, j->obs_start, 4 * (P + 1));
    if (table) memcpy(U + oItems, j->items, 8 * P);
    if (j->desc_start) memcpy(U + oDStart, j->desc_start, 4 * (P + 1));
    if (keyed) memcpy(U + oRec, rec.data(), 8 * dtotal);
    else if (j->do_desc) memcpy(U + oDesc, j->obs_desc, 32 * dtotal);
    if (j->do_normal) { memcpy(U + oOw, j->obs_Ow, 12 * total); memcpy(U + oPos, j->Pos, 12 * P); memcpy(U + oRef, j->OwRef, 12 * P); memcpy(U + oLsf, j->levelScaleFactor, 4 * P); }
    lap_(2);
    // The job block is read by the kernels where it is (pinned, mapped into the device's address space) unless it carries the observations' descriptors, which
    // k_distinctive reads many times: every array of the keyed form is read once or twice, and the copy engine hop + its dependency cost more than the PCIe
    // reads (Fuse: +4.4 % frames/s, same box).  Results come back through the copy kernel for the same reason.  OSLAM_SLAM_MPU_UPLOAD=1: the staged path.
    static const bool force_upload = getenv("OSLAM_SLAM_MPU_UPLOAD") != nullptr;
    // (only the SMALL jobs — the per-round descriptor updates of SearchInNeighbors: the large ones after a local BA read 12 bytes per observation, their kernels
    // then hold the device 20 % longer and the whole bench loses 4 %)
    const bool upload = force_upload || (j->do_desc && !keyed) || in_bytes > (size_t)(getenv("OSLAM_SLAM_MPU_ZC_BYTES") ? atol(getenv("OSLAM_SLAM_MPU_ZC_BYTES")) : 262144);
    const uint8_t* In = upload ? Dv : U;
    if (upload) OSLAM_HIP_CHECK(hipMemcpyAsync(Dv, U, in_bytes, hipMemcpyHostToDevice, o->strm));
    Layout R;
    const size_t rBest = R.take(4 * P), rOut = R.take(32 * P), rOut5 = R.take(20 * P);
    OPS_CHECK(o->ensure_dn(R.off));
    // The small keyed jobs (every Fuse round's descriptor updates, the new keyframe's and the new points' updates) in ONE launch: k_mp_update_fused reads the
    // observations' descriptors from the resident keyframes, writes the results straight into the (device-accessible) result block and into the resident records.
    // OSLAM_SLAM_MPU_FUSED=0: the separate kernels.
    static const bool fused_on = !(getenv("OSLAM_SLAM_MPU_FUSED") && atoi(getenv("OSLAM_SLAM_MPU_FUSED")) == 0);
    bool fused = fused_on && (!j->do_desc || keyed);
    if (fused && j->do_desc) for (size_t i = 0; i < P && fused; i++) fused = dstart[i + 1] - dstart[i] <= 128;   // (kDdMaxObs of csrc/mappoint.hip: beyond it k_distinctive reads from memory)
    if (fused) {
        // (the deferred form does not even enqueue the kernel here: it is launched right in front of the next operator's own kernel — the next Fuse round's search —
        // or by mp_update_collect, so that ONE wait covers both and nothing runs on the card while the driver does the round's bookkeeping)
        const bool do_desc = j->do_desc != 0, do_normal = j->do_normal != 0, has_ds = j->desc_start != nullptr;
        uint8_t* const dn = o->dn_h;
        hipEvent_t e0 = o->tev0, e1 = o->tev1;
        std::function<int()> launch = [=]() -> int {
            if (o->timing) (void)hipEventRecord(e0, o->strm);
            const int rcl = oslam_mp_update_fused_device((int)P, do_desc, do_normal, (const int32_t*)(In + oStart), (const int32_t*)(In + (has_ds ? oDStart : oStart)),
                                                         keyed ? (const int32_t*)(In + oRec) : nullptr, (const uint8_t* const*)o->d_rec_desc, (const float*)(In + oOw), (const float*)(In + oPos),
                                                         (const float*)(In + oRef), (const float*)(In + oLsf), o->scale[o->cfg.nLevels - 1], table ? (const int32_t*)(In + oItems) : nullptr,
                                                         table ? o->d_mp_tab : nullptr, (int32_t*)(dn + rBest), dn + rOut, (float*)(dn + rOut5), o->strm);
            if (o->timing) (void)hipEventRecord(e1, o->strm);
            return rcl;
        };
        if (defer) {   // (h_mp_update_collect finishes it)
            o->mpu_pend.on = true; o->mpu_pend.j = j; o->mpu_pend.P = P; o->mpu_pend.rBest = rBest; o->mpu_pend.rOut = rOut; o->mpu_pend.rOut5 = rOut5; o->mpu_pend.dtotal = (double)dtotal;
            o->mpu_pend.launch = launch;
            return OSLAM_OK;
        }
        OPS_CHECK(launch());
        lap_(3);
        OSLAM_HIP_CHECK(oslam::stream_wait(o->strm));
        lap_(4);
        o->t_collect(6, 1, (double)dtotal);
        if (j->do_desc) { memcpy(j->best_idx, o->dn_h + rBest, 4 * P); memcpy(j->out_desc, o->dn_h + rOut, 32 * P); }
        if (j->do_normal) memcpy(j->out5, o->dn_h + rOut5, 20 * P);
        lap_(5);
        return OSLAM_OK;
    }
    o->t_begin();
    if (keyed) OPS_CHECK(oslam_gather_descriptors_device((const uint8_t* const*)o->d_rec_desc, (const int32_t*)(In + oRec), (int)dtotal, Dv + oDesc, o->strm));
    if (j->do_desc) {
        OSLAM_HIP_CHECK(hipMemsetAsync(Dv + oOut, 0, 32 * P, o->strm));
        OPS_CHECK(oslam_mp_distinctive_descriptors_device((int)P, (const int32_t*)(In + (j->desc_start ? oDStart : oStart)), Dv + oDesc, (int32_t*)(Dv + oBest), Dv + oOut, o->strm));
        OSLAM_HIP_CHECK(oslam::copy_to_host_async(o->dn_h + rBest, Dv + oBest, 4 * P, o->strm));
        OSLAM_HIP_CHECK(oslam::copy_to_host_async(o->dn_h + rOut, Dv + oOut, 32 * P, o->strm));
    }
    if (j->do_normal) {
        OPS_CHECK(oslam_mp_update_normal_depth_device((int)P, (const float*)(In + oPos), (const int32_t*)(In + oStart), (const float*)(In + oOw), (const float*)(In + oRef),
                                                      (const float*)(In + oLsf), o->scale[o->cfg.nLevels - 1], (float*)(Dv + oOut5), o->strm));
        OSLAM_HIP_CHECK(oslam::copy_to_host_async(o->dn_h + rOut5, Dv + oOut5, 20 * P, o->strm));
    }
    if (table)
        OPS_CHECK(oslam_mp_table_write_device((int)P, (const int32_t*)(In + oItems), o->d_mp_tab, (const int32_t*)(In + oStart), (const int32_t*)(In + (j->desc_start ? oDStart : oStart)),
                                              (const float*)(In + oPos), (const float*)(Dv + oOut5), Dv + oOut, j->do_desc, j->do_normal, o->strm));
    o->t_end();
    lap_(3);
    OSLAM_HIP_CHECK(oslam::stream_wait(o->strm));
    lap_(4);
    o->t_collect(6, (keyed ? 1 : 0) + (j->do_desc ? 2 : 0) + (j->do_normal ? 1 : 0) + (table ? 1 : 0), (double)dtotal);
    if (j->do_desc) { memcpy(j->best_idx, o->dn_h + rBest, 4 * P); memcpy(j->out_desc, o->dn_h + rOut, 32 * P); }
    if (j->do_normal) memcpy(j->out5, o->dn_h + rOut5, 20 * P);
    lap_(5);
    return OSLAM_OK;
}

// MapPoint::UpdateNormalAndDepth after a local BA, from the solved windows themselves (oslam_job_mp_window_t): the windows' arrays go up as they are (5 bytes per
// edge, ~40 per point — no walk over the map), one launch over all points of all windows, the 20-byte results come back; the resident records are updated in
// the same kernel.
int h_mp_update_windows(void* p, int n, oslam_job_mp_window_t* wins) {
    HipOps* o = (HipOps*)p;
    OSLAM_HIP_CHECK(hipSetDevice(o->cfg.device));
    size_t P = 0, E = 0, K = 0;
    for (int i = 0; i < n; i++) {
        const oslam_job_mp_window_t& w = wins[i];
        if (w.nP < 0 || w.nE < 0 || w.nK < 1 || w.slot < 0 || w.slot >= o->S || (w.nP > 0 && (!w.pt_ids || !w.pt_start || !w.skip || !w.ref_kf || !w.lsf || !w.pos || !w.out5)) ||
            (w.nE > 0 && (!w.edge_kf || !w.erase)) || !w.Ow) { oslam::set_error("mp_update_windows: bad window"); return OSLAM_E_INVALID; }
        P += w.nP; E += w.nE; K += w.nK;
    }
    if (P == 0) return OSLAM_OK;
    const bool table = o->mp_tab_on;
    if (table) {
        for (int i = 0; i < n; i++) {
            int mx = -1;
            for (int j = 0; j < wins[i].nP; j++) mx = std::max(mx, wins[i].pt_ids[j]);
            if (mx >= 0) OPS_CHECK(o->ensure_mp_records(wins[i].slot, (size_t)mx + 1));
        }
        OPS_CHECK(o->sync_mp_table());
    }
    Layout L;
    const size_t oItems = L.take(8 * P), oE0 = L.take(4 * P), oNe = L.take(4 * P), oKb = L.take(4 * P), oRef = L.take(4 * P), oLsf = L.take(4 * P), oSkip = L.take(P), oPos = L.take(12 * P),
                 oEkf = L.take(4 * E), oEr = L.take(E), oOw = L.take(12 * K);
    const size_t in_bytes = L.off;
    const size_t oOut = L.take(20 * P);
    OPS_CHECK(o->ensure_up(L.off));
    uint8_t* U = o->up_h;
    uint8_t* Dv = o->up_d;
    std::vector<size_t> pb(n + 1, 0), eb(n + 1, 0), kb(n + 1, 0);
    for (int i = 0; i < n; i++) { pb[i + 1] = pb[i] + wins[i].nP; eb[i + 1] = eb[i] + wins[i].nE; kb[i + 1] = kb[i] + wins[i].nK; }
    o->pool->parallel_for(n, [&](int i) {
        const oslam_job_mp_window_t& w = wins[i];
        int32_t* it = (int32_t*)(U + oItems) + 2 * pb[i];
        int32_t* e0 = (int32_t*)(U + oE0) + pb[i];
        int32_t* ne = (int32_t*)(U + oNe) + pb[i];
        int32_t* kq = (int32_t*)(U + oKb) + pb[i];
        for (int j = 0; j < w.nP; j++) {
            it[2 * j] = w.slot; it[2 * j + 1] = w.pt_ids[j];
            e0[j] = (int32_t)eb[i] + w.pt_start[j]; ne[j] = w.pt_start[j + 1] - w.pt_start[j]; kq[j] = (int32_t)kb[i];
        }
        memcpy((int32_t*)(U + oRef) + pb[i], w.ref_kf, 4 * (size_t)w.nP);
        memcpy((float*)(U + oLsf) + pb[i], w.lsf, 4 * (size_t)w.nP);
        memcpy(U + oSkip + pb[i], w.skip, (size_t)w.nP);
        memcpy((float*)(U + oPos) + 3 * pb[i], w.pos, 12 * (size_t)w.nP);
        memcpy((int32_t*)(U + oEkf) + eb[i], w.edge_kf, 4 * (size_t)w.nE);
        memcpy(U + oEr + eb[i], w.erase, (size_t)w.nE);
        memcpy((float*)(U + oOw) + 3 * kb[i], w.Ow, 12 * (size_t)w.nK);
    });
    OSLAM_HIP_CHECK(hipMemcpyAsync(Dv, U, in_bytes, hipMemcpyHostToDevice, o->strm));
    o->t_begin();
    OPS_CHECK(oslam_mp_update_windows_device((int)P, (const int32_t*)(Dv + oItems), table ? o->d_mp_tab : nullptr, (const int32_t*)(Dv + oE0), (const int32_t*)(Dv + oNe),
                                             (const int32_t*)(Dv + oKb), (const int32_t*)(Dv + oRef), (const float*)(Dv + oLsf), Dv + oSkip, (const float*)(Dv + oPos),
                                             (const int32_t*)(Dv + oEkf), Dv + oEr, (const float*)(Dv + oOw), o->scale[o->cfg.nLevels - 1], (float*)(Dv + oOut), o->strm));
    o->t_end();
    OPS_CHECK(o->ensure_dn(20 * P));
    OSLAM_HIP_CHECK(oslam::copy_to_host_async(o->dn_h, Dv + oOut, 20 * P, o->strm));
    OSLAM_HIP_CHECK(oslam::stream_wait(o->strm));
    o->t_collect(6, 1, (double)E);
    o->pool->parallel_for(n, [&](int i) { memcpy(wins[i].out5, o->dn_h + 20 * pb[i], 20 * (size_t)wins[i].nP); });
    return OSLAM_OK;
}

// fp64 work of one local BA: per LM trial 700 flop per edge (linearise + accumulate), Schur 324 k_p^2 per point, Cholesky (6K)^3/3,
// back-substitution 2(6K)^2 + 45P; trials = stats[1] + stats[3]
static double lba_flop(const oslam_lba_problem_t& q, const int32_t st[4]) {
    std::vector<int> k(q.nP, 0);
    for (int e = 0; e < q.nE; e++) k[q.edge_pt[e]]++;
    double schur = 0;
    for (int i = 0; i < q.nP; i++) schur += 324.0 * k[i] * k[i];
    int nfree = 0;
    for (int i = 0; i < q.nKF; i++) nfree += q.fixed[i] == 0;
    const double n6 = 6.0 * nfree;
    return (double)(st[1] + st[3]) * (700.0 * q.nE + schur + n6 * n6 * n6 / 3.0 + 2.0 * n6 * n6 + 45.0 * q.nP);
}

// ---- local-BA service (deferred schedule, include/oslam_slam.h: lba_submit / lba_wait) ----
// The handles of a process step independently, each on its own host thread and stream.  In the synchronous schedule every handle runs its own local-BA call
// (~40-80 windows) on its own stream: eight such calls interleave ~180 dependent launches each on one card and every launch runs several times longer than
// alone.  In the deferred schedule a handle only SUBMITS its windows and tracks the next frame; ONE service thread per device takes whatever the handles have
// submitted meanwhile (same camera), solves it as ONE batch on its own stream and wakes the submitters.  A window's result does not depend on the batch it is
// solved in (tests/test_lba_gpu.py), so the schedule of the service does not enter the results.
struct LbaService {
    typedef LbaJob Job;
    int device = 0;
    // OSLAM_LBA_SERVICE_THREADS workers (default 2), each with its own solver handle and stream: while one has the device the other prepares the next batch
    // and scatters the previous one's results (window preparation, upload and result scatter are ~a third of a call's wall time).  The device part of a call
    // (upload .. download) is taken in turns (launch_mu): two overlapping calls ran 7.1 s of stream time per 20 bench steps against 5.3 s in turns, at the
    // same frames/s (same-box A/B: 26.6 k overlapping, 26.3-27.1 k in turns)
    struct Worker { oslam_lba_t* ba = nullptr; hipStream_t strm = nullptr; std::thread th; };
    std::vector<std::unique_ptr<Worker>> workers;
    int max_batch = 0, mode_small = 1, mode_big = 1, big_from = 1 << 30;
    std::mutex launch_mu;   // one call on the device at a time: the other worker prepares / scatters meanwhile (OSLAM_LBA_SERVICE_OVERLAP=1 lets the calls overlap)
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::deque<Job*> queue;
    bool stop = false;
    int users = 0;
    long long calls = 0, windows = 0, max_windows = 0;

    static std::mutex& reg_mu() { static std::mutex m; return m; }
    static LbaService*& slot(int device) { static LbaService* s[64] = {nullptr}; return s[device & 63]; }

    static LbaService* acquire(int device) {
        std::lock_guard<std::mutex> lk(reg_mu());
        LbaService*& sv = slot(device);
        if (!sv) {
            LbaService* n = new LbaService;
            n->device = device;
            if (n->start()) { delete n; return nullptr; }
            sv = n;
        }
        sv->users++;
        return sv;
    }
    static void release(LbaService* sv) {
        if (!sv) return;
        std::lock_guard<std::mutex> lk(reg_mu());
        if (--sv->users > 0) return;
        { std::lock_guard<std::mutex> l2(sv->mu); sv->stop = true; }
        sv->cv_work.notify_all();
        for (auto& wk : sv->workers) if (wk->th.joinable()) wk->th.join();
        if (getenv("OSLAM_LBA_SERVICE_STATS")) fprintf(stderr, "[lba service] %lld calls, %lld windows (%.1f per call, max %lld)\n", sv->calls, sv->windows, sv->calls ? (double)sv->windows / sv->calls : 0.0, sv->max_windows);
        (void)hipSetDevice(sv->device);
        for (auto& wk : sv->workers) { oslam_lba_destroy(wk->ba); if (wk->strm) (void)hipStreamDestroy(wk->strm); }
        slot(sv->device) = nullptr;
        delete sv;
    }
    int start() {
        OSLAM_HIP_CHECK(hipSetDevice(device));
        max_batch = getenv("OSLAM_LBA_SERVICE_MAX_BATCH") ? atoi(getenv("OSLAM_LBA_SERVICE_MAX_BATCH")) : 4096;
        const int nthreads = std::max(1, getenv("OSLAM_LBA_SERVICE_THREADS") ? atoi(getenv("OSLAM_LBA_SERVICE_THREADS")) : 2);
        // OSLAM_LBA_SERVICE_CUS=k: the service's streams may only use k of the card's CUs (the tracking kernels of the handles keep the others to themselves)
        const int cus = getenv("OSLAM_LBA_SERVICE_CUS") ? atoi(getenv("OSLAM_LBA_SERVICE_CUS")) : 0;
        for (int t = 0; t < nthreads; t++) {
            std::unique_ptr<Worker> wk(new Worker);
            const int rc = oslam_lba_create(&wk->ba, max_batch, 1 << 16, 4096, 32768, device);
            if (rc) return rc;
            if (cus > 0) {
                hipDeviceProp_t pr;
                OSLAM_HIP_CHECK(hipGetDeviceProperties(&pr, device));
                const int total = pr.multiProcessorCount;
                std::vector<uint32_t> mask((total + 31) / 32, 0u);
                for (int i = 0; i < std::min(cus, total); i++) mask[i >> 5] |= 1u << (i & 31);
                OSLAM_HIP_CHECK(hipExtStreamCreateWithCUMask(&wk->strm, (uint32_t)mask.size(), mask.data()));
                oslam::lba_use_stream(wk->ba, wk->strm);
            } else if (const char* pe = getenv("OSLAM_LBA_SERVICE_PRIORITY")) {
                // Rounds 4-5 gave the service's streams the LOWEST priority (the handles' short tracking / mapping kernels dispatched ahead of the queued local-BA
                // launches).  Since the end of round 5 they are ordinary streams (the solver handle's own): same frames/s (41.10 / 41.17 k against 41.13 k, same box,
                // alternating), 6.5 % less local-BA device time (roofline.frac 0.056 against 0.052), and no stream of the rank left that the other streams' work can
                // starve ("A starved side stream").  OSLAM_LBA_SERVICE_PRIORITY=low / high are the A/B knobs (section 7.2).
                if (!strcmp(pe, "low") || !strcmp(pe, "high")) {
                    int lo = 0, hi = 0;
                    OSLAM_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));   // (lo = numerically greatest = lowest priority)
                    OSLAM_HIP_CHECK(hipStreamCreateWithPriority(&wk->strm, hipStreamNonBlocking, !strcmp(pe, "high") ? hi : lo));
                    oslam::lba_use_stream(wk->ba, wk->strm);
                }
            }
            if (!getenv("OSLAM_LBA_SERVICE_OVERLAP")) oslam::lba_use_gate(wk->ba, &launch_mu);
            workers.push_back(std::move(wk));
        }
        // layout per call: batches of at least `big_from` windows go through the one-workgroup-per-window kernel (mode 2), smaller ones through the multi-launch layout
        mode_small = getenv("OSLAM_LBA_SERVICE_MODE") ? atoi(getenv("OSLAM_LBA_SERVICE_MODE")) : 1;
        mode_big = getenv("OSLAM_LBA_SERVICE_MODE_BIG") ? atoi(getenv("OSLAM_LBA_SERVICE_MODE_BIG")) : mode_small;
        big_from = getenv("OSLAM_LBA_SERVICE_BIG_FROM") ? atoi(getenv("OSLAM_LBA_SERVICE_BIG_FROM")) : (1 << 30);
        for (auto& wk : workers) { Worker* w = wk.get(); w->th = std::thread([this, w] { run(*w); }); }
        return OSLAM_OK;
    }
    void submit(Job* j) {
        { std::lock_guard<std::mutex> lk(mu); j->done = false; queue.push_back(j); }
        cv_work.notify_one();
    }
    void wait(Job* j) {
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return j->done; });
    }
    void run(Worker& wk) {
        (void)hipSetDevice(device);
        oslam::stream_wait_thread_mode(getenv("OSLAM_LBA_SERVICE_SPIN_US") ? atoi(getenv("OSLAM_LBA_SERVICE_SPIN_US")) : 0);   // sleep, do not spin: off the critical path
        oslam_lba_t* ba = wk.ba;
        std::vector<Job*> take;
        std::vector<oslam_lba_problem_t> probs;
        std::vector<int32_t> st;
        for (;;) {
            take.clear();
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || !queue.empty(); });
                if (queue.empty()) return;   // (stop with nothing left)
                // everything submitted so far with the camera of the oldest job, up to the batch capacity
                Job* first = queue.front();
                int total = 0;
                for (auto it = queue.begin(); it != queue.end();) {
                    Job* j = *it;
                    if (memcmp(j->K5, first->K5, sizeof(first->K5)) == 0 && (take.empty() || total + j->n <= max_batch)) { take.push_back(j); total += j->n; it = queue.erase(it); }
                    else ++it;
                }
            }
            probs.clear();
            bool timing = false;
            for (Job* j : take) { probs.insert(probs.end(), j->probs, j->probs + j->n); timing = timing || j->timing; }
            const int n = (int)probs.size();
            st.assign((size_t)4 * n, 0);
            std::vector<int32_t*> caller_stats(n);
            for (int i = 0; i < n; i++) { caller_stats[i] = probs[i].stats; probs[i].stats = &st[4 * (size_t)i]; }
            int rc = OSLAM_OK;
            double ms = 0; long long launches = 0;
            if (n > 0) {
                rc = oslam_lba_set_mode(ba, n >= big_from ? mode_big : mode_small);
                if (!rc) rc = oslam_lba_kernel_time(ba, timing ? 1 : 0, nullptr, nullptr);
                for (int at = 0; !rc && at < n; at += max_batch) rc = oslam_lba_optimize_batch(ba, std::min(max_batch, n - at), probs.data() + at, take[0]->K5);
                if (!rc && timing) rc = oslam_lba_kernel_time(ba, 0, &ms, &launches);
            }
            // A window the solver refused (stats[0] < 0, oslam_lba_optimize_batch) fails alone: its submitter sees it in the stats array it passed, or — when it
            // passed none — as the error of ITS job; the other jobs of the batch are complete.
            std::vector<int> job_rc(take.size(), 0);
            if (!rc) {
                size_t at = 0;
                for (size_t q = 0; q < take.size(); q++)
                    for (int i = 0; i < take[q]->n; i++, at++) {
                        if (caller_stats[at]) memcpy(caller_stats[at], probs[at].stats, 16);
                        else if (probs[at].stats[0] < 0 && !job_rc[q]) job_rc[q] = probs[at].stats[1];
                    }
            }
            // the call's kernel time and launches are shared out by the windows' flop / count (the sums over the handles are the call's)
            std::vector<double> fl(take.size(), 0.0);
            double fl_all = 0;
            if (!rc && timing) {
                size_t at = 0;
                for (size_t q = 0; q < take.size(); q++) {
                    for (int i = 0; i < take[q]->n; i++, at++) {
                        const oslam_lba_problem_t& w = probs[at];
                        if (w.stats[0] >= 0) fl[q] += lba_flop(w, w.stats);
                    }
                    fl_all += fl[q];
                }
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                calls++; windows += n; max_windows = std::max<long long>(max_windows, n);
                for (size_t q = 0; q < take.size(); q++) {
                    Job* j = take[q];
                    j->rc = rc ? rc : job_rc[q];
                    if (rc) snprintf(j->err, sizeof(j->err), "%s", oslam_last_error());
                    else if (job_rc[q]) snprintf(j->err, sizeof(j->err), "a window of the submission was refused by the solver (pass stats to learn which)");
                    j->flop = fl[q]; j->ms = fl_all > 0 ? ms * fl[q] / fl_all : 0; j->launches = n > 0 ? (double)launches * j->n / n : 0;
                    j->done = true;
                }
            }
            cv_done.notify_all();
        }
    }
};

int h_lba_submit(void* p, int n, const oslam_lba_problem_t* pr);
int h_lba_wait(void* p);
int h_lba(void* p, int n, const oslam_lba_problem_t* pr) {
    HipOps* o = (HipOps*)p;
    OSLAM_HIP_CHECK(hipSetDevice(o->cfg.device));   // the HIP current device is per host thread: a handle may be stepped from any thread
    if (o->svc && n > 1) {   // Also the synchronous schedule solves through the process-wide service: the windows of the handles that are at this point together
                             // form shared batches, one call on the device at a time, and this handle SLEEPS meanwhile.  On its own stream a handle's call ran
                             // against the other handles' calls (6.2 s of device time per 20 steps of the headline against 4.3 s) and its thread spun through
                             // the wait (73 against 58 busy core-seconds): 31.8 k -> 35.4-36.3 k frames/s, same box.  OSLAM_LBA_SYNC_OWN_STREAM=1: as before.
        const int rc = h_lba_submit(p, n, pr);
        return rc ? rc : h_lba_wait(p);
    }
    std::vector<oslam_lba_problem_t> tp;
    std::vector<int32_t> st;
    const oslam_lba_problem_t* caller = pr;
    if (o->timing) {
        tp.assign(pr, pr + n); st.assign((size_t)4 * n, 0);
        for (int i = 0; i < n; i++) tp[i].stats = &st[4 * (size_t)i];
        pr = tp.data();
    }
    oslam_lba_t* ba = n == 1 ? o->ba1 : o->ba;
    int rc;
    if (n == 1) {   // one window: spread over the whole GPU
        rc = oslam_lba_optimize(o->ba1, pr[0].nKF, pr[0].poses, pr[0].fixed, pr[0].nP, pr[0].points, pr[0].nE, pr[0].edge_kf, pr[0].edge_pt, pr[0].edge_obs,
                                pr[0].edge_invSigma2, o->K5, 0, pr[0].poses_out, pr[0].points_out, pr[0].erase, pr[0].stats);
        if ((rc == OSLAM_E_CAPACITY || rc == OSLAM_E_INVALID) && caller[0].stats) {   // the window was refused: it fails alone, like a window of a batch (oslam_lba_optimize_batch)
            const oslam_lba_problem_t& q = pr[0];
            if (q.poses_out && q.poses && q.nKF > 0) memcpy(q.poses_out, q.poses, (size_t)q.nKF * 64);
            if (q.points_out && q.points && q.nP > 0) memcpy(q.points_out, q.points, (size_t)q.nP * 12);
            if (q.erase && q.nE > 0) memset(q.erase, 0, (size_t)q.nE);
            q.stats[0] = -1; q.stats[1] = rc; q.stats[2] = q.stats[3] = 0;
            rc = OSLAM_OK;
        }
    } else rc = oslam_lba_optimize_batch(o->ba, n, pr, o->K5);   // one workgroup per window, one launch
    if (!rc && pr != caller)
        for (int i = 0; i < n; i++) if (caller[i].stats) memcpy(caller[i].stats, pr[i].stats, 16);
    if (!rc && o->timing) {
        double ms = 0; long long launches = 0;
        OPS_CHECK(oslam_lba_kernel_time(ba, 1, &ms, &launches));
        std::vector<double> fl(n, 0.0);   // (instrumentation inside the timed region of bench.py: on the shared workers, not serially on the stepping thread)
        o->pool->parallel_for(n, [&](int i) { fl[i] = pr[i].stats[0] >= 0 ? lba_flop(pr[i], pr[i].stats) : 0.0; });
        double flop = 0;
        for (int i = 0; i < n; i++) flop += fl[i];
        o->kt[6] += ms; o->kt[7] += (double)launches; o->kt[8] += flop;
    }
    return rc;
}

static void lba_service_release(LbaService* s) { LbaService::release(s); }

int h_lba_submit(void* p, int n, const oslam_lba_problem_t* pr) {
    HipOps* o = (HipOps*)p;
    if (!o->svc) { oslam::set_error("lba_submit: no local-BA service"); return OSLAM_E_INVALID; }
    if (o->job_active) { oslam::set_error("lba_submit: a submission is already in flight"); return OSLAM_E_INVALID; }
    LbaService::Job& j = o->job;
    j = LbaService::Job();
    j.n = n; j.probs = pr; memcpy(j.K5, o->K5, sizeof(j.K5)); j.timing = o->timing != 0;
    o->job_active = true;
    o->svc->submit(&j);
    return OSLAM_OK;
}

int h_lba_wait(void* p) {
    HipOps* o = (HipOps*)p;
    if (!o->job_active) return OSLAM_OK;
    o->svc->wait(&o->job);
    o->job_active = false;
    const LbaService::Job& j = o->job;
    if (j.rc) { oslam::set_error("local-BA service: %s", j.err); return j.rc; }
    if (o->timing) { o->kt[6] += j.ms; o->kt[7] += j.launches; o->kt[8] += j.flop; }
    return OSLAM_OK;
}

// ---- resident keyframes ----

int h_register_keyframes(void* p, int n, const int32_t* slots, const int32_t* kf_ids) {
    HipOps* o = (HipOps*)p;
    OSLAM_HIP_CHECK(hipSetDevice(o->cfg.device));
    if (n == 0) return OSLAM_OK;
    if ((int)o->rec_of_kf.size() < o->S) o->rec_of_kf.resize(o->S);
    const size_t cap = o->cap, rb = o->rec_bytes();
    std::vector<CopySegH> segs;
    for (int i = 0; i < n; i++) {
        const int slot = slots[i], kf = kf_ids[i];
        if (slot < 0 || slot >= o->S || kf < 0) { oslam::set_error("register_keyframes: bad slot / id"); return OSLAM_E_INVALID; }
        if (kf == 0) {   // the sequence was reset: its keyframe ids restart, its records are free again
            for (int r0 : o->rec_of_kf[slot]) if (r0 >= 0) o->free_recs.push_back(r0);
            o->rec_of_kf[slot].clear();
        }
        if ((int)o->rec_of_kf[slot].size() <= kf) o->rec_of_kf[slot].resize(kf + 1, -1);
        int r;
        if (!o->free_recs.empty()) { r = o->free_recs.back(); o->free_recs.pop_back(); }   // a record of a map that was reset
        else {
            r = o->n_rec++;
            if (r / HipOps::kRecChunk >= (int)o->rec_chunks.size()) {
                uint8_t* c = nullptr;
                OSLAM_HIP_CHECK(hipMalloc((void**)&c, rb * HipOps::kRecChunk));
                o->rec_chunks.push_back(c);
            }
            o->h_rec_desc.push_back((uint8_t*)o->rec_desc(r));
        }
        o->rec_of_kf[slot][kf] = r;
        // the frame built for `slot` in this step is still in the batch arrays
        segs.push_back({(const uint8_t*)(o->d_keysUn + cap * slot), (uint8_t*)o->rec_keys(r), (uint32_t)(cap * sizeof(oslam_keypoint_t)), 0});
        segs.push_back({o->d_desc + 32 * cap * slot, (uint8_t*)o->rec_desc(r), (uint32_t)(cap * 32), 0});
        segs.push_back({(const uint8_t*)(o->cur_uRight + cap * slot), (uint8_t*)o->rec_ur(r), (uint32_t)(cap * 4), 0});
    }
    if (o->lazy_keys && (!o->d_kp || !o->d_desc)) { oslam::set_error("register_keyframes: no frame has been built yet"); return OSLAM_E_INVALID; }
    if (o->lazy_keys) {   // mvKeys of the new keyframes: written by the same launch into a pinned block the device can address (keyframe_raw_keys hands them out)
        const size_t need = (size_t)n * cap * (sizeof(oslam_keypoint_t) + (o->lazy_desc ? 32 : 0));
        if (need > o->kfk_cap) {
            OSLAM_HIP_CHECK(oslam::stream_wait(o->strm));
            if (o->kfk_h) (void)hipHostFree(o->kfk_h);
    if (o->fc_d) (void)hipFree(o->fc_d);
            o->kfk_h = nullptr; o->kfk_cap = 0;
            OSLAM_HIP_CHECK(hipHostMalloc((void**)&o->kfk_h, need + need / 2 + 4096, 0));
            o->kfk_cap = need + need / 2 + 4096;
        }
        o->kfk_slots.assign(slots, slots + n);
        for (int i = 0; i < n; i++)
            segs.push_back({(const uint8_t*)(o->d_kp + cap * slots[i]), o->kfk_h + (size_t)i * cap * sizeof(oslam_keypoint_t), (uint32_t)(cap * sizeof(oslam_keypoint_t)), 0});
        if (o->lazy_desc)
            for (int i = 0; i < n; i++)
                segs.push_back({o->d_desc + 32 * cap * slots[i], o->kfk_h + (size_t)n * cap * sizeof(oslam_keypoint_t) + (size_t)i * cap * 32, (uint32_t)(cap * 32), 0});
    }
    const size_t oJobs = oslam::align_up(segs.size() * sizeof(CopySegH), 256), up_bytes = oJobs + (size_t)n * sizeof(oslam_kf_grid_job_t);
    OPS_CHECK(o->ensure_up(up_bytes));
    memcpy(o->up_h, segs.data(), segs.size() * sizeof(CopySegH));
    oslam_kf_grid_job_t* gj = (oslam_kf_grid_job_t*)(o->up_h + oJobs);
    for (int i = 0; i < n; i++) {
        const int r = o->rec_of_kf[slots[i]][kf_ids[i]];
        gj[i].keys = o->rec_keys(r); gj[i].uRight = o->rec_ur(r); gj[i].cell_end = o->rec_cell_end(r); gj[i].cand = o->rec_cand(r); gj[i].slot = slots[i]; gj[i].pad_ = 0;
    }
    OSLAM_HIP_CHECK(hipMemcpyAsync(o->up_d, o->up_h, up_bytes, hipMemcpyHostToDevice, o->strm));
    o->t_begin();
    OPS_CHECK(oslam_copy_segments_device(o->up_d, (int)segs.size(), o->strm));
    // KeyFrame::mGrid of the new keyframes (fixed from here on): sorted once, read by every later Fuse against them
    OPS_CHECK(oslam_kf_grid_build_device(n, (const oslam_kf_grid_job_t*)(o->up_d + oJobs), o->d_cnt, o->bounds, (int)cap, o->d_status + 8, o->strm));
    o->t_end();
    // descriptor-array table for the observation gathers
    if ((size_t)o->n_rec > o->rec_desc_cap) {
        OSLAM_HIP_CHECK(oslam::stream_wait(o->strm));
        if (o->d_rec_desc) (void)hipFree(o->d_rec_desc);
        o->rec_desc_cap = (size_t)o->n_rec * 2 + 1024;
        OSLAM_HIP_CHECK(hipMalloc((void**)&o->d_rec_desc, o->rec_desc_cap * sizeof(uint8_t*)));
        o->rec_desc_n = 0;
    }
    OSLAM_HIP_CHECK(hipMemcpyAsync(o->d_rec_desc + o->rec_desc_n, o->h_rec_desc.data() + o->rec_desc_n, (size_t)(o->n_rec - o->rec_desc_n) * sizeof(uint8_t*),
                                   hipMemcpyHostToDevice, o->strm));
    o->rec_desc_n = o->n_rec;
    OPS_CHECK(o->ensure_dn(4));
    OSLAM_HIP_CHECK(oslam::copy_to_host_async(o->dn_h, o->d_status + 8, 4, o->strm));
    OSLAM_HIP_CHECK(oslam::stream_wait(o->strm));   // the records are complete when this returns
    o->t_collect(7, 2, 0);
    if (*(const int32_t*)o->dn_h != 0) { oslam::set_error("register_keyframes: a keyframe has more keypoints than the extractor's capacity"); return OSLAM_E_CAPACITY; }
    return OSLAM_OK;
}

// the records of culled keyframes go back to the free list (every operator synchronises before it returns: nothing in flight reads them)
int h_release_keyframes(void* p, int n, const int32_t* slots, const int32_t* kf_ids) {
    HipOps* o = (HipOps*)p;
    for (int i = 0; i < n; i++) {
        const int slot = slots[i], kf = kf_ids[i];
        if (slot < 0 || slot >= (int)o->rec_of_kf.size() || kf < 0 || kf >= (int)o->rec_of_kf[slot].size()) continue;
        const int r = o->rec_of_kf[slot][kf];
        if (r < 0) continue;
        o->rec_of_kf[slot][kf] = -1;
        o->free_recs.push_back(r);
        o->n_released++;
    }
    return OSLAM_OK;
}

// ---- device mirror of the observation graph (round 5, include/oslam_slam.h oslam_slam_ops_t::map_journal / kf_culling_counts) ----
// Per keyframe record: mp[cap] (KeyFrame::mvpMapPoints), okf[cap] (the point that holds the observation (kf, i): MapPoint::mObservations seen from the keyframe's
// side; the LAST AddObservation wins when two points claim one keypoint) and good[(cap + 31) / 32] (usable-depth bits).  Per slot: 16 bytes per point id
// (Observations(), isBad(), octave histogram).  The host translates keyframe ids to record indices while it copies a journal into the upload block; one wavefront
// per sequence applies its records in program order (lane 0; the bulk record of a new keyframe by all lanes).
struct MirrorGeom { uint64_t rec_bytes, core_bytes, okf_off, good_off; int32_t cap, chunk; };
struct MirrorBulk { int32_t r, N; uint32_t word_off, seq; };   // a new keyframe: p[N] then good[ceil(N/32)] at word_off of the bulk payload
__device__ __forceinline__ int32_t* mirror_mp(uint8_t* const* chunks, const MirrorGeom& g, int r) { return (int32_t*)(chunks[r / g.chunk] + (size_t)(r % g.chunk) * g.rec_bytes + g.core_bytes); }
// one workgroup per new keyframe: its point list, its usable-depth bits, and every okf cell back to "nobody" at the bulk's event number
__global__ __launch_bounds__(256) void k_mirror_bulk(const MirrorBulk* bulks, const uint32_t* payload, uint8_t* const* chunks, MirrorGeom g) {
    const MirrorBulk b = bulks[blockIdx.x];
    if (b.r < 0) return;
    int32_t* mp = mirror_mp(chunks, g, b.r);
    unsigned long long* okf = (unsigned long long*)((uint8_t*)mp + g.okf_off);
    uint32_t* good = (uint32_t*)((uint8_t*)mp + g.good_off);
    const uint32_t* w = payload + b.word_off;
    for (int i = threadIdx.x; i < g.cap; i += 256) { mp[i] = i < b.N ? (int32_t)w[i] : -1; okf[i] = ((unsigned long long)b.seq << 32) | 0xFFFFFFFFull; }
    for (int i = threadIdx.x; i < (g.cap + 31) / 32; i += 256) good[i] = i < (b.N + 31) / 32 ? w[b.N + i] : 0u;
}
// one thread per record: dirty cells of the point lists (current values), observation events (the event with the largest number stays in a cell: 64-bit
// max of number << 32 | point), dirty points (current scalars)
__global__ __launch_bounds__(256) void k_mirror_ops(const int4* cells, int ncell, const uint4* okfs, int nokf, const uint4* pts, int npt, uint8_t* const* chunks, uint8_t* const* pt_aux,
                                                    MirrorGeom g) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < ncell) {
        const int4 c = cells[i];
        if (c.x >= 0 && c.y >= 0 && c.y < g.cap) mirror_mp(chunks, g, c.x)[c.y] = c.z;
    } else if (i < ncell + nokf) {
        const uint4 e = okfs[i - ncell];   // (record, idx, point, event number)
        const int r = (int)e.x, idx = (int)e.y;
        if (r >= 0 && idx >= 0 && idx < g.cap) {
            unsigned long long* okf = (unsigned long long*)((uint8_t*)mirror_mp(chunks, g, r) + g.okf_off);
            atomicMax(okf + idx, ((unsigned long long)e.w << 32) | (unsigned long long)e.z);
        }
    } else if (i < ncell + nokf + npt) {
        const uint4 a = pts[2 * (i - ncell - nokf)], b = pts[2 * (i - ncell - nokf) + 1];   // (slot, p, nObs, bad), (lvl lo, lvl hi, -, -)
        ((uint4*)pt_aux[a.x])[2 * (size_t)a.y] = make_uint4(a.z, a.w, b.x, b.y);   // (32-byte records: the marks in the second half are not touched)
    }
}

// LocalMapping::KeyFrameCulling's counting loop (src/LocalMapping.cc:649-690) for one candidate keyframe per wavefront, from the mirror: out = (slots with a point
// at a usable depth, nMPs, nRedundantObservations, ambiguous).  "At least three OTHER observations at octave <= level + 1" from the point's octave histogram: four or
// more qualifying observations -> yes, two or fewer -> no; exactly three -> yes iff this keyframe's own observation is not one of them — it is when okf[i] == p
// (its octave IS the level); otherwise the mirror cannot tell (a displaced claim) and the host recounts the keyframe.
struct CullCand { int32_t rec, slot; };
__global__ __launch_bounds__(64) void k_cull_counts(const CullCand* cands, uint8_t* const* chunks, uint8_t* const* pt_aux, MirrorGeom g, int32_t* out) {
    const CullCand cd = cands[blockIdx.x];
    const int lane = threadIdx.x;
    int ub = 0, nMPs = 0, nRed = 0, amb = 0;
    if (cd.rec >= 0) {
        const uint8_t* rec = chunks[cd.rec / g.chunk] + (size_t)(cd.rec % g.chunk) * g.rec_bytes;
        const oslam_keypoint_t* keys = (const oslam_keypoint_t*)rec;
        const int32_t* mp = (const int32_t*)(rec + g.core_bytes);
        const unsigned long long* okf = (const unsigned long long*)((const uint8_t*)mp + g.okf_off);
        const uint32_t* good = (const uint32_t*)((const uint8_t*)mp + g.good_off);
        const uint4* aux = (const uint4*)pt_aux[cd.slot];
        for (int i = lane; i < g.cap; i += 64) {
            const int p = mp[i];
            if (p < 0) continue;
            if (!((good[i >> 5] >> (i & 31)) & 1u)) continue;
            ub++;
            const uint4 a = aux[2 * (size_t)p];   // (nObs, bad, lvl lo, lvl hi)
            if (a.y != 0u) continue;
            nMPs++;
            if ((int)a.x > 3) {
                const int lvl = keys[i].octave + 1;
                const unsigned long long hgram = ((unsigned long long)a.w << 32) | a.z;
                const unsigned long long hh = lvl >= 7 ? hgram : (hgram & ((1ull << (8 * (lvl + 1))) - 1ull));
                unsigned long long s2 = (hh & 0x00FF00FF00FF00FFull) + ((hh >> 8) & 0x00FF00FF00FF00FFull);
                s2 = (s2 & 0x0000FFFF0000FFFFull) + ((s2 >> 16) & 0x0000FFFF0000FFFFull);
                const int all_le = (int)((s2 & 0xFFFFFFFFull) + (s2 >> 32));
                if (all_le >= 4) nRed++;
                else if (all_le == 3 && (int)(uint32_t)(okf[i] & 0xFFFFFFFFull) != p) amb++;
            }
        }
    } else amb = 1;   // (the keyframe has no resident record: the host counts)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { ub += __shfl_xor(ub, d, 64); nMPs += __shfl_xor(nMPs, d, 64); nRed += __shfl_xor(nRed, d, 64); amb += __shfl_xor(amb, d, 64); }
    if (lane == 0) { out[4 * blockIdx.x] = ub; out[4 * blockIdx.x + 1] = nMPs; out[4 * blockIdx.x + 2] = nRed; out[4 * blockIdx.x + 3] = amb; }
}

static MirrorGeom mirror_geom(const HipOps* o) {
    MirrorGeom g;
    g.rec_bytes = o->rec_bytes(); g.core_bytes = o->rec_core_bytes(); g.okf_off = oslam::align_up((size_t)o->cap * 4, 256); g.good_off = g.okf_off + oslam::align_up((size_t)o->cap * 8, 256);
    g.cap = o->cap; g.chunk = HipOps::kRecChunk;
    return g;
}

int h_map_journal(void* p, int n, const oslam_map_changes_t* ch) {
    HipOps* o = (HipOps*)p;
    OSLAM_HIP_CHECK(hipSetDevice(o->cfg.device));
    if (n <= 0) return OSLAM_OK;
    if ((int)o->pt_aux.size() < o->S) { o->pt_aux.resize(o->S, nullptr); o->pt_aux_cap.resize(o->S, 0); }
    if ((int)o->okf_seq.size() < o->S) o->okf_seq.resize(o->S, 0u);
    if (o->mir_used) { OSLAM_HIP_CHECK(oslam::stream_wait(o->strm)); o->mir_used = 0; }   // (a flush without a collected count request before it: its block must have landed)
    size_t nbulk = 0, ncell = 0, nokf = 0, npt = 0, bw = 0;
    std::vector<size_t> oB(n), oC(n), oO(n), oP(n), oW(n);
    std::vector<long long> maxp(n, -1);
    for (int i = 0; i < n; i++) {
        const oslam_map_changes_t& c = ch[i];
        if (c.slot < 0 || c.slot >= o->S || c.n_new < 0 || c.n_cells < 0 || c.n_events < 0 || c.n_points < 0) { oslam::set_error("map_journal: bad change set"); return OSLAM_E_INVALID; }
        oB[i] = nbulk; oC[i] = ncell; oO[i] = nokf; oP[i] = npt; oW[i] = bw;
        nbulk += c.n_new; ncell += c.n_cells; nokf += c.n_events; npt += c.n_points;
        for (int k = 0; k < c.n_new; k++) { if (c.new_kfs[k].N < 0 || c.new_kfs[k].N > o->cap) { oslam::set_error("map_journal: keyframe with more keypoints than the capacity"); return OSLAM_E_CAPACITY; } bw += (size_t)c.new_kfs[k].N + ((size_t)c.new_kfs[k].N + 31) / 32; }
    }
    // the largest point id named anywhere: the per-point records of a slot cover it before anything on the device indexes them
    o->pool->parallel_for(n, [&](int i) {
        const oslam_map_changes_t& c = ch[i];
        long long m = -1;
        for (int k = 0; k < c.n_new; k++) for (int q = 0; q < c.new_kfs[k].N; q++) m = std::max<long long>(m, c.new_kfs[k].mp[q]);
        for (int q = 0; q < c.n_cells; q++) m = std::max<long long>(m, c.cells[3 * q + 2]);
        for (int q = 0; q < c.n_points; q++) m = std::max<long long>(m, (long long)c.points[5 * q]);
        maxp[i] = m;
    });
    for (int i = 0; i < n; i++) if (maxp[i] >= 0) OPS_CHECK(o->ensure_pt_aux(ch[i].slot, (size_t)maxp[i] + 1));
    Layout L;
    const size_t aB = L.take(sizeof(MirrorBulk) * nbulk), aW = L.take(4 * bw), aC = L.take(16 * ncell), aO = L.take(16 * nokf), aP = L.take(32 * npt);
    OPS_CHECK(o->ensure_mir(L.off + (1 << 20)));   // (+ room for the count request that usually follows)
    OPS_CHECK(o->sync_mirror_tables());
    // event numbers: the new keyframes of a flush take the slot's counter, its observation events counter + 1 + position
    std::vector<uint32_t> seq0(n);
    for (int i = 0; i < n; i++) { seq0[i] = o->okf_seq[ch[i].slot]; o->okf_seq[ch[i].slot] += 1u + (uint32_t)ch[i].n_events; }
    uint8_t* U = o->mir_h;
    // keyframe id -> record index (a keyframe whose record was released — culled — or never registered: -1, skipped on the device)
    o->pool->parallel_for(n, [&](int i) {
        const oslam_map_changes_t& c = ch[i];
        const int slot = c.slot;
        MirrorBulk* B = (MirrorBulk*)(U + aB) + oB[i];
        uint32_t* PW = (uint32_t*)(U + aW);
        int4* Cc = (int4*)(U + aC) + oC[i];
        uint4* Oo = (uint4*)(U + aO) + oO[i];
        uint4* Pp = (uint4*)(U + aP) + 2 * oP[i];
        size_t wo = oW[i];
        for (int k = 0; k < c.n_new; k++) {
            const oslam_map_new_kf_t& e = c.new_kfs[k];
            B[k].r = o->rec_lookup(slot, e.kf); B[k].N = e.N; B[k].word_off = (uint32_t)wo; B[k].seq = seq0[i];
            memcpy(PW + wo, e.mp, 4 * (size_t)e.N);
            memcpy(PW + wo + e.N, e.good, 4 * (((size_t)e.N + 31) / 32));
            wo += (size_t)e.N + ((size_t)e.N + 31) / 32;
        }
        for (int q = 0; q < c.n_cells; q++) Cc[q] = make_int4(o->rec_lookup(slot, c.cells[3 * q]), c.cells[3 * q + 1], c.cells[3 * q + 2], 0);
        uint32_t ev = seq0[i] + 1u;
        for (int q = 0; q < c.n_events; q++) {
            const uint32_t w1 = c.events[3 * q + 1];
            Oo[q] = make_uint4((uint32_t)o->rec_lookup(slot, (int)c.events[3 * q]), w1 & 0x7FFFFFFFu, (w1 & 0x80000000u) ? c.events[3 * q + 2] : 0xFFFFFFFFu, ev++);
        }
        for (int q = 0; q < c.n_points; q++) {
            Pp[2 * q] = make_uint4((uint32_t)slot, c.points[5 * q], c.points[5 * q + 1], c.points[5 * q + 2]);
            Pp[2 * q + 1] = make_uint4(c.points[5 * q + 3], c.points[5 * q + 4], 0u, 0u);
        }
    });
    OSLAM_HIP_CHECK(hipMemcpyAsync(o->mir_d, o->mir_h, L.off, hipMemcpyHostToDevice, o->strm));
    o->mir_used = L.off;
    const MirrorGeom g = mirror_geom(o);
    if (nbulk) hipLaunchKernelGGL(k_mirror_bulk, dim3((unsigned)nbulk), dim3(256), 0, o->strm, (const MirrorBulk*)(o->mir_d + aB), (const uint32_t*)(o->mir_d + aW), (uint8_t* const*)o->d_rec_chunk, g);
    const size_t nops = ncell + nokf + npt;
    if (nops) hipLaunchKernelGGL(k_mirror_ops, dim3((unsigned)((nops + 255) / 256)), dim3(256), 0, o->strm, (const int4*)(o->mir_d + aC), (int)ncell, (const uint4*)(o->mir_d + aO), (int)nokf,
                                 (const uint4*)(o->mir_d + aP), (int)npt, (uint8_t* const*)o->d_rec_chunk, (uint8_t* const*)o->d_pt_aux, g);
    OSLAM_HIP_CHECK(hipGetLastError());
    // (no wait: the consumer that follows on this stream — kf_culling_counts — synchronises; the mirror's upload block is not touched before the next flush, and
    // every operator of the step after this one waits for the stream before it returns)
    return OSLAM_OK;
}

int h_kf_culling_counts(void* p, int n, const oslam_job_cull_t* jobs, float thDepth) {
    HipOps* o = (HipOps*)p;
    (void)thDepth;   // (the usable-depth bits came with the keyframes' change sets)
    OSLAM_HIP_CHECK(hipSetDevice(o->cfg.device));
    o->cull_pending.clear();
    size_t total = 0;
    for (int i = 0; i < n; i++) { if (jobs[i].slot < 0 || jobs[i].slot >= o->S || jobs[i].n < 0 || !jobs[i].out) { oslam::set_error("kf_culling_counts: bad job"); return OSLAM_E_INVALID; } total += (size_t)jobs[i].n; }
    if (total == 0) return OSLAM_OK;
    // the request goes behind the flush in the mirror's upload block (both may be in flight together)
    const size_t base = oslam::align_up(o->mir_used, 256), oC = base, oOut = oslam::align_up(oC + sizeof(CullCand) * total, 256), end = oOut + 16 * total;
    if (end > o->mir_cap) {   // (rare: the block is grown with nothing in flight; the flush, if any, has completed by then)
        OSLAM_HIP_CHECK(hipStreamSynchronize(o->strm));
        o->mir_used = 0;
        OPS_CHECK(o->ensure_mir(sizeof(CullCand) * total + 16 * total + 1024));
        return h_kf_culling_counts(p, n, jobs, thDepth);
    }
    if (16 * total > o->cull_cap) {
        OSLAM_HIP_CHECK(hipStreamSynchronize(o->strm));
        if (o->cull_h) (void)hipHostFree(o->cull_h);
        o->cull_h = nullptr; o->cull_cap = 0;
        OSLAM_HIP_CHECK(hipHostMalloc((void**)&o->cull_h, 32 * total + 4096, 0));
        o->cull_cap = 32 * total + 4096;
    }
    CullCand* cc = (CullCand*)(o->mir_h + oC);
    size_t at = 0;
    if ((int)o->pt_aux.size() < o->S) { o->pt_aux.resize(o->S, nullptr); o->pt_aux_cap.resize(o->S, 0); }
    for (int i = 0; i < n; i++) {
        if (!o->pt_aux[jobs[i].slot]) OPS_CHECK(o->ensure_pt_aux(jobs[i].slot, 1));
        for (int q = 0; q < jobs[i].n; q++, at++) { cc[at].rec = o->rec_lookup(jobs[i].slot, jobs[i].kf_ids[q]); cc[at].slot = jobs[i].slot; }
        o->cull_pending.push_back({jobs[i].out, jobs[i].n});
    }
    OPS_CHECK(o->sync_mirror_tables());
    OSLAM_HIP_CHECK(hipMemcpyAsync(o->mir_d + oC, o->mir_h + oC, sizeof(CullCand) * total, hipMemcpyHostToDevice, o->strm));
    hipLaunchKernelGGL(k_cull_counts, dim3((unsigned)total), dim3(64), 0, o->strm, (const CullCand*)(o->mir_d + oC), (uint8_t* const*)o->d_rec_chunk, (uint8_t* const*)o->d_pt_aux,
                       mirror_geom(o), (int32_t*)(o->mir_d + oOut));
    OSLAM_HIP_CHECK(hipGetLastError());
    OSLAM_HIP_CHECK(oslam::copy_to_host_async(o->cull_h, o->mir_d + oOut, 16 * total, o->strm));
    o->mir_used = end;
    return OSLAM_OK;   // (results: h_kf_culling_collect)
}

int h_kf_culling_collect(void* p) {
    HipOps* o = (HipOps*)p;
    OSLAM_HIP_CHECK(hipSetDevice(o->cfg.device));
    if (o->cull_pending.empty()) { o->mir_used = 0; return OSLAM_OK; }
    OSLAM_HIP_CHECK(oslam::stream_wait(o->strm));   // (usually already drained: the MapPoint-update operator between request and collection synchronises the same stream)
    size_t at = 0;
    for (const HipOps::CullPending& c : o->cull_pending) { memcpy(c.out, o->cull_h + 16 * at, 16 * (size_t)c.n); at += (size_t)c.n; }
    o->cull_pending.clear();
    o->mir_used = 0;
    return OSLAM_OK;
}

// ---- SearchInNeighbors, second direction, from the mirror (oslam_job_fuse_cur_t) ----
// vpFuseCandidates of one sequence = the targets' point lists in order, bad points skipped, every point once.  "Once, at its FIRST occurrence in (target, slot)
// order" in parallel: every (target t, slot i) proposes the key stamp << 32 | (0xFFFF - t) << 16 | (0xFFFF - i) for its point with a 64-bit atomicMax (a newer stamp
// beats an old one; within a stamp the smallest (t, i) is the largest key); the slot whose key stays is the point's first occurrence.  The current keyframe's own
// points get the stamp in a second word: a candidate already observed by the keyframe is excluded (MapPoint::IsInKeyFrame, src/ORBmatcher.cc:849).
struct FuseCurJob { int32_t slot, cur_rec, nT, t_off; };
__device__ __forceinline__ unsigned long long fusecur_key(uint32_t stamp, int t, int i) { return ((unsigned long long)stamp << 32) | ((unsigned long long)(0xFFFF - t) << 16) | (unsigned long long)(0xFFFF - i); }
__global__ __launch_bounds__(256) void k_fusecur_mark(const FuseCurJob* jobs, const int32_t* recs, uint8_t* const* chunks, uint8_t* const* pt_aux, MirrorGeom g, uint32_t stamp) {
    const FuseCurJob j = jobs[blockIdx.z];
    const int t = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (t > j.nT || i >= g.cap) return;
    const int rec = t == j.nT ? j.cur_rec : recs[j.t_off + t];
    if (rec < 0) return;
    const int p = mirror_mp(chunks, g, rec)[i];
    if (p < 0) return;
    uint4* aux = (uint4*)pt_aux[j.slot] + 2 * (size_t)p;
    if (t == j.nT) { ((uint32_t*)(aux + 1))[2] = stamp; return; }
    if (aux[0].y != 0u) return;
    atomicMax((unsigned long long*)(aux + 1), fusecur_key(stamp, t, i));
}
__global__ __launch_bounds__(1024) void k_fusecur_list(const FuseCurJob* jobs, const int32_t* recs, uint8_t* const* chunks, uint8_t* const* pt_aux, MirrorGeom g, uint32_t stamp, int stride,
                                                        int32_t* ids, uint8_t* excl, int32_t* Mout) {
    const FuseCurJob j = jobs[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    __shared__ int s_cnt[16];
    __shared__ int s_base;
    if (tid == 0) s_base = 0;
    __syncthreads();
    const uint4* aux = (const uint4*)pt_aux[j.slot];
    for (int t = 0; t < j.nT; t++) {
        const int rec = recs[j.t_off + t];
        if (rec < 0) continue;   // (uniform)
        const int32_t* mp = mirror_mp(chunks, g, rec);
        for (int i0 = 0; i0 < g.cap; i0 += 1024) {
            const int i = i0 + tid;
            int p = -1;
            bool first = false;
            if (i < g.cap) {
                p = mp[i];
                if (p >= 0) {
                    const uint4 a0 = aux[2 * (size_t)p];
                    first = a0.y == 0u && *(const unsigned long long*)(aux + 2 * (size_t)p + 1) == fusecur_key(stamp, t, i);
                }
            }
            const unsigned long long bal = __ballot(first);
            if (lane == 0) s_cnt[wv] = __popcll(bal);
            __syncthreads();
            int off = s_base, tot = 0;
            for (int w2 = 0; w2 < 16; w2++) { if (w2 < wv) off += s_cnt[w2]; tot += s_cnt[w2]; }
            if (first) {
                const int pos = off + __popcll(bal & ((1ull << lane) - 1ull));
                if (pos < stride) {
                    ids[(size_t)blockIdx.x * stride + pos] = p;
                    excl[(size_t)blockIdx.x * stride + pos] = ((const uint32_t*)(aux + 2 * (size_t)p + 1))[2] == stamp ? 1 : 0;
                }
            }
            __syncthreads();
            if (tid == 0) s_base += tot;
            __syncthreads();
        }
    }
    if (tid == 0) { const int b = s_base; Mout[2 * blockIdx.x] = b < stride ? b : stride; Mout[2 * blockIdx.x + 1] = b > stride ? 1 : 0; }
}
// the matches of one job in candidate order: (point, keypoint) pairs straight into the pinned result block
__global__ __launch_bounds__(1024) void k_fusecur_pairs(const int32_t* Mn, const int32_t* ids, const int32_t* q_match, int stride, int max_pairs, int32_t* pairs, int32_t* counts) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int M = Mn[2 * blockIdx.x];
    __shared__ int s_cnt[16];
    __shared__ int s_base;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < M; i0 += 1024) {
        const int i = i0 + tid;
        const int best = i < M ? q_match[(size_t)blockIdx.x * stride + i] : -1;
        const bool hit = best >= 0;
        const unsigned long long bal = __ballot(hit);
        if (lane == 0) s_cnt[wv] = __popcll(bal);
        __syncthreads();
        int off = s_base, tot = 0;
        for (int w2 = 0; w2 < 16; w2++) { if (w2 < wv) off += s_cnt[w2]; tot += s_cnt[w2]; }
        if (hit) {
            const int pos = off + __popcll(bal & ((1ull << lane) - 1ull));
            if (pos < max_pairs) { pairs[((size_t)blockIdx.x * max_pairs + pos) * 2] = ids[(size_t)blockIdx.x * stride + i]; pairs[((size_t)blockIdx.x * max_pairs + pos) * 2 + 1] = best; }
        }
        __syncthreads();
        if (tid == 0) s_base += tot;
        __syncthreads();
    }
    if (tid == 0) { counts[3 * blockIdx.x] = s_base; counts[3 * blockIdx.x + 1] = M; counts[3 * blockIdx.x + 2] = Mn[2 * blockIdx.x + 1]; }
}

__global__ void k_fusecur_counts(const int32_t* Mn, int n, int32_t* M) { const int i = blockIdx.x * 256 + threadIdx.x; if (i < n) M[i] = Mn[2 * i]; }

int h_fuse_into_current(void* p, int n, oslam_job_fuse_cur_t* jobs) {
    HipOps* o = (HipOps*)p;
    OSLAM_HIP_CHECK(hipSetDevice(o->cfg.device));
    if (n == 0) return OSLAM_OK;
    if (n > o->S) { oslam::set_error("fuse_into_current: n > n_sequences"); return OSLAM_E_INVALID; }
    const int stride = HipOps::kFuseCurStride, maxPairs = HipOps::kFuseCurPairs;
    size_t nrec = 0;
    int maxT = 0;
    for (int i = 0; i < n; i++) {
        const oslam_job_fuse_cur_t& j = jobs[i];
        if (j.slot < 0 || j.slot >= o->S || j.n_targets < 0 || j.n_targets > 0xFFFF || !j.pairs || j.max_pairs < 0 || (j.n_targets > 0 && !j.targets)) { oslam::set_error("fuse_into_current: bad job"); return OSLAM_E_INVALID; }
        nrec += (size_t)j.n_targets; maxT = std::max(maxT, j.n_targets);
    }
    if (o->cap > 0xFFFF) { oslam::set_error("fuse_into_current: more than 65535 keypoints per keyframe"); return OSLAM_E_CAPACITY; }
    OPS_CHECK(o->sync_mp_table());
    if ((int)o->pt_aux.size() < o->S) { o->pt_aux.resize(o->S, nullptr); o->pt_aux_cap.resize(o->S, 0); }
    for (int i = 0; i < n; i++) if (!o->pt_aux[jobs[i].slot]) OPS_CHECK(o->ensure_pt_aux(jobs[i].slot, 1));
    OPS_CHECK(o->sync_mirror_tables());
    // job block (pinned, read in place) | device scratch: candidate ids, flags, match table, counts
    const size_t B = n;
    Layout L;
    const size_t oJ = L.take(sizeof(FuseCurJob) * B), oR = L.take(4 * std::max<size_t>(nrec, 1)), oSl = L.take(4 * B), oT = L.take(64 * B), oOw = L.take(12 * B), oRef = L.take(sizeof(oslam_kf_grid_ref_t) * B);
    OPS_CHECK(o->ensure_up(L.off));
    const size_t sIds = 0, sEx = oslam::align_up(sIds + 4 * (size_t)stride * B, 256), sQm = oslam::align_up(sEx + (size_t)stride * B, 256), sM = oslam::align_up(sQm + 4 * (size_t)stride * B, 256),
                 sM1 = oslam::align_up(sM + 8 * B, 256), sEnd = sM1 + 4 * B;
    if (sEnd > o->fc_cap) {
        OSLAM_HIP_CHECK(oslam::stream_wait(o->strm));
        if (o->fc_d) (void)hipFree(o->fc_d);
        o->fc_d = nullptr; o->fc_cap = 0;
        OSLAM_HIP_CHECK(hipMalloc((void**)&o->fc_d, sEnd + sEnd / 2));
        o->fc_cap = sEnd + sEnd / 2;
    }
    Layout R;
    const size_t rCnt = R.take(12 * B), rPairs = R.take(8 * (size_t)maxPairs * B);
    bool dbg = false;
    for (int i = 0; i < n; i++) dbg = dbg || (jobs[i].dbg_cap > 0 && jobs[i].dbg_ids && jobs[i].dbg_excl);
    const size_t rIds = R.take(dbg ? 4 * (size_t)stride * B : 0), rEx = R.take(dbg ? (size_t)stride * B : 0);
    OPS_CHECK(o->ensure_dn(R.off));
    uint8_t* U = o->up_h;
    FuseCurJob* fj = (FuseCurJob*)(U + oJ);
    int32_t* recs = (int32_t*)(U + oR);
    size_t at = 0;
    for (int i = 0; i < n; i++) {
        const oslam_job_fuse_cur_t& j = jobs[i];
        const int cr = o->rec_lookup(j.slot, j.kf);
        if (cr < 0) { oslam::set_error("fuse_into_current: the current keyframe is not resident"); return OSLAM_E_INVALID; }
        fj[i].slot = j.slot; fj[i].cur_rec = cr; fj[i].nT = j.n_targets; fj[i].t_off = (int32_t)at;
        for (int t = 0; t < j.n_targets; t++) recs[at++] = o->rec_lookup(j.slot, j.targets[t]);   // (-1: not resident — cannot happen for a keyframe of the map; skipped)
        ((int32_t*)(U + oSl))[i] = j.slot;
        memcpy(U + oT + 64 * i, j.Tcw, 64); memcpy(U + oOw + 12 * i, j.Ow, 12);
        oslam_kf_grid_ref_t& ref = ((oslam_kf_grid_ref_t*)(U + oRef))[i];
        ref.cell_end = o->rec_cell_end(cr); ref.cand = o->rec_cand(cr); ref.desc = o->rec_desc(cr);
    }
    const uint32_t stamp = ++o->fc_stamp;
    const MirrorGeom g = mirror_geom(o);
    uint8_t* D = o->fc_d;
    o->t_begin();
    hipLaunchKernelGGL(k_fusecur_mark, dim3((unsigned)((g.cap + 255) / 256), (unsigned)(maxT + 1), (unsigned)n), dim3(256), 0, o->strm, (const FuseCurJob*)(U + oJ), (const int32_t*)(U + oR),
                       (uint8_t* const*)o->d_rec_chunk, (uint8_t* const*)o->d_pt_aux, g, stamp);
    hipLaunchKernelGGL(k_fusecur_list, dim3((unsigned)n), dim3(1024), 0, o->strm, (const FuseCurJob*)(U + oJ), (const int32_t*)(U + oR), (uint8_t* const*)o->d_rec_chunk,
                       (uint8_t* const*)o->d_pt_aux, g, stamp, stride, (int32_t*)(D + sIds), D + sEx, (int32_t*)(D + sM));
    OSLAM_HIP_CHECK(hipGetLastError());
    // (k_fuse_search reads M at d_M[b]; the list kernel's table holds [M, overflow] pairs: the search gets a compacted copy)
    hipLaunchKernelGGL(k_fusecur_counts, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, o->strm, (const int32_t*)(D + sM), n, (int32_t*)(D + sM1));
    OPS_CHECK(oslam_fuse_search_device(n, stride, (const oslam_kf_grid_ref_t*)(U + oRef), (const int32_t*)(U + oSl), (const int32_t*)(D + sM1), (const int32_t*)(D + sIds), D + sEx,
                                       o->d_mp_tab, (const float*)(U + oT), (const float*)(U + oOw), o->K5, o->bounds, jobs[0].th, o->logScale, o->scale, o->invSigma2, o->cfg.nLevels,
                                       (int32_t*)(D + sQm), o->strm));
    hipLaunchKernelGGL(k_fusecur_pairs, dim3((unsigned)n), dim3(1024), 0, o->strm, (const int32_t*)(D + sM), (const int32_t*)(D + sIds), (const int32_t*)(D + sQm), stride, maxPairs,
                       (int32_t*)(o->dn_h + rPairs), (int32_t*)(o->dn_h + rCnt));
    OSLAM_HIP_CHECK(hipGetLastError());
    if (dbg) {
        OSLAM_HIP_CHECK(oslam::copy_to_host_async(o->dn_h + rIds, D + sIds, 4 * (size_t)stride * B, o->strm));
        OSLAM_HIP_CHECK(oslam::copy_to_host_async(o->dn_h + rEx, D + sEx, (size_t)stride * B, o->strm));
    }
    o->t_end();
    OSLAM_HIP_CHECK(oslam::stream_wait(o->strm));
    o->t_collect(4, 4, 0);
    const int32_t* cnt = (const int32_t*)(o->dn_h + rCnt);
    for (int i = 0; i < n; i++) {
        oslam_job_fuse_cur_t& j = jobs[i];
        j.n_pairs = cnt[3 * i]; j.n_candidates = cnt[3 * i + 1]; j.overflow = (cnt[3 * i + 2] != 0 || j.n_pairs > maxPairs || j.n_pairs > j.max_pairs) ? 1 : 0;
        if (!j.overflow) memcpy(j.pairs, o->dn_h + rPairs + 8 * (size_t)maxPairs * i, 8 * (size_t)j.n_pairs);
        if (j.dbg_cap > 0 && j.dbg_ids && j.dbg_excl) {
            const int m = std::min(j.n_candidates, j.dbg_cap);
            memcpy(j.dbg_ids, o->dn_h + rIds + 4 * (size_t)stride * i, 4 * (size_t)m);
            memcpy(j.dbg_excl, o->dn_h + rEx + (size_t)stride * i, (size_t)m);
        }
    }
    return OSLAM_OK;
}

// Tracking::UpdateLocalPoints from the mirror (oslam_job_local_list_t): the same first-occurrence compaction as the Fuse candidates, over the local keyframes' lists;
// the ids go straight into the pinned result block.
int h_local_points_list(void* p, int n, oslam_job_local_list_t* jobs) {
    HipOps* o = (HipOps*)p;
    OSLAM_HIP_CHECK(hipSetDevice(o->cfg.device));
    if (n == 0) return OSLAM_OK;
    if (n > o->S) { oslam::set_error("local_points_list: n > n_sequences"); return OSLAM_E_INVALID; }
    const int stride = HipOps::kFuseCurStride;
    size_t nrec = 0;
    int maxT = 0;
    for (int i = 0; i < n; i++) {
        const oslam_job_local_list_t& j = jobs[i];
        if (j.slot < 0 || j.slot >= o->S || j.n_kfs < 0 || j.n_kfs > 0xFFFF || !j.ids || j.cap < 0 || (j.n_kfs > 0 && !j.kfs)) { oslam::set_error("local_points_list: bad job"); return OSLAM_E_INVALID; }
        nrec += (size_t)j.n_kfs; maxT = std::max(maxT, j.n_kfs);
    }
    if (o->cap > 0xFFFF) { oslam::set_error("local_points_list: more than 65535 keypoints per keyframe"); return OSLAM_E_CAPACITY; }
    if ((int)o->pt_aux.size() < o->S) { o->pt_aux.resize(o->S, nullptr); o->pt_aux_cap.resize(o->S, 0); }
    for (int i = 0; i < n; i++) if (!o->pt_aux[jobs[i].slot]) OPS_CHECK(o->ensure_pt_aux(jobs[i].slot, 1));
    OPS_CHECK(o->sync_mirror_tables());
    const size_t B = n;
    Layout L;
    const size_t oJ = L.take(sizeof(FuseCurJob) * B), oR = L.take(4 * std::max<size_t>(nrec, 1));
    OPS_CHECK(o->ensure_up(L.off));
    const size_t sEx = 0, sEnd = (size_t)stride * B;
    if (sEnd > o->fc_cap) {
        OSLAM_HIP_CHECK(oslam::stream_wait(o->strm));
        if (o->fc_d) (void)hipFree(o->fc_d);
        o->fc_d = nullptr; o->fc_cap = 0;
        OSLAM_HIP_CHECK(hipMalloc((void**)&o->fc_d, sEnd + sEnd / 2));
        o->fc_cap = sEnd + sEnd / 2;
    }
    Layout R;
    const size_t rM = R.take(8 * B), rIds = R.take(4 * (size_t)stride * B);
    OPS_CHECK(o->ensure_dn(R.off));
    uint8_t* U = o->up_h;
    FuseCurJob* fj = (FuseCurJob*)(U + oJ);
    int32_t* recs = (int32_t*)(U + oR);
    size_t at = 0;
    for (int i = 0; i < n; i++) {
        const oslam_job_local_list_t& j = jobs[i];
        fj[i].slot = j.slot; fj[i].cur_rec = -1; fj[i].nT = j.n_kfs; fj[i].t_off = (int32_t)at;
        for (int t = 0; t < j.n_kfs; t++) recs[at++] = o->rec_lookup(j.slot, j.kfs[t]);
    }
    const uint32_t stamp = ++o->fc_stamp;
    const MirrorGeom g = mirror_geom(o);
    o->t_begin();
    hipLaunchKernelGGL(k_fusecur_mark, dim3((unsigned)((g.cap + 255) / 256), (unsigned)(maxT + 1), (unsigned)n), dim3(256), 0, o->strm, (const FuseCurJob*)(U + oJ), (const int32_t*)(U + oR),
                       (uint8_t* const*)o->d_rec_chunk, (uint8_t* const*)o->d_pt_aux, g, stamp);
    hipLaunchKernelGGL(k_fusecur_list, dim3((unsigned)n), dim3(1024), 0, o->strm, (const FuseCurJob*)(U + oJ), (const int32_t*)(U + oR), (uint8_t* const*)o->d_rec_chunk,
                       (uint8_t* const*)o->d_pt_aux, g, stamp, stride, (int32_t*)(o->dn_h + rIds), o->fc_d + sEx, (int32_t*)(o->dn_h + rM));
    OSLAM_HIP_CHECK(hipGetLastError());
    o->t_end();
    OSLAM_HIP_CHECK(oslam::stream_wait(o->strm));
    o->t_collect(3, 2, 0);
    const int32_t* Mn = (const int32_t*)(o->dn_h + rM);
    for (int i = 0; i < n; i++) {
        oslam_job_local_list_t& j = jobs[i];
        j.n_ids = Mn[2 * i]; j.overflow = (Mn[2 * i + 1] != 0 || j.n_ids > j.cap) ? 1 : 0;
        if (!j.overflow) memcpy(j.ids, o->dn_h + rIds + 4 * (size_t)stride * i, 4 * (size_t)j.n_ids);
    }
    return OSLAM_OK;
}

int h_resident_points(void* p) { return ((HipOps*)p)->mp_tab_on ? 1 : 0; }

int h_point_record(void* p, int slot, int id, uint8_t out[64]) {
    HipOps* o = (HipOps*)p;
    OSLAM_HIP_CHECK(hipSetDevice(o->cfg.device));
    if (slot < 0 || slot >= (int)o->mp_tab.size() || id < 0 || (size_t)id >= o->mp_cap[slot]) { oslam::set_error("point_record: no such record"); return OSLAM_E_INVALID; }
    OSLAM_HIP_CHECK(oslam::stream_wait(o->strm));
    OSLAM_HIP_CHECK(hipMemcpy(out, o->mp_tab[slot] + (size_t)id * 64, 64, hipMemcpyDeviceToHost));
    return OSLAM_OK;
}

// KeyFrame::ComputeBoW node assignment of registered keyframes from their resident descriptors (see oslam_slam_ops_t::bow_nodes_keyed)
int h_bow_nodes_keyed(void* p, int n, const int32_t* slots, const int32_t* kf_ids, const uint64_t* top, const uint64_t* sub, const int32_t* counts, uint32_t* const* out) {
    HipOps* o = (HipOps*)p;
    OSLAM_HIP_CHECK(hipSetDevice(o->cfg.device));
    if (n == 0) return OSLAM_OK;
    const size_t cap = o->cap;
    Layout L;
    const size_t oPtr = L.take(8 * (size_t)n), oCnt = L.take(4 * (size_t)n), oTop = L.take(320), oSub = L.take(3200);
    const size_t in_bytes = L.off;
    const size_t oOut = L.take(4 * cap * (size_t)n);
    OPS_CHECK(o->ensure_up(L.off));
    OPS_CHECK(o->ensure_dn(4 * cap * (size_t)n));
    uint8_t* U = o->up_h;
    for (int i = 0; i < n; i++) {
        const int r = o->rec_lookup(slots[i], kf_ids[i]);
        if (r < 0 || counts[i] < 0 || counts[i] > (int)cap) { oslam::set_error("bow_nodes: keyframe not resident / bad count"); return OSLAM_E_INVALID; }
        ((const uint8_t**)(U + oPtr))[i] = o->rec_desc(r);
        ((int32_t*)(U + oCnt))[i] = counts[i];
    }
    memcpy(U + oTop, top, 320); memcpy(U + oSub, sub, 3200);
    OSLAM_HIP_CHECK(hipMemcpyAsync(o->up_d, U, in_bytes, hipMemcpyHostToDevice, o->strm));
    uint8_t* Dv = o->up_d;
    o->t_begin();
    OPS_CHECK(oslam_bow_nodes_device((const uint8_t* const*)(Dv + oPtr), (const int32_t*)(Dv + oCnt), n, (int)cap, (const uint64_t*)(Dv + oTop), (const uint64_t*)(Dv + oSub),
                                     (uint32_t*)(Dv + oOut), o->strm));
    o->t_end();
    OSLAM_HIP_CHECK(hipMemcpyAsync(o->dn_h, Dv + oOut, 4 * cap * (size_t)n, hipMemcpyDeviceToHost, o->strm));
    OSLAM_HIP_CHECK(oslam::stream_wait(o->strm));
    o->t_collect(7, 1, 0);
    o->pool->parallel_for(n, [&](int i) { memcpy(out[i], o->dn_h + 4 * cap * (size_t)i, 4 * (size_t)counts[i]); });
    return OSLAM_OK;
}

int h_bow_keyed(void* p, int n, oslam_job_bow_t* jobs, const oslam_kf_key_t* keys) {
    HipOps* o = (HipOps*)p;
    OSLAM_HIP_CHECK(hipSetDevice(o->cfg.device));
    std::vector<oslam_bow_resident_t> res(n);
    const size_t cap = o->cap;
    for (int i = 0; i < n; i++) {
        memset(&res[i], 0, sizeof(res[i]));
        const int r1 = o->rec_lookup(keys[i].slot, keys[i].kf1);
        if (r1 >= 0) { res[i].d_keys1 = o->rec_keys(r1); res[i].d_desc1 = o->rec_desc(r1); res[i].d_uRight1 = o->rec_ur(r1); }
        if (keys[i].kf2 == -2 && keys[i].slot >= 0 && keys[i].slot < o->S) {   // side 2 = the current frame of the slot
            res[i].d_keys2 = o->d_keysUn + cap * keys[i].slot; res[i].d_desc2 = o->d_desc + 32 * cap * keys[i].slot; res[i].d_uRight2 = o->cur_uRight + cap * keys[i].slot;
        } else {
            const int r2 = o->rec_lookup(keys[i].slot, keys[i].kf2);
            if (r2 >= 0) { res[i].d_keys2 = o->rec_keys(r2); res[i].d_desc2 = o->rec_desc(r2); res[i].d_uRight2 = o->rec_ur(r2); }
        }
    }
    return oslam_match_bow_batch_resident(o->bow, n, jobs, res.data(), o->scale, o->sigma2, o->cfg.nLevels);
}

int h_kernel_times(void* p, int enable, double* out) {
    HipOps* o = (HipOps*)p;
    OSLAM_HIP_CHECK(hipSetDevice(o->cfg.device));
    if (enable && !o->tev0) { OSLAM_HIP_CHECK(hipEventCreate(&o->tev0)); OSLAM_HIP_CHECK(hipEventCreate(&o->tev1)); OSLAM_HIP_CHECK(hipEventCreate(&o->tevB0)); OSLAM_HIP_CHECK(hipEventCreate(&o->tevB1)); }
    {   // the BoW matchers and the triangulation run on their handles' own streams
        double ms = 0; long long ln = 0;
        OPS_CHECK(oslam_bow_kernel_time(o->bow, enable, &ms, &ln));
        o->kt[15] += ms; o->kt[16] += (double)ln;
        OPS_CHECK(oslam_mappoint_kernel_time(o->mp, enable, &ms, &ln));
        o->kt[15] += ms; o->kt[16] += (double)ln;
    }
    if (out) memcpy(out, o->kt, sizeof(o->kt));
    memset(o->kt, 0, sizeof(o->kt));
    o->timing = enable;
    OPS_CHECK(oslam_lba_kernel_time(o->ba, enable, nullptr, nullptr));
    OPS_CHECK(oslam_lba_kernel_time(o->ba1, enable, nullptr, nullptr));
    return OSLAM_OK;
}

// search half of ORBmatcher::Fuse for n (keyframe, candidate list) jobs in one launch (one workgroup per keyframe)
static int fuse_impl(HipOps* o, int n, oslam_job_fuse_t* jobs, const oslam_kf_key_t* keys);
int h_fuse(void* p, int n, oslam_job_fuse_t* jobs) { return fuse_impl((HipOps*)p, n, jobs, nullptr); }
int h_fuse_keyed(void* p, int n, oslam_job_fuse_t* jobs, const oslam_kf_key_t* keys) { return fuse_impl((HipOps*)p, n, jobs, keys); }

static int fuse_impl(HipOps* o, int n, oslam_job_fuse_t* jobs, const oslam_kf_key_t* keys) {
    OSLAM_HIP_CHECK(hipSetDevice(o->cfg.device));   // the HIP current device is per host thread: a handle may be stepped from any thread
    if (n == 0) return OSLAM_OK;
    if (n > o->S) { oslam::set_error("fuse: n > n_sequences"); return OSLAM_E_INVALID; }
    const size_t cap = o->cap, B = n;
    int maxM = 1;
    for (int i = 0; i < n; i++) {
        if (jobs[i].N > (int)cap) { oslam::set_error("fuse: %d keypoints exceed capacity", jobs[i].N); return OSLAM_E_CAPACITY; }
        maxM = std::max(maxM, jobs[i].M);
    }
    if (maxM > o->max_local) {
        OSLAM_HIP_CHECK(oslam::stream_wait(o->strm));
        oslam_matcher_destroy(o->m_map);
        o->m_map = nullptr;
        o->max_local = (int)oslam::align_up((size_t)maxM + maxM / 2, 64);
        OPS_CHECK(oslam_matcher_create(&o->m_map, o->S, o->cap, o->max_local, o->cfg.device));
    }
    const size_t st = oslam::align_up((size_t)maxM, 64);
    Layout L;
    // upload block: counts, queries, the segment table of the resident keyframes, and (last, so that a fully resident batch does not ship them) the
    // keypoint / stereo / descriptor arrays of the keyframes that are not resident
    const size_t oN = L.take(4 * B), oM = L.take(4 * B), oQ = L.take(sizeof(oslam_proj_query_t) * st * B), oSeg = L.take(sizeof(CopySegH) * 3 * B);
    const size_t small_bytes = L.off;
    const size_t oKeys = L.take(sizeof(oslam_keypoint_t) * cap * B), oUr = L.take(4 * cap * B), oDesc = L.take(32 * cap * B);
    OPS_CHECK(o->ensure_up(L.off));
    uint8_t* U = o->up_h;
    uint8_t* Dv = o->up_d;
    std::vector<int> rec(n, -1);
    int nres = 0;
    if (keys) for (int i = 0; i < n; i++) { rec[i] = o->rec_lookup(keys[i].slot, keys[i].kf1); nres += rec[i] >= 0; }
    CopySegH* segs = (CopySegH*)(U + oSeg);
    for (int i = 0, q = 0; i < n; i++)
        if (rec[i] >= 0) {
            const size_t N = jobs[i].N;
            segs[q++] = {(const uint8_t*)o->rec_keys(rec[i]), Dv + oKeys + sizeof(oslam_keypoint_t) * cap * i, (uint32_t)(sizeof(oslam_keypoint_t) * N), 0};
            segs[q++] = {(const uint8_t*)o->rec_ur(rec[i]), Dv + oUr + 4 * cap * i, (uint32_t)(4 * N), 0};
            segs[q++] = {o->rec_desc(rec[i]), Dv + oDesc + 32 * cap * i, (uint32_t)(32 * N), 0};
        }
    o->pool->parallel_for(n, [&](int i) {
        const oslam_job_fuse_t& j = jobs[i];
        const size_t N = j.N, M = j.M;
        ((int32_t*)(U + oN))[i] = j.N; ((int32_t*)(U + oM))[i] = j.M;
        if (rec[i] < 0) {
            memcpy(U + oKeys + sizeof(oslam_keypoint_t) * cap * i, j.keysUn, sizeof(oslam_keypoint_t) * N);
            memcpy(U + oUr + 4 * cap * i, j.uRight, 4 * N); memcpy(U + oDesc + 32 * cap * i, j.desc, 32 * N);
        }
        memcpy(U + oQ + sizeof(oslam_proj_query_t) * st * i, j.queries, sizeof(oslam_proj_query_t) * M);
    });
    OSLAM_HIP_CHECK(hipMemcpyAsync(Dv, U, nres == n ? small_bytes : L.off, hipMemcpyHostToDevice, o->strm));
    o->t_begin();
    if (nres) OPS_CHECK(oslam_copy_segments_device(Dv + oSeg, 3 * nres, o->strm));
    oslam_match_frames_t fr;
    fr.keysUn = (const oslam_keypoint_t*)(Dv + oKeys); fr.kp_stride = (int)cap; fr.uRight = (const float*)(Dv + oUr); fr.desc = Dv + oDesc; fr.blocked = nullptr;
    fr.n_kps = (const int32_t*)(Dv + oN); fr.n_kps_const = 0;
    fr.minX = o->bounds[0]; fr.minY = o->bounds[1]; fr.maxX = o->bounds[2]; fr.maxY = o->bounds[3];
    OPS_CHECK(oslam_match_fuse_batch_device(o->m_map, &fr, (const oslam_proj_query_t*)(Dv + oQ), (int)st, (const int32_t*)(Dv + oM), 0, n, o->invSigma2, o->cfg.nLevels, o->strm));
    o->t_end();
    const int32_t* d_qm;
    OPS_CHECK(oslam_match_results_device(o->m_map, &d_qm, nullptr, nullptr, nullptr, nullptr, nullptr));
    OPS_CHECK(o->ensure_dn(4 * st * B));
    OSLAM_HIP_CHECK(hipMemcpyAsync(o->dn_h, d_qm, 4 * st * B, hipMemcpyDeviceToHost, o->strm));
    OSLAM_HIP_CHECK(oslam::stream_wait(o->strm));
    o->t_collect(4, 2, 0);
    o->pool->parallel_for(n, [&](int i) { memcpy(jobs[i].q_match, o->dn_h + 4 * st * i, 4 * (size_t)jobs[i].M); });
    return OSLAM_OK;
}

// ORBmatcher::Fuse search half with the candidates named by map-point id: the projection gates run on the device from the resident records, the search reads the
// resident keyframe; what travels is 5 bytes per candidate up and 4 back
int h_fuse_points_keyed(void* p, int n, oslam_job_fuse_pts_t* jobs) {
    HipOps* o = (HipOps*)p;
    OSLAM_HIP_CHECK(hipSetDevice(o->cfg.device));
    if (n == 0) return OSLAM_OK;
    if (n > o->S) { oslam::set_error("fuse_points: n > n_sequences"); return OSLAM_E_INVALID; }
    const size_t cap = o->cap, B = n;
    int maxM = 1;
    std::vector<int> rec(n);
    for (int i = 0; i < n; i++) {
        rec[i] = o->rec_lookup(jobs[i].slot, jobs[i].kf);
        if (rec[i] < 0 || jobs[i].M < 0 || jobs[i].N < 0 || jobs[i].N > (int)cap) { oslam::set_error("fuse_points: keyframe not resident / bad size"); return OSLAM_E_INVALID; }
        maxM = std::max(maxM, jobs[i].M);
    }
    OPS_CHECK(o->mpu_launch_pending());   // (the previous round's descriptor updates, if they were deferred: in front of this search, behind one wait)
    static const bool staged = getenv("OSLAM_SLAM_FUSE_STAGED") != nullptr;   // A/B knob: the round-2 path (copy the records into a batch, queries, LDS window search)
    if (!staged) {
        // gates + window search of every candidate in ONE launch, straight from the resident records and their grids (oslam_fuse_search_device)
        OPS_CHECK(o->sync_mp_table());
        const size_t st = oslam::align_up((size_t)maxM, 64);
        Layout L;
        const size_t oM = L.take(4 * B), oSl = L.take(4 * B), oT = L.take(64 * B), oOw = L.take(12 * B), oRef = L.take(sizeof(oslam_kf_grid_ref_t) * B), oIds = L.take(4 * st * B),
                     oEx = L.take(st * B);
        const size_t head = L.off;
        OPS_CHECK(o->ensure_up(L.off));
        OPS_CHECK(o->ensure_dn(4 * st * B));   // the match table: written by the kernel itself
        uint8_t* U = o->up_h;
        uint8_t* Dv = o->up_d;
        o->pool->parallel_for(n, [&](int i) {
            const oslam_job_fuse_pts_t& j = jobs[i];
            const size_t M = j.M;
            ((int32_t*)(U + oM))[i] = j.M; ((int32_t*)(U + oSl))[i] = j.slot;
            memcpy(U + oT + 64 * i, j.Tcw, 64); memcpy(U + oOw + 12 * i, j.Ow, 12);
            memcpy(U + oIds + 4 * st * i, j.ids, 4 * M); memcpy(U + oEx + st * i, j.excl, M);
            oslam_kf_grid_ref_t& ref = ((oslam_kf_grid_ref_t*)(U + oRef))[i];
            ref.cell_end = o->rec_cell_end(rec[i]); ref.cand = o->rec_cand(rec[i]); ref.desc = o->rec_desc(rec[i]);
        });
        // The job block (~5 KB per job) is read by the kernel where it is — the pinned staging block is mapped into the device's address space — instead of
        // being copied first: one dependent copy-engine hop less per SearchInNeighbors round (OSLAM_SLAM_FUSE_UPLOAD=1 restores the copy: A/B knob).
        static const bool upload = getenv("OSLAM_SLAM_FUSE_UPLOAD") != nullptr;
        const uint8_t* In = upload ? Dv : U;
        if (upload) OSLAM_HIP_CHECK(hipMemcpyAsync(Dv, U, head, hipMemcpyHostToDevice, o->strm));
        o->t_begin();
        OPS_CHECK(oslam_fuse_search_device(n, (int)st, (const oslam_kf_grid_ref_t*)(In + oRef), (const int32_t*)(In + oSl), (const int32_t*)(In + oM), (const int32_t*)(In + oIds),
                                           In + oEx, o->d_mp_tab, (const float*)(In + oT), (const float*)(In + oOw), o->K5, o->bounds, jobs[0].th, o->logScale, o->scale,
                                           o->invSigma2, o->cfg.nLevels, (int32_t*)o->dn_h, o->strm));   // (every query writes its one result: straight into the pinned result block, no copy kernel behind it)
        o->t_end();
        OSLAM_HIP_CHECK(oslam::stream_wait(o->strm));
        o->t_collect(4, 1, 0);
        o->pool->parallel_for(n, [&](int i) { memcpy(jobs[i].q_match, o->dn_h + 4 * st * i, 4 * (size_t)jobs[i].M); });
        return OSLAM_OK;
    }
    if (maxM > o->max_local) {
        OSLAM_HIP_CHECK(oslam::stream_wait(o->strm));
        oslam_matcher_destroy(o->m_map);
        o->m_map = nullptr;
        o->max_local = (int)oslam::align_up((size_t)maxM + maxM / 2, 64);
        OPS_CHECK(oslam_matcher_create(&o->m_map, o->S, o->cap, o->max_local, o->cfg.device));
    }
    OPS_CHECK(o->sync_mp_table());
    const size_t st = oslam::align_up((size_t)maxM, 64);
    Layout L;
    const size_t oN = L.take(4 * B), oM = L.take(4 * B), oSl = L.take(4 * B), oT = L.take(64 * B), oOw = L.take(12 * B), oSeg = L.take(sizeof(CopySegH) * 3 * B),
                 oIds = L.take(4 * st * B), oEx = L.take(st * B);
    const size_t head = L.off;
    const size_t oQ = L.take(sizeof(oslam_proj_query_t) * st * B), oKeys = L.take(sizeof(oslam_keypoint_t) * cap * B), oUr = L.take(4 * cap * B), oDesc = L.take(32 * cap * B);
    OPS_CHECK(o->ensure_up(L.off));
    uint8_t* U = o->up_h;
    uint8_t* Dv = o->up_d;
    CopySegH* segs = (CopySegH*)(U + oSeg);
    o->pool->parallel_for(n, [&](int i) {
        const oslam_job_fuse_pts_t& j = jobs[i];
        const size_t N = j.N, M = j.M;
        ((int32_t*)(U + oN))[i] = j.N; ((int32_t*)(U + oM))[i] = j.M; ((int32_t*)(U + oSl))[i] = j.slot;
        memcpy(U + oT + 64 * i, j.Tcw, 64); memcpy(U + oOw + 12 * i, j.Ow, 12);
        memcpy(U + oIds + 4 * st * i, j.ids, 4 * M); memcpy(U + oEx + st * i, j.excl, M);
        segs[3 * i] = {(const uint8_t*)o->rec_keys(rec[i]), Dv + oKeys + sizeof(oslam_keypoint_t) * cap * i, (uint32_t)(sizeof(oslam_keypoint_t) * N), 0};
        segs[3 * i + 1] = {(const uint8_t*)o->rec_ur(rec[i]), Dv + oUr + 4 * cap * i, (uint32_t)(4 * N), 0};
        segs[3 * i + 2] = {o->rec_desc(rec[i]), Dv + oDesc + 32 * cap * i, (uint32_t)(32 * N), 0};
    });
    OSLAM_HIP_CHECK(hipMemcpyAsync(Dv, U, head, hipMemcpyHostToDevice, o->strm));
    o->t_begin();
    OPS_CHECK(oslam_copy_segments_device(Dv + oSeg, 3 * n, o->strm));
    OPS_CHECK(oslam_fuse_queries_device(n, (int)st, (const int32_t*)(Dv + oSl), (const int32_t*)(Dv + oM), (const int32_t*)(Dv + oIds), Dv + oEx, o->d_mp_tab,
                                        (const float*)(Dv + oT), (const float*)(Dv + oOw), o->K5, o->bounds, jobs[0].th, o->logScale, o->scale, o->cfg.nLevels,
                                        (oslam_proj_query_t*)(Dv + oQ), o->strm));
    oslam_match_frames_t fr;
    fr.keysUn = (const oslam_keypoint_t*)(Dv + oKeys); fr.kp_stride = (int)cap; fr.uRight = (const float*)(Dv + oUr); fr.desc = Dv + oDesc; fr.blocked = nullptr;
    fr.n_kps = (const int32_t*)(Dv + oN); fr.n_kps_const = 0;
    fr.minX = o->bounds[0]; fr.minY = o->bounds[1]; fr.maxX = o->bounds[2]; fr.maxY = o->bounds[3];
    OPS_CHECK(oslam_match_fuse_batch_device(o->m_map, &fr, (const oslam_proj_query_t*)(Dv + oQ), (int)st, (const int32_t*)(Dv + oM), 0, n, o->invSigma2, o->cfg.nLevels, o->strm));
    o->t_end();
    const int32_t* d_qm;
    OPS_CHECK(oslam_match_results_device(o->m_map, &d_qm, nullptr, nullptr, nullptr, nullptr, nullptr));
    OPS_CHECK(o->ensure_dn(4 * st * B));
    OSLAM_HIP_CHECK(hipMemcpyAsync(o->dn_h, d_qm, 4 * st * B, hipMemcpyDeviceToHost, o->strm));
    OSLAM_HIP_CHECK(oslam::stream_wait(o->strm));
    o->t_collect(4, 3, 0);
    o->pool->parallel_for(n, [&](int i) { memcpy(jobs[i].q_match, o->dn_h + 4 * st * i, 4 * (size_t)jobs[i].M); });
    return OSLAM_OK;
}

int h_bow(void* p, int n, oslam_job_bow_t* jobs) {
    HipOps* o = (HipOps*)p;
    OSLAM_HIP_CHECK(hipSetDevice(o->cfg.device));   // the HIP current device is per host thread: a handle may be stepped from any thread
    return oslam_match_bow_batch(o->bow, n, jobs, o->scale, o->sigma2, o->cfg.nLevels);
}

int h_triangulate(void* p, int n, oslam_job_triangulate_t* jobs) {
    HipOps* o = (HipOps*)p;
    OSLAM_HIP_CHECK(hipSetDevice(o->cfg.device));   // the HIP current device is per host thread: a handle may be stepped from any thread
    std::vector<oslam_tri_kf_t> k1(n), k2(n);
    std::vector<int32_t> ps(n + 1, 0), i1, i2;
    for (int i = 0; i < n; i++) {
        k1[i] = jobs[i].kf1; k2[i] = jobs[i].kf2;
        ps[i + 1] = ps[i] + jobs[i].M;
        i1.insert(i1.end(), jobs[i].idx1, jobs[i].idx1 + jobs[i].M);
        i2.insert(i2.end(), jobs[i].idx2, jobs[i].idx2 + jobs[i].M);
    }
    const int M = ps[n];
    if (M == 0) return OSLAM_OK;
    std::vector<uint8_t> ok(M);
    std::vector<float> x3(3 * (size_t)M);
    OPS_CHECK(oslam_mp_triangulate_pairs(o->mp, n, k1.data(), k2.data(), ps.data(), i1.data(), i2.data(), o->scale, o->sigma2, o->cfg.nLevels,
                                         1.5f * o->cfg.scaleFactor, ok.data(), x3.data()));
    for (int i = 0; i < n; i++) {
        memcpy(jobs[i].ok, &ok[ps[i]], (size_t)jobs[i].M);
        memcpy(jobs[i].x3D, &x3[3 * (size_t)ps[i]], 12 * (size_t)jobs[i].M);
    }
    return OSLAM_OK;
}

void h_destroy(void* p) {
    HipOps* o = (HipOps*)p;
    (void)hipSetDevice(o->cfg.device);
    if (o->job_active) (void)h_lba_wait(o);   // (the submitted windows' arrays belong to the driver handle being destroyed)
    lba_service_release(o->svc);
    oslam_orb_destroy(o->orb); oslam_orb_destroy(o->orbR); oslam_stereo_destroy(o->stereo); (void)hipFree(o->d_grayR); oslam_matcher_destroy(o->m_last); oslam_matcher_destroy(o->m_map); oslam_poseopt_destroy(o->po);
    oslam_lba_destroy(o->ba); oslam_lba_destroy(o->ba1); oslam_mappoint_destroy(o->mp); oslam_frame_destroy(o->fr); oslam_bow_destroy(o->bow);
    if (o->up_h) (void)hipHostFree(o->up_h);
    if (o->dn_h) (void)hipHostFree(o->dn_h);
    for (uint8_t* c : o->rec_chunks) (void)hipFree(c);
    if (o->d_rec_desc) (void)hipFree(o->d_rec_desc);
    for (uint8_t* q : o->mp_tab) if (q) (void)hipFree(q);
    (void)hipFree(o->d_mp_tab);
    for (uint8_t* q : o->pt_aux) if (q) (void)hipFree(q);
    if (o->d_pt_aux) (void)hipFree(o->d_pt_aux);
    if (o->d_rec_chunk) (void)hipFree(o->d_rec_chunk);
    if (o->cull_h) (void)hipHostFree(o->cull_h);
    if (o->mir_h) (void)hipHostFree(o->mir_h);
    if (o->mir_d) (void)hipFree(o->mir_d);
    (void)hipFree(o->up_d); (void)hipFree(o->d_maskbits); (void)hipFree(o->d_loc); (void)hipFree(o->d_lq); (void)hipFree(o->d_inview); (void)hipFree(o->d_objbits); (void)hipFree(o->d_maskstage);
    delete o->pool;
    if (o->tev0) (void)hipEventDestroy(o->tev0);
    if (o->tevB0) (void)hipEventDestroy(o->tevB0);
    if (o->tevB1) (void)hipEventDestroy(o->tevB1);
    if (o->upB_h) (void)hipHostFree(o->upB_h);
    if (o->kfk_h) (void)hipHostFree(o->kfk_h);
    if (o->upB_d) (void)hipFree(o->upB_d);
    if (o->dnB_h) (void)hipHostFree(o->dnB_h);
    if (o->tev1) (void)hipEventDestroy(o->tev1);
    if (o->strm) (void)hipStreamDestroy(o->strm);
    (void)hipFree(o->d_gray); (void)hipFree(o->d_depth); (void)hipFree(o->d_keysUn); (void)hipFree(o->d_keysUn_prev); (void)hipFree(o->d_uRight); (void)hipFree(o->d_mvDepth); (void)hipFree(o->d_status);
    delete o;
}

}  // namespace

int oslam_slam_make_hip_ops(const oslam_slam_config_t* cfg, oslam_slam_ops_t* ops) {
    if (oslam_device_count() <= 0 || hipSetDevice(cfg->device) != hipSuccess) {
        oslam::set_error("oslam_slam_create: no HIP device (the tracking driver has no CPU fallback)");
        return OSLAM_E_HIP;
    }
    memset(ops, 0, sizeof(*ops));
    HipOps* o = new HipOps;
    o->cfg = *cfg; o->S = cfg->n_sequences;
    o->pool = new oslam_drv::Pool(cfg->host_threads > 1 ? cfg->host_threads : 1, /*own_workers*/ false);
    if (hipStreamCreateWithFlags(&o->strm, hipStreamNonBlocking) != hipSuccess) o->strm = nullptr;
    const int dev = cfg->device;
    int rc = oslam_orb_create(&o->orb, cfg->nFeatures, cfg->scaleFactor, cfg->nLevels, cfg->iniThFAST, cfg->minThFAST, cfg->width, cfg->height, o->S, dev);
    if (!rc) {
        o->cap = oslam_orb_max_keypoints(o->orb);
        o->max_local = getenv("OSLAM_SLAM_MAX_LOCAL") ? atoi(getenv("OSLAM_SLAM_MAX_LOCAL")) : 8192;   // first reservation; grows on demand
        rc = oslam_matcher_create(&o->m_last, o->S, o->cap, o->cap, dev);
    }
    if (!rc) rc = oslam_matcher_create(&o->m_map, o->S, o->cap, o->max_local, dev);
    if (!rc && cfg->sensor == 1) rc = oslam_orb_create(&o->orbR, cfg->nFeatures, cfg->scaleFactor, cfg->nLevels, cfg->iniThFAST, cfg->minThFAST, cfg->width, cfg->height, o->S, dev);
    if (!rc && cfg->sensor == 1) rc = oslam_stereo_create(&o->stereo, o->S, o->cap, dev);
    if (!rc) rc = oslam_poseopt_create(&o->po, o->S, o->cap, dev);
    if (!rc) rc = oslam_lba_create(&o->ba, o->S, 1 << 16, 4096, 32768, dev);    // keyframes / points / edges per window grow on demand (include/oslam_hip.h); <= 128 FREE keyframes
    if (!rc) rc = oslam_lba_create(&o->ba1, 1, 1 << 16, 4096, 32768, dev);
    // batches of windows: every LM trial of ALL windows as short whole-GPU launches (mode 1: the windows of a call spread over all CUs and the kernels of the other
    // handles interleave; the Schur complement is formed on chip by tiles).  OSLAM_LBA_BATCH_MODE = 2 (one workgroup per window, the whole schedule in one launch:
    // the most work per CU-second, but a call of ~40 windows then holds 40 CUs for tens of milliseconds and the other handles' short kernels queue behind the
    // windows of all handles: measured 7.6 k against 16.8 k frames/s in bench.py's steady state) or 0 (the round-1 compact kernel) are A/B knobs
    if (!rc) rc = oslam_lba_set_mode(o->ba, getenv("OSLAM_LBA_BATCH_MODE") ? atoi(getenv("OSLAM_LBA_BATCH_MODE")) : (getenv("OSLAM_LBA_BATCH_COMPACT") ? 0 : 1));
    if (!rc) rc = oslam_mappoint_create(&o->mp, dev);
    if (!rc) rc = oslam_frame_create(&o->fr, dev);
    if (!rc) rc = oslam_bow_create(&o->bow, o->cap, dev);
    // One stream per driver handle: the operators run one after the other on the handle's thread, so the solvers and batch matchers use o->strm instead of
    // a stream each (OSLAM_SLAM_OWN_STREAMS=1 restores the separate streams: an A/B knob).
    if (!rc && o->strm && !getenv("OSLAM_SLAM_OWN_STREAMS")) {
        oslam::lba_use_stream(o->ba, o->strm); oslam::lba_use_stream(o->ba1, o->strm);
        oslam::bow_use_stream(o->bow, o->strm); oslam::mappoint_use_stream(o->mp, o->strm);
    }
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
        if (e == hipSuccess && cfg->sensor == 1) e = hipMalloc((void**)&o->d_grayR, o->gray_pitch * cfg->height * S);
        if (e == hipSuccess) e = hipMalloc((void**)&o->d_depth, (size_t)cfg->width * cfg->height * 4 * S);
        if (e == hipSuccess) e = hipMalloc((void**)&o->d_keysUn, sizeof(oslam_keypoint_t) * o->cap * S);
        if (e == hipSuccess) e = hipMalloc((void**)&o->d_keysUn_prev, sizeof(oslam_keypoint_t) * o->cap * S);
        if (e == hipSuccess) e = hipMalloc((void**)&o->d_uRight, 4 * (size_t)o->cap * S);
        if (e == hipSuccess) e = hipMalloc((void**)&o->d_mvDepth, 4 * (size_t)o->cap * S);
        if (e == hipSuccess) e = hipMalloc((void**)&o->d_objbits, (size_t)o->cap * S);
        if (e == hipSuccess) e = hipMalloc((void**)&o->d_status, 64);
        if (e == hipSuccess) e = hipMemset(o->d_status, 0, 64);
        if (e != hipSuccess) { oslam::set_error("slam ops: hipMalloc failed: %s", hipGetErrorString(e)); rc = OSLAM_E_HIP; }
    }
    if (!rc && hipDeviceSynchronize() != hipSuccess) { oslam::set_error("slam ops: device synchronisation failed"); rc = OSLAM_E_HIP; }   // creation-time fills ran on the null stream
    if (rc) { h_destroy(o); return rc; }
    ops->ctx = o;
    ops->max_keypoints = h_max_keypoints; ops->scale_tables = h_scale_tables; ops->image_bounds = h_image_bounds; ops->frames_rgbd = h_frames; ops->frames_rgbd_raw16 = h_frames_raw16;
    ops->search_last = h_search_last; ops->search_local = h_search_local; ops->pose_opt = h_pose_opt; ops->mp_update = h_mp_update; ops->lba = h_lba;
    ops->fuse = h_fuse; ops->bow = h_bow; ops->triangulate = h_triangulate; ops->destroy = h_destroy; ops->frames_stereo = cfg->sensor == 1 ? h_frames_stereo : nullptr;
    ops->kernel_times = h_kernel_times; ops->object_kps = h_object_kps; ops->pose_opt2 = h_pose_opt2;
    const bool deferred_cfg = (cfg->local_mapping & OSLAM_SLAM_LM_DEFERRED) != 0;
    if (deferred_cfg ? !getenv("OSLAM_LBA_NO_SERVICE") : !getenv("OSLAM_LBA_SYNC_OWN_STREAM")) {   // (OSLAM_LBA_NO_SERVICE=1: the deferred schedule with the handle's own solver at collection time)
        o->svc = LbaService::acquire(dev);
        if (!o->svc) { h_destroy(o); return OSLAM_E_HIP; }
        if (deferred_cfg) { ops->lba_submit = h_lba_submit; ops->lba_wait = h_lba_wait; }
    }
    o->mp_tab_on = getenv("OSLAM_SLAM_NO_RESIDENT_POINTS") == nullptr;
    if (o->mp_tab_on) { ops->point_record = h_point_record; ops->resident_points = h_resident_points; }
    if (!getenv("OSLAM_SLAM_NO_WINDOW_UPDATES")) ops->mp_update_windows = h_mp_update_windows;   // (A/B: the MapPoint updates after a local BA through mp_update as before)
    if (!getenv("OSLAM_SLAM_NO_RESIDENT_KF") && !getenv("OSLAM_SLAM_NO_MIRROR")) { ops->map_journal = h_map_journal; ops->kf_culling_counts = h_kf_culling_counts; ops->kf_culling_collect = h_kf_culling_collect; if (o->mp_tab_on && !getenv("OSLAM_SLAM_FUSECUR_HOST")) ops->fuse_into_current = h_fuse_into_current; ops->local_points_list = h_local_points_list; }
    if (!getenv("OSLAM_SLAM_NO_RESIDENT_KF")) { ops->register_keyframes = h_register_keyframes; if (!getenv("OSLAM_SLAM_KEEP_CULLED_RECORDS")) ops->release_keyframes = h_release_keyframes; ops->bow_keyed = h_bow_keyed; ops->fuse_keyed = h_fuse_keyed; ops->mp_update_keyed = h_mp_update_keyed; if (!getenv("OSLAM_SLAM_EAGER_KEYS")) { o->lazy_keys = true; ops->keyframe_raw_keys = h_keyframe_raw_keys; if (!getenv("OSLAM_SLAM_EAGER_DESC")) { o->lazy_desc = true; ops->keyframe_descriptors = h_keyframe_descriptors; ops->frame_descriptors = h_frame_descriptors; } } ops->mp_update_keyed_async = h_mp_update_keyed_async; ops->mp_update_collect = h_mp_update_collect;
        if (!getenv("OSLAM_SLAM_HOST_BOW_NODES")) ops->bow_nodes_keyed = h_bow_nodes_keyed;
        if (o->mp_tab_on && !getenv("OSLAM_SLAM_HOST_FUSE_QUERIES")) ops->fuse_points_keyed = h_fuse_points_keyed; }
    return OSLAM_OK;
}
