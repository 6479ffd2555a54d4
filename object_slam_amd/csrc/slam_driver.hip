// slam_driver.hip — batch-of-sequences Tracking + LocalMapping driver (include/oslam_slam.h).
// Host C++ only: the control flow of the reference's Tracking::Track (src/Tracking.cc:310-587) and LocalMapping::Run
// (src/LocalMapping.cc:48-113) restated as lockstep STAGES over S sequences; every hot-path operator is issued once per
// stage for all sequences through the operator table (slam_ops_hip.hip binds the HIP kernels).  No CPU fallback lives
// here: oslam_slam_create() fails without a HIP device.  Never includes oracle/.
#include <atomic>
#include <chrono>
#include <map>
#include <memory>

#include "common.h"
#include "slam_map.h"
#include "slam_pool.h"

int oslam_slam_make_hip_ops(const oslam_slam_config_t* cfg, oslam_slam_ops_t* out);   // slam_ops_hip.hip

namespace oslam_drv {

// ---- substitute vocabulary (see include/oslam_slam.h): k = 10, two levels -> 100 feature-vector nodes ----
struct Vocab {
    uint64_t top[10][4], sub[10][10][4];
    Vocab() {
        uint64_t s = 0x853c49e6748fea9bULL;   // PCG32, fixed seed
        auto next = [&]() {
            const uint64_t old = s;
            s = old * 6364136223846793005ULL + 1442695040888963407ULL;
            const uint32_t x = (uint32_t)(((old >> 18u) ^ old) >> 27u), r = (uint32_t)(old >> 59u);
            return (x >> r) | (x << ((-r) & 31));
        };
        auto word = [&]() { const uint64_t hi = next(); const uint64_t lo = next(); return (hi << 32) | lo; };   // high half drawn first
        for (int i = 0; i < 10; i++)
            for (int w = 0; w < 4; w++) top[i][w] = word();
        for (int i = 0; i < 10; i++)
            for (int j = 0; j < 10; j++)
                for (int w = 0; w < 4; w++) sub[i][j][w] = word();
    }
    static int dist(const uint64_t* a, const uint64_t* b) {
        return __builtin_popcountll(a[0] ^ b[0]) + __builtin_popcountll(a[1] ^ b[1]) + __builtin_popcountll(a[2] ^ b[2]) + __builtin_popcountll(a[3] ^ b[3]);
    }
    uint32_t node(const uint8_t* d) const {   // DBoW2 tree descent: nearest child, first on ties
        uint64_t v[4];
        memcpy(v, d, 32);
        int b1 = 0, bd = 1 << 30;
        for (int i = 0; i < 10; i++) { const int dd = dist(v, top[i]); if (dd < bd) { bd = dd; b1 = i; } }
        int b2 = 0; bd = 1 << 30;
        for (int j = 0; j < 10; j++) { const int dd = dist(v, sub[b1][j]); if (dd < bd) { bd = dd; b2 = j; } }
        return 11u + (uint32_t)(b1 * 10 + b2);
    }
};

struct Timer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    long long c0 = thread_cpu_ns();
    double cpu = 0;   // core-seconds of the last lap: this thread's CPU time + what the workers spent on its batches
    double lap() {
        auto t1 = std::chrono::steady_clock::now();
        const double d = std::chrono::duration<double>(t1 - t0).count();
        t0 = t1;
        const long long c1 = thread_cpu_ns();
        CpuAccount* a = thread_account();
        cpu = 1e-9 * (double)((c1 - c0) + (a ? a->worker_ns.exchange(0, std::memory_order_relaxed) : 0));
        c0 = c1;
        return d;
    }
};

// flat FeatureVector views for the BoW matchers (see oslam_hip.h): side 1 = (node asc, index order) list, side 2 = CSR over node ids
struct BowViews {
    std::vector<int32_t> q_idx; std::vector<uint32_t> q_node;
    std::vector<uint32_t> nodes; std::vector<int32_t> start, items;
    // Both views order the keypoints by (node id, keypoint index).  Node ids are small integers (the vocabulary substitute has 100 leaves, DBoW2's level-4
    // nodes of the reference vocabulary 10^4), so this is a stable counting sort by node, not a comparison sort of 1000-2000 pairs per keyframe.
    static void by_node(const std::vector<uint32_t>& node, std::vector<int32_t>& order, std::vector<int32_t>& count /* [max node + 2], prefix sums on return */) {
        const int N = (int)node.size();
        uint32_t mx = 0;
        for (int i = 0; i < N; i++) mx = std::max(mx, node[i]);
        order.resize(N);
        if (mx > (1u << 20)) {   // not a small-id vocabulary: comparison sort
            std::vector<std::pair<uint32_t, int>> v(N);
            for (int i = 0; i < N; i++) v[i] = std::make_pair(node[i], i);
            std::sort(v.begin(), v.end());
            for (int i = 0; i < N; i++) order[i] = v[i].second;
            count.clear();
            return;
        }
        count.assign((size_t)mx + 2, 0);
        for (int i = 0; i < N; i++) count[node[i] + 1]++;
        for (size_t k = 1; k < count.size(); k++) count[k] += count[k - 1];
        std::vector<int32_t> cur(count.begin(), count.end() - 1);
        for (int i = 0; i < N; i++) order[cur[node[i]]++] = i;
    }
    void side1(const std::vector<uint32_t>& node) {
        const int N = (int)node.size();
        std::vector<int32_t> cnt;
        by_node(node, q_idx, cnt);
        q_node.resize(N);
        for (int i = 0; i < N; i++) q_node[i] = node[q_idx[i]];
    }
    void side2(const std::vector<uint32_t>& node) {
        const int N = (int)node.size();
        std::vector<int32_t> cnt;
        by_node(node, items, cnt);
        nodes.clear(); start.clear();
        for (int i = 0; i < N; i++)
            if (i == 0 || node[items[i]] != node[items[i - 1]]) { nodes.push_back(node[items[i]]); start.push_back(i); }
        start.push_back(N);
    }
};

// (local-BA windows: the solver's reduced system holds 6 x 128 unknowns, i.e. at most 128 FREE keyframes; fixed keyframes are not limited.  OSLAM_SLAM_LBA_MAX_FREE
// lowers the bound — a test knob for the degraded-window path of run_local_mapping)
static const int kLbaMaxFreeKFs = [] { const char* e = getenv("OSLAM_SLAM_LBA_MAX_FREE"); const int v = e ? atoi(e) : 0; return (v > 0 && v < 128) ? v : 128; }();

enum { ST_NOT_INITIALIZED = OSLAM_SLAM_NOT_INITIALIZED, ST_OK = OSLAM_SLAM_OK, ST_LOST = OSLAM_SLAM_LOST };

struct Ctx;
struct MpUpdate;

// Per-sequence Tracking + LocalMapping state (include/Tracking.h, include/LocalMapping.h members).
struct Seq {
    Map map;
    Frame fa, fb;
    Frame* cur = &fa;
    Frame* last = &fb;
    int state = ST_NOT_INITIALIZED;
    int nextFrameId = 0;
    bool hasVelocity = false;
    M4 velocity;
    int refKF = -1, lastKFFrameId = 0, lastRelocFrameId = 0;
    int matchesInliers = 0;
    std::vector<int> localKFs, localMPs;
    std::vector<uint64_t> mpMark;         // mnTrackReferenceForFrame of the map points as ONE BIT per point id: set <=> the point was visited by the cached walk (update_local_map)
    std::vector<int> baMark;              // mnBALocalForKF per map point id (dense, see the local-BA gather in run_local_mapping)
    std::vector<int> fuseMark;            // mnFuseCandidateForKF per map point id (dense, SearchInNeighbors' second direction)
    // UpdateLocalKeyFrames' keyframeCounter kept from frame to frame (update_local_map): vote[k] = matched points of the last voted frame that keyframe k observes,
    // votePts = those points (one entry per keypoint), valid while voteVersion == mapVersion (observation lists only change under a version bump)
    std::vector<int> vote, votePts, votePrev;
    std::vector<int8_t> voteDelta;        // dense by map point id, all zero between calls
    long long voteVersion = -1;
    // Local-map cache: mvpLocalMapPoints is a function of the ordered local keyframe list and of the map, and the map only changes when this sequence
    // creates a keyframe or its local mapping runs (mapVersion counts both).  A frame whose list and version equal the cached ones reuses the walk of
    // the keyframes' map points AND the packed SearchLocalPoints arrays (whose copy the operator table may keep resident: content id).
    long long mapVersion = 0, locVersion = -1, locContentId = 0;
    int locWalkFrame = -1;                // frame id of the cached walk (-1: none)
    bool locReused = false;
    bool locListDev = false, locListPending = false;   // UpdateLocalPoints' walk is the operator table's (oslam_slam_ops_t::local_points_list); this frame's list is still to come
    int trkKF = -1, trkMinObs = 0, trkCount = 0; long long trkVersion = -1;   // cached KeyFrame::TrackedMapPoints of the reference keyframe (NeedNewKeyFrame)
    std::vector<int> locKFs, mpPos;       // cached keyframe list; position of a visited point in localMPs (-1: visited but bad)
    std::vector<int> seenList;            // points given mnLastFrameSeen = this frame outside SearchLocalPoints (outliers of the initial pose optimisation)
    std::vector<float> locPw, locPn, locMax, locMin;
    std::vector<uint8_t> locObs, locDesc, jSkip;
    std::vector<RelPose> rel;
    std::vector<int> recentAdded;          // mlpRecentAddedMapPoints
    std::vector<int> newKFs;               // mlNewKeyFrames (at most one per step)
    std::vector<int> pendingKF;            // keyframes created in this step, not yet announced to the operator table (register_keyframes)
    std::vector<int> culledKFs;            // keyframes culled in this pass, not yet announced to the operator table (release_keyframes)
    std::vector<int> counter;              // scratch, indexed by keyframe id
    int64_t st[16] = {0};
    // per-step scratch
    int path = 0;                          // 0 none, 1 motion model, 2 reference keyframe
    bool ok = false;
    int curKF = -1;                        // keyframe being processed by local mapping this step
    std::vector<float> jXw, jObs, jInv;
    std::vector<uint8_t> jHas, jDesc, jBlocked, jInView, jOutlier;
    std::vector<int> jMatch;
    oslam_job_search_last_t jSL; oslam_job_search_local_t jLoc; oslam_job_pose_t jPose;
    bool hasSL = false, hasLoc = false;
    // object layer substitute (include/oslam_slam.h head comment): Object3Ds keyed by the caller's track id
    struct Obj3D { int track = -1, updateCnt = 0; std::vector<int> mps; std::vector<uint8_t> member; };   // mvpMapPoints in insertion order + membership by point id
    std::vector<Obj3D> obj3ds;
    std::map<int, int> objOfTrack;
    int64_t sem[8] = {0};
    int64_t lbaWindowsDegraded = 0;           // local-BA windows DEGRADED because they had more than 128 free keyframes (the operator's bound): the weakest covisible keyframes entered as fixed cameras
    int64_t lbaWin[4] = {0, 0, 0, 0};      // local-BA window sizes summed over the sequence's windows: local keyframes, fixed keyframes, points, edges
    std::vector<uint8_t> jInMask;                       // object_kps output
    std::vector<const uint8_t*> jMaskPtrs;              // masks of the matched objects (idx_obj order)
    std::vector<float> jObjXw; std::vector<int32_t> jObjOf, jJointKp, jJointObj, jObjIds;
    oslam_job_pose2_t jPose2; bool hasPose2 = false;
    const oslam_slam_objects_t* det = nullptr;           // this step's detections (NULL or n == 0: none)
    bool lazyDesc = false;                // the same for mDescriptors (keyframe_descriptors / frame_descriptors)
    bool lazyKeys = false;                // the table hands mvKeys out for keyframes only (oslam_slam_ops_t::keyframe_raw_keys)
    int64_t opFailures = 0;               // operator errors confined to this sequence (its map was reset: local_mapping_back)
    bool resetRequested = false;          // System::Reset() asked by Track() (lost with <= 5 keyframes, reference src/Tracking.cc:553-561)
    void reset() {                        // Tracking::Reset (:1769-1815) + LocalMapping::ResetIfRequested + Map::clear; id counters restart at 0
        const bool jr_on = map.jrOn;
        map = Map();
        map.jrOn = jr_on; map.jrReset = jr_on;   // (device mirror of the observation graph: oslam_slam_ops_t::map_journal)
        state = ST_NOT_INITIALIZED; nextFrameId = 0; refKF = -1;
        localKFs.clear(); localMPs.clear(); rel.clear(); recentAdded.clear(); newKFs.clear(); kfBow.clear(); pendingKF.clear(); culledKFs.clear();
        obj3ds.clear(); objOfTrack.clear();   // Map::clear() drops the Object3Ds too; the counters in sem[] run on like N_AllSemanticConstraintNum
        std::fill(counter.begin(), counter.end(), 0);
        std::fill(mpMark.begin(), mpMark.end(), 0ull);
        std::fill(baMark.begin(), baMark.end(), 0);
        std::fill(fuseMark.begin(), fuseMark.end(), 0);
        voteVersion = -1; votePts.clear();
        mapVersion++; locVersion = -1; locWalkFrame = -1; locKFs.clear(); locListPending = false;
        resetRequested = false;
    }
    std::vector<int> updList;             // points created by tracking this step (descriptor / normal pending)
    std::vector<std::unique_ptr<BowViews>> kfBow;   // FeatureVector views per keyframe id (built once: the descriptors are immutable)
    BowViews& bow_views(const Ctx& c, int kf);
};

struct Ctx {
    oslam_slam_config_t cfg;
    oslam_slam_ops_t ops;
    int S = 0, cap = 0;
    float scale[OSLAM_MAX_LEVELS], invScale[OSLAM_MAX_LEVELS], sigma2[OSLAM_MAX_LEVELS], invSigma2[OSLAM_MAX_LEVELS];
    float bounds[4];
    float invfx, invfy, thDepth, mb, logScale;
    int maxFrames, minFrames;
    bool stereo = false;
    Vocab voc;
    std::vector<std::unique_ptr<Seq>> seq;
    std::unique_ptr<Pool> pool;
    double sec[16] = {0};
    double cpu[16] = {0};   // core-seconds of the same stages (host thread + workers)
    double fine[8] = {0};   // OSLAM_SLAM_SN_STATS: core-seconds inside SearchInNeighbors (target lists + masks, first-direction rounds, second-direction list, its round, updates + connections)
    CpuAccount acct;
    int shard = 0;   // worker set of this handle (slam_pool.h)
    struct Win {   // one local-BA window (run_local_mapping)
        int si = -1, nLocal = 0, nFree = 0;   // nFree: window indices below it are free poses (== nLocal unless the window was degraded)
        std::vector<int> kfs, pts; std::vector<float> poses, points, eobs, einv, poses_out, points_out; std::vector<uint8_t> fixed, erase;
        std::vector<int32_t> ekf, ept; std::vector<std::pair<int, int>> eref;
        // for the MapPoint updates after the solve (oslam_job_mp_window_t): first edge of every point, octave of every edge's keypoint, points with an observation
        // outside the window (in a culled keyframe), and the job's own arrays
        std::vector<int32_t> pstart, uref; std::vector<uint8_t> eoct, pquirk, uskip; std::vector<float> ulsf, uOw, uout5;
        int32_t st[4] = {0, 0, 0, 0};   // the solver's statistics of the window; st[0] < 0: the operator refused THIS window (st[1] = its error code), see local_mapping_back
        void reset() { kfs.clear(); pts.clear(); poses.clear(); points.clear(); eobs.clear(); einv.clear(); poses_out.clear(); points_out.clear(); fixed.clear(); erase.clear(); ekf.clear(); ept.clear(); eref.clear(); nLocal = 0; }
    };
    std::vector<Win> winPool;
    // deferred schedule (OSLAM_SLAM_LM_DEFERRED): the second half of the local-mapping pass of the previous step — local-BA solve in flight or not yet started,
    // write-back, MapPoint updates, KeyFrameCulling — applied by finish_local_mapping after the next step's tracking
    struct PendingLM { bool active = false, submitted = false; std::vector<int> who; std::vector<Win*> wins; std::vector<oslam_lba_problem_t> probs; } pend;
    std::unique_ptr<MpUpdate> updTrack, updMap;   // batched MapPoint updates of the two halves of a step (arrays keep their capacity)
    int mapStep = 0;            // local-mapping passes of this handle (MapPt::updStep)
    std::vector<Map::JrScratch> jrScratch;   // per sequence: the arrays of the change set handed to oslam_slam_ops_t::map_journal in this pass (kept for their capacity)
    int injectLbaFailure = -1;  // oslam_slam_inject_failure: sequence whose next local-BA window is made invalid (tests of the per-sequence failure isolation)
    bool residentPts = false;   // the operator table serves pose jobs from map-point ids (oslam_slam_ops_t::resident_points)
    std::atomic<long long> badKFObs{0};   // observations in culled keyframes left out by ComputeDistinctiveDescriptors (oslam_slam_bad_keyframe_observations)
    std::atomic<long long> contentCounter{0}, locReuse{0}, locFrames{0};   // content ids of the packed local maps; frames that reused theirs / all tracked frames
    long long next_content_id() { return contentCounter.fetch_add(1, std::memory_order_relaxed) + 1; }
    int mask_stride = 0, masks_on_device = 0;
};

}  // namespace oslam_drv

using namespace oslam_drv;

struct oslam_slam { Ctx c; };

namespace oslam_drv {

static void compute_bow(const Ctx& c, int N, const uint8_t* desc, std::vector<uint32_t>& out) {
    if (!out.empty()) return;   // Frame::ComputeBoW / KeyFrame::ComputeBoW: only once
    out.resize(N);
    for (int i = 0; i < N; i++) out[i] = c.voc.node(desc + (size_t)i * 32);
}

BowViews& Seq::bow_views(const Ctx& c, int kf) {
    if ((int)kfBow.size() <= kf) kfBow.resize(kf + 1);
    if (!kfBow[kf]) {
        KeyFrm& k = map.kfs[kf];
        compute_bow(c, k.N, k.desc.data(), k.bowNode);
        kfBow[kf].reset(new BowViews);
        kfBow[kf]->side1(k.bowNode);
        kfBow[kf]->side2(k.bowNode);
    }
    return *kfBow[kf];
}

// Frame::UnprojectStereo (src/Frame.cc:904-919): mRwc*x3Dc + mOw (gemm small-matrix branch with C)
static bool unproject_frame(const Ctx& c, const Frame& f, int i, float out[3]) {
    const float z = f.depth[i];
    if (!(z > 0)) return false;
    const float u = f.keysUn[i].x, v = f.keysUn[i].y;
    const float x = (u - c.cfg.cx) * z * c.invfx, y = (v - c.cfg.cy) * z * c.invfy;
    for (int r = 0; r < 3; r++) {
        float s = f.pose.Rwc[r * 3] * x;
        s += f.pose.Rwc[r * 3 + 1] * y;
        s += f.pose.Rwc[r * 3 + 2] * z;
        out[r] = (float)((double)s + (double)f.pose.Ow[r]);
    }
    return true;
}

// KeyFrame::KeyFrame(Frame&, ...) (src/KeyFrame.cc:30-58)
static int new_keyframe(Seq& s, const Frame& f, float thDepth) {
    s.mapVersion++;
    s.map.kfs.emplace_back();
    KeyFrm& k = s.map.kfs.back();
    k.id = (int)s.map.kfs.size() - 1;
    k.frameId = f.id; k.stamp = f.stamp; k.N = f.N;
    if (!s.lazyKeys) k.keys.assign(f.keys.begin(), f.keys.begin() + f.N);   // (else: fetched for the new keyframes only, after register_keyframes)
    k.keysUn.assign(f.keysUn.begin(), f.keysUn.begin() + f.N);
    k.oct.resize(f.N);
    for (int i = 0; i < f.N; i++) k.oct[i] = (uint8_t)f.keysUn[i].octave;   // (KeyFrameCulling reads the octaves alone: 1 byte per keypoint instead of a pass over the 28-byte records)
    if (!s.lazyDesc) k.desc.assign(f.desc.begin(), f.desc.begin() + (size_t)f.N * 32);   // (else: after register_keyframes, like mvKeys)
    k.uRight.assign(f.uRight.begin(), f.uRight.begin() + f.N);
    k.depth.assign(f.depth.begin(), f.depth.begin() + f.N);
    k.mp.assign(f.mp.begin(), f.mp.begin() + f.N);
    k.bowNode = f.bowNode;
    k.pose.set_keyframe(f.pose.Tcw);
    k.Tcp = eye4();
    s.counter.resize(s.map.kfs.size() + 8, 0);
    s.st[1]++;
    s.pendingKF.push_back(k.id);
    (void)thDepth;
    s.map.jr_new_kf(k.id);
    return k.id;
}

// ---- batched MapPoint::ComputeDistinctiveDescriptors / UpdateNormalAndDepth over points of several sequences ----
struct MpUpdate {
    struct Item { int seq, p; };
    std::vector<Item> items;
    std::vector<int> start;
    std::vector<uint8_t> odesc, outdesc;
    std::vector<int32_t> okey;
    std::vector<float> oOw, pos, owref, lsf, out5;
    std::vector<int> best;
    void clear() { items.clear(); }   // (never while a job is pending: finish() first)
    void add(int seq, int p) { items.push_back({seq, p}); }
    std::vector<int> dstart;
    std::vector<int32_t> ikey;
    // OSLAM_MPU_PROF=1: wall-clock split of run() summed over all calls of the process (count / pack / operator / scatter), printed at exit
    struct Prof {
        std::atomic<long long> ns[4], calls{0}, pts{0}, obs{0};
        Prof() { for (auto& x : ns) x = 0; }
        ~Prof() {
            if (calls.load()) fprintf(stderr, "[mp_update prof] %lld calls, %lld points, %lld observations: count %.1f ms, pack %.1f ms, operator %.1f ms, scatter %.1f ms\n", calls.load(), pts.load(),
                                      obs.load(), ns[0] * 1e-6, ns[1] * 1e-6, ns[2] * 1e-6, ns[3] * 1e-6);
        }
    };
    static Prof* prof() { static Prof* p = getenv("OSLAM_MPU_PROF") ? new Prof : nullptr; static struct D { ~D() { delete prof(); } } d; return p; }
    // submit(..., defer = true) + finish(): the operator call may return before its results are there (oslam_slam_ops_t::mp_update_keyed_async); finish() waits for
    // them and scatters them into the maps.  Nothing may touch the items' observation lists or this object in between.  run() = both at once.
    bool pending = false, pDesc = false, pNormal = false;
    oslam_job_mp_update_t pjob;
    int run(Ctx& c, bool do_desc, bool do_normal) {
        const int rc = submit(c, do_desc, do_normal, false);
        return rc ? rc : finish(c);
    }
    int submit(Ctx& c, bool do_desc, bool do_normal, bool defer) {
        if (pending) { const int rc0 = finish(c); if (rc0) return rc0; }
        const int P = (int)items.size();
        if (P == 0) return OSLAM_OK;
        Prof* pf = prof();
        auto t0_ = std::chrono::steady_clock::now();
        auto lap_ = [&](int k) { if (pf) { auto t1_ = std::chrono::steady_clock::now(); pf->ns[k] += std::chrono::duration_cast<std::chrono::nanoseconds>(t1_ - t0_).count(); t0_ = t1_; } };
        // Two observation lists per point: UpdateNormalAndDepth reads every observation (src/MapPoint.cc:441-453), ComputeDistinctiveDescriptors only those
        // whose keyframe is not bad (:362-368).  They differ when a point still observes a culled keyframe: KeyFrame::SetBadFlag erases the observations of
        // the keyframe's own mvpMapPoints only, and a keypoint's slot can have gone to another point (two new points triangulated against the same neighbour
        // keypoint, src/LocalMapping.cc:440-446).
        // (the counts on the shared workers — after a local BA this pass walks ~10^5 point records and ~10^6 observations at random — then a serial prefix sum)
        start.assign(P + 1, 0); dstart.assign(P + 1, 0);
        const int chunk = 256, nchunks = (P + chunk - 1) / chunk;
        c.pool->parallel_for(nchunks, [&](int ch) {
            const int i0 = ch * chunk, i1 = std::min(P, i0 + chunk);
            for (int i = i0; i < i1; i++) {
                if (i + kPF < i1) prefetch_mp(&c.seq[items[i + kPF].seq]->map.mps[items[i + kPF].p]);
                if (i + kPF / 2 < i1) __builtin_prefetch(c.seq[items[i + kPF / 2].seq]->map.mps[items[i + kPF / 2].p].obs.data());
                const Map& m = c.seq[items[i].seq]->map;
                const MapPt& p = m.mps[items[i].p];
                int n = 0, nd = 0;
                if (!m.pBad[items[i].p]) { n = (int)p.obs.size(); for (auto& e : p.obs) nd += !m.kfs[e.first].bad; }
                start[i + 1] = n; dstart[i + 1] = nd;
            }
        });
        for (int i = 0; i < P; i++) { start[i + 1] += start[i]; dstart[i + 1] += dstart[i]; }
        lap_(0);
        const size_t total = (size_t)start[P], dtotal = (size_t)dstart[P];
        const bool split = dtotal != total;
        if (do_desc) c.badKFObs.fetch_add((long long)(total - dtotal), std::memory_order_relaxed);
        const bool keyed = c.ops.mp_update_keyed != nullptr && do_desc;   // the table gathers the descriptors from its resident keyframes
        if (!keyed) odesc.resize(do_desc ? std::max<size_t>(dtotal, 1) * 32 : 32);   // (UpdateNormalAndDepth alone reads no descriptors)
        else okey.resize(std::max<size_t>(dtotal, 1) * 3);
        oOw.resize(std::max<size_t>(total, 1) * 3);
        pos.resize((size_t)P * 3); owref.resize((size_t)P * 3); lsf.resize(P);
        c.pool->parallel_for(nchunks, [&](int ch) {
            const int i0 = ch * chunk, i1 = std::min(P, i0 + chunk);
            for (int i = i0; i < i1; i++) {
                if (i + kPF < i1) prefetch_mp(&c.seq[items[i + kPF].seq]->map.mps[items[i + kPF].p]);
                if (i + kPF / 2 < i1) __builtin_prefetch(c.seq[items[i + kPF / 2].seq]->map.mps[items[i + kPF / 2].p].obs.data());
                if (do_normal && i + kPF / 2 < i1) __builtin_prefetch(c.seq[items[i + kPF / 2].seq]->map.mps[items[i + kPF / 2].p].okp.data());   // (the reference keyframe's octave is read from the list beside obs)
                const Map& m = c.seq[items[i].seq]->map;
                const MapPt& p = m.mps[items[i].p];
                const int n = start[i + 1] - start[i];
                size_t at = (size_t)start[i], dat = (size_t)dstart[i];
                if (n > 0)
                    for (auto& e : p.obs) {
                        const KeyFrm& k = m.kfs[e.first];
                        if (do_desc && !k.bad) {
                            if (keyed) { okey[dat * 3] = items[i].seq; okey[dat * 3 + 1] = e.first; okey[dat * 3 + 2] = e.second; }
                            else memcpy(&odesc[dat * 32], &k.desc[(size_t)e.second * 32], 32);
                            dat++;
                        }
                        oOw[at * 3] = k.pose.Ow[0]; oOw[at * 3 + 1] = k.pose.Ow[1]; oOw[at * 3 + 2] = k.pose.Ow[2];
                        at++;
                    }
                for (int d = 0; d < 3; d++) pos[(size_t)i * 3 + d] = p.pos[d];
                float lf = 1.f;
                const float* ow = p.pos;
                if (n > 0 && p.refKF >= 0) {
                    const KeyFrm& rk = m.kfs[p.refKF];
                    ow = rk.pose.Ow;
                    for (size_t oi = 0; oi < p.obs.size(); oi++)
                        if (p.obs[oi].first == p.refKF) { lf = c.scale[p.okp[oi].octave]; break; }   // mvKeysUn[idx].octave of the reference keyframe's observation (src/MapPoint.cc:455-457), cached in ObsKp
                }
                for (int d = 0; d < 3; d++) owref[(size_t)i * 3 + d] = ow[d];
                lsf[i] = lf;
            }
        });
        lap_(1);
        best.resize(P); outdesc.resize((size_t)P * 32); out5.resize((size_t)P * 5);
        oslam_job_mp_update_t& j = pjob;
        j.P = P; j.obs_start = start.data(); j.obs_desc = keyed ? nullptr : odesc.data(); j.obs_Ow = oOw.data(); j.Pos = pos.data(); j.OwRef = owref.data();
        j.levelScaleFactor = lsf.data(); j.do_desc = do_desc; j.do_normal = do_normal; j.best_idx = best.data(); j.out_desc = outdesc.data(); j.out5 = out5.data();
        j.desc_start = split ? dstart.data() : nullptr;
        ikey.resize((size_t)P * 2);
        for (int i = 0; i < P; i++) { ikey[2 * (size_t)i] = items[i].seq; ikey[2 * (size_t)i + 1] = items[i].p; }
        j.items = ikey.data();
        const bool async = defer && keyed && c.ops.mp_update_keyed_async && c.ops.mp_update_collect;
        const int rc = async ? c.ops.mp_update_keyed_async(c.ops.ctx, &j, okey.data()) : keyed ? c.ops.mp_update_keyed(c.ops.ctx, &j, okey.data()) : c.ops.mp_update(c.ops.ctx, &j);
        if (rc) return rc;
        lap_(2);
        pending = true; pDesc = do_desc; pNormal = do_normal; pAsync = async;
        if (pf) { pf->calls++; pf->pts += P; pf->obs += (long long)total; }
        return OSLAM_OK;
    }
    bool pAsync = false;
    struct SpecStats { std::atomic<long long> upd{0}, changed{0}, changed_in_list{0}; ~SpecStats() { fprintf(stderr, "[fuse spec stats] per-round descriptor updates %lld, changed %lld, changed and in the current keyframe's candidate list %lld\n", upd.load(), changed.load(), changed_in_list.load()); } };
    int finish(Ctx& c) {
        static SpecStats* spec_stats = getenv("OSLAM_SLAM_SPEC_STATS") ? new SpecStats : nullptr;
        static struct SpecAtExit { ~SpecAtExit() { delete spec_stats; } } spec_at_exit;
        if (!pending) return OSLAM_OK;
        pending = false;
        if (pAsync) { const int rc = c.ops.mp_update_collect(c.ops.ctx); if (rc) return rc; }
        const bool do_desc = pDesc, do_normal = pNormal;
        const int P = (int)items.size();
        const int chunk = 256, nchunks = (P + chunk - 1) / chunk;
        Prof* pf = prof();
        auto t0_ = std::chrono::steady_clock::now();
        auto lap_ = [&](int k) { if (pf) { auto t1_ = std::chrono::steady_clock::now(); pf->ns[k] += std::chrono::duration_cast<std::chrono::nanoseconds>(t1_ - t0_).count(); t0_ = t1_; } };
        // (an item can be listed twice after fusions; both copies carry the same result, so concurrent writers store the same bytes)
        c.pool->parallel_for(nchunks, [&](int ch) {
            const int i0 = ch * chunk, i1 = std::min(P, i0 + chunk);
            for (int i = i0; i < i1; i++) {
                if (start[i + 1] == start[i]) continue;   // no observations: both methods return early
                MapPt& p = c.seq[items[i].seq]->map.mps[items[i].p];
                if (spec_stats && do_desc && !do_normal && dstart[i + 1] > dstart[i]) {   // (how often a Fuse round changes the descriptor of a point that later rounds still search)
                    spec_stats->upd++;
                    const Seq& sq = *c.seq[items[i].seq];
                    if (memcmp(p.desc, &outdesc[(size_t)i * 32], 32) != 0) { spec_stats->changed++; if (p.fuseListStamp == sq.curKF + 1) spec_stats->changed_in_list++; }
                }
                if (do_desc && dstart[i + 1] > dstart[i]) memcpy(p.desc, &outdesc[(size_t)i * 32], 32);   // every observing keyframe bad: the descriptor stays (:370-371)
                if (do_normal) {
                    const float* o = &out5[(size_t)i * 5];
                    p.normal[0] = o[0]; p.normal[1] = o[1]; p.normal[2] = o[2]; p.maxD = o[3]; p.minD = o[4];
                }
                if (do_desc && do_normal) { p.updVer = p.obsVer; p.updStep = c.mapStep; }   // (nothing touches the observation lists while a batch runs)
            }
        });
        lap_(3);
        return OSLAM_OK;
    }
};

// Creates the stereo points of StereoInitialization (:603-619) / CreateNewKeyFrame (:1338-1398) for frame f, keyframe kf.
static void create_stereo_points(Ctx& c, Seq& s, Frame& f, int kf, bool all) {
    Map& m = s.map;
    auto make = [&](int i) {
        float x[3];
        if (!unproject_frame(c, f, i, x)) return;
        const int p = m.new_point(x, kf, m.kfs[kf].frameId);
        m.add_observation(p, kf, i);
        m.set_kf_mp(kf, i, p);
        m.nMPsInMap++; s.st[3]++;
        f.mp[i] = p;
        s.updList.push_back(p);
    };
    if (all) {
        for (int i = 0; i < f.N; i++) if (f.depth[i] > 0) make(i);
        return;
    }
    static thread_local std::vector<std::pair<float, int>> v;   // scratch
    v.clear();
    for (int i = 0; i < f.N; i++) if (f.depth[i] > 0) v.push_back(std::make_pair(f.depth[i], i));
    if (v.empty()) return;
    std::sort(v.begin(), v.end());
    int nPoints = 0;
    for (size_t j = 0; j < v.size(); j++) {
        const int i = v[j].second;
        bool create = false;
        const int p = f.mp[i];
        if (p < 0) create = true;
        else if (m.pNObs[p] < 1) { create = true; f.mp[i] = -1; }
        if (create) make(i);
        nPoints++;
        if (v[j].first > c.thDepth && nPoints > 100) break;
    }
}

// Tracking::UpdateLocalKeyFrames (:1496-1604) + UpdateLocalPoints (:1470-1493)
static void update_local_map(Seq& s) {
    Map& m = s.map;
    Frame& f = *s.cur;
    // keyframeCounter (:1499-1517): for every keypoint with a map point, +1 for every keyframe that observes the point.  The matched set of consecutive frames
    // differs by 10-20 % and the observation lists only change under a map-version bump, so the counts are kept and only the points that entered or left the
    // set since the last voted frame walk their observation lists (a full recount after a version change).  OSLAM_SLAM_VOTE_CHECK=1 compares with the recount.
    s.votePrev.swap(s.votePts);
    s.votePts.clear();
    for (int i = 0; i < f.N; i++) {
        const int p = f.mp[i];
        if (p < 0) continue;
        if (m.pBad[p]) { f.mp[i] = -1; continue; }
        s.votePts.push_back(p);
    }
    if (s.vote.size() < m.kfs.size()) s.vote.resize(m.kfs.size() + 16, 0);
    const std::vector<int>& cur = s.votePts;
    if (s.voteVersion != s.mapVersion) {   // recount
        std::fill(s.vote.begin(), s.vote.end(), 0);
        for (size_t i = 0; i < cur.size(); i++) {
            prefetch_obs_ahead(m.mps, cur, i, cur.size());
            for (auto& e : m.mps[cur[i]].obs) s.vote[e.first]++;
        }
        s.voteVersion = s.mapVersion;
    } else {
        if (s.voteDelta.size() < m.mps.size()) s.voteDelta.resize(m.mps.size() + m.mps.size() / 2 + 64, 0);
        int8_t* dl = s.voteDelta.data();
        for (int p : s.votePrev) dl[p]--;
        for (int p : cur) dl[p]++;
        static thread_local std::vector<std::pair<int, int>> chg;   // (point, change of its multiplicity) for the points that entered or left the matched set
        chg.clear();
        auto collect = [&](const std::vector<int>& pts) {
            for (int p : pts) { const int d = dl[p]; if (d != 0) { dl[p] = 0; chg.push_back(std::make_pair(p, d)); } }
        };
        collect(s.votePrev);
        collect(cur);
        const size_t nc = chg.size();
        for (size_t i = 0; i < nc; i++) {   // (records and lists of later entries requested ahead: every one of them is a cache miss)
            if (i + kPF < nc) __builtin_prefetch(&m.mps[chg[i + kPF].first].obs);
            if (i + kPF / 2 < nc) __builtin_prefetch(m.mps[chg[i + kPF / 2].first].obs.data());
            const int d = chg[i].second;
            for (auto& e : m.mps[chg[i].first].obs) s.vote[e.first] += d;
        }
    }
    static const bool vote_check = getenv("OSLAM_SLAM_VOTE_CHECK") != nullptr;
    if (vote_check) {
        std::vector<int> ref(s.vote.size(), 0);
        for (int p : cur) for (auto& e : m.mps[p].obs) ref[e.first]++;
        if (ref != s.vote) { fprintf(stderr, "OSLAM_SLAM_VOTE_CHECK: the kept keyframe counter differs from the recount (frame %d)\n", f.id); abort(); }
    }
    bool any = false;
    for (size_t k = 0; k < m.kfs.size() && !any; k++) any = s.vote[k] > 0;
    if (any) {
        int mx = 0, kmax = -1;
        s.localKFs.clear();
        for (int k = 0; k < (int)m.kfs.size(); k++) {   // (ascending keyframe id: the order of the reference's map, see slam_map.h)
            const int cnt = s.vote[k];
            if (cnt == 0) continue;
            if (m.kfs[k].bad) continue;
            if (cnt > mx) { mx = cnt; kmax = k; }
            s.localKFs.push_back(k);
            m.kfs[k].trackRefForFrame = f.id;
        }
        const size_t n0 = s.localKFs.size();
        for (size_t q = 0; q < n0; q++) {
            if (s.localKFs.size() > 80) break;
            const int k = s.localKFs[q];
            for (int nb : m.best_covisibles(k, 10))
                if (!m.kfs[nb].bad && m.kfs[nb].trackRefForFrame != f.id) { s.localKFs.push_back(nb); m.kfs[nb].trackRefForFrame = f.id; break; }
            for (int ch : m.kfs[k].children)
                if (!m.kfs[ch].bad && m.kfs[ch].trackRefForFrame != f.id) { s.localKFs.push_back(ch); m.kfs[ch].trackRefForFrame = f.id; break; }
            const int par = m.kfs[k].parent;
            if (par >= 0 && m.kfs[par].trackRefForFrame != f.id) {
                s.localKFs.push_back(par);
                m.kfs[par].trackRefForFrame = f.id;
                break;   // the reference leaves the loop here (:1593)
            }
        }
        if (kmax >= 0) { s.refKF = kmax; f.refKF = kmax; }
    }
    s.locReused = s.locVersion == s.mapVersion && s.locWalkFrame >= 0 && s.localKFs == s.locKFs;
    if (s.locReused) return;   // same ordered keyframe list over an unchanged map: the walk below would rebuild the same list
    if (s.locListDev) { s.locListPending = true; return; }   // (the table walks the keyframes' lists where they are resident: stage_local_map_lists)
    // mnTrackReferenceForFrame of the map points as a per-sequence BITMAP, cleared per walk: the loop below visits 20-80 k keyframe slots and most of them hit
    // an already marked point; one bit per point id keeps the marks of a whole map (~20 k points) in 2.5 KB — first-level cache — where a stamped int per point
    // was an 80 KB array that every other sequence's walk had evicted.  A bad point is marked too (it is never pushed either way), which leaves the list unchanged.
    {
        const size_t nw = (m.mps.size() + 63) / 64;
        if (s.mpMark.size() < nw) s.mpMark.resize(nw + nw / 2 + 4, 0ull);
        if (s.mpPos.size() < m.mps.size()) s.mpPos.resize(m.mps.size() + m.mps.size() / 2 + 64, -1);
        std::fill(s.mpMark.begin(), s.mpMark.end(), 0ull);
    }
    uint64_t* mark = s.mpMark.data();
    int* pos = s.mpPos.data();
    s.localMPs.clear();
    for (int k : s.localKFs) {
        const KeyFrm& kf = m.kfs[k];
        const int* kmp = kf.mp.data();
        for (int i = 0; i < kf.N; i++) {
            const int p = kmp[i];
            if (p < 0) continue;
            const uint64_t bit = 1ull << (p & 63);
            uint64_t& wd = mark[p >> 6];
            if (wd & bit) continue;
            wd |= bit;
            if (!m.pBad[p]) { pos[p] = (int)s.localMPs.size(); s.localMPs.push_back(p); }
            else pos[p] = -1;
        }
    }
    s.locKFs = s.localKFs; s.locVersion = s.mapVersion; s.locWalkFrame = f.id;
}

// the walk above, on the host, for one sequence whose list the table could not deliver (overflow) — and the bookkeeping around a delivered list
static void local_points_walk_host(Seq& s) {
    const bool dev = s.locListDev;
    s.locListDev = false; s.locListPending = false;
    // (re-enter update_local_map's tail: the vote and the keyframe list are done; votePts were swapped by that call, so only the walk is repeated here)
    Map& m = s.map;
    Frame& f = *s.cur;
    const size_t nw = (m.mps.size() + 63) / 64;
    if (s.mpMark.size() < nw) s.mpMark.resize(nw + nw / 2 + 4, 0ull);
    if (s.mpPos.size() < m.mps.size()) s.mpPos.resize(m.mps.size() + m.mps.size() / 2 + 64, -1);
    std::fill(s.mpMark.begin(), s.mpMark.end(), 0ull);
    uint64_t* mark = s.mpMark.data();
    int* pos = s.mpPos.data();
    s.localMPs.clear();
    for (int k : s.localKFs) {
        const KeyFrm& kf = m.kfs[k];
        const int* kmp = kf.mp.data();
        for (int i = 0; i < kf.N; i++) {
            const int p = kmp[i];
            if (p < 0) continue;
            const uint64_t bit = 1ull << (p & 63);
            uint64_t& wd = mark[p >> 6];
            if (wd & bit) continue;
            wd |= bit;
            if (!m.pBad[p]) { pos[p] = (int)s.localMPs.size(); s.localMPs.push_back(p); }
            else pos[p] = -1;
        }
    }
    s.locKFs = s.localKFs; s.locVersion = s.mapVersion; s.locWalkFrame = f.id;
    s.locListDev = dev;
}
static void local_points_from_list(Seq& s, int n) {   // s.localMPs[0 .. n) came from the table: marks and positions of the listed points (a bad point is in neither)
    Map& m = s.map;
    Frame& f = *s.cur;
    const size_t nw = (m.mps.size() + 63) / 64;
    if (s.mpMark.size() < nw) s.mpMark.resize(nw + nw / 2 + 4, 0ull);
    if (s.mpPos.size() < m.mps.size()) s.mpPos.resize(m.mps.size() + m.mps.size() / 2 + 64, -1);
    std::fill(s.mpMark.begin(), s.mpMark.end(), 0ull);
    s.localMPs.resize(n);
    uint64_t* mark = s.mpMark.data();
    int* pos = s.mpPos.data();
    for (int q = 0; q < n; q++) { const int p = s.localMPs[q]; mark[p >> 6] |= 1ull << (p & 63); pos[p] = q; }
    s.locKFs = s.localKFs; s.locVersion = s.mapVersion; s.locWalkFrame = f.id;
    s.locListPending = false;
}

// fills the PoseOptimization job arrays of the current frame
static void fill_pose_job(Ctx& c, Seq& s, int si, oslam_job_pose_t& j) {
    Frame& f = *s.cur;
    const int N = f.N;
    if (c.residentPts) {   // the table reads the map points' positions from its records and the keypoints from the frame it still holds: nothing to gather
        s.jOutlier.assign(N, 0);
        j.slot = si; j.N = N;
        memcpy(j.Tcw_in, f.pose.Tcw.m, 64);
        j.Xw = nullptr; j.obs = nullptr; j.invSigma2 = nullptr; j.has_mp = nullptr; j.mp_ids = f.mp.data();
        j.outlier = s.jOutlier.data(); j.n_inliers = 0;
        memcpy(j.Tcw_out, f.pose.Tcw.m, 64);
        return;
    }
    s.jXw.assign((size_t)N * 3, 0.f); s.jObs.resize((size_t)N * 3); s.jInv.resize(N); s.jHas.assign(N, 0); s.jOutlier.assign(N, 0);
    for (int i = 0; i < N; i++) {
        prefetch_ahead(s.map.mps, f.mp, i, N);
        s.jObs[(size_t)i * 3] = f.keysUn[i].x; s.jObs[(size_t)i * 3 + 1] = f.keysUn[i].y; s.jObs[(size_t)i * 3 + 2] = f.uRight[i];
        s.jInv[i] = c.invSigma2[f.keysUn[i].octave];
        const int p = f.mp[i];
        if (p >= 0) {
            s.jHas[i] = 1;
            const float* x = s.map.mps[p].pos;
            s.jXw[(size_t)i * 3] = x[0]; s.jXw[(size_t)i * 3 + 1] = x[1]; s.jXw[(size_t)i * 3 + 2] = x[2];
        }
    }
    j.slot = si; j.N = N;
    memcpy(j.Tcw_in, f.pose.Tcw.m, 64);
    j.Xw = s.jXw.data(); j.obs = s.jObs.data(); j.invSigma2 = s.jInv.data(); j.has_mp = s.jHas.data(); j.mp_ids = nullptr;
    j.outlier = s.jOutlier.data(); j.n_inliers = 0;
    memcpy(j.Tcw_out, f.pose.Tcw.m, 64);
}

// after PoseOptimization in TrackWithMotionModel / TrackReferenceKeyFrame: discard outliers (:858-879, :981-1008)
static bool finish_initial_pose(Seq& s, const oslam_job_pose_t& j) {
    Frame& f = *s.cur;
    M4 T; memcpy(T.m, j.Tcw_out, 64);
    f.pose.set_frame(T);
    int nmatchesMap = 0;
    for (int i = 0; i < f.N; i++) {
        const int p = f.mp[i];
        if (p < 0) continue;
        if (j.outlier[i]) {
            f.mp[i] = -1; f.outlier[i] = 0;
            s.map.pLastSeen[p] = f.id;
            s.seenList.push_back(p);
        } else {
            f.outlier[i] = 0;
            if (s.map.pNObs[p] > 0) nmatchesMap++;
        }
    }
    return nmatchesMap >= 10;
}

// ------------------------------------------------------------------------------------------------------------------
// Local mapping for the sequences that inserted a keyframe this step (LocalMapping::Run, src/LocalMapping.cc:48-113)
// ------------------------------------------------------------------------------------------------------------------
static void map_point_culling(Seq& s) {   // :171-206
    Map& m = s.map;
    const int cur = s.curKF;
    std::vector<int> keep;
    for (int p : s.recentAdded) {
        MapPt& mp = m.mps[p];
        if (m.pBad[p]) continue;
        if ((float)m.pFound[p] / m.pVisible[p] < 0.25f) { m.set_bad_point(p); s.st[12]++; continue; }
        if (cur - mp.firstKF >= 2 && m.pNObs[p] <= 3) { m.set_bad_point(p); s.st[12]++; continue; }
        if (cur - mp.firstKF >= 3) continue;
        keep.push_back(p);
    }
    s.recentAdded.swap(keep);
}

// 3x3 helpers for ComputeF12 (src/LocalMapping.cc:537-554); the cv::Mat chain is evaluated in double and rounded once per product
static void mat3_mul(const double* a, const double* b, double* r) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += a[i * 3 + k] * b[k * 3 + j]; r[i * 3 + j] = (double)(float)s; }
}
static void compute_F12(const Ctx& c, const KeyFrm& k1, const KeyFrm& k2, float F12[9]) {
    double R1[9], R2t[9], t1[3], t2[3];
    for (int r = 0; r < 3; r++) {
        for (int cc = 0; cc < 3; cc++) { R1[r * 3 + cc] = k1.pose.Tcw.m[r * 4 + cc]; R2t[r * 3 + cc] = k2.pose.Tcw.m[cc * 4 + r]; }
        t1[r] = k1.pose.Tcw.m[r * 4 + 3]; t2[r] = k2.pose.Tcw.m[r * 4 + 3];
    }
    double R12[9];
    mat3_mul(R1, R2t, R12);
    double t12[3];
    for (int r = 0; r < 3; r++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += -R12[r * 3 + k] * t2[k];
        t12[r] = (double)(float)((double)(float)s + t1[r]);
    }
    const double tx[9] = {0, -t12[2], t12[1], t12[2], 0, -t12[0], -t12[1], t12[0], 0};
    const double fx = c.cfg.fx, fy = c.cfg.fy, cx = c.cfg.cx, cy = c.cfg.cy;
    // K^-1 = [[1/fx, 0, -cx/fx], [0, 1/fy, -cy/fy], [0, 0, 1]]; K1.t().inv() = (K^-1)^T
    const double Ki[9] = {(double)(float)(1.0 / fx), 0, (double)(float)(-cx / fx), 0, (double)(float)(1.0 / fy), (double)(float)(-cy / fy), 0, 0, 1};
    const double KiT[9] = {Ki[0], 0, 0, 0, Ki[4], 0, Ki[2], Ki[5], 1};
    double a[9], b[9], f[9];
    mat3_mul(KiT, tx, a);
    mat3_mul(a, R12, b);
    mat3_mul(b, Ki, f);
    for (int i = 0; i < 9; i++) F12[i] = (float)f[i];
}

static void fill_tri_kf(const Ctx& c, const KeyFrm& k, oslam_tri_kf_t& t) {
    memcpy(t.Tcw, k.pose.Tcw.m, 64); memcpy(t.Twc, k.pose.Twc.m, 64);
    t.fx = c.cfg.fx; t.fy = c.cfg.fy; t.cx = c.cfg.cx; t.cy = c.cfg.cy; t.invfx = c.invfx; t.invfy = c.invfy; t.mbf = c.cfg.bf; t.mb = c.mb;
    t.keysUn = k.keysUn.data(); t.keys = k.keys.data(); t.uRight = k.uRight.data(); t.depth = k.depth.data(); t.n_kps = k.N;
}

// MapPoint::PredictScale(dist, KeyFrame*) (src/MapPoint.cc:488-503)
static int predict_scale(const Ctx& c, float maxD, float dist) {
    const float ratio = maxD / dist;
    int n = (int)std::ceil(std::log(ratio) / c.logScale);
    if (n < 0) n = 0;
    else if (n >= c.cfg.nLevels) n = c.cfg.nLevels - 1;
    return n;
}

// projection gates of ORBmatcher::Fuse (src/ORBmatcher.cc:840-890) for candidate points `pts` against keyframe k
static void fuse_queries(const Ctx& c, const Map& m, int k, const std::vector<int>& pts, float th, std::vector<oslam_proj_query_t>& q, std::vector<int>& qpt) {
    const KeyFrm& kf = m.kfs[k];
    const float* T = kf.pose.Tcw.m;
    q.clear(); qpt.clear();
    for (size_t pi = 0; pi < pts.size(); pi++) {
        prefetch_obs_ahead(m.mps, pts, pi, pts.size());
        const int p = pts[pi];
        if (p < 0) continue;
        const MapPt& mp = m.mps[p];
        if (m.pBad[p] || mp.obs_index(k) >= 0) continue;
        float pc[3];
        for (int r = 0; r < 3; r++) {
            float s = T[r * 4] * mp.pos[0];
            s += T[r * 4 + 1] * mp.pos[1];
            s += T[r * 4 + 2] * mp.pos[2];
            pc[r] = (float)((double)s + (double)T[r * 4 + 3]);
        }
        if (pc[2] < 0.0f) continue;
        const float invz = 1 / pc[2];
        const float x = pc[0] * invz, y = pc[1] * invz;
        const float u = c.cfg.fx * x + c.cfg.cx, v = c.cfg.fy * y + c.cfg.cy;
        if (!(u >= c.bounds[0] && u < c.bounds[2] && v >= c.bounds[1] && v < c.bounds[3])) continue;   // KeyFrame::IsInImage
        const float ur = u - c.cfg.bf * invz;
        const float maxDistance = 1.2f * mp.maxD, minDistance = 0.8f * mp.minD;
        const float PO[3] = {mp.pos[0] - kf.pose.Ow[0], mp.pos[1] - kf.pose.Ow[1], mp.pos[2] - kf.pose.Ow[2]};
        const float dist3D = norm3(PO);
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        const double dot = (double)PO[0] * mp.normal[0] + (double)PO[1] * mp.normal[1] + (double)PO[2] * mp.normal[2];
        if (dot < 0.5 * dist3D) continue;
        const int lvl = predict_scale(c, mp.maxD, dist3D);
        oslam_proj_query_t e;
        memset(&e, 0, sizeof(e));
        e.u = u; e.v = v; e.ur = ur; e.radius = th * c.scale[lvl]; e.minLevel = lvl - 1; e.maxLevel = lvl; e.flags = 1;
        memcpy(e.desc, mp.desc, 32);
        q.push_back(e); qpt.push_back(p);
    }
}

// surgery of ORBmatcher::Fuse (:950-970) in query order; points whose descriptor must be recomputed go to `upd`
// `touched` (optional): the points whose bad flag or observation list this call changed
static inline void fuse_apply_one(Seq& s, int k, int p, int best, std::vector<int>* touched) {
    Map& m = s.map;
    if (m.pBad[p]) return;
    const int inKF = m.kfs[k].mp[best];
    if (inKF >= 0) {
        if (!m.pBad[inKF]) {
            if (m.pNObs[inKF] > m.pNObs[p]) { if (m.replace_point(p, inKF)) s.updList.push_back(inKF); }
            else { if (m.replace_point(inKF, p)) s.updList.push_back(p); }
            if (touched) { touched->push_back(p); touched->push_back(inKF); }
        }
    } else {
        m.add_observation(p, k, best);
        m.set_kf_mp(k, best, p);
        if (touched) touched->push_back(p);
    }
    s.st[9]++;
}
static void fuse_apply(Seq& s, int k, const std::vector<int>& qpt, const int32_t* q_match, std::vector<int>* touched = nullptr) {
    for (size_t i = 0; i < qpt.size(); i++)
        if (q_match[i] >= 0) fuse_apply_one(s, k, qpt[i], q_match[i], touched);
}
// the same from the (point, keypoint) pairs of oslam_slam_ops_t::fuse_into_current (candidate order)
static void fuse_apply_pairs(Seq& s, int k, const int32_t* pairs, int n) {
    for (int q = 0; q < n; q++) fuse_apply_one(s, k, pairs[2 * q], pairs[2 * q + 1], nullptr);
}

// Second half of a local-mapping pass: Optimizer::LocalBundleAdjustment's write-back (src/Optimizer.cc:711-777) from the solved windows, the MapPoint updates of
// their points, KeyFrameCulling (src/LocalMapping.cc:633-697) for every sequence of the pass.  Synchronous schedule: called at the end of run_local_mapping;
// deferred schedule: by finish_local_mapping after the next step's tracking.
static int local_mapping_back_ok(Ctx& c, const std::vector<int>& who, const std::vector<Ctx::Win*>& wins);
// Per-sequence failure isolation (the reference: one System that loses track resets ITSELF, src/Tracking.cc:553-560).  A local-BA window the operator refused
// (Win::st[0] < 0: beyond its bounds, malformed) fails its own sequence only: that sequence leaves the pass here — no write-back, no culling —, is counted
// (oslam_slam_lba_window_stats [7]) and resets before its next frame like a system that lost track right after initialisation; the other sequences of the handle
// go on, with the results they would have had without it (tests/test_slam_driver_gpu.py).  Device / runtime errors (OSLAM_E_HIP) are not per-sequence and still
// fail the step.
static int local_mapping_back(Ctx& c, const std::vector<int>& who, const std::vector<Ctx::Win*>& wins) {
    bool any = false;
    for (const Ctx::Win* W : wins) any = any || W->st[0] < 0;
    if (!any) return local_mapping_back_ok(c, who, wins);
    std::vector<char> failed(c.S, 0);
    std::vector<Ctx::Win*> wins2;
    for (Ctx::Win* W : wins) {
        if (W->st[0] < 0) {
            Seq& s = *c.seq[W->si];
            failed[W->si] = 1;
            s.opFailures++;
            s.resetRequested = true;
            s.state = ST_LOST;
        } else wins2.push_back(W);
    }
    std::vector<int> who2;
    for (int si : who) if (!failed[si]) who2.push_back(si);
    return local_mapping_back_ok(c, who2, wins2);
}

static int local_mapping_back_ok(Ctx& c, const std::vector<int>& who, const std::vector<Ctx::Win*>& wins) {
    typedef Ctx::Win Win;
    Timer tm;
    int rc;
    if (!c.updMap) c.updMap.reset(new MpUpdate);
    MpUpdate& upd = *c.updMap;
    Pool& pool = *c.pool;
    const int flags = c.cfg.local_mapping;
    const int nW = (int)who.size();
    auto merge_upd = [&]() { upd.clear(); for (int si : who) { Seq& s = *c.seq[si]; for (int p : s.updList) upd.add(si, p); s.updList.clear(); } };
    const bool useWin = c.ops.mp_update_windows != nullptr;
    std::vector<std::vector<int32_t>> cullOut;   // [w][4 per candidate] device counts (empty: host path)
    std::vector<std::vector<int32_t>> cullIds;
    std::vector<oslam_job_cull_t> cullJobs;
    static const bool cull_dev = !getenv("OSLAM_SLAM_CULL_HOST");
    static const bool cull_check = getenv("OSLAM_SLAM_CULL_CHECK") != nullptr;   // debugging: every device verdict is compared with the host count
    struct CullStats { std::atomic<long long> dev{0}, amb{0}, after{0}; ~CullStats() { fprintf(stderr, "[cull stats] candidates decided from device counts %lld, ambiguous (host recount) %lld, after a cull of the pass (host) %lld\n", dev.load(), amb.load(), after.load()); } };
    static CullStats* cull_stats = getenv("OSLAM_SLAM_CULL_STATS") ? new CullStats : nullptr;
    static struct CullStatsAtExit { ~CullStatsAtExit() { delete cull_stats; } } cull_stats_at_exit;
    bool cull_requested = false, cull_collected = false;
    auto request_cull = [&]() -> int {
        cull_requested = true;
        if (!((flags & 16) && cull_dev && c.ops.map_journal && c.ops.kf_culling_counts)) { cull_collected = true; return OSLAM_OK; }
        cullOut.resize(nW); cullIds.resize(nW);
        std::vector<oslam_map_changes_t> chg(nW), chs;
        std::vector<uint8_t> has(nW, 0);
        std::vector<oslam_job_cull_t> jobs;
        if (c.jrScratch.size() < (size_t)c.S) c.jrScratch.resize(c.S);
        pool.parallel_for(nW, [&](int w) { Seq& s = *c.seq[who[w]]; if (s.map.jr_pending()) { s.map.journal_changes(who[w], c.thDepth, c.jrScratch[who[w]], chg[w]); has[w] = 1; } });
        for (int w = 0; w < nW; w++) {
            Seq& s = *c.seq[who[w]];
            Map& m = s.map;
            if (has[w]) chs.push_back(chg[w]);
            if (m.lvlOverflow) continue;
            for (int k : m.kfs[s.curKF].ordered) if (k != 0) cullIds[w].push_back(k);
            if (cullIds[w].empty()) continue;
            cullOut[w].assign(cullIds[w].size() * 4, 0);
            oslam_job_cull_t j; j.slot = who[w]; j.n = (int32_t)cullIds[w].size(); j.kf_ids = cullIds[w].data(); j.out = cullOut[w].data();
            jobs.push_back(j);
        }
        int rc2;
        if (!chs.empty() && (rc2 = c.ops.map_journal(c.ops.ctx, (int)chs.size(), chs.data()))) return rc2;
        cullJobs.swap(jobs);   // (the jobs name arrays of cullIds / cullOut: all stay alive until the collection)
        if (!cullJobs.empty() && (rc2 = c.ops.kf_culling_counts(c.ops.ctx, (int)cullJobs.size(), cullJobs.data(), c.thDepth))) return rc2;
        if (!c.ops.kf_culling_collect) cull_collected = true;
        { const double d_ = tm.lap(); c.sec[7] += d_; c.sec[13] += d_; c.cpu[7] += tm.cpu; c.cpu[13] += tm.cpu; }
        return OSLAM_OK;
    };
    if (flags & 8) {
        pool.parallel_for((int)wins.size(), [&](int wi) {
            Win& W = *wins[wi];
            Seq& s = *c.seq[W.si];
            Map& m = s.map;
            // erase list: mono edges first, then stereo edges (:711-757)
            for (int pass = 0; pass < 2; pass++)
                for (size_t e = 0; e < W.ekf.size(); e++) {
                    if (!W.erase[e]) continue;
                    const bool stereo = W.eobs[e * 3 + 2] >= 0;
                    if ((pass == 1) != stereo) continue;
                    const int k = W.eref[e].first, p = W.eref[e].second;
                    const int idx = m.mps[p].obs_index(k);
                    if (idx >= 0) m.set_kf_mp(k, idx, -1);
                    m.erase_observation(p, k);
                }
            for (int q = 0; q < W.nFree; q++) {   // (the local keyframes a degraded window held fixed keep their poses)
                M4 T; memcpy(T.m, &W.poses_out[(size_t)q * 16], 64);
                m.kfs[W.kfs[q]].pose.set_keyframe(T);
            }
            if (!useWin) {
                for (size_t j = 0; j < W.pts.size(); j++) {
                    if (j + kPF < W.pts.size()) __builtin_prefetch(&m.mps[W.pts[j + kPF]]);
                    MapPt& mp = m.mps[W.pts[j]];
                    for (int d = 0; d < 3; d++) mp.pos[d] = W.points_out[j * 3 + d];
                    s.updList.push_back(W.pts[j]);
                }
                return;
            }
            // UpdateNormalAndDepth from the window itself (oslam_job_mp_window_t): per point only the reference keyframe's window index and level scale factor
            // are looked up here; the positions are written together with the results below.  A point with an observation the window does not carry, or
            // whose reference keyframe's observation is not among its surviving edges, takes the general path (position now, update through mp_update).
            const size_t nP = W.pts.size(), nK = W.kfs.size();
            std::vector<int>& slot = s.counter;   // keyframe id -> window index + 1 (restored to 0 below)
            for (size_t q = 0; q < nK; q++) slot[W.kfs[q]] = (int)q + 1;
            W.uOw.resize(nK * 3);
            for (size_t q = 0; q < nK; q++) memcpy(&W.uOw[q * 3], m.kfs[W.kfs[q]].pose.Ow, 12);
            W.uskip.assign(nP, 0); W.uref.assign(nP, 0); W.ulsf.assign(nP, 1.f); W.uout5.resize(nP * 5 + 5);
            for (size_t j = 0; j < nP; j++) {
                const int p = W.pts[j];
                if (j + kPF < nP && !m.pBad[W.pts[j + kPF]]) __builtin_prefetch(&m.mps[W.pts[j + kPF]].refKF);
                if (m.pBad[p]) { W.uskip[j] = 1; continue; }
                int q = -1, eRef = -1;
                if (!W.pquirk[j]) {
                    const int rk = m.mps[p].refKF;
                    q = rk >= 0 ? slot[rk] - 1 : -1;
                    if (q >= 0)
                        for (int e = W.pstart[j]; e < W.pstart[j + 1]; e++)
                            if (W.ekf[e] == q && !W.erase[e]) { eRef = e; break; }
                }
                if (eRef < 0) {   // general path
                    W.uskip[j] = 2;
                    MapPt& mp = m.mps[p];
                    for (int d = 0; d < 3; d++) mp.pos[d] = W.points_out[j * 3 + d];
                    s.updList.push_back(p);
                    continue;
                }
                W.uref[j] = q; W.ulsf[j] = c.scale[W.eoct[eRef]];
            }
            for (size_t q = 0; q < nK; q++) slot[W.kfs[q]] = 0;
        });
        // the observation lists are final for this pass (the MapPoint updates below do not touch them): the culling counts are requested now and collected after them
        { const double d_ = tm.lap(); c.sec[7] += d_; c.sec[12] += d_; c.cpu[7] += tm.cpu; c.cpu[12] += tm.cpu; }
        if ((rc = request_cull())) return rc;
        if (useWin && !wins.empty()) {
            { const double d_ = tm.lap(); c.sec[7] += d_; c.sec[12] += d_; c.cpu[7] += tm.cpu; c.cpu[12] += tm.cpu; }
            std::vector<oslam_job_mp_window_t> wj(wins.size());
            for (size_t wi = 0; wi < wins.size(); wi++) {
                Win& W = *wins[wi];
                oslam_job_mp_window_t& j = wj[wi];
                j.slot = W.si; j.nP = (int32_t)W.pts.size(); j.nE = (int32_t)W.ekf.size(); j.nK = (int32_t)W.kfs.size();
                j.pt_ids = W.pts.data(); j.pt_start = W.pstart.data(); j.edge_kf = W.ekf.data(); j.erase = W.erase.data(); j.skip = W.uskip.data(); j.ref_kf = W.uref.data();
                j.lsf = W.ulsf.data(); j.Ow = W.uOw.data(); j.pos = W.points_out.data(); j.out5 = W.uout5.data();
            }
            if ((rc = c.ops.mp_update_windows(c.ops.ctx, (int)wj.size(), wj.data()))) return rc;
            { c.sec[5] += tm.lap(); c.cpu[5] += tm.cpu; }
            pool.parallel_for((int)wins.size(), [&](int wi) {
                Win& W = *wins[wi];
                Map& m = c.seq[W.si]->map;
                for (size_t j = 0; j < W.pts.size(); j++) {
                    if (j + kPF < W.pts.size() && W.uskip[j + kPF] != 2) __builtin_prefetch(&m.mps[W.pts[j + kPF]]);
                    if (W.uskip[j] == 2) continue;   // (took the general path: position already written)
                    MapPt& mp = m.mps[W.pts[j]];
                    for (int d = 0; d < 3; d++) mp.pos[d] = W.points_out[j * 3 + d];
                    if (W.uskip[j]) continue;
                    const float* o = &W.uout5[j * 5];
                    mp.normal[0] = o[0]; mp.normal[1] = o[1]; mp.normal[2] = o[2]; mp.maxD = o[3]; mp.minD = o[4];
                }
            });
        }
        merge_upd();
        { const double d_ = tm.lap(); c.sec[7] += d_; c.sec[12] += d_; c.cpu[7] += tm.cpu; c.cpu[12] += tm.cpu; }
        if ((rc = upd.run(c, false, true))) return rc;
        { c.sec[5] += tm.lap(); c.cpu[5] += tm.cpu; }
    }
    // --- KeyFrameCulling (:633-697) ---
    // Round 5: the counting loop of the candidates runs on the device from the table's mirror of the observation graph (oslam_slam_ops_t::map_journal /
    // kf_culling_counts); the host takes the verdicts in the reference's order.  SetBadFlag changes the counts of the candidates behind a culled keyframe, so from
    // the first cull of a pass on — and for a keyframe the mirror flags as ambiguous, and for a map whose octave histogram overflowed — the host counts itself.
    if (!cull_requested && (rc = request_cull())) return rc;   // (a pass without local BA)
    if (!cull_collected) { if ((rc = c.ops.kf_culling_collect(c.ops.ctx))) return rc; cull_collected = true; }
    if (flags & 16)
        pool.parallel_for(nW, [&](int w) {
            Seq& s = *c.seq[who[w]];
            Map& m = s.map;
            const std::vector<int> local = m.kfs[s.curKF].ordered;
            const bool useHist = !m.lvlOverflow;
            const int32_t* dev = (!cullOut.empty() && !cullOut[w].empty()) ? cullOut[w].data() : nullptr;
            bool culled_any = false;
            long long n_dev = 0, n_amb = 0, n_after = 0;
            int qi = 0;   // index of the candidate among the non-zero ids (cullIds order)
            for (int k : local) {
                if (k == 0) continue;
                const int q = qi++;
                if (dev) { if (culled_any) n_after++; else if (dev[4 * q + 3] != 0) n_amb++; else n_dev++; }
                if (dev && !culled_any && dev[4 * q + 3] == 0) {
                    const int ub = dev[4 * q], nMPs = dev[4 * q + 1], nRed = dev[4 * q + 2];
                    const int keepAt = ub / 10 + 2;
                    const bool cull = (nMPs - nRed < keepAt) && (nRed > 0.9 * nMPs);
                    if (cull_check) {
                        const KeyFrm& kfc = m.kfs[k];
                        int ub2 = 0, r2 = 0, n2 = 0;
                        for (int i = 0; i < kfc.N; i++) {
                            const int p = kfc.mp[i];
                            const bool good = !(kfc.depth[i] > c.thDepth || kfc.depth[i] < 0);
                            ub2 += p >= 0 && good;
                            if (p < 0 || m.pBad[p] || !good) continue;
                            n2++;
                            if (m.pNObs[p] > 3) {
                                int nn = 0;
                                const MapPt& mq = m.mps[p];
                                for (size_t oi = 0; oi < mq.obs.size(); oi++) if (mq.obs[oi].first != k && mq.okp[oi].octave <= kfc.oct[i] + 1) nn++;
                                r2 += nn >= 3;
                            }
                        }
                        if (ub2 != ub || n2 != nMPs || r2 != nRed) { fprintf(stderr, "OSLAM_SLAM_CULL_CHECK: sequence %d keyframe %d: device (%d, %d, %d), host (%d, %d, %d)\n", who[w], k, ub, nMPs, nRed, ub2, n2, r2); abort(); }
                    }
                    if (cull) { m.set_bad_keyframe(k); s.st[11]++; s.culledKFs.push_back(k); culled_any = true; }
                    continue;
                }
                const KeyFrm& kf = m.kfs[k];
                // The verdict is nRed > 0.9 * nMPs.  nMPs is at most the number of slots that hold a point at a usable depth (counted from the keyframe's own
                // arrays, no map access), so once the points found NOT redundant reach a tenth of that bound (+ 2: away from the rounding of 0.9 * nMPs) the keyframe
                // stays whatever the remaining slots hold, and the walk over their observation lists is skipped.  Most keyframes leave the loop this way.
                int ub = 0;
                for (int i = 0; i < kf.N; i++) ub += kf.mp[i] >= 0 && !(kf.depth[i] > c.thDepth || kf.depth[i] < 0);
                const int keepAt = ub / 10 + 2;
                int nRed = 0, nMPs = 0;
                for (int i = 0; i < kf.N && nMPs - nRed < keepAt; i++) {
                    if (!useHist) prefetch_okp_ahead(m.mps, kf.mp, i, kf.N);
                    const int p = kf.mp[i];
                    if (p < 0 || m.pBad[p]) continue;
                    if (kf.depth[i] > c.thDepth || kf.depth[i] < 0) continue;
                    nMPs++;
                    if (m.pNObs[p] > 3) {
                        const int lvl = kf.oct[i];
                        // "three OTHER observations at octave <= lvl + 1": the point's octave histogram counts ALL its observations there; with four or more the
                        // answer is yes and with two or fewer no, whether or not this keyframe's own observation is among them (Map::pLvl).  Exactly three: the lists.
                        const int all_le = useHist ? m.lvl_count_le(p, lvl + 1) : 3;
                        if (all_le != 3) { nRed += all_le >= 4; continue; }
                        int n = 0;
                        const MapPt& mq = m.mps[p];
                        for (size_t oi = 0; oi < mq.obs.size(); oi++) {
                            if (mq.obs[oi].first == k) continue;
                            if (mq.okp[oi].octave <= lvl + 1) { n++; if (n >= 3) break; }   // (the observing keypoint's octave, cached beside the observation: slam_map.h ObsKp)
                        }
                        if (n >= 3) nRed++;
                    }
                }
                if (nMPs - nRed < keepAt && nRed > 0.9 * nMPs) { m.set_bad_keyframe(k); s.st[11]++; s.culledKFs.push_back(k); culled_any = true; }
            }
            if (cull_stats) { cull_stats->dev += n_dev; cull_stats->amb += n_amb; cull_stats->after += n_after; }
        });
    if (c.ops.release_keyframes) {   // the table may recycle the resident records of the keyframes culled above
        std::vector<int32_t> rs, rk;
        for (int si : who) { Seq& s = *c.seq[si]; for (int k : s.culledKFs) { rs.push_back(si); rk.push_back(k); } s.culledKFs.clear(); }
        if (!rs.empty() && (rc = c.ops.release_keyframes(c.ops.ctx, (int)rs.size(), rs.data(), rk.data()))) return rc;
    } else
        for (int si : who) c.seq[si]->culledKFs.clear();
    { const double d_ = tm.lap(); c.sec[7] += d_; c.sec[13] += d_; c.cpu[7] += tm.cpu; c.cpu[13] += tm.cpu; }
    return OSLAM_OK;
}


// Deferred schedule: what is left of the previous step's local-mapping pass.  Waits for the local BA in flight (or runs it now when the table has no
// asynchronous form), then write-back, MapPoint updates and KeyFrameCulling.
static int finish_local_mapping(Ctx& c) {
    Ctx::PendingLM& pd = c.pend;
    if (!pd.active) return OSLAM_OK;
    pd.active = false;
    Timer tm;
    int rc = OSLAM_OK;
    if (!pd.probs.empty()) {
        if (pd.submitted) rc = c.ops.lba_wait(c.ops.ctx);
        else rc = c.ops.lba(c.ops.ctx, (int)pd.probs.size(), pd.probs.data());
        pd.submitted = false;
        if (rc) return rc;
    }
    { c.sec[6] += tm.lap(); c.cpu[6] += tm.cpu; }
    for (int si : pd.who) c.seq[si]->mapVersion++;   // poses, positions and the keyframe set of these sequences change: their cached local maps are stale
    return local_mapping_back(c, pd.who, pd.wins);
}

static int run_local_mapping(Ctx& c, const std::vector<int>& who) {
    if (who.empty()) return OSLAM_OK;
    Timer tm;
    int rc;
    if (!c.updMap) c.updMap.reset(new MpUpdate);
    MpUpdate& upd = *c.updMap;
    upd.clear();
    const int flags = c.cfg.local_mapping;
    // --- ProcessNewKeyFrame (:129-169) ---
    Pool& pool = *c.pool;
    const int nW = (int)who.size();
    for (int si : who) c.seq[si]->mapVersion++;   // the map of these sequences changes below: their cached local maps are stale
    c.mapStep++;
    auto merge_upd = [&]() { upd.clear(); for (int si : who) { Seq& s = *c.seq[si]; for (int p : s.updList) upd.add(si, p); s.updList.clear(); } };
    pool.parallel_for(nW, [&](int w) {
        Seq& s = *c.seq[who[w]];
        Map& m = s.map;
        s.updList.clear();
        s.curKF = s.newKFs.front();
        s.newKFs.clear();
        KeyFrm& kf = m.kfs[s.curKF];
        compute_bow(c, kf.N, kf.desc.data(), kf.bowNode);
        for (int i = 0; i < kf.N; i++) {
            prefetch_obs_ahead(m.mps, kf.mp, i, kf.N);
            const int p = kf.mp[i];
            if (p < 0 || m.pBad[p]) continue;
            if (m.mps[p].obs_index(s.curKF) < 0) { m.add_observation(p, s.curKF, i); s.updList.push_back(p); }
            else s.recentAdded.push_back(p);
        }
    });
    merge_upd();
    { const double d_ = tm.lap(); c.sec[7] += d_; c.sec[9] += d_; c.cpu[7] += tm.cpu; c.cpu[9] += tm.cpu; }
    if ((rc = upd.run(c, true, true))) return rc;
    { c.sec[5] += tm.lap(); c.cpu[5] += tm.cpu; }
    pool.parallel_for(nW, [&](int w) {
        Seq& s = *c.seq[who[w]];
        s.map.update_connections(s.curKF, s.counter);
        s.map.nKFsInMap++;
        if (flags & 1) map_point_culling(s);
    });
    { const double d_ = tm.lap(); c.sec[7] += d_; c.sec[9] += d_; c.cpu[7] += tm.cpu; c.cpu[9] += tm.cpu; }

    // --- CreateNewMapPoints (:208-453) ---
    // The reference handles the neighbours one after the other, and neighbour i sees the points created from neighbours < i in exactly one way: a keypoint of the
    // current keyframe that has received a point is skipped (`if(pMP1) continue`, src/ORBmatcher.cc:711-716).  Everything else a neighbour's search and
    // triangulation read is fixed during the pass: the poses, the neighbour's own map points (a neighbour occurs once, and new points only touch the current
    // keyframe and THEIR neighbour), and — SearchForTriangulation never sets vbMatched2 — every keypoint of the current keyframe picks its partner independently
    // of the others.  So ALL neighbours of all sequences are searched and triangulated in ONE batch each from the state before the pass, and the results are
    // applied in neighbour order with the skip test at application time: the same points in the same order as the reference's loop, with 3 operator calls instead
    // of 3 per neighbour round.  OSLAM_SLAM_CNMP_ROUNDS=1 keeps the lockstep rounds (A/B: bit-identical runs, tests/test_slam_driver_gpu.py).
    const bool cnmp_rounds = getenv("OSLAM_SLAM_CNMP_ROUNDS") != nullptr;
    if ((flags & 2) && !cnmp_rounds) {
        struct Pair { int w, ni, k2; };
        std::vector<std::vector<int>> neigh(who.size());
        std::vector<Pair> pairs;
        for (size_t w = 0; w < who.size(); w++) {
            Seq& s = *c.seq[who[w]];
            const Map::IntSpan bc = s.map.best_covisibles(s.curKF, 10);
            neigh[w].assign(bc.begin(), bc.end());
            for (size_t ni = 0; ni < neigh[w].size(); ni++) pairs.push_back({(int)w, (int)ni, neigh[w][ni]});
        }
        const int nPairs = (int)pairs.size();
        std::vector<std::vector<uint8_t>> flag1(who.size());
        pool.parallel_for(nW, [&](int w) {
            Seq& s = *c.seq[who[w]];
            const KeyFrm& k1 = s.map.kfs[s.curKF];
            flag1[w].resize(k1.N);
            for (int i = 0; i < k1.N; i++) flag1[w][i] = k1.mp[i] >= 0;
            (void)s.bow_views(c, s.curKF);   // (built once per keyframe: not from several pair jobs at a time)
            for (int k2 : neigh[w]) (void)s.bow_views(c, k2);
        });
        std::vector<std::vector<uint8_t>> has2(nPairs);
        std::vector<std::vector<int32_t>> match(nPairs);
        std::vector<oslam_job_bow_t> cand(nPairs);
        std::vector<uint8_t> have(nPairs, 0);
        pool.parallel_for(nPairs, [&](int q) {
            const Pair& pq = pairs[q];
            Seq& s = *c.seq[who[pq.w]];
            Map& m = s.map;
            const KeyFrm& k1 = m.kfs[s.curKF];
            KeyFrm& k2 = m.kfs[pq.k2];
            const float vb[3] = {k2.pose.Ow[0] - k1.pose.Ow[0], k2.pose.Ow[1] - k1.pose.Ow[1], k2.pose.Ow[2] - k1.pose.Ow[2]};
            if (norm3(vb) < c.mb) return;   // :251-254
            oslam_job_bow_t& j = cand[q];
            memset(&j, 0, sizeof(j));
            compute_F12(c, k1, k2, j.F12);
            // epipole of camera 1 in image 2 (src/ORBmatcher.cc:663-670)
            float C2[3];
            for (int r = 0; r < 3; r++) {
                float sacc = k2.pose.Tcw.m[r * 4] * k1.pose.Ow[0];
                sacc += k2.pose.Tcw.m[r * 4 + 1] * k1.pose.Ow[1];
                sacc += k2.pose.Tcw.m[r * 4 + 2] * k1.pose.Ow[2];
                C2[r] = (float)((double)sacc + (double)k2.pose.Tcw.m[r * 4 + 3]);
            }
            const float invz = 1.0f / C2[2];
            j.ex = c.cfg.fx * C2[0] * invz + c.cfg.cx; j.ey = c.cfg.fy * C2[1] * invz + c.cfg.cy;
            const BowViews& v1 = s.bow_views(c, s.curKF);
            const BowViews& v2 = s.bow_views(c, pq.k2);
            has2[q].resize(k2.N);
            for (int i = 0; i < k2.N; i++) has2[q][i] = k2.mp[i] >= 0;
            match[q].assign(k1.N, -1);
            j.s1.N = k1.N; j.s1.keys = k1.keysUn.data(); j.s1.desc = k1.desc.data(); j.s1.uRight = k1.uRight.data(); j.s1.flag = flag1[pq.w].data();
            j.s1.nq = k1.N; j.s1.q_idx = v1.q_idx.data(); j.s1.q_node = v1.q_node.data();
            j.s2.N = k2.N; j.s2.keys = k2.keysUn.data(); j.s2.desc = k2.desc.data(); j.s2.uRight = k2.uRight.data(); j.s2.has_mp = has2[q].data();
            j.s2.nNodes = (int)v2.nodes.size(); j.s2.nodes = v2.nodes.data(); j.s2.start = v2.start.data(); j.s2.items = v2.items.data();
            j.triangulation = 1; j.nnratio = 0.6f; j.checkOri = 0; j.match = match[q].data();
            have[q] = 1;
        });
        std::vector<oslam_job_bow_t> bj;
        std::vector<int> bjq;
        std::vector<oslam_kf_key_t> bkey;
        for (int q = 0; q < nPairs; q++)
            if (have[q]) { bj.push_back(cand[q]); bjq.push_back(q); bkey.push_back({who[pairs[q].w], c.seq[who[pairs[q].w]]->curKF, pairs[q].k2}); }
        { const double d_ = tm.lap(); c.sec[7] += d_; c.sec[10] += d_; c.cpu[7] += tm.cpu; c.cpu[10] += tm.cpu; }
        if (!bj.empty()) {
            if ((rc = c.ops.bow_keyed ? c.ops.bow_keyed(c.ops.ctx, (int)bj.size(), bj.data(), bkey.data()) : c.ops.bow(c.ops.ctx, (int)bj.size(), bj.data()))) return rc;
            const int nJ = (int)bj.size();
            std::vector<oslam_job_triangulate_t> tj(nJ);
            std::vector<std::vector<int32_t>> i1(nJ), i2(nJ);
            std::vector<std::vector<uint8_t>> okv(nJ);
            std::vector<std::vector<float>> x3(nJ);
            pool.parallel_for(nJ, [&](int jq) {
                const int q = bjq[jq];
                Seq& s = *c.seq[who[pairs[q].w]];
                const KeyFrm& k1 = s.map.kfs[s.curKF];
                // vMatchedIndices order (:815-820): ascending index of keyframe 1
                for (int i = 0; i < k1.N; i++) if (match[q][i] >= 0) { i1[jq].push_back(i); i2[jq].push_back(match[q][i]); }
                oslam_job_triangulate_t& t = tj[jq];
                fill_tri_kf(c, k1, t.kf1); fill_tri_kf(c, s.map.kfs[pairs[q].k2], t.kf2);
                t.M = (int)i1[jq].size(); t.idx1 = i1[jq].data(); t.idx2 = i2[jq].data();
                okv[jq].assign(t.M + 1, 0); x3[jq].assign((size_t)t.M * 3 + 3, 0.f);
                t.ok = okv[jq].data(); t.x3D = x3[jq].data();
            });
            if ((rc = c.ops.triangulate(c.ops.ctx, nJ, tj.data()))) return rc;
            { c.sec[8] += tm.lap(); c.cpu[8] += tm.cpu; }
            // application in neighbour order per sequence (the jobs of a sequence are consecutive and in neighbour order)
            std::vector<int> first(who.size() + 1, 0);
            for (int jq = 0; jq < nJ; jq++) first[pairs[bjq[jq]].w + 1]++;
            for (size_t w = 0; w < who.size(); w++) first[w + 1] += first[w];
            pool.parallel_for(nW, [&](int w) {
                Seq& s = *c.seq[who[w]];
                Map& m = s.map;
                for (int jq = first[w]; jq < first[w + 1]; jq++) {
                    const int k2 = pairs[bjq[jq]].k2;
                    for (int e = 0; e < tj[jq].M; e++) {
                        if (!okv[jq][e]) continue;
                        if (m.kfs[s.curKF].mp[i1[jq][e]] >= 0) continue;   // the keypoint received a point from an earlier neighbour: the reference's search skipped it
                        const int p = m.new_point(&x3[jq][(size_t)e * 3], s.curKF, m.kfs[s.curKF].frameId);   // :408-430
                        m.add_observation(p, s.curKF, i1[jq][e]);
                        m.add_observation(p, k2, i2[jq][e]);
                        m.set_kf_mp(s.curKF, i1[jq][e], p);
                        m.set_kf_mp(k2, i2[jq][e], p);
                        m.nMPsInMap++; s.st[3]++; s.st[10]++;
                        s.recentAdded.push_back(p);
                        s.updList.push_back(p);
                    }
                }
            });
            merge_upd();
            { const double d_ = tm.lap(); c.sec[7] += d_; c.sec[10] += d_; c.cpu[7] += tm.cpu; c.cpu[10] += tm.cpu; }
            if ((rc = upd.run(c, true, true))) return rc;
            { c.sec[5] += tm.lap(); c.cpu[5] += tm.cpu; }
        }
    }
    // the same in lockstep rounds (round i = neighbour i of every sequence; the matches of neighbour i see the points created from neighbour i-1): A/B form
    if ((flags & 2) && cnmp_rounds) {
        std::vector<std::vector<int>> neigh(who.size());
        size_t maxn = 0;
        for (size_t w = 0; w < who.size(); w++) {
            Seq& s = *c.seq[who[w]];
            const Map::IntSpan bc = s.map.best_covisibles(s.curKF, 10);
            neigh[w].assign(bc.begin(), bc.end());
            maxn = std::max(maxn, neigh[w].size());
        }
        std::vector<std::vector<uint8_t>> flag1(who.size()), has2(who.size());
        std::vector<std::vector<int32_t>> match(who.size());
        std::vector<oslam_job_bow_t> bj;
        std::vector<int> bjw;
        for (size_t ni = 0; ni < maxn; ni++) {
            bj.clear(); bjw.clear();
            std::vector<oslam_job_bow_t> cand(who.size());
            std::vector<uint8_t> have(who.size(), 0);
            pool.parallel_for(nW, [&](int w) {
                if (ni >= neigh[w].size()) return;
                Seq& s = *c.seq[who[w]];
                Map& m = s.map;
                const KeyFrm& k1 = m.kfs[s.curKF];
                KeyFrm& k2 = m.kfs[neigh[w][ni]];
                const float vb[3] = {k2.pose.Ow[0] - k1.pose.Ow[0], k2.pose.Ow[1] - k1.pose.Ow[1], k2.pose.Ow[2] - k1.pose.Ow[2]};
                if (norm3(vb) < c.mb) return;   // :251-254
                oslam_job_bow_t& j = cand[w];
                memset(&j, 0, sizeof(j));
                compute_F12(c, k1, k2, j.F12);
                // epipole of camera 1 in image 2 (src/ORBmatcher.cc:663-670)
                float C2[3];
                for (int r = 0; r < 3; r++) {
                    float sacc = k2.pose.Tcw.m[r * 4] * k1.pose.Ow[0];
                    sacc += k2.pose.Tcw.m[r * 4 + 1] * k1.pose.Ow[1];
                    sacc += k2.pose.Tcw.m[r * 4 + 2] * k1.pose.Ow[2];
                    C2[r] = (float)((double)sacc + (double)k2.pose.Tcw.m[r * 4 + 3]);
                }
                const float invz = 1.0f / C2[2];
                j.ex = c.cfg.fx * C2[0] * invz + c.cfg.cx; j.ey = c.cfg.fy * C2[1] * invz + c.cfg.cy;
                const BowViews& v1 = s.bow_views(c, s.curKF);
                const BowViews& v2 = s.bow_views(c, neigh[w][ni]);
                flag1[w].resize(k1.N); has2[w].resize(k2.N);
                for (int i = 0; i < k1.N; i++) flag1[w][i] = k1.mp[i] >= 0;
                for (int i = 0; i < k2.N; i++) has2[w][i] = k2.mp[i] >= 0;
                match[w].assign(k1.N, -1);
                j.s1.N = k1.N; j.s1.keys = k1.keysUn.data(); j.s1.desc = k1.desc.data(); j.s1.uRight = k1.uRight.data(); j.s1.flag = flag1[w].data();
                j.s1.nq = k1.N; j.s1.q_idx = v1.q_idx.data(); j.s1.q_node = v1.q_node.data();
                j.s2.N = k2.N; j.s2.keys = k2.keysUn.data(); j.s2.desc = k2.desc.data(); j.s2.uRight = k2.uRight.data(); j.s2.has_mp = has2[w].data();
                j.s2.nNodes = (int)v2.nodes.size(); j.s2.nodes = v2.nodes.data(); j.s2.start = v2.start.data(); j.s2.items = v2.items.data();
                j.triangulation = 1; j.nnratio = 0.6f; j.checkOri = 0; j.match = match[w].data();
                have[w] = 1;
            });
            std::vector<oslam_kf_key_t> bkey;
            for (size_t w = 0; w < who.size(); w++)
                if (have[w]) { bj.push_back(cand[w]); bjw.push_back((int)w); bkey.push_back({who[w], c.seq[who[w]]->curKF, neigh[w][ni]}); }
            if (bj.empty()) continue;
            { const double d_ = tm.lap(); c.sec[7] += d_; c.sec[10] += d_; c.cpu[7] += tm.cpu; c.cpu[10] += tm.cpu; }
            if ((rc = c.ops.bow_keyed ? c.ops.bow_keyed(c.ops.ctx, (int)bj.size(), bj.data(), bkey.data()) : c.ops.bow(c.ops.ctx, (int)bj.size(), bj.data()))) return rc;
            std::vector<oslam_job_triangulate_t> tj(bj.size());
            std::vector<std::vector<int32_t>> i1(bj.size()), i2(bj.size());
            std::vector<std::vector<uint8_t>> okv(bj.size());
            std::vector<std::vector<float>> x3(bj.size());
            pool.parallel_for((int)bj.size(), [&](int q) {
                const size_t w = bjw[q];
                Seq& s = *c.seq[who[w]];
                const KeyFrm& k1 = s.map.kfs[s.curKF];
                // vMatchedIndices order (:815-820): ascending index of keyframe 1
                for (int i = 0; i < k1.N; i++) if (match[w][i] >= 0) { i1[q].push_back(i); i2[q].push_back(match[w][i]); }
                oslam_job_triangulate_t& t = tj[q];
                fill_tri_kf(c, k1, t.kf1); fill_tri_kf(c, s.map.kfs[neigh[w][ni]], t.kf2);
                t.M = (int)i1[q].size(); t.idx1 = i1[q].data(); t.idx2 = i2[q].data();
                okv[q].assign(t.M + 1, 0); x3[q].assign((size_t)t.M * 3 + 3, 0.f);
                t.ok = okv[q].data(); t.x3D = x3[q].data();
            });
            if ((rc = c.ops.triangulate(c.ops.ctx, (int)tj.size(), tj.data()))) return rc;
            { c.sec[8] += tm.lap(); c.cpu[8] += tm.cpu; }
            pool.parallel_for((int)bj.size(), [&](int q) {   // one job per sequence: independent maps
                const size_t w = bjw[q];
                Seq& s = *c.seq[who[w]];
                Map& m = s.map;
                const int k2 = neigh[w][ni];
                for (int e = 0; e < tj[q].M; e++) {
                    if (!okv[q][e]) continue;
                    const int p = m.new_point(&x3[q][(size_t)e * 3], s.curKF, m.kfs[s.curKF].frameId);   // :408-430
                    m.add_observation(p, s.curKF, i1[q][e]);
                    m.add_observation(p, k2, i2[q][e]);
                    m.set_kf_mp(s.curKF, i1[q][e], p);
                    m.set_kf_mp(k2, i2[q][e], p);
                    m.nMPsInMap++; s.st[3]++; s.st[10]++;
                    s.recentAdded.push_back(p);
                    s.updList.push_back(p);
                }
            });
            merge_upd();
            { const double d_ = tm.lap(); c.sec[7] += d_; c.sec[10] += d_; c.cpu[7] += tm.cpu; c.cpu[10] += tm.cpu; }
            if ((rc = upd.run(c, true, true))) return rc;
            { c.sec[5] += tm.lap(); c.cpu[5] += tm.cpu; }
        }
    }

    // --- SearchInNeighbors (:455-535) ---
    // The target keyframes of one sequence are fused one after the other like the reference's loop (a fusion changes descriptors and
    // observations the next target sees); round t handles target t of every sequence in one batch.
    if (flags & 4) {
        // badf / mask / dup / slotOf / touched: what only the host knows of ORBmatcher::Fuse's gates (:849: the point is bad, or already observed in the target) for
        // the points of the current keyframe against ALL targets, computed once before the rounds (bit t of mask[i] = point i is observed in the keyframe with
        // slot t) and refreshed after every round for the points that round changed — instead of a pass over all ~1000 point records and their observation
        // lists per round and sequence.
        struct FuseSeq {
            std::vector<int> targets; std::vector<int> pts; std::vector<oslam_proj_query_t> q; std::vector<int> qpt; std::vector<int32_t> qm; std::vector<uint8_t> excl; int kf;
            std::vector<uint8_t> badf, slotOf; std::vector<uint64_t> mask; std::vector<int> dup, touched; bool cached = false;
        };
        static const bool excl_check = getenv("OSLAM_SLAM_FUSE_EXCL_CHECK") != nullptr;   // debugging: compare the cached flags with the full pass every round
        static const bool excl_cache = !getenv("OSLAM_SLAM_FUSE_EXCL_FULL");               // A/B knob: the full pass every round
        const bool fuse_by_id = c.residentPts && c.ops.fuse_points_keyed != nullptr;   // the table runs the projection gates itself from its map-point records
        // A round's descriptor updates (MapPoint::Replace -> ComputeDistinctiveDescriptors) are handed to the table WITHOUT a wait (mp_update_keyed_async): the table launches
        // them right in front of the NEXT round's search, so one wait covers both — one device round trip per round instead of two.  Same-box A/B, alternating,
        // identical results: 39.7 / 40.9 k against 38.6 / 40.6 k frames/s.  (The first form enqueued the update kernel at once and let it run beside the round's host
        // bookkeeping: 33.4 / 35.4 k against 36.1 / 37.8 k — slower; OSLAM_SLAM_MPU_SYNC=1 restores the wait per update.)
        static const bool mpu_async = getenv("OSLAM_SLAM_MPU_SYNC") == nullptr;
        std::vector<oslam_job_fuse_pts_t> pjobs;
        std::vector<FuseSeq> fs(who.size());
        std::vector<oslam_job_fuse_t> jobs;
        std::vector<int> jw;
        std::vector<oslam_kf_key_t> fkey;
        size_t maxt = 0;
        pool.parallel_for(nW, [&](int w) {
            Seq& s = *c.seq[who[w]];
            Map& m = s.map;
            const int cur = s.curKF;
            for (int k : m.best_covisibles(cur, 10)) {
                if (m.kfs[k].bad || m.kfs[k].fuseTargetForKF == cur) continue;
                fs[w].targets.push_back(k);
                m.kfs[k].fuseTargetForKF = cur;
                for (int k2 : m.best_covisibles(k, 5)) {
                    if (m.kfs[k2].bad || m.kfs[k2].fuseTargetForKF == cur || k2 == cur) continue;
                    fs[w].targets.push_back(k2);
                }
            }
            fs[w].pts.assign(m.kfs[cur].mp.begin(), m.kfs[cur].mp.end());   // vpMapPointMatches snapshot (:484)
            FuseSeq& f = fs[w];
            f.cached = false;
            if (!fuse_by_id || !excl_cache) return;
            f.slotOf.assign(m.kfs.size(), 255);
            int nslot = 0;
            for (int k : f.targets) if (f.slotOf[k] == 255) f.slotOf[k] = (uint8_t)nslot++;   // (the target list may name a keyframe twice, like the reference's)
            if (nslot > 64) return;   // (cannot happen with 10 + 10 x 5 targets; the full pass serves it)
            const size_t n = f.pts.size();
            f.badf.assign(n, 1); f.mask.assign(n, 0); f.dup.assign(n, -1); f.touched.clear();
            const int stamp = cur + 1;
            for (size_t pi = 0; pi < n; pi++) {
                prefetch_obs_ahead(m.mps, f.pts, pi, n);
                const int p = f.pts[pi];
                if (p < 0) continue;
                MapPt& mp = m.mps[p];
                if (mp.fuseListStamp == stamp) f.dup[pi] = mp.fuseListIdx;   // the same point at two keypoints of the keyframe: a chain through its positions
                mp.fuseListStamp = stamp; mp.fuseListIdx = (int)pi;
                f.badf[pi] = m.pBad[p] ? 1 : 0;
                uint64_t mk = 0;
                for (auto& e : mp.obs) { const uint8_t sl = f.slotOf[e.first]; if (sl != 255) mk |= 1ull << sl; }
                f.mask[pi] = mk;
            }
            f.cached = true;
        });
        auto fine = [&](int i) { const double d_ = tm.lap(); c.sec[7] += d_; c.sec[11] += d_; c.cpu[7] += tm.cpu; c.cpu[11] += tm.cpu; c.fine[i] += tm.cpu; };
        fine(0);
        auto refresh_touched = [&](Seq& s, FuseSeq& f) {   // after a round: the points it changed, if they are in the list
            const Map& m = s.map;
            const int stamp = s.curKF + 1;
            for (int x : f.touched) {
                const MapPt& mp = m.mps[x];
                if (mp.fuseListStamp != stamp) continue;
                uint64_t mk = 0;
                for (auto& e : mp.obs) { const uint8_t sl = (size_t)e.first < f.slotOf.size() ? f.slotOf[e.first] : 255; if (sl != 255) mk |= 1ull << sl; }
                for (int pi = mp.fuseListIdx; pi >= 0; pi = f.dup[pi]) { f.badf[pi] = m.pBad[x] ? 1 : 0; f.mask[pi] = mk; }
            }
            f.touched.clear();
        };
        for (size_t w = 0; w < who.size(); w++) maxt = std::max(maxt, fs[w].targets.size());
        auto fuse_round_by_id = [&](bool into_current, size_t t) -> int {
            jw.clear(); pjobs.clear();
            pool.parallel_for(nW, [&](int w) {
                Seq& s = *c.seq[who[w]];
                fs[w].kf = -1;
                if (!into_current && t >= fs[w].targets.size()) return;
                if (into_current && fs[w].targets.empty()) return;
                const int k = into_current ? s.curKF : fs[w].targets[t];
                fs[w].kf = k;
                const Map& m = s.map;
                const std::vector<int>& pts = fs[w].pts;
                fs[w].excl.assign(pts.size() + 1, 1);
                bool any = false;
                const bool cached = !into_current && fs[w].cached;
                if (cached) {
                    const FuseSeq& f = fs[w];
                    const int sl = f.slotOf[k];
                    for (size_t pi = 0; pi < pts.size(); pi++) {
                        const uint8_t ex = f.badf[pi] | (uint8_t)((f.mask[pi] >> sl) & 1);
                        fs[w].excl[pi] = ex; any = any || !ex;
                    }
                }
                if (!cached || excl_check)
                    for (size_t pi = 0; pi < pts.size(); pi++) {   // what only the host knows of ORBmatcher::Fuse's gates (:849): bad, or already in the keyframe
                        prefetch_obs_ahead(m.mps, pts, pi, pts.size());
                        const int p = pts[pi];
                        uint8_t ex = 1;
                        if (p >= 0) { const MapPt& mp = m.mps[p]; ex = (m.pBad[p] || mp.obs_index(k) >= 0) ? 1 : 0; }
                        if (cached) { if (ex != fs[w].excl[pi]) { fprintf(stderr, "[fuse excl check] mismatch: sequence %d keyframe %d point %d (list index %zu): cached %d, full %d\n", who[w], k, p, pi, fs[w].excl[pi], ex); abort(); } }
                        else { fs[w].excl[pi] = ex; any = any || !ex; }
                    }
                if (!any) fs[w].kf = -1;
                fs[w].qm.assign(pts.size() + 1, -1);
            });
            for (size_t w = 0; w < who.size(); w++) {
                if (fs[w].kf < 0) continue;
                const KeyFrm& kf = c.seq[who[w]]->map.kfs[fs[w].kf];
                oslam_job_fuse_pts_t j;
                j.slot = who[w]; j.kf = fs[w].kf; j.N = kf.N; j.M = (int)fs[w].pts.size(); j.ids = fs[w].pts.data(); j.excl = fs[w].excl.data();
                memcpy(j.Tcw, kf.pose.Tcw.m, 64); memcpy(j.Ow, kf.pose.Ow, 12); j.th = 3.0f; j.q_match = fs[w].qm.data();
                pjobs.push_back(j); jw.push_back((int)w);
            }
            fine(into_current ? 3 : 1);
            int rc2 = OSLAM_OK;
            if (pjobs.empty()) {
                if ((rc2 = upd.finish(c))) return rc2;
                { c.sec[5] += tm.lap(); c.cpu[5] += tm.cpu; }
                return OSLAM_OK;
            }
            // (the descriptor updates of the previous round may still be in flight: this search is queued behind them and reads their results from the resident
            // records; their host copies are scattered right after it)
            rc2 = c.ops.fuse_points_keyed(c.ops.ctx, (int)pjobs.size(), pjobs.data());
            if (rc2) return rc2;
            { c.sec[8] += tm.lap(); c.cpu[8] += tm.cpu; }
            if ((rc2 = upd.finish(c))) return rc2;
            { c.sec[5] += tm.lap(); c.cpu[5] += tm.cpu; }
            pool.parallel_for((int)jw.size(), [&](int q) {
                const int w = jw[q];
                const bool cached = !into_current && fs[w].cached;
                fuse_apply(*c.seq[who[w]], fs[w].kf, fs[w].pts, fs[w].qm.data(), cached ? &fs[w].touched : nullptr);
                if (cached) refresh_touched(*c.seq[who[w]], fs[w]);
            });
            merge_upd();
            fine(into_current ? 3 : 1);
            rc2 = upd.submit(c, true, false, mpu_async);   // Replace -> ComputeDistinctiveDescriptors (src/MapPoint.cc:314); collected behind the next round's search
            { c.sec[5] += tm.lap(); c.cpu[5] += tm.cpu; }
            return rc2;
        };
        auto fuse_round = [&](bool into_current, size_t t) -> int {
            if (fuse_by_id) return fuse_round_by_id(into_current, t);
            jobs.clear(); jw.clear(); fkey.clear();
            pool.parallel_for(nW, [&](int w) {
                Seq& s = *c.seq[who[w]];
                fs[w].q.clear(); fs[w].qpt.clear();
                if (!into_current && t >= fs[w].targets.size()) return;
                if (into_current && fs[w].targets.empty()) return;
                const int k = into_current ? s.curKF : fs[w].targets[t];
                fs[w].kf = k;
                fuse_queries(c, s.map, k, fs[w].pts, 3.0f, fs[w].q, fs[w].qpt);
                fs[w].qm.assign(fs[w].q.size() + 1, -1);
            });
            for (size_t w = 0; w < who.size(); w++) {
                Seq& s = *c.seq[who[w]];
                Map& m = s.map;
                if (fs[w].q.empty()) continue;
                const int k = fs[w].kf;
                const KeyFrm& kf = m.kfs[k];
                oslam_job_fuse_t j;
                j.N = kf.N; j.keysUn = kf.keysUn.data(); j.uRight = kf.uRight.data(); j.desc = kf.desc.data();
                j.M = (int)fs[w].q.size(); j.queries = fs[w].q.data(); j.q_match = fs[w].qm.data();
                jobs.push_back(j); jw.push_back((int)w); fkey.push_back({who[w], k, -1});
            }
            fine(into_current ? 3 : 1);
            if (jobs.empty()) return OSLAM_OK;
            int rc2 = c.ops.fuse_keyed ? c.ops.fuse_keyed(c.ops.ctx, (int)jobs.size(), jobs.data(), fkey.data()) : c.ops.fuse(c.ops.ctx, (int)jobs.size(), jobs.data());
            if (rc2) return rc2;
            { c.sec[8] += tm.lap(); c.cpu[8] += tm.cpu; }
            pool.parallel_for((int)jw.size(), [&](int q) { const int w = jw[q]; fuse_apply(*c.seq[who[w]], fs[w].kf, fs[w].qpt, fs[w].qm.data()); });
            merge_upd();
            fine(into_current ? 3 : 1);
            rc2 = upd.run(c, true, false);   // Replace -> ComputeDistinctiveDescriptors (src/MapPoint.cc:314)
            { c.sec[5] += tm.lap(); c.cpu[5] += tm.cpu; }
            return rc2;
        };
        for (size_t t = 0; t < maxt; t++)
            if ((rc = fuse_round(false, t))) return rc;
        if ((rc = upd.finish(c))) return rc;
        { c.sec[5] += tm.lap(); c.cpu[5] += tm.cpu; }
        // the targets' points into the current keyframe (:492-515)
        auto build_lists = [&]() {
            pool.parallel_for(nW, [&](int w) {
                Seq& s = *c.seq[who[w]];
                Map& m = s.map;
                const int cur = s.curKF;
                fs[w].pts.clear();
                // (mnFuseCandidateForKF as a dense per-sequence array, like baMark: up to 60 targets x 1000 slots per keyframe, most of them already marked)
                if (s.fuseMark.size() < m.mps.size()) s.fuseMark.resize(m.mps.size() + m.mps.size() / 2 + 64, 0);
                int* fm = s.fuseMark.data();
                for (int k : fs[w].targets) {
                    const int* kmp = m.kfs[k].mp.data();
                    for (size_t i = 0, n = m.kfs[k].mp.size(); i < n; i++) {
                        const int p = kmp[i];
                        if (p < 0 || m.pBad[p] || fm[p] == cur) continue;
                        fm[p] = cur;
                        fs[w].pts.push_back(p);
                    }
                }
            });
        };
        // With a table that mirrors the observation graph the candidate list is the TABLE's: it walks the targets' point lists where they are resident, runs Fuse's gates
        // and search and returns the matches (oslam_slam_ops_t::fuse_into_current) — no 60 x 1000-slot walk, no observation-list lookups, no candidate upload.  The change
        // sets of this pass so far (new keyframe, triangulated points, the first direction's fusions) go to the table first.  OSLAM_SLAM_FUSECUR_CHECK=1: the list and
        // its flags are also built here and compared entry by entry.
        static const bool fusecur_check = getenv("OSLAM_SLAM_FUSECUR_CHECK") != nullptr;
        bool cur_done = false;
        if (fuse_by_id && c.ops.fuse_into_current && c.ops.map_journal) {
            bool all_on = true;
            for (int w = 0; w < nW; w++) all_on = all_on && c.seq[who[w]]->map.jrOn;
            if (all_on) {
                std::vector<oslam_map_changes_t> chg(nW), chs;
                std::vector<uint8_t> has(nW, 0);
                if (c.jrScratch.size() < (size_t)c.S) c.jrScratch.resize(c.S);
                pool.parallel_for(nW, [&](int w) { Seq& s = *c.seq[who[w]]; if (s.map.jr_pending()) { s.map.journal_changes(who[w], c.thDepth, c.jrScratch[who[w]], chg[w]); has[w] = 1; } });
                for (int w = 0; w < nW; w++) if (has[w]) chs.push_back(chg[w]);
                if (!chs.empty() && (rc = c.ops.map_journal(c.ops.ctx, (int)chs.size(), chs.data()))) return rc;
                std::vector<oslam_job_fuse_cur_t> cj;
                std::vector<int> cw;
                std::vector<std::vector<int32_t>> cpairs(nW), dIds(nW);
                std::vector<std::vector<uint8_t>> dEx(nW);
                for (int w = 0; w < nW; w++) {
                    if (fs[w].targets.empty()) continue;
                    Seq& s = *c.seq[who[w]];
                    const KeyFrm& kf = s.map.kfs[s.curKF];
                    oslam_job_fuse_cur_t j;
                    memset(&j, 0, sizeof(j));
                    j.slot = who[w]; j.kf = s.curKF; j.n_targets = (int32_t)fs[w].targets.size(); j.targets = fs[w].targets.data();
                    memcpy(j.Tcw, kf.pose.Tcw.m, 64); memcpy(j.Ow, kf.pose.Ow, 12); j.th = 3.0f;
                    cpairs[w].resize(2 * 2048); j.max_pairs = 2048; j.pairs = cpairs[w].data();
                    if (fusecur_check) { dIds[w].resize(16384); dEx[w].resize(16384); j.dbg_cap = 16384; j.dbg_ids = dIds[w].data(); j.dbg_excl = dEx[w].data(); }
                    cj.push_back(j); cw.push_back(w);
                }
                fine(2);
                if (!cj.empty() && (rc = c.ops.fuse_into_current(c.ops.ctx, (int)cj.size(), cj.data()))) return rc;
                { c.sec[8] += tm.lap(); c.cpu[8] += tm.cpu; }
                bool overflow = false;
                for (auto& j : cj) overflow = overflow || j.overflow != 0;
                if (!overflow) {
                    if (fusecur_check) {
                        build_lists();
                        for (size_t q = 0; q < cj.size(); q++) {
                            const int w = cw[q];
                            const Seq& s = *c.seq[who[w]];
                            const std::vector<int>& pts = fs[w].pts;
                            bool same = (size_t)cj[q].n_candidates == pts.size();
                            for (size_t pi = 0; same && pi < pts.size(); pi++) {
                                const int pp = pts[pi];
                                const uint8_t ex = (s.map.pBad[pp] || s.map.mps[pp].obs_index(s.curKF) >= 0) ? 1 : 0;
                                same = dIds[w][pi] == pp && dEx[w][pi] == ex;
                                if (!same) fprintf(stderr, "OSLAM_SLAM_FUSECUR_CHECK: sequence %d keyframe %d candidate %zu: table (%d, %d), driver (%d, %d)\n", who[w], s.curKF, pi, dIds[w][pi], dEx[w][pi], pp, ex);
                            }
                            if (!same) { fprintf(stderr, "OSLAM_SLAM_FUSECUR_CHECK: the table's candidate list differs from the driver's (sequence %d keyframe %d: %d against %zu candidates)\n", who[w], s.curKF, cj[q].n_candidates, pts.size()); abort(); }
                        }
                    }
                    pool.parallel_for((int)cj.size(), [&](int q) { Seq& s = *c.seq[who[cw[q]]]; fuse_apply_pairs(s, s.curKF, cj[q].pairs, cj[q].n_pairs); });
                    merge_upd();
                    fine(3);
                    if ((rc = upd.run(c, true, false))) return rc;   // Replace -> ComputeDistinctiveDescriptors (src/MapPoint.cc:314)
                    { c.sec[5] += tm.lap(); c.cpu[5] += tm.cpu; }
                    cur_done = true;
                }
            }
        }
        if (!cur_done) {
            build_lists();
            fine(2);
            if ((rc = fuse_round(true, 0))) return rc;
            if ((rc = upd.finish(c))) return rc;
            { c.sec[5] += tm.lap(); c.cpu[5] += tm.cpu; }
        }
        // update points of the current keyframe (:517-531) and its connections
        pool.parallel_for(nW, [&](int w) {
            Seq& s = *c.seq[who[w]];
            // The reference recomputes descriptor and normal of every point of the keyframe here.  A point that went through a full update earlier in THIS
            // pass (ProcessNewKeyFrame, triangulation) and whose observation list has not changed since would get the same result: its position, the poses and
            // the bad flags of the observing keyframes only change later in the pass (local BA, culling).
            s.updList.clear();
            const std::vector<int>& kmp = s.map.kfs[s.curKF].mp;
            for (size_t i = 0; i < kmp.size(); i++) {
                prefetch_ahead(s.map.mps, kmp, i, kmp.size());
                const int p = kmp[i];
                if (p < 0) continue;
                const MapPt& mp = s.map.mps[p];
                if (s.map.pBad[p] || (mp.updStep == c.mapStep && mp.updVer == mp.obsVer)) continue;
                s.updList.push_back(p);
            }
        });
        merge_upd();   // (sequence order, then keypoint order: the order of the serial loop this replaces)
        fine(4);
        if ((rc = upd.run(c, true, true))) return rc;
        { c.sec[5] += tm.lap(); c.cpu[5] += tm.cpu; }
        pool.parallel_for(nW, [&](int w) { Seq& s = *c.seq[who[w]]; s.map.update_connections(s.curKF, s.counter); });
        fine(4);
    }

    // --- Optimizer::LocalBundleAdjustment (src/Optimizer.cc:453-778), all windows in one batch ---
    if (flags & 8) {
        typedef Ctx::Win Win;
        std::vector<Win>& pool_w = c.winPool;   // the windows' arrays keep their capacity from step to step
        if (pool_w.size() < who.size()) pool_w.resize(who.size());
        pool.parallel_for(nW, [&](int w) {
            const int si = who[w];
            Seq& s = *c.seq[si];
            Map& m = s.map;
            Win& W = pool_w[w];
            W.reset();
            W.si = -1;
            if (m.nKFsInMap <= 2) return;   // src/LocalMapping.cc:81
            const int cur = s.curKF;
            W.si = si;
            W.kfs.push_back(cur);
            for (int k : m.kfs[cur].ordered)
                if (!m.kfs[k].bad) W.kfs.push_back(k);
            W.nLocal = (int)W.kfs.size();
            // lLocalMapPoints (src/Optimizer.cc:470-487).  mnBALocalForKF of the points is a dense per-sequence array (as mpMark in update_local_map): the loop visits
            // ~1000 slots per local keyframe and most of them hold a point that is already in the list, so it should touch 4 bytes per slot, not a MapPt record.
            if (s.baMark.size() < m.mps.size()) s.baMark.resize(m.mps.size() + m.mps.size() / 2 + 64, 0);
            int* bam = s.baMark.data();
            size_t edgeCap = 0;
            for (int q = 0; q < W.nLocal; q++) {
                const std::vector<int>& kmp = m.kfs[W.kfs[q]].mp;
                const int* kp_ = kmp.data();
                for (size_t i = 0, n = kmp.size(); i < n; i++) {
                    if (i + kPF < n) { const int pq = kp_[i + kPF]; if (pq >= 0 && bam[pq] != cur) __builtin_prefetch(&m.mps[pq].obs); }
                    const int p = kp_[i];
                    if (p < 0 || bam[p] == cur) continue;
                    bam[p] = cur;

                    if (!m.pBad[p]) { W.pts.push_back(p); edgeCap += m.mps[p].obs.size(); }
                }
            }
            // Fixed keyframes (:489-504: every observer outside the local set, no bound) and the edges (:560-650) in ONE walk over the observation lists: a fixed
            // keyframe takes the next window index the first time an observation names it, which is the order the reference's lFixedCameras list is built in.
            std::vector<int>& slot = s.counter;   // keyframe id -> window index + 1 (restored to 0 below)
            for (int q = 0; q < W.nLocal; q++) slot[W.kfs[q]] = q + 1;
            W.points.resize(W.pts.size() * 3);
            W.ekf.resize(edgeCap); W.ept.resize(edgeCap); W.eobs.resize(edgeCap * 3); W.einv.resize(edgeCap); W.eref.resize(edgeCap); W.eoct.resize(edgeCap);
            W.pstart.resize(W.pts.size() + 1); W.pquirk.assign(W.pts.size(), 0);
            size_t ne = 0;
            for (size_t j = 0; j < W.pts.size(); j++) {
                prefetch_okp_ahead(m.mps, W.pts, j, W.pts.size());
                const int p = W.pts[j];
                const MapPt& mp = m.mps[p];
                W.pstart[j] = (int32_t)ne;
                for (int d = 0; d < 3; d++) W.points[j * 3 + d] = mp.pos[d];
                for (size_t oi = 0; oi < mp.obs.size(); oi++) {
                    const int kid = mp.obs[oi].first;
                    int q = slot[kid];
                    if (q == 0) {
                        if (m.kfs[kid].bad) { W.pquirk[j] = 1; continue; }   // (an observation the window does not carry: this point's update takes the general path)
                        W.kfs.push_back(kid);
                        slot[kid] = q = (int)W.kfs.size();
                    }
                    const ObsKp& kp = mp.okp[oi];   // mvKeysUn[idx].pt, mvuRight[idx], octave of the observing keypoint (cached beside the observation)
                    W.ekf[ne] = q - 1; W.ept[ne] = (int)j;
                    W.eobs[ne * 3] = kp.x; W.eobs[ne * 3 + 1] = kp.y; W.eobs[ne * 3 + 2] = kp.ur;
                    W.einv[ne] = c.invSigma2[kp.octave];
                    W.eoct[ne] = (uint8_t)kp.octave;
                    W.eref[ne] = std::make_pair(kid, p);
                    ne++;
                }
            }
            W.pstart[W.pts.size()] = (int32_t)ne;
            W.ekf.resize(ne); W.ept.resize(ne); W.eobs.resize(ne * 3); W.einv.resize(ne); W.eref.resize(ne); W.eoct.resize(ne);
            // A window with more than 128 FREE keyframes (768 unknowns) is beyond the local-BA operator (OSLAM_E_CAPACITY).  The reference has no such bound.  The
            // window is kept, DEGRADED: m.kfs[cur].ordered is weight-descending, so the current keyframe and its strongest covisible keyframes stay free up to the
            // bound and the remaining local keyframes enter as fixed cameras (fixed = 1): their points and edges are still in the window, points and the strongest
            // poses are still refined and outlier observations still erased.  Counted (oslam_slam_lba_window_stats [5]: degraded windows).
            int nFreeCap = W.nLocal;   // window indices >= nFreeCap are fixed cameras
            {
                int nFreeKF = 0;
                for (int q = 0; q < W.nLocal; q++) {
                    nFreeKF += W.kfs[q] != 0;
                    if (nFreeKF > kLbaMaxFreeKFs) { nFreeCap = q; break; }
                }
                if (nFreeCap < W.nLocal) s.lbaWindowsDegraded++;
            }
            W.nFree = nFreeCap;
            W.poses.resize(W.kfs.size() * 16); W.fixed.resize(W.kfs.size());
            for (size_t q = 0; q < W.kfs.size(); q++) {
                memcpy(&W.poses[q * 16], m.kfs[W.kfs[q]].pose.Tcw.m, 64);
                W.fixed[q] = (int)q >= nFreeCap ? 1 : (W.kfs[q] == 0 ? 2 : 0);
            }
            for (size_t q = 0; q < W.kfs.size(); q++) slot[W.kfs[q]] = 0;
            W.poses_out.resize(W.poses.size()); W.points_out.resize(W.points.size() + 3); W.erase.assign(W.ekf.size() + 1, 0);
            s.st[5]++; s.st[14] += (int64_t)W.ekf.size();
            s.lbaWin[0] += W.nLocal; s.lbaWin[1] += (int64_t)W.kfs.size() - W.nLocal; s.lbaWin[2] += (int64_t)W.pts.size(); s.lbaWin[3] += (int64_t)W.ekf.size();
        });
        std::vector<Win*> wins;
        for (int w = 0; w < nW; w++) if (pool_w[w].si >= 0) wins.push_back(&pool_w[w]);
        std::vector<oslam_lba_problem_t> probs(wins.size());
        for (size_t i = 0; i < wins.size(); i++) {
            Win& W = *wins[i];
            oslam_lba_problem_t& p = probs[i];
            p.nKF = (int)W.kfs.size(); p.poses = W.poses.data(); p.fixed = W.fixed.data(); p.nP = (int)W.pts.size(); p.points = W.points.data();
            p.nE = (int)W.ekf.size(); p.edge_kf = W.ekf.data(); p.edge_pt = W.ept.data(); p.edge_obs = W.eobs.data(); p.edge_invSigma2 = W.einv.data();
            p.poses_out = W.poses_out.data(); p.points_out = W.points_out.data(); p.erase = W.erase.data();
            W.st[0] = W.st[1] = W.st[2] = W.st[3] = 0;
            p.stats = W.st;
            if (c.injectLbaFailure >= 0 && W.si == c.injectLbaFailure && !W.ekf.empty()) { W.ekf[0] = (int32_t)W.kfs.size(); c.injectLbaFailure = -1; }   // test hook (oslam_slam_inject_failure): an edge that names a keyframe outside the window — the operator refuses the window (OSLAM_E_INVALID)
        }
        { const double d_ = tm.lap(); c.sec[7] += d_; c.sec[12] += d_; c.cpu[7] += tm.cpu; c.cpu[12] += tm.cpu; }
        if (flags & OSLAM_SLAM_LM_DEFERRED) {
            // deferred schedule: the solve runs while the next frame is tracked; write-back, MapPoint updates and KeyFrameCulling follow in finish_local_mapping
            Ctx::PendingLM& pd = c.pend;
            pd.active = true; pd.who = who; pd.wins = wins; pd.probs.swap(probs); pd.submitted = false;
            if (!pd.probs.empty() && c.ops.lba_submit) {
                if ((rc = c.ops.lba_submit(c.ops.ctx, (int)pd.probs.size(), pd.probs.data()))) return rc;
                pd.submitted = true;
            }
            { c.sec[6] += tm.lap(); c.cpu[6] += tm.cpu; }
            return OSLAM_OK;
        }
        if (!probs.empty() && (rc = c.ops.lba(c.ops.ctx, (int)probs.size(), probs.data()))) return rc;
        { c.sec[6] += tm.lap(); c.cpu[6] += tm.cpu; }
        return local_mapping_back(c, who, wins);
    }
    if (flags & OSLAM_SLAM_LM_DEFERRED) {   // (no local BA in this configuration: only the culling is deferred)
        c.pend.active = true; c.pend.who = who; c.pend.wins.clear(); c.pend.probs.clear(); c.pend.submitted = false;
        return OSLAM_OK;
    }
    return local_mapping_back(c, who, std::vector<Ctx::Win*>());
}

// ------------------------------------------------------------------------------------------------------------------
// Object layer (include/oslam_slam.h head comment)
// ------------------------------------------------------------------------------------------------------------------
// Frame::BuildObject2DsRGBD / BuildObject2DsStereo (src/Frame.cc:240-312, :314-386) from the keypoint test bits: detection i takes, in keypoint order, the
// keypoints still in the pool whose window lies in its mask and whose depth is in (0, mThDepth]; they leave the pool even if the detection then
// fails the "> 5 keypoints" test and builds no Object2D (the erase at :283 comes before the test at :291).
static void build_object2ds(const Ctx& c, Seq& s) {
    Frame& f = *s.cur;
    const oslam_slam_objects_t& d = *s.det;
    std::vector<uint8_t> taken(f.N, 0);
    for (int i = 0; i < d.n; i++) {
        Frame::Obj2D o;
        o.det = i; o.track = d.track_id ? d.track_id[i] : -1;
        for (int k = 0; k < f.N; k++) {
            if (taken[k] || !((s.jInMask[k] >> i) & 1)) continue;
            const float z = f.depth[k];
            if (!(z > 0 && z <= c.thDepth)) continue;
            taken[k] = 1;
            o.kps.push_back(k);
        }
        if ((int)o.kps.size() > 5) {
            const int idx = (int)f.objs.size();
            for (int k : o.kps) f.objOfKp[k] = idx;
            f.objs.push_back(std::move(o));
            s.sem[5]++;
        }
    }
}

// Tracking::TrackObject substitute: mvpObject3Ds[o] = the Object3D of the detection's track id, if one exists
static void track_objects(Seq& s) {
    for (auto& o : s.cur->objs) {
        o.obj3d = -1;
        if (o.track < 0) continue;
        auto it = s.objOfTrack.find(o.track);
        if (it != s.objOfTrack.end()) o.obj3d = it->second;
    }
}

// ObjectOptimizer::PoseOptimization2 inputs (src/ObjectOptimizer.cc:685-767, :978-990): masks and map points of the matched objects, M_joint candidates
static void fill_pose2_job(Ctx& c, Seq& s, int si) {
    Frame& f = *s.cur;
    s.hasPose2 = false;
    s.jMaskPtrs.clear(); s.jObjXw.clear(); s.jObjOf.clear(); s.jJointKp.clear(); s.jJointObj.clear(); s.jObjIds.clear();
    int m = 0;
    for (size_t io = 0; io < f.objs.size(); io++) {
        const Frame::Obj2D& o = f.objs[io];
        if (o.obj3d < 0) continue;
        const Seq::Obj3D& ob = s.obj3ds[o.obj3d];
        s.jMaskPtrs.push_back(s.det->masks[o.det]);
        const size_t n0 = s.jObjOf.size(), nob = ob.mps.size();
        s.jObjOf.resize(n0 + nob, m);
        if (c.residentPts) s.jObjIds.insert(s.jObjIds.end(), ob.mps.begin(), ob.mps.end());   // the table reads the positions from its records
        else s.jObjXw.resize((n0 + nob) * 3);
        for (size_t q = 0; q < nob && !c.residentPts; q++) {
            if (q + kPF < nob) __builtin_prefetch(&s.map.mps[ob.mps[q + kPF]]);
            const float* x = s.map.mps[ob.mps[q]].pos;
            float* d = &s.jObjXw[(n0 + q) * 3];
            d[0] = x[0]; d[1] = x[1]; d[2] = x[2];
        }
        for (int k = 0; k < f.N; k++) {
            const int p = f.mp[k];
            if (p < 0 || p >= (int)ob.member.size() || !ob.member[p]) continue;
            if (f.objOfKp[k] != (int)io) { s.jJointKp.push_back(k); s.jJointObj.push_back(m); }
        }
        m++;
    }
    if (m == 0) return;
    oslam_job_pose2_t& j = s.jPose2;
    j.base = s.jPose;
    j.nObj = m; j.masks = s.jMaskPtrs.data(); j.mask_stride = c.mask_stride; j.on_device = c.masks_on_device;
    j.nObjMp = (int)s.jObjOf.size(); j.objmp_Xw = c.residentPts ? nullptr : s.jObjXw.data(); j.objmp_obj = s.jObjOf.data();
    j.objmp_ids = c.residentPts ? s.jObjIds.data() : nullptr;
    j.nJoint = (int)s.jJointKp.size(); j.joint_kp = s.jJointKp.data(); j.joint_obj = s.jJointObj.data();
    j.n_semantic = 0;
    s.hasPose2 = true;
    (void)si;
}

// Tracking::UpdateCurrentObject (src/Tracking.cc:1079-1210) + Object3D::Update (src/ObjectTypes.cc:56-140), list logic only
static void update_current_objects(Seq& s) {
    Frame& f = *s.cur;
    Map& m = s.map;
    auto mark = [](Seq::Obj3D& ob, int p) { if ((int)ob.member.size() <= p) ob.member.resize(p + 1 + p / 2, 0); ob.member[p] = 1; };
    for (auto& o : f.objs) {
        if (o.obj3d >= 0) {
            Seq::Obj3D& ob = s.obj3ds[o.obj3d];
            ob.updateCnt++;
            std::vector<int> cand;
            for (int k : o.kps) {
                const int p = f.mp[k];
                if (p < 0 || m.pBad[p] || f.outlier[k]) continue;
                if (!(p < (int)ob.member.size() && ob.member[p])) cand.push_back(p);
            }
            for (int p : cand) { ob.mps.push_back(p); mark(ob, p); s.sem[4]++; }
        } else {
            std::vector<int> cand;
            for (int k : o.kps) if (f.mp[k] >= 0) cand.push_back(f.mp[k]);
            if ((int)cand.size() > 5) {   // MIN_OBJ3DMP_NUM (include/ObjectTypes.h:19)
                Seq::Obj3D ob;
                ob.track = o.track; ob.mps = cand;
                for (int p : cand) mark(ob, p);
                o.obj3d = (int)s.obj3ds.size();
                if (o.track >= 0 && !s.objOfTrack.count(o.track)) s.objOfTrack[o.track] = o.obj3d;
                s.obj3ds.push_back(std::move(ob));
                s.sem[3]++; s.sem[4] += (int64_t)cand.size();
            }
        }
    }
}

// Stage helpers of track_step (each runs for one sequence; the sequences of a batch are independent, so a stage is a parallel_for).
static void stage_motion_model_prepare(Ctx& c, int i) {
    Seq& s = *c.seq[i];
    Frame& f = *s.cur; Frame& l = *s.last;
    s.hasSL = false;
    // CheckReplacedInLastFrame (:820-835)
    for (int k = 0; k < l.N; k++) {
        const int p = l.mp[k];
        if (p >= 0 && s.map.pReplaced[p] >= 0) l.mp[k] = s.map.pReplaced[p];
    }
    if (!s.hasVelocity || f.id < s.lastRelocFrameId + 2) { s.path = 2; return; }
    s.path = 1;
    // UpdateLastFrame (:882-891)
    l.pose.set_frame(mul4(s.rel.back().Tcr, s.map.kfs[l.refKF].pose.Tcw));
    f.pose.set_frame(mul4(s.velocity, l.pose.Tcw));
    const int NL = l.N;
    const bool by_id = c.residentPts;   // the table takes positions / descriptors from its records and the last frame's keypoints from the previous step
    s.jHas.assign(NL, 0);
    if (!by_id) { s.jXw.assign((size_t)NL * 3, 0.f); s.jDesc.assign((size_t)NL * 32, 0); }
    for (int k = 0; k < NL; k++) {
        if (!by_id) prefetch_ahead(s.map.mps, l.mp, k, NL);
        const int p = l.mp[k];
        if (p < 0 || l.outlier[k]) continue;
        const MapPt& mp = s.map.mps[p];
        s.jHas[k] = 1 | (s.map.pNObs[p] > 0 ? 2 : 0);
        if (by_id) continue;
        for (int d = 0; d < 3; d++) s.jXw[(size_t)k * 3 + d] = mp.pos[d];
        memcpy(&s.jDesc[(size_t)k * 32], mp.desc, 32);
    }
    s.jMatch.assign(f.N + 1, -1);
    oslam_job_search_last_t& j = s.jSL;
    j.slot = i; j.cur = &f.view; j.Nlast = NL; j.has_mp = s.jHas.data();
    if (by_id) { j.Xw = nullptr; j.last_keysUn = nullptr; j.mp_desc = nullptr; j.mp_ids = l.mp.data(); }
    else { j.Xw = s.jXw.data(); j.last_keysUn = l.keysUn.data(); j.mp_desc = s.jDesc.data(); j.mp_ids = nullptr; }
    memcpy(j.Tcw, f.pose.Tcw.m, 64); memcpy(j.Tlw, l.pose.Tcw.m, 64);
    j.th = c.stereo ? 7.f : 15.f;   // :961-965
    j.kp_match = s.jMatch.data(); j.nmatches = 0;
    s.hasSL = true;
}

static void stage_local_map_prepare(Ctx& c, int i) {
    Seq& s = *c.seq[i];
    Frame& f = *s.cur;
    s.hasLoc = false;
    f.refKF = s.refKF;   // :446
    if (!s.ok) return;
    update_local_map(s);
}
// (second half: after the table has delivered the lists of the sequences whose local map changed — stage_local_map_lists)
static void stage_local_map_prepare_b(Ctx& c, int i) {
    Seq& s = *c.seq[i];
    Frame& f = *s.cur;
    if (!s.ok) return;
    // SearchLocalPoints (:1408-1458)
    Map& m = s.map;
    s.jBlocked.assign(f.N, 0);
    for (int k = 0; k < f.N; k++) {
        const int p = f.mp[k];
        if (p < 0) continue;
        if (m.pBad[p]) { f.mp[k] = -1; continue; }
        m.pVisible[p]++;
        m.pLastSeen[p] = f.id;
        s.jBlocked[k] = m.pNObs[p] > 0;
    }
    // The job lists ALL local points; those the reference leaves out of the projection (mnLastFrameSeen == this frame, :1413-1427: matched above or
    // discarded as outliers of the initial pose optimisation) are flagged in `skip`, so the packed arrays only depend on the local map and stay valid
    // (here and, by content id, in the operator table's resident copy) until the keyframe list or the map changes.
    const int M = (int)s.localMPs.size();
    if (!s.locReused) {
        s.locObs.resize(M + 1);
        if (c.residentPts) {   // the table gathers the points' arrays from its records: only Observations() > 0 is the driver's to tell
            for (int q = 0; q < M; q++) {
                s.locObs[q] = m.pNObs[s.localMPs[q]] > 0;
            }
        } else {
            s.locPw.resize((size_t)M * 3 + 3); s.locPn.resize((size_t)M * 3 + 3); s.locMax.resize(M + 1); s.locMin.resize(M + 1);
            s.locDesc.resize((size_t)M * 32 + 32);
        }
        for (int q = 0; q < M && !c.residentPts; q++) {
            prefetch_ahead(m.mps, s.localMPs, q, M);
            const MapPt& mp = m.mps[s.localMPs[q]];
            for (int d = 0; d < 3; d++) { s.locPw[(size_t)q * 3 + d] = mp.pos[d]; s.locPn[(size_t)q * 3 + d] = mp.normal[d]; }
            s.locMax[q] = mp.maxD; s.locMin[q] = mp.minD; s.locObs[q] = m.pNObs[s.localMPs[q]] > 0;
            memcpy(&s.locDesc[(size_t)q * 32], mp.desc, 32);
        }
        s.locContentId = c.next_content_id();
    } else c.locReuse.fetch_add(1, std::memory_order_relaxed);
    c.locFrames.fetch_add(1, std::memory_order_relaxed);
    s.jSkip.assign(M + 1, 0);
    {
        const uint64_t* mark = s.mpMark.data();
        const size_t markBits = s.mpMark.size() * 64;
        const int* pos = s.mpPos.data();
        auto flag = [&](int p) { if (s.locWalkFrame >= 0 && (size_t)p < markBits && ((mark[p >> 6] >> (p & 63)) & 1ull) && pos[p] >= 0) s.jSkip[pos[p]] = 1; };
        for (int k = 0; k < f.N; k++) if (f.mp[k] >= 0) flag(f.mp[k]);
        for (int p : s.seenList) flag(p);
    }
    s.jInView.assign(M + 1, 0); s.jMatch.assign(f.N + 1, -1);
    oslam_job_search_local_t& j = s.jLoc;
    j.slot = i; j.cur = &f.view; j.blocked = s.jBlocked.data(); j.M = M; j.obs_gt0 = s.locObs.data(); j.skip = s.jSkip.data(); j.content_id = s.locContentId;
    if (c.residentPts) { j.Pw = nullptr; j.Pn = nullptr; j.maxDist = nullptr; j.minDist = nullptr; j.mp_desc = nullptr; j.local_ids = s.localMPs.data(); }
    else { j.Pw = s.locPw.data(); j.Pn = s.locPn.data(); j.maxDist = s.locMax.data(); j.minDist = s.locMin.data(); j.mp_desc = s.locDesc.data(); j.local_ids = nullptr; }
    memcpy(j.Tcw, f.pose.Tcw.m, 64);
    j.th = f.id < s.lastRelocFrameId + 2 ? 5.f : (c.stereo ? 1.f : 3.f);   // th = 1, RGB-D 3, after a relocalisation 5 (:1450-1455)
    j.in_view = s.jInView.data(); j.kp_match = s.jMatch.data(); j.nmatches = 0;
    s.hasLoc = true;
}

// Tracking::UpdateLocalPoints (:1470-1493) for the sequences whose local keyframe list or map changed, by the table (oslam_slam_ops_t::local_points_list): their
// change sets go to the mirror first.  OSLAM_SLAM_LOCLIST_CHECK=1: every delivered list is compared with the driver's own walk.
static int stage_local_map_lists(Ctx& c, const std::vector<int>& tracking) {
    std::vector<int> need;
    for (int i : tracking) if (c.seq[i]->ok && c.seq[i]->locListPending) need.push_back(i);
    if (need.empty()) return OSLAM_OK;
    static const bool check = getenv("OSLAM_SLAM_LOCLIST_CHECK") != nullptr;
    Pool& pool = *c.pool;
    const int n = (int)need.size();
    std::vector<oslam_map_changes_t> chg(n), chs;
    std::vector<uint8_t> has(n, 0);
    if (c.jrScratch.size() < (size_t)c.S) c.jrScratch.resize(c.S);
    pool.parallel_for(n, [&](int q) { Seq& s = *c.seq[need[q]]; if (s.map.jr_pending()) { s.map.journal_changes(need[q], c.thDepth, c.jrScratch[need[q]], chg[q]); has[q] = 1; } });
    for (int q = 0; q < n; q++) if (has[q]) chs.push_back(chg[q]);
    int rc;
    if (!chs.empty() && (rc = c.ops.map_journal(c.ops.ctx, (int)chs.size(), chs.data()))) return rc;
    std::vector<oslam_job_local_list_t> jobs(n);
    for (int q = 0; q < n; q++) {
        Seq& s = *c.seq[need[q]];
        s.localMPs.resize(16384);
        oslam_job_local_list_t& j = jobs[q];
        j.slot = need[q]; j.n_kfs = (int32_t)s.localKFs.size(); j.kfs = s.localKFs.data(); j.cap = 16384; j.ids = s.localMPs.data(); j.n_ids = 0; j.overflow = 0;
    }
    if ((rc = c.ops.local_points_list(c.ops.ctx, n, jobs.data()))) return rc;
    pool.parallel_for(n, [&](int q) {
        Seq& s = *c.seq[need[q]];
        if (jobs[q].overflow) { local_points_walk_host(s); return; }
        if (check) {
            std::vector<int> got(s.localMPs.begin(), s.localMPs.begin() + jobs[q].n_ids);
            local_points_walk_host(s);
            if (got != s.localMPs) { fprintf(stderr, "OSLAM_SLAM_LOCLIST_CHECK: the table's local point list differs from the driver's walk (sequence %d frame %d: %zu against %zu points)\n", need[q], s.cur->id, got.size(), s.localMPs.size()); abort(); }
            return;
        }
        local_points_from_list(s, jobs[q].n_ids);
    });
    return OSLAM_OK;
}

static void stage_after_local_pose(Ctx& c, int i) {
    Seq& s = *c.seq[i];
    Frame& f = *s.cur;
    const oslam_job_pose_t& j = s.jPose;
    M4 T; memcpy(T.m, j.Tcw_out, 64);
    f.pose.set_frame(T);
    s.matchesInliers = 0;
    for (int k = 0; k < f.N; k++) {
        const int p = f.mp[k];
        if (p < 0) continue;
        f.outlier[k] = j.outlier[k];
        if (!f.outlier[k]) {
            s.map.pFound[p]++;
            if (s.map.pNObs[p] > 0) s.matchesInliers++;
        } else if (c.stereo) {
            f.mp[k] = -1;   // :1041-1042
        }
    }
    s.st[13] = s.matchesInliers;
    if (f.id < s.lastRelocFrameId + c.maxFrames && s.matchesInliers < 50) s.ok = false;
    else s.ok = s.matchesInliers >= 30;
}

// the part of Tracking::Track after TrackLocalMap (:470-566)
static void stage_after_tracking(Ctx& c, int i) {
    Seq& s = *c.seq[i];
    Frame& f = *s.cur; Frame& l = *s.last;
    Map& m = s.map;
    s.state = s.ok ? ST_OK : ST_LOST;
    if (s.ok) {
        if (l.pose.valid) { s.velocity = mul4(f.pose.Tcw, l.pose.Twc); s.hasVelocity = true; }
        else s.hasVelocity = false;
        for (int k = 0; k < f.N; k++) {   // clean VO matches
            const int p = f.mp[k];
            if (p >= 0 && m.pNObs[p] < 1) { f.outlier[k] = 0; f.mp[k] = -1; }
        }
        // NeedNewKeyFrame (:1242-1326) with an idle local mapper
        bool need = false;
        {
            const int nKFs = m.nKFsInMap;
            if (!(f.id < s.lastRelocFrameId + c.maxFrames && nKFs > c.maxFrames)) {
                const int nMinObs = nKFs <= 2 ? 2 : 3;
                // KeyFrame::TrackedMapPoints depends on the map alone: one count per (reference keyframe, nMinObs, map version)
                if (!(s.trkKF == s.refKF && s.trkMinObs == nMinObs && s.trkVersion == s.mapVersion)) {
                    s.trkCount = m.tracked_map_points(s.refKF, nMinObs);
                    s.trkKF = s.refKF; s.trkMinObs = nMinObs; s.trkVersion = s.mapVersion;
                }
                const int nRefMatches = s.trkCount;
                int nNonTrackedClose = 0, nTrackedClose = 0;
                for (int k = 0; k < f.N; k++)
                    if (f.depth[k] > 0 && f.depth[k] < c.thDepth) {
                        if (f.mp[k] >= 0 && !f.outlier[k]) nTrackedClose++;
                        else nNonTrackedClose++;
                    }
                const bool bNeedToInsertClose = (nTrackedClose < 100) && (nNonTrackedClose > 70);
                const float thRefRatio = nKFs < 2 ? 0.4f : 0.75f;
                const bool c1a = f.id >= s.lastKFFrameId + c.maxFrames;
                const bool c1b = f.id >= s.lastKFFrameId + c.minFrames;
                const bool c1c = s.matchesInliers < nRefMatches * 0.25 || bNeedToInsertClose;
                const bool c2 = (s.matchesInliers < nRefMatches * thRefRatio || bNeedToInsertClose) && s.matchesInliers > 15;
                need = (c1a || c1b || c1c) && c2;
            }
        }
        if (need) {   // CreateNewKeyFrame (:1328-1406)
            const int kf = new_keyframe(s, f, c.thDepth);
            s.refKF = kf; f.refKF = kf;
            create_stereo_points(c, s, f, kf, false);
            s.newKFs.push_back(kf);
            s.lastKFFrameId = f.id;
        }
        for (int k = 0; k < f.N; k++)
            if (f.mp[k] >= 0 && f.outlier[k]) f.mp[k] = -1;
        if (!f.objs.empty()) update_current_objects(s);   // UpdateCurrentObject(true) (:537)
    } else {
        s.st[8]++;
        if (m.nKFsInMap <= 5) s.resetRequested = true;   // "Track lost soon after initialisation, reseting..." (:553-561): Track() returns here
    }
    if (f.refKF < 0) f.refKF = s.refKF;
}

static int run_pose_jobs(Ctx& c, const std::vector<int>& who) {
    if (who.empty()) return OSLAM_OK;
    std::vector<oslam_job_pose_t> pj(who.size());
    for (size_t q = 0; q < who.size(); q++) pj[q] = c.seq[who[q]]->jPose;
    const int rc = c.ops.pose_opt(c.ops.ctx, (int)pj.size(), pj.data());
    for (size_t q = 0; q < who.size(); q++) c.seq[who[q]]->jPose = pj[q];
    return rc;
}

// ------------------------------------------------------------------------------------------------------------------
// One lockstep step of Tracking::Track for all sequences
// ------------------------------------------------------------------------------------------------------------------
static int track_step(Ctx& c, const uint8_t* const* gray, const uint8_t* const* right, int gray_stride, const float* const* depth, int depth_pitch,
                      int on_device, const double* stamps, const oslam_slam_objects_t* objs, int mask_stride, float* Tcw_out, int32_t* state_out,
                      const uint16_t* const* depth16 = nullptr, float depth_factor = 1.f) {
    const int S = c.S;
    oslam_drv::ShardScope shard_scope(c.shard);   // library code below the operator table (local-BA preparation, BoW views) uses this handle's worker set
    AccountScope acct_scope(&c.acct);   // the workers bill the tasks of this thread's batches (driver and operator table) to this handle; unbound again on every return
    c.acct.worker_ns.store(0, std::memory_order_relaxed);
    Timer tm;
    int rc;
    Pool& pool = *c.pool;
    // Frame::Frame for every sequence
    {
        std::vector<int32_t> slots(S);
        std::vector<oslam_slam_frame_t*> outs(S);
        for (int i = 0; i < S; i++) { slots[i] = i; outs[i] = &c.seq[i]->cur->view; }
        if (right) rc = c.ops.frames_stereo(c.ops.ctx, S, slots.data(), gray, right, gray_stride, on_device, outs.data());
        else if (depth16) rc = c.ops.frames_rgbd_raw16(c.ops.ctx, S, slots.data(), gray, gray_stride, depth16, depth_pitch, depth_factor, on_device, outs.data());
        else rc = c.ops.frames_rgbd(c.ops.ctx, S, slots.data(), gray, gray_stride, depth, depth_pitch, on_device, outs.data());
        if (rc) return rc;
    }
    { c.sec[0] += tm.lap(); c.cpu[0] += tm.cpu; }
    std::vector<int> tracking;   // sequences in the "system is initialised" branch
    pool.parallel_for(S, [&](int i) {
        Seq& s = *c.seq[i];
        if (s.resetRequested) s.reset();   // System::TrackRGBD: if(mbReset) mpTracker->Reset() before the frame is grabbed (src/System.cc:262-266)
        const int fid = s.nextFrameId++;
        s.cur->begin(fid, stamps ? stamps[i] : (double)fid);
        s.seenList.clear();
        s.st[0]++;
        s.path = 0; s.ok = false; s.hasSL = s.hasLoc = false;
        s.updList.clear();
        Frame& f = *s.cur;
        if (s.state == ST_NOT_INITIALIZED && f.N > 500) {
            // StereoInitialization (:590-642)
            f.pose.set_frame(eye4());
            const int kf = new_keyframe(s, f, c.thDepth);
            // mpMap->AddKeyFrame (:601): counted when ProcessNewKeyFrame inserts it (same std::set entry in the reference)
            create_stereo_points(c, s, f, kf, true);
            s.newKFs.push_back(kf);
            s.lastKFFrameId = f.id;
            s.localKFs.assign(1, kf);
            s.localMPs.clear();
            for (int p : s.map.kfs[kf].mp) if (p >= 0) s.localMPs.push_back(p);
            s.locVersion = -1;   // not a walk of update_local_map: nothing cached
            s.refKF = kf; f.refKF = kf;
            s.state = ST_OK;
            s.path = -1;   // initialised in this step: not tracked
        }
    });
    // Frame::BuildObject2DsRGBD / BuildObject2DsStereo (src/Frame.cc:179, :119): part of the Frame constructor, whatever the tracking state
    {
        std::vector<oslam_job_object_kps_t> oj;
        std::vector<int> ojw;
        c.mask_stride = mask_stride; c.masks_on_device = on_device;
        for (int i = 0; i < S; i++) {
            Seq& s = *c.seq[i];
            s.det = (objs && objs[i].n > 0) ? &objs[i] : nullptr;
            if (!s.det) continue;
            if (s.det->n > OSLAM_SLAM_MAX_OBJECTS || !s.det->masks) { oslam::set_error("track: sequence %d has %d detections (max %d) or no masks", i, s.det->n, OSLAM_SLAM_MAX_OBJECTS); return OSLAM_E_INVALID; }
            if (!c.ops.object_kps || !c.ops.pose_opt2) { oslam::set_error("track: this operator table has no semantic operators"); return OSLAM_E_INVALID; }
            s.jInMask.assign(s.cur->N + 1, 0);
            oslam_job_object_kps_t j;
            j.slot = i; j.cur = &s.cur->view; j.n_masks = s.det->n; j.masks = s.det->masks; j.mask_stride = mask_stride; j.on_device = on_device; j.in_mask = s.jInMask.data();
            oj.push_back(j); ojw.push_back(i);
        }
        if (!oj.empty()) {
            if ((rc = c.ops.object_kps(c.ops.ctx, (int)oj.size(), oj.data()))) return rc;
            pool.parallel_for((int)ojw.size(), [&](int q) { build_object2ds(c, *c.seq[ojw[q]]); });
        }
        { c.sec[0] += tm.lap(); c.cpu[0] += tm.cpu; }
    }
    for (int i = 0; i < S; i++) if (c.seq[i]->state == ST_OK && c.seq[i]->path != -1) tracking.push_back(i);
    const int nT = (int)tracking.size();
    // ---------------- initial pose: motion model or reference keyframe ----------------
    pool.parallel_for(nT, [&](int q) { stage_motion_model_prepare(c, tracking[q]); });
    std::vector<oslam_job_search_last_t> sl;
    std::vector<int> slw;
    for (int i : tracking) if (c.seq[i]->hasSL) { sl.push_back(c.seq[i]->jSL); slw.push_back(i); }
    { const double d_ = tm.lap(); c.sec[4] += d_; c.sec[14] += d_; c.cpu[4] += tm.cpu; c.cpu[14] += tm.cpu; }
    if (!sl.empty()) {
        if ((rc = c.ops.search_last(c.ops.ctx, (int)sl.size(), sl.data()))) return rc;
        std::vector<oslam_job_search_last_t> again;
        std::vector<size_t> againAt;
        for (size_t q = 0; q < sl.size(); q++)
            if (sl[q].nmatches < 20) { sl[q].th = c.stereo ? 14.f : 30.f; again.push_back(sl[q]); againAt.push_back(q); }   // :969-973 (2*th)
        if (!again.empty()) {
            if ((rc = c.ops.search_last(c.ops.ctx, (int)again.size(), again.data()))) return rc;
            for (size_t q = 0; q < again.size(); q++) sl[againAt[q]].nmatches = again[q].nmatches;
        }
    }
    { c.sec[1] += tm.lap(); c.cpu[1] += tm.cpu; }
    std::vector<int> pjw;
    for (size_t q = 0; q < sl.size(); q++) {
        Seq& s = *c.seq[slw[q]];
        if (sl[q].nmatches < 20) { s.path = 2; continue; }   // TrackWithMotionModel failed -> TrackReferenceKeyFrame (:365-366)
        pjw.push_back(slw[q]);
    }
    pool.parallel_for((int)pjw.size(), [&](int q) {
        Seq& s = *c.seq[pjw[q]];
        Frame& f = *s.cur; Frame& l = *s.last;
        for (int k = 0; k < f.N; k++) f.mp[k] = s.jMatch[k] >= 0 ? l.mp[s.jMatch[k]] : -1;
        fill_pose_job(c, s, pjw[q], s.jPose);
    });
    { const double d_ = tm.lap(); c.sec[4] += d_; c.sec[14] += d_; c.cpu[4] += tm.cpu; c.cpu[14] += tm.cpu; }
    if ((rc = run_pose_jobs(c, pjw))) return rc;
    { c.sec[2] += tm.lap(); c.cpu[2] += tm.cpu; }
    pool.parallel_for((int)pjw.size(), [&](int q) {
        Seq& s = *c.seq[pjw[q]];
        s.ok = finish_initial_pose(s, s.jPose);
        if (s.ok) s.st[6]++;
        else s.path = 2;
    });
    // TrackReferenceKeyFrame (:838-880) for the sequences without a motion model or whose motion-model tracking failed
    {
        std::vector<int> rk;
        for (int i : tracking) if (c.seq[i]->path == 2) rk.push_back(i);
        if (!rk.empty()) {
            std::vector<oslam_job_bow_t> bj(rk.size());
            std::vector<BowViews> bv(rk.size());
            std::vector<std::vector<uint8_t>> flag(rk.size());
            std::vector<std::vector<int32_t>> match(rk.size());
            if (c.seq[rk[0]]->lazyDesc) {   // Frame::ComputeBoW below runs on the host: these frames' descriptors are fetched now
                std::vector<int32_t> cnt(rk.size());
                std::vector<uint8_t*> outd(rk.size());
                for (size_t q = 0; q < rk.size(); q++) { Frame& f = *c.seq[rk[q]]->cur; cnt[q] = f.N; outd[q] = f.desc.data(); }
                if ((rc = c.ops.frame_descriptors(c.ops.ctx, (int)rk.size(), rk.data(), cnt.data(), outd.data()))) return rc;
            }
            pool.parallel_for((int)rk.size(), [&](int q) {
                Seq& s = *c.seq[rk[q]];
                Frame& f = *s.cur;
                KeyFrm& kf = s.map.kfs[s.refKF];
                compute_bow(c, f.N, f.desc.data(), f.bowNode);
                const BowViews& vk = s.bow_views(c, s.refKF);
                bv[q].side2(f.bowNode);
                flag[q].resize(kf.N);
                for (int k = 0; k < kf.N; k++) flag[q][k] = kf.mp[k] >= 0 && !s.map.pBad[kf.mp[k]];
                match[q].assign(f.N + 1, -1);
                oslam_job_bow_t& j = bj[q];
                memset(&j, 0, sizeof(j));
                j.s1.N = kf.N; j.s1.keys = kf.keysUn.data(); j.s1.desc = kf.desc.data(); j.s1.uRight = nullptr; j.s1.flag = flag[q].data();
                j.s1.nq = kf.N; j.s1.q_idx = vk.q_idx.data(); j.s1.q_node = vk.q_node.data();
                j.s2.N = f.N; j.s2.keys = f.keysUn.data(); j.s2.desc = f.desc.data(); j.s2.uRight = nullptr; j.s2.has_mp = nullptr;
                j.s2.nNodes = (int)bv[q].nodes.size(); j.s2.nodes = bv[q].nodes.data(); j.s2.start = bv[q].start.data(); j.s2.items = bv[q].items.data();
                j.triangulation = 0; j.nnratio = 0.7f; j.checkOri = 1; j.match = match[q].data();
            });
            { const double d_ = tm.lap(); c.sec[4] += d_; c.sec[14] += d_; c.cpu[4] += tm.cpu; c.cpu[14] += tm.cpu; }
            std::vector<oslam_kf_key_t> bkey(rk.size());
            for (size_t q = 0; q < rk.size(); q++) bkey[q] = {rk[q], c.seq[rk[q]]->refKF, -2};   // side 2 = the current frame, still on the device
            if ((rc = c.ops.bow_keyed ? c.ops.bow_keyed(c.ops.ctx, (int)bj.size(), bj.data(), bkey.data()) : c.ops.bow(c.ops.ctx, (int)bj.size(), bj.data()))) return rc;
            { c.sec[8] += tm.lap(); c.cpu[8] += tm.cpu; }
            pjw.clear();
            for (size_t q = 0; q < rk.size(); q++) {
                Seq& s = *c.seq[rk[q]];
                Frame& f = *s.cur;
                s.ok = false;
                if (bj[q].nmatches < 15) continue;
                const KeyFrm& kf = s.map.kfs[s.refKF];
                for (int k = 0; k < f.N; k++) f.mp[k] = match[q][k] >= 0 ? kf.mp[match[q][k]] : -1;
                f.pose.set_frame(s.last->pose.Tcw);
                fill_pose_job(c, s, rk[q], s.jPose);
                pjw.push_back(rk[q]);
            }
            if ((rc = run_pose_jobs(c, pjw))) return rc;
            { c.sec[2] += tm.lap(); c.cpu[2] += tm.cpu; }
            for (int i : pjw) {
                Seq& s = *c.seq[i];
                s.ok = finish_initial_pose(s, s.jPose);
                if (s.ok) s.st[7]++;
            }
        }
    }
    // ---------------- TrackLocalMap (:1011-1056) ----------------
    pool.parallel_for(nT, [&](int q) { stage_local_map_prepare(c, tracking[q]); });
    if ((rc = stage_local_map_lists(c, tracking))) return rc;
    pool.parallel_for(nT, [&](int q) { stage_local_map_prepare_b(c, tracking[q]); });
    std::vector<oslam_job_search_local_t> lj;
    std::vector<int> ljw;
    for (int i : tracking) if (c.seq[i]->hasLoc) { lj.push_back(c.seq[i]->jLoc); ljw.push_back(i); }
    { const double d_ = tm.lap(); c.sec[4] += d_; c.sec[15] += d_; c.cpu[4] += tm.cpu; c.cpu[15] += tm.cpu; }
    if (!lj.empty() && (rc = c.ops.search_local(c.ops.ctx, (int)lj.size(), lj.data()))) return rc;
    { c.sec[3] += tm.lap(); c.cpu[3] += tm.cpu; }
    pool.parallel_for((int)ljw.size(), [&](int q) {
        Seq& s = *c.seq[ljw[q]];
        Frame& f = *s.cur;
        for (int e = 0, Me = lj[q].M; e < Me; e++) {
            if (s.jInView[e]) s.map.pVisible[s.localMPs[e]]++;
        }
        for (int k = 0; k < f.N; k++) if (s.jMatch[k] >= 0) f.mp[k] = s.localMPs[s.jMatch[k]];
        fill_pose_job(c, s, ljw[q], s.jPose);
        s.hasPose2 = false;
        if (s.det && !f.objs.empty()) {   // TrackObject (:453) then ObjectOptimizer::PoseOptimization2 (:1022); without matched objects it is PoseOptimization
            track_objects(s);
            fill_pose2_job(c, s, ljw[q]);
        }
    });
    { const double d_ = tm.lap(); c.sec[4] += d_; c.sec[15] += d_; c.cpu[4] += tm.cpu; c.cpu[15] += tm.cpu; }
    {
        std::vector<int> plain, sem;
        for (int i : ljw) (c.seq[i]->hasPose2 ? sem : plain).push_back(i);
        if ((rc = run_pose_jobs(c, plain))) return rc;
        if (!sem.empty()) {
            std::vector<oslam_job_pose2_t> pj(sem.size());
            for (size_t q = 0; q < sem.size(); q++) pj[q] = c.seq[sem[q]]->jPose2;
            if ((rc = c.ops.pose_opt2(c.ops.ctx, (int)pj.size(), pj.data()))) return rc;
            for (size_t q = 0; q < sem.size(); q++) {
                Seq& s = *c.seq[sem[q]];
                s.jPose = pj[q].base;
                s.sem[0] += pj[q].n_semantic; s.sem[1]++; s.sem[2] += pj[q].n_semantic > 0;
            }
        }
    }
    { c.sec[2] += tm.lap(); c.cpu[2] += tm.cpu; }
    pool.parallel_for((int)ljw.size(), [&](int q) { stage_after_local_pose(c, ljw[q]); });
    // ---------------- after tracking (:470-566) ----------------
    pool.parallel_for(nT, [&](int q) { stage_after_tracking(c, tracking[q]); });
    std::vector<int> mapping;
    if (!c.updTrack) c.updTrack.reset(new MpUpdate);
    MpUpdate& upd = *c.updTrack;
    upd.clear();
    for (int i = 0; i < S; i++) {
        Seq& s = *c.seq[i];
        if (!s.newKFs.empty()) mapping.push_back(i);
        for (int p : s.updList) upd.add(i, p);
    }
    { const double d_ = tm.lap(); c.sec[4] += d_; c.sec[14] += d_; c.cpu[4] += tm.cpu; c.cpu[14] += tm.cpu; }
    if (c.ops.register_keyframes) {   // the frames that became keyframes are still on the device: the table keeps them (include/oslam_slam.h)
        std::vector<int32_t> rs, rk;
        for (int i = 0; i < S; i++) {
            for (int kf : c.seq[i]->pendingKF) { rs.push_back(i); rk.push_back(kf); }
            c.seq[i]->pendingKF.clear();
        }
        if (!rs.empty() && (rc = c.ops.register_keyframes(c.ops.ctx, (int)rs.size(), rs.data(), rk.data()))) return rc;
        if (!rs.empty() && c.ops.keyframe_raw_keys) {   // mvKeys of the new keyframes (KeyFrame::UnprojectStereo in CreateNewMapPoints reads them; no frame does)
            std::vector<int32_t> cnt(rs.size());
            std::vector<oslam_keypoint_t*> outp(rs.size());
            for (size_t q = 0; q < rs.size(); q++) {
                KeyFrm& k = c.seq[rs[q]]->map.kfs[rk[q]];
                k.keys.resize(k.N);
                cnt[q] = k.N; outp[q] = (oslam_keypoint_t*)k.keys.data();
            }
            if ((rc = c.ops.keyframe_raw_keys(c.ops.ctx, (int)rs.size(), rs.data(), cnt.data(), outp.data()))) return rc;
            if (c.seq[rs[0]]->lazyDesc) {   // KeyFrame::mDescriptors of the new keyframes (the frames' own descriptors stay on the device)
                std::vector<uint8_t*> outd(rs.size());
                for (size_t q = 0; q < rs.size(); q++) {
                    KeyFrm& k = c.seq[rs[q]]->map.kfs[rk[q]];
                    k.desc.resize((size_t)k.N * 32);
                    outd[q] = k.desc.data();
                }
                if ((rc = c.ops.keyframe_descriptors(c.ops.ctx, (int)rs.size(), rs.data(), cnt.data(), outd.data()))) return rc;
            }
        }
        if (!rs.empty() && c.ops.bow_nodes_keyed) {   // KeyFrame::ComputeBoW of the new keyframes from their resident descriptors (ProcessNewKeyFrame finds it done)
            std::vector<int32_t> cnt(rs.size());
            std::vector<uint32_t*> outp(rs.size());
            for (size_t q = 0; q < rs.size(); q++) {
                KeyFrm& k = c.seq[rs[q]]->map.kfs[rk[q]];
                k.bowNode.resize(k.N);
                cnt[q] = k.N; outp[q] = k.bowNode.data();
            }
            if ((rc = c.ops.bow_nodes_keyed(c.ops.ctx, (int)rs.size(), rs.data(), rk.data(), &c.voc.top[0][0], &c.voc.sub[0][0][0], cnt.data(), outp.data()))) return rc;
        }
    }
    if ((rc = upd.run(c, true, true))) return rc;   // descriptors / normals of the points created this step
    { c.sec[5] += tm.lap(); c.cpu[5] += tm.cpu; }
    // store relative poses (:569-585), swap frames
    for (int i = 0; i < S; i++) {
        Seq& s = *c.seq[i];
        Frame& f = *s.cur;
        if (s.resetRequested) {
            // Track() returned before the relative pose was stored and before mLastFrame was replaced
        } else if (f.pose.valid) {
            RelPose r;
            r.Tcr = mul4(f.pose.Tcw, s.map.kfs[f.refKF].pose.Twc);
            r.refKF = s.refKF; r.stamp = f.stamp; r.lost = s.state == ST_LOST;
            s.rel.push_back(r);
        } else if (!s.rel.empty()) {
            RelPose r = s.rel.back();
            r.lost = s.state == ST_LOST;
            s.rel.push_back(r);
        }
        if (Tcw_out) {
            if (f.pose.valid) memcpy(Tcw_out + (size_t)i * 16, f.pose.Tcw.m, 64);
            else memset(Tcw_out + (size_t)i * 16, 0, 64);
        }
        if (state_out) state_out[i] = s.state;
        if (s.state != ST_NOT_INITIALIZED) std::swap(s.cur, s.last);   // mLastFrame = Frame(mCurrentFrame)
    }
    { const double d_ = tm.lap(); c.sec[4] += d_; c.sec[14] += d_; c.cpu[4] += tm.cpu; c.cpu[14] += tm.cpu; }
    if ((rc = finish_local_mapping(c))) return rc;   // (deferred schedule: the second half of the previous step's pass, before this step's pass)
    return run_local_mapping(c, mapping);
}


// consistency of the observation graph (stats[15]): observation indices in range, Observations() = sum over the list, no observations on
// bad points, spanning-tree parents alive (covisibility lists may legitimately keep one-sided links to culled keyframes, src/KeyFrame.cc:367)
static int64_t map_violations(const Map& m) {
    int64_t bad = 0;
    for (size_t p = 0; p < m.mps.size(); p++) {
        const MapPt& mp = m.mps[p];
        if (m.pBad[p]) { bad += !mp.obs.empty(); continue; }
        int n = 0;
        for (auto& e : mp.obs) {
            const KeyFrm& k = m.kfs[e.first];
            // (k.mp[idx] may legitimately differ from p: SearchForTriangulation never sets vbMatched2, reference src/ORBmatcher.cc:722, so two
            // keypoints of the current keyframe can triangulate against the same neighbour keypoint and the second AddMapPoint wins)
            if (e.second < 0 || e.second >= k.N) { bad++; continue; }
            n += k.uRight[e.second] >= 0 ? 2 : 1;
        }
        if (!m.lvlOverflow) {   // the octave histogram KeyFrameCulling trusts equals the lists
            uint64_t h = 0;
            for (auto& o : mp.okp) h += 1ull << (8 * (o.octave & 7));
            if (h != m.pLvl[p]) bad++;
        }
        if (n != m.pNObs[p]) { bad++; if (getenv("OSLAM_SLAM_DEBUG")) fprintf(stderr, "violation: point %d nObs %d != %d\n", (int)p, m.pNObs[p], n); }
    }
    for (size_t k = 0; k < m.kfs.size(); k++) {
        const KeyFrm& kf = m.kfs[k];
        if (kf.bad) continue;
        if (kf.id != 0 && !kf.firstConnection && (kf.parent < 0 || m.kfs[kf.parent].bad)) bad++;
    }
    return bad;
}

}  // namespace oslam_drv

extern "C" {

int oslam_slam_create_with_ops(oslam_slam_t** out, const oslam_slam_config_t* cfg, const oslam_slam_ops_t* ops) {
    if (!out || !cfg || !ops || cfg->n_sequences <= 0 || cfg->nLevels <= 0 || cfg->nLevels > OSLAM_MAX_LEVELS || !(cfg->fx > 0) || !(cfg->fy > 0)) {
        oslam::set_error("oslam_slam_create: bad argument");
        return OSLAM_E_INVALID;
    }
    oslam_slam* h = new oslam_slam;
    Ctx& c = h->c;
    c.cfg = *cfg; c.ops = *ops; c.S = cfg->n_sequences;
    c.cap = c.ops.max_keypoints(c.ops.ctx);
    int rc = c.ops.scale_tables(c.ops.ctx, c.scale, c.invScale, c.sigma2, c.invSigma2);
    if (!rc) rc = c.ops.image_bounds(c.ops.ctx, c.bounds);
    if (rc) { if (c.ops.destroy) c.ops.destroy(c.ops.ctx); delete h; return rc; }
    c.invfx = 1.0f / cfg->fx; c.invfy = 1.0f / cfg->fy;
    c.mb = cfg->bf / cfg->fx;                       // src/Frame.cc: mb = mbf/fx
    c.thDepth = cfg->bf * cfg->thDepth / cfg->fx;   // src/Tracking.cc:159
    c.logScale = std::log(cfg->scaleFactor);
    c.maxFrames = (int)cfg->fps; c.minFrames = 0;
    c.stereo = cfg->sensor == 1;
    c.residentPts = c.ops.resident_points && c.ops.resident_points(c.ops.ctx) != 0;
    c.shard = oslam_drv::next_pool_shard();
    c.pool.reset(new Pool(cfg->host_threads > 1 ? cfg->host_threads : 1, true, c.shard));
    for (int i = 0; i < c.S; i++) {
        c.seq.emplace_back(new Seq);
        c.seq.back()->fa.alloc(c.cap);
        c.seq.back()->fb.alloc(c.cap);
        c.seq.back()->counter.assign(64, 0);
        // the table keeps a device mirror of the observation graph: the maps journal their changes (slam_map.h)
        c.seq.back()->lazyKeys = c.ops.keyframe_raw_keys != nullptr && c.ops.register_keyframes != nullptr;
        c.seq.back()->lazyDesc = c.seq.back()->lazyKeys && c.ops.keyframe_descriptors != nullptr && c.ops.frame_descriptors != nullptr && c.ops.bow_keyed != nullptr;
        // (opt-in, OSLAM_SLAM_LOCLIST_DEV=1: exact — OSLAM_SLAM_LOCLIST_CHECK — and measured slower: the list costs a device round trip in every step's tracking
        // stage, 38.5 / 39.0 k against 40.4 / 40.0 k frames/s for 0.7 of 10.7 core-seconds saved)
        c.seq.back()->locListDev = getenv("OSLAM_SLAM_LOCLIST_DEV") != nullptr && c.ops.local_points_list != nullptr && c.ops.map_journal != nullptr && c.ops.kf_culling_counts != nullptr &&
                                   (cfg->local_mapping & 16) && !getenv("OSLAM_SLAM_CULL_HOST");
        c.seq.back()->map.jrOn = c.ops.map_journal && c.ops.kf_culling_counts && (cfg->local_mapping & 16) && !getenv("OSLAM_SLAM_CULL_HOST");
    }
    *out = h;
    return OSLAM_OK;
}

int oslam_slam_create(oslam_slam_t** out, const oslam_slam_config_t* cfg) {
    if (!out || !cfg) { oslam::set_error("oslam_slam_create: bad argument"); return OSLAM_E_INVALID; }
    oslam_slam_ops_t ops;
    const int rc = oslam_slam_make_hip_ops(cfg, &ops);   // fails without a HIP device: there is no CPU fallback
    if (rc) return rc;
    return oslam_slam_create_with_ops(out, cfg, &ops);
}

void oslam_slam_destroy(oslam_slam_t* h) {
    if (h && getenv("OSLAM_SLAM_SN_STATS")) { const double* f = h->c.fine; fprintf(stderr, "[search-neighbors core-s] targets+masks %.3f rounds %.3f second-direction list %.3f its round %.3f updates+connections %.3f\n", f[0], f[1], f[2], f[3], f[4]); }
    if (!h) return;
    if (thread_account() == &h->c.acct) thread_account() = nullptr;   // never leave a dangling account bound to the destroying thread
    if (h->c.ops.destroy) h->c.ops.destroy(h->c.ops.ctx);
    delete h;
}

int oslam_slam_track_rgbd(oslam_slam_t* h, const uint8_t* const* gray, int gray_stride, const float* const* depth, int depth_pitch,
                          int on_device, const double* timestamps, float* Tcw_out, int32_t* state_out) {
    if (!h || !gray || !depth) { oslam::set_error("oslam_slam_track_rgbd: bad argument"); return OSLAM_E_INVALID; }
    if (h->c.stereo) { oslam::set_error("oslam_slam_track_rgbd on a STEREO handle"); return OSLAM_E_INVALID; }
    return track_step(h->c, gray, nullptr, gray_stride, depth, depth_pitch, on_device, timestamps, nullptr, 0, Tcw_out, state_out);
}

int oslam_slam_track_rgbd_objects(oslam_slam_t* h, const uint8_t* const* gray, int gray_stride, const float* const* depth, int depth_pitch, int on_device,
                                  const double* timestamps, const oslam_slam_objects_t* objs, int mask_stride, float* Tcw_out, int32_t* state_out) {
    if (!h || !gray || !depth) { oslam::set_error("oslam_slam_track_rgbd_objects: bad argument"); return OSLAM_E_INVALID; }
    if (h->c.stereo) { oslam::set_error("oslam_slam_track_rgbd_objects on a STEREO handle"); return OSLAM_E_INVALID; }
    if (objs && mask_stride < h->c.cfg.width) { oslam::set_error("oslam_slam_track_rgbd_objects: mask_stride < width"); return OSLAM_E_INVALID; }
    return track_step(h->c, gray, nullptr, gray_stride, depth, depth_pitch, on_device, timestamps, objs, mask_stride, Tcw_out, state_out);
}

int oslam_slam_track_rgbd_raw16(oslam_slam_t* h, const uint8_t* const* gray, int gray_stride, const uint16_t* const* depth16, int depth_pitch, float depth_factor,
                                int on_device, const double* timestamps, const oslam_slam_objects_t* objs, int mask_stride, float* Tcw_out, int32_t* state_out) {
    if (!h || !gray || !depth16) { oslam::set_error("oslam_slam_track_rgbd_raw16: bad argument"); return OSLAM_E_INVALID; }
    if (h->c.stereo) { oslam::set_error("oslam_slam_track_rgbd_raw16 on a STEREO handle"); return OSLAM_E_INVALID; }
    if (!h->c.ops.frames_rgbd_raw16) { oslam::set_error("oslam_slam_track_rgbd_raw16: the operator table has no frames_rgbd_raw16"); return OSLAM_E_INVALID; }
    if (objs && mask_stride != 0 && mask_stride < h->c.cfg.width) { oslam::set_error("oslam_slam_track_rgbd_raw16: mask_stride must be 0 (bitmaps) or >= width"); return OSLAM_E_INVALID; }
    if (objs && mask_stride == 0 && !on_device) { oslam::set_error("oslam_slam_track_rgbd_raw16: one-bit masks must be device-accessible (on_device != 0)"); return OSLAM_E_INVALID; }
    return track_step(h->c, gray, nullptr, gray_stride, nullptr, depth_pitch, on_device, timestamps, objs, mask_stride, Tcw_out, state_out, depth16, depth_factor);
}

int oslam_slam_finish(oslam_slam_t* h) {
    if (!h) { oslam::set_error("oslam_slam_finish: NULL handle"); return OSLAM_E_INVALID; }
    AccountScope acct_scope(&h->c.acct);
    return finish_local_mapping(h->c);
}

int oslam_slam_inject_failure(oslam_slam_t* h, int seq) {
    if (!h || seq < 0 || seq >= h->c.S) { oslam::set_error("oslam_slam_inject_failure: bad argument"); return OSLAM_E_INVALID; }
    h->c.injectLbaFailure = seq;
    return OSLAM_OK;
}

int oslam_slam_lba_window_stats(oslam_slam_t* h, int seq, int64_t out[8]) {
    if (!h || seq < 0 || seq >= h->c.S || !out) { oslam::set_error("oslam_slam_lba_window_stats: bad argument"); return OSLAM_E_INVALID; }
    const auto& s = *h->c.seq[seq];
    memset(out, 0, 8 * sizeof(int64_t));
    out[0] = s.st[5]; out[1] = s.lbaWin[0]; out[2] = s.lbaWin[1]; out[3] = s.lbaWin[2]; out[4] = s.lbaWin[3]; out[5] = s.lbaWindowsDegraded; out[7] = s.opFailures;
    return OSLAM_OK;
}

int oslam_slam_object_stats(oslam_slam_t* h, int seq, int64_t out[8]) {
    if (!h || seq < 0 || seq >= h->c.S || !out) { oslam::set_error("oslam_slam_object_stats: bad argument"); return OSLAM_E_INVALID; }
    memcpy(out, h->c.seq[seq]->sem, sizeof(h->c.seq[seq]->sem));
    out[6] = h->c.seq[seq]->lbaWindowsDegraded;
    return OSLAM_OK;
}

int oslam_slam_track_stereo(oslam_slam_t* h, const uint8_t* const* left, const uint8_t* const* right, int gray_stride, int on_device,
                            const double* timestamps, float* Tcw_out, int32_t* state_out) {
    if (!h || !left || !right) { oslam::set_error("oslam_slam_track_stereo: bad argument"); return OSLAM_E_INVALID; }
    if (!h->c.stereo || !h->c.ops.frames_stereo) { oslam::set_error("oslam_slam_track_stereo needs cfg.sensor = 1 (STEREO) and a stereo-capable operator table"); return OSLAM_E_INVALID; }
    return track_step(h->c, left, right, gray_stride, nullptr, 0, on_device, timestamps, nullptr, 0, Tcw_out, state_out);
}

int oslam_slam_track_stereo_objects(oslam_slam_t* h, const uint8_t* const* left, const uint8_t* const* right, int gray_stride, int on_device,
                                    const double* timestamps, const oslam_slam_objects_t* objs, int mask_stride, float* Tcw_out, int32_t* state_out) {
    if (!h || !left || !right) { oslam::set_error("oslam_slam_track_stereo_objects: bad argument"); return OSLAM_E_INVALID; }
    if (!h->c.stereo || !h->c.ops.frames_stereo) { oslam::set_error("oslam_slam_track_stereo_objects needs cfg.sensor = 1 (STEREO) and a stereo-capable operator table"); return OSLAM_E_INVALID; }
    if (objs && mask_stride < h->c.cfg.width) { oslam::set_error("oslam_slam_track_stereo_objects: mask_stride < width"); return OSLAM_E_INVALID; }
    return track_step(h->c, left, right, gray_stride, nullptr, 0, on_device, timestamps, objs, mask_stride, Tcw_out, state_out);
}

static void twc_rows(const M4& Tcw, float* o) {   // Rwc = Rcw.t(), twc = -Rwc*tcw (src/System.cc:423-424)
    for (int r = 0; r < 3; r++) {
        for (int cc = 0; cc < 3; cc++) o[r * 4 + cc] = Tcw.m[cc * 4 + r];
        float s = Tcw.m[r] * Tcw.m[3];
        s += Tcw.m[4 + r] * Tcw.m[7];
        s += Tcw.m[8 + r] * Tcw.m[11];
        o[r * 4 + 3] = (float)((double)s * -1.0);
    }
}

int oslam_slam_trajectory(oslam_slam_t* h, int seq, int cap, double* stamps, float* Twc, int32_t* n_out) {
    if (!h || seq < 0 || seq >= h->c.S || !n_out) { oslam::set_error("oslam_slam_trajectory: bad argument"); return OSLAM_E_INVALID; }
    { const int rc_ = oslam_slam_finish(h); if (rc_) return rc_; }   // System::Shutdown waits for the local mapper before the trajectory is saved
    const Seq& s = *h->c.seq[seq];
    const Map& m = s.map;
    int n = 0;
    if (m.kfs.empty()) { *n_out = 0; return OSLAM_OK; }
    const M4 Two = m.kfs[0].pose.Twc;   // first keyframe by id (:389-393)
    for (const RelPose& r : s.rel) {
        if (r.lost) continue;
        int k = r.refKF;
        M4 Trw = eye4();
        while (m.kfs[k].bad) { Trw = mul4(Trw, m.kfs[k].Tcp); k = m.kfs[k].parent; }
        Trw = mul4(mul4(Trw, m.kfs[k].pose.Tcw), Two);
        const M4 Tcw = mul4(r.Tcr, Trw);
        if (n < cap) {
            if (stamps) stamps[n] = r.stamp;
            if (Twc) twc_rows(Tcw, Twc + (size_t)n * 12);
        }
        n++;
    }
    *n_out = n;
    if (n > cap) { oslam::set_error("trajectory: cap %d < %d", cap, n); return OSLAM_E_CAPACITY; }
    return OSLAM_OK;
}

int oslam_slam_keyframe_trajectory(oslam_slam_t* h, int seq, int cap, double* stamps, float* Twc, int32_t* n_out) {
    if (!h || seq < 0 || seq >= h->c.S || !n_out) { oslam::set_error("oslam_slam_keyframe_trajectory: bad argument"); return OSLAM_E_INVALID; }
    { const int rc_ = oslam_slam_finish(h); if (rc_) return rc_; }   // System::Shutdown waits for the local mapper before the trajectory is saved
    const Map& m = h->c.seq[seq]->map;
    int n = 0;
    for (const KeyFrm& k : m.kfs) {
        if (k.bad) continue;
        if (n < cap) {
            if (stamps) stamps[n] = k.stamp;
            if (Twc)
                for (int r = 0; r < 3; r++) {
                    for (int cc = 0; cc < 3; cc++) Twc[(size_t)n * 12 + r * 4 + cc] = k.pose.Rwc[r * 3 + cc];
                    Twc[(size_t)n * 12 + r * 4 + 3] = k.pose.Ow[r];
                }
        }
        n++;
    }
    *n_out = n;
    if (n > cap) { oslam::set_error("keyframe trajectory: cap %d < %d", cap, n); return OSLAM_E_CAPACITY; }
    return OSLAM_OK;
}

int oslam_slam_stats(oslam_slam_t* h, int seq, int64_t out[16]) {
    if (!h || seq < 0 || seq >= h->c.S || !out) { oslam::set_error("oslam_slam_stats: bad argument"); return OSLAM_E_INVALID; }
    Seq& s = *h->c.seq[seq];
    memcpy(out, s.st, sizeof(s.st));
    out[2] = s.map.nKFsInMap; out[4] = s.map.nMPsInMap;
    out[15] = map_violations(s.map);
    return OSLAM_OK;
}

int oslam_slam_kernel_times(oslam_slam_t* h, int enable, double out[OSLAM_SLAM_KT_GROUPS * 3]) {
    if (!h) { oslam::set_error("oslam_slam_kernel_times: bad argument"); return OSLAM_E_INVALID; }
    if (!h->c.ops.kernel_times) { oslam::set_error("oslam_slam_kernel_times: this operator table has no device timing"); return OSLAM_E_INVALID; }
    return h->c.ops.kernel_times(h->c.ops.ctx, enable, out);
}

int oslam_slam_debug_point(oslam_slam_t* h, int seq, int id, uint8_t host[64], uint8_t resident[64], int32_t* bad) {
    if (!h || !host || !resident || !bad || seq < 0 || seq >= h->c.S) { oslam::set_error("oslam_slam_debug_point: bad argument"); return OSLAM_E_INVALID; }
    const Map& m = h->c.seq[seq]->map;
    if (id < 0 || id >= (int)m.mps.size()) { oslam::set_error("oslam_slam_debug_point: no such point"); return OSLAM_E_INVALID; }
    const MapPt& p = m.mps[id];
    static_assert(offsetof(MapPt, desc) == 32 && offsetof(MapPt, minD) == 24, "the first 64 bytes of MapPt are the resident record");
    memcpy(host, &p, 64);
    *bad = m.pBad[id] || p.obs.empty();
    if (!h->c.ops.point_record) { oslam::set_error("oslam_slam_debug_point: the operator table keeps no resident map points"); return OSLAM_E_INVALID; }
    return h->c.ops.point_record(h->c.ops.ctx, seq, id, resident);
}

int oslam_slam_bad_keyframe_observations(oslam_slam_t* h, int64_t* out) {
    if (!h || !out) { oslam::set_error("oslam_slam_bad_keyframe_observations: bad argument"); return OSLAM_E_INVALID; }
    *out = h->c.badKFObs.load();
    return OSLAM_OK;
}

int oslam_slam_local_map_reuse(oslam_slam_t* h, int64_t out[2]) {
    if (!h || !out) { oslam::set_error("oslam_slam_local_map_reuse: bad argument"); return OSLAM_E_INVALID; }
    out[0] = h->c.locReuse.load(); out[1] = h->c.locFrames.load();
    return OSLAM_OK;
}

int oslam_slam_struct_sizes(int32_t out[4]) {
    if (!out) return OSLAM_E_INVALID;
    out[0] = (int32_t)sizeof(oslam_slam_config_t); out[1] = (int32_t)sizeof(oslam_slam_ops_t); out[2] = (int32_t)sizeof(oslam_slam_objects_t); out[3] = (int32_t)sizeof(oslam_map_changes_t);
    return OSLAM_OK;
}

int oslam_slam_stage_cpu_seconds(oslam_slam_t* h, double out[16]) {
    if (!h || !out) { oslam::set_error("oslam_slam_stage_cpu_seconds: bad argument"); return OSLAM_E_INVALID; }
    memcpy(out, h->c.cpu, sizeof(h->c.cpu));
    return OSLAM_OK;
}

int oslam_slam_stage_seconds(oslam_slam_t* h, double out[16]) {
    if (!h || !out) { oslam::set_error("oslam_slam_stage_seconds: bad argument"); return OSLAM_E_INVALID; }
    memcpy(out, h->c.sec, sizeof(h->c.sec));
    return OSLAM_OK;
}

}  // extern "C"
