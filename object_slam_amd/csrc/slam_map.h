// slam_map.h — host data model of the batch-of-sequences driver (include/oslam_slam.h): the reference's Frame / KeyFrame /
// MapPoint / Map graph (include/Frame.h, include/KeyFrame.h, include/MapPoint.h, include/Map.h) restated over index-based
// arrays, one Map per sequence.  Containers the reference orders by pointer value are ordered by keyframe id here.
// Product code: never includes oracle/.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <set>
#include <utility>
#include <vector>

#include "../../include/oslam_slam.h"

namespace oslam_drv {

typedef oslam_keypoint_t KP;

struct M4 { float m[16]; };
inline M4 eye4() { M4 r; memset(r.m, 0, sizeof(r.m)); r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.f; return r; }
// cv::Mat CV_32F 4x4 * 4x4 (cv::gemm small-matrix branch: float products accumulated in float, left to right)
inline M4 mul4(const M4& a, const M4& b) {
    M4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            float s = a.m[i * 4] * b.m[j];
            s += a.m[i * 4 + 1] * b.m[4 + j];
            s += a.m[i * 4 + 2] * b.m[8 + j];
            s += a.m[i * 4 + 3] * b.m[12 + j];
            r.m[i * 4 + j] = s;
        }
    return r;
}
inline float norm3(const float* v) {   // cv::norm L2 of a CV_32F 3x1: double accumulation, sqrt in double
    double s = 0;
    for (int k = 0; k < 3; k++) s += (double)v[k] * (double)v[k];
    return (float)std::sqrt(s);
}

// Pose with the derived matrices the reference caches.
struct PoseM {
    M4 Tcw, Twc;
    float Rwc[9], Ow[3];
    bool valid = false;
    // Frame::SetPose / UpdatePoseMatrices (src/Frame.cc:478-505): mOw = -mRcw.t()*mtcw (transposed operand -> generic gemm, fp64 sums)
    void set_frame(const M4& T) {
        Tcw = T; valid = true;
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) Rwc[r * 3 + c] = T.m[c * 4 + r];
        for (int r = 0; r < 3; r++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += (double)T.m[k * 4 + r] * (double)T.m[k * 4 + 3];
            Ow[r] = (float)(-1.0 * s);
        }
        fill_twc();
    }
    // KeyFrame::SetPose (src/KeyFrame.cc:78-92): Rwc materialised, Ow = -Rwc*tcw (small-matrix branch, float sums)
    void set_keyframe(const M4& T) {
        Tcw = T; valid = true;
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) Rwc[r * 3 + c] = T.m[c * 4 + r];
        for (int r = 0; r < 3; r++) {
            float s = Rwc[r * 3] * T.m[3];
            s += Rwc[r * 3 + 1] * T.m[7];
            s += Rwc[r * 3 + 2] * T.m[11];
            Ow[r] = (float)((double)s * -1.0);
        }
        fill_twc();
    }
    void fill_twc() {
        Twc = eye4();
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) Twc.m[r * 4 + c] = Rwc[r * 3 + c];
            Twc.m[r * 4 + 3] = Ow[r];
        }
    }
};

struct Frame {
    int id = -1;
    double stamp = 0;
    int N = 0;
    std::vector<KP> keys, keysUn;
    std::vector<uint8_t> desc;
    std::vector<float> uRight, depth;
    std::vector<int> mp;            // mvpMapPoints: map point id or -1
    std::vector<uint8_t> outlier;   // mvbOutlier
    std::vector<uint32_t> bowNode;  // substitute FeatureVector: node id per keypoint (empty = not computed)
    PoseM pose;
    int refKF = -1;
    oslam_slam_frame_t view;
    // object layer (reference include/Frame.h:110-119): mvObject2Ds (detection index, caller's track id, mvFrameKpIndices), mvpObject3Ds, mvObjectKpIndices[k].first
    struct Obj2D { int det = -1, track = -1, obj3d = -1; std::vector<int> kps; };
    std::vector<Obj2D> objs;
    std::vector<int> objOfKp;
    void alloc(int cap) {
        keys.resize(cap); keysUn.resize(cap); desc.resize((size_t)cap * 32); uRight.resize(cap); depth.resize(cap);
        mp.assign(cap, -1); outlier.assign(cap, 0);
        view.N = 0; view.keys = keys.data(); view.keysUn = keysUn.data(); view.desc = desc.data(); view.uRight = uRight.data(); view.depth = depth.data();
    }
    void begin(int id_, double stamp_) {
        id = id_; stamp = stamp_; N = view.N;
        std::fill(mp.begin(), mp.end(), -1);
        std::fill(outlier.begin(), outlier.end(), 0);
        bowNode.clear(); pose.valid = false; refKF = -1;
        objs.clear(); objOfKp.assign(N, -1);
    }
};

// The keypoint behind an observation: mvKeysUn.pt, mvuRight and octave of keypoint `second` of keyframe `first`, copied when the observation is added (a
// keyframe's keypoints never change: include/KeyFrame.h:163-175 are const members).  Kept in a list PARALLEL to MapPt::obs: the loops that need the keypoint
// (the local-BA graph gather, KeyFrameCulling, the observation counts) then stay inside the point's two lists instead of taking a dependent cache miss in the
// keyframe's arrays per observation, and the loops that only need (keyframe, index) — UpdateLocalKeyFrames — keep walking 8 bytes per observation.
struct ObsKp { float x, y, ur; int octave; };

struct MapPt {
    float pos[3];
    float normal[3] = {0, 0, 0};
    float minD = 0, maxD = 0;
    uint8_t desc[32];
    // (nObs, mnVisible, mnFound, mbBad, mpReplaced and mnLastFrameSeen live in Map's dense per-point arrays: Map::pNObs ...)
    int firstKF = 0, firstFrame = 0, refKF = -1;
    std::vector<std::pair<int, int>> obs;        // (keyframe id, keypoint index), ascending keyframe id
    std::vector<ObsKp> okp;                      // okp[i] = the keypoint of obs[i]
    int trackRefForFrame = 0;   // zero-initialised like the reference
    // driver scratch of SearchInNeighbors: position of the point in the current keyframe's point list (valid while fuseListStamp == current keyframe id + 1)
    int fuseListIdx = 0, fuseListStamp = 0;
    // bookkeeping of the driver (not in the reference): obsVer counts the changes of the observation list; (updVer, updStep) = obsVer and the local-mapping
    // step of the last full update (descriptor + normal / depth), so that an update whose inputs cannot have changed since is not repeated
    int obsVer = 0, updVer = -1, updStep = -1;
    int obs_index(int kf) const {
        for (size_t i = 0; i < obs.size(); i++) if (obs[i].first == kf) return obs[i].second;
        return -1;
    }
};

// The driver's per-frame loops gather MapPt records by index in keypoint / list order, i.e. at random over a table of several MB per sequence: each
// loop requests the records of a later iteration early (a MapPt spans three cache lines; its observation list is a second, dependent access).
constexpr int kPF = 12;
inline void prefetch_mp(const MapPt* p) { __builtin_prefetch(p); __builtin_prefetch((const char*)p + 64); __builtin_prefetch((const char*)p + 128); }
template <class Ids>
inline void prefetch_ahead(const std::vector<MapPt>& mps, const Ids& ids, size_t i, size_t n) {
    if (i + kPF < n) { const int q = ids[i + kPF]; if (q >= 0) prefetch_mp(&mps[q]); }
}
template <class Ids>
inline void prefetch_obs_ahead(const std::vector<MapPt>& mps, const Ids& ids, size_t i, size_t n) {   // with prefetch_ahead: the list of a record requested kPF / 2 iterations ago
    prefetch_ahead(mps, ids, i, n);
    if (i + kPF / 2 < n) { const int q = ids[i + kPF / 2]; if (q >= 0) __builtin_prefetch(mps[q].obs.data()); }
}

template <class Ids>
inline void prefetch_okp_ahead(const std::vector<MapPt>& mps, const Ids& ids, size_t i, size_t n) {   // prefetch_obs_ahead + the parallel keypoint list
    prefetch_obs_ahead(mps, ids, i, n);
    if (i + kPF / 2 < n) { const int q = ids[i + kPF / 2]; if (q >= 0) __builtin_prefetch(mps[q].okp.data()); }
}

struct KeyFrm {
    int id = 0, frameId = 0;
    double stamp = 0;
    int N = 0;
    std::vector<KP> keys, keysUn;
    std::vector<uint8_t> desc, oct;               // oct[i] = keysUn[i].octave
    std::vector<float> uRight, depth;
    std::vector<int> mp;
    std::vector<uint32_t> bowNode;
    PoseM pose;
    M4 Tcp;                                       // pose relative to the parent, set when the keyframe is culled
    std::map<int, int> connW;                     // mConnectedKeyFrameWeights
    std::vector<int> ordered, orderedW;           // mvpOrderedConnectedKeyFrames / mvOrderedWeights
    int parent = -1;
    std::set<int> children;
    bool firstConnection = true, bad = false;
    int trackRefForFrame = 0, fuseTargetForKF = 0;
};

struct RelPose { M4 Tcr; int refKF; double stamp; bool lost; };

// One sequence's Map + the map-side methods of KeyFrame / MapPoint.
struct Map {
    std::vector<MapPt> mps;
    std::vector<KeyFrm> kfs;
    int nKFsInMap = 0, nMPsInMap = 0;
    int64_t nCulledKF = 0, nCulledMP = 0;

    // ---- MapPoint (src/MapPoint.cc) ----
    // The scalars the per-frame loops of tracking read or bump for every matched / visible point, dense by point id: nObs (Observations()), mnVisible, mnFound,
    // mnLastFrameSeen, mpReplaced (-1: none) and mbBad.  A loop over a frame's points then touches 4 bytes at neighbouring ids (the points a keyframe created have
    // consecutive ids) instead of one or two cache lines of a 170-byte MapPt record per point, whose maps (several MB per sequence, thousands of sequences per
    // rank) never stay in a cache from one frame to the next.
    std::vector<int> pNObs, pVisible, pFound, pLastSeen, pReplaced;
    std::vector<uint8_t> pBad;
    // Octave histogram of a point's observations, byte o of pLvl[p] = observations whose keypoint has octave o (kept beside obs / okp by the four functions that
    // change the lists).  KeyFrameCulling asks "at least three OTHER observations at octave <= l + 1": the histogram answers it without walking the lists unless
    // exactly three observations qualify (then it matters whether the keyframe's own observation is one of them).  lvlOverflow: a count reached 255 or an octave
    // was >= 8 — the histogram is then not trusted for this map and the lists are walked as before.
    std::vector<uint64_t> pLvl;
    bool lvlOverflow = false;
    void lvl_add(int p, int oct) {
        if ((unsigned)oct >= 8u || ((pLvl[p] >> (8 * oct)) & 0xFFull) == 0xFFull) { lvlOverflow = true; return; }
        pLvl[p] += 1ull << (8 * oct);
    }
    void lvl_sub(int p, int oct) {
        if ((unsigned)oct >= 8u || ((pLvl[p] >> (8 * oct)) & 0xFFull) == 0ull) { lvlOverflow = true; return; }
        pLvl[p] -= 1ull << (8 * oct);
    }
    int lvl_count_le(int p, int oct) const {   // observations of p at octave <= oct (valid while !lvlOverflow)
        const uint64_t h = oct >= 7 ? pLvl[p] : (pLvl[p] & ((1ull << (8 * (oct + 1))) - 1ull));
        uint64_t s2 = (h & 0x00FF00FF00FF00FFull) + ((h >> 8) & 0x00FF00FF00FF00FFull);   // four 16-bit partial sums
        s2 = (s2 & 0x0000FFFF0000FFFFull) + ((s2 >> 16) & 0x0000FFFF0000FFFFull);
        return (int)((s2 & 0xFFFFFFFFull) + (s2 >> 32));
    }
    // ---- change log for a DEVICE MIRROR of the observation graph (round 5; include/oslam_slam.h oslam_slam_ops_t::map_journal) ----
    // When an operator table keeps device copies of the keyframes' point lists, of "which point holds the observation (kf, idx)" and of the per-point scalars
    // (Observations(), bad flag, octave histogram), the map notes what changed — dirty cells of the point lists, the ordered AddObservation / EraseObservation
    // events, dirty points, new keyframes — and journal_words() turns the notes into the table's record stream with the CURRENT values of everything that only
    // needs its final state.  Off (jrOn = false) the map behaves as before.
    enum { JR_KFMP = 1, JR_KFMP_BULK = 2, JR_OKF_SET = 3, JR_OKF_CLR = 4, JR_PT = 5, JR_RESET = 6 };
    bool jrOn = false, jrReset = false;
    std::vector<std::pair<int, int>> jrCells;   // (kf, idx) whose mp entry changed
    std::vector<uint32_t> jrOkf;                // (kf, idx | set << 31, p) triples in program order
    std::vector<int> jrPts, jrNewKFs;           // dirty points (deduped through jrPtMark), keyframes created since the last flush
    std::vector<uint8_t> jrPtMark;
    bool jr_pending() const { return jrReset || !jrCells.empty() || !jrOkf.empty() || !jrPts.empty() || !jrNewKFs.empty(); }
    void jr_kfmp(int kf, int idx, int) { if (jrOn) jrCells.push_back(std::make_pair(kf, idx)); }
    void jr_okf(int op, int kf, int idx, int p) { if (jrOn) { jrOkf.push_back((uint32_t)kf); jrOkf.push_back((uint32_t)idx | (op == JR_OKF_SET ? 0x80000000u : 0u)); jrOkf.push_back((uint32_t)p); } }
    void jr_pt(int p) {
        if (!jrOn) return;
        if (jrPtMark.size() <= (size_t)p) jrPtMark.resize((size_t)p + 1 + jrPtMark.size() / 2, 0);
        if (!jrPtMark[p]) { jrPtMark[p] = 1; jrPts.push_back(p); }
    }
    void jr_new_kf(int kf) { if (jrOn) jrNewKFs.push_back(kf); }
    // The change set of oslam_slam_ops_t::map_journal for everything noted since the last call (the notes are cleared; the arrays the change set points into
    // live in `sc` and in the map's keyframes until the next call).
    struct JrScratch { std::vector<oslam_map_new_kf_t> kfs; std::vector<uint32_t> good, pts; std::vector<int32_t> cells; std::vector<uint32_t> events; };
    void journal_changes(int slot, float thDepth, JrScratch& sc, oslam_map_changes_t& ch) {
        ch.slot = slot; ch.reset = jrReset ? 1 : 0; jrReset = false;
        sc.kfs.clear(); sc.good.clear();
        size_t gw = 0;
        for (int kf : jrNewKFs) gw += (kfs[kf].mp.size() + 31) / 32;
        sc.good.resize(gw);
        gw = 0;
        for (int kf : jrNewKFs) {
            const KeyFrm& k = kfs[kf];
            const size_t N = k.mp.size(), nw = (N + 31) / 32;
            for (size_t wI = 0; wI < nw; wI++) {
                uint32_t bits = 0;
                for (size_t i = wI * 32; i < std::min(N, wI * 32 + 32); i++) bits |= (uint32_t)(!(k.depth[i] > thDepth || k.depth[i] < 0)) << (i & 31);
                sc.good[gw + wI] = bits;
            }
            oslam_map_new_kf_t e; e.kf = kf; e.N = (int32_t)N; e.mp = k.mp.data(); e.good = nullptr;
            sc.kfs.push_back(e);
            gw += nw;
        }
        gw = 0;
        for (auto& e : sc.kfs) { e.good = sc.good.data() + gw; gw += ((size_t)e.N + 31) / 32; }
        sc.cells.resize(jrCells.size() * 3);
        for (size_t i = 0; i < jrCells.size(); i++) { sc.cells[3 * i] = jrCells[i].first; sc.cells[3 * i + 1] = jrCells[i].second; sc.cells[3 * i + 2] = kfs[jrCells[i].first].mp[jrCells[i].second]; }
        sc.events.swap(jrOkf);
        sc.pts.resize(jrPts.size() * 5);
        for (size_t i = 0; i < jrPts.size(); i++) {
            const int p = jrPts[i];
            jrPtMark[p] = 0;
            sc.pts[5 * i] = (uint32_t)p; sc.pts[5 * i + 1] = (uint32_t)pNObs[p]; sc.pts[5 * i + 2] = (uint32_t)pBad[p];
            sc.pts[5 * i + 3] = (uint32_t)(pLvl[p] & 0xFFFFFFFFull); sc.pts[5 * i + 4] = (uint32_t)(pLvl[p] >> 32);
        }
        ch.n_new = (int32_t)sc.kfs.size(); ch.new_kfs = sc.kfs.data();
        ch.n_cells = (int32_t)jrCells.size(); ch.cells = sc.cells.data();
        ch.n_events = (int32_t)(sc.events.size() / 3); ch.events = sc.events.data();
        ch.n_points = (int32_t)jrPts.size(); ch.points = sc.pts.data();
        jrNewKFs.clear(); jrCells.clear(); jrOkf.clear(); jrPts.clear();
    }
    // kfs[kf].mp[idx] = p, noted (every write to a keyframe's point list outside this file goes through here)
    void set_kf_mp(int kf, int idx, int p) { kfs[kf].mp[idx] = p; jr_kfmp(kf, idx, p); }

    int new_point(const float x[3], int refKF, int refFrame) {
        MapPt p;
        p.pos[0] = x[0]; p.pos[1] = x[1]; p.pos[2] = x[2];
        memset(p.desc, 0, 32);
        p.firstKF = refKF; p.firstFrame = refFrame; p.refKF = refKF;
        mps.push_back(p);
        pNObs.push_back(0); pVisible.push_back(1); pFound.push_back(1);   // src/MapPoint.cc:33-46
        pLastSeen.push_back(0); pReplaced.push_back(-1); pBad.push_back(0); pLvl.push_back(0);
        return (int)mps.size() - 1;
    }
    void add_observation(int p, int kf, int idx) {   // :196-207
        MapPt& m = mps[p];
        size_t at = 0;
        while (at < m.obs.size() && m.obs[at].first < kf) at++;
        if (at < m.obs.size() && m.obs[at].first == kf) return;
        const KeyFrm& k = kfs[kf];
        const ObsKp o = {k.keysUn[idx].x, k.keysUn[idx].y, k.uRight[idx], k.keysUn[idx].octave};
        m.obs.insert(m.obs.begin() + at, std::make_pair(kf, idx));
        m.okp.insert(m.okp.begin() + at, o);
        lvl_add(p, o.octave);
        pNObs[p] += o.ur >= 0 ? 2 : 1;
        m.obsVer++;
        jr_okf(JR_OKF_SET, kf, idx, p); jr_pt(p);
    }
    void set_bad_point(int p) {                      // :253-270
        MapPt& m = mps[p];
        if (!pBad[p]) { nMPsInMap--; nCulledMP++; }
        pBad[p] = 1;
        std::vector<std::pair<int, int>> o;
        o.swap(m.obs);
        std::vector<ObsKp>().swap(m.okp);   // (a bad point never gets an observation again: its lists' memory goes back, 40 % of the points a sequence creates end here)
        pLvl[p] = 0;
        for (auto& e : o) { set_kf_mp(e.first, e.second, -1); jr_okf(JR_OKF_CLR, e.first, e.second, p); }
        jr_pt(p);
    }
    void erase_observation(int p, int kf) {          // :209-239
        MapPt& m = mps[p];
        bool bad = false;
        for (size_t i = 0; i < m.obs.size(); i++)
            if (m.obs[i].first == kf) {
                pNObs[p] -= m.okp[i].ur >= 0 ? 2 : 1;
                jr_okf(JR_OKF_CLR, kf, m.obs[i].second, p);
                m.obs.erase(m.obs.begin() + i);
                lvl_sub(p, m.okp[i].octave);
                m.okp.erase(m.okp.begin() + i);
                m.obsVer++;
                if (m.refKF == kf && !m.obs.empty()) m.refKF = m.obs.front().first;
                if (pNObs[p] <= 2) bad = true;
                jr_pt(p);
                break;
            }
        if (bad) set_bad_point(p);
    }
    // MapPoint::Replace (:279-318) without the descriptor recomputation (the caller batches it): returns true if `by` changed
    bool replace_point(int p, int by) {
        if (p == by) return false;
        MapPt& m = mps[p];
        std::vector<std::pair<int, int>> o;
        o.swap(m.obs);
        std::vector<ObsKp>().swap(m.okp);
        pLvl[p] = 0;
        if (!pBad[p]) { nMPsInMap--; }
        pBad[p] = 1;
        pReplaced[p] = by;
        const int nvisible = pVisible[p], nfound = pFound[p];
        for (auto& e : o) {
            jr_okf(JR_OKF_CLR, e.first, e.second, p);
            if (mps[by].obs_index(e.first) < 0) {
                set_kf_mp(e.first, e.second, by);
                add_observation(by, e.first, e.second);
            } else {
                set_kf_mp(e.first, e.second, -1);
            }
        }
        jr_pt(p);
        pFound[by] += nfound;
        pVisible[by] += nvisible;
        return true;
    }

    // ---- KeyFrame covisibility graph (src/KeyFrame.cc:123-379) ----
    void update_best_covisibles(int k) {             // :138-157: descending (weight, id)
        KeyFrm& f = kfs[k];
        static thread_local std::vector<std::pair<int, int>> v;   // scratch
        v.clear();
        for (auto& e : f.connW) v.push_back(std::make_pair(e.second, e.first));
        std::sort(v.begin(), v.end());
        f.ordered.clear(); f.orderedW.clear();
        for (size_t i = v.size(); i-- > 0;) { f.ordered.push_back(v[i].second); f.orderedW.push_back(v[i].first); }
    }
    void add_connection(int k, int other, int w) {   // :123-136
        KeyFrm& f = kfs[k];
        auto it = f.connW.find(other);
        if (it == f.connW.end()) f.connW[other] = w;
        else if (it->second != w) it->second = w;
        else return;
        update_best_covisibles(k);
    }
    void erase_connection(int k, int other) {        // :553-567
        if (kfs[k].connW.erase(other)) update_best_covisibles(k);
    }
    struct IntSpan {   // view of the first entries of a keyframe's ordered list (valid until that list changes)
        const int* b; const int* e;
        const int* begin() const { return b; }
        const int* end() const { return e; }
        size_t size() const { return (size_t)(e - b); }
    };
    IntSpan best_covisibles(int k, int n) const {   // :174-182 (the reference returns a copy: per frame and local keyframe that is an allocation)
        const KeyFrm& f = kfs[k];
        const size_t m = std::min<size_t>(f.ordered.size(), (size_t)std::max(n, 0));
        return IntSpan{f.ordered.data(), f.ordered.data() + m};
    }
    int weight(int k, int other) const {
        auto it = kfs[k].connW.find(other);
        return it == kfs[k].connW.end() ? 0 : it->second;
    }
    void update_connections(int k, std::vector<int>& counter /* scratch, size >= kfs.size(), zeros */) {   // :289-379
        KeyFrm& f = kfs[k];
        static thread_local std::vector<int> touched;   // scratch
        touched.clear();
        for (int i = 0; i < f.N; i++) {
            prefetch_obs_ahead(mps, f.mp, i, f.N);
            const int p = f.mp[i];
            if (p < 0 || pBad[p]) continue;
            for (auto& e : mps[p].obs) {
                if (e.first == k) continue;
                if (counter[e.first]++ == 0) touched.push_back(e.first);
            }
        }
        if (touched.empty()) return;
        std::sort(touched.begin(), touched.end());
        int nmax = 0, kmax = -1;
        const int th = 15;
        static thread_local std::vector<std::pair<int, int>> v2;   // scratch (add_connection below uses update_best_covisibles' own)
        std::vector<std::pair<int, int>>& v = v2;
        v.clear();
        for (int o : touched) {
            const int c = counter[o];
            if (c > nmax) { nmax = c; kmax = o; }
            if (c >= th) { v.push_back(std::make_pair(c, o)); add_connection(o, k, c); }
        }
        if (v.empty()) { v.push_back(std::make_pair(nmax, kmax)); add_connection(kmax, k, nmax); }
        std::sort(v.begin(), v.end());
        f.connW.clear();
        for (int o : touched) { f.connW[o] = counter[o]; counter[o] = 0; }
        f.ordered.clear(); f.orderedW.clear();
        for (size_t i = v.size(); i-- > 0;) { f.ordered.push_back(v[i].second); f.orderedW.push_back(v[i].first); }
        if (f.firstConnection && f.id != 0) {
            f.parent = f.ordered.front();
            kfs[f.parent].children.insert(k);
            f.firstConnection = false;
        }
    }
    int tracked_map_points(int k, int minObs) const {   // :250-275
        const KeyFrm& f = kfs[k];
        int n = 0;
        for (int i = 0; i < f.N; i++) {
            const int p = f.mp[i];
            if (p < 0 || pBad[p]) continue;
            if (minObs > 0) { if (pNObs[p] >= minObs) n++; }
            else n++;
        }
        return n;
    }
    void set_bad_keyframe(int k) {                   // :453-545 (mbNotErase is only set by the loop closer: never here)
        KeyFrm& f = kfs[k];
        if (f.id == 0) return;
        const bool first = !f.bad;   // KeyFrameCulling can reach a culled keyframe again through a stale one-sided link; the reference then repeats the body
        for (auto& e : f.connW) erase_connection(e.first, k);
        for (int i = 0; i < f.N; i++)
            if (f.mp[i] >= 0) erase_observation(f.mp[i], k);
        f.connW.clear(); f.ordered.clear(); f.orderedW.clear();
        std::set<int> cand;
        cand.insert(f.parent);
        while (!f.children.empty()) {
            bool cont = false;
            int maxw = -1, pC = -1, pP = -1;
            for (int c : f.children) {
                if (kfs[c].bad) continue;
                for (int conn : kfs[c].ordered)
                    for (int pc : cand)
                        if (conn == pc) {
                            const int w = weight(c, conn);
                            if (w > maxw) { pC = c; pP = conn; maxw = w; cont = true; }
                        }
            }
            if (!cont) break;
            kfs[pC].parent = pP;
            kfs[pP].children.insert(pC);
            cand.insert(pC);
            f.children.erase(pC);
        }
        for (int c : f.children) { kfs[c].parent = f.parent; kfs[f.parent].children.insert(c); }
        kfs[f.parent].children.erase(k);
        f.Tcp = mul4(f.pose.Tcw, kfs[f.parent].pose.Twc);
        f.bad = true;
        if (first) { nKFsInMap--; nCulledKF++; }
    }
};

}  // namespace oslam_drv
