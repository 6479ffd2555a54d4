// Host-side unit checks of the driver's map bookkeeping (object_slam_amd/csrc/slam_map.h) against the behaviour of the reference's
// KeyFrame / MapPoint methods (src/KeyFrame.cc:123-567, src/MapPoint.cc:196-318), on a hand-built map.  No GPU, no oracle.
#include <cstdio>
#include <cstdlib>

#include "../object_slam_amd/csrc/slam_map.h"

using namespace oslam_drv;

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "CHECK failed at line %d: %s\n", __LINE__, #cond); return 1; } } while (0)

static int add_kf(Map& m, int nkp, bool stereo) {
    m.kfs.emplace_back();
    KeyFrm& k = m.kfs.back();
    k.id = (int)m.kfs.size() - 1; k.frameId = k.id * 5; k.N = nkp;
    k.keys.resize(nkp); k.keysUn.resize(nkp); k.desc.assign((size_t)nkp * 32, 0); k.uRight.assign(nkp, stereo ? 10.f : -1.f); k.depth.assign(nkp, stereo ? 2.f : -1.f);
    for (int i = 0; i < nkp; i++) { k.keysUn[i].octave = 0; k.keys[i].octave = 0; }
    k.mp.assign(nkp, -1);
    M4 T = eye4(); T.m[3] = -0.1f * k.id;
    k.pose.set_keyframe(T);
    k.Tcp = eye4();
    m.nKFsInMap++;
    return k.id;
}
static void observe(Map& m, int p, int kf, int idx) { m.add_observation(p, kf, idx); m.kfs[kf].mp[idx] = p; }

int main() {
    Map m;
    std::vector<int> counter(64, 0);
    const int NK = 50;   // slots 40..49 stay free for the single-point scenarios
    for (int k = 0; k < 5; k++) add_kf(m, NK, true);
    // points 0..19 seen by keyframes 0,1,2; points 20..39 seen by 1,2,3,4; point 40 only by 4 and 3
    const float x[3] = {0, 0, 2};
    for (int p = 0; p < 41; p++) { m.new_point(x, 0, 0); m.nMPsInMap++; }
    for (int p = 0; p < 20; p++) for (int k = 0; k <= 2; k++) observe(m, p, k, p);
    for (int p = 20; p < 40; p++) for (int k = 1; k <= 4; k++) observe(m, p, k, p);
    CHECK(m.pNObs[0] == 6);     // stereo observations count twice (src/MapPoint.cc:203-206)
    CHECK(m.pNObs[20] == 8);
    for (int k = 0; k < 5; k++) m.update_connections(k, counter);
    for (int v : counter) CHECK(v == 0);
    // weights: (0,1)=20 (0,2)=20 (1,2)=40 (1,3)=20 (1,4)=20 (2,3)=20 (2,4)=20 (3,4)=20
    CHECK(m.weight(1, 2) == 40 && m.weight(2, 1) == 40 && m.weight(0, 3) == 0 && m.weight(3, 4) == 20);
    // ordered by descending (weight, id): keyframe 1 sees 2 (40) first, then ids 4, 3, 0 (20 each)
    CHECK(m.kfs[1].ordered.size() == 4 && m.kfs[1].ordered[0] == 2 && m.kfs[1].ordered[1] == 4 && m.kfs[1].ordered[2] == 3 && m.kfs[1].ordered[3] == 0);
    // spanning tree (first connection, src/KeyFrame.cc:371-376): 1 -> 0 (only keyframe 0 existed with links when 1 connected)
    CHECK(m.kfs[0].parent == -1);
    CHECK(m.kfs[1].parent == 2 || m.kfs[1].parent == 0);   // all observations already present in this construction: best covisible
    CHECK(m.kfs[3].parent >= 0 && m.kfs[m.kfs[3].parent].children.count(3) == 1);
    // th = 15: a link below the threshold only survives as the single best link (src/KeyFrame.cc:348-352)
    const int k5 = add_kf(m, NK, true);
    for (int p = 0; p < 5; p++) observe(m, p, k5, p);
    m.update_connections(k5, counter);
    CHECK(m.kfs[k5].ordered.size() == 1 && m.kfs[k5].orderedW[0] == 5);
    CHECK(m.kfs[k5].connW.size() == 3);   // mConnectedKeyFrameWeights keeps every counter (KFcounter), :367
    // erase_observation: refKF moves to the first remaining observer; <= 2 observations left -> point culled (src/MapPoint.cc:209-239)
    // (free slot 38 of keyframes 3 and 4 first: point 38 loses two stereo observations)
    m.erase_observation(38, 3); m.erase_observation(38, 4);
    CHECK(m.pNObs[38] == 4 && !m.pBad[38] && m.kfs[3].mp[38] == 38);   // EraseObservation does not touch the keyframe slot; the caller does (src/Optimizer.cc:752-753)
    m.kfs[3].mp[38] = -1; m.kfs[4].mp[38] = -1;
    m.kfs[3].uRight[38] = -1.f; m.kfs[4].uRight[38] = -1.f;   // two monocular observations: nObs = 2
    observe(m, 40, 3, 38); observe(m, 40, 4, 38);
    m.mps[40].refKF = 3;
    CHECK(m.pNObs[40] == 2);
    m.erase_observation(40, 3);
    CHECK(m.pBad[40] && m.mps[40].obs.empty() && m.kfs[4].mp[38] == -1);   // nObs <= 2 -> SetBadFlag clears the remaining slot
    // Replace (src/MapPoint.cc:279-318): observations move unless the target is already in that keyframe
    const int pa = m.new_point(x, 0, 0), pb = m.new_point(x, 0, 0);
    m.nMPsInMap += 2;
    observe(m, pa, 0, 45); observe(m, pa, 1, 46); observe(m, pb, 1, 47); observe(m, pb, 2, 48);
    m.pFound[pa] = 3; m.pVisible[pa] = 7;
    const int f0 = m.pFound[pb], v0 = m.pVisible[pb];
    CHECK(m.replace_point(pa, pb));
    CHECK(m.pBad[pa] && m.pReplaced[pa] == pb && m.mps[pa].obs.empty());
    CHECK(m.kfs[0].mp[45] == pb && m.kfs[1].mp[46] == -1 && m.kfs[1].mp[47] == pb);   // keyframe 1 already observed pb: pa's slot is erased
    CHECK(m.mps[pb].obs.size() == 3 && m.pFound[pb] == f0 + 3 && m.pVisible[pb] == v0 + 7);
    // SetBadFlag of keyframe 2 (src/KeyFrame.cc:453-545): links and observations removed, children re-attached, mTcp stored, id 0 is immune
    const int nk = m.nKFsInMap;
    m.set_bad_keyframe(0);
    CHECK(!m.kfs[0].bad && m.nKFsInMap == nk);
    const int par2 = m.kfs[2].parent;
    std::set<int> ch2 = m.kfs[2].children;
    m.set_bad_keyframe(2);
    CHECK(m.kfs[2].bad && m.nKFsInMap == nk - 1 && m.kfs[2].ordered.empty() && m.kfs[2].connW.empty());
    for (int k = 0; k < (int)m.kfs.size(); k++) {
        if (k == 2) continue;
        // keyframe 5 keeps its one-sided counter for keyframe 2: mConnectedKeyFrameWeights = KFcounter holds every co-observer (src/KeyFrame.cc:367) while
        // only links >= 15 (or the single best) are mirrored (:341-352), and SetBadFlag erases the mirrored ones (:466-467)
        if (k != k5) CHECK(m.weight(k, 2) == 0);
        else CHECK(m.weight(k, 2) == 5);
        for (int o : m.kfs[k].ordered) CHECK(o != 2);
        if (m.kfs[k].parent >= 0) CHECK(!m.kfs[m.kfs[k].parent].bad);
    }
    for (int c : ch2) CHECK(m.kfs[c].parent != 2 && m.kfs[c].parent >= 0);
    if (par2 >= 0) CHECK(m.kfs[par2].children.count(2) == 0);
    for (int p = 0; p < 40; p++) CHECK(m.mps[p].obs_index(2) < 0);
    // points 0..19 had observers 0,1,2 (+5 for the first five): after losing keyframe 2 they keep nObs = 4 (or 6)
    CHECK(m.pNObs[10] == 4 && !m.pBad[10] && m.pNObs[0] == 6);
    // Tcp = Tcw * parent.Twc
    const M4 Tcp = mul4(m.kfs[2].pose.Tcw, m.kfs[par2].pose.Twc);
    for (int i = 0; i < 16; i++) CHECK(m.kfs[2].Tcp.m[i] == Tcp.m[i]);
    printf("slam_map_check ok\n");
    return 0;
}
