// A small caller written ONLY against the adapter classes of include/orb_slam2_adapter.hpp (ORB_SLAM2::ORBextractor, ORBmatcher, Optimizer,
// ObjectOptimizer, ComputeStereoMatches) — the way the reference's Frame / Tracking / LocalMapping code would call them.  It reads raw arrays from
// a directory (written by tests/test_adapter_gpu.py), runs every adapter method once and writes the results back; the test compares them with
// the ctypes path.  Usage: adapter_program <dir>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>

#include "../include/orb_slam2_adapter.hpp"

static std::string g_dir;
template <class T>
static std::vector<T> rd(const std::string& name) {
    std::ifstream f(g_dir + "/" + name + ".bin", std::ios::binary | std::ios::ate);
    if (!f) { std::cerr << "missing " << name << "\n"; exit(2); }
    const size_t n = (size_t)f.tellg();
    std::vector<T> v(n / sizeof(T));
    f.seekg(0);
    f.read((char*)v.data(), n);
    return v;
}
template <class T>
static void wr(const std::string& name, const T* p, size_t n) {
    std::ofstream f(g_dir + "/out_" + name + ".bin", std::ios::binary);
    f.write((const char*)p, n * sizeof(T));
}
template <class T>
static void wr(const std::string& name, const std::vector<T>& v) { wr(name, v.data(), v.size()); }

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    g_dir = argv[1];
    std::map<std::string, double> meta;
    {
        std::ifstream f(g_dir + "/meta.txt");
        std::string k; double v;
        while (f >> k >> v) meta[k] = v;
    }
    auto I = [&](const char* k) { return (int)meta.at(k); };
    auto F = [&](const char* k) { return (float)meta.at(k); };
    using namespace ORB_SLAM2;
    try {
        // ---- ORBextractor on the left and the right image of a stereo pair ----
        const int W = I("W"), H = I("H");
        std::vector<uint8_t> imL = rd<uint8_t>("imL"), imR = rd<uint8_t>("imR");
        ORBextractor exL(I("nFeatures"), 1.2f, 8, 20, 7), exR(I("nFeatures"), 1.2f, 8, 20, 7);
        std::vector<oslam::KeyPoint> kL, kR;
        std::vector<uint8_t> dL, dR;
        oslam::Image8 noMask = {nullptr, 0, 0, 0};
        exL(oslam::Image8{imL.data(), W, H, W}, noMask, kL, dL);
        exR(oslam::Image8{imR.data(), W, H, W}, kR, dR);
        wr("kL", kL); wr("dL", dL); wr("kR", kR); wr("dR", dR);
        wr("scale", exL.GetScaleFactors()); wr("invsigma2", exL.GetInverseScaleSigmaSquares());
        // ---- Frame::ComputeStereoMatches ----
        std::vector<float> uR, depth;
        ComputeStereoMatches(exL, exR, kL, dL, kR, dR, F("bf"), F("b"), uR, depth);
        wr("uR", uR); wr("depth", depth);
        // ---- ORBmatcher: SearchByProjection(Cur, Last), SearchByProjection(F, points), Fuse ----
        std::vector<oslam::KeyPoint> curK = rd<oslam::KeyPoint>("cur_keys"), lastK = rd<oslam::KeyPoint>("last_keys");
        std::vector<uint8_t> curD = rd<uint8_t>("cur_desc"), lastD = rd<uint8_t>("last_desc"), lastHas = rd<uint8_t>("last_has");
        std::vector<float> curUR = rd<float>("cur_uR"), lastXw = rd<float>("last_Xw"), Tcw = rd<float>("Tcw"), Tlw = rd<float>("Tlw");
        FrameView cur = {(int)curK.size(), curK.data(), curUR.data(), curD.data(), nullptr, 0.f, 0.f, (float)I("mW"), (float)I("mH")};
        oslam_camera_t cam = {F("fx"), F("fy"), F("cx"), F("cy"), F("mbf"), F("mbf") / F("fx")};
        ORBmatcher m9(0.9f, true);
        std::vector<int32_t> km;
        std::vector<float> sf = rd<float>("scaleFactors");
        const int nm = m9.SearchByProjection(cur, (int)lastK.size(), lastXw.data(), lastHas.data(), lastK.data(), lastD.data(), Tcw.data(), Tlw.data(), cam, sf, 15.f, false, km);
        wr("last_kp_match", km);
        std::vector<oslam_proj_query_t> q = rd<oslam_proj_query_t>("queries");
        ORBmatcher m8(0.8f, true);
        std::vector<int32_t> km2, qm2;
        const int nm2 = m8.SearchByProjection(cur, q, km2, &qm2);
        wr("proj_kp_match", km2); wr("proj_q_match", qm2);
        std::vector<float> invs2 = rd<float>("invSigma2");
        std::vector<int32_t> qf;
        const int nf = m8.Fuse(cur, q, invs2, qf);
        wr("fuse_q_match", qf);
        // ---- ORBmatcher: SearchByBoW / SearchForTriangulation with a caller-supplied FeatureVector ----
        ORBmatcher::FeatureVector fvA, fvB;
        fvA.q_idx = rd<int32_t>("fvA_qidx"); fvA.q_node = rd<uint32_t>("fvA_qnode"); fvA.nodes = rd<uint32_t>("fvA_nodes"); fvA.start = rd<int32_t>("fvA_start"); fvA.items = rd<int32_t>("fvA_items");
        fvB.q_idx = rd<int32_t>("fvB_qidx"); fvB.q_node = rd<uint32_t>("fvB_qnode"); fvB.nodes = rd<uint32_t>("fvB_nodes"); fvB.start = rd<int32_t>("fvB_start"); fvB.items = rd<int32_t>("fvB_items");
        FrameView last = {(int)lastK.size(), lastK.data(), nullptr, lastD.data(), nullptr, 0.f, 0.f, (float)I("mW"), (float)I("mH")};
        std::vector<uint8_t> flagA(lastK.size(), 1);
        ORBmatcher m7(0.7f, true);
        std::vector<int32_t> bm;
        const int nb = m7.SearchByBoW(last, fvA, flagA.data(), cur, fvB, bm);
        wr("bow_match", bm);
        std::vector<float> F12 = rd<float>("F12"), s2 = rd<float>("sigma2");
        std::vector<uint8_t> none1(lastK.size(), 0), none2(curK.size(), 0);
        std::vector<float> lastUR(lastK.size(), -1.f);
        FrameView k1 = {(int)lastK.size(), lastK.data(), lastUR.data(), lastD.data(), nullptr, 0, 0, 0, 0};
        ORBmatcher m6(0.6f, false);
        std::vector<int32_t> tm;
        const int nt = m6.SearchForTriangulation(k1, fvA, none1.data(), cur, fvB, none2.data(), F12.data(), F("ex"), F("ey"), sf, s2, false, tm);
        wr("tri_match", tm);
        const int dd = ORBmatcher::DescriptorDistance(curD.data(), lastD.data());
        // ---- Optimizer::PoseOptimization / ObjectOptimizer::PoseOptimization2 ----
        std::vector<float> pT = rd<float>("pose_T"), pXw = rd<float>("pose_Xw"), pUR = rd<float>("pose_uR");
        std::vector<oslam::KeyPoint> pK = rd<oslam::KeyPoint>("pose_keys");
        std::vector<uint8_t> pHas = rd<uint8_t>("pose_has"), outl(pK.size() + 1);
        PoseFrameView pf = {(int)pK.size(), pT.data(), pXw.data(), pHas.data(), pK.data(), pUR.data(), invs2.data(), outl.data(), F("fx"), F("fy"), F("cx"), F("cy"), F("mbf")};
        const int ninl = Optimizer::PoseOptimization(pf);
        wr("pose_T", pT); wr("pose_outlier", outl.data(), pK.size());
        std::vector<float> pT2 = rd<float>("pose_T");
        std::vector<uint8_t> masks = rd<uint8_t>("sem_masks"), outl2(pK.size() + 1);
        std::vector<float> oXw = rd<float>("sem_objmp_Xw");
        std::vector<int32_t> oObj = rd<int32_t>("sem_objmp_obj"), jk = rd<int32_t>("sem_joint_kp"), jo = rd<int32_t>("sem_joint_obj");
        PoseFrameView pf2 = pf;
        pf2.mTcw = pT2.data(); pf2.mvbOutlier = outl2.data();
        SemanticView sv = {I("sem_nObj"), I("mH"), I("mW"), masks.data(), (int)oObj.size(), oXw.data(), oObj.data(), (int)jk.size(), jk.data(), jo.data(), 0.f, 0.f, (float)I("mW"), (float)I("mH")};
        int nsem = 0;
        const int ninl2 = ObjectOptimizer::PoseOptimization2(pf2, sv, &nsem);
        wr("pose2_T", pT2); wr("pose2_outlier", outl2.data(), pK.size());
        // ---- Optimizer::LocalBundleAdjustment / BundleAdjustment ----
        std::vector<float> poses = rd<float>("ba_poses"), points = rd<float>("ba_points"), eobs = rd<float>("ba_eobs"), einv = rd<float>("ba_einv"), K5 = rd<float>("ba_K5");
        std::vector<uint8_t> fixed = rd<uint8_t>("ba_fixed");
        std::vector<int32_t> ekf = rd<int32_t>("ba_ekf"), ept = rd<int32_t>("ba_ept");
        std::vector<uint8_t> erase(ekf.size() + 1);
        std::vector<float> poses2 = poses, points2 = points;
        BAGraph g = {(int)fixed.size(), poses.data(), fixed.data(), (int)points.size() / 3, points.data(), (int)ekf.size(), ekf.data(), ept.data(), eobs.data(), einv.data(), erase.data(),
                     K5[0], K5[1], K5[2], K5[3], K5[4]};
        bool stop = false;
        Optimizer::LocalBundleAdjustment(g, &stop);
        wr("lba_poses", poses); wr("lba_points", points); wr("lba_erase", erase.data(), ekf.size());
        BAGraph g2 = g;
        g2.poses = poses2.data(); g2.points = points2.data(); g2.erase = nullptr;
        Optimizer::BundleAdjustment(g2, 5, nullptr, true);
        wr("ba_poses", poses2); wr("ba_points", points2);
        std::ofstream r(g_dir + "/out_results.txt");
        r << "nmatches_last " << nm << "\nnmatches_proj " << nm2 << "\nnfused " << nf << "\nnbow " << nb << "\nntri " << nt << "\ndist " << dd << "\nninliers " << ninl
          << "\nninliers2 " << ninl2 << "\nnsem " << nsem << "\n";
    } catch (const std::exception& e) {
        std::cerr << "adapter_program: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
