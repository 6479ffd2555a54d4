// Compile-only check of include/orb_slam2_adapter.hpp (no OpenCV, no GPU needed): every adapter class is instantiated in dead code so that
// all member templates and inline bodies are compiled and the C ABI symbols they use must resolve at link time.
#include "../include/orb_slam2_adapter.hpp"
int main(int argc, char**) {
    if (argc > 1000) {   // never taken
        ORB_SLAM2::ORBextractor e(1000, 1.2f, 8, 20, 7);
        ORB_SLAM2::ORBmatcher m(0.9f, true);
        ORB_SLAM2::PoseFrameView f{};
        ORB_SLAM2::SemanticView s{};
        ORB_SLAM2::BAGraph g{};
        ORB_SLAM2::Optimizer::PoseOptimization(f);
        ORB_SLAM2::ObjectOptimizer::PoseOptimization2(f, s);
        ORB_SLAM2::Optimizer::LocalBundleAdjustment(g);
        ORB_SLAM2::Optimizer::BundleAdjustment(g);
        (void)e; (void)m;
    }
    return sizeof(oslam::KeyPoint) == 28 ? 0 : 1;
}
