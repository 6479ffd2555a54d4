// Compile-only check of include/orb_slam2_adapter.hpp (no OpenCV, no GPU needed).
#include "../include/orb_slam2_adapter.hpp"
int main() {
    ORB_SLAM2::ORBextractor* e = nullptr;
    ORB_SLAM2::ORBmatcher* m = nullptr;
    (void)e; (void)m;
    return sizeof(oslam::KeyPoint) == 28 ? 0 : 1;
}
