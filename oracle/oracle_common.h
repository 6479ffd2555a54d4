// ORACLE — TEST INFRASTRUCTURE ONLY.
// CPU restatement of the reference hot path (yangliu9527/Object_SLAM, an ORB_SLAM2 fork).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything
// under oracle/.  The product (object_slam_amd/, include/) never links or calls it.
//
// PARITY UNPINNED: the reference has no tests / golden vectors and cannot be built here
// (OpenCV 3.2, g2o, PCL, DBoW2 absent).  The third-party arithmetic
// (cv::FAST, cv::resize, cv::GaussianBlur, cv::fastAtan2, g2o LM) is restated from the
// published OpenCV-3.2 / ORB_SLAM2-g2o algorithms; the in-tree reference code is followed
// line by line (file:line cited at each function).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace oracle {

// cv::KeyPoint POD mirror (OpenCV 3.2 types.hpp): 28 bytes.
struct KeyPoint {
    float x, y;      // pt
    float size;
    float angle;
    float response;
    int octave;
    int class_id;
};
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint layout");

// cvRound(float/double): SSE cvtss2si / lrint under the default rounding mode
// = round-half-to-even (OpenCV 3.2 fast_math.hpp).
static inline int cvRound(float v) { return (int)lrintf(v); }
static inline int cvRound(double v) { return (int)lrint(v); }
static inline int cvFloor(double v) { return (int)std::floor(v); }
static inline int cvCeil(double v) { return (int)std::ceil(v); }

struct Image {
    int w = 0, h = 0;
    std::vector<uint8_t> d;
    Image() {}
    Image(int w_, int h_) : w(w_), h(h_), d((size_t)w_ * h_) {}
    uint8_t* row(int y) { return d.data() + (size_t)y * w; }
    const uint8_t* row(int y) const { return d.data() + (size_t)y * w; }
};

}  // namespace oracle
