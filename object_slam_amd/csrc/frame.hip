// gfx950 Frame construction steps between the extractor and the grid / matchers (the callers' side of SURVEY.md §8 A-8/A-9):
//   Frame::UndistortKeyPoints      reference src/Frame.cc:644-675 (cv::undistortPoints, OpenCV 3.2 cvUndistortPoints)
//   Frame::ComputeImageBounds      reference src/Frame.cc:677-704
//   Frame::ComputeStereoFromRGBD   reference src/Frame.cc:883-904
// Batched over the frames of one extractor batch; everything stays in HBM between the extractor and the matcher.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "common.h"

namespace oslam {

struct UndistortCtx {
    const oslam_keypoint_t* keys; oslam_keypoint_t* keysUn; const int* counts; int n_const; int stride;
    double fx, fy, cx, cy, ifx, ify, k[12];
    int iters, passthrough;
};

// cvUndistortPoints with R = I, P = K: fp64 fixed-point inversion of the distortion model
__device__ __forceinline__ void undistort_point(const UndistortCtx& c, float xin, float yin, float& xo, float& yo) {
    double x = (double)xin, y = (double)yin, x0, y0;
    x0 = x = (x - c.cx) * c.ifx;
    y0 = y = (y - c.cy) * c.ify;
    for (int j = 0; j < c.iters; j++) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((c.k[7] * r2 + c.k[6]) * r2 + c.k[5]) * r2) / (1 + ((c.k[4] * r2 + c.k[1]) * r2 + c.k[0]) * r2);
        const double deltaX = 2 * c.k[2] * x * y + c.k[3] * (r2 + 2 * x * x) + c.k[8] * r2 + c.k[9] * r2 * r2;
        const double deltaY = c.k[2] * (r2 + 2 * y * y) + 2 * c.k[3] * x * y + c.k[10] * r2 + c.k[11] * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    // RR = K * I: rows (fx, 0, cx), (0, fy, cy), (0, 0, 1)
    const double xx = c.fx * x + 0.0 * y + c.cx;
    const double yy = 0.0 * x + c.fy * y + c.cy;
    const double ww = 1. / (0.0 * x + 0.0 * y + 1.0);
    xo = (float)(xx * ww);
    yo = (float)(yy * ww);
}

__global__ __launch_bounds__(256) void k_undistort(UndistortCtx c) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int n = c.counts ? c.counts[b] : c.n_const;
    if (i >= n || i >= c.stride) return;
    oslam_keypoint_t kp = c.keys[(long long)b * c.stride + i];
    if (!c.passthrough) undistort_point(c, kp.x, kp.y, kp.x, kp.y);
    c.keysUn[(long long)b * c.stride + i] = kp;
}

struct RgbdCtx {
    const oslam_keypoint_t* keys; const oslam_keypoint_t* keysUn; const int* counts; int n_const; int stride;
    const float* depth; int pitch; long long image_stride; int rows, cols;
    float mbf; float* uRight; float* mvDepth; int* status;
    const float* const* depth_ptrs;   // batch of depth images given by a pointer table (image b = depth_ptrs[b]) instead of base + b * image_stride
    const uint16_t* const* depth16_ptrs;   // the same with RAW 16-bit depth images, scaled on lookup by `factor` like imDepth.convertTo(CV_32F, mDepthMapFactor)
    float factor;                          // (reference src/Tracking.cc:262; OpenCV 3.2 cvtScale_<ushort, float, float>: (float)d * scale in float arithmetic)
};

__global__ __launch_bounds__(256) void k_stereo_from_rgbd(RgbdCtx c) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int n = c.counts ? c.counts[b] : c.n_const;
    if (i >= c.stride) return;
    const long long o = (long long)b * c.stride + i;
    float ur = -1.f, dp = -1.f;
    if (i < n) {
        const oslam_keypoint_t kp = c.keys[o];
        const int row = (int)kp.y, col = (int)kp.x;   // Mat::at<float>(v, u) with float arguments: truncation
        if (row < 0 || row >= c.rows || col < 0 || col >= c.cols) atomicOr(c.status, 1);   // the reference would read out of the image
        else {
            const float d = c.depth16_ptrs ? __fmul_rn((float)c.depth16_ptrs[b][(long long)row * c.pitch + col], c.factor)
                          : c.depth_ptrs ? c.depth_ptrs[b][(long long)row * c.pitch + col] : c.depth[(long long)b * c.image_stride + (long long)row * c.pitch + col];
            if (d > 0) { dp = d; ur = c.keysUn[o].x - __fdiv_rn(c.mbf, d); }
        }
    }
    c.uRight[o] = ur;
    c.mvDepth[o] = dp;
}

static int fill_undistort(UndistortCtx& c, const float K4[4], const float* dist, int ndist) {
    if (!K4 || ndist < 0 || ndist > 12 || (ndist > 0 && !dist) || (ndist != 0 && ndist != 4 && ndist != 5 && ndist != 8 && ndist != 12)) {
        set_error("distortion vector must have 0, 4, 5, 8 or 12 entries");
        return OSLAM_E_INVALID;
    }
    for (int i = 0; i < 12; i++) c.k[i] = i < ndist ? (double)dist[i] : 0.0;
    c.iters = ndist > 0 ? 5 : 1;
    c.passthrough = (ndist == 0 || dist[0] == 0.0f) ? 1 : 0;   // mDistCoef.at<float>(0)==0.0 -> mvKeysUn = mvKeys
    c.fx = (double)K4[0]; c.fy = (double)K4[1]; c.cx = (double)K4[2]; c.cy = (double)K4[3];
    c.ifx = 1. / c.fx; c.ify = 1. / c.fy;
    return OSLAM_OK;
}

// Gathers n images given by a table of device pointers into one contiguous batch [n][rows][dst_pitch] (the extractor's input layout): ONE launch
// instead of n 2-D copies.  16-byte chunks where source and destination rows are 16-byte aligned, bytes otherwise.
__global__ __launch_bounds__(256) void k_gather_images(const uint8_t* const* src, int src_pitch, int row_bytes, uint8_t* dst, size_t dst_image_stride, int dst_pitch) {
    const int row = blockIdx.y, img = blockIdx.z;
    const uint8_t* s = src[img] + (size_t)row * src_pitch;
    uint8_t* d = dst + (size_t)img * dst_image_stride + (size_t)row * dst_pitch;
    const bool wide = (((uintptr_t)s | (uintptr_t)d) & 15) == 0;
    const int nchunk = wide ? row_bytes >> 4 : 0;
    for (int c = blockIdx.x * 256 + threadIdx.x; c < nchunk; c += gridDim.x * 256) ((uint4*)d)[c] = ((const uint4*)s)[c];
    for (int b = nchunk * 16 + blockIdx.x * 256 + threadIdx.x; b < row_bytes; b += gridDim.x * 256) d[b] = s[b];
}

// n byte segments copied device to device in one launch (one workgroup per segment): the gather of resident keyframe arrays into a stage's batch layout
struct CopySeg { const uint8_t* src; uint8_t* dst; uint32_t bytes, pad; };
__global__ __launch_bounds__(256) void k_copy_segments(const CopySeg* segs) {
    const CopySeg g = segs[blockIdx.x];
    const bool wide = (((uintptr_t)g.src | (uintptr_t)g.dst) & 15) == 0;
    const uint32_t nchunk = wide ? g.bytes >> 4 : 0;
    for (uint32_t c = threadIdx.x; c < nchunk; c += 256) ((uint4*)g.dst)[c] = ((const uint4*)g.src)[c];
    for (uint32_t b = nchunk * 16 + threadIdx.x; b < g.bytes; b += 256) g.dst[b] = g.src[b];
}

// out[i] = the 32-byte descriptor at desc_base[rec[i].x] + 32 * rec[i].y (observations of map points gathered from resident keyframes)
__global__ __launch_bounds__(256) void k_gather_desc(const uint8_t* const* desc_base, const int2* rec, int n, uint8_t* out) {
    const int i = blockIdx.x * 32 + (threadIdx.x >> 3), part = threadIdx.x & 7;
    if (i >= n) return;
    const int2 r = rec[i];
    ((uint32_t*)out)[(size_t)i * 8 + part] = ((const uint32_t*)(desc_base[r.x] + (size_t)r.y * 32))[part];
}

// node of every descriptor: nearest of the 10 top centres, then nearest of that centre's 10 children (Hamming, first minimum); the 110 centres sit in LDS
__global__ __launch_bounds__(256) void k_bow_nodes(const uint8_t* const* desc_ptrs, const int* counts, int stride, const unsigned long long* top,
                                                   const unsigned long long* sub, uint32_t* out) {
    __shared__ unsigned long long s_c[110 * 4];
    for (int i = threadIdx.x; i < 440; i += 256) s_c[i] = i < 40 ? top[i] : sub[i - 40];
    __syncthreads();
    const int b = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    if (k >= counts[b]) return;
    const uint4* d4 = (const uint4*)(desc_ptrs[b] + (size_t)k * 32);
    const uint4 lo = d4[0], hi = d4[1];
    const unsigned long long v[4] = {(unsigned long long)lo.x | ((unsigned long long)lo.y << 32), (unsigned long long)lo.z | ((unsigned long long)lo.w << 32),
                                     (unsigned long long)hi.x | ((unsigned long long)hi.y << 32), (unsigned long long)hi.z | ((unsigned long long)hi.w << 32)};
    auto dist = [&](const unsigned long long* c) { return __popcll(v[0] ^ c[0]) + __popcll(v[1] ^ c[1]) + __popcll(v[2] ^ c[2]) + __popcll(v[3] ^ c[3]); };
    int b1 = 0, bd = 1 << 30;
    for (int i = 0; i < 10; i++) { const int dd = dist(s_c + 4 * i); if (dd < bd) { bd = dd; b1 = i; } }
    int b2 = 0; bd = 1 << 30;
    for (int j = 0; j < 10; j++) { const int dd = dist(s_c + 40 + 4 * (b1 * 10 + j)); if (dd < bd) { bd = dd; b2 = j; } }
    out[(size_t)b * stride + k] = 11u + (uint32_t)(b1 * 10 + b2);
}

}  // namespace oslam

using namespace oslam;

struct oslam_frame {
    int device = 0;
    struct Buf { void* p = nullptr; size_t cap = 0; };
    Buf keys, keysUn, depth, ur, dp, status;
    PinStage pin;
};

static int fr_ensure(oslam_frame::Buf& b, size_t bytes) {
    if (b.p && bytes <= b.cap) return OSLAM_OK;
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = bytes + bytes / 2 + 256;
    OSLAM_HIP_CHECK(hipMalloc(&b.p, b.cap));
    return OSLAM_OK;
}

extern "C" {

void oslam_frame_destroy(oslam_frame_t* h) {
    if (!h) return;
    oslam_frame::Buf* bs[] = {&h->keys, &h->keysUn, &h->depth, &h->ur, &h->dp, &h->status};
    for (auto* b : bs)
        if (b->p) (void)hipFree(b->p);
    h->pin.release();
    delete h;
}

int oslam_frame_create(oslam_frame_t** out, int device) {
    if (!out) { set_error("out is NULL"); return OSLAM_E_INVALID; }
    *out = nullptr;
    int ndev = oslam_device_count();
    if (ndev <= 0) { set_error("no HIP device visible: the gfx950 frame kernels have no CPU fallback"); return OSLAM_E_HIP; }
    if (device < 0 || device >= ndev) { set_error("device out of range"); return OSLAM_E_INVALID; }
    OSLAM_HIP_CHECK(hipSetDevice(device));
    oslam_frame* h = new oslam_frame();
    h->device = device;
    *out = h;
    return OSLAM_OK;
}

int oslam_frame_undistort_batch_device(const oslam_keypoint_t* d_keys, oslam_keypoint_t* d_keysUn, const int32_t* d_counts, int n_const, int stride,
                                       int batch, const float K4[4], const float* dist, int ndist, void* stream) {
    if (!d_keys || !d_keysUn || batch < 1 || stride < 1 || (!d_counts && (n_const < 0 || n_const > stride))) { set_error("bad argument"); return OSLAM_E_INVALID; }
    UndistortCtx c;
    int rc = fill_undistort(c, K4, dist, ndist);
    if (rc) return rc;
    c.keys = d_keys; c.keysUn = d_keysUn; c.counts = d_counts; c.n_const = n_const; c.stride = stride;
    hipLaunchKernelGGL(k_undistort, dim3(div_up(stride, 256), batch), dim3(256), 0, (hipStream_t)stream, c);
    OSLAM_HIP_CHECK(hipGetLastError());
    return OSLAM_OK;
}

int oslam_frame_gather_images_device(const void* const* d_src_ptrs, int n, int src_pitch, int row_bytes, int rows, void* d_dst, size_t dst_image_stride, int dst_pitch, void* stream) {
    if (!d_src_ptrs || !d_dst || n < 1 || rows < 1 || row_bytes < 1 || src_pitch < row_bytes || dst_pitch < row_bytes) { set_error("gather_images: bad argument"); return OSLAM_E_INVALID; }
    hipLaunchKernelGGL(k_gather_images, dim3(1, rows, n), dim3(256), 0, (hipStream_t)stream, (const uint8_t* const*)d_src_ptrs, src_pitch, row_bytes, (uint8_t*)d_dst, dst_image_stride, dst_pitch);
    OSLAM_HIP_CHECK(hipGetLastError());
    return OSLAM_OK;
}

int oslam_copy_segments_device(const void* d_segs, int n, void* stream) {
    if (!d_segs || n < 1) { set_error("copy_segments: bad argument"); return OSLAM_E_INVALID; }
    hipLaunchKernelGGL(k_copy_segments, dim3(n), dim3(256), 0, (hipStream_t)stream, (const CopySeg*)d_segs);
    OSLAM_HIP_CHECK(hipGetLastError());
    return OSLAM_OK;
}

int oslam_bow_nodes_device(const uint8_t* const* d_desc_ptrs, const int32_t* d_counts, int n, int stride, const uint64_t* d_top, const uint64_t* d_sub,
                           uint32_t* d_out, void* stream) {
    if (!d_desc_ptrs || !d_counts || !d_top || !d_sub || !d_out || n < 1 || stride < 1) { set_error("bow_nodes: bad argument"); return OSLAM_E_INVALID; }
    hipLaunchKernelGGL(k_bow_nodes, dim3(div_up(stride, 256), n), dim3(256), 0, (hipStream_t)stream, d_desc_ptrs, d_counts, stride, (const unsigned long long*)d_top,
                       (const unsigned long long*)d_sub, d_out);
    OSLAM_HIP_CHECK(hipGetLastError());
    return OSLAM_OK;
}

int oslam_gather_descriptors_device(const uint8_t* const* d_desc_base, const int32_t* d_rec /*[n][2]*/, int n, uint8_t* d_out, void* stream) {
    if (!d_desc_base || !d_rec || !d_out || n < 1) { set_error("gather_descriptors: bad argument"); return OSLAM_E_INVALID; }
    hipLaunchKernelGGL(k_gather_desc, dim3(div_up(n, 32)), dim3(256), 0, (hipStream_t)stream, d_desc_base, (const int2*)d_rec, n, d_out);
    OSLAM_HIP_CHECK(hipGetLastError());
    return OSLAM_OK;
}

int oslam_frame_stereo_from_rgbd_batch_device(const oslam_keypoint_t* d_keys, const oslam_keypoint_t* d_keysUn, const int32_t* d_counts, int n_const,
                                              int stride, int batch, const float* d_depth, int rows, int cols, int pitch, size_t image_stride,
                                              float mbf, float* d_uRight, float* d_mvDepth, int32_t* d_status, void* stream) {
    if (!d_keys || !d_keysUn || !d_depth || !d_uRight || !d_mvDepth || !d_status || batch < 1 || stride < 1 || rows < 1 || cols < 1 || pitch < cols ||
        (!d_counts && (n_const < 0 || n_const > stride))) {
        set_error("bad argument");
        return OSLAM_E_INVALID;
    }
    RgbdCtx c;
    c.keys = d_keys; c.keysUn = d_keysUn; c.counts = d_counts; c.n_const = n_const; c.stride = stride;
    c.depth = d_depth; c.pitch = pitch; c.image_stride = (long long)image_stride; c.rows = rows; c.cols = cols;
    c.mbf = mbf; c.uRight = d_uRight; c.mvDepth = d_mvDepth; c.status = d_status; c.depth_ptrs = nullptr; c.depth16_ptrs = nullptr; c.factor = 1.f;
    hipLaunchKernelGGL(k_stereo_from_rgbd, dim3(div_up(stride, 256), batch), dim3(256), 0, (hipStream_t)stream, c);
    OSLAM_HIP_CHECK(hipGetLastError());
    return OSLAM_OK;
}

// The same with the depth images where they are: image b = d_depth_ptrs[b] (device table of device pointers), rows `pitch` floats apart — only the ~1000
// depth values at the keypoints are read, so the images need not be gathered into a batch first.
int oslam_frame_stereo_from_rgbd_batch_ptrs_device(const oslam_keypoint_t* d_keys, const oslam_keypoint_t* d_keysUn, const int32_t* d_counts, int n_const, int stride,
                                                   int batch, const float* const* d_depth_ptrs, int rows, int cols, int pitch, float mbf, float* d_uRight,
                                                   float* d_mvDepth, int32_t* d_status, void* stream) {
    if (!d_keys || !d_keysUn || !d_depth_ptrs || !d_uRight || !d_mvDepth || !d_status || batch < 1 || stride < 1 || rows < 1 || cols < 1 || pitch < cols ||
        (!d_counts && (n_const < 0 || n_const > stride))) { set_error("stereo_from_rgbd: bad argument"); return OSLAM_E_INVALID; }
    RgbdCtx c;
    c.keys = d_keys; c.keysUn = d_keysUn; c.counts = d_counts; c.n_const = n_const; c.stride = stride;
    c.depth = nullptr; c.pitch = pitch; c.image_stride = 0; c.rows = rows; c.cols = cols;
    c.mbf = mbf; c.uRight = d_uRight; c.mvDepth = d_mvDepth; c.status = d_status; c.depth_ptrs = d_depth_ptrs; c.depth16_ptrs = nullptr; c.factor = 1.f;
    hipLaunchKernelGGL(k_stereo_from_rgbd, dim3(div_up(stride, 256), batch), dim3(256), 0, (hipStream_t)stream, c);
    OSLAM_HIP_CHECK(hipGetLastError());
    return OSLAM_OK;
}

// The same on RAW 16-bit depth images (the TUM PNGs as Tracking::GrabImageRGBD receives them): image b = d_depth16_ptrs[b], rows `pitch` elements apart;
// a value is scaled on lookup by depth_factor = mDepthMapFactor exactly as imDepth.convertTo(CV_32F, mDepthMapFactor) would have scaled the whole image.
int oslam_frame_stereo_from_rgbd_batch_ptrs_u16_device(const oslam_keypoint_t* d_keys, const oslam_keypoint_t* d_keysUn, const int32_t* d_counts, int n_const, int stride,
                                                       int batch, const uint16_t* const* d_depth16_ptrs, int rows, int cols, int pitch, float depth_factor, float mbf,
                                                       float* d_uRight, float* d_mvDepth, int32_t* d_status, void* stream) {
    if (!d_keys || !d_keysUn || !d_depth16_ptrs || !d_uRight || !d_mvDepth || !d_status || batch < 1 || stride < 1 || rows < 1 || cols < 1 || pitch < cols ||
        (!d_counts && (n_const < 0 || n_const > stride))) { set_error("stereo_from_rgbd (u16): bad argument"); return OSLAM_E_INVALID; }
    RgbdCtx c;
    c.keys = d_keys; c.keysUn = d_keysUn; c.counts = d_counts; c.n_const = n_const; c.stride = stride;
    c.depth = nullptr; c.pitch = pitch; c.image_stride = 0; c.rows = rows; c.cols = cols;
    c.mbf = mbf; c.uRight = d_uRight; c.mvDepth = d_mvDepth; c.status = d_status; c.depth_ptrs = nullptr; c.depth16_ptrs = d_depth16_ptrs; c.factor = depth_factor;
    hipLaunchKernelGGL(k_stereo_from_rgbd, dim3(div_up(stride, 256), batch), dim3(256), 0, (hipStream_t)stream, c);
    OSLAM_HIP_CHECK(hipGetLastError());
    return OSLAM_OK;
}

int oslam_frame_undistort_keypoints(oslam_frame_t* h, int n, const oslam_keypoint_t* keys, const float K4[4], const float* dist, int ndist,
                                    oslam_keypoint_t* keysUn) {
    if (!h || n < 0 || (n > 0 && (!keys || !keysUn))) { set_error("bad argument"); return OSLAM_E_INVALID; }
    if (n == 0) return OSLAM_OK;
    OSLAM_HIP_CHECK(hipSetDevice(h->device));
    const size_t bytes = (size_t)n * sizeof(oslam_keypoint_t);
    int rc;
    if ((rc = fr_ensure(h->keys, bytes)) || (rc = fr_ensure(h->keysUn, bytes)) || (rc = h->pin.reserve_total(2 * bytes + 4096))) return rc;
    h->pin.reset();
    if ((rc = h->pin.upload(h->keys.p, keys, bytes))) return rc;
    if ((rc = oslam_frame_undistort_batch_device((const oslam_keypoint_t*)h->keys.p, (oslam_keypoint_t*)h->keysUn.p, nullptr, n, n, 1, K4, dist, ndist, nullptr))) return rc;
    uint8_t* at = nullptr;
    if ((rc = h->pin.download(h->keysUn.p, bytes, &at))) return rc;
    OSLAM_HIP_CHECK(hipStreamSynchronize(nullptr));
    memcpy(keysUn, at, bytes);
    return OSLAM_OK;
}

int oslam_frame_image_bounds(oslam_frame_t* h, int cols, int rows, const float K4[4], const float* dist, int ndist, float bounds[4]) {
    if (!h || !bounds || cols < 1 || rows < 1 || !K4) { set_error("bad argument"); return OSLAM_E_INVALID; }
    if (ndist == 0 || (dist && dist[0] == 0.0f)) { bounds[0] = 0.f; bounds[1] = 0.f; bounds[2] = (float)cols; bounds[3] = (float)rows; return OSLAM_OK; }
    oslam_keypoint_t c[4], u[4];
    memset(c, 0, sizeof(c));
    c[1].x = (float)cols; c[2].y = (float)rows; c[3].x = (float)cols; c[3].y = (float)rows;
    int rc = oslam_frame_undistort_keypoints(h, 4, c, K4, dist, ndist, u);
    if (rc) return rc;
    bounds[0] = fminf(u[0].x, u[2].x);   // mnMinX
    bounds[2] = fmaxf(u[1].x, u[3].x);   // mnMaxX
    bounds[1] = fminf(u[0].y, u[1].y);   // mnMinY
    bounds[3] = fmaxf(u[2].y, u[3].y);   // mnMaxY
    return OSLAM_OK;
}

int oslam_frame_stereo_from_rgbd(oslam_frame_t* h, int n, const oslam_keypoint_t* keys, const oslam_keypoint_t* keysUn, const float* depth, int rows,
                                 int cols, int pitch, float mbf, float* uRight, float* mvDepth) {
    if (!h || n < 0 || !depth || rows < 1 || cols < 1 || pitch < cols || (n > 0 && (!keys || !keysUn || !uRight || !mvDepth))) { set_error("bad argument"); return OSLAM_E_INVALID; }
    if (n == 0) return OSLAM_OK;
    OSLAM_HIP_CHECK(hipSetDevice(h->device));
    const size_t kb = (size_t)n * sizeof(oslam_keypoint_t), db = (size_t)rows * pitch * 4;
    int rc;
    if ((rc = fr_ensure(h->keys, kb)) || (rc = fr_ensure(h->keysUn, kb)) || (rc = fr_ensure(h->depth, db)) || (rc = fr_ensure(h->ur, (size_t)n * 4)) ||
        (rc = fr_ensure(h->dp, (size_t)n * 4)) || (rc = fr_ensure(h->status, 4)) || (rc = h->pin.reserve_total(2 * kb + db + (size_t)n * 8 + 8192)))
        return rc;
    h->pin.reset();
    OSLAM_HIP_CHECK(hipMemsetAsync(h->status.p, 0, 4, nullptr));
    if ((rc = h->pin.upload(h->keys.p, keys, kb)) || (rc = h->pin.upload(h->keysUn.p, keysUn, kb)) || (rc = h->pin.upload(h->depth.p, depth, db))) return rc;
    if ((rc = oslam_frame_stereo_from_rgbd_batch_device((const oslam_keypoint_t*)h->keys.p, (const oslam_keypoint_t*)h->keysUn.p, nullptr, n, n, 1,
                                                        (const float*)h->depth.p, rows, cols, pitch, 0, mbf, (float*)h->ur.p, (float*)h->dp.p,
                                                        (int32_t*)h->status.p, nullptr)))
        return rc;
    uint8_t *a_u = nullptr, *a_d = nullptr, *a_s = nullptr;
    if ((rc = h->pin.download(h->ur.p, (size_t)n * 4, &a_u)) || (rc = h->pin.download(h->dp.p, (size_t)n * 4, &a_d)) || (rc = h->pin.download(h->status.p, 4, &a_s))) return rc;
    OSLAM_HIP_CHECK(hipStreamSynchronize(nullptr));
    int st;
    memcpy(&st, a_s, 4);
    if (st) { set_error("a keypoint lies outside the depth image"); return OSLAM_E_INVALID; }
    memcpy(uRight, a_u, (size_t)n * 4);
    memcpy(mvDepth, a_d, (size_t)n * 4);
    return OSLAM_OK;
}

}  // extern "C"
