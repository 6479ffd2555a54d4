/*
 * oslam_hip.h — C ABI of the MI355X-native (gfx950) front-end + local-BA hot path of
 * yangliu9527/Object_SLAM (an ORB_SLAM2 fork).  Plain pointers and sizes only; no C++ / torch
 * types.  Every entry point returns 0 on success or a negative OSLAM_E_* code;
 * oslam_last_error() gives a thread-local message.  One handle per caller thread / HIP stream;
 * handles are not re-entrant (same contract as the reference classes, which own their buffers).
 *
 * Each declaration cites the reference interface (file:line under the reference tree) it
 * replaces.  INTEGRATION.md shows the adapter a maintainer adds on the reference side.
 */
#ifndef OSLAM_HIP_H
#define OSLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OSLAM_OK 0
#define OSLAM_E_INVALID (-1)   /* bad argument / unsupported shape */
#define OSLAM_E_HIP (-2)       /* HIP runtime error (no device, OOM, launch failure) */
#define OSLAM_E_CAPACITY (-3)  /* caller buffer or internal arena too small (never truncates silently) */
#define OSLAM_E_NUMERIC (-4)   /* solver failure */

#define OSLAM_MAX_LEVELS 16

/* cv::KeyPoint POD mirror (OpenCV 3.2 core/types.hpp; used at include/ORBextractor.h:59-61). */
typedef struct oslam_keypoint {
    float x, y;       /* pt */
    float size;
    float angle;      /* degrees [0,360) */
    float response;   /* FAST score */
    int32_t octave;
    int32_t class_id; /* -1 */
} oslam_keypoint_t;

const char* oslam_last_error(void);
/* Number of visible HIP devices (0 if none); never throws. */
int oslam_device_count(void);

/* ------------------------------------------------------------------------------------------
 * ORBextractor — replaces ORB_SLAM2::ORBextractor (include/ORBextractor.h:45-110,
 * src/ORBextractor.cc:410-470 ctor, :1043-1105 operator()).
 * The handle is created for one image geometry and a maximum batch of images per call;
 * batch > 1 is the batch-of-sequences mode (independent images, same arithmetic per image).
 * ---------------------------------------------------------------------------------------- */
typedef struct oslam_orb oslam_orb_t;

int oslam_orb_create(oslam_orb_t** out, int nfeatures, float scaleFactor, int nlevels, int iniThFAST,
                     int minThFAST, int width, int height, int max_batch, int device);
void oslam_orb_destroy(oslam_orb_t* h);

/* Getters of include/ORBextractor.h:63-83; each array has nlevels entries (any may be NULL). */
int oslam_orb_get_scale_tables(const oslam_orb_t* h, float* scaleFactors, float* invScaleFactors,
                               float* levelSigma2, float* invLevelSigma2, int* nFeaturesPerLevel);
/* cv::GaussianBlur column-pass rounding model (DESIGN.md "blur rounding"): 1 (default) = OpenCV 3.2
 * on x86 SSE2 (half-to-even for x < (w & ~3), half-up tail); 0 = scalar fixed point everywhere. */
int oslam_orb_set_blur_rounding(oslam_orb_t* h, int sse2);
/* Per image capacity of the keypoint / descriptor outputs (sum of per-level node caps). */
int oslam_orb_max_keypoints(const oslam_orb_t* h);

/* operator() drop-in (src/ORBextractor.cc:1043): host gray image in, host keypoints +
 * descriptors (n x 32 bytes, row-major) out.  Empty image -> *n_out = 0, returns 0 (the
 * reference returns silently, :1046).  cap < n -> OSLAM_E_CAPACITY with *n_out = n. */
int oslam_orb_extract(oslam_orb_t* h, const uint8_t* gray, int width, int height, int stride,
                      oslam_keypoint_t* keypoints, uint8_t* descriptors, int cap, int* n_out);

/* Batch mode, everything resident in HBM: d_gray holds `batch` images (row pitch `stride`
 * bytes, `image_stride` bytes between images).  Asynchronous on `stream` (a hipStream_t, may be
 * NULL = default stream); results stay on the device, see oslam_orb_results_device(). */
int oslam_orb_extract_batch_device(oslam_orb_t* h, const uint8_t* d_gray, int batch, int stride,
                                   size_t image_stride, void* stream);
/* Device pointers of the last batch: keypoints [batch][cap] (oslam_keypoint_t), descriptors
 * [batch][cap][32], counts [batch] (int32), cap = oslam_orb_max_keypoints().  status[0] != 0
 * after the stream has drained means an internal arena overflowed (OSLAM_E_CAPACITY). */
int oslam_orb_results_device(const oslam_orb_t* h, const oslam_keypoint_t** d_keypoints,
                             const uint8_t** d_descriptors, const int32_t** d_counts,
                             const int32_t** d_status);
/* Synchronise the stream and copy image b's results of the last batch to the host. */
int oslam_orb_fetch(oslam_orb_t* h, int b, oslam_keypoint_t* keypoints, uint8_t* descriptors, int cap,
                    int* n_out);

/* mvImagePyramid (public member, include/ORBextractor.h:85; read by Frame::ComputeStereoMatches,
 * src/Frame.cc:713,803,815): level image b of the last batch, host copy (w*h bytes, tight) or
 * device view. */
int oslam_orb_level_size(const oslam_orb_t* h, int level, int* width, int* height);
int oslam_orb_get_pyramid_level(oslam_orb_t* h, int b, int level, uint8_t* out);
int oslam_orb_pyramid_level_device(const oslam_orb_t* h, int b, int level, const uint8_t** d_ptr,
                                   int* pitch);

/* Stage outputs for parity tests (host copies; call after a batch has run). */
int oslam_orb_debug_get_blurred(oslam_orb_t* h, int b, int level, uint8_t* out);
/* FAST candidates of a level in reference order (x, y in region coords, response):
 * src/ORBextractor.cc:789-829.  out is [cap][3] int32; returns count in *n_out. */
int oslam_orb_debug_get_candidates(oslam_orb_t* h, int b, int level, int32_t* out, int cap, int* n_out);
/* Quad-tree survivors of a level in reference order (level coords): src/ORBextractor.cc:834-847. */
int oslam_orb_debug_get_level_keys(oslam_orb_t* h, int b, int level, int32_t* out, int cap, int* n_out);

/* Work model of one extract call: algorithmic bytes per image. */
int64_t oslam_orb_algorithmic_bytes(const oslam_orb_t* h, int n_keypoints);

#ifdef __cplusplus
}
#endif
#endif /* OSLAM_HIP_H */
