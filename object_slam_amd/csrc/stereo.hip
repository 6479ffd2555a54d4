// gfx950 stereo association: Frame::ComputeStereoMatches (reference src/Frame.cc:706-880) —
// row-band Hamming search left->right, 11-shift 11x11 SAD refinement on the pyramid level of the
// left keypoint, parabola sub-pixel fit, depth = bf/disparity, median-SAD outlier cut.
// One 1024-thread workgroup per stereo pair; right keypoints + descriptors staged in LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace oslam {

constexpr int kStereoThreads = 1024;
constexpr int kStereoMaxKps = 2400;

struct StereoPyr {
    const uint8_t* lv[OSLAM_MAX_LEVELS];
    int pitch[OSLAM_MAX_LEVELS], w[OSLAM_MAX_LEVELS], h[OSLAM_MAX_LEVELS];
};

struct StereoCtx {
    const oslam_keypoint_t* kpL; const uint8_t* descL; const int* nL; int nL_const;
    const oslam_keypoint_t* kpR; const uint8_t* descR; const int* nR; int nR_const;
    int kp_stride;
    long long pyr_stride_L, pyr_stride_R;     // bytes between images of consecutive batch entries (levels >= 1)
    long long img0_stride_L, img0_stride_R;   // level 0 lives in the caller's image buffer
    StereoPyr L, R;
    float scale[OSLAM_MAX_LEVELS], invScale[OSLAM_MAX_LEVELS];
    float bf, b;
    float* uRight; float* depth;   // [B][kp_stride]
    int* sad;                      // [B][kp_stride] scratch: best SAD of accepted matches or -1
    int* n_matched;                // [B]
};

__global__ __launch_bounds__(kStereoThreads) void k_stereo(StereoCtx c, int ncap) {
    const int bidx = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int N = c.nL ? c.nL[bidx] : c.nL_const;
    const int Nr = c.nR ? c.nR[bidx] : c.nR_const;
    const oslam_keypoint_t* kpL = c.kpL + (long long)bidx * c.kp_stride;
    const oslam_keypoint_t* kpR = c.kpR + (long long)bidx * c.kp_stride;
    const uint32_t* dL = (const uint32_t*)(c.descL + (long long)bidx * c.kp_stride * 32);
    const uint32_t* dR = (const uint32_t*)(c.descR + (long long)bidx * c.kp_stride * 32);
    float* uRight = c.uRight + (long long)bidx * c.kp_stride;
    float* depth = c.depth + (long long)bidx * c.kp_stride;
    int* sad = c.sad + (long long)bidx * c.kp_stride;

    extern __shared__ __align__(16) uint8_t smem[];
    uint32_t* s_desc = (uint32_t*)smem;                 // [ncap][8] right descriptors
    float* s_x = (float*)(s_desc + (size_t)ncap * 8);   // [ncap] right u
    short* s_minr = (short*)(s_x + ncap);               // [ncap]
    short* s_maxr = s_minr + ncap;                      // [ncap]
    short* s_best = s_maxr + ncap;                      // [ncap] per LEFT keypoint: best right index or -1
    uint8_t* s_oct = (uint8_t*)(s_best + ncap);         // [ncap]
    __shared__ int s_cnt, s_median;

    if (N > ncap || Nr > ncap || N < 0 || Nr < 0) {
        if (tid == 0) c.n_matched[bidx] = -1;
        return;
    }
    const int nRows = c.L.h[0];
    // ---- row table (:716-733) kept implicit: right keypoint iR is a candidate of row y iff minr <= y <= maxr ----
    for (int i = tid; i < Nr; i += kStereoThreads) {
        const oslam_keypoint_t kp = kpR[i];
        const float r = 2.0f * c.scale[kp.octave];
        s_maxr[i] = (short)(int)ceilf(kp.y + r);
        s_minr[i] = (short)(int)floorf(kp.y - r);
        s_x[i] = kp.x;
        s_oct[i] = (uint8_t)kp.octave;
    }
    for (int i = tid; i < Nr * 8; i += kStereoThreads) s_desc[i] = dR[i];
    if (tid == 0) s_cnt = 0;
    __syncthreads();

    const float minZ = c.b, minD = 0.f, maxD = c.bf / minZ;
    const int thOrbDist = (100 + 50) / 2;
    // ---- Hamming scan (:744-789): candidates of the row in right-keypoint index order ----
    for (int iL = tid; iL < N; iL += kStereoThreads) {
        const oslam_keypoint_t kp = kpL[iL];
        uRight[iL] = -1.0f;
        depth[iL] = -1.0f;
        sad[iL] = -1;
        int best = -1;
        const int row = (int)kp.y;
        const float minU = kp.x - maxD, maxU = kp.x - minD;
        if (row >= 0 && row < nRows && !(maxU < 0)) {
            uint32_t q[8];
#pragma unroll
            for (int w = 0; w < 8; w++) q[w] = dL[(size_t)iL * 8 + w];
            int bestDist = 100;   // TH_HIGH
            const int levelL = kp.octave;
            for (int iR = 0; iR < Nr; iR++) {
                if (row < s_minr[iR] || row > s_maxr[iR]) continue;
                const int o = s_oct[iR];
                if (o < levelL - 1 || o > levelL + 1) continue;
                const float uR = s_x[iR];
                if (uR >= minU && uR <= maxU) {
                    const uint32_t* d = s_desc + iR * 8;
                    int dist = 0;
#pragma unroll
                    for (int w = 0; w < 8; w++) dist += __popc(q[w] ^ d[w]);
                    if (dist < bestDist) { bestDist = dist; best = iR; }
                }
            }
            if (!(bestDist < thOrbDist)) best = -1;
        }
        s_best[iL] = (short)best;
    }
    __syncthreads();

    // ---- SAD refinement (:792-863), one wavefront per left keypoint ----
    for (int iL = wv; iL < N; iL += kStereoThreads / 64) {
        const int bestIdxR = s_best[iL];
        if (bestIdxR < 0) continue;
        const oslam_keypoint_t kp = kpL[iL];
        const int oct = kp.octave;
        const float uR0 = s_x[bestIdxR];
        const float sf = c.invScale[oct];
        const float scaleduL = roundf(kp.x * sf), scaledvL = roundf(kp.y * sf), scaleduR0 = roundf(uR0 * sf);
        const int w = 5, L = 5;
        const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
        if (iniu < 0 || endu >= (float)c.R.w[oct]) continue;
        const uint8_t* imL = c.L.lv[oct] + (oct == 0 ? c.img0_stride_L : c.pyr_stride_L) * bidx;
        const uint8_t* imR = c.R.lv[oct] + (oct == 0 ? c.img0_stride_R : c.pyr_stride_R) * bidx;
        const int pL = c.L.pitch[oct], pR = c.R.pitch[oct];
        const int cy = (int)scaledvL, cxL = (int)scaleduL, cxR0 = (int)scaleduR0;
        // guard the window reads (the reference does not check the left window / the right -10 side)
        if (cy - w < 0 || cy + w >= c.L.h[oct] || cxL - w < 0 || cxL + w >= c.L.w[oct] || cxR0 - L - w < 0 || cxR0 + L + w >= c.R.w[oct]) continue;
        const int centerL = imL[(long long)cy * pL + cxL];
        // lanes hold the 121 left-patch values (2 per lane)
        int lv0 = 0, lv1 = 0, dy0 = 0, dx0 = 0, dy1 = 0, dx1 = 0;
        const bool has1 = lane + 64 < 121;
        dy0 = lane / 11 - w; dx0 = lane % 11 - w;
        lv0 = (int)imL[(long long)(cy + dy0) * pL + cxL + dx0] - centerL;
        if (has1) {
            dy1 = (lane + 64) / 11 - w; dx1 = (lane + 64) % 11 - w;
            lv1 = (int)imL[(long long)(cy + dy1) * pL + cxL + dx1] - centerL;
        }
        int bestDist = 0x7fffffff, bestinc = 0;
        float d_m1 = 0, d_0 = 0, d_p1 = 0;
        float vd[11];
#pragma unroll
        for (int inc = -5; inc <= 5; inc++) {
            const int cxR = cxR0 + inc;
            const int centerR = imR[(long long)cy * pR + cxR];
            int s = abs(lv0 - ((int)imR[(long long)(cy + dy0) * pR + cxR + dx0] - centerR));
            if (has1) s += abs(lv1 - ((int)imR[(long long)(cy + dy1) * pR + cxR + dx1] - centerR));
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
            vd[inc + 5] = (float)s;
            if ((float)s < (float)bestDist) { bestDist = s; bestinc = inc; }
        }
        if (bestinc == -L || bestinc == L) continue;
#pragma unroll
        for (int k = 1; k < 10; k++)
            if (k == bestinc + 5) { d_m1 = vd[k - 1]; d_0 = vd[k]; d_p1 = vd[k + 1]; }
        const float deltaR = __fdiv_rn(d_m1 - d_p1, 2.0f * (d_m1 + d_p1 - 2.0f * d_0));
        if (deltaR < -1 || deltaR > 1) continue;   // NaN (flat SAD) passes, exactly like the reference's comparison
        float bestuR = c.scale[oct] * ((float)scaleduR0 + (float)bestinc + deltaR);
        float disparity = kp.x - bestuR;
        if (disparity >= minD && disparity < maxD) {
            if (disparity <= 0) { disparity = 0.01f; bestuR = (float)((double)kp.x - 0.01); }
            if (lane == 0) {
                depth[iL] = __fdiv_rn(c.bf, disparity);
                uRight[iL] = bestuR;
                sad[iL] = bestDist;
                atomicAdd(&s_cnt, 1);
            }
        }
    }
    __syncthreads();
    // ---- median-SAD outlier cut (:866-879): median = element size/2 of the (dist, iL)-sorted list ----
    const int M = s_cnt;
    if (tid == 0) s_median = -1;
    __syncthreads();
    if (M > 0) {
        for (int i = tid; i < N; i += kStereoThreads) {
            const int di = sad[i];
            if (di < 0) continue;
            int rank = 0;
            for (int j = 0; j < N; j++) {
                const int dj = sad[j];
                if (dj >= 0 && (dj < di || (dj == di && j < i))) rank++;
            }
            if (rank == M / 2) s_median = di;
        }
        __syncthreads();
        const float median = (float)s_median;
        const float thDist = 1.5f * 1.4f * median;
        for (int i = tid; i < N; i += kStereoThreads) {
            const int di = sad[i];
            if (di >= 0 && !((float)di < thDist)) { uRight[i] = -1.0f; depth[i] = -1.0f; }
        }
    }
    if (tid == 0) c.n_matched[bidx] = M;
}

}  // namespace oslam

using namespace oslam;

struct oslam_stereo {
    int device = 0, max_batch = 0, max_kps = 0;
    size_t lds = 0;
    float* d_uRight = nullptr; float* d_depth = nullptr; int* d_sad = nullptr; int* d_nm = nullptr;
    oslam_keypoint_t* d_kpL = nullptr; oslam_keypoint_t* d_kpR = nullptr; uint8_t* d_descL = nullptr; uint8_t* d_descR = nullptr;
};

extern "C" {

void oslam_stereo_destroy(oslam_stereo_t* h) {
    if (!h) return;
    void* ptrs[] = {h->d_uRight, h->d_depth, h->d_sad, h->d_nm, h->d_kpL, h->d_kpR, h->d_descL, h->d_descR};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    delete h;
}

int oslam_stereo_create(oslam_stereo_t** out, int max_batch, int max_keypoints, int device) {
    if (!out) { set_error("out is NULL"); return OSLAM_E_INVALID; }
    *out = nullptr;
    if (max_batch < 1 || max_keypoints < 1 || max_keypoints > kStereoMaxKps) { set_error("oslam_stereo_create: invalid argument (max_keypoints <= %d)", kStereoMaxKps); return OSLAM_E_INVALID; }
    int ndev = oslam_device_count();
    if (ndev <= 0) { set_error("no HIP device visible: the gfx950 stereo matcher has no CPU fallback"); return OSLAM_E_HIP; }
    if (device < 0 || device >= ndev) { set_error("device out of range"); return OSLAM_E_INVALID; }
    OSLAM_HIP_CHECK(hipSetDevice(device));
    oslam_stereo* h = new oslam_stereo();
    h->device = device; h->max_batch = max_batch; h->max_kps = max_keypoints;
    h->lds = (size_t)max_keypoints * (32 + 4 + 2 + 2 + 2 + 1) + 64;
    const size_t B = max_batch, NK = max_keypoints;
#define ALLOC(ptr, bytes)                                                         \
    do {                                                                          \
        hipError_t e_ = hipMalloc((void**)&(ptr), (bytes));                       \
        if (e_ != hipSuccess) {                                                   \
            set_error("hipMalloc(%zu) failed: %s", (size_t)(bytes), hipGetErrorString(e_)); \
            oslam_stereo_destroy(h);                                              \
            return OSLAM_E_HIP;                                                   \
        }                                                                         \
    } while (0)
    ALLOC(h->d_uRight, B * NK * 4); ALLOC(h->d_depth, B * NK * 4); ALLOC(h->d_sad, B * NK * 4); ALLOC(h->d_nm, B * 4);
    ALLOC(h->d_kpL, NK * sizeof(oslam_keypoint_t)); ALLOC(h->d_kpR, NK * sizeof(oslam_keypoint_t)); ALLOC(h->d_descL, NK * 32); ALLOC(h->d_descR, NK * 32);
#undef ALLOC
    OSLAM_HIP_CHECK(hipFuncSetAttribute((const void*)k_stereo, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds));
    *out = h;
    return OSLAM_OK;
}

static int fill_pyr(oslam_orb_t* orb, int nlevels, StereoPyr* P, long long* pyr_stride, long long* img0_stride) {
    // strides: distance between batch entries = (pointer of b=1) - (pointer of b=0) when the last batch had > 1 image
    for (int l = 0; l < nlevels; l++) {
        int rc = oslam_orb_pyramid_level_device(orb, 0, l, &P->lv[l], &P->pitch[l]);
        if (rc) return rc;
        rc = oslam_orb_level_size(orb, l, &P->w[l], &P->h[l]);
        if (rc) return rc;
    }
    const uint8_t* p1; int pitch;
    *pyr_stride = 0; *img0_stride = 0;
    if (oslam_orb_pyramid_level_device(orb, 1, 0, &p1, &pitch) == OSLAM_OK) {
        *img0_stride = p1 - P->lv[0];
        if (nlevels > 1 && oslam_orb_pyramid_level_device(orb, 1, 1, &p1, &pitch) == OSLAM_OK) *pyr_stride = p1 - P->lv[1];
    }
    return OSLAM_OK;
}

int oslam_stereo_match_batch_device(oslam_stereo_t* h, oslam_orb_t* orbL, oslam_orb_t* orbR, int batch, int kp_stride,
                                    const oslam_keypoint_t* d_kpL, const uint8_t* d_descL, const int32_t* d_nL, int nL_const,
                                    const oslam_keypoint_t* d_kpR, const uint8_t* d_descR, const int32_t* d_nR, int nR_const,
                                    int nlevels, float bf, float b, void* stream) {
    if (!h || !orbL || !orbR || !d_kpL || !d_descL || !d_kpR || !d_descR) { set_error("NULL argument"); return OSLAM_E_INVALID; }
    if (batch < 1 || batch > h->max_batch || nlevels < 1 || nlevels > OSLAM_MAX_LEVELS) { set_error("bad batch/nlevels"); return OSLAM_E_INVALID; }
    if (kp_stride < 1 || kp_stride > h->max_kps) { set_error("kp_stride %d > max_keypoints %d", kp_stride, h->max_kps); return OSLAM_E_CAPACITY; }
    if ((!d_nL && (nL_const < 0 || nL_const > kp_stride)) || (!d_nR && (nR_const < 0 || nR_const > kp_stride))) { set_error("keypoint count exceeds stride"); return OSLAM_E_CAPACITY; }
    if (!(b > 0) || !(bf > 0)) { set_error("bf and b must be positive"); return OSLAM_E_INVALID; }
    OSLAM_HIP_CHECK(hipSetDevice(h->device));
    StereoCtx c;
    memset(&c, 0, sizeof(c));
    int rc = fill_pyr(orbL, nlevels, &c.L, &c.pyr_stride_L, &c.img0_stride_L);
    if (rc) return rc;
    rc = fill_pyr(orbR, nlevels, &c.R, &c.pyr_stride_R, &c.img0_stride_R);
    if (rc) return rc;
    float s2[OSLAM_MAX_LEVELS], is2[OSLAM_MAX_LEVELS];
    rc = oslam_orb_get_scale_tables(orbL, c.scale, c.invScale, s2, is2, nullptr);
    if (rc) return rc;
    c.kpL = d_kpL; c.descL = d_descL; c.nL = d_nL; c.nL_const = nL_const;
    c.kpR = d_kpR; c.descR = d_descR; c.nR = d_nR; c.nR_const = nR_const;
    c.kp_stride = kp_stride; c.bf = bf; c.b = b;
    c.uRight = h->d_uRight; c.depth = h->d_depth; c.sad = h->d_sad; c.n_matched = h->d_nm;
    hipLaunchKernelGGL(k_stereo, dim3(batch), dim3(kStereoThreads), h->lds, (hipStream_t)stream, c, h->max_kps);
    OSLAM_HIP_CHECK(hipGetLastError());
    return OSLAM_OK;
}

int oslam_stereo_results_device(const oslam_stereo_t* h, const float** d_uRight, const float** d_depth, const int32_t** d_n_matched) {
    if (!h) { set_error("NULL handle"); return OSLAM_E_INVALID; }
    if (d_uRight) *d_uRight = h->d_uRight;
    if (d_depth) *d_depth = h->d_depth;
    if (d_n_matched) *d_n_matched = h->d_nm;
    return OSLAM_OK;
}

int oslam_stereo_match(oslam_stereo_t* h, oslam_orb_t* orbL, oslam_orb_t* orbR, int N, const oslam_keypoint_t* keysL, const uint8_t* descL,
                       int Nr, const oslam_keypoint_t* keysR, const uint8_t* descR, int nlevels, float bf, float b, float* uRight,
                       float* depth) {
    if (!h || (N > 0 && (!keysL || !descL || !uRight || !depth)) || (Nr > 0 && (!keysR || !descR))) { set_error("NULL argument"); return OSLAM_E_INVALID; }
    if (N < 0 || Nr < 0 || N > h->max_kps || Nr > h->max_kps) { set_error("keypoint count exceeds capacity %d", h->max_kps); return OSLAM_E_CAPACITY; }
    OSLAM_HIP_CHECK(hipSetDevice(h->device));
    if (N > 0) {
        OSLAM_HIP_CHECK(hipMemcpy(h->d_kpL, keysL, (size_t)N * sizeof(oslam_keypoint_t), hipMemcpyHostToDevice));
        OSLAM_HIP_CHECK(hipMemcpy(h->d_descL, descL, (size_t)N * 32, hipMemcpyHostToDevice));
    }
    if (Nr > 0) {
        OSLAM_HIP_CHECK(hipMemcpy(h->d_kpR, keysR, (size_t)Nr * sizeof(oslam_keypoint_t), hipMemcpyHostToDevice));
        OSLAM_HIP_CHECK(hipMemcpy(h->d_descR, descR, (size_t)Nr * 32, hipMemcpyHostToDevice));
    }
    int rc = oslam_stereo_match_batch_device(h, orbL, orbR, 1, h->max_kps, h->d_kpL, h->d_descL, nullptr, N, h->d_kpR, h->d_descR, nullptr, Nr,
                                             nlevels, bf, b, nullptr);
    if (rc) return rc;
    OSLAM_HIP_CHECK(hipDeviceSynchronize());
    int nm = 0;
    OSLAM_HIP_CHECK(hipMemcpy(&nm, h->d_nm, 4, hipMemcpyDeviceToHost));
    if (nm < 0) { set_error("stereo kernel rejected the frame (capacity)"); return OSLAM_E_CAPACITY; }
    if (N > 0) {
        OSLAM_HIP_CHECK(hipMemcpy(uRight, h->d_uRight, (size_t)N * 4, hipMemcpyDeviceToHost));
        OSLAM_HIP_CHECK(hipMemcpy(depth, h->d_depth, (size_t)N * 4, hipMemcpyDeviceToHost));
    }
    return OSLAM_OK;
}

}  // extern "C"
