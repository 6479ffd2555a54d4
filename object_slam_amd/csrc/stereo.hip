// gfx950 stereo association: Frame::ComputeStereoMatches (reference src/Frame.cc:706-880) —
// row-band Hamming search left->right, 11-shift 11x11 SAD refinement on the pyramid level of the
// left keypoint, parabola sub-pixel fit, depth = bf/disparity, median-SAD outlier cut.
// Scan and outlier cut: one 1024-thread workgroup per stereo pair (right keypoints + descriptors staged in LDS); SAD refinement: one wavefront per left keypoint over the whole GPU.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "lane_ops.h"

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
    short* best;                   // [B][kp_stride] scratch: best right index of the Hamming scan or -1
    unsigned short* row_items;     // [B][ncap * kRowItemsPerKp] scratch: row table items
};

// ------------------------------------------------------------------------------------------
// Three kernels.  k_stereo_scan (one workgroup per pair): right keypoints + descriptors into LDS, the reference's row table
// (vRowIndices, :716-733) as a counting sort over the image rows (items in HBM scratch), then one thread per left keypoint scans
// the candidates of its row.  The reference keeps the FIRST minimum in right-keypoint order; the lists here are unordered, so the
// scan takes the minimum of (distance, right index), which is the same keypoint.  k_stereo_sad (one wavefront per left keypoint,
// the whole GPU): 11-shift SAD + parabola.  k_stereo_cut (one workgroup per pair): median SAD by bisection, outlier cut.
// ------------------------------------------------------------------------------------------
constexpr int kStereoMaxRows = 4096;      // level-0 height limit of the extractor
constexpr int kRowItemsPerKp = 24;        // capacity of the row table: items per right keypoint (band of 2 * 2 * scale + 1 rows; 17 at 1.2^7)

__device__ __forceinline__ bool stereo_counts(const StereoCtx& c, int bidx, int ncap, int& N, int& Nr) {
    N = c.nL ? c.nL[bidx] : c.nL_const;
    Nr = c.nR ? c.nR[bidx] : c.nR_const;
    return !(N > ncap || Nr > ncap || N < 0 || Nr < 0);
}

__global__ __launch_bounds__(kStereoThreads) void k_stereo_scan(StereoCtx c, int ncap) {
    const int bidx = blockIdx.x, tid = threadIdx.x;
    int N, Nr;
    const bool ok = stereo_counts(c, bidx, ncap, N, Nr);
    const oslam_keypoint_t* kpL = c.kpL + (long long)bidx * c.kp_stride;
    const oslam_keypoint_t* kpR = c.kpR + (long long)bidx * c.kp_stride;
    const uint32_t* dL = (const uint32_t*)(c.descL + (long long)bidx * c.kp_stride * 32);
    const uint32_t* dR = (const uint32_t*)(c.descR + (long long)bidx * c.kp_stride * 32);
    float* uRight = c.uRight + (long long)bidx * c.kp_stride;
    float* depth = c.depth + (long long)bidx * c.kp_stride;
    int* sad = c.sad + (long long)bidx * c.kp_stride;
    short* best_out = c.best + (long long)bidx * c.kp_stride;
    unsigned short* items = c.row_items + (long long)bidx * ncap * kRowItemsPerKp;

    extern __shared__ __align__(16) uint8_t smem[];
    uint32_t* s_desc = (uint32_t*)smem;                 // [ncap][8] right descriptors
    float* s_x = (float*)(s_desc + (size_t)ncap * 8);   // [ncap] right u
    int* s_row = (int*)(s_x + ncap);                    // [kStereoMaxRows + 1] row starts (counting sort)
    short* s_minr = (short*)(s_row + kStereoMaxRows + 1);   // [ncap]
    short* s_maxr = s_minr + ncap;                      // [ncap]
    uint8_t* s_oct = (uint8_t*)(s_maxr + ncap);         // [ncap]
    __shared__ int s_total;
    __shared__ int s_wtot[kStereoThreads / 64];

    if (tid == 0) c.n_matched[bidx] = ok ? 0 : -1;
    if (!ok) return;
    const int nRows = min(c.L.h[0], kStereoMaxRows);
    for (int i = tid; i <= nRows; i += kStereoThreads) s_row[i] = 0;
    __syncthreads();
    for (int i = tid; i < Nr; i += kStereoThreads) {
        const oslam_keypoint_t kp = kpR[i];
        const float r = 2.0f * c.scale[kp.octave];
        const int maxr = (int)ceilf(kp.y + r), minr = (int)floorf(kp.y - r);
        s_maxr[i] = (short)maxr;
        s_minr[i] = (short)minr;
        s_x[i] = kp.x;
        s_oct[i] = (uint8_t)kp.octave;
        for (int y = max(minr, 0); y <= min(maxr, nRows - 1); y++) atomicAdd(&s_row[y], 1);
    }
    for (int i = tid; i < Nr * 8; i += kStereoThreads) s_desc[i] = dR[i];
    __syncthreads();
    // exclusive scan of the row counts (4 rows per thread)
    {
        const int lane = tid & 63, wv = tid >> 6;
        int cnt[4], local = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { const int y = tid * 4 + k; cnt[k] = y < nRows ? s_row[y] : 0; local += cnt[k]; }
        int incl = local;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t;
        }
        if (lane == 63) s_wtot[wv] = incl;
        __syncthreads();
        int start = incl - local;
        for (int i = 0; i < wv; i++) start += s_wtot[i];
#pragma unroll
        for (int k = 0; k < 4; k++) { const int y = tid * 4 + k; if (y < nRows) s_row[y] = start; start += cnt[k]; }
        if (tid == kStereoThreads - 1) s_total = start;
    }
    __syncthreads();
    const bool table = s_total <= ncap * kRowItemsPerKp;   // otherwise (extreme scale factors) every left keypoint walks all right keypoints
    if (table) {
        // scatter with the row starts as cursors (afterwards s_row[y] = end of row y = start of row y + 1)
        for (int i = tid; i < Nr; i += kStereoThreads)
            for (int y = max((int)s_minr[i], 0); y <= min((int)s_maxr[i], nRows - 1); y++) items[atomicAdd(&s_row[y], 1)] = (unsigned short)i;
        __threadfence_block();
        __syncthreads();
    }

    const float minZ = c.b, minD = 0.f, maxD = c.bf / minZ;
    const int thOrbDist = (100 + 50) / 2;
    // ---- Hamming scan (:744-789) ----
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
            int bestKey = (100 << 16);   // TH_HIGH | index: strict minimum of (distance, right index)
            const int levelL = kp.octave;
            auto consider = [&](int iR) {
                const int o = s_oct[iR];
                if (o < levelL - 1 || o > levelL + 1) return;
                const float uR = s_x[iR];
                if (uR >= minU && uR <= maxU) {
                    const uint4* d = (const uint4*)(s_desc + iR * 8);
                    const uint4 d0 = d[0], d1 = d[1];
                    const int dist = __popc(q[0] ^ d0.x) + __popc(q[1] ^ d0.y) + __popc(q[2] ^ d0.z) + __popc(q[3] ^ d0.w) +
                                     __popc(q[4] ^ d1.x) + __popc(q[5] ^ d1.y) + __popc(q[6] ^ d1.z) + __popc(q[7] ^ d1.w);
                    const int key = (dist << 16) | iR;
                    if (key < bestKey) bestKey = key;
                }
            };
            if (table) {
                const int st = row > 0 ? s_row[row - 1] : 0, en = s_row[row];
                for (int t = st; t < en; t++) consider((int)items[t]);
            } else {
                for (int iR = 0; iR < Nr; iR++)
                    if (row >= s_minr[iR] && row <= s_maxr[iR]) consider(iR);
            }
            const int bestDist = bestKey >> 16;
            if (bestDist < 100 && bestDist < thOrbDist) best = bestKey & 0xFFFF;
        }
        best_out[iL] = (short)best;
    }
}

// ---- SAD refinement (:792-863), one wavefront per left keypoint ----
__global__ __launch_bounds__(256) void k_stereo_sad(StereoCtx c, int ncap) {
    const int bidx = blockIdx.y, lane = threadIdx.x & 63;
    const int iL = blockIdx.x * 4 + (threadIdx.x >> 6);
    int N, Nr;
    if (!stereo_counts(c, bidx, ncap, N, Nr) || iL >= N) return;
    const int bestIdxR = c.best[(long long)bidx * c.kp_stride + iL];
    if (bestIdxR < 0) return;
    const oslam_keypoint_t kp = c.kpL[(long long)bidx * c.kp_stride + iL];
    const float uR0 = c.kpR[(long long)bidx * c.kp_stride + bestIdxR].x;
    float* uRight = c.uRight + (long long)bidx * c.kp_stride;
    float* depth = c.depth + (long long)bidx * c.kp_stride;
    int* sad = c.sad + (long long)bidx * c.kp_stride;
    const float minZ = c.b, minD = 0.f, maxD = c.bf / minZ;
    {
        const int oct = kp.octave;
        const float sf = c.invScale[oct];
        const float scaleduL = roundf(kp.x * sf), scaledvL = roundf(kp.y * sf), scaleduR0 = roundf(uR0 * sf);
        const int w = 5, L = 5;
        const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
        if (iniu < 0 || endu >= (float)c.R.w[oct]) return;
        const uint8_t* imL = c.L.lv[oct] + (oct == 0 ? c.img0_stride_L : c.pyr_stride_L) * bidx;
        const uint8_t* imR = c.R.lv[oct] + (oct == 0 ? c.img0_stride_R : c.pyr_stride_R) * bidx;
        const int pL = c.L.pitch[oct], pR = c.R.pitch[oct];
        const int cy = (int)scaledvL, cxL = (int)scaleduL, cxR0 = (int)scaleduR0;
        // guard the window reads (the reference does not check the left window / the right -10 side)
        if (cy - w < 0 || cy + w >= c.L.h[oct] || cxL - w < 0 || cxL + w >= c.L.w[oct] || cxR0 - L - w < 0 || cxR0 + L + w >= c.R.w[oct]) return;
        // lanes hold the 121 left-patch values (2 per lane).  The 11 x 11 left patch and the 11 x 21 right strip are fetched ONCE per wavefront as aligned
        // words (44 + 66 words: three load instructions) into LDS and the lanes pick their bytes there.  Round 4 had every lane load its 35 bytes from global
        // memory itself: 35 byte-gather instructions per keypoint, and the kernel — the largest of the stereo workload, 1.98 ms per 256 frames — ran at the
        // texture addresser's rate (rocprofv3: profiles/r05_stereo_bench_kernel_stats_before.csv).  A keypoint lies >= 19 pixels inside its level and the
        // guard above keeps the strip inside the row, so the words that hold the first / last bytes are inside the image buffer.
        __shared__ uint32_t s_pat[4][112];
        uint32_t* sp = s_pat[threadIdx.x >> 6];
        const uint8_t* spb = (const uint8_t*)sp;
        {
            if (lane < 44) {
                const uint8_t* a = imL + (long long)(cy + (lane >> 2) - w) * pL + cxL - w;
                sp[lane] = *(const uint32_t*)((const uint8_t*)((unsigned long long)a & ~3ull) + 4 * (lane & 3));
            }
            const int r = lane / 6, wd = lane - 6 * r;   // right strip rows 0 .. 10 (lanes 0 .. 63 cover words 0 .. 63; lanes 0, 1 add words 64, 65)
            const uint8_t* a = imR + (long long)(cy + r - w) * pR + cxR0 - L - w;
            sp[44 + lane] = *(const uint32_t*)((const uint8_t*)((unsigned long long)a & ~3ull) + 4 * wd);
            if (lane < 2) {
                const uint8_t* a2 = imR + (long long)(cy + 10 - w) * pR + cxR0 - L - w;
                sp[44 + 64 + lane] = *(const uint32_t*)((const uint8_t*)((unsigned long long)a2 & ~3ull) + 4 * (4 + lane));
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        auto left_px = [&](int dy, int dx) -> int {    // imL[(cy + dy) * pL + cxL + dx]
            const unsigned sh = (unsigned)((unsigned long long)(imL + (long long)(cy + dy) * pL + cxL - w) & 3ull);
            return (int)spb[(dy + w) * 16 + sh + dx + w];
        };
        auto right_px = [&](int dy, int x) -> int {    // imR[(cy + dy) * pR + cxR0 - 10 + x], x = 0 .. 20
            const unsigned sh = (unsigned)((unsigned long long)(imR + (long long)(cy + dy) * pR + cxR0 - L - w) & 3ull);
            return (int)spb[176 + (dy + w) * 24 + sh + x];
        };
        const bool has1 = lane + 64 < 121;
        const int dy0 = lane / 11 - w, dx0 = lane % 11 - w;
        const int dy1 = has1 ? (lane + 64) / 11 - w : 0, dx1 = has1 ? (lane + 64) % 11 - w : 0;
        const int centerL = left_px(0, 0);
        const int l0 = left_px(dy0, dx0);
        const int l1 = left_px(dy1, dx1);
        int r0[11], r1[11], rc[11];
#pragma unroll
        for (int k = 0; k < 11; k++) {   // cxR = cxR0 + k - 5: strip column k + 5 (+ dx)
            rc[k] = right_px(0, k + 5);
            r0[k] = right_px(dy0, k + 5 + dx0);
            r1[k] = right_px(dy1, k + 5 + dx1);
        }
        const int lv0 = l0 - centerL, lv1 = l1 - centerL;
        int bestDist = 0x7fffffff, bestinc = 0;
        float d_m1 = 0, d_0 = 0, d_p1 = 0;
        float vd[11];
        {   // the 11 SADs: lane partials, then all 11 wavefront sums at once in the vector ALU (lane_ops.h; 66 ds_bpermute before)
            int part[11], sums[3];
#pragma unroll
            for (int inc = -5; inc <= 5; inc++) {
                int s = abs(lv0 - (r0[inc + 5] - rc[inc + 5]));
                if (has1) s += abs(lv1 - (r1[inc + 5] - rc[inc + 5]));
                part[inc + 5] = s;
            }
            wave_sums_i32<11>(part, sums);
#pragma unroll
            for (int inc = -5; inc <= 5; inc++) {
                const int s = wave_sums_get<3>(sums, inc + 5);
                vd[inc + 5] = (float)s;
                if ((float)s < (float)bestDist) { bestDist = s; bestinc = inc; }
            }
        }
        if (bestinc == -L || bestinc == L) return;
#pragma unroll
        for (int k = 1; k < 10; k++)
            if (k == bestinc + 5) { d_m1 = vd[k - 1]; d_0 = vd[k]; d_p1 = vd[k + 1]; }
        const float deltaR = __fdiv_rn(d_m1 - d_p1, 2.0f * (d_m1 + d_p1 - 2.0f * d_0));
        if (deltaR < -1 || deltaR > 1) return;   // NaN (flat SAD) passes, exactly like the reference's comparison
        float bestuR = c.scale[oct] * ((float)scaleduR0 + (float)bestinc + deltaR);
        float disparity = kp.x - bestuR;
        if (disparity >= minD && disparity < maxD) {
            if (disparity <= 0) { disparity = 0.01f; bestuR = (float)((double)kp.x - 0.01); }
            if (lane == 0) {
                depth[iL] = __fdiv_rn(c.bf, disparity);
                uRight[iL] = bestuR;
                sad[iL] = bestDist;
                atomicAdd(&c.n_matched[bidx], 1);
            }
        }
    }
}

// ---- median-SAD outlier cut (:866-879): median = element size/2 of the distance-sorted list, found by bisection on the value ----
__global__ __launch_bounds__(kStereoThreads) void k_stereo_cut(StereoCtx c, int ncap) {
    const int bidx = blockIdx.x, tid = threadIdx.x;
    int N, Nr;
    if (!stereo_counts(c, bidx, ncap, N, Nr)) return;
    const int M = c.n_matched[bidx];
    if (M <= 0) return;
    float* uRight = c.uRight + (long long)bidx * c.kp_stride;
    float* depth = c.depth + (long long)bidx * c.kp_stride;
    const int* sad = c.sad + (long long)bidx * c.kp_stride;
    constexpr int kPer = (kStereoMaxKps + kStereoThreads - 1) / kStereoThreads;
    int mine[kPer];
#pragma unroll
    for (int k = 0; k < kPer; k++) { const int i = tid + k * kStereoThreads; mine[k] = i < N ? sad[i] : -1; }
    __shared__ int s_count[2];
    // smallest d with #{sad <= d} >= M/2 + 1: the value of element M/2 of the sorted list (SAD <= 121 * 510 < 2^16)
    int lo = 0, hi = 65535;
    if (tid < 2) s_count[tid] = 0;
    __syncthreads();
    int ph = 0;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < kPer; k++) cnt += (mine[k] >= 0 && mine[k] <= mid);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
        if ((tid & 63) == 0 && cnt) atomicAdd(&s_count[ph], cnt);
        __syncthreads();
        const int total = s_count[ph];
        if (tid == 0) s_count[ph ^ 1] = 0;
        ph ^= 1;
        __syncthreads();
        if (total >= M / 2 + 1) hi = mid; else lo = mid + 1;
    }
    const float median = (float)lo;
    const float thDist = 1.5f * 1.4f * median;
#pragma unroll
    for (int k = 0; k < kPer; k++) {
        const int i = tid + k * kStereoThreads;
        if (i < N && mine[k] >= 0 && !((float)mine[k] < thDist)) { uRight[i] = -1.0f; depth[i] = -1.0f; }
    }
}

}  // namespace oslam

using namespace oslam;

struct oslam_stereo {
    int device = 0, max_batch = 0, max_kps = 0;
    size_t lds = 0;
    float* d_uRight = nullptr; float* d_depth = nullptr; int* d_sad = nullptr; int* d_nm = nullptr;
    short* d_best = nullptr; unsigned short* d_items = nullptr;
    oslam_keypoint_t* d_kpL = nullptr; oslam_keypoint_t* d_kpR = nullptr; uint8_t* d_descL = nullptr; uint8_t* d_descR = nullptr;
};

extern "C" {

void oslam_stereo_destroy(oslam_stereo_t* h) {
    if (!h) return;
    void* ptrs[] = {h->d_uRight, h->d_depth, h->d_sad, h->d_nm, h->d_best, h->d_items, h->d_kpL, h->d_kpR, h->d_descL, h->d_descR};
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
    h->lds = (size_t)max_keypoints * (32 + 4 + 2 + 2 + 1) + (kStereoMaxRows + 1) * 4 + 64;   // k_stereo_scan: descriptors, u, row band, octave + row starts
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
    ALLOC(h->d_best, B * NK * sizeof(short)); ALLOC(h->d_items, B * NK * kRowItemsPerKp * sizeof(unsigned short));
    ALLOC(h->d_kpL, NK * sizeof(oslam_keypoint_t)); ALLOC(h->d_kpR, NK * sizeof(oslam_keypoint_t)); ALLOC(h->d_descL, NK * 32); ALLOC(h->d_descR, NK * 32);
#undef ALLOC
    OSLAM_HIP_CHECK(hipFuncSetAttribute((const void*)k_stereo_scan, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds));
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
    c.uRight = h->d_uRight; c.depth = h->d_depth; c.sad = h->d_sad; c.n_matched = h->d_nm; c.best = h->d_best; c.row_items = h->d_items;
    if (c.L.h[0] > kStereoMaxRows) { set_error("image height %d above the stereo row table (%d)", c.L.h[0], kStereoMaxRows); return OSLAM_E_CAPACITY; }
    hipLaunchKernelGGL(k_stereo_scan, dim3(batch), dim3(kStereoThreads), h->lds, (hipStream_t)stream, c, h->max_kps);
    hipLaunchKernelGGL(k_stereo_sad, dim3((kp_stride + 3) / 4, batch), dim3(256), 0, (hipStream_t)stream, c, h->max_kps);
    hipLaunchKernelGGL(k_stereo_cut, dim3(batch), dim3(kStereoThreads), 0, (hipStream_t)stream, c, h->max_kps);
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
