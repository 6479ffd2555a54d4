// gfx950 BoW-guided matchers: ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) (reference
// src/ORBmatcher.cc:159-288) and ORBmatcher::SearchForTriangulation (:657-823).  The DBoW2 vocabulary
// is not part of the reference tree, so the FeatureVectors are inputs: side 1 as a flat list in the
// std::map iteration order (node ascending, indices in vector order), side 2 as CSR over its sorted
// node ids.  One 1024-thread workgroup per keyframe/frame pair; side-2 descriptors live in LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <vector>

#include "common.h"
#include "slam_pool.h"

namespace oslam {

constexpr int kBowThreads = 1024;
constexpr int kBowMaxKps = 2400;
constexpr int kBowHisto = 30;

struct BowCtx {
    int mode;   // 0 SearchByBoW, 1 SearchForTriangulation
    int nq; const int* q_idx1; const uint32_t* q_node;
    int N1; const oslam_keypoint_t* keys1; const uint8_t* desc1; const float* uRight1; const uint8_t* flag1;
    int N2; const oslam_keypoint_t* keys2; const uint8_t* desc2; const float* uRight2; const uint8_t* has_mp2;
    int nNodes; const uint32_t* nodes; const int* start; const int* items;
    float nnratio; int checkOri; int bOnlyStereo;
    float F12[9], ex, ey, scale[OSLAM_MAX_LEVELS], sigma2[OSLAM_MAX_LEVELS];
    int* out; int* nmatches; int* q_best;   // q_best [nq] scratch: accepted side-2 index per query or -1
};

__device__ __forceinline__ int bow_find_node(const BowCtx& c, uint32_t node) {
    int lo = 0, hi = c.nNodes;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (c.nodes[mid] < node) lo = mid + 1; else hi = mid;
    }
    return (lo < c.nNodes && c.nodes[lo] == node) ? lo : -1;
}

__device__ __forceinline__ void bow_body(const BowCtx& c, const int ncap) {
    const int tid = threadIdx.x;
    extern __shared__ __align__(16) uint8_t smem[];
    uint32_t* s_desc = (uint32_t*)smem;               // [ncap][8] side-2 descriptors
    int* s_Bcur = (int*)(s_desc + (size_t)ncap * 8);  // [ncap]
    int* s_Bprev = s_Bcur + ncap;                     // [ncap]
    __shared__ int s_hist[kBowHisto], s_changed, s_nm, s_ind[3];
    const int N2 = c.N2, nq = c.nq;
    if (N2 > ncap || N2 < 0 || nq < 0) {
        if (tid == 0) *c.nmatches = -1;
        return;
    }
    const uint32_t* g2 = (const uint32_t*)c.desc2;
    for (int i = tid; i < N2 * 8; i += kBowThreads) s_desc[i] = g2[i];
    for (int i = tid; i < N2; i += kBowThreads) s_Bprev[i] = 0x7fffffff;
    if (c.mode == 0) for (int i = tid; i < N2; i += kBowThreads) c.out[i] = -1;
    else for (int i = tid; i < c.N1; i += kBowThreads) c.out[i] = -1;
    __syncthreads();

    const int maxit = c.mode == 0 ? nq + 2 : 1;
    for (int it = 0; it < maxit; it++) {
        for (int i = tid; i < N2; i += kBowThreads) s_Bcur[i] = 0x7fffffff;
        if (tid == 0) s_changed = 0;
        __syncthreads();
        for (int q = tid; q < nq; q += kBowThreads) {
            const int idx1 = c.q_idx1[q];
            int best = -1;
            bool ok = c.mode == 0 ? (c.flag1[idx1] != 0) : (c.flag1[idx1] == 0);
            const bool bStereo1 = c.mode == 1 && c.uRight1[idx1] >= 0;
            if (c.mode == 1 && c.bOnlyStereo && !bStereo1) ok = false;
            const int nd = ok ? bow_find_node(c, c.q_node[q]) : -1;
            if (nd >= 0) {
                uint32_t qd[8];
                const uint32_t* qp = (const uint32_t*)(c.desc1 + (size_t)idx1 * 32);
#pragma unroll
                for (int w = 0; w < 8; w++) qd[w] = qp[w];
                if (c.mode == 0) {
                    int bestDist1 = 256, bestDist2 = 256;
                    for (int t = c.start[nd]; t < c.start[nd + 1]; t++) {
                        const int k = c.items[t];
                        if (s_Bprev[k] < q) continue;   // vpMapPointMatches[realIdxF] already set (:211-212)
                        const uint32_t* d = s_desc + k * 8;
                        int dist = 0;
#pragma unroll
                        for (int w = 0; w < 8; w++) dist += __popc(qd[w] ^ d[w]);
                        if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; best = k; }
                        else if (dist < bestDist2) bestDist2 = dist;
                    }
                    if (!(bestDist1 <= 50 && (float)bestDist1 < c.nnratio * (float)bestDist2)) best = -1;
                } else {
                    const oslam_keypoint_t kp1 = c.keys1[idx1];
                    // epipolar line l = x1' F12 (:142-145)
                    const float la = kp1.x * c.F12[0] + kp1.y * c.F12[3] + c.F12[6];
                    const float lb = kp1.x * c.F12[1] + kp1.y * c.F12[4] + c.F12[7];
                    const float lc = kp1.x * c.F12[2] + kp1.y * c.F12[5] + c.F12[8];
                    const float den = la * la + lb * lb;
                    int bestDist = 50;
                    for (int t = c.start[nd]; t < c.start[nd + 1]; t++) {
                        const int k = c.items[t];
                        if (c.has_mp2[k]) continue;
                        const bool bStereo2 = c.uRight2[k] >= 0;
                        if (c.bOnlyStereo && !bStereo2) continue;
                        const uint32_t* d = s_desc + k * 8;
                        int dist = 0;
#pragma unroll
                        for (int w = 0; w < 8; w++) dist += __popc(qd[w] ^ d[w]);
                        if (dist > 50 || dist > bestDist) continue;
                        const oslam_keypoint_t kp2 = c.keys2[k];
                        if (!bStereo1 && !bStereo2) {
                            const float dx = c.ex - kp2.x, dy = c.ey - kp2.y;
                            if (dx * dx + dy * dy < 100 * c.scale[kp2.octave]) continue;
                        }
                        const float num = la * kp2.x + lb * kp2.y + lc;
                        if (den == 0) continue;
                        const float dsqr = __fdiv_rn(num * num, den);
                        if ((double)dsqr < 3.84 * (double)c.sigma2[kp2.octave]) { best = k; bestDist = dist; }
                    }
                }
            }
            c.q_best[q] = best;
            if (c.mode == 0 && best >= 0) atomicMin(&s_Bcur[best], q);
        }
        __syncthreads();
        int diff = 0;
        for (int i = tid; i < N2; i += kBowThreads) diff |= (s_Bcur[i] != s_Bprev[i]);
        if (diff) s_changed = 1;
        __syncthreads();
        const int changed = s_changed;
        __syncthreads();
        if (!changed || c.mode == 1) break;
        { int* t = s_Bcur; s_Bcur = s_Bprev; s_Bprev = t; }
    }

    // results + rotation consistency (:264-283 / :785-805)
    if (tid < kBowHisto) s_hist[tid] = 0;
    if (tid == 0) s_nm = 0;
    __syncthreads();
    const float factor = 1.0f / kBowHisto;
    int local = 0;
    for (int q = tid; q < nq; q += kBowThreads) {
        const int k = c.q_best[q];
        if (k < 0) continue;
        local++;
        const int idx1 = c.q_idx1[q];
        if (c.mode == 0) c.out[k] = idx1; else c.out[idx1] = k;
        if (c.checkOri) {
            float rot = c.keys1[idx1].angle - c.keys2[k].angle;
            if (rot < 0.0f) rot += 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == kBowHisto) bin = 0;
            atomicAdd(&s_hist[bin], 1);
        }
    }
    if (local) atomicAdd(&s_nm, local);
    __syncthreads();
    if (c.checkOri) {
        if (tid == 0) {
            int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
            for (int i = 0; i < kBowHisto; i++) {
                const int s = s_hist[i];
                if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
                else if (s > max3) { max3 = s; ind3 = i; }
            }
            if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
            else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
            s_ind[0] = ind1; s_ind[1] = ind2; s_ind[2] = ind3;
        }
        __syncthreads();
        int removed = 0;
        for (int q = tid; q < nq; q += kBowThreads) {
            const int k = c.q_best[q];
            if (k < 0) continue;
            const int idx1 = c.q_idx1[q];
            float rot = c.keys1[idx1].angle - c.keys2[k].angle;
            if (rot < 0.0f) rot += 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == kBowHisto) bin = 0;
            if (bin != s_ind[0] && bin != s_ind[1] && bin != s_ind[2]) {
                removed++;
                if (c.mode == 0) c.out[k] = -2; else c.out[idx1] = -1;
            }
        }
        if (removed) atomicSub(&s_nm, removed);
        __syncthreads();
    }
    if (tid == 0) *c.nmatches = s_nm;
}

__global__ __launch_bounds__(kBowThreads) void k_search_bow(BowCtx c, int ncap) { bow_body(c, ncap); }
// one workgroup per pair; the contexts live in device memory
__global__ __launch_bounds__(kBowThreads) void k_search_bow_batch(const BowCtx* cs, int ncap) { bow_body(cs[blockIdx.x], ncap); }

}  // namespace oslam

using namespace oslam;

struct oslam_bow {
    int device = 0, max_kps = 0;
    size_t lds = 0;
    struct Buf { void* p = nullptr; size_t cap = 0; };
    Buf q_idx1, q_node, keys1, desc1, ur1, flag1, keys2, desc2, ur2, mp2, nodes, start, items, out, qbest, nm;
    uint8_t* st_h = nullptr; uint8_t* st_d = nullptr; size_t st_cap = 0;   // batch staging: pinned block mirrored on the device
    hipStream_t strm = nullptr;   // the batch form runs on the handle's own non-blocking stream (created on first use) or on the stream given by bow_use_stream
    bool owns_strm = true;
    // device time of the batch kernel (bench.py's kernel-time groups): HIP events on `strm`
    int timing = 0; hipEvent_t ev0 = nullptr, ev1 = nullptr; double kern_ms = 0; long long kern_n = 0;
};

static int bow_ensure(oslam_bow::Buf& b, size_t bytes) {
    if (b.p && bytes <= b.cap) return OSLAM_OK;
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = bytes + bytes / 2 + 256;
    OSLAM_HIP_CHECK(hipMalloc(&b.p, b.cap));
    return OSLAM_OK;
}
static int bow_up(oslam_bow::Buf& b, const void* src, size_t bytes) {
    int rc = bow_ensure(b, bytes ? bytes : 4);
    if (rc) return rc;
    if (bytes && src) OSLAM_HIP_CHECK(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    return OSLAM_OK;
}

extern "C" {

int oslam_bow_kernel_time(oslam_bow_t* h, int enable, double* ms_out, long long* launches_out) {
    if (!h) { set_error("NULL handle"); return OSLAM_E_INVALID; }
    OSLAM_HIP_CHECK(hipSetDevice(h->device));
    if (enable && !h->ev0) { OSLAM_HIP_CHECK(hipEventCreate(&h->ev0)); OSLAM_HIP_CHECK(hipEventCreate(&h->ev1)); }
    if (ms_out) *ms_out = h->kern_ms;
    if (launches_out) *launches_out = h->kern_n;
    h->kern_ms = 0; h->kern_n = 0; h->timing = enable;
    return OSLAM_OK;
}

void oslam_bow_destroy(oslam_bow_t* h) {
    if (!h) return;
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    oslam_bow::Buf* bs[] = {&h->q_idx1, &h->q_node, &h->keys1, &h->desc1, &h->ur1, &h->flag1, &h->keys2, &h->desc2, &h->ur2, &h->mp2,
                            &h->nodes, &h->start, &h->items, &h->out, &h->qbest, &h->nm};
    for (auto* b : bs)
        if (b->p) (void)hipFree(b->p);
    if (h->st_h) (void)hipHostFree(h->st_h);
    if (h->st_d) (void)hipFree(h->st_d);
    if (h->strm && h->owns_strm) (void)hipStreamDestroy(h->strm);
    delete h;
}

extern "C++" {
namespace oslam {
void bow_use_stream(oslam_bow* h, hipStream_t s) {
    if (!h || !s) return;
    if (h->strm && h->owns_strm) (void)hipStreamDestroy(h->strm);
    h->strm = s; h->owns_strm = false;
}
}  // namespace oslam
}  // extern "C++"

int oslam_bow_create(oslam_bow_t** out, int max_keypoints, int device) {
    if (!out) { set_error("out is NULL"); return OSLAM_E_INVALID; }
    *out = nullptr;
    if (max_keypoints < 1 || max_keypoints > kBowMaxKps) { set_error("oslam_bow_create: max_keypoints must be in [1,%d]", kBowMaxKps); return OSLAM_E_INVALID; }
    int ndev = oslam_device_count();
    if (ndev <= 0) { set_error("no HIP device visible: the gfx950 BoW matchers have no CPU fallback"); return OSLAM_E_HIP; }
    if (device < 0 || device >= ndev) { set_error("device out of range"); return OSLAM_E_INVALID; }
    OSLAM_HIP_CHECK(hipSetDevice(device));
    oslam_bow* h = new oslam_bow();
    h->device = device; h->max_kps = max_keypoints;
    h->lds = (size_t)max_keypoints * 40 + 64;
    OSLAM_HIP_CHECK(hipFuncSetAttribute((const void*)k_search_bow, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds));
    *out = h;
    return OSLAM_OK;
}

static int bow_run(oslam_bow_t* h, int mode, const oslam_bow_side1_t* s1, const oslam_bow_side2_t* s2, float nnratio, int checkOri,
                   const float* F12, float ex, float ey, const float* scaleFactors, const float* levelSigma2, int nlevels, int bOnlyStereo,
                   int32_t* out, int32_t* nmatches) {
    if (!h || !s1 || !s2 || !out || !nmatches) { set_error("NULL argument"); return OSLAM_E_INVALID; }
    if (s1->N < 0 || s2->N < 0 || s1->nq < 0 || s2->nNodes < 0 || s2->N > h->max_kps || s1->N > 65536 * 16) { set_error("keypoint counts exceed capacity %d", h->max_kps); return OSLAM_E_CAPACITY; }
    for (int q = 0; q < s1->nq; q++)
        if (s1->q_idx[q] < 0 || s1->q_idx[q] >= s1->N) { set_error("side-1 index out of range"); return OSLAM_E_INVALID; }
    const int nitems = s2->nNodes ? s2->start[s2->nNodes] : 0;
    for (int i = 0; i < nitems; i++)
        if (s2->items[i] < 0 || s2->items[i] >= s2->N) { set_error("side-2 index out of range"); return OSLAM_E_INVALID; }
    for (int i = 1; i < s2->nNodes; i++)
        if (!(s2->nodes[i - 1] < s2->nodes[i])) { set_error("side-2 node ids must be strictly ascending"); return OSLAM_E_INVALID; }
    OSLAM_HIP_CHECK(hipSetDevice(h->device));
    int rc;
    std::vector<float> negs((size_t)std::max(std::max(s1->N, s2->N), 1), -1.0f);
    std::vector<uint8_t> zeros((size_t)std::max(std::max(s1->N, s2->N), 1), 0);
    if ((rc = bow_up(h->q_idx1, s1->q_idx, (size_t)s1->nq * 4)) || (rc = bow_up(h->q_node, s1->q_node, (size_t)s1->nq * 4)) ||
        (rc = bow_up(h->keys1, s1->keys, (size_t)s1->N * sizeof(oslam_keypoint_t))) || (rc = bow_up(h->desc1, s1->desc, (size_t)s1->N * 32)) ||
        (rc = bow_up(h->ur1, s1->uRight ? s1->uRight : negs.data(), (size_t)s1->N * 4)) || (rc = bow_up(h->flag1, s1->flag ? s1->flag : zeros.data(), (size_t)s1->N)) ||
        (rc = bow_up(h->keys2, s2->keys, (size_t)s2->N * sizeof(oslam_keypoint_t))) || (rc = bow_up(h->desc2, s2->desc, (size_t)s2->N * 32)) ||
        (rc = bow_up(h->ur2, s2->uRight ? s2->uRight : negs.data(), (size_t)s2->N * 4)) || (rc = bow_up(h->mp2, s2->has_mp ? s2->has_mp : zeros.data(), (size_t)s2->N)) ||
        (rc = bow_up(h->nodes, s2->nodes, (size_t)s2->nNodes * 4)) || (rc = bow_up(h->start, s2->start, (size_t)(s2->nNodes + 1) * 4)) ||
        (rc = bow_up(h->items, s2->items, (size_t)nitems * 4)) || (rc = bow_ensure(h->out, (size_t)std::max(std::max(s1->N, s2->N), 1) * 4)) ||
        (rc = bow_ensure(h->qbest, (size_t)std::max(s1->nq, 1) * 4)) || (rc = bow_ensure(h->nm, 4)))
        return rc;
    BowCtx c;
    memset(&c, 0, sizeof(c));
    c.mode = mode; c.nq = s1->nq; c.q_idx1 = (const int*)h->q_idx1.p; c.q_node = (const uint32_t*)h->q_node.p;
    c.N1 = s1->N; c.keys1 = (const oslam_keypoint_t*)h->keys1.p; c.desc1 = (const uint8_t*)h->desc1.p; c.uRight1 = (const float*)h->ur1.p; c.flag1 = (const uint8_t*)h->flag1.p;
    c.N2 = s2->N; c.keys2 = (const oslam_keypoint_t*)h->keys2.p; c.desc2 = (const uint8_t*)h->desc2.p; c.uRight2 = (const float*)h->ur2.p; c.has_mp2 = (const uint8_t*)h->mp2.p;
    c.nNodes = s2->nNodes; c.nodes = (const uint32_t*)h->nodes.p; c.start = (const int*)h->start.p; c.items = (const int*)h->items.p;
    c.nnratio = nnratio; c.checkOri = checkOri; c.bOnlyStereo = bOnlyStereo;
    if (F12) for (int i = 0; i < 9; i++) c.F12[i] = F12[i];
    c.ex = ex; c.ey = ey;
    for (int i = 0; i < OSLAM_MAX_LEVELS; i++) {
        c.scale[i] = (scaleFactors && i < nlevels) ? scaleFactors[i] : 0.f;
        c.sigma2[i] = (levelSigma2 && i < nlevels) ? levelSigma2[i] : 0.f;
    }
    c.out = (int*)h->out.p; c.nmatches = (int*)h->nm.p; c.q_best = (int*)h->qbest.p;
    hipLaunchKernelGGL(k_search_bow, dim3(1), dim3(kBowThreads), h->lds, nullptr, c, h->max_kps);
    OSLAM_HIP_CHECK(hipGetLastError());
    OSLAM_HIP_CHECK(hipDeviceSynchronize());
    OSLAM_HIP_CHECK(hipMemcpy(nmatches, h->nm.p, 4, hipMemcpyDeviceToHost));
    if (*nmatches < 0) { set_error("BoW matcher kernel rejected the pair (capacity)"); return OSLAM_E_CAPACITY; }
    const int nout = mode == 0 ? s2->N : s1->N;
    if (nout > 0) OSLAM_HIP_CHECK(hipMemcpy(out, h->out.p, (size_t)nout * 4, hipMemcpyDeviceToHost));
    return OSLAM_OK;
}

int oslam_match_search_by_bow(oslam_bow_t* h, const oslam_bow_side1_t* kf, const oslam_bow_side2_t* frame, float nnratio, int checkOri,
                              int32_t* match_f, int32_t* nmatches) {
    return bow_run(h, 0, kf, frame, nnratio, checkOri, nullptr, 0.f, 0.f, nullptr, nullptr, 0, 0, match_f, nmatches);
}

int oslam_match_search_for_triangulation(oslam_bow_t* h, const oslam_bow_side1_t* kf1, const oslam_bow_side2_t* kf2, const float F12[9], float ex,
                                         float ey, const float* scaleFactors, const float* levelSigma2, int nlevels, int bOnlyStereo,
                                         int checkOri, int32_t* match12, int32_t* nmatches) {
    if (!F12 || !scaleFactors || !levelSigma2 || nlevels < 1 || nlevels > OSLAM_MAX_LEVELS) { set_error("bad F12 / scale tables"); return OSLAM_E_INVALID; }
    return bow_run(h, 1, kf1, kf2, 0.f, checkOri, F12, ex, ey, scaleFactors, levelSigma2, nlevels, bOnlyStereo, match12, nmatches);
}

}  // extern "C"

// Batch of independent pairs (SearchByBoW and / or SearchForTriangulation), one workgroup each in ONE launch: all inputs are packed into one
// pinned block and reach the device in one copy; the results come back in one copy after one synchronisation.
static int bow_batch(oslam_bow_t* h, int n, oslam_bow_job_t* jobs, const oslam_bow_resident_t* res, const float* scaleFactors, const float* levelSigma2, int nlevels);

extern "C" int oslam_match_bow_batch(oslam_bow_t* h, int n, oslam_bow_job_t* jobs, const float* scaleFactors, const float* levelSigma2, int nlevels) {
    return bow_batch(h, n, jobs, nullptr, scaleFactors, levelSigma2, nlevels);
}

// The same with device-resident keypoint / descriptor / stereo-coordinate arrays for either side of a job (res[i] members that are NULL fall back to the
// job's host arrays): only the FeatureVector index lists and the flags travel.  The caller guarantees that the resident arrays are complete (its copies
// have been synchronised) — the launch runs on the handle's own stream.
extern "C" int oslam_match_bow_batch_resident(oslam_bow_t* h, int n, oslam_bow_job_t* jobs, const oslam_bow_resident_t* res, const float* scaleFactors,
                                              const float* levelSigma2, int nlevels) {
    return bow_batch(h, n, jobs, res, scaleFactors, levelSigma2, nlevels);
}

static int bow_batch(oslam_bow_t* h, int n, oslam_bow_job_t* jobs, const oslam_bow_resident_t* res, const float* scaleFactors, const float* levelSigma2, int nlevels) {
    if (!h || n < 0 || (n > 0 && !jobs)) { set_error("NULL argument"); return OSLAM_E_INVALID; }
    if (n == 0) return OSLAM_OK;
    bool anyTri = false;
    for (int i = 0; i < n; i++) anyTri = anyTri || jobs[i].triangulation;
    if (anyTri && (!scaleFactors || !levelSigma2 || nlevels < 1 || nlevels > OSLAM_MAX_LEVELS)) { set_error("bad scale tables"); return OSLAM_E_INVALID; }
    struct Off { size_t q_idx, q_node, keys1, desc1, ur1, flag1, keys2, desc2, ur2, mp2, nodes, start, items, out, qbest, nm; int nitems, nout; };
    std::vector<Off> off(n);
    size_t at = align_up(sizeof(BowCtx) * (size_t)n, 256);
    auto take = [&](size_t bytes) { const size_t a = at; at += align_up(bytes ? bytes : 4, 256); return a; };
    for (int i = 0; i < n; i++) {
        const oslam_bow_side1_t& s1 = jobs[i].s1;
        const oslam_bow_side2_t& s2 = jobs[i].s2;
        if (!jobs[i].match || s1.N < 0 || s2.N < 0 || s1.nq < 0 || s2.nNodes < 0 || s2.N > h->max_kps || (s1.N > 0 && (!s1.keys || !s1.desc)) || (s2.N > 0 && (!s2.keys || !s2.desc)) ||
            (s1.nq > 0 && (!s1.q_idx || !s1.q_node)) || (s2.nNodes > 0 && (!s2.nodes || !s2.start || !s2.items))) { set_error("bow batch: bad job %d", i); return OSLAM_E_INVALID; }
        for (int q = 0; q < s1.nq; q++)
            if (s1.q_idx[q] < 0 || s1.q_idx[q] >= s1.N) { set_error("side-1 index out of range"); return OSLAM_E_INVALID; }
        const int nitems = s2.nNodes ? s2.start[s2.nNodes] : 0;
        for (int k = 0; k < nitems; k++)
            if (s2.items[k] < 0 || s2.items[k] >= s2.N) { set_error("side-2 index out of range"); return OSLAM_E_INVALID; }
        for (int k = 1; k < s2.nNodes; k++)
            if (!(s2.nodes[k - 1] < s2.nodes[k])) { set_error("side-2 node ids must be strictly ascending"); return OSLAM_E_INVALID; }
        Off& o = off[i];
        o.nitems = nitems; o.nout = jobs[i].triangulation ? s1.N : s2.N;
        const bool r1 = res && res[i].d_keys1 && res[i].d_desc1, r2 = res && res[i].d_keys2 && res[i].d_desc2;
        const bool u1 = r1 && res[i].d_uRight1 && s1.uRight, u2 = r2 && res[i].d_uRight2 && s2.uRight;   // a NULL host uRight means "all -1": that is packed, resident or not
        o.q_idx = take((size_t)s1.nq * 4); o.q_node = take((size_t)s1.nq * 4);
        o.keys1 = r1 ? 0 : take((size_t)s1.N * sizeof(oslam_keypoint_t)); o.desc1 = r1 ? 0 : take((size_t)s1.N * 32);
        o.ur1 = u1 ? 0 : take((size_t)s1.N * 4); o.flag1 = take((size_t)s1.N);
        o.keys2 = r2 ? 0 : take((size_t)s2.N * sizeof(oslam_keypoint_t)); o.desc2 = r2 ? 0 : take((size_t)s2.N * 32);
        o.ur2 = u2 ? 0 : take((size_t)s2.N * 4); o.mp2 = take((size_t)s2.N); o.nodes = take((size_t)s2.nNodes * 4); o.start = take((size_t)(s2.nNodes + 1) * 4);
        o.items = take((size_t)nitems * 4);
    }
    const size_t in_bytes = at;
    for (int i = 0; i < n; i++) { off[i].out = take((size_t)std::max(off[i].nout, 1) * 4); off[i].nm = take(4); }
    const size_t io_bytes = at;
    for (int i = 0; i < n; i++) off[i].qbest = take((size_t)std::max(jobs[i].s1.nq, 1) * 4);
    OSLAM_HIP_CHECK(hipSetDevice(h->device));
    if (at > h->st_cap) {
        OSLAM_HIP_CHECK(hipDeviceSynchronize());
        if (h->st_h) (void)hipHostFree(h->st_h);
        if (h->st_d) (void)hipFree(h->st_d);
        h->st_h = nullptr; h->st_d = nullptr; h->st_cap = 0;
        const size_t ncap = at + at / 2;
        OSLAM_HIP_CHECK(hipHostMalloc((void**)&h->st_h, ncap, 0));
        OSLAM_HIP_CHECK(hipMalloc((void**)&h->st_d, ncap));
        h->st_cap = ncap;
    }
    uint8_t* H = h->st_h;
    uint8_t* D = h->st_d;
    BowCtx* cs = (BowCtx*)H;
    oslam_drv::shared_parallel_for(n, [&](int i) {   // ~140 KB of keypoints / descriptors per job
        const oslam_bow_job_t& j = jobs[i];
        const oslam_bow_side1_t& s1 = j.s1;
        const oslam_bow_side2_t& s2 = j.s2;
        const Off& o = off[i];
        memcpy(H + o.q_idx, s1.q_idx, (size_t)s1.nq * 4); memcpy(H + o.q_node, s1.q_node, (size_t)s1.nq * 4);
        if (o.keys1) { memcpy(H + o.keys1, s1.keys, (size_t)s1.N * sizeof(oslam_keypoint_t)); memcpy(H + o.desc1, s1.desc, (size_t)s1.N * 32); }
        if (o.ur1) { if (s1.uRight) memcpy(H + o.ur1, s1.uRight, (size_t)s1.N * 4); else for (int k = 0; k < s1.N; k++) ((float*)(H + o.ur1))[k] = -1.0f; }
        if (s1.flag) memcpy(H + o.flag1, s1.flag, (size_t)s1.N); else memset(H + o.flag1, 0, (size_t)s1.N);
        if (o.keys2) { memcpy(H + o.keys2, s2.keys, (size_t)s2.N * sizeof(oslam_keypoint_t)); memcpy(H + o.desc2, s2.desc, (size_t)s2.N * 32); }
        if (o.ur2) { if (s2.uRight) memcpy(H + o.ur2, s2.uRight, (size_t)s2.N * 4); else for (int k = 0; k < s2.N; k++) ((float*)(H + o.ur2))[k] = -1.0f; }
        if (s2.has_mp) memcpy(H + o.mp2, s2.has_mp, (size_t)s2.N); else memset(H + o.mp2, 0, (size_t)s2.N);
        memcpy(H + o.nodes, s2.nodes, (size_t)s2.nNodes * 4);
        if (s2.nNodes) memcpy(H + o.start, s2.start, (size_t)(s2.nNodes + 1) * 4); else *(int*)(H + o.start) = 0;
        memcpy(H + o.items, s2.items, (size_t)o.nitems * 4);
        BowCtx& c = cs[i];
        memset(&c, 0, sizeof(c));
        c.mode = j.triangulation ? 1 : 0; c.nq = s1.nq; c.q_idx1 = (const int*)(D + o.q_idx); c.q_node = (const uint32_t*)(D + o.q_node);
        c.N1 = s1.N; c.keys1 = o.keys1 ? (const oslam_keypoint_t*)(D + o.keys1) : res[i].d_keys1; c.desc1 = o.desc1 ? D + o.desc1 : res[i].d_desc1;
        c.uRight1 = o.ur1 ? (const float*)(D + o.ur1) : res[i].d_uRight1; c.flag1 = D + o.flag1;
        c.N2 = s2.N; c.keys2 = o.keys2 ? (const oslam_keypoint_t*)(D + o.keys2) : res[i].d_keys2; c.desc2 = o.desc2 ? D + o.desc2 : res[i].d_desc2;
        c.uRight2 = o.ur2 ? (const float*)(D + o.ur2) : res[i].d_uRight2; c.has_mp2 = D + o.mp2;
        c.nNodes = s2.nNodes; c.nodes = (const uint32_t*)(D + o.nodes); c.start = (const int*)(D + o.start); c.items = (const int*)(D + o.items);
        c.nnratio = j.nnratio; c.checkOri = j.checkOri; c.bOnlyStereo = 0;
        for (int k = 0; k < 9; k++) c.F12[k] = j.F12[k];
        c.ex = j.ex; c.ey = j.ey;
        for (int k = 0; k < OSLAM_MAX_LEVELS; k++) {
            c.scale[k] = (scaleFactors && k < nlevels) ? scaleFactors[k] : 0.f;
            c.sigma2[k] = (levelSigma2 && k < nlevels) ? levelSigma2[k] : 0.f;
        }
        c.out = (int*)(D + o.out); c.nmatches = (int*)(D + o.nm); c.q_best = (int*)(D + o.qbest);
    });
    if (!h->strm) OSLAM_HIP_CHECK(hipStreamCreateWithFlags(&h->strm, hipStreamNonBlocking));
    OSLAM_HIP_CHECK(hipMemcpyAsync(D, H, in_bytes, hipMemcpyHostToDevice, h->strm));
    if (h->timing) (void)hipEventRecord(h->ev0, h->strm);
    hipLaunchKernelGGL(k_search_bow_batch, dim3(n), dim3(kBowThreads), h->lds, h->strm, (const BowCtx*)D, h->max_kps);
    if (h->timing) (void)hipEventRecord(h->ev1, h->strm);
    OSLAM_HIP_CHECK(hipGetLastError());
    OSLAM_HIP_CHECK(hipMemcpyAsync(H + in_bytes, D + in_bytes, io_bytes - in_bytes, hipMemcpyDeviceToHost, h->strm));
    OSLAM_HIP_CHECK(stream_wait(h->strm));
    if (h->timing) { float ms = 0.f; if (hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess) { h->kern_ms += ms; h->kern_n += 1; } }
    for (int i = 0; i < n; i++) {
        jobs[i].nmatches = *(const int*)(H + off[i].nm);
        if (jobs[i].nmatches < 0) { set_error("BoW matcher kernel rejected pair %d (capacity)", i); return OSLAM_E_CAPACITY; }
        memcpy(jobs[i].match, H + off[i].out, (size_t)off[i].nout * 4);
    }
    return OSLAM_OK;
}
