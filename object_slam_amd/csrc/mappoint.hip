// gfx950 MapPoint maintenance + Frame::isInFrustum (the steps right before
// SearchByProjection(F, vpMPs) and right after LocalBundleAdjustment), batched over map points:
//   MapPoint::ComputeDistinctiveDescriptors  reference src/MapPoint.cc:345-410
//   MapPoint::UpdateNormalAndDepth           reference src/MapPoint.cc:433-474
//   Frame::isInFrustum + MapPoint::PredictScale + the query fields of SearchByProjection
//                                            reference src/Frame.cc:509-565, src/MapPoint.cc:505-521, src/ORBmatcher.cc:57-67
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>

#include "common.h"

namespace oslam {

// one wavefront per map point; descriptors of its observations staged in LDS (chunks of 128)
constexpr int kDdMaxObs = 128;

__global__ __launch_bounds__(256) void k_distinctive(int P, const int* obs_start, const uint8_t* obs_desc, int* best_idx, uint8_t* out_desc) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int p = blockIdx.x * 4 + wv;
    __shared__ uint32_t s_d[4][kDdMaxObs * 8];
    if (p >= P) return;
    const int s = obs_start[p], N = obs_start[p + 1] - s;
    if (N <= 0) { if (lane == 0) best_idx[p] = -1; return; }
    const uint32_t* g = (const uint32_t*)(obs_desc + (size_t)s * 32);
    const bool in_lds = N <= kDdMaxObs;
    uint32_t* d = s_d[wv];
    if (in_lds) for (int i = lane; i < N * 8; i += 64) d[i] = g[i];
    __builtin_amdgcn_wave_barrier();
    const uint32_t* D = in_lds ? d : g;
    const int k = (int)(0.5 * (N - 1));   // vDists[0.5*(N-1)]
    int bestMedian = 0x7fffffff, bestI = 0;
    for (int i = lane; i < N; i += 64) {
        uint32_t qi[8];
#pragma unroll
        for (int w = 0; w < 8; w++) qi[w] = D[i * 8 + w];
        // k-th smallest of row i by bisection on the value (distances are integers in [0,256])
        int lo = 0, hi = 256;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            int cnt = 0;
            for (int j = 0; j < N; j++) {
                int dist = 0;
#pragma unroll
                for (int w = 0; w < 8; w++) dist += __popc(qi[w] ^ D[j * 8 + w]);
                cnt += dist <= mid;
            }
            if (cnt >= k + 1) hi = mid; else lo = mid + 1;
        }
        if (lo < bestMedian) { bestMedian = lo; bestI = i; }   // lanes visit i ascending: first minimum kept
    }
    // (median, index) lexicographic minimum over lanes = the reference's first strict minimum
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) {
        const int om = __shfl_xor(bestMedian, sft, 64), oi = __shfl_xor(bestI, sft, 64);
        if (om < bestMedian || (om == bestMedian && oi < bestI)) { bestMedian = om; bestI = oi; }
    }
    if (lane == 0) best_idx[p] = bestI;
    if (lane < 8) ((uint32_t*)(out_desc + (size_t)p * 32))[lane] = D[bestI * 8 + lane];
}

__device__ __forceinline__ double norm3d(float a, float b, float c) { return sqrt((double)a * a + (double)b * b + (double)c * c); }

__global__ __launch_bounds__(256) void k_update_normal_depth(int P, const float* Pos, const int* obs_start, const float* obs_Ow, const float* OwRef,
                                                             const float* levelScale, float lastScale, float* out) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const float px = Pos[p * 3], py = Pos[p * 3 + 1], pz = Pos[p * 3 + 2];
    const int s = obs_start[p], n = obs_start[p + 1] - s;
    if (n <= 0) { for (int k = 0; k < 5; k++) out[p * 5 + k] = 0.f; return; }
    float nx = 0, ny = 0, nz = 0;
    for (int i = 0; i < n; i++) {
        const float ax = px - obs_Ow[(s + i) * 3], ay = py - obs_Ow[(s + i) * 3 + 1], az = pz - obs_Ow[(s + i) * 3 + 2];
        const float inv = (float)(1.0 / norm3d(ax, ay, az));   // cv::scaleAdd with float alpha
        nx = ax * inv + nx;
        ny = ay * inv + ny;
        nz = az * inv + nz;
    }
    const float dist = (float)norm3d(px - OwRef[p * 3], py - OwRef[p * 3 + 1], pz - OwRef[p * 3 + 2]);
    const float maxD = dist * levelScale[p];
    const float invn = (float)(1.0 / (double)n);   // convertTo(alpha = 1/n): float scale
    out[p * 5] = nx * invn + 0.0f;
    out[p * 5 + 1] = ny * invn + 0.0f;
    out[p * 5 + 2] = nz * invn + 0.0f;
    out[p * 5 + 3] = maxD;
    out[p * 5 + 4] = __fdiv_rn(maxD, lastScale);
}

struct FrustumCtx {
    int M;
    const float* Pw; const float* Pn; const float* maxDist; const float* minDist; const uint8_t* obs_gt0; const uint8_t* mp_desc;
    float T[16], fx, fy, cx, cy, bf, minX, minY, maxX, maxY, cosLimit, logScale, th;
    float scale[OSLAM_MAX_LEVELS]; int nLevels;
    oslam_proj_query_t* out;
};

// one map point against one frame pose; T = the frame's Tcw, th = the search-radius factor
__device__ __forceinline__ bool frustum_point(const FrustumCtx& c, const float* __restrict__ T, const float th, const int i, oslam_proj_query_t* out);

__global__ __launch_bounds__(256) void k_is_in_frustum(FrustumCtx c) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= c.M) return;
    frustum_point(c, c.T, c.th, i, c.out + i);
}

// batch of frames: point arrays [batch][stride], one pose / th / count per frame; in_view[b][i] = mbTrackInView
// the point arrays are [batch][stride_in], the queries / in_view / skip flags [batch][stride_out]; skip[b][i] != 0: the point is not projected (an
// inactive query, in_view 0)
__global__ __launch_bounds__(256) void k_is_in_frustum_batch(FrustumCtx c, int stride_in, int stride_out, const int* __restrict__ d_M, const float* __restrict__ d_Tcw,
                                                             const float* __restrict__ d_th, const uint8_t* __restrict__ skip, uint8_t* __restrict__ in_view) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= d_M[b]) return;
    const size_t at = (size_t)b * stride_in + i, ao = (size_t)b * stride_out + i;
    if (skip && skip[ao]) {
        oslam_proj_query_t q;
        q.u = q.v = q.ur = q.radius = 0.f; q.minLevel = -1; q.maxLevel = -1; q.flags = 0; q.angle = 0.f;
        uint32_t* qd = (uint32_t*)q.desc;
#pragma unroll
        for (int w = 0; w < 8; w++) qd[w] = 0u;
        c.out[ao] = q;
        if (in_view) in_view[ao] = 0;
        return;
    }
    const bool ok = frustum_point(c, d_Tcw + (size_t)b * 16, d_th[b], (int)at, c.out + ao);
    if (in_view) in_view[ao] = ok;
}

__device__ __forceinline__ bool frustum_point(const FrustumCtx& c, const float* __restrict__ Tm, const float th, const int i, oslam_proj_query_t* out) {
    oslam_proj_query_t q;
    q.u = q.v = q.ur = q.radius = 0.f; q.minLevel = -1; q.maxLevel = -1; q.flags = 0; q.angle = 0.f;
    float Ow[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) s += (double)Tm[k * 4 + r] * (double)Tm[k * 4 + 3];
        Ow[r] = (float)(-1.0 * s);
    }
    const float P[3] = {c.Pw[i * 3], c.Pw[i * 3 + 1], c.Pw[i * 3 + 2]};
    float Pc[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {   // cv::gemm small-matrix branch: float accumulation
        const float t0 = Tm[r * 4] * P[0] + Tm[r * 4 + 1] * P[1] + Tm[r * 4 + 2] * P[2];
        Pc[r] = (float)((double)t0 + (double)Tm[r * 4 + 3]);
    }
    bool ok = !(Pc[2] < 0.0f);
    float u = 0, v = 0, invz = 0, viewCos = 0;
    int nScale = 0;
    if (ok) {
        invz = __fdiv_rn(1.0f, Pc[2]);
        u = c.fx * Pc[0] * invz + c.cx;
        v = c.fy * Pc[1] * invz + c.cy;
        ok = !(u < c.minX || u > c.maxX) && !(v < c.minY || v > c.maxY);
    }
    if (ok) {
        const float maxDistance = 1.2f * c.maxDist[i], minDistance = 0.8f * c.minDist[i];
        const float PO[3] = {P[0] - Ow[0], P[1] - Ow[1], P[2] - Ow[2]};
        const float dist = (float)norm3d(PO[0], PO[1], PO[2]);
        ok = !(dist < minDistance || dist > maxDistance);
        if (ok) {
            const float* n = c.Pn + i * 3;
            viewCos = (float)(((double)PO[0] * n[0] + (double)PO[1] * n[1] + (double)PO[2] * n[2]) / (double)dist);
            ok = !(viewCos < c.cosLimit);
            const float ratio = __fdiv_rn(c.maxDist[i], dist);
            // std::log(float): glibc logf is within 0.82 ulp; fp64 log rounded to float is the correctly rounded value
            nScale = (int)ceilf(__fdiv_rn((float)log((double)ratio), c.logScale));
            if (nScale < 0) nScale = 0;
            else if (nScale >= c.nLevels) nScale = c.nLevels - 1;
        }
    }
    if (ok) {
        float r = (double)viewCos > 0.998 ? 2.5f : 4.0f;
        if (th != 1.0f) r *= th;
        q.u = u; q.v = v; q.ur = u - c.bf * invz;
        q.radius = r * c.scale[nScale];
        q.minLevel = nScale - 1; q.maxLevel = nScale;
        q.flags = 1 | (c.obs_gt0[i] ? 2 : 0);
        q.angle = viewCos;
    }
    const uint32_t* sd = (const uint32_t*)(c.mp_desc + (size_t)i * 32);
    uint32_t* qd = (uint32_t*)q.desc;
#pragma unroll
    for (int w = 0; w < 8; w++) qd[w] = ok ? sd[w] : 0u;
    *out = q;
    return ok;
}

}  // namespace oslam

using namespace oslam;

struct oslam_mappoint {
    int device = 0;
    struct Buf { void* p = nullptr; size_t cap = 0; };
    Buf a, b, c, d, e, e2, f, g, g2, o1, o2;
    // batch form (oslam_mp_triangulate_pairs): one pinned block mirrored on the device, one stream — ONE upload, one launch, ONE download per call
    uint8_t* st_h = nullptr; uint8_t* st_d = nullptr; size_t st_cap = 0;
    hipStream_t strm = nullptr;
    bool owns_strm = true;
    int timing = 0; hipEvent_t ev0 = nullptr, ev1 = nullptr; double kern_ms = 0; long long kern_n = 0;   // device time of the batched triangulation kernel
};

static int mp_ensure(oslam_mappoint::Buf& b, size_t bytes) {
    if (b.p && bytes <= b.cap) return OSLAM_OK;
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = bytes + bytes / 2 + 256;
    OSLAM_HIP_CHECK(hipMalloc(&b.p, b.cap));
    return OSLAM_OK;
}
static int mp_up(oslam_mappoint::Buf& b, const void* src, size_t bytes) {
    int rc = mp_ensure(b, bytes ? bytes : 4);
    if (rc) return rc;
    if (bytes && src) OSLAM_HIP_CHECK(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    return OSLAM_OK;
}

extern "C" {

int oslam_mappoint_kernel_time(oslam_mappoint_t* h, int enable, double* ms_out, long long* launches_out) {
    if (!h) { set_error("NULL handle"); return OSLAM_E_INVALID; }
    OSLAM_HIP_CHECK(hipSetDevice(h->device));
    if (enable && !h->ev0) { OSLAM_HIP_CHECK(hipEventCreate(&h->ev0)); OSLAM_HIP_CHECK(hipEventCreate(&h->ev1)); }
    if (ms_out) *ms_out = h->kern_ms;
    if (launches_out) *launches_out = h->kern_n;
    h->kern_ms = 0; h->kern_n = 0; h->timing = enable;
    return OSLAM_OK;
}

void oslam_mappoint_destroy(oslam_mappoint_t* h) {
    if (!h) return;
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    oslam_mappoint::Buf* bs[] = {&h->a, &h->b, &h->c, &h->d, &h->e, &h->e2, &h->f, &h->g, &h->g2, &h->o1, &h->o2};
    for (auto* b : bs)
        if (b->p) (void)hipFree(b->p);
    if (h->st_h) (void)hipHostFree(h->st_h);
    if (h->st_d) (void)hipFree(h->st_d);
    if (h->strm && h->owns_strm) (void)hipStreamDestroy(h->strm);
    delete h;
}

extern "C++" {
namespace oslam {
void mappoint_use_stream(oslam_mappoint* h, hipStream_t s) {
    if (!h || !s) return;
    if (h->strm && h->owns_strm) (void)hipStreamDestroy(h->strm);
    h->strm = s; h->owns_strm = false;
}
}  // namespace oslam
}  // extern "C++"

int oslam_mappoint_create(oslam_mappoint_t** out, int device) {
    if (!out) { set_error("out is NULL"); return OSLAM_E_INVALID; }
    *out = nullptr;
    int ndev = oslam_device_count();
    if (ndev <= 0) { set_error("no HIP device visible: the gfx950 map-point kernels have no CPU fallback"); return OSLAM_E_HIP; }
    if (device < 0 || device >= ndev) { set_error("device out of range"); return OSLAM_E_INVALID; }
    OSLAM_HIP_CHECK(hipSetDevice(device));
    oslam_mappoint* h = new oslam_mappoint();
    h->device = device;
    *out = h;
    return OSLAM_OK;
}

int oslam_mp_distinctive_descriptors(oslam_mappoint_t* h, int P, const int32_t* obs_start, const uint8_t* obs_desc, int32_t* best_idx, uint8_t* out_desc) {
    if (!h || P < 0 || (P > 0 && (!obs_start || !best_idx || !out_desc))) { set_error("bad argument"); return OSLAM_E_INVALID; }
    if (P == 0) return OSLAM_OK;
    const int total = obs_start[P];
    if (total < 0 || (total > 0 && !obs_desc)) { set_error("bad observation table"); return OSLAM_E_INVALID; }
    for (int p = 0; p < P; p++) if (obs_start[p + 1] < obs_start[p]) { set_error("obs_start not monotone"); return OSLAM_E_INVALID; }
    OSLAM_HIP_CHECK(hipSetDevice(h->device));
    int rc;
    if ((rc = mp_up(h->a, obs_start, (size_t)(P + 1) * 4)) || (rc = mp_up(h->b, obs_desc, (size_t)total * 32)) || (rc = mp_ensure(h->o1, (size_t)P * 4)) ||
        (rc = mp_ensure(h->o2, (size_t)P * 32)))
        return rc;
    OSLAM_HIP_CHECK(hipMemset(h->o2.p, 0, (size_t)P * 32));
    hipLaunchKernelGGL(k_distinctive, dim3(div_up(P, 4)), dim3(256), 0, nullptr, P, (const int*)h->a.p, (const uint8_t*)h->b.p, (int*)h->o1.p, (uint8_t*)h->o2.p);
    OSLAM_HIP_CHECK(hipGetLastError());
    OSLAM_HIP_CHECK(hipMemcpy(best_idx, h->o1.p, (size_t)P * 4, hipMemcpyDeviceToHost));
    OSLAM_HIP_CHECK(hipMemcpy(out_desc, h->o2.p, (size_t)P * 32, hipMemcpyDeviceToHost));
    return OSLAM_OK;
}

// Device-pointer forms of the two MapPoint updates (everything in HBM, asynchronous on `stream`): the batch-of-sequences driver packs the touched
// points of all sequences into one block.  d_out_desc must be zero-filled by the caller for points without observations (best_idx = -1).
// Resident map-point table (see oslam_hip.h): the results of a MapPoint update are written into the 64-byte records of the points they belong to
__global__ __launch_bounds__(256) void k_mp_table_write(int P, const int32_t* items, uint8_t* const* tab, const int32_t* obs_start, const int32_t* desc_start,
                                                        const float* Pos, const float* out5, const uint8_t* out_desc, int do_desc, int do_normal) {
    const int i = blockIdx.x * 32 + (threadIdx.x >> 3), part = threadIdx.x & 7;
    if (i >= P) return;
    uint32_t* rec = (uint32_t*)(tab[items[2 * i]] + (size_t)items[2 * i + 1] * 64);
    const bool has_obs = obs_start[i + 1] > obs_start[i];   // none (a bad point): both methods return early, only the position the job carries is current
    if (do_normal) {   // words 0-2 position, 3-5 normal, 6 minimum distance, 7 maximum distance
        const uint32_t* ps = (const uint32_t*)(Pos + (size_t)i * 3);
        const uint32_t* o5 = (const uint32_t*)(out5 + (size_t)i * 5);
        if (part < 3) rec[part] = ps[part];   // SetWorldPos happened before this job whatever became of the point (local BA moves a point, then may cull it)
        else if (has_obs) rec[part] = part < 6 ? o5[part - 3] : part == 6 ? o5[4] : o5[3];
    }
    if (!has_obs) return;
    if (do_desc && desc_start[i + 1] > desc_start[i]) rec[8 + part] = ((const uint32_t*)(out_desc + (size_t)i * 32))[part];   // (every observing keyframe bad: it stays)
}

// eight lanes per point: lane part copies word `part` of the descriptor, lanes 0-2 the position
__global__ __launch_bounds__(256) void k_mp_table_gather(int stride, const int32_t* n, const int32_t* ids, uint8_t* const* tab, float* Xw, uint8_t* desc) {
    const int b = blockIdx.y, i = blockIdx.x * 32 + (threadIdx.x >> 3), part = threadIdx.x & 7;
    if (i >= n[b]) return;
    const size_t at = (size_t)b * stride + i;
    const int id = ids[at];
    const uint32_t* rec = id >= 0 ? (const uint32_t*)(tab[b] + (size_t)id * 64) : nullptr;
    ((uint32_t*)desc)[at * 8 + part] = rec ? rec[8 + part] : 0u;
    if (part < 3) ((uint32_t*)Xw)[at * 3 + part] = rec ? rec[part] : 0u;
}

struct FuseQCtx { float fx, fy, cx, cy, bf, minX, minY, maxX, maxY, th, logScale; float scale[OSLAM_MAX_LEVELS]; int nLevels; };
// The projection gates of ORBmatcher::Fuse (src/ORBmatcher.cc:840-890) for one candidate record r = (pos[3], normal[3], minD, maxD, desc); the arithmetic is the
// driver's fuse_queries (the host form the oracle table receives), operator by operator.  Returns false when a gate rejects the candidate.
__device__ __forceinline__ bool fuse_gates(const FuseQCtx& c, const float* r, const float* T, const float* O, float& u, float& v, float& ur, int& lvl) {
    float pc[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float s = T[k * 4] * r[0];
        s += T[k * 4 + 1] * r[1];
        s += T[k * 4 + 2] * r[2];
        pc[k] = (float)((double)s + (double)T[k * 4 + 3]);
    }
    if (pc[2] < 0.0f) return false;
    const float invz = __fdiv_rn(1.0f, pc[2]);
    const float x = pc[0] * invz, y = pc[1] * invz;
    u = c.fx * x + c.cx; v = c.fy * y + c.cy;
    if (!(u >= c.minX && u < c.maxX && v >= c.minY && v < c.maxY)) return false;   // KeyFrame::IsInImage
    ur = u - c.bf * invz;
    const float maxDistance = 1.2f * r[7], minDistance = 0.8f * r[6];
    const float PO[3] = {r[0] - O[0], r[1] - O[1], r[2] - O[2]};
    const float dist3D = (float)norm3d(PO[0], PO[1], PO[2]);
    if (dist3D < minDistance || dist3D > maxDistance) return false;
    const double dot = (double)PO[0] * r[3] + (double)PO[1] * r[4] + (double)PO[2] * r[5];
    if (dot < 0.5 * dist3D) return false;
    const float ratio = __fdiv_rn(r[7], dist3D);
    lvl = (int)ceilf(__fdiv_rn((float)log((double)ratio), c.logScale));   // std::log(float) / logScaleFactor, as Frame::isInFrustum's kernel does
    if (lvl < 0) lvl = 0;
    else if (lvl >= c.nLevels) lvl = c.nLevels - 1;
    return true;
}

// one thread per candidate
__global__ __launch_bounds__(256) void k_fuse_queries(FuseQCtx c, int stride, const int32_t* slots, const int32_t* Mn, const int32_t* ids, const uint8_t* excl,
                                                      uint8_t* const* tab, const float* Tcw, const float* Ow, oslam_proj_query_t* qout) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Mn[b]) return;
    const size_t at = (size_t)b * stride + i;
    oslam_proj_query_t q;
    q.u = q.v = q.ur = q.radius = 0.f; q.minLevel = -1; q.maxLevel = -1; q.flags = 0; q.angle = 0.f;
    uint32_t* qd = (uint32_t*)q.desc;
#pragma unroll
    for (int w = 0; w < 8; w++) qd[w] = 0u;
    const int id = ids[at];
    if (id >= 0 && !excl[at]) {
        const float* r = (const float*)(tab[slots[b]] + (size_t)id * 64);   // pos[3], normal[3], minD, maxD, desc
        float u, v, ur; int lvl;
        if (fuse_gates(c, r, Tcw + (size_t)b * 16, Ow + (size_t)b * 3, u, v, ur, lvl)) {
            q.u = u; q.v = v; q.ur = ur; q.radius = c.th * c.scale[lvl]; q.minLevel = lvl - 1; q.maxLevel = lvl; q.flags = 1;
            const uint32_t* rd = (const uint32_t*)r + 8;
#pragma unroll
            for (int w = 0; w < 8; w++) qd[w] = rd[w];
        }
    }
    qout[at] = q;
}

// ---- Fuse against RESIDENT keyframes: the keyframe's feature grid is built once, when the keyframe is registered ----
// KeyFrame::mGrid never changes after the constructor (src/KeyFrame.cc:44-52 copies Frame::mGrid), but the batched window search rebuilt it in LDS for every
// Fuse job (staging of ~52 KB per keyframe + a counting sort: most of that kernel's 83 us).  k_kf_grid_build sorts a registered keyframe's keypoints by cell
// (cells in the reference's (x, y) nesting, index order inside a cell = the push_back order of Frame::AssignFeaturesToGrid, src/Frame.cc:455-470) into
//   cell_end[kGridCells]  (uint16: end of cell c in the sorted list; start = end of c - 1)
//   cand[N]               (x, y, uRight, octave << 16 | keypoint index) in sorted order
// and k_fuse_search does gates + window search of one candidate per THREAD straight from these arrays (no LDS, no per-job staging).
constexpr int kFGridCols = 64, kFGridRows = 48, kFGridCells = kFGridCols * kFGridRows;   // include/Frame.h:43-44
constexpr int kGridBuildThreads = 1024;

__global__ __launch_bounds__(kGridBuildThreads) void k_kf_grid_build(const oslam_kf_grid_job_t* jobs, const int32_t* cnt, float minX, float minY, float invW, float invH, int ncap,
                                                                      int32_t* status) {
    const oslam_kf_grid_job_t j = jobs[blockIdx.x];
    const int tid = threadIdx.x, N = cnt[j.slot];
    extern __shared__ __align__(16) uint8_t smem[];
    int* s_cell = (int*)smem;                                 // [kFGridCells + 1]
    uint16_t* s_items = (uint16_t*)(s_cell + kFGridCells + 1);   // [ncap]
    __shared__ int s_wtot[kGridBuildThreads / 64];
    if (N < 0 || N > ncap) { if (tid == 0) atomicExch(status, 1); return; }   // host validates; never truncate silently
    for (int i = tid; i <= kFGridCells; i += kGridBuildThreads) s_cell[i] = 0;
    __syncthreads();
    for (int i = tid; i < N; i += kGridBuildThreads) {
        const oslam_keypoint_t kp = j.keys[i];
        const int px = (int)roundf((kp.x - minX) * invW);
        const int py = (int)roundf((kp.y - minY) * invH);
        if (px >= 0 && px < kFGridCols && py >= 0 && py < kFGridRows) atomicAdd(&s_cell[px * kFGridRows + py], 1);
    }
    __syncthreads();
    {   // exclusive scan of the cell counts
        constexpr int kCellsPer = kFGridCells / kGridBuildThreads;
        static_assert(kCellsPer * kGridBuildThreads == kFGridCells, "the cell scan gives every thread the same number of cells");
        const int lane = tid & 63, wv = tid >> 6, base = tid * kCellsPer;
        int cc[kCellsPer], incl = 0;
#pragma unroll
        for (int k = 0; k < kCellsPer; k++) { cc[k] = s_cell[base + k]; incl += cc[k]; }
        const int local = incl;
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
        for (int k = 0; k < kCellsPer; k++) { s_cell[base + k] = start; start += cc[k]; }
    }
    __syncthreads();
    for (int i = tid; i < N; i += kGridBuildThreads) {   // scatter with the cell starts as cursors: afterwards s_cell[c] = end of cell c
        const oslam_keypoint_t kp = j.keys[i];
        const int px = (int)roundf((kp.x - minX) * invW);
        const int py = (int)roundf((kp.y - minY) * invH);
        if (px >= 0 && px < kFGridCols && py >= 0 && py < kFGridRows) s_items[atomicAdd(&s_cell[px * kFGridRows + py], 1)] = (uint16_t)i;
    }
    __syncthreads();
    for (int cell = tid; cell < kFGridCells; cell += kGridBuildThreads) {   // index order inside every cell (insertion sort: cells hold a handful of points)
        const int st = cell > 0 ? s_cell[cell - 1] : 0, en = s_cell[cell];
        for (int a = st + 1; a < en; a++) {
            const uint16_t v = s_items[a];
            int q = a - 1;
            while (q >= st && s_items[q] > v) { s_items[q + 1] = s_items[q]; q--; }
            s_items[q + 1] = v;
        }
        j.cell_end[cell] = (uint16_t)en;
    }
    __syncthreads();
    const int total = s_cell[kFGridCells - 1];
    for (int t = tid; t < total; t += kGridBuildThreads) {
        const int k = s_items[t];
        const oslam_keypoint_t kp = j.keys[k];
        ((float4*)j.cand)[t] = make_float4(kp.x, kp.y, j.uRight ? j.uRight[k] : -1.0f, __uint_as_float(((uint32_t)kp.octave << 16) | (uint32_t)k));
    }
}

struct FuseSearchCtx { FuseQCtx q; float invW, invH; float invSigma2[OSLAM_MAX_LEVELS]; int th_high; };
__global__ __launch_bounds__(256) void k_fuse_search(FuseSearchCtx c, int stride, const oslam_kf_grid_ref_t* kfs, const int32_t* slots, const int32_t* Mn, const int32_t* ids,
                                                     const uint8_t* excl, uint8_t* const* tab, const float* Tcw, const float* Ow, int32_t* q_match) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Mn[b]) return;
    const size_t at = (size_t)b * stride + i;
    int bestIdx = -1;
    const int id = ids[at];
    if (id >= 0 && !excl[at]) {
        const float* r = (const float*)(tab[slots[b]] + (size_t)id * 64);
        float x, y, qur; int lvl;
        if (fuse_gates(c.q, r, Tcw + (size_t)b * 16, Ow + (size_t)b * 3, x, y, qur, lvl)) {
            const float rad = c.q.th * c.q.scale[lvl];
            const int minLevel = lvl - 1, maxLevel = lvl;
            // KeyFrame::GetFeaturesInArea (src/KeyFrame.cc:418-455; the window arithmetic of the batched search, matcher.hip)
            const int nMinCellX = max(0, (int)floorf((x - c.q.minX - rad) * c.invW));
            const int nMaxCellX = min(kFGridCols - 1, (int)ceilf((x - c.q.minX + rad) * c.invW));
            const int nMinCellY = max(0, (int)floorf((y - c.q.minY - rad) * c.invH));
            const int nMaxCellY = min(kFGridRows - 1, (int)ceilf((y - c.q.minY + rad) * c.invH));
            if (nMinCellX < kFGridCols && nMaxCellX >= 0 && nMinCellY < kFGridRows && nMaxCellY >= 0) {
                const oslam_kf_grid_ref_t kf = kfs[b];
                const uint4 qd0 = ((const uint4*)r)[2], qd1 = ((const uint4*)r)[3];
                int bestDist = 256;
                for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
                    const int c0 = ix * kFGridRows + nMinCellY;
                    const int s = c0 > 0 ? kf.cell_end[c0 - 1] : 0, e = kf.cell_end[ix * kFGridRows + nMaxCellY];
                    for (int t = s; t < e; t++) {   // cells iy = min..max are contiguous in this layout
                        const float4 cd = ((const float4*)kf.cand)[t];
                        if (!(fabsf(cd.x - x) < rad && fabsf(cd.y - y) < rad)) continue;
                        const uint32_t ok = __float_as_uint(cd.w);
                        const int oct = (int)(ok >> 16), k = (int)(ok & 0xFFFFu);
                        if (oct < minLevel || oct > maxLevel) continue;   // :903-906
                        const float ex = x - cd.x, ey = y - cd.y;
                        if (cd.z >= 0) {   // chi2 gates :908-934
                            const float er = qur - cd.z;
                            const float e2 = ex * ex + ey * ey + er * er;
                            if ((double)(e2 * c.invSigma2[oct]) > 7.8) continue;
                        } else {
                            const float e2 = ex * ex + ey * ey;
                            if ((double)(e2 * c.invSigma2[oct]) > 5.99) continue;
                        }
                        const uint4* d = (const uint4*)(kf.desc + (size_t)k * 32);
                        const uint4 d0 = d[0], d1 = d[1];
                        const int dist = __popc(qd0.x ^ d0.x) + __popc(qd0.y ^ d0.y) + __popc(qd0.z ^ d0.z) + __popc(qd0.w ^ d0.w) + __popc(qd1.x ^ d1.x) + __popc(qd1.y ^ d1.y) +
                                         __popc(qd1.z ^ d1.z) + __popc(qd1.w ^ d1.w);
                        if (dist < bestDist) { bestDist = dist; bestIdx = k; }
                    }
                }
                if (bestDist > c.th_high) bestIdx = -1;
            }
        }
    }
    q_match[at] = bestIdx;
}

int oslam_kf_grid_build_device(int n, const oslam_kf_grid_job_t* d_jobs, const int32_t* d_counts, const float bounds[4], int max_keypoints, int32_t* d_status, void* stream) {
    if (n < 1 || !d_jobs || !d_counts || !bounds || max_keypoints < 1 || max_keypoints > 65535 || !d_status) { set_error("kf_grid_build: bad argument"); return OSLAM_E_INVALID; }
    const size_t lds = (size_t)(kFGridCells + 1) * 4 + (size_t)max_keypoints * 2;
    {   // once per DEVICE (the attribute belongs to the function on the current device) and safe against the handles' threads calling concurrently
        static std::once_flag attr_once[64];
        static std::atomic<int> attr_rc[64];
        int dev = 0;
        OSLAM_HIP_CHECK(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64) { set_error("device index %d outside [0,64)", dev); return OSLAM_E_INVALID; }
        std::call_once(attr_once[dev], [dev] { attr_rc[dev] = (int)hipFuncSetAttribute((const void*)k_kf_grid_build, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048); });
        if (attr_rc[dev] != (int)hipSuccess) { set_error("hipFuncSetAttribute(k_kf_grid_build): %s", hipGetErrorString((hipError_t)attr_rc[dev].load())); return OSLAM_E_HIP; }
    }
    if (lds > 160 * 1024 - 2048) { set_error("kf_grid_build: max_keypoints too large"); return OSLAM_E_CAPACITY; }
    const float invW = (float)kFGridCols / (float)(bounds[2] - bounds[0]), invH = (float)kFGridRows / (float)(bounds[3] - bounds[1]);   // src/Frame.cc:160-161
    hipLaunchKernelGGL(k_kf_grid_build, dim3(n), dim3(kGridBuildThreads), lds, (hipStream_t)stream, d_jobs, d_counts, bounds[0], bounds[1], invW, invH, max_keypoints, d_status);
    OSLAM_HIP_CHECK(hipGetLastError());
    return OSLAM_OK;
}

int oslam_fuse_search_device(int n, int stride, const oslam_kf_grid_ref_t* d_kfs, const int32_t* d_slots, const int32_t* d_M, const int32_t* d_ids, const uint8_t* d_excl,
                             uint8_t* const* d_tab, const float* d_Tcw, const float* d_Ow, const float K5[5], const float bounds[4], float th, float logScaleFactor,
                             const float* scaleFactors, const float* invLevelSigma2, int nLevels, int32_t* d_q_match, void* stream) {
    if (n < 1 || stride < 1 || !d_kfs || !d_slots || !d_M || !d_ids || !d_excl || !d_tab || !d_Tcw || !d_Ow || !K5 || !bounds || !scaleFactors || !invLevelSigma2 || nLevels < 1 ||
        nLevels > OSLAM_MAX_LEVELS || !d_q_match) {
        set_error("fuse_search: bad argument"); return OSLAM_E_INVALID;
    }
    FuseSearchCtx c;
    c.q.fx = K5[0]; c.q.fy = K5[1]; c.q.cx = K5[2]; c.q.cy = K5[3]; c.q.bf = K5[4];
    c.q.minX = bounds[0]; c.q.minY = bounds[1]; c.q.maxX = bounds[2]; c.q.maxY = bounds[3]; c.q.th = th; c.q.logScale = logScaleFactor; c.q.nLevels = nLevels;
    for (int l = 0; l < OSLAM_MAX_LEVELS; l++) { c.q.scale[l] = l < nLevels ? scaleFactors[l] : 1.f; c.invSigma2[l] = l < nLevels ? invLevelSigma2[l] : 0.f; }
    c.invW = (float)kFGridCols / (float)(bounds[2] - bounds[0]); c.invH = (float)kFGridRows / (float)(bounds[3] - bounds[1]);
    c.th_high = 50;   // ORBmatcher::TH_LOW (src/ORBmatcher.cc:37, :936)
    hipLaunchKernelGGL(k_fuse_search, dim3(div_up(stride, 256), n), dim3(256), 0, (hipStream_t)stream, c, stride, d_kfs, d_slots, d_M, d_ids, d_excl, d_tab, d_Tcw, d_Ow, d_q_match);
    OSLAM_HIP_CHECK(hipGetLastError());
    return OSLAM_OK;
}

int oslam_fuse_queries_device(int n, int stride, const int32_t* d_slots, const int32_t* d_M, const int32_t* d_ids, const uint8_t* d_excl, uint8_t* const* d_tab,
                              const float* d_Tcw, const float* d_Ow, const float K5[5], const float bounds[4], float th, float logScaleFactor,
                              const float* scaleFactors, int nLevels, oslam_proj_query_t* d_q, void* stream) {
    if (n < 1 || stride < 1 || !d_slots || !d_M || !d_ids || !d_excl || !d_tab || !d_Tcw || !d_Ow || !K5 || !bounds || !scaleFactors || nLevels < 1 || nLevels > OSLAM_MAX_LEVELS || !d_q) {
        set_error("fuse_queries: bad argument"); return OSLAM_E_INVALID;
    }
    FuseQCtx c;
    c.fx = K5[0]; c.fy = K5[1]; c.cx = K5[2]; c.cy = K5[3]; c.bf = K5[4];
    c.minX = bounds[0]; c.minY = bounds[1]; c.maxX = bounds[2]; c.maxY = bounds[3]; c.th = th; c.logScale = logScaleFactor; c.nLevels = nLevels;
    for (int l = 0; l < OSLAM_MAX_LEVELS; l++) c.scale[l] = l < nLevels ? scaleFactors[l] : 1.f;
    hipLaunchKernelGGL(k_fuse_queries, dim3(div_up(stride, 256), n), dim3(256), 0, (hipStream_t)stream, c, stride, d_slots, d_M, d_ids, d_excl, d_tab, d_Tcw, d_Ow, d_q);
    OSLAM_HIP_CHECK(hipGetLastError());
    return OSLAM_OK;
}

// eight lanes per local point: lane `part` moves word `part` of the record's first half (position, normal, distances) and of its descriptor
__global__ __launch_bounds__(256) void k_local_gather(const oslam_local_gather_t* jobs, const uint8_t* stage, uint8_t* const* tab, int stride, float* Pw, float* Pn,
                                                      float* maxD, float* minD, uint8_t* obs, uint8_t* desc) {
    const oslam_local_gather_t j = jobs[blockIdx.y];
    const int q = blockIdx.x * 32 + (threadIdx.x >> 3), part = threadIdx.x & 7;
    if (q >= j.M) return;
    const int id = ((const int32_t*)(stage + j.ids_off))[q];
    const uint32_t* rec = (const uint32_t*)(tab[j.slot] + (size_t)id * 64);
    const size_t at = (size_t)j.slot * stride + q;
    const uint32_t w = rec[part];
    if (part < 3) ((uint32_t*)Pw)[at * 3 + part] = w;
    else if (part < 6) ((uint32_t*)Pn)[at * 3 + part - 3] = w;
    else if (part == 6) ((uint32_t*)minD)[at] = w;
    else ((uint32_t*)maxD)[at] = w;
    ((uint32_t*)desc)[at * 8 + part] = rec[8 + part];
    if (part == 0) obs[at] = (stage + j.obs_off)[q];
}

int oslam_mp_table_local_gather_device(int n, int maxM, const oslam_local_gather_t* d_jobs, const uint8_t* d_stage, uint8_t* const* d_tab, int stride, float* d_Pw,
                                       float* d_Pn, float* d_maxDist, float* d_minDist, uint8_t* d_obs_gt0, uint8_t* d_mp_desc, void* stream) {
    if (n < 0 || maxM < 0 || (n > 0 && maxM > 0 && (!d_jobs || !d_stage || !d_tab || stride < maxM || !d_Pw || !d_Pn || !d_maxDist || !d_minDist || !d_obs_gt0 || !d_mp_desc))) {
        set_error("mp_table_local_gather: bad argument"); return OSLAM_E_INVALID;
    }
    if (n == 0 || maxM == 0) return OSLAM_OK;
    hipLaunchKernelGGL(k_local_gather, dim3(div_up(maxM, 32), n), dim3(256), 0, (hipStream_t)stream, d_jobs, d_stage, d_tab, stride, d_Pw, d_Pn, d_maxDist, d_minDist,
                       d_obs_gt0, d_mp_desc);
    OSLAM_HIP_CHECK(hipGetLastError());
    return OSLAM_OK;
}

__global__ __launch_bounds__(256) void k_mp_table_positions(int n, const int32_t* slots, const int32_t* ids, uint8_t* const* tab, float* Xw) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* r = (const float*)(tab[slots[i]] + (size_t)ids[i] * 64);
    Xw[(size_t)i * 3] = r[0]; Xw[(size_t)i * 3 + 1] = r[1]; Xw[(size_t)i * 3 + 2] = r[2];
}

int oslam_mp_table_positions_device(int n, const int32_t* d_slots, const int32_t* d_ids, uint8_t* const* d_tab, float* d_Xw, void* stream) {
    if (n < 0 || (n > 0 && (!d_slots || !d_ids || !d_tab || !d_Xw))) { set_error("mp_table_positions: bad argument"); return OSLAM_E_INVALID; }
    if (n == 0) return OSLAM_OK;
    hipLaunchKernelGGL(k_mp_table_positions, dim3(div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, n, d_slots, d_ids, d_tab, d_Xw);
    OSLAM_HIP_CHECK(hipGetLastError());
    return OSLAM_OK;
}

int oslam_mp_table_gather_device(int batch, int stride, const int32_t* d_n, const int32_t* d_ids, uint8_t* const* d_tab, float* d_Xw, uint8_t* d_desc, void* stream) {
    if (batch < 1 || stride < 1 || !d_n || !d_ids || !d_tab || !d_Xw || !d_desc) { set_error("mp_table_gather: bad argument"); return OSLAM_E_INVALID; }
    hipLaunchKernelGGL(k_mp_table_gather, dim3(div_up(stride, 32), batch), dim3(256), 0, (hipStream_t)stream, stride, d_n, d_ids, d_tab, d_Xw, d_desc);
    OSLAM_HIP_CHECK(hipGetLastError());
    return OSLAM_OK;
}

struct InvSigma { float v[OSLAM_MAX_LEVELS]; };
__global__ __launch_bounds__(256) void k_pose_inputs(int stride, const int32_t* slots, const int32_t* n, const int32_t* ids, uint8_t* const* tab, const oslam_keypoint_t* keysUn,
                                                     const float* uRight, int kp_stride, InvSigma inv, float* Xw, float* obs, float* invS, uint8_t* has) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n[b]) return;
    const int slot = slots[b];
    const size_t at = (size_t)b * stride + i, src = (size_t)slot * kp_stride + i;
    const oslam_keypoint_t kp = keysUn[src];
    obs[at * 3] = kp.x; obs[at * 3 + 1] = kp.y; obs[at * 3 + 2] = uRight[src];
    invS[at] = inv.v[kp.octave];
    const int id = ids[at];
    float x = 0.f, y = 0.f, z = 0.f;
    if (id >= 0) { const float* r = (const float*)(tab[slot] + (size_t)id * 64); x = r[0]; y = r[1]; z = r[2]; }
    Xw[at * 3] = x; Xw[at * 3 + 1] = y; Xw[at * 3 + 2] = z;
    has[at] = id >= 0;
}

int oslam_pose_inputs_gather_device(int batch, int stride, const int32_t* d_slots, const int32_t* d_n, const int32_t* d_ids, uint8_t* const* d_tab,
                                    const oslam_keypoint_t* d_keysUn, const float* d_uRight, int kp_stride, const float* invLevelSigma2, int nLevels,
                                    float* d_Xw, float* d_obs, float* d_invSigma2, uint8_t* d_has_mp, void* stream) {
    if (batch < 1 || stride < 1 || kp_stride < 1 || !d_slots || !d_n || !d_ids || !d_tab || !d_keysUn || !d_uRight || !invLevelSigma2 || nLevels < 1 || nLevels > OSLAM_MAX_LEVELS ||
        !d_Xw || !d_obs || !d_invSigma2 || !d_has_mp) { set_error("pose_inputs_gather: bad argument"); return OSLAM_E_INVALID; }
    InvSigma inv;
    for (int l = 0; l < OSLAM_MAX_LEVELS; l++) inv.v[l] = l < nLevels ? invLevelSigma2[l] : 0.f;
    hipLaunchKernelGGL(k_pose_inputs, dim3(div_up(stride, 256), batch), dim3(256), 0, (hipStream_t)stream, stride, d_slots, d_n, d_ids, d_tab, d_keysUn, d_uRight, kp_stride, inv,
                       d_Xw, d_obs, d_invSigma2, d_has_mp);
    OSLAM_HIP_CHECK(hipGetLastError());
    return OSLAM_OK;
}

int oslam_mp_table_write_device(int P, const int32_t* d_items, uint8_t* const* d_tab, const int32_t* d_obs_start, const int32_t* d_desc_start, const float* d_Pos,
                                const float* d_out5, const uint8_t* d_out_desc, int do_desc, int do_normal, void* stream) {
    if (P < 0 || (P > 0 && (!d_items || !d_tab || !d_obs_start || !d_desc_start || (do_normal && (!d_Pos || !d_out5)) || (do_desc && !d_out_desc)))) { set_error("mp_table_write: bad argument"); return OSLAM_E_INVALID; }
    if (P == 0) return OSLAM_OK;
    hipLaunchKernelGGL(k_mp_table_write, dim3(div_up(P, 32)), dim3(256), 0, (hipStream_t)stream, P, d_items, d_tab, d_obs_start, d_desc_start, d_Pos, d_out5, d_out_desc,
                       do_desc, do_normal);
    OSLAM_HIP_CHECK(hipGetLastError());
    return OSLAM_OK;
}

// MapPoint::UpdateNormalAndDepth for the points of solved local-BA windows (include/oslam_slam.h oslam_job_mp_window_t), one thread per point: the observations
// are the point's window edges that the write-back did not erase, the camera centres come from the windows' own keyframe tables.  Same float arithmetic, in
// the same order, as k_update_normal_depth over the packed observation list; the resident record (position always, normal / distances unless skipped) is written
// in the same pass (k_mp_table_write's layout).
__global__ __launch_bounds__(256) void k_mp_windows(int P, const int32_t* items, uint8_t* const* tab, const int32_t* e0, const int32_t* ne, const int32_t* kbase, const int32_t* ref,
                                                    const float* lsf, const uint8_t* skip, const float* Pos, const int32_t* edge_kf, const uint8_t* erase, const float* Ow,
                                                    float lastScale, float* out5) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const float px = Pos[p * 3], py = Pos[p * 3 + 1], pz = Pos[p * 3 + 2];
    float* rec = tab ? (float*)(tab[items[2 * p]] + (size_t)items[2 * p + 1] * 64) : nullptr;
    if (rec) { rec[0] = px; rec[1] = py; rec[2] = pz; }   // SetWorldPos happened whatever became of the point
    float* o = out5 + (size_t)p * 5;
    if (skip[p]) { for (int k = 0; k < 5; k++) o[k] = 0.f; return; }
    const int s = e0[p], n = ne[p], kb = kbase[p];
    float nx = 0, ny = 0, nz = 0;
    int cnt = 0;
    for (int i = 0; i < n; i++) {
        if (erase[s + i]) continue;
        const float* c = Ow + (size_t)(kb + edge_kf[s + i]) * 3;
        const float ax = px - c[0], ay = py - c[1], az = pz - c[2];
        const float inv = (float)(1.0 / norm3d(ax, ay, az));   // cv::scaleAdd with float alpha
        nx = ax * inv + nx;
        ny = ay * inv + ny;
        nz = az * inv + nz;
        cnt++;
    }
    if (cnt == 0) { for (int k = 0; k < 5; k++) o[k] = 0.f; return; }
    const float* cr = Ow + (size_t)(kb + ref[p]) * 3;
    const float dist = (float)norm3d(px - cr[0], py - cr[1], pz - cr[2]);
    const float maxD = dist * lsf[p];
    const float invn = (float)(1.0 / (double)cnt);   // convertTo(alpha = 1/n): float scale
    const float r0 = nx * invn + 0.0f, r1 = ny * invn + 0.0f, r2 = nz * invn + 0.0f, minD = __fdiv_rn(maxD, lastScale);
    o[0] = r0; o[1] = r1; o[2] = r2; o[3] = maxD; o[4] = minD;
    if (rec) { rec[3] = r0; rec[4] = r1; rec[5] = r2; rec[6] = minD; rec[7] = maxD; }
}

int oslam_mp_update_windows_device(int P, const int32_t* d_items, uint8_t* const* d_tab, const int32_t* d_e0, const int32_t* d_ne, const int32_t* d_kbase, const int32_t* d_ref,
                                   const float* d_lsf, const uint8_t* d_skip, const float* d_Pos, const int32_t* d_edge_kf, const uint8_t* d_erase, const float* d_Ow, float lastScale,
                                   float* d_out5, void* stream) {
    if (P < 0 || (P > 0 && (!d_e0 || !d_ne || !d_kbase || !d_ref || !d_lsf || !d_skip || !d_Pos || !d_edge_kf || !d_erase || !d_Ow || !d_out5 || (d_tab && !d_items)))) {
        set_error("mp_update_windows: bad argument"); return OSLAM_E_INVALID;
    }
    if (P == 0) return OSLAM_OK;
    hipLaunchKernelGGL(k_mp_windows, dim3(div_up(P, 256)), dim3(256), 0, (hipStream_t)stream, P, d_items, d_tab, d_e0, d_ne, d_kbase, d_ref, d_lsf, d_skip, d_Pos, d_edge_kf, d_erase,
                       d_Ow, lastScale, d_out5);
    OSLAM_HIP_CHECK(hipGetLastError());
    return OSLAM_OK;
}

// The driver's small MapPoint updates (new points, the descriptor updates after every Fuse round: tens to hundreds of points per call, dozens of calls per
// keyframe) in ONE launch instead of up to eight (descriptor gather, memset, k_distinctive, k_update_normal_depth, k_mp_table_write, three copy kernels): one
// wavefront per point reads its observations' descriptors straight from the resident keyframe records (rec = (record, keypoint) per observation of the
// descriptor list), runs k_distinctive's selection and k_update_normal_depth's sums (lane 0), and writes the results to the caller's device-accessible host
// block AND into the point's resident record.  Same arithmetic, same order as the separate kernels.  Points with more than kDdMaxObs observations: not here
// (the caller checks).
struct MpFused {
    int P, do_desc, do_normal;
    const int32_t* obs_start; const int32_t* desc_start; const int32_t* rec; const uint8_t* const* rec_desc;
    const float* obs_Ow; const float* Pos; const float* OwRef; const float* lsf; float lastScale;
    const int32_t* items; uint8_t* const* tab;
    int32_t* o_best; uint8_t* o_desc; float* o_out5;
};
__global__ __launch_bounds__(256) void k_mp_update_fused(MpFused c) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int p = blockIdx.x * 4 + wv;
    __shared__ uint32_t s_d[4][kDdMaxObs * 8];
    if (p >= c.P) return;
    const bool has_obs = c.obs_start[p + 1] > c.obs_start[p];
    uint32_t* recw = c.tab ? (uint32_t*)(c.tab[c.items[2 * p]] + (size_t)c.items[2 * p + 1] * 64) : nullptr;
    if (c.do_desc) {
        const int s = c.desc_start[p], N = c.desc_start[p + 1] - s;
        uint32_t* od = (uint32_t*)(c.o_desc + (size_t)p * 32);
        if (N <= 0 || N > kDdMaxObs) {   // (more observations than the staging slice holds: reported as best = -2, the point's descriptor is left to the caller)
            if (lane == 0) c.o_best[p] = N <= 0 ? -1 : -2;
            if (lane < 8) od[lane] = 0u;
        } else {
            uint32_t* D = s_d[wv];
            for (int i = lane; i < N * 8; i += 64) {
                const int o = i >> 3, w = i & 7;
                D[i] = ((const uint32_t*)(c.rec_desc[c.rec[2 * (s + o)]] + (size_t)c.rec[2 * (s + o) + 1] * 32))[w];
            }
            __builtin_amdgcn_wave_barrier();
            const int k = (int)(0.5 * (N - 1));   // vDists[0.5*(N-1)]
            int bestMedian = 0x7fffffff, bestI = 0;
            for (int i = lane; i < N; i += 64) {
                uint32_t qi[8];
#pragma unroll
                for (int w = 0; w < 8; w++) qi[w] = D[i * 8 + w];
                int lo = 0, hi = 256;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    int cnt = 0;
                    for (int j = 0; j < N; j++) {
                        int dist = 0;
#pragma unroll
                        for (int w = 0; w < 8; w++) dist += __popc(qi[w] ^ D[j * 8 + w]);
                        cnt += dist <= mid;
                    }
                    if (cnt >= k + 1) hi = mid; else lo = mid + 1;
                }
                if (lo < bestMedian) { bestMedian = lo; bestI = i; }
            }
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) {
                const int om = __shfl_xor(bestMedian, sft, 64), oi = __shfl_xor(bestI, sft, 64);
                if (om < bestMedian || (om == bestMedian && oi < bestI)) { bestMedian = om; bestI = oi; }
            }
            if (lane == 0) c.o_best[p] = bestI;
            if (lane < 8) {
                const uint32_t v = D[bestI * 8 + lane];
                od[lane] = v;
                if (recw && has_obs) recw[8 + lane] = v;   // (every observing keyframe bad: the record's descriptor stays, k_mp_table_write)
            }
        }
    }
    if (c.do_normal && lane == 0) {
        const float px = c.Pos[p * 3], py = c.Pos[p * 3 + 1], pz = c.Pos[p * 3 + 2];
        float* o = c.o_out5 + (size_t)p * 5;
        float* rf = (float*)recw;
        if (rf) { rf[0] = px; rf[1] = py; rf[2] = pz; }
        const int s = c.obs_start[p], n = c.obs_start[p + 1] - s;
        if (n <= 0) { for (int k = 0; k < 5; k++) o[k] = 0.f; }
        else {
            float nx = 0, ny = 0, nz = 0;
            for (int i = 0; i < n; i++) {
                const float ax = px - c.obs_Ow[(s + i) * 3], ay = py - c.obs_Ow[(s + i) * 3 + 1], az = pz - c.obs_Ow[(s + i) * 3 + 2];
                const float inv = (float)(1.0 / norm3d(ax, ay, az));   // cv::scaleAdd with float alpha
                nx = ax * inv + nx;
                ny = ay * inv + ny;
                nz = az * inv + nz;
            }
            const float dist = (float)norm3d(px - c.OwRef[p * 3], py - c.OwRef[p * 3 + 1], pz - c.OwRef[p * 3 + 2]);
            const float maxD = dist * c.lsf[p];
            const float invn = (float)(1.0 / (double)n);   // convertTo(alpha = 1/n): float scale
            const float r0 = nx * invn + 0.0f, r1 = ny * invn + 0.0f, r2 = nz * invn + 0.0f, minD = __fdiv_rn(maxD, c.lastScale);
            o[0] = r0; o[1] = r1; o[2] = r2; o[3] = maxD; o[4] = minD;
            if (rf) { rf[3] = r0; rf[4] = r1; rf[5] = r2; rf[6] = minD; rf[7] = maxD; }
        }
    }
}

int oslam_mp_update_fused_device(int P, int do_desc, int do_normal, const int32_t* d_obs_start, const int32_t* d_desc_start, const int32_t* d_rec, const uint8_t* const* d_rec_desc,
                                 const float* d_obs_Ow, const float* d_Pos, const float* d_OwRef, const float* d_lsf, float lastScale, const int32_t* d_items, uint8_t* const* d_tab,
                                 int32_t* o_best, uint8_t* o_desc, float* o_out5, void* stream) {
    if (P < 0 || (P > 0 && (!d_obs_start || !d_desc_start || (do_desc && (!d_rec || !d_rec_desc || !o_best || !o_desc)) ||
                            (do_normal && (!d_obs_Ow || !d_Pos || !d_OwRef || !d_lsf || !o_out5)) || (d_tab && !d_items)))) { set_error("mp_update_fused: bad argument"); return OSLAM_E_INVALID; }
    if (P == 0) return OSLAM_OK;
    MpFused c;
    c.P = P; c.do_desc = do_desc; c.do_normal = do_normal; c.obs_start = d_obs_start; c.desc_start = d_desc_start; c.rec = d_rec; c.rec_desc = d_rec_desc;
    c.obs_Ow = d_obs_Ow; c.Pos = d_Pos; c.OwRef = d_OwRef; c.lsf = d_lsf; c.lastScale = lastScale; c.items = d_items; c.tab = d_tab;
    c.o_best = o_best; c.o_desc = o_desc; c.o_out5 = o_out5;
    hipLaunchKernelGGL(k_mp_update_fused, dim3(div_up(P, 4)), dim3(256), 0, (hipStream_t)stream, c);
    OSLAM_HIP_CHECK(hipGetLastError());
    return OSLAM_OK;
}

int oslam_mp_distinctive_descriptors_device(int P, const int32_t* d_obs_start, const uint8_t* d_obs_desc, int32_t* d_best_idx, uint8_t* d_out_desc, void* stream) {
    if (P < 0 || (P > 0 && (!d_obs_start || !d_obs_desc || !d_best_idx || !d_out_desc))) { set_error("bad argument"); return OSLAM_E_INVALID; }
    if (P == 0) return OSLAM_OK;
    hipLaunchKernelGGL(k_distinctive, dim3(div_up(P, 4)), dim3(256), 0, (hipStream_t)stream, P, d_obs_start, d_obs_desc, d_best_idx, d_out_desc);
    OSLAM_HIP_CHECK(hipGetLastError());
    return OSLAM_OK;
}
int oslam_mp_update_normal_depth_device(int P, const float* d_Pos, const int32_t* d_obs_start, const float* d_obs_Ow, const float* d_OwRef,
                                        const float* d_levelScaleFactor, float lastScaleFactor, float* d_out, void* stream) {
    if (P < 0 || (P > 0 && (!d_Pos || !d_obs_start || !d_obs_Ow || !d_OwRef || !d_levelScaleFactor || !d_out))) { set_error("bad argument"); return OSLAM_E_INVALID; }
    if (P == 0) return OSLAM_OK;
    hipLaunchKernelGGL(k_update_normal_depth, dim3(div_up(P, 256)), dim3(256), 0, (hipStream_t)stream, P, d_Pos, d_obs_start, d_obs_Ow, d_OwRef, d_levelScaleFactor,
                       lastScaleFactor, d_out);
    OSLAM_HIP_CHECK(hipGetLastError());
    return OSLAM_OK;
}

int oslam_mp_update_normal_depth(oslam_mappoint_t* h, int P, const float* Pos, const int32_t* obs_start, const float* obs_Ow, const float* OwRef,
                                 const float* levelScaleFactor, float lastScaleFactor, float* out) {
    if (!h || P < 0 || (P > 0 && (!Pos || !obs_start || !OwRef || !levelScaleFactor || !out))) { set_error("bad argument"); return OSLAM_E_INVALID; }
    if (P == 0) return OSLAM_OK;
    const int total = obs_start[P];
    if (total < 0 || (total > 0 && !obs_Ow)) { set_error("bad observation table"); return OSLAM_E_INVALID; }
    OSLAM_HIP_CHECK(hipSetDevice(h->device));
    int rc;
    if ((rc = mp_up(h->a, obs_start, (size_t)(P + 1) * 4)) || (rc = mp_up(h->b, obs_Ow, (size_t)total * 12)) || (rc = mp_up(h->c, Pos, (size_t)P * 12)) ||
        (rc = mp_up(h->d, OwRef, (size_t)P * 12)) || (rc = mp_up(h->e, levelScaleFactor, (size_t)P * 4)) || (rc = mp_ensure(h->o1, (size_t)P * 20)))
        return rc;
    hipLaunchKernelGGL(k_update_normal_depth, dim3(div_up(P, 256)), dim3(256), 0, nullptr, P, (const float*)h->c.p, (const int*)h->a.p, (const float*)h->b.p,
                       (const float*)h->d.p, (const float*)h->e.p, lastScaleFactor, (float*)h->o1.p);
    OSLAM_HIP_CHECK(hipGetLastError());
    OSLAM_HIP_CHECK(hipMemcpy(out, h->o1.p, (size_t)P * 20, hipMemcpyDeviceToHost));
    return OSLAM_OK;
}

static void fill_frustum(FrustumCtx& c, int M, const float Tcw[16], const float K5[5], const float bounds[4], float viewingCosLimit,
                         float logScaleFactor, const float* scaleFactors, int nLevels, float th) {
    c.M = M;
    for (int i = 0; i < 16; i++) c.T[i] = Tcw[i];
    c.fx = K5[0]; c.fy = K5[1]; c.cx = K5[2]; c.cy = K5[3]; c.bf = K5[4];
    c.minX = bounds[0]; c.minY = bounds[1]; c.maxX = bounds[2]; c.maxY = bounds[3];
    c.cosLimit = viewingCosLimit; c.logScale = logScaleFactor; c.th = th; c.nLevels = nLevels;
    for (int i = 0; i < OSLAM_MAX_LEVELS; i++) c.scale[i] = i < nLevels ? scaleFactors[i] : 0.f;
}

int oslam_frame_is_in_frustum_device(int M, const float* d_Pw, const float* d_Pn, const float* d_maxDist, const float* d_minDist, const uint8_t* d_obs_gt0,
                                     const uint8_t* d_mp_desc, const float Tcw[16], const float K5[5], const float bounds[4], float viewingCosLimit,
                                     float logScaleFactor, const float* scaleFactors, int nLevels, float th, oslam_proj_query_t* d_out, void* stream) {
    if (M < 0 || !Tcw || !K5 || !bounds || !scaleFactors || nLevels < 1 || nLevels > OSLAM_MAX_LEVELS || (M > 0 && (!d_Pw || !d_Pn || !d_maxDist || !d_minDist || !d_obs_gt0 || !d_mp_desc || !d_out))) {
        set_error("bad argument");
        return OSLAM_E_INVALID;
    }
    if (M == 0) return OSLAM_OK;
    FrustumCtx c;
    fill_frustum(c, M, Tcw, K5, bounds, viewingCosLimit, logScaleFactor, scaleFactors, nLevels, th);
    c.Pw = d_Pw; c.Pn = d_Pn; c.maxDist = d_maxDist; c.minDist = d_minDist; c.obs_gt0 = d_obs_gt0; c.mp_desc = d_mp_desc; c.out = d_out;
    hipLaunchKernelGGL(k_is_in_frustum, dim3(div_up(M, 256)), dim3(256), 0, (hipStream_t)stream, c);
    OSLAM_HIP_CHECK(hipGetLastError());
    return OSLAM_OK;
}

int oslam_frame_is_in_frustum_batch_resident_device(int batch, int stride_in, int stride_out, const int32_t* d_M, const float* d_Pw, const float* d_Pn,
                                                    const float* d_maxDist, const float* d_minDist, const uint8_t* d_obs_gt0, const uint8_t* d_mp_desc,
                                                    const uint8_t* d_skip, const float* d_Tcw, const float* d_th, const float K5[5], const float bounds[4],
                                                    float viewingCosLimit, float logScaleFactor, const float* scaleFactors, int nLevels,
                                                    oslam_proj_query_t* d_out, uint8_t* d_in_view, void* stream) {
    if (batch < 0 || stride_in < 0 || stride_out < 0 || stride_out > stride_in || !K5 || !bounds || !scaleFactors || nLevels < 1 || nLevels > OSLAM_MAX_LEVELS ||
        (batch > 0 && stride_out > 0 && (!d_M || !d_Pw || !d_Pn || !d_maxDist || !d_minDist || !d_obs_gt0 || !d_mp_desc || !d_Tcw || !d_th || !d_out))) {
        set_error("bad argument");
        return OSLAM_E_INVALID;
    }
    if (batch == 0 || stride_out == 0) return OSLAM_OK;
    FrustumCtx c;
    const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    fill_frustum(c, 0, I, K5, bounds, viewingCosLimit, logScaleFactor, scaleFactors, nLevels, 1.0f);
    c.Pw = d_Pw; c.Pn = d_Pn; c.maxDist = d_maxDist; c.minDist = d_minDist; c.obs_gt0 = d_obs_gt0; c.mp_desc = d_mp_desc; c.out = d_out;
    hipLaunchKernelGGL(k_is_in_frustum_batch, dim3(div_up(stride_out, 256), batch), dim3(256), 0, (hipStream_t)stream, c, stride_in, stride_out, d_M, d_Tcw, d_th,
                       d_skip, d_in_view);
    OSLAM_HIP_CHECK(hipGetLastError());
    return OSLAM_OK;
}

int oslam_frame_is_in_frustum_batch_device(int batch, int stride, const int32_t* d_M, const float* d_Pw, const float* d_Pn, const float* d_maxDist,
                                           const float* d_minDist, const uint8_t* d_obs_gt0, const uint8_t* d_mp_desc, const float* d_Tcw, const float* d_th,
                                           const float K5[5], const float bounds[4], float viewingCosLimit, float logScaleFactor, const float* scaleFactors,
                                           int nLevels, oslam_proj_query_t* d_out, uint8_t* d_in_view, void* stream) {
    return oslam_frame_is_in_frustum_batch_resident_device(batch, stride, stride, d_M, d_Pw, d_Pn, d_maxDist, d_minDist, d_obs_gt0, d_mp_desc, nullptr, d_Tcw, d_th, K5,
                                                           bounds, viewingCosLimit, logScaleFactor, scaleFactors, nLevels, d_out, d_in_view, stream);
}

int oslam_frame_is_in_frustum(oslam_mappoint_t* h, int M, const float* Pw, const float* Pn, const float* maxDist, const float* minDist, const uint8_t* obs_gt0,
                              const uint8_t* mp_desc, const float Tcw[16], const float K5[5], const float bounds[4], float viewingCosLimit,
                              float logScaleFactor, const float* scaleFactors, int nLevels, float th, oslam_proj_query_t* out) {
    if (!h || M < 0 || (M > 0 && (!Pw || !Pn || !maxDist || !minDist || !obs_gt0 || !mp_desc || !out))) { set_error("bad argument"); return OSLAM_E_INVALID; }
    if (M == 0) return OSLAM_OK;
    OSLAM_HIP_CHECK(hipSetDevice(h->device));
    int rc;
    if ((rc = mp_up(h->a, Pw, (size_t)M * 12)) || (rc = mp_up(h->b, Pn, (size_t)M * 12)) || (rc = mp_up(h->c, maxDist, (size_t)M * 4)) ||
        (rc = mp_up(h->d, minDist, (size_t)M * 4)) || (rc = mp_up(h->e, obs_gt0, (size_t)M)) || (rc = mp_up(h->f, mp_desc, (size_t)M * 32)) ||
        (rc = mp_ensure(h->o1, (size_t)M * sizeof(oslam_proj_query_t))))
        return rc;
    rc = oslam_frame_is_in_frustum_device(M, (const float*)h->a.p, (const float*)h->b.p, (const float*)h->c.p, (const float*)h->d.p, (const uint8_t*)h->e.p,
                                          (const uint8_t*)h->f.p, Tcw, K5, bounds, viewingCosLimit, logScaleFactor, scaleFactors, nLevels, th,
                                          (oslam_proj_query_t*)h->o1.p, nullptr);
    if (rc) return rc;
    OSLAM_HIP_CHECK(hipMemcpy(out, h->o1.p, (size_t)M * sizeof(oslam_proj_query_t), hipMemcpyDeviceToHost));
    return OSLAM_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// LocalMapping::CreateNewMapPoints, per-match numeric core (reference src/LocalMapping.cc:291-432).
// One thread per candidate match; all neighbour keyframes of one current keyframe in one launch.
// ---------------------------------------------------------------------------------------------------------------
namespace oslam {

struct TriKfDev {
    float Tcw[16], Twc[16];
    float fx, fy, cx, cy, invfx, invfy, mbf, mb;
    int kp_off;   // first keypoint of this keyframe in the concatenated per-keypoint tables
};

struct TriCtx {
    int M, nLevels;
    TriKfDev kf1;
    const TriKfDev* kf1s;         // [nPairs] or NULL: one current keyframe (kf1) for all pairs
    const TriKfDev* kf2;          // [nPairs]
    const int* pair_of;           // [M]
    const int* idx1; const int* idx2;
    const oslam_keypoint_t* keysUn; const oslam_keypoint_t* keys; const float* uRight; const float* depth;   // concatenated: kf1 first
    float scale[OSLAM_MAX_LEVELS], sigma2[OSLAM_MAX_LEVELS];
    float ratioFactor;
    uint8_t* ok; float* x3D;
};

__device__ __forceinline__ double dot3d(const float* a, const float* b) {
    double r = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) r += (double)a[k] * (double)b[k];
    return r;
}
// cv::gemm small-matrix branch (float accumulation), optional + c
__device__ __forceinline__ float rowmul3(const float* row, const float* x) { return row[0] * x[0] + row[1] * x[1] + row[2] * x[2]; }

// one Jacobi rotation between rows I and J of At (and of Vt); returns whether it rotated
template <int I, int J>
__device__ __forceinline__ bool jacobi_pair(float (&At)[16], float (&Vt)[16], double (&W)[4]) {
    double a = W[I], b = W[J], p = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) p += (double)At[I * 4 + k] * (double)At[J * 4 + k];
    const float eps = 2.0f * 1.1920928955078125e-07f;
    if (fabs(p) <= (double)eps * sqrt(a * b)) return false;
    p *= 2;
    const double beta = a - b, gamma = hypot(p, beta);
    float c, s;
    if (beta < 0) {
        const double delta = (gamma - beta) * 0.5;
        s = (float)sqrt(delta / gamma);
        c = (float)(p / (gamma * (double)s * 2));
    } else {
        c = (float)sqrt((gamma + beta) / (gamma * 2));
        s = (float)(p / (gamma * (double)c * 2));
    }
    a = b = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float ai = At[I * 4 + k], aj = At[J * 4 + k];
        const float t0 = c * ai + s * aj, t1 = -s * ai + c * aj;
        At[I * 4 + k] = t0; At[J * 4 + k] = t1;
        a += (double)t0 * (double)t0; b += (double)t1 * (double)t1;
    }
    W[I] = a; W[J] = b;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float vi = Vt[I * 4 + k], vj = Vt[J * 4 + k];
        Vt[I * 4 + k] = c * vi + s * vj;
        Vt[J * 4 + k] = -s * vi + c * vj;
    }
    return true;
}

// last right singular vector of a 4x4 float matrix (row-major A), as cv::SVD::compute(...).vt.row(3)
__device__ void svd4_last_row(const float (&A)[16], float (&out)[4]) {
    float At[16], Vt[16];
    double W[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { At[i * 4 + k] = A[k * 4 + i]; sd += (double)At[i * 4 + k] * (double)At[i * 4 + k]; Vt[i * 4 + k] = (i == k) ? 1.f : 0.f; }
        W[i] = sd;
    }
    for (int iter = 0; iter < 30; iter++) {
        bool changed = false;
        changed |= jacobi_pair<0, 1>(At, Vt, W);
        changed |= jacobi_pair<0, 2>(At, Vt, W);
        changed |= jacobi_pair<0, 3>(At, Vt, W);
        changed |= jacobi_pair<1, 2>(At, Vt, W);
        changed |= jacobi_pair<1, 3>(At, Vt, W);
        changed |= jacobi_pair<2, 3>(At, Vt, W);
        if (!changed) break;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) sd += (double)At[i * 4 + k] * (double)At[i * 4 + k];
        W[i] = sqrt(sd);
    }
    // selection sort by decreasing W (strict <), tracking only which original row ends up last
    int perm[4] = {0, 1, 2, 3};
#pragma unroll
    for (int i = 0; i < 3; i++) {
        int j = i;
#pragma unroll
        for (int k = i + 1; k < 4; k++)
            if (W[j] < W[k]) j = k;
        if (j != i) {
            // static-index swap
#pragma unroll
            for (int k = i + 1; k < 4; k++)
                if (k == j) { const double tw = W[i]; W[i] = W[k]; W[k] = tw; const int tp = perm[i]; perm[i] = perm[k]; perm[k] = tp; }
        }
    }
    const int r = perm[3];
#pragma unroll
    for (int k = 0; k < 4; k++) out[k] = r == 0 ? Vt[k] : r == 1 ? Vt[4 + k] : r == 2 ? Vt[8 + k] : Vt[12 + k];
}

__device__ __forceinline__ bool tri_reproj_ok(const TriKfDev& kf, float mbf, float kx, float ky, float kur, bool bStereo, float x, float y, float z, float sig) {
    const float invz = (float)(1.0 / (double)z);
    const float u = kf.fx * x * invz + kf.cx;
    const float v = kf.fy * y * invz + kf.cy;
    const float ex = u - kx, ey = v - ky;
    if (!bStereo) return !((double)(ex * ex + ey * ey) > 5.991 * (double)sig);
    const float u_r = u - mbf * invz;
    const float er = u_r - kur;
    return !((double)(ex * ex + ey * ey + er * er) > 7.8 * (double)sig);
}

__device__ __forceinline__ bool tri_unproject(const TriKfDev& kf, float rawx, float rawy, float z, float (&out)[3]) {
    if (!(z > 0)) return false;
    const float x = (rawx - kf.cx) * z * kf.invfx;
    const float y = (rawy - kf.cy) * z * kf.invfy;
    const float xc[3] = {x, y, z};
#pragma unroll
    for (int r = 0; r < 3; r++) out[r] = (float)((double)rowmul3(kf.Twc + r * 4, xc) + (double)kf.Twc[r * 4 + 3]);
    return true;
}

__global__ __launch_bounds__(128) void k_triangulate(TriCtx c) {
    const int m = blockIdx.x * 128 + threadIdx.x;
    if (m >= c.M) return;
    const TriKfDev k1 = c.kf1s ? c.kf1s[c.pair_of[m]] : c.kf1;
    const TriKfDev k2 = c.kf2[c.pair_of[m]];
    const int i1 = k1.kp_off + c.idx1[m], i2 = k2.kp_off + c.idx2[m];
    const oslam_keypoint_t kp1 = c.keysUn[i1], kp2 = c.keysUn[i2];
    const float ur1 = c.uRight[i1], ur2 = c.uRight[i2];
    const bool bS1 = ur1 >= 0, bS2 = ur2 >= 0;
    bool ok = true;
    float X[3] = {0, 0, 0};

    const float xn1[3] = {(kp1.x - k1.cx) * k1.invfx, (kp1.y - k1.cy) * k1.invfy, 1.0f};
    const float xn2[3] = {(kp2.x - k2.cx) * k2.invfx, (kp2.y - k2.cy) * k2.invfy, 1.0f};
    float ray1[3], ray2[3];
#pragma unroll
    for (int r = 0; r < 3; r++) { ray1[r] = rowmul3(k1.Twc + r * 4, xn1); ray2[r] = rowmul3(k2.Twc + r * 4, xn2); }
    const float cosRays = (float)(dot3d(ray1, ray2) / (sqrt(dot3d(ray1, ray1)) * sqrt(dot3d(ray2, ray2))));
    float cs1 = cosRays + 1, cs2 = cosRays + 1;
    // std::atan2(float,float) / std::cos(float): evaluated in fp64 and rounded (glibc's float versions are within 1 ulp)
    if (bS1) cs1 = (float)cos((double)(2 * (float)atan2((double)(k1.mb / 2), (double)c.depth[i1])));
    else if (bS2) cs2 = (float)cos((double)(2 * (float)atan2((double)(k2.mb / 2), (double)c.depth[i2])));
    const float cosStereo = fminf(cs1, cs2);

    if (cosRays < cosStereo && cosRays > 0 && (bS1 || bS2 || (double)cosRays < 0.9998)) {
        float A[16];
#pragma unroll
        for (int k = 0; k < 4; k++) {   // cv::addWeighted in double, one rounding
            A[k] = (float)((double)k1.Tcw[8 + k] * (double)xn1[0] + (double)k1.Tcw[k] * -1.0 + 0.0);
            A[4 + k] = (float)((double)k1.Tcw[8 + k] * (double)xn1[1] + (double)k1.Tcw[4 + k] * -1.0 + 0.0);
            A[8 + k] = (float)((double)k2.Tcw[8 + k] * (double)xn2[0] + (double)k2.Tcw[k] * -1.0 + 0.0);
            A[12 + k] = (float)((double)k2.Tcw[8 + k] * (double)xn2[1] + (double)k2.Tcw[4 + k] * -1.0 + 0.0);
        }
        float v[4];
        svd4_last_row(A, v);
        if (v[3] == 0) ok = false;
        else {
            const float inv = (float)(1.0 / (double)v[3]);
#pragma unroll
            for (int k = 0; k < 3; k++) X[k] = v[k] * inv + 0.0f;
        }
    } else if (bS1 && cs1 < cs2) {
        ok = tri_unproject(k1, c.keys[i1].x, c.keys[i1].y, c.depth[i1], X);
    } else if (bS2 && cs2 < cs1) {
        ok = tri_unproject(k2, c.keys[i2].x, c.keys[i2].y, c.depth[i2], X);
    } else
        ok = false;

    if (ok) {
        const float z1 = (float)(dot3d(k1.Tcw + 8, X) + (double)k1.Tcw[11]);
        const float z2 = (float)(dot3d(k2.Tcw + 8, X) + (double)k2.Tcw[11]);
        ok = !(z1 <= 0) && !(z2 <= 0);
        if (ok) {
            const float x1 = (float)(dot3d(k1.Tcw, X) + (double)k1.Tcw[3]), y1 = (float)(dot3d(k1.Tcw + 4, X) + (double)k1.Tcw[7]);
            ok = tri_reproj_ok(k1, k1.mbf, kp1.x, kp1.y, ur1, bS1, x1, y1, z1, c.sigma2[kp1.octave]);
        }
        if (ok) {
            const float x2 = (float)(dot3d(k2.Tcw, X) + (double)k2.Tcw[3]), y2 = (float)(dot3d(k2.Tcw + 4, X) + (double)k2.Tcw[7]);
            ok = tri_reproj_ok(k2, k1.mbf, kp2.x, kp2.y, ur2, bS2, x2, y2, z2, c.sigma2[kp2.octave]);
        }
        if (ok) {
            const float n1[3] = {X[0] - k1.Twc[3], X[1] - k1.Twc[7], X[2] - k1.Twc[11]};
            const float n2[3] = {X[0] - k2.Twc[3], X[1] - k2.Twc[7], X[2] - k2.Twc[11]};
            const float d1 = (float)sqrt(dot3d(n1, n1)), d2 = (float)sqrt(dot3d(n2, n2));
            ok = !(d1 == 0 || d2 == 0);
            if (ok) {
                const float ratioDist = __fdiv_rn(d2, d1);
                const float ratioOct = __fdiv_rn(c.scale[kp1.octave], c.scale[kp2.octave]);
                ok = !(ratioDist * c.ratioFactor < ratioOct || ratioDist > ratioOct * c.ratioFactor);
            }
        }
    }
    c.ok[m] = ok ? 1 : 0;
#pragma unroll
    for (int k = 0; k < 3; k++) c.x3D[m * 3 + k] = ok ? X[k] : 0.f;
}

}  // namespace oslam

extern "C" int oslam_mp_triangulate(oslam_mappoint_t* h, const oslam_tri_kf_t* kf1, int nPairs, const oslam_tri_kf_t* kf2, const int32_t* pair_start,
                                    const int32_t* idx1, const int32_t* idx2, const float* scaleFactors, const float* levelSigma2, int nLevels,
                                    float ratioFactor, uint8_t* ok, float* x3D, int32_t* nnew) {
    if (!h || !kf1 || nPairs < 0 || (nPairs > 0 && (!kf2 || !pair_start)) || !scaleFactors || !levelSigma2 || nLevels < 1 || nLevels > OSLAM_MAX_LEVELS) {
        set_error("bad argument");
        return OSLAM_E_INVALID;
    }
    if (nnew) *nnew = 0;
    const int M = nPairs ? pair_start[nPairs] : 0;
    if (M == 0) return OSLAM_OK;
    if (M < 0 || !idx1 || !idx2 || !ok || !x3D) { set_error("bad match table"); return OSLAM_E_INVALID; }
    // validate every index on the host: the kernel trusts them
    auto kf_ok = [&](const oslam_tri_kf_t& k) { return k.n_kps >= 0 && (k.n_kps == 0 || (k.keysUn && k.keys && k.uRight && k.depth)); };
    if (!kf_ok(*kf1)) { set_error("kf1 tables missing"); return OSLAM_E_INVALID; }
    std::vector<int> pair_of(M);
    std::vector<TriKfDev> dev2(nPairs);
    size_t total = (size_t)kf1->n_kps;
    auto fill = [](TriKfDev& d, const oslam_tri_kf_t& k, int off) {
        for (int i = 0; i < 16; i++) { d.Tcw[i] = k.Tcw[i]; d.Twc[i] = k.Twc[i]; }
        d.fx = k.fx; d.fy = k.fy; d.cx = k.cx; d.cy = k.cy; d.invfx = k.invfx; d.invfy = k.invfy; d.mbf = k.mbf; d.mb = k.mb;
        d.kp_off = off;
    };
    for (int p = 0; p < nPairs; p++) {
        if (!kf_ok(kf2[p]) || pair_start[p + 1] < pair_start[p] || pair_start[0] != 0) { set_error("bad pair %d", p); return OSLAM_E_INVALID; }
        fill(dev2[p], kf2[p], (int)total);
        for (int m = pair_start[p]; m < pair_start[p + 1]; m++) {
            pair_of[m] = p;
            if (idx1[m] < 0 || idx1[m] >= kf1->n_kps || idx2[m] < 0 || idx2[m] >= kf2[p].n_kps) { set_error("match %d: keypoint index out of range", m); return OSLAM_E_INVALID; }
        }
        total += (size_t)kf2[p].n_kps;
    }
    // levels of the matched keypoints index the scale tables
    for (int m = 0; m < M; m++) {
        const int o1 = kf1->keysUn[idx1[m]].octave, o2 = kf2[pair_of[m]].keysUn[idx2[m]].octave;
        if (o1 < 0 || o1 >= nLevels || o2 < 0 || o2 >= nLevels) { set_error("match %d: octave out of range", m); return OSLAM_E_INVALID; }
    }
    OSLAM_HIP_CHECK(hipSetDevice(h->device));
    std::vector<oslam_keypoint_t> kun(total), kraw(total);
    std::vector<float> ur(total), dep(total);
    auto put = [&](const oslam_tri_kf_t& k, size_t off) {
        if (!k.n_kps) return;
        memcpy(&kun[off], k.keysUn, (size_t)k.n_kps * sizeof(oslam_keypoint_t));
        memcpy(&kraw[off], k.keys, (size_t)k.n_kps * sizeof(oslam_keypoint_t));
        memcpy(&ur[off], k.uRight, (size_t)k.n_kps * 4);
        memcpy(&dep[off], k.depth, (size_t)k.n_kps * 4);
    };
    put(*kf1, 0);
    for (int p = 0; p < nPairs; p++) put(kf2[p], (size_t)dev2[p].kp_off);
    int rc;
    if ((rc = mp_up(h->a, kun.data(), total * sizeof(oslam_keypoint_t))) || (rc = mp_up(h->b, kraw.data(), total * sizeof(oslam_keypoint_t))) ||
        (rc = mp_up(h->c, ur.data(), total * 4)) || (rc = mp_up(h->d, dep.data(), total * 4)) || (rc = mp_up(h->e, dev2.data(), dev2.size() * sizeof(TriKfDev))) ||
        (rc = mp_up(h->f, pair_of.data(), (size_t)M * 4)) || (rc = mp_up(h->g, idx1, (size_t)M * 4)) || (rc = mp_up(h->g2, idx2, (size_t)M * 4)) ||
        (rc = mp_ensure(h->o1, (size_t)M)) || (rc = mp_ensure(h->o2, (size_t)M * 12)))
        return rc;
    TriCtx c;
    c.M = M; c.nLevels = nLevels;
    fill(c.kf1, *kf1, 0);
    c.kf1s = nullptr;
    c.kf2 = (const TriKfDev*)h->e.p; c.pair_of = (const int*)h->f.p; c.idx1 = (const int*)h->g.p; c.idx2 = (const int*)h->g2.p;
    c.keysUn = (const oslam_keypoint_t*)h->a.p; c.keys = (const oslam_keypoint_t*)h->b.p; c.uRight = (const float*)h->c.p; c.depth = (const float*)h->d.p;
    for (int i = 0; i < OSLAM_MAX_LEVELS; i++) { c.scale[i] = i < nLevels ? scaleFactors[i] : 1.f; c.sigma2[i] = i < nLevels ? levelSigma2[i] : 1.f; }
    c.ratioFactor = ratioFactor;
    c.ok = (uint8_t*)h->o1.p; c.x3D = (float*)h->o2.p;
    hipLaunchKernelGGL(k_triangulate, dim3(div_up(M, 128)), dim3(128), 0, nullptr, c);
    OSLAM_HIP_CHECK(hipGetLastError());
    OSLAM_HIP_CHECK(hipMemcpy(ok, h->o1.p, (size_t)M, hipMemcpyDeviceToHost));
    OSLAM_HIP_CHECK(hipMemcpy(x3D, h->o2.p, (size_t)M * 12, hipMemcpyDeviceToHost));
    if (nnew) { int n = 0; for (int m = 0; m < M; m++) n += ok[m]; *nnew = n; }
    return OSLAM_OK;
}

// Batch form for several (current keyframe, neighbour) pairs that may belong to different maps: pair p = (kf1[p], kf2[p]).  Only the matched
// keypoints travel: the kernel gathers through per-match rows (2m, 2m+1) of compact tables built here.
extern "C" int oslam_mp_triangulate_pairs(oslam_mappoint_t* h, int nPairs, const oslam_tri_kf_t* kf1, const oslam_tri_kf_t* kf2, const int32_t* pair_start,
                                          const int32_t* idx1, const int32_t* idx2, const float* scaleFactors, const float* levelSigma2, int nLevels,
                                          float ratioFactor, uint8_t* ok, float* x3D) {
    if (!h || nPairs < 0 || (nPairs > 0 && (!kf1 || !kf2 || !pair_start)) || !scaleFactors || !levelSigma2 || nLevels < 1 || nLevels > OSLAM_MAX_LEVELS) {
        set_error("bad argument");
        return OSLAM_E_INVALID;
    }
    const int M = nPairs ? pair_start[nPairs] : 0;
    if (M == 0) return OSLAM_OK;
    if (M < 0 || !idx1 || !idx2 || !ok || !x3D || pair_start[0] != 0) { set_error("bad match table"); return OSLAM_E_INVALID; }
    OSLAM_HIP_CHECK(hipSetDevice(h->device));
    // layout of the staging block: inputs, then outputs
    size_t at = 0;
    auto take = [&](size_t bytes) { const size_t o = at; at += (bytes + 255) & ~(size_t)255; return o; };
    const size_t total = (size_t)2 * M;
    const size_t o_kun = take(total * sizeof(oslam_keypoint_t)), o_kraw = take(total * sizeof(oslam_keypoint_t)), o_ur = take(total * 4), o_dep = take(total * 4),
                 o_dev1 = take((size_t)nPairs * sizeof(TriKfDev)), o_dev2 = take((size_t)nPairs * sizeof(TriKfDev)), o_pair = take((size_t)M * 4), o_d1 = take((size_t)M * 4),
                 o_d2 = take((size_t)M * 4);
    const size_t in_bytes = at;
    const size_t o_ok = take((size_t)M), o_x3 = take((size_t)M * 12);
    if (at > h->st_cap) {
        OSLAM_HIP_CHECK(hipDeviceSynchronize());
        if (h->st_h) (void)hipHostFree(h->st_h);
        if (h->st_d) (void)hipFree(h->st_d);
        h->st_h = nullptr; h->st_d = nullptr; h->st_cap = 0;
        const size_t ncap = at + at / 2 + 4096;
        OSLAM_HIP_CHECK(hipHostMalloc((void**)&h->st_h, ncap, 0));
        OSLAM_HIP_CHECK(hipMalloc((void**)&h->st_d, ncap));
        h->st_cap = ncap;
    }
    if (!h->strm) OSLAM_HIP_CHECK(hipStreamCreateWithFlags(&h->strm, hipStreamNonBlocking));
    uint8_t* H = h->st_h;
    uint8_t* D = h->st_d;
    int* pair_of = (int*)(H + o_pair); int* d1 = (int*)(H + o_d1); int* d2 = (int*)(H + o_d2);
    TriKfDev* dev1 = (TriKfDev*)(H + o_dev1); TriKfDev* dev2 = (TriKfDev*)(H + o_dev2);
    oslam_keypoint_t* kun = (oslam_keypoint_t*)(H + o_kun); oslam_keypoint_t* kraw = (oslam_keypoint_t*)(H + o_kraw);
    float* ur = (float*)(H + o_ur); float* dep = (float*)(H + o_dep);
    auto fill = [](TriKfDev& d, const oslam_tri_kf_t& k) {
        for (int i = 0; i < 16; i++) { d.Tcw[i] = k.Tcw[i]; d.Twc[i] = k.Twc[i]; }
        d.fx = k.fx; d.fy = k.fy; d.cx = k.cx; d.cy = k.cy; d.invfx = k.invfx; d.invfy = k.invfy; d.mbf = k.mbf; d.mb = k.mb;
        d.kp_off = 0;
    };
    for (int p = 0; p < nPairs; p++) {
        const oslam_tri_kf_t& a = kf1[p];
        const oslam_tri_kf_t& b = kf2[p];
        if (pair_start[p + 1] < pair_start[p]) { set_error("bad pair %d", p); return OSLAM_E_INVALID; }
        fill(dev1[p], a); fill(dev2[p], b);
        for (int m = pair_start[p]; m < pair_start[p + 1]; m++) {
            if (idx1[m] < 0 || idx1[m] >= a.n_kps || idx2[m] < 0 || idx2[m] >= b.n_kps || !a.keysUn || !a.keys || !a.uRight || !a.depth || !b.keysUn || !b.keys ||
                !b.uRight || !b.depth) { set_error("match %d: keypoint index out of range", m); return OSLAM_E_INVALID; }
            const int o1 = a.keysUn[idx1[m]].octave, o2 = b.keysUn[idx2[m]].octave;
            if (o1 < 0 || o1 >= nLevels || o2 < 0 || o2 >= nLevels) { set_error("match %d: octave out of range", m); return OSLAM_E_INVALID; }
            pair_of[m] = p; d1[m] = 2 * m; d2[m] = 2 * m + 1;
            kun[(size_t)2 * m] = a.keysUn[idx1[m]]; kraw[(size_t)2 * m] = a.keys[idx1[m]]; ur[(size_t)2 * m] = a.uRight[idx1[m]]; dep[(size_t)2 * m] = a.depth[idx1[m]];
            kun[(size_t)2 * m + 1] = b.keysUn[idx2[m]]; kraw[(size_t)2 * m + 1] = b.keys[idx2[m]]; ur[(size_t)2 * m + 1] = b.uRight[idx2[m]]; dep[(size_t)2 * m + 1] = b.depth[idx2[m]];
        }
    }
    OSLAM_HIP_CHECK(hipMemcpyAsync(D, H, in_bytes, hipMemcpyHostToDevice, h->strm));
    TriCtx c;
    c.M = M; c.nLevels = nLevels;
    fill(c.kf1, kf1[0]);
    c.kf1s = (const TriKfDev*)(D + o_dev1);
    c.kf2 = (const TriKfDev*)(D + o_dev2); c.pair_of = (const int*)(D + o_pair); c.idx1 = (const int*)(D + o_d1); c.idx2 = (const int*)(D + o_d2);
    c.keysUn = (const oslam_keypoint_t*)(D + o_kun); c.keys = (const oslam_keypoint_t*)(D + o_kraw); c.uRight = (const float*)(D + o_ur); c.depth = (const float*)(D + o_dep);
    for (int i = 0; i < OSLAM_MAX_LEVELS; i++) { c.scale[i] = i < nLevels ? scaleFactors[i] : 1.f; c.sigma2[i] = i < nLevels ? levelSigma2[i] : 1.f; }
    c.ratioFactor = ratioFactor;
    c.ok = D + o_ok; c.x3D = (float*)(D + o_x3);
    if (h->timing) (void)hipEventRecord(h->ev0, h->strm);
    hipLaunchKernelGGL(k_triangulate, dim3(div_up(M, 128)), dim3(128), 0, h->strm, c);
    if (h->timing) (void)hipEventRecord(h->ev1, h->strm);
    OSLAM_HIP_CHECK(hipGetLastError());
    OSLAM_HIP_CHECK(hipMemcpyAsync(H + o_ok, D + o_ok, at - in_bytes, hipMemcpyDeviceToHost, h->strm));
    OSLAM_HIP_CHECK(stream_wait(h->strm));
    if (h->timing) { float ms = 0.f; if (hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess) { h->kern_ms += ms; h->kern_n += 1; } }
    memcpy(ok, H + o_ok, (size_t)M);
    memcpy(x3D, H + o_x3, (size_t)M * 12);
    return OSLAM_OK;
}
