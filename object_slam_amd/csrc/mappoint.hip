// gfx950 MapPoint maintenance + Frame::isInFrustum (the steps right before
// SearchByProjection(F, vpMPs) and right after LocalBundleAdjustment), batched over map points:
//   MapPoint::ComputeDistinctiveDescriptors  reference src/MapPoint.cc:345-410
//   MapPoint::UpdateNormalAndDepth           reference src/MapPoint.cc:433-474
//   Frame::isInFrustum + MapPoint::PredictScale + the query fields of SearchByProjection
//                                            reference src/Frame.cc:509-565, src/MapPoint.cc:505-521, src/ORBmatcher.cc:57-67
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
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
        const double inv = 1.0 / norm3d(ax, ay, az);
        nx = nx + (float)((double)ax * inv);
        ny = ny + (float)((double)ay * inv);
        nz = nz + (float)((double)az * inv);
    }
    const float dist = (float)norm3d(px - OwRef[p * 3], py - OwRef[p * 3 + 1], pz - OwRef[p * 3 + 2]);
    const float maxD = dist * levelScale[p];
    const double invn = 1.0 / n;
    out[p * 5] = (float)((double)nx * invn);
    out[p * 5 + 1] = (float)((double)ny * invn);
    out[p * 5 + 2] = (float)((double)nz * invn);
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

__global__ __launch_bounds__(256) void k_is_in_frustum(FrustumCtx c) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= c.M) return;
    oslam_proj_query_t q;
    q.u = q.v = q.ur = q.radius = 0.f; q.minLevel = -1; q.maxLevel = -1; q.flags = 0; q.angle = 0.f;
    float Ow[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) s += (double)c.T[k * 4 + r] * (double)c.T[k * 4 + 3];
        Ow[r] = (float)(-1.0 * s);
    }
    const float P[3] = {c.Pw[i * 3], c.Pw[i * 3 + 1], c.Pw[i * 3 + 2]};
    float Pc[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {   // cv::gemm small-matrix branch: float accumulation
        const float t0 = c.T[r * 4] * P[0] + c.T[r * 4 + 1] * P[1] + c.T[r * 4 + 2] * P[2];
        Pc[r] = (float)((double)t0 + (double)c.T[r * 4 + 3]);
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
        if (c.th != 1.0f) r *= c.th;
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
    c.out[i] = q;
}

}  // namespace oslam

using namespace oslam;

struct oslam_mappoint {
    int device = 0;
    struct Buf { void* p = nullptr; size_t cap = 0; };
    Buf a, b, c, d, e, f, g, o1, o2;
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

void oslam_mappoint_destroy(oslam_mappoint_t* h) {
    if (!h) return;
    oslam_mappoint::Buf* bs[] = {&h->a, &h->b, &h->c, &h->d, &h->e, &h->f, &h->g, &h->o1, &h->o2};
    for (auto* b : bs)
        if (b->p) (void)hipFree(b->p);
    delete h;
}

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
