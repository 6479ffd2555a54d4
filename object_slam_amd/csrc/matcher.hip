// gfx950 ORB matcher: Frame grid binning (reference src/Frame.cc:455-470,567-632), windowed
// 256-bit Hamming search of ORBmatcher::SearchByProjection (reference src/ORBmatcher.cc:45-129 and
// :1328-1470) with the reference's sequential claim semantics reproduced by a fixpoint iteration,
// and the rotation-consistency histogram (:1431-1467, :1601-1642).
// One workgroup per frame; keypoints, descriptors and the 64x48 grid live in LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <vector>

#include "common.h"

namespace oslam {

constexpr int kGridCols = 64, kGridRows = 48;   // reference include/Frame.h:43-44
constexpr int kGridCells = kGridCols * kGridRows;
constexpr int kHistoLen = 30;                   // src/ORBmatcher.cc:39
#ifndef OSLAM_MATCH_THREADS
#define OSLAM_MATCH_THREADS 1024
#endif
constexpr int kMatchThreads = OSLAM_MATCH_THREADS;   // kernel experiments: -DOSLAM_MATCH_THREADS=512 (two queries per thread, two workgroups per CU)
constexpr int kMaxMatchKps = 2400;              // LDS budget, see lds_bytes()
constexpr int kCacheCap = 48;                   // cached candidates per query for the claim-order fixpoint

struct MatchCtx {
    // frame side
    const oslam_keypoint_t* keysUn; int kp_stride;
    const float* uRight;      // may be NULL (monocular: all -1)
    const uint8_t* desc;
    const uint8_t* blocked;   // may be NULL
    const int* n_kps;         // [B] (device) or NULL with n_kps_const
    int n_kps_const;
    float minX, minY, invW, invH;
    // query side
    const oslam_proj_query_t* q; int q_stride;
    const int* n_q; int n_q_const;
    float nnratio; int use_ratio, check_ori, th_high;
    int fuse;                 // Fuse search (src/ORBmatcher.cc:888-947): chi2 reprojection gate, no claims
    float invSigma2[OSLAM_MAX_LEVELS];
    // outputs
    int* q_match; int* q_dist; int* kp_match; int* nmatches;
    int* iters;               // [B] fixpoint iterations used (diagnostic)
    int* pairs;               // [B] descriptor pairs compared in the first pass (work model of SURVEY.md §8(d))
    uint32_t* cache;          // [B][q_stride][kCacheCap] gate-passing candidates of every query: dist << 16 | index, reference order
    int* ccount;              // [B][q_stride] cached count, or -1 if the query has more than kCacheCap candidates
    long long* dbg;           // [8] phase stamps of frame 0 (profiling builds, -DOSLAM_MATCH_PROFILE)
};

__host__ __device__ inline size_t match_lds_bytes(int ncap) {
    // desc 32 + xy 8 + uRight 4 + items 2 + Bcur 4 + Bprev 4 + owner 4 + octave 1 + blocked 1 per keypoint
    return (size_t)ncap * 60 + (kGridCells + 1) * 4 + 256;
}

__global__ __launch_bounds__(kMatchThreads) void k_search_window(MatchCtx c, int ncap) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int N = c.n_kps ? c.n_kps[b] : c.n_kps_const;
    const int M = c.n_q ? c.n_q[b] : c.n_q_const;
    const oslam_keypoint_t* kps = c.keysUn + (long long)b * c.kp_stride;
    const float* uR = c.uRight ? c.uRight + (long long)b * c.kp_stride : nullptr;
    const uint32_t* gdesc = (const uint32_t*)(c.desc + (long long)b * c.kp_stride * 32);
    const uint8_t* gblocked = c.blocked ? c.blocked + (long long)b * c.kp_stride : nullptr;
    const oslam_proj_query_t* Q = c.q + (long long)b * c.q_stride;
    int* q_match = c.q_match + (long long)b * c.q_stride;
    int* q_dist = c.q_dist + (long long)b * c.q_stride;
    int* kp_match = c.kp_match + (long long)b * c.kp_stride;
    uint32_t* cache = c.cache + (long long)b * c.q_stride * kCacheCap;
    int* ccount = c.ccount + (long long)b * c.q_stride;

    extern __shared__ __align__(16) uint8_t smem[];
    uint32_t* s_desc = (uint32_t*)smem;                     // [ncap][8] (a word-major layout was measured slower: 8 ds_read_b32 instead of 2 ds_read_b128 per candidate)
    float2* s_xy = (float2*)(s_desc + (size_t)ncap * 8);    // [ncap]
    float* s_ur = (float*)(s_xy + ncap);                    // [ncap]
    int* s_Bcur = (int*)(s_ur + ncap);                      // [ncap] min blocking claimer (this iteration)
    int* s_Bprev = s_Bcur + ncap;                           // [ncap] (previous iteration)
    int* s_owner = s_Bprev + ncap;                          // [ncap]
    int* s_cell = s_owner + ncap;                           // [kGridCells + 1] cell start
    uint16_t* s_items = (uint16_t*)(s_cell + kGridCells + 1);   // [ncap] keypoints sorted by cell, index order
    uint8_t* s_oct = (uint8_t*)(s_items + ncap);            // [ncap]
    uint8_t* s_blk = s_oct + ncap;                          // [ncap]
    __shared__ int s_hist[kHistoLen];
    __shared__ int s_changed, s_nm, s_ind[3], s_pairs;
    if (threadIdx.x == 0) s_pairs = 0;   // (barriers of the staging phase follow before the first use)
    __shared__ int s_wtot[kMatchThreads / 64];

#ifdef OSLAM_MATCH_PROFILE
    long long tw_ = wall_clock64();
#define MSTAMP(i) do { __syncthreads(); if (b == 0 && tid == 0) { const long long t_ = wall_clock64(); c.dbg[i] += t_ - tw_; tw_ = t_; } } while (0)
#else
#define MSTAMP(i) do { } while (0)
#endif
    if (N > ncap || N < 0 || M < 0 || M > c.q_stride) {   // host validates; never truncate silently
        if (tid == 0) c.nmatches[b] = -1;
        return;
    }

    // ---- load keypoints, AssignFeaturesToGrid (:455-470): cell = round((x-minX)*invW) ----
    for (int i = tid; i <= kGridCells; i += kMatchThreads) s_cell[i] = 0;
    __syncthreads();
    for (int i = tid; i < N; i += kMatchThreads) {
        const oslam_keypoint_t kp = kps[i];
        s_xy[i] = make_float2(kp.x, kp.y);
        s_oct[i] = (uint8_t)kp.octave;
        s_ur[i] = uR ? uR[i] : -1.0f;
        s_blk[i] = gblocked ? gblocked[i] : 0;
        s_Bprev[i] = 0x7fffffff;
        s_owner[i] = -1;
        const int px = (int)roundf((kp.x - c.minX) * c.invW);
        const int py = (int)roundf((kp.y - c.minY) * c.invH);
        if (px >= 0 && px < kGridCols && py >= 0 && py < kGridRows) atomicAdd(&s_cell[px * kGridRows + py], 1);
    }
    for (int i = tid; i < N * 8; i += kMatchThreads) s_desc[i] = gdesc[i];
    __syncthreads();
    MSTAMP(0);
    // exclusive scan of the 3072 cell counts (kCellsPer per thread)
    {
        constexpr int kCellsPer = kGridCells / kMatchThreads;
        static_assert(kCellsPer * kMatchThreads == kGridCells, "the cell scan gives every thread the same number of cells");
        const int lane = tid & 63, wv = tid >> 6;
        const int base = tid * kCellsPer;
        int cc[kCellsPer];
        int incl = 0;
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
    // scatter with the cell starts as cursors (afterwards s_cell[c] = end of cell c = start of c+1),
    // then restore index order inside every cell (mGrid push_back order, :467)
    for (int i = tid; i < N; i += kMatchThreads) {
        const float2 p = s_xy[i];
        const int px = (int)roundf((p.x - c.minX) * c.invW);
        const int py = (int)roundf((p.y - c.minY) * c.invH);
        if (px >= 0 && px < kGridCols && py >= 0 && py < kGridRows) {
            const int pos = atomicAdd(&s_cell[px * kGridRows + py], 1);
            s_items[pos] = (uint16_t)i;
        }
    }
    __syncthreads();
    for (int cell = tid; cell < kGridCells; cell += kMatchThreads) {
        const int st = cell > 0 ? s_cell[cell - 1] : 0, en = s_cell[cell];
        for (int a = st + 1; a < en; a++) {   // insertion sort, cells hold a handful of points
            const uint16_t v = s_items[a];
            int q = a - 1;
            while (q >= st && s_items[q] > v) { s_items[q + 1] = s_items[q]; q--; }
            s_items[q + 1] = v;
        }
    }
    __syncthreads();

    MSTAMP(1);
#if defined(OSLAM_MATCH_ABLATE) && OSLAM_MATCH_ABLATE == 1
    if (tid == 0) { c.nmatches[b] = 0; if (c.iters) c.iters[b] = 0; if (c.pairs) c.pairs[b] = 0; }
    return;   // timing experiment: staging + grid only
#endif
    // ---- fixpoint over the sequential claim order (:87-89, :123 / :1402-1404, :1428) ----
    // A thread's first query (j == tid) keeps its flags, its cached-candidate count and the first kRegCache cached
    // candidates in registers after iteration 0, so the replay iterations of frames with <= 1024 queries touch no
    // global memory (each replay used to walk the per-query list in HBM/L2 with one exposed latency per entry).
    constexpr int kRegCache = 16;
    uint32_t rc[kRegCache];
#pragma unroll
    for (int t = 0; t < kRegCache; t++) rc[t] = 0;
    int my_flags = 0, my_nc = -1;
    int it = 0;
    for (; it < M + 2; it++) {
        for (int i = tid; i < N; i += kMatchThreads) s_Bcur[i] = 0x7fffffff;
        if (tid == 0) s_changed = 0;
        __syncthreads();
        if (it == 1 && tid < M) {   // this thread wrote the list in iteration 0: read it back once, as vectors
            my_nc = ccount[tid];
            const uint4* cv = (const uint4*)(cache + (long long)tid * kCacheCap);
#pragma unroll
            for (int t = 0; t < kRegCache / 4; t++) {
                const uint4 v = cv[t];
                rc[4 * t] = v.x; rc[4 * t + 1] = v.y; rc[4 * t + 2] = v.z; rc[4 * t + 3] = v.w;
            }
        }
        for (int j = tid; j < M; j += kMatchThreads) {
            const oslam_proj_query_t* qp = Q + j;
            int bestIdx = -1, bestDist = 256;
            const bool mine = (j == tid) && it > 0;
            const int flags = mine ? my_flags : qp->flags;
            if (j == tid && it == 0) my_flags = flags;
            const int ncached = it > 0 ? (mine ? my_nc : ccount[j]) : -1;
            if ((flags & 1) && ncached >= 0) {
                // iterations >= 1: the gate-passing candidates and their distances do not change, only the
                // claims do: replay the cached list (reference order) against the current claim table
                int bestLevel = -1, bestDist2 = 256, bestLevel2 = -1;
                const uint32_t* cl = cache + (long long)j * kCacheCap;
                auto consider = [&](uint32_t en) {
                    const int k = en & 0xFFFF, dist = en >> 16;
                    if (s_Bprev[k] < j) return;
                    const int oct = s_oct[k];
                    if (dist < bestDist) {
                        bestDist2 = bestDist; bestDist = dist;
                        bestLevel2 = bestLevel; bestLevel = oct;
                        bestIdx = k;
                    } else if (dist < bestDist2) {
                        bestLevel2 = oct; bestDist2 = dist;
                    }
                };
                if (mine) {
#pragma unroll
                    for (int t = 0; t < kRegCache; t++)
                        if (t < ncached) consider(rc[t]);
                    for (int t = kRegCache; t < ncached; t++) consider(cl[t]);
                } else {
                    for (int t = 0; t < ncached; t++) consider(cl[t]);
                }
                if (bestDist <= c.th_high) {
                    if (c.use_ratio && bestLevel == bestLevel2 && (float)bestDist > c.nnratio * (float)bestDist2) bestIdx = -1;
                } else
                    bestIdx = -1;
            } else if (flags & 1) {
                int nc = 0;   // candidates cached in iteration 0
                uint32_t* cl = cache + (long long)j * kCacheCap;
                const uint4 qa = ((const uint4*)qp)[0], qb = ((const uint4*)qp)[1];   // u, v, ur, radius | minLevel, maxLevel, flags, angle
                const float x = __uint_as_float(qa.x), y = __uint_as_float(qa.y), qur = __uint_as_float(qa.z), r = __uint_as_float(qa.w);
                const int minLevel = (int)qb.x, maxLevel = (int)qb.y;
                // GetFeaturesInArea (:567-620)
                const int nMinCellX = max(0, (int)floorf((x - c.minX - r) * c.invW));
                const int nMaxCellX = min(kGridCols - 1, (int)ceilf((x - c.minX + r) * c.invW));
                const int nMinCellY = max(0, (int)floorf((y - c.minY - r) * c.invH));
                const int nMaxCellY = min(kGridRows - 1, (int)ceilf((y - c.minY + r) * c.invH));
                if (nMinCellX < kGridCols && nMaxCellX >= 0 && nMinCellY < kGridRows && nMaxCellY >= 0) {
                    const bool bCheckLevels = !c.fuse && ((minLevel > 0) || (maxLevel >= 0));   // KeyFrame::GetFeaturesInArea has no level filter
                    const uint4 qd0 = ((const uint4*)qp)[2], qd1 = ((const uint4*)qp)[3];
                    const uint32_t qd[8] = {qd0.x, qd0.y, qd0.z, qd0.w, qd1.x, qd1.y, qd1.z, qd1.w};
                    int bestLevel = -1, bestDist2 = 256, bestLevel2 = -1;
                    for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
                        const int c0 = ix * kGridRows + nMinCellY;
                        const int s = c0 > 0 ? s_cell[c0 - 1] : 0, e = s_cell[ix * kGridRows + nMaxCellY];
                        for (int t = s; t < e; t++) {   // cells iy = min..max are contiguous in this layout
                            const int k = s_items[t];
                            const int oct = s_oct[k];
                            if (bCheckLevels) {
                                if (oct < minLevel) continue;
                                if (maxLevel >= 0 && oct > maxLevel) continue;
                            }
                            const float2 p = s_xy[k];
                            if (!(fabsf(p.x - x) < r && fabsf(p.y - y) < r)) continue;
                            const float kur = s_ur[k];
                            if (c.fuse) {
                                // level gate is explicit in Fuse (:903-906); chi2 gates :908-934
                                if (oct < minLevel || oct > maxLevel) continue;
                                const float ex = x - p.x, ey = y - p.y;
                                if (kur >= 0) {
                                    const float er = qur - kur;
                                    const float e2 = ex * ex + ey * ey + er * er;
                                    if ((double)(e2 * c.invSigma2[oct]) > 7.8) continue;
                                } else {
                                    const float e2 = ex * ex + ey * ey;
                                    if ((double)(e2 * c.invSigma2[oct]) > 5.99) continue;
                                }
                            } else {
                                if (s_blk[k]) continue;
                                if (kur > 0) {
                                    const float er = fabsf(qur - kur);
                                    if (er > r) continue;
                                }
                            }
                            const uint32_t* d = s_desc + k * 8;
                            int dist = 0;
#pragma unroll
                            for (int w = 0; w < 8; w++) dist += __popc(qd[w] ^ d[w]);
                            if (it == 0) {
                                if (nc < kCacheCap) cl[nc] = ((uint32_t)dist << 16) | (uint32_t)k;
                                nc++;
                            }
                            if (!c.fuse && s_Bprev[k] < j) continue;   // claimed by an earlier observed map point (:87-89)
                            if (dist < bestDist) {
                                bestDist2 = bestDist; bestDist = dist;
                                bestLevel2 = bestLevel; bestLevel = oct;
                                bestIdx = k;
                            } else if (dist < bestDist2) {
                                bestLevel2 = oct; bestDist2 = dist;
                            }
                        }
                    }
                    if (bestDist <= c.th_high) {
                        if (c.use_ratio && bestLevel == bestLevel2 && (float)bestDist > c.nnratio * (float)bestDist2) bestIdx = -1;
                    } else
                        bestIdx = -1;
                }
                if (it == 0) { ccount[j] = nc <= kCacheCap ? nc : -1; if (nc) atomicAdd(&s_pairs, nc); }
            }
            q_match[j] = bestIdx;
            q_dist[j] = bestIdx >= 0 ? bestDist : 256;
            if (bestIdx >= 0 && (flags & 2)) atomicMin(&s_Bcur[bestIdx], j);
        }
        __syncthreads();
        int diff = 0;
        for (int i = tid; i < N; i += kMatchThreads) diff |= (s_Bcur[i] != s_Bprev[i]);
        if (diff) s_changed = 1;
        __syncthreads();
        const int changed = s_changed;
        __syncthreads();
        if (it == 0) MSTAMP(2); else MSTAMP(3);
#if defined(OSLAM_MATCH_ABLATE) && OSLAM_MATCH_ABLATE == 2
        break;   // timing experiment: first pass only
#endif
        if (!changed || c.fuse) break;   // Fuse has no claims: one pass
        { int* t = s_Bcur; s_Bcur = s_Bprev; s_Bprev = t; }
    }

    // ---- mvpMapPoints after the loop: last accepted claimer wins; rotation consistency ----
    if (tid < kHistoLen) s_hist[tid] = 0;
    if (tid == 0) s_nm = 0;
    __syncthreads();
    const float factor = 1.0f / kHistoLen;
    int local_nm = 0;
    for (int j = tid; j < M; j += kMatchThreads) {
        const int k = q_match[j];
        if (k >= 0) {
            local_nm++;
            atomicMax(&s_owner[k], j);
            if (c.check_ori) {
                float rot = Q[j].angle - kps[k].angle;
                if (rot < 0.0f) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == kHistoLen) bin = 0;
                atomicAdd(&s_hist[bin], 1);
            }
        }
    }
    if (local_nm) atomicAdd(&s_nm, local_nm);
    __syncthreads();
    if (c.check_ori) {
        if (tid == 0) {   // ComputeThreeMaxima (:1601-1642)
            int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
            for (int i = 0; i < kHistoLen; i++) {
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
        for (int j = tid; j < M; j += kMatchThreads) {
            const int k = q_match[j];
            if (k >= 0) {
                float rot = Q[j].angle - kps[k].angle;
                if (rot < 0.0f) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == kHistoLen) bin = 0;
                if (bin != s_ind[0] && bin != s_ind[1] && bin != s_ind[2]) {
                    removed++;
                    s_owner[k] = -2 - M;   // below every query index: marks "set to NULL" (applied after all claims)
                }
            }
        }
        if (removed) atomicSub(&s_nm, removed);
        __syncthreads();
    }
    // a keypoint nulled by one claimer stays NULL even if another (kept) claimer also wrote it,
    // because the reference nulls after the whole loop (:1447-1463); atomicMax above ran before.
    for (int i = tid; i < N; i += kMatchThreads) {
        const int o = s_owner[i];
        kp_match[i] = o >= 0 ? o : (o == -1 ? -1 : -2);
    }
    if (tid == 0) {
        c.nmatches[b] = s_nm;
        if (c.iters) c.iters[b] = it + 1;
        if (c.pairs) c.pairs[b] = s_pairs;
    }
    MSTAMP(4);
}

// Projection half of SearchByProjection(Cur, Last), reference src/ORBmatcher.cc:1338-1392.
// cv::Mat products are single cv::gemm calls: fp64 accumulation, one rounding to float.
struct ProjectCtx {
    const float* Xw; const uint8_t* has_mp; const oslam_keypoint_t* keys; const uint8_t* mp_desc;
    int kp_stride; const int* n_last; int n_last_const;
    const float* Tcw; const float* Tlw;   // [B][16]
    float fx, fy, cx, cy, bf, b;
    float minX, minY, maxX, maxY;
    float scale[OSLAM_MAX_LEVELS];
    float th; int bMono;
    oslam_proj_query_t* out; int q_stride; int* n_q;
};

// cv::Mat 3x3*3x1+3x1 = one cv::gemm, flags==0, len==3: float accumulation (OpenCV small-matrix branch),
// then (float)(t0*1.0 + c*1.0) in double
__device__ __forceinline__ float gemm_row(const float* a, const float* x, float cc) {
    const float t0 = a[0] * x[0] + a[1] * x[1] + a[2] * x[2];
    return (float)((double)t0 + (double)cc);
}

__global__ __launch_bounds__(256) void k_project_last(ProjectCtx c) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int N = c.n_last ? c.n_last[b] : c.n_last_const;
    if (i == 0) c.n_q[b] = N;
    if (i >= N) return;
    const float* T = c.Tcw + b * 16;
    const float* Tl = c.Tlw + b * 16;
    float Rcw[3][3], tcw[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
        for (int k = 0; k < 3; k++) Rcw[r][k] = T[r * 4 + k];
        tcw[r] = T[r * 4 + 3];
    }
    // twc = -Rcw^T tcw ; tlc.z = Rlw[2,:] twc + tlw.z
    float twc[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) s += (double)Rcw[k][r] * (double)tcw[k];
        twc[r] = (float)(-1.0 * s);
    }
    const float Rl2[3] = {Tl[8], Tl[9], Tl[10]};
    const float tlcz = gemm_row(Rl2, twc, Tl[11]);
    const bool bForward = tlcz > c.b && !c.bMono;
    const bool bBackward = -tlcz > c.b && !c.bMono;

    const long long o = (long long)b * c.kp_stride + i;
    oslam_proj_query_t q;
    q.u = q.v = q.ur = q.radius = 0.f;
    q.minLevel = -1; q.maxLevel = -1; q.flags = 0; q.angle = 0.f;
    const uint8_t hm = c.has_mp[o];
    bool ok = hm & 1;
    if (ok) {
        const float X[3] = {c.Xw[o * 3], c.Xw[o * 3 + 1], c.Xw[o * 3 + 2]};
        const float xc = gemm_row(Rcw[0], X, tcw[0]);
        const float yc = gemm_row(Rcw[1], X, tcw[1]);
        const float zc = gemm_row(Rcw[2], X, tcw[2]);
        const float invzc = (float)(1.0 / (double)zc);
        ok = !(invzc < 0);
        if (ok) {
            const float u = c.fx * xc * invzc + c.cx;
            const float v = c.fy * yc * invzc + c.cy;
            ok = !(u < c.minX || u > c.maxX) && !(v < c.minY || v > c.maxY);
            if (ok) {
                const oslam_keypoint_t kp = c.keys[o];
                const int oct = kp.octave;
                if (bForward) { q.minLevel = oct; q.maxLevel = -1; }
                else if (bBackward) { q.minLevel = 0; q.maxLevel = oct; }
                else { q.minLevel = oct - 1; q.maxLevel = oct + 1; }
                q.u = u; q.v = v;
                q.ur = u - c.bf * invzc;
                q.radius = c.th * c.scale[oct];
                q.flags = 1 | ((hm & 2) ? 2 : 0);
                q.angle = kp.angle;
            }
        }
    }
    oslam_proj_query_t* dst = c.out + (long long)b * c.q_stride + i;
    const uint32_t* sd = (const uint32_t*)(c.mp_desc + o * 32);
    uint32_t* qd = (uint32_t*)q.desc;
#pragma unroll
    for (int w = 0; w < 8; w++) qd[w] = ok ? sd[w] : 0u;
    *dst = q;
}

}  // namespace oslam

using namespace oslam;

struct oslam_matcher {
    int device = 0, max_batch = 0, max_kps = 0, max_q = 0;
    size_t lds = 0;
    // device-owned
    int* d_q_match = nullptr; int* d_q_dist = nullptr; int* d_kp_match = nullptr; int* d_nm = nullptr; int* d_iters = nullptr; int* d_pairs = nullptr;
    uint32_t* d_cache = nullptr; int* d_ccount = nullptr;
    oslam_proj_query_t* d_queries = nullptr;   // internal query buffer (project_last / host API)
    long long* d_dbg = nullptr;
    int* d_nq = nullptr;
    // staging for the host API (batch 1)
    oslam_keypoint_t* d_kps = nullptr; float* d_ur = nullptr; uint8_t* d_desc = nullptr; uint8_t* d_blocked = nullptr;
    float* d_Xw = nullptr; uint8_t* d_has = nullptr; oslam_keypoint_t* d_lkeys = nullptr; uint8_t* d_ldesc = nullptr; float* d_T = nullptr;
    PinStage pin;   // host-pointer entry points
};

extern "C" {

void oslam_matcher_destroy(oslam_matcher_t* h) {
    if (!h) return;
    void* ptrs[] = {h->d_dbg, h->d_cache, h->d_ccount, h->d_q_match, h->d_q_dist, h->d_kp_match, h->d_nm, h->d_iters, h->d_pairs, h->d_queries, h->d_nq, h->d_kps, h->d_ur,
                    h->d_desc, h->d_blocked, h->d_Xw, h->d_has, h->d_lkeys, h->d_ldesc, h->d_T};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    h->pin.release();
    delete h;
}

int oslam_matcher_create(oslam_matcher_t** out, int max_batch, int max_keypoints, int max_queries, int device) {
    if (!out) { set_error("out is NULL"); return OSLAM_E_INVALID; }
    *out = nullptr;
    if (max_batch < 1 || max_keypoints < 1 || max_queries < 1) { set_error("oslam_matcher_create: invalid argument"); return OSLAM_E_INVALID; }
    if (max_keypoints > kMaxMatchKps) {
        set_error("max_keypoints %d > %d (keypoints + descriptors + grid of one frame must fit 160 KiB of LDS)", max_keypoints, kMaxMatchKps);
        return OSLAM_E_INVALID;
    }
    int ndev = oslam_device_count();
    if (ndev <= 0) { set_error("no HIP device visible: the gfx950 matcher has no CPU fallback"); return OSLAM_E_HIP; }
    if (device < 0 || device >= ndev) { set_error("device out of range"); return OSLAM_E_INVALID; }
    OSLAM_HIP_CHECK(hipSetDevice(device));
    oslam_matcher* h = new oslam_matcher();
    h->device = device; h->max_batch = max_batch; h->max_kps = max_keypoints; h->max_q = max_queries;
    h->lds = match_lds_bytes(max_keypoints);
    const size_t B = max_batch, NK = max_keypoints, NQ = max_queries;
#define ALLOC(ptr, bytes)                                                         \
    do {                                                                          \
        hipError_t e_ = hipMalloc((void**)&(ptr), (bytes));                       \
        if (e_ != hipSuccess) {                                                   \
            set_error("hipMalloc(%zu) failed: %s", (size_t)(bytes), hipGetErrorString(e_)); \
            oslam_matcher_destroy(h);                                             \
            return OSLAM_E_HIP;                                                   \
        }                                                                         \
    } while (0)
    ALLOC(h->d_q_match, B * NQ * 4); ALLOC(h->d_q_dist, B * NQ * 4); ALLOC(h->d_kp_match, B * NK * 4);
    ALLOC(h->d_cache, B * NQ * kCacheCap * 4); ALLOC(h->d_ccount, B * NQ * 4);
    ALLOC(h->d_nm, B * 4); ALLOC(h->d_iters, B * 4); ALLOC(h->d_pairs, B * 4); ALLOC(h->d_dbg, 64); ALLOC(h->d_queries, B * NQ * sizeof(oslam_proj_query_t)); ALLOC(h->d_nq, B * 4);
    ALLOC(h->d_kps, NK * sizeof(oslam_keypoint_t)); ALLOC(h->d_ur, NK * 4); ALLOC(h->d_desc, NK * 32); ALLOC(h->d_blocked, NK);
    ALLOC(h->d_Xw, NQ * 12); ALLOC(h->d_has, NQ); ALLOC(h->d_lkeys, NQ * sizeof(oslam_keypoint_t)); ALLOC(h->d_ldesc, NQ * 32); ALLOC(h->d_T, 32 * 4);
#undef ALLOC
    OSLAM_HIP_CHECK(hipFuncSetAttribute((const void*)k_search_window, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds));
    *out = h;
    return OSLAM_OK;
}

static int check_frames(const oslam_matcher* h, const oslam_match_frames_t* f, int batch) {
    if (!h || !f) { set_error("NULL argument"); return OSLAM_E_INVALID; }
    if (batch < 1 || batch > h->max_batch) { set_error("batch %d outside [1,%d]", batch, h->max_batch); return OSLAM_E_INVALID; }
    if (!f->keysUn || !f->desc) { set_error("keysUn/desc NULL"); return OSLAM_E_INVALID; }
    if (f->kp_stride < 1 || f->kp_stride > h->max_kps * 64) { set_error("bad kp_stride"); return OSLAM_E_INVALID; }
    if (!f->n_kps && (f->n_kps_const < 0 || f->n_kps_const > h->max_kps || f->n_kps_const > f->kp_stride)) { set_error("n_kps %d exceeds capacity %d", f->n_kps_const, h->max_kps); return OSLAM_E_CAPACITY; }
    if (!(f->maxX > f->minX) || !(f->maxY > f->minY)) { set_error("empty image bounds"); return OSLAM_E_INVALID; }
    return OSLAM_OK;
}

static int search_impl(oslam_matcher_t* h, const oslam_match_frames_t* f, const oslam_proj_query_t* d_queries, int q_stride,
                       const int32_t* d_n_queries, int n_queries_const, int batch, float nnratio, int use_ratio, int check_ori,
                       int th_high, const float* invLevelSigma2, int nlevels, void* stream);

int oslam_match_search_batch_device(oslam_matcher_t* h, const oslam_match_frames_t* f, const oslam_proj_query_t* d_queries,
                                    int q_stride, const int32_t* d_n_queries, int n_queries_const, int batch, float nnratio,
                                    int use_ratio, int check_ori, int th_high, void* stream) {
    return search_impl(h, f, d_queries, q_stride, d_n_queries, n_queries_const, batch, nnratio, use_ratio, check_ori, th_high, nullptr, 0, stream);
}

int oslam_match_fuse_batch_device(oslam_matcher_t* h, const oslam_match_frames_t* f, const oslam_proj_query_t* d_queries, int q_stride,
                                  const int32_t* d_n_queries, int n_queries_const, int batch, const float* invLevelSigma2, int nlevels,
                                  void* stream) {
    if (!invLevelSigma2 || nlevels < 1 || nlevels > OSLAM_MAX_LEVELS) { set_error("bad invLevelSigma2/nlevels"); return OSLAM_E_INVALID; }
    return search_impl(h, f, d_queries, q_stride, d_n_queries, n_queries_const, batch, 0.f, 0, 0, 50, invLevelSigma2, nlevels, stream);
}

static int search_impl(oslam_matcher_t* h, const oslam_match_frames_t* f, const oslam_proj_query_t* d_queries, int q_stride,
                       const int32_t* d_n_queries, int n_queries_const, int batch, float nnratio, int use_ratio, int check_ori,
                       int th_high, const float* invLevelSigma2, int nlevels, void* stream) {
    int rc = check_frames(h, f, batch);
    if (rc) return rc;
    if (!d_queries) d_queries = h->d_queries;
    if (q_stride < 1 || q_stride > h->max_q) { set_error("q_stride %d outside [1,%d]", q_stride, h->max_q); return OSLAM_E_INVALID; }
    if (!d_n_queries && (n_queries_const < 0 || n_queries_const > q_stride)) { set_error("n_queries exceeds q_stride"); return OSLAM_E_CAPACITY; }
    OSLAM_HIP_CHECK(hipSetDevice(h->device));
    MatchCtx c;
    c.keysUn = f->keysUn; c.kp_stride = f->kp_stride; c.uRight = f->uRight; c.desc = f->desc; c.blocked = f->blocked;
    c.n_kps = f->n_kps; c.n_kps_const = f->n_kps_const;
    c.minX = f->minX; c.minY = f->minY;
    c.invW = (float)kGridCols / (float)(f->maxX - f->minX);   // src/Frame.cc:160-161
    c.invH = (float)kGridRows / (float)(f->maxY - f->minY);
    c.q = d_queries; c.q_stride = q_stride; c.n_q = d_n_queries; c.n_q_const = n_queries_const;
    c.nnratio = nnratio; c.use_ratio = use_ratio; c.check_ori = check_ori; c.th_high = th_high;
    c.fuse = invLevelSigma2 ? 1 : 0;
    for (int i = 0; i < OSLAM_MAX_LEVELS; i++) c.invSigma2[i] = (invLevelSigma2 && i < nlevels) ? invLevelSigma2[i] : 0.f;
    c.q_match = h->d_q_match; c.q_dist = h->d_q_dist; c.kp_match = h->d_kp_match; c.nmatches = h->d_nm; c.iters = h->d_iters; c.pairs = h->d_pairs; c.dbg = h->d_dbg;
    c.cache = h->d_cache; c.ccount = h->d_ccount;
    // per-frame output strides equal the input strides; outputs were sized for max_q / max_kps
    if ((size_t)f->kp_stride > (size_t)h->max_kps) { set_error("kp_stride %d > max_keypoints %d", f->kp_stride, h->max_kps); return OSLAM_E_CAPACITY; }
    hipLaunchKernelGGL(k_search_window, dim3(batch), dim3(kMatchThreads), h->lds, (hipStream_t)stream, c, h->max_kps);
    OSLAM_HIP_CHECK(hipGetLastError());
    return OSLAM_OK;
}

int oslam_match_project_last_batch_device(oslam_matcher_t* h, const oslam_match_last_t* last, const float* d_Tcw,
                                          const float* d_Tlw, const oslam_camera_t* cam, const oslam_match_frames_t* cur,
                                          const float* scaleFactors, int nlevels, float th, int bMono, int batch,
                                          void* stream) {
    if (!h || !last || !d_Tcw || !d_Tlw || !cam || !cur || !scaleFactors) { set_error("NULL argument"); return OSLAM_E_INVALID; }
    if (batch < 1 || batch > h->max_batch || nlevels < 1 || nlevels > OSLAM_MAX_LEVELS) { set_error("bad batch/nlevels"); return OSLAM_E_INVALID; }
    if (last->kp_stride > h->max_q || (!last->n_kps && last->n_kps_const > last->kp_stride)) { set_error("last frame exceeds max_queries %d", h->max_q); return OSLAM_E_CAPACITY; }
    OSLAM_HIP_CHECK(hipSetDevice(h->device));
    ProjectCtx c;
    c.Xw = last->Xw; c.has_mp = last->has_mp; c.keys = last->keys; c.mp_desc = last->mp_desc; c.kp_stride = last->kp_stride;
    c.n_last = last->n_kps; c.n_last_const = last->n_kps_const;
    c.Tcw = d_Tcw; c.Tlw = d_Tlw;
    c.fx = cam->fx; c.fy = cam->fy; c.cx = cam->cx; c.cy = cam->cy; c.bf = cam->bf; c.b = cam->b;
    c.minX = cur->minX; c.minY = cur->minY; c.maxX = cur->maxX; c.maxY = cur->maxY;
    for (int i = 0; i < OSLAM_MAX_LEVELS; i++) c.scale[i] = i < nlevels ? scaleFactors[i] : 0.f;
    c.th = th; c.bMono = bMono;
    c.out = h->d_queries; c.q_stride = last->kp_stride; c.n_q = h->d_nq;
    hipLaunchKernelGGL(k_project_last, dim3(div_up(last->kp_stride, 256), batch), dim3(256), 0, (hipStream_t)stream, c);
    OSLAM_HIP_CHECK(hipGetLastError());
    return OSLAM_OK;
}

int oslam_match_results_device(const oslam_matcher_t* h, const int32_t** q_match, const int32_t** q_dist, const int32_t** kp_match,
                               const int32_t** nmatches, const oslam_proj_query_t** queries, const int32_t** n_queries) {
    if (!h) { set_error("NULL handle"); return OSLAM_E_INVALID; }
    if (q_match) *q_match = h->d_q_match;
    if (q_dist) *q_dist = h->d_q_dist;
    if (kp_match) *kp_match = h->d_kp_match;
    if (nmatches) *nmatches = h->d_nm;
    if (queries) *queries = h->d_queries;
    if (n_queries) *n_queries = h->d_nq;
    return OSLAM_OK;
}

int oslam_match_hamming_pairs(oslam_matcher_t* h, int batch, int64_t* total) {
    if (!h || !total || batch < 1 || batch > h->max_batch) { set_error("bad argument"); return OSLAM_E_INVALID; }
    OSLAM_HIP_CHECK(hipSetDevice(h->device));
    std::vector<int> v(batch);
    OSLAM_HIP_CHECK(hipDeviceSynchronize());
    OSLAM_HIP_CHECK(hipMemcpy(v.data(), h->d_pairs, (size_t)batch * 4, hipMemcpyDeviceToHost));
    int64_t t = 0;
    for (int x : v) t += x;
    *total = t;
    return OSLAM_OK;
}

int oslam_match_debug_counters(oslam_matcher_t* h, long long out[8], int reset) {
    if (!h || !out) { set_error("bad argument"); return OSLAM_E_INVALID; }
    OSLAM_HIP_CHECK(hipMemcpy(out, h->d_dbg, 64, hipMemcpyDeviceToHost));
    if (reset) OSLAM_HIP_CHECK(hipMemset(h->d_dbg, 0, 64));
    return OSLAM_OK;
}

int oslam_match_fetch(oslam_matcher_t* h, int b, int q_stride, int n_q, int kp_stride, int n_kps, int32_t* q_match, int32_t* q_dist,
                      int32_t* kp_match, int32_t* nmatches, int32_t* iterations, void* stream) {
    if (!h || b < 0 || b >= h->max_batch) { set_error("bad argument"); return OSLAM_E_INVALID; }
    OSLAM_HIP_CHECK(hipSetDevice(h->device));
    OSLAM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    int nm = 0;
    OSLAM_HIP_CHECK(hipMemcpy(&nm, h->d_nm + b, 4, hipMemcpyDeviceToHost));
    if (nm < 0) { set_error("matcher kernel rejected frame %d (keypoints or queries exceed the handle capacity)", b); return OSLAM_E_CAPACITY; }
    if (nmatches) *nmatches = nm;
    if (iterations) OSLAM_HIP_CHECK(hipMemcpy(iterations, h->d_iters + b, 4, hipMemcpyDeviceToHost));
    if (q_match && n_q > 0) OSLAM_HIP_CHECK(hipMemcpy(q_match, h->d_q_match + (size_t)b * q_stride, (size_t)n_q * 4, hipMemcpyDeviceToHost));
    if (q_dist && n_q > 0) OSLAM_HIP_CHECK(hipMemcpy(q_dist, h->d_q_dist + (size_t)b * q_stride, (size_t)n_q * 4, hipMemcpyDeviceToHost));
    if (kp_match && n_kps > 0) OSLAM_HIP_CHECK(hipMemcpy(kp_match, h->d_kp_match + (size_t)b * kp_stride, (size_t)n_kps * 4, hipMemcpyDeviceToHost));
    return OSLAM_OK;
}

// results of frame 0 through the pinned block: five async copies, one synchronisation
static int fetch_host(oslam_matcher* h, int n_q, int n_kps, int32_t* q_match, int32_t* q_dist, int32_t* kp_match, int32_t* nmatches) {
    uint8_t *a_nm = nullptr, *a_qm = nullptr, *a_qd = nullptr, *a_km = nullptr;
    int rc;
    if ((rc = h->pin.download(h->d_nm, 4, &a_nm))) return rc;
    if (q_match && n_q > 0 && (rc = h->pin.download(h->d_q_match, (size_t)n_q * 4, &a_qm))) return rc;
    if (q_dist && n_q > 0 && (rc = h->pin.download(h->d_q_dist, (size_t)n_q * 4, &a_qd))) return rc;
    if (kp_match && n_kps > 0 && (rc = h->pin.download(h->d_kp_match, (size_t)n_kps * 4, &a_km))) return rc;
    OSLAM_HIP_CHECK(hipStreamSynchronize(nullptr));
    int nm;
    memcpy(&nm, a_nm, 4);
    if (nm < 0) { set_error("matcher kernel rejected the frame (keypoints or queries exceed the handle capacity)"); return OSLAM_E_CAPACITY; }
    if (nmatches) *nmatches = nm;
    if (a_qm) memcpy(q_match, a_qm, (size_t)n_q * 4);
    if (a_qd) memcpy(q_dist, a_qd, (size_t)n_q * 4);
    if (a_km) memcpy(kp_match, a_km, (size_t)n_kps * 4);
    return OSLAM_OK;
}

static int stage_frame(oslam_matcher* h, int N, const oslam_keypoint_t* keysUn, const float* uRight, const uint8_t* desc,
                       const uint8_t* blocked, const float bounds[4], oslam_match_frames_t* f) {
    if (N < 0 || N > h->max_kps) { set_error("%d keypoints > capacity %d", N, h->max_kps); return OSLAM_E_CAPACITY; }
    // one pinned block sized for everything a host-pointer call moves (frame, queries / last frame, results)
    int rc = h->pin.reserve_total((size_t)h->max_kps * (sizeof(oslam_keypoint_t) + 32 + 4 + 1 + 4) +
                                  (size_t)h->max_q * (sizeof(oslam_proj_query_t) + 12 + 1 + sizeof(oslam_keypoint_t) + 32 + 8) + 64 * 1024);
    if (rc) return rc;
    h->pin.reset();
    if (N > 0) {
        if ((rc = h->pin.upload(h->d_kps, keysUn, (size_t)N * sizeof(oslam_keypoint_t))) || (rc = h->pin.upload(h->d_desc, desc, (size_t)N * 32))) return rc;
        if (uRight && (rc = h->pin.upload(h->d_ur, uRight, (size_t)N * 4))) return rc;
        if (blocked && (rc = h->pin.upload(h->d_blocked, blocked, (size_t)N))) return rc;
    }
    f->keysUn = h->d_kps; f->kp_stride = h->max_kps; f->uRight = uRight ? h->d_ur : nullptr; f->desc = h->d_desc;
    f->blocked = blocked ? h->d_blocked : nullptr; f->n_kps = nullptr; f->n_kps_const = N;
    f->minX = bounds[0]; f->minY = bounds[1]; f->maxX = bounds[2]; f->maxY = bounds[3];
    return OSLAM_OK;
}

int oslam_match_search_by_projection(oslam_matcher_t* h, int N, const oslam_keypoint_t* keysUn, const float* uRight,
                                     const uint8_t* desc, const uint8_t* blocked, const float bounds[4],
                                     const oslam_proj_query_t* queries, int M, float nnratio, int use_ratio, int check_ori,
                                     int32_t* q_match, int32_t* q_dist, int32_t* kp_match, int32_t* nmatches) {
    if (!h || !bounds || (N > 0 && (!keysUn || !desc)) || (M > 0 && !queries)) { set_error("NULL argument"); return OSLAM_E_INVALID; }
    if (M < 0 || M > h->max_q) { set_error("%d queries > capacity %d", M, h->max_q); return OSLAM_E_CAPACITY; }
    OSLAM_HIP_CHECK(hipSetDevice(h->device));
    oslam_match_frames_t f;
    int rc = stage_frame(h, N, keysUn, uRight, desc, blocked, bounds, &f);
    if (rc) return rc;
    if (M > 0 && (rc = h->pin.upload(h->d_queries, queries, (size_t)M * sizeof(oslam_proj_query_t)))) return rc;
    rc = oslam_match_search_batch_device(h, &f, h->d_queries, h->max_q, nullptr, M, 1, nnratio, use_ratio, check_ori, 100, nullptr);
    if (rc) return rc;
    return fetch_host(h, M, N, q_match, q_dist, kp_match, nmatches);
}

int oslam_match_project_last_frame(oslam_matcher_t* h, int N, const oslam_keypoint_t* keysUn, const float* uRight, const uint8_t* desc,
                                   const uint8_t* blocked, const float bounds[4], int Nlast, const float* Xw, const uint8_t* has_mp,
                                   const oslam_keypoint_t* last_keys, const uint8_t* mp_desc, const float Tcw[16], const float Tlw[16],
                                   const oslam_camera_t* cam, const float* scaleFactors, int nlevels, float th, int bMono,
                                   int check_ori, int32_t* q_match, int32_t* q_dist, int32_t* kp_match, int32_t* nmatches) {
    if (!h || !bounds || !Tcw || !Tlw || !cam || !scaleFactors) { set_error("NULL argument"); return OSLAM_E_INVALID; }
    if (Nlast < 0 || Nlast > h->max_q) { set_error("%d last-frame points > capacity %d", Nlast, h->max_q); return OSLAM_E_CAPACITY; }
    OSLAM_HIP_CHECK(hipSetDevice(h->device));
    oslam_match_frames_t f;
    int rc = stage_frame(h, N, keysUn, uRight, desc, blocked, bounds, &f);
    if (rc) return rc;
    if (Nlast > 0) {
        if ((rc = h->pin.upload(h->d_Xw, Xw, (size_t)Nlast * 12)) || (rc = h->pin.upload(h->d_has, has_mp, (size_t)Nlast)) ||
            (rc = h->pin.upload(h->d_lkeys, last_keys, (size_t)Nlast * sizeof(oslam_keypoint_t))) || (rc = h->pin.upload(h->d_ldesc, mp_desc, (size_t)Nlast * 32)))
            return rc;
    }
    if ((rc = h->pin.upload(h->d_T, Tcw, 64)) || (rc = h->pin.upload(h->d_T + 16, Tlw, 64))) return rc;
    oslam_match_last_t last;
    last.Xw = h->d_Xw; last.has_mp = h->d_has; last.keys = h->d_lkeys; last.mp_desc = h->d_ldesc;
    last.kp_stride = h->max_q; last.n_kps = nullptr; last.n_kps_const = Nlast;
    rc = oslam_match_project_last_batch_device(h, &last, h->d_T, h->d_T + 16, cam, &f, scaleFactors, nlevels, th, bMono, 1, nullptr);
    if (rc) return rc;
    rc = oslam_match_search_batch_device(h, &f, nullptr, h->max_q, nullptr, Nlast, 1, 0.f, 0, check_ori, 100, nullptr);
    if (rc) return rc;
    return fetch_host(h, Nlast, N, q_match, q_dist, kp_match, nmatches);
}

int oslam_match_fuse_search(oslam_matcher_t* h, int N, const oslam_keypoint_t* keysUn, const float* uRight, const uint8_t* desc,
                            const float bounds[4], const oslam_proj_query_t* queries, int M, const float* invLevelSigma2, int nlevels,
                            int32_t* q_match, int32_t* q_dist, int32_t* n_fused) {
    if (!h || !bounds || (N > 0 && (!keysUn || !desc)) || (M > 0 && !queries) || !invLevelSigma2) { set_error("NULL argument"); return OSLAM_E_INVALID; }
    if (M < 0 || M > h->max_q) { set_error("%d queries > capacity %d", M, h->max_q); return OSLAM_E_CAPACITY; }
    OSLAM_HIP_CHECK(hipSetDevice(h->device));
    oslam_match_frames_t f;
    int rc = stage_frame(h, N, keysUn, uRight, desc, nullptr, bounds, &f);
    if (rc) return rc;
    if (M > 0 && (rc = h->pin.upload(h->d_queries, queries, (size_t)M * sizeof(oslam_proj_query_t)))) return rc;
    rc = oslam_match_fuse_batch_device(h, &f, h->d_queries, h->max_q, nullptr, M, 1, invLevelSigma2, nlevels, nullptr);
    if (rc) return rc;
    return fetch_host(h, M, 0, q_match, q_dist, nullptr, n_fused);
}

int oslam_match_debug_get_queries(oslam_matcher_t* h, int b, int q_stride, int n, oslam_proj_query_t* out) {
    if (!h || !out) { set_error("NULL argument"); return OSLAM_E_INVALID; }
    OSLAM_HIP_CHECK(hipSetDevice(h->device));
    OSLAM_HIP_CHECK(hipDeviceSynchronize());
    if (n > 0) OSLAM_HIP_CHECK(hipMemcpy(out, h->d_queries + (size_t)b * q_stride, (size_t)n * sizeof(oslam_proj_query_t), hipMemcpyDeviceToHost));
    return OSLAM_OK;
}

}  // extern "C"
