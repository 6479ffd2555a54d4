// gfx950 motion-only bundle adjustment: Optimizer::PoseOptimization (reference
// src/Optimizer.cc:239-451) with the g2o Levenberg-Marquardt loop it drives,
// one 256-thread workgroup per frame, whole 4 x optimize(10) schedule in a single launch.
// Per-edge residuals / Jacobians / Huber weights in fp64, block reduction of the 6x6 normal
// equations (21 + 6 + 1 doubles), in-register 6x6 Cholesky, everything wave-uniform after the
// reduction so no broadcast is needed.
#include <hip/hip_runtime.h>
#include <string.h>
#include <stdint.h>

#include <algorithm>

#include "common.h"
#include "se3_math.h"
#include "lane_ops.h"

namespace oslam {

#ifndef OSLAM_POSE_THREADS
#define OSLAM_POSE_THREADS 128
#endif
constexpr int kPoseThreads = OSLAM_POSE_THREADS;   // threads per frame.  Round 4: 128 (two wavefronts), four frames per CU — with the lean passes the card-filling rate is 0.72 us per frame
                                                   // against 0.90 with 256 threads / two frames per CU and 0.78 with 64 / eight (tools/pose_prof.py, 4096 frames per launch)
constexpr int kPoseWaves = kPoseThreads / 64;
constexpr int kRedN = 29;   // 21 H (upper) + 6 b + chi + 1 spare

// Semantic constraints of ObjectOptimizer::PoseOptimization2 (reference src/ObjectOptimizer.cc:624-1240)
struct SemCtx {
    int nObj;
    const short2* area;        // BOUNDARY pixels of the mask (== 255 with a 4-neighbour that is not, or on the image border) as (col,row), row-major scan order per object
    const int* area_start;     // [nObj+1]
    const int* row_start;      // [nObj][H]: boundary pixels of the object in the rows before a row (k_mask_rowscan), relative to area_start[o]
    const unsigned long long* bits; const int* bits_index; int WB;   // optional one-bit-per-pixel masks (object o = bitmap bits_index[o]): then no mask byte is read
    const uint8_t* masks; const uint8_t* const* mask_ptrs; int H, W, pitch;   // the masks themselves (mask o = mask_ptrs[o] or masks + o*H*pitch), for the pixels around a query
    int nObjMp; const float* objmp_Xw; const int* objmp_obj;          // object map points, object-major
    int nJoint; const int* joint_kp; const int* joint_obj;             // M_joint candidates (:721-726)
    const float* kp_uv;        // [N][2] mvKeysUn[i].pt
    float minX, minY, maxX, maxY, invSigma2_0;
    // semantic edge store (capacity nJoint + nObjMp): M_joint edges first, then M_semantic
    float* e_Xw; float* e_obs; uint8_t* e_level; double* e_chi2; int* e_obj; uint8_t* e_out; int* e_tmp;
    int* nSem;                 // [B] semantic constraints used (nSemNum, :1232)
    const oslam_sem_frame_t* frames;   // batch form: frame b uses objects [obj0, obj0 + nObj) of the pools above, its own slice of the objmp / joint /
                                       // edge-store pools, and the keypoint coordinates of its obs rows (kp_uv == nullptr)
};

struct PoseCtx {
    const float* Tcw;        // [B][16]
    const float* Xw;         // [B][stride][3]
    const float* obs;        // [B][stride][3]  (u, v, uR; uR < 0 => monocular edge)
    const float* invSigma2;  // [B][stride]
    const uint8_t* has_mp;   // [B][stride]
    const int* n; int n_const; int stride;
    float fx, fy, cx, cy, bf;
    float* Tcw_out;          // [B][16]
    uint8_t* outlier;        // [B][stride]
    int* n_inliers;          // [B]
    int* stats;              // [B][2] LM iterations, trials
    double* trace; int trace_cap;   // optional LM trace of frame 0: trace[0] = records written, then [cap][6] = (F before, F of the trial, rho, lambda, accepted, first trial of a round)
    SemCtx sem;
};

__device__ __forceinline__ double wave_sum(double v) { return wave_sum_xor(v); }   // (lane_ops.h: same operand pairs as the __shfl_xor butterfly, no LDS crossbar)

// Block-wide sums of NV doubles; every thread returns the same totals (fixed summation order).
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* s_red /* [2][kPoseWaves][kRedN] */, int& phase) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double* buf = s_red + phase * kPoseWaves * kRedN;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const double s = wave_sum(v[i]);
        if (lane == 0) buf[wv * kRedN + i] = s;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; i++) {
        double s = buf[i];
#pragma unroll
        for (int w = 1; w < kPoseWaves; w++) s += buf[w * kRedN + i];
        v[i] = s;
    }
    phase ^= 1;   // next reduction uses the other buffer: one barrier per reduction is enough
}

// Exact nearest mask pixel under FLANN's float L2 (squared) over ALL pixels == 255 (the pcl cloud of reference src/ObjectOptimizer.cc:699-710); ties ->
// first in row-major scan order, i.e. smaller (row, col).  Only the mask's boundary pixels and the (at most four) pixels of the unit cell around the
// query can be that pixel: an interior pixel more than half a pixel away in x or y has a 4-neighbour inside the mask that is strictly nearer, and a
// pixel within one pixel in both x and y is one of the floor / ceil combinations.  (Pixel minus query is exact in float for |difference| <= 1.)
// px = col | row << 15.
// cutoff: the caller only distinguishes distances up to it (d2 < 10 / d2 > 10 gates of the M_semantic creation and the re-gating): rows whose vertical distance
// alone exceeds it are not visited, and a non-empty mask without a pixel that near reports d2 = +inf (px = 0) — exact whenever the true minimum is <= cutoff.
__device__ __forceinline__ bool mask_nearest(const SemCtx& sm, int o, float u, float v, int& px, float& d2, const float cutoff = __builtin_inff()) {
    const int s0 = sm.area_start[o], s1 = sm.area_start[o + 1];
    if (s1 <= s0) return false;   // no boundary pixel = empty mask
    float best = 0;
    int bx = -1, by = -1;
    // the candidate with the smallest (distance, row, col) wins: the first minimum in row-major scan order
    auto consider = [&](int x, int y) {
        const float dx = (float)x - u, dy = (float)y - v;
        float d = 0;
        d += dx * dx;
        d += dy * dy;
        if (bx < 0 || d < best || (d == best && (y < by || (y == by && x < bx)))) { best = d; bx = x; by = y; }
    };
    const unsigned long long* Bm = sm.bits ? sm.bits + (long long)sm.bits_index[o] * sm.H * sm.WB : nullptr;
    const uint8_t* m = Bm ? nullptr : (sm.mask_ptrs ? sm.mask_ptrs[o] : sm.masks + (long long)o * sm.H * sm.pitch);
    const float fu = floorf(u), fv = floorf(v);
    if (fu >= -1.f && fu < (float)sm.W && fv >= -1.f && fv < (float)sm.H) {   // unit cell first: a query inside the mask starts with best <= 2
        const int x0 = (int)fu, y0 = (int)fv;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int x = x0 + (k & 1), y = y0 + (k >> 1);
            if (x < 0 || y < 0 || x >= sm.W || y >= sm.H) continue;
            if (Bm ? !((Bm[(long long)y * sm.WB + (x >> 6)] >> (x & 63)) & 1ull) : m[(long long)y * sm.pitch + x] != 255) continue;
            consider(x, y);
        }
    }
    if (!(fabsf(v) < 1.0e6f)) {   // no usable row order (never seen on this path): the whole list
        for (int i = s0; i < s1; i++) { const short2 p = sm.area[i]; consider(p.x, p.y); }
    } else {
        // Boundary pixels row by row, outwards from the query's row in both directions: a row whose vertical distance alone exceeds the best distance so far
        // cannot hold the minimum (d = fl(fl(dx^2) + fl(dy^2)) >= fl(dy^2) in float; equality still scans the row, for the tie rule), and the vertical
        // distance grows monotonically in each direction.  For a query inside or next to the mask this visits two or three rows instead of the whole list.
        const int* rs = sm.row_start + (long long)o * sm.H;
        const int nb = s1 - s0;
        int yu = min(sm.H - 1, (int)fv), yd = max(0, (int)fv + 1);
        bool up = yu >= 0, dn = yd < sm.H;
        while (up || dn) {
            if (up) {
                const float dy = (float)yu - v;
                if ((bx >= 0 && dy * dy > best) || dy * dy > cutoff) up = false;
                else {
                    const int a = rs[yu], e = yu + 1 < sm.H ? rs[yu + 1] : nb;
                    for (int i = s0 + a; i < s0 + e; i++) consider(sm.area[i].x, yu);
                    if (--yu < 0) up = false;
                }
            }
            if (dn) {
                const float dy = (float)yd - v;
                if ((bx >= 0 && dy * dy > best) || dy * dy > cutoff) dn = false;
                else {
                    const int a = rs[yd], e = yd + 1 < sm.H ? rs[yd + 1] : nb;
                    for (int i = s0 + a; i < s0 + e; i++) consider(sm.area[i].x, yd);
                    if (++yd >= sm.H) dn = false;
                }
            }
        }
    }
    if (bx < 0) { px = 0; d2 = __builtin_inff(); return true; }   // (only with a cutoff: nothing that near)
    px = bx | (by << 15);
    d2 = best;
    return true;
}

// cv::Mat R*P + t as a single cv::gemm (flags==0, len==3): float accumulation, then (float)(t0 + t) in double
__device__ __forceinline__ void project_f32(const float* T, const float* P, float Pc[3]) {
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const float t0 = T[r * 4] * P[0] + T[r * 4 + 1] * P[1] + T[r * 4 + 2] * P[2];
        Pc[r] = (float)((double)t0 + (double)T[r * 4 + 3]);
    }
}

// ---- per-edge arithmetic of the passes over the edges (round 4) -------------------------------------------------------------------------------------
// The kernel is bound by fp64 instruction issue once the card is full (1.3 us per frame whether a frame has 64, 128 or 256 threads), so the passes are written
// for instruction count: one reciprocal per edge by v_rcp_f64 + two Newton steps instead of IEEE divisions (~14 instructions each), the Huber square root and
// its quotient from one v_rsq_f64 + two Newton steps, fused multiply-adds, and the normal equations accumulated over the NON-ZERO entries of the Jacobian rows
// only (g2o's rows have J[4] = J[9] = J[16] = 0; the generic 3 x 6 x 6 triple product spent a third of its multiplications on exact zeros).  All of it is a few ulp
// away from the divided / unfused form: far inside the 1e-4 bar the parity tests hold (the float `invz` of the fork's stereo projection is kept).
// (rcp_nr, rsq_nr, PoseRt, edge_residual_fast, huber_fast, accumulate_row: se3_math.h, shared with the local-BA linearisation)
// the OnlyPose Jacobian rows (se3_math.h jac_pose_onlypose) from iz = 1 / z, accumulated into the normal equations with weight wi = rho' * info
__device__ __forceinline__ void accumulate_edge(const Cam& c, const double p[3], double iz, const double e[3], double wi, bool stereo, double (&acc)[kRedN]) {
#pragma clang fp contract(fast)
    double Ju[6], Jv[6], Jr[6];
    pose_jac_rows(c, p, iz, stereo, Ju, Jv, Jr);
    accumulate_row<0>(Ju, e[0], wi, acc);
    accumulate_row<1>(Jv, e[1], wi, acc);
    if (stereo) accumulate_row<2>(Jr, e[2], wi, acc);
}

#ifdef OSLAM_POSE_PROFILE
__device__ unsigned long long g_pose_prof[8];   // cycles of frame 0 per phase: build, sum28, solve+exp, eval, sum1, classify, prologue, total
#define PSTAMP(i) do { if (b == 0 && tid == 0) { const long long t_ = clock64(); pacc[i] += (unsigned long long)(t_ - tp_); tp_ = t_; } } while (0)
#else
#define PSTAMP(i) do { } while (0)
#endif

// Block-wide sums of the 28 accumulators of the build pass (21 H + 6 b + chi2) into s_tot (fixed order).
// Transposed through LDS: every thread parks its 28 partials, 8 threads per value add 32 partials each and combine with three
// shuffle steps -- 28 writes + 32 reads per thread instead of the 168 double-precision shuffle steps of 28 wavefront butterflies.
// Neighbouring lanes are added first (one DPP step per value), so only every second thread parks a partial: the buffer is 30 KB instead of 59 KB and
// TWO frames fit into a CU's LDS beside their staged edges (the kernel's 214-235 registers allow two wavefronts per SIMD; one workgroup per CU left every
// SIMD with a single wavefront of dependent fp64 chains).
constexpr int kSumN = 28, kSumCols = kPoseThreads / 2, kSumPitch = kSumCols + 8;   // pitch = 8 mod 32 doubles: the value rows of a wavefront's reads fall on different banks
constexpr int kSumLanes = kSumN * 8 <= kPoseThreads ? 8 : (kSumN * 4 <= kPoseThreads ? 4 : 2);   // summing threads per value
static_assert(kSumN * kSumLanes <= kPoseThreads && kSumCols % kSumLanes == 0, "block_sum_wide: summing threads per value");
__device__ __forceinline__ void block_sum_wide(double (&v)[kRedN], double* s_part /* [kSumN][kSumPitch] */, double* s_tot /* [kSumN] */) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < kSumN; i++) {
        const double s = v[i] + lane_xor1(v[i]);   // (DPP quad_perm instead of two ds_bpermute per value)
        if (!(tid & 1)) s_part[i * kSumPitch + (tid >> 1)] = s;
    }
    __syncthreads();
    if (tid < kSumN * kSumLanes) {
        const double* row = s_part + (tid / kSumLanes) * kSumPitch + (tid % kSumLanes);
        double s = row[0];
#pragma unroll
        for (int k = 1; k < kSumCols / kSumLanes; k++) s += row[kSumLanes * k];
#pragma unroll
        for (int d = 1; d < kSumLanes; d <<= 1) s += d == 1 ? lane_xor1(s) : (d == 2 ? lane_xor2(s) : __shfl_xor(s, d, 64));
        if ((tid % kSumLanes) == 0) s_tot[tid / kSumLanes] = s;
    }
    __syncthreads();   // the totals are in s_tot: the callers read what they need (H is only needed by the wavefront that solves)
}

// STAGE: the edge data (Xw, obs, invSigma2: 28 B per edge) are copied into LDS once; every pass of the ~100 over the edges then reads them at LDS
// latency instead of paying a global-memory round trip per edge (one wavefront per SIMD: nothing else hides it).
// two wavefronts per SIMD (at most 256 registers): left to itself the compiler takes a 257th for the semantic variant, which halves the frames per CU
#ifndef OSLAM_POSE_WAVES_PER_EU
#define OSLAM_POSE_WAVES_PER_EU 2
#endif
#define POSE_OCC __attribute__((amdgpu_waves_per_eu(OSLAM_POSE_WAVES_PER_EU, OSLAM_POSE_WAVES_PER_EU)))
template <bool SEM, bool STAGE>
__global__ __launch_bounds__(kPoseThreads) POSE_OCC void k_pose_optimize(PoseCtx c) {
    const int b = blockIdx.x, tid = threadIdx.x;
#ifdef OSLAM_POSE_PROFILE
    unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tp_ = clock64();
    const long long tstart_ = tp_;
#endif
    const int N = c.n ? c.n[b] : c.n_const;
    const float* gXw = c.Xw + (long long)b * c.stride * 3;
    const float* gobs = c.obs + (long long)b * c.stride * 3;
    const float* ginv = c.invSigma2 + (long long)b * c.stride;
    const uint8_t* has = c.has_mp + (long long)b * c.stride;
    uint8_t* outl = c.outlier + (long long)b * c.stride;
    const float* T0f = c.Tcw + b * 16;

    extern __shared__ __align__(16) uint8_t smem[];
    double* s_chi2 = (double*)smem;                         // [stride] _error chi2 as last computed (may be stale)
    float* s_edge = (float*)(s_chi2 + c.stride);            // STAGE: [stride][3] Xw, [stride][3] obs, [stride] invSigma2
    uint8_t* s_level = (uint8_t*)(s_edge + (STAGE ? 7 * c.stride : 0));   // [stride] 0 active, 1 excluded, 255 no edge
    uint16_t* s_act = (uint16_t*)(s_level + ((c.stride + 1) & ~1));         // [stride] the active edges of the current round, ascending (compact_active)
    const float* Xw = STAGE ? s_edge : gXw;
    const float* obsp = STAGE ? s_edge + 3 * c.stride : gobs;
    const float* inv = STAGE ? s_edge + 6 * c.stride : ginv;
    if (STAGE) {
        for (int i = tid; i < 3 * N; i += kPoseThreads) { s_edge[i] = gXw[i]; s_edge[3 * c.stride + i] = gobs[i]; }
        for (int i = tid; i < N; i += kPoseThreads) s_edge[6 * c.stride + i] = ginv[i];
    }
    __shared__ double s_red[2 * kPoseWaves * kRedN];
    __shared__ double s_part[kSumN * kSumPitch], s_tot[kSumN];   // block_sum_wide (59 KB)
    constexpr int kCandN = 14;                                   // candidate pose (q, t), step x, solve ok
    constexpr int kMaxTrials = 10;                               // g2o's maxTrialsAfterFailure
    // Poses live in LDS, not in registers (they are wave-uniform and read a few times per iteration; the kernel sits at the 256-register limit of two wavefronts
    // per SIMD): the candidates of an iteration in s_cand[iteration parity], the input pose in s_T0, and `Tp` points at the current estimate — the input pose at
    // a round's start, then the accepted candidate (an iteration without an accepted trial ends the round, so the slot Tp points into is never the one the next
    // iteration's candidates are written to).
    __shared__ double s_cand[2 * kMaxTrials * kCandN];
    __shared__ double s_T0[8];
    constexpr int kMaxChunks = (11000 + 63) / 64;                // 64-edge chunks of the largest frame oslam_poseopt_create accepts
    __shared__ int s_coff[kMaxChunks + 1];
    const int wv = tid >> 6, lane = tid & 63;
    __shared__ int s_cnt[2];
    int phase = 0;

    const Cam cam = {(double)c.fx, (double)c.fy, (double)c.cx, (double)c.cy, (double)c.bf};
    const double deltaMono = (double)(float)sqrt(5.991), deltaStereo = (double)(float)sqrt(7.815);
    const float chi2Mono = 5.991f, chi2Stereo = 7.815f;

    if (tid == 0) { s_cnt[0] = 0; s_cnt[1] = 0; }
    __syncthreads();
    int ncorr = 0;
    for (int i = tid; i < N; i += kPoseThreads) {
        const bool h = has[i] != 0;
        s_level[i] = h ? 0 : 255;
        s_chi2[i] = 0;
        outl[i] = 0;   // mvbOutlier[i] = false (:289,:323); entries without a map point are reported 0
        if (h) ncorr++;
    }
    if (ncorr) atomicAdd(&s_cnt[0], ncorr);
    __syncthreads();
    const int nInitial = s_cnt[0];
    if (nInitial < 3) {   // reference :364-365: return 0, pose untouched
        if (tid < 16) c.Tcw_out[b * 16 + tid] = T0f[tid];
        if (tid == 0) { c.n_inliers[b] = 0; if (c.stats) { c.stats[b * 2] = 0; c.stats[b * 2 + 1] = 0; } if (SEM) c.sem.nSem[b] = 0; }
        return;
    }

    // Ordered compaction, the one pattern behind every list this kernel builds: emit(i, base + rank of i among the flagged indices below n), in index order, returns
    // base + their number to every thread.  Chunk ballots, a scan of the chunk counts by wavefront 0, ordered scatter; ends with a barrier (the emitted data and
    // s_coff are then free to use).  flag(i) is evaluated twice and must not change in between.
    auto ordered_scatter = [&](int n, int base, auto flag, auto emit) -> int {
        for (int i0 = 0; i0 < n; i0 += kMaxChunks * 64) {
            const int nn = min(n - i0, kMaxChunks * 64), nch = (nn + 63) >> 6;
            for (int ch = wv; ch < nch; ch += kPoseWaves) {
                const int i = i0 + ch * 64 + lane;
                const unsigned long long m = __ballot(i < i0 + nn && flag(i));
                if (lane == 0) s_coff[ch] = __popcll(m);
            }
            __syncthreads();
            if (wv == 0) {
                int carry = 0;
                for (int cb = 0; cb < nch; cb += 64) {
                    const int v = cb + lane < nch ? s_coff[cb + lane] : 0;
                    int incl = v;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
                    if (cb + lane < nch) s_coff[cb + lane] = carry + incl - v;
                    carry += __shfl(incl, 63, 64);
                }
                if (lane == 0) s_coff[nch] = carry;
            }
            __syncthreads();
            for (int ch = wv; ch < nch; ch += kPoseWaves) {
                const int i = i0 + ch * 64 + lane;
                const bool a = i < i0 + nn && flag(i);
                const unsigned long long m = __ballot(a);
                if (a) emit(i, base + s_coff[ch] + __popcll(m & ((1ull << lane) - 1ull)));
            }
            base += s_coff[nch];
            __syncthreads();
        }
        return base;
    };
    // The passes of a round visit the ACTIVE edges only (level 0: 40-70 % of a frame's keypoint slots hold a map point, fewer after the outlier rounds), so they
    // run over a compacted index list instead of masking idle lanes.  The list is ascending, so every thread's share — and with it the summation order — is a
    // function of the levels alone.  Callers: s_level complete and visible.
    auto compact_active = [&]() -> int {
        return ordered_scatter(N, 0, [&](int i) { return s_level[i] == 0; }, [&](int i, int pos) { s_act[pos] = (uint16_t)i; });
    };
    int nAct = 0;

    if (tid == 0) {
        const SE3 T0 = se3_from_T(T0f);
        for (int k = 0; k < 4; k++) s_T0[k] = T0.q[k];
        for (int k = 0; k < 3; k++) s_T0[4 + k] = T0.t[k];
    }
    const double* Tp = s_T0;   // (visible to the other threads after the barriers of the first compact_active)
    auto load_pose = [&](const double* q) -> SE3 { SE3 o; for (int k = 0; k < 4; k++) o.q[k] = q[k]; for (int k = 0; k < 3; k++) o.t[k] = q[4 + k]; return o; };
    int nBad = 0, tot_its = 0, tot_trials = 0;

    // ---- semantic edges (PoseOptimization2) ----
    SemCtx sm = c.sem;
    if (SEM && c.sem.frames) {
        const oslam_sem_frame_t f = c.sem.frames[b];
        const int e0 = f.objmp0 + f.joint0;
        sm.nObj = f.nObj; sm.area_start += f.obj0; sm.row_start += (long long)f.obj0 * sm.H;
        if (sm.bits) sm.bits_index += f.obj0; else sm.mask_ptrs += f.obj0;
        sm.nObjMp = f.nObjMp; sm.objmp_Xw += 3 * (long long)f.objmp0; sm.objmp_obj += f.objmp0;
        sm.nJoint = f.nJoint; sm.joint_kp += f.joint0; sm.joint_obj += f.joint0;
        sm.e_Xw += 3 * (long long)e0; sm.e_obs += 2 * (long long)e0; sm.e_level += e0; sm.e_chi2 += e0; sm.e_obj += e0; sm.e_out += e0; sm.e_tmp += e0;
        sm.nSem = c.sem.nSem;
    }
    __shared__ int s_nsem, s_ninit, s_semnum;
    int nsem = 0, ninit = 0;
    if (SEM) {
        // M_joint constraints (:719-767): NN of the keypoint in its object's mask, skipped if d^2 < 1
        for (int j = tid; j < sm.nJoint; j += kPoseThreads) {
            const int kp = sm.joint_kp[j], o = sm.joint_obj[j];
            int idx; float d2;
            int ok = 0;
            const float ku = sm.kp_uv ? sm.kp_uv[kp * 2] : gobs[kp * 3], kv = sm.kp_uv ? sm.kp_uv[kp * 2 + 1] : gobs[kp * 3 + 1];   // mvKeysUn[kp].pt
            if (mask_nearest(sm, o, ku, kv, idx, d2) && !(d2 < 1.0f)) ok = 1 + idx;
            sm.e_tmp[j] = ok;
        }
        __syncthreads();
        // ordered compaction (creation order of the reference)
        const int nj = ordered_scatter(sm.nJoint, 0, [&](int j) { return sm.e_tmp[j] != 0; }, [&](int j, int n) {
            const int kp = sm.joint_kp[j];
            const int px = sm.e_tmp[j] - 1;
            sm.e_Xw[n * 3] = Xw[kp * 3]; sm.e_Xw[n * 3 + 1] = Xw[kp * 3 + 1]; sm.e_Xw[n * 3 + 2] = Xw[kp * 3 + 2];
            sm.e_obs[n * 2] = (float)(px & 0x7FFF); sm.e_obs[n * 2 + 1] = (float)(px >> 15);
            sm.e_level[n] = 0; sm.e_obj[n] = sm.joint_obj[j]; sm.e_out[n] = 0;
        });
        if (tid == 0) { s_ninit = nj; s_nsem = nj; s_semnum = nj; }
        __syncthreads();
        nsem = s_nsem; ninit = s_ninit;
    }
    const double infoSem = (double)sm.invSigma2_0;

    // residual pass over the active edges at pose P: stores chi2 per edge, returns sum of robust chi2
    // Edge inputs of the passes.  Without staging they come through the L2 (and the semantic edges always do), and two wavefronts per SIMD do not hide that
    // latency behind ~60-130 instructions of arithmetic per edge: every pass requests the inputs of its NEXT edge before it works on the current one.
    struct EdgeIn { float x0, x1, x2, o0, o1, o2, iv; };
    auto load_edge = [&](int i) -> EdgeIn { return EdgeIn{Xw[i * 3], Xw[i * 3 + 1], Xw[i * 3 + 2], obsp[i * 3], obsp[i * 3 + 1], obsp[i * 3 + 2], inv[i]}; };
    struct SemIn { float x0, x1, x2, o0, o1; int lv; };
    auto load_sem = [&](int i) -> SemIn { return SemIn{sm.e_Xw[i * 3], sm.e_Xw[i * 3 + 1], sm.e_Xw[i * 3 + 2], sm.e_obs[i * 2], sm.e_obs[i * 2 + 1], (int)sm.e_level[i]}; };
    // for_edges(f): f(i, in) for the active edges of this thread; for_sem(f): f(i, in) for its active semantic edges
    auto for_edges = [&](auto f) {
        int j = tid, ic = 0;
        EdgeIn cur = {};
        if (j < nAct) { ic = s_act[j]; cur = load_edge(ic); }
        while (j < nAct) {
            const int jn = j + kPoseThreads;
            int in_ = 0;
            EdgeIn nxt = cur;
            if (jn < nAct) { in_ = s_act[jn]; nxt = load_edge(in_); }
            f(ic, cur);
            cur = nxt; ic = in_; j = jn;
        }
    };
    auto for_sem = [&](auto f) {
        int i = tid;
        SemIn cur = {};
        if (i < nsem) cur = load_sem(i);
        while (i < nsem) {
            const int in_ = i + kPoseThreads;
            SemIn nxt = cur;
            if (in_ < nsem) nxt = load_sem(in_);
            if (cur.lv == 0) f(i, cur);
            cur = nxt; i = in_;
        }
    };
    auto eval = [&](const SE3& Pq, bool robust) -> double {
        double F = 0;
        const PoseRt P = pose_rt(Pq);
        for_edges([&](int i, const EdgeIn& in) {
            const double X[3] = {(double)in.x0, (double)in.x1, (double)in.x2};
            const bool stereo = !(in.o2 < 0);
            const double ob[3] = {(double)in.o0, (double)in.o1, (double)in.o2};
            double p[3], e[3], iz;
            const double c2 = edge_residual_fast(cam, P, X, ob, stereo, (double)in.iv, p, e, iz);
            s_chi2[i] = c2;
            if (robust) {
                double r0, r1;
                huber_fast(c2, stereo ? deltaStereo : deltaMono, r0, r1);
                F += r0;
            } else
                F += c2;
        });
        if (SEM) {   // semantic edges keep their Huber kernel in every round
            for_sem([&](int, const SemIn& in) {
                const double X[3] = {(double)in.x0, (double)in.x1, (double)in.x2};
                const double ob[3] = {(double)in.o0, (double)in.o1, 0.0};
                double p[3], e[3], iz;
                const double c2 = edge_residual_fast(cam, P, X, ob, false, infoSem, p, e, iz);
                double r0, r1;
                huber_fast(c2, deltaMono, r0, r1);
                F += r0;
            });
        }
        return F;
    };

    PSTAMP(6);
    for (int it = 0; it < 4; it++) {
        const bool robust = it < 3;   // setRobustKernel(0) after round index 2 (:407,:436)
        Tp = s_T0;                    // every round restarts from the input pose (:377)
        // initializeOptimization(0): any active edge?
        nAct = compact_active();   // (s_level: written before the barrier that ended the previous round / the prologue)
        int nact = nAct;
        if (SEM) {
            int ns = 0;
            for (int i = tid; i < nsem; i += kPoseThreads) ns += (sm.e_level[i] == 0);
            double v[1] = {(double)ns};
            block_sum<1>(v, s_red, phase);
            nact += (int)v[0];
        }
        if (nact > 0) {
            double lambda = 0, ni = 2;
            for (int iter = 0; iter < 10; iter++) {
                // computeActiveErrors + activeRobustChi2 + buildSystem in one pass over the edges
                double acc[kRedN];
#pragma unroll
                for (int k = 0; k < kRedN; k++) acc[k] = 0;
                const PoseRt Trt = pose_rt(load_pose(Tp));
                for_edges([&](int i, const EdgeIn& in) {
                    const double X[3] = {(double)in.x0, (double)in.x1, (double)in.x2};
                    const bool stereo = !(in.o2 < 0);
                    const double ob[3] = {(double)in.o0, (double)in.o1, (double)in.o2};
                    const double info = (double)in.iv;
                    double p[3], e[3], iz;
                    const double c2 = edge_residual_fast(cam, Trt, X, ob, stereo, info, p, e, iz);
                    s_chi2[i] = c2;
                    double r0 = c2, w = 1.0;
                    if (robust) huber_fast(c2, stereo ? deltaStereo : deltaMono, r0, w);
                    acc[27] += r0;
                    accumulate_edge(cam, p, iz, e, w * info, stereo, acc);
                });
                if (SEM) {
                    for_sem([&](int, const SemIn& in) {
                        const double X[3] = {(double)in.x0, (double)in.x1, (double)in.x2};
                        const double ob[3] = {(double)in.o0, (double)in.o1, 0.0};
                        double p[3], e[3], iz;
                        const double c2 = edge_residual_fast(cam, Trt, X, ob, false, infoSem, p, e, iz);
                        double r0, w;
                        huber_fast(c2, deltaMono, r0, w);
                        acc[27] += r0;
                        accumulate_edge(cam, p, iz, e, w * infoSem, false, acc);
                    });
                }
                PSTAMP(0);
                block_sum_wide(acc, s_part, s_tot);
                PSTAMP(1);
                double currentChi = s_tot[27];
                double g[6];
                for (int a = 0; a < 6; a++) g[a] = s_tot[21 + a];
                if (iter == 0) {   // computeLambdaInit: 1e-5 * max |H_jj|
                    double md = 0;
                    for (int a = 0; a < 6; a++) md = fmax(fabs(s_tot[h_idx(a, a)]), md);
                    lambda = 1e-5 * md;
                    ni = 2;
                }
                double rho = 0;
                int qmax = 0;
                do {
                    // A rejected trial multiplies lambda by ni and doubles ni, so the damping of every trial this iteration can make (at most 10) is known in
                    // advance: LANE l of ONE wavefront solves for trial l (the 6x6 Cholesky and the exponential map are ~5 k cycles of dependent fp64 divisions,
                    // square roots, sin / cos — the same instruction stream whatever the damping, so ten candidates cost what one does), the candidates wait in LDS and are
                    // evaluated one after the other exactly as before.  (Rounds 1-3 had wavefront w solve for trial q + w: four instruction streams per group of
                    // four trials, three of them wasted whenever the first trial was accepted — which is the usual case — and the kernel is bound by issue.)
                    if (qmax == 0) {
                        if (wv == (tot_its % kPoseWaves)) {   // the wavefronts take the solves in turns: each SIMD of the CU hosts one of them, and the chain is issue-bound
                            const int tl = min(lane, kMaxTrials - 1);
                            double lw = lambda, nw = ni;
                            for (int k = 0; k < kMaxTrials - 1; k++)
                                if (k < tl) { lw *= nw; nw *= 2; }
                            double A[36], xw[6];
                            for (int a = 0; a < 6; a++)
                                for (int cc = a; cc < 6; cc++) { const double hv = s_tot[h_idx(a, cc)]; A[a * 6 + cc] = hv; A[cc * 6 + a] = hv; }
                            for (int a = 0; a < 6; a++) { A[a * 7] += lw; xw[a] = g[a]; }
                            const bool okw = solve6(A, xw);
                            if (!okw) for (int a = 0; a < 6; a++) xw[a] = 0;
                            const SE3 Tw = se3_mul(se3_exp(xw), load_pose(Tp));
                            if (lane < kMaxTrials) {
                                double* cd = s_cand + ((tot_its & 1) * kMaxTrials + lane) * kCandN;
                                for (int k = 0; k < 4; k++) cd[k] = Tw.q[k];
                                for (int k = 0; k < 3; k++) cd[4 + k] = Tw.t[k];
                                for (int k = 0; k < 6; k++) cd[7 + k] = xw[k];
                                cd[13] = okw ? 1.0 : 0.0;
                            }
                        }
                        __syncthreads();
                    }
                    double x[6];
                    SE3 Tn;
                    bool ok2;
                    {
                        const double* cd = s_cand + ((tot_its & 1) * kMaxTrials + qmax) * kCandN;
                        for (int k = 0; k < 4; k++) Tn.q[k] = cd[k];
                        for (int k = 0; k < 3; k++) Tn.t[k] = cd[4 + k];
                        for (int k = 0; k < 6; k++) x[k] = cd[7 + k];
                        ok2 = cd[13] != 0.0;
                    }
                    PSTAMP(2);
                    double v[1] = {eval(Tn, robust)};
                    PSTAMP(3);
                    block_sum<1>(v, s_red, phase);
                    PSTAMP(4);
                    double tempChi = v[0];
                    if (!ok2) tempChi = 1.7976931348623157e308;
                    rho = currentChi - tempChi;
                    double scale = 0;
                    for (int a = 0; a < 6; a++) scale += x[a] * (lambda * x[a] + g[a]);
                    scale += 1e-3;
                    rho /= scale;
                    const bool finite = (tempChi - tempChi) == 0;
                    if (c.trace && b == 0 && tid == 0) {
                        const int r = (int)c.trace[0];
                        if (r < c.trace_cap) {
                            double* t = c.trace + 1 + 6 * r;
                            t[0] = currentChi; t[1] = tempChi; t[2] = rho; t[3] = lambda; t[4] = (rho > 0 && finite) ? 1.0 : 0.0; t[5] = (iter == 0 && qmax == 0) ? 1.0 : 0.0;
                        }
                        c.trace[0] = r + 1;
                    }
                    if (rho > 0 && finite) {
                        double alpha = 1. - (2 * rho - 1) * (2 * rho - 1) * (2 * rho - 1);
                        alpha = fmin(alpha, 2. / 3.);
                        lambda *= fmax(1. / 3., alpha);
                        ni = 2;
                        currentChi = tempChi;
                        Tp = s_cand + ((tot_its & 1) * kMaxTrials + qmax) * kCandN;
                    } else {
                        lambda *= ni;
                        ni *= 2;
                    }
                    qmax++;
                    tot_trials++;
                } while (rho < 0 && qmax < 10);
                tot_its++;
                if (qmax == 10 || rho == 0) break;   // OptimizationAlgorithm::Terminate
            }
        }
        if (SEM) {
            float Pose[16];
            se3_to_T(load_pose(Tp), Pose);   // Converter::toCvMat(vSE3->estimate())
            // re-gate the M_joint edges (:928-973 after round 0, :1042-1098 afterwards)
            int dsem = 0;
            for (int i = tid; i < ninit; i += kPoseThreads) {
                const float Pw[3] = {sm.e_Xw[i * 3], sm.e_Xw[i * 3 + 1], sm.e_Xw[i * 3 + 2]};
                float Pc[3];
                project_f32(Pose, Pw, Pc);
                const float x = __fdiv_rn(Pc[0], Pc[2]), y = __fdiv_rn(Pc[1], Pc[2]);
                const float u = c.fx * x + c.cx, v = c.fy * y + c.cy;
                bool out;
                if (u < sm.minX || v < sm.minY || u > sm.maxX || v > sm.maxY) out = true;
                else {
                    int ni; float d2;
                    if (!mask_nearest(sm, sm.e_obj[i], u, v, ni, d2, 10.f)) continue;
                    out = d2 > 10;
                    if (!out) { sm.e_obs[i * 2] = u; sm.e_obs[i * 2 + 1] = v; }   // measurement := projection (:969-971)
                }
                if (out) {
                    sm.e_level[i] = 1;
                    if (!sm.e_out[i]) { sm.e_out[i] = 1; dsem--; }
                } else {
                    sm.e_level[i] = 0;
                    if (sm.e_out[i]) { sm.e_out[i] = 0; dsem++; }
                }
            }
            if (dsem) atomicAdd(&s_semnum, dsem);
            if (it == 0) {
                // M_semantic constraints (:978-1032)
                for (int m = tid; m < sm.nObjMp; m += kPoseThreads) {
                    const int o = sm.objmp_obj[m];
                    float Pc[3];
                    project_f32(Pose, sm.objmp_Xw + 3 * m, Pc);
                    const float x = __fdiv_rn(Pc[0], Pc[2]), y = __fdiv_rn(Pc[1], Pc[2]);
                    const float u = c.fx * x + c.cx, v = c.fy * y + c.cy;
                    int ni; float d2;
                    sm.e_tmp[m] = (mask_nearest(sm, o, u, v, ni, d2, 10.f) && d2 < 10) ? 1 + ni : 0;
                }
                __syncthreads();
                const int n0 = s_nsem;
                const int n1 = ordered_scatter(sm.nObjMp, n0, [&](int m) { return sm.e_tmp[m] != 0; }, [&](int m, int n) {
                    const int px = sm.e_tmp[m] - 1;
                    sm.e_Xw[n * 3] = sm.objmp_Xw[m * 3]; sm.e_Xw[n * 3 + 1] = sm.objmp_Xw[m * 3 + 1]; sm.e_Xw[n * 3 + 2] = sm.objmp_Xw[m * 3 + 2];
                    sm.e_obs[n * 2] = (float)(px & 0x7FFF); sm.e_obs[n * 2 + 1] = (float)(px >> 15);
                    sm.e_level[n] = 0; sm.e_obj[n] = sm.objmp_obj[m]; sm.e_out[n] = 0;
                });
                if (tid == 0) { s_semnum += n1 - n0; s_nsem = n1; }
            }
            __syncthreads();
            nsem = s_nsem;
        }
        // inlier / outlier classification (:380-438): excluded edges get a fresh error at the final
        // pose, active edges keep the error of the last LM trial (g2o's _error buffer)
        int bad = 0;
        const SE3 Tfin = load_pose(Tp);
        for (int i = tid; i < N; i += kPoseThreads) {
            const int lv = s_level[i];
            if (lv == 255) continue;
            const float ur = obsp[i * 3 + 2];
            const bool stereo = !(ur < 0);
            if (lv == 1) {
                const double X[3] = {(double)Xw[i * 3], (double)Xw[i * 3 + 1], (double)Xw[i * 3 + 2]};
                const double ob[3] = {(double)obsp[i * 3], (double)obsp[i * 3 + 1], (double)ur};
                double p[3], e[3];
                se3_map(Tfin, X, p);
                s_chi2[i] = edge_error(cam, p, ob, stereo, (double)inv[i], e);
            }
            const float chi2 = (float)s_chi2[i];
            if (chi2 > (stereo ? chi2Stereo : chi2Mono)) { s_level[i] = 1; outl[i] = 1; bad++; }
            else { s_level[i] = 0; outl[i] = 0; }
        }
        {
            double v[1] = {(double)bad};
            block_sum<1>(v, s_red, phase);
            nBad = (int)v[0];
        }
        PSTAMP(5);
        if (nInitial + nsem < 10) break;   // optimizer.edges().size()<10 (:440); semantic edges count too
    }

    if (tid == 0) {
        float To[16];
        se3_to_T(load_pose(Tp), To);
        for (int k = 0; k < 16; k++) c.Tcw_out[b * 16 + k] = To[k];
        c.n_inliers[b] = nInitial - nBad;
        if (c.stats) { c.stats[b * 2] = tot_its; c.stats[b * 2 + 1] = tot_trials; }
        if (SEM) sm.nSem[b] = s_semnum;
#ifdef OSLAM_POSE_PROFILE
        if (b == 0) { pacc[7] = (unsigned long long)(clock64() - tstart_); for (int k = 0; k < 8; k++) g_pose_prof[k] = pacc[k]; }
#endif
    }
}

// Object2D mask -> ordered list of its 255-pixels (pcl cloud order of reference src/ObjectOptimizer.cc:699-710).
// Mask o is masks + o*H*pitch (ptrs == nullptr) or ptrs[o], rows `pitch` bytes apart.  Three steps: per-row counts, per-object exclusive
// scan of the rows + a scan over the objects (area_start), then the ordered fill.
__device__ __forceinline__ const uint8_t* mask_row(const uint8_t* masks, const uint8_t* const* ptrs, int o, int H, int pitch, int row) {
    return (ptrs ? ptrs[o] : masks + (long long)o * H * pitch) + (long long)row * pitch;
}

// boundary pixel of the mask: == 255 and on the image border or next to a 4-neighbour that is not 255 (see mask_nearest)
__device__ __forceinline__ bool mask_boundary(const uint8_t* m, int pitch, int x, int row, int H, int W) {
    if (m[x] != 255) return false;
    if (x == 0 || x == W - 1 || row == 0 || row == H - 1) return true;
    return m[x - 1] != 255 || m[x + 1] != 255 || m[x - pitch] != 255 || m[x + pitch] != 255;
}

__global__ __launch_bounds__(64) void k_mask_rowcount(const uint8_t* masks, const uint8_t* const* ptrs, int H, int W, int pitch, int* rowcnt) {
    const int row = blockIdx.x, o = blockIdx.y, lane = threadIdx.x;
    const uint8_t* m = mask_row(masks, ptrs, o, H, pitch, row);
    int n = 0;
    for (int x = lane; x < W; x += 64) n += mask_boundary(m, pitch, x, row, H, W);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) n += __shfl_xor(n, d, 64);
    if (lane == 0) rowcnt[o * H + row] = n;
}

// one wavefront per object: rowcnt[o*H + row] becomes the number of mask pixels in the rows before `row`; objcnt[o] = the object's total
__global__ __launch_bounds__(64) void k_mask_rowscan(int H, int* rowcnt, int* objcnt) {
    const int o = blockIdx.x, lane = threadIdx.x;
    int* rc = rowcnt + (long long)o * H;
    int carry = 0;
    for (int r0 = 0; r0 < H; r0 += 64) {
        const int r = r0 + lane;
        const int v = r < H ? rc[r] : 0;
        int incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
        if (r < H) rc[r] = carry + incl - v;
        carry += __shfl(incl, 63, 64);
    }
    if (lane == 0) objcnt[o] = carry;
}

// one wavefront: area_start[o] = pixels of the objects before o, area_start[nObj] = total
__global__ __launch_bounds__(64) void k_mask_objscan(int nObj, const int* objcnt, int* area_start) {
    const int lane = threadIdx.x;
    int carry = 0;
    for (int o0 = 0; o0 < nObj; o0 += 64) {
        const int o = o0 + lane;
        const int v = o < nObj ? objcnt[o] : 0;
        int incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
        if (o < nObj) area_start[o] = carry + incl - v;
        carry += __shfl(incl, 63, 64);
    }
    if (lane == 0) area_start[nObj] = carry;
}

__global__ __launch_bounds__(64) void k_mask_fill(const uint8_t* masks, const uint8_t* const* ptrs, int H, int W, int pitch, const int* rowstart,
                                                  const int* area_start, short2* area) {
    const int row = blockIdx.x, o = blockIdx.y, lane = threadIdx.x;
    const uint8_t* m = mask_row(masks, ptrs, o, H, pitch, row);
    int base = area_start[o] + rowstart[o * H + row];
    for (int x0 = 0; x0 < W; x0 += 64) {
        const int x = x0 + lane;
        const bool f = x < W && mask_boundary(m, pitch, x, row, H, W);
        const unsigned long long bal = __ballot(f);
        if (f) area[base + __popcll(bal & ((1ull << lane) - 1ull))] = make_short2((short)x, (short)row);
        base += __popcll(bal);
    }
}

// Frame::BuildObject2DsRGBD keypoint test (reference src/Frame.cc:262-272): every mask pixel (int)(kp.pt.y + row), (int)(kp.pt.x + col), row / col in
// [-10, 10), equals 255.  One wavefront per (keypoint, mask): 400 byte reads, one ballot.  A pixel outside the image fails the test (the reference
// reads out of bounds there).  out[b][k] gets bit o.
__global__ __launch_bounds__(256) void k_object_kp_test(const oslam_keypoint_t* keysUn, int kp_stride, const int* n_kps, const uint8_t* const* mask_ptrs,
                                                        const int* mask0, const int* n_masks, int H, int W, int pitch, uint8_t* out) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= n_kps[b]) return;
    const oslam_keypoint_t kp = keysUn[(long long)b * kp_stride + k];
    uint8_t bits = 0;
    for (int o = 0; o < n_masks[b]; o++) {
        const uint8_t* m = mask_ptrs[mask0[b] + o];
        bool ok = true;
        for (int i = lane; i < 400; i += 64) {
            const int row = i / 20 - 10, col = i % 20 - 10;
            const int y = (int)(kp.y + (float)row), x = (int)(kp.x + (float)col);
            ok = ok && y >= 0 && y < H && x >= 0 && x < W && m[(long long)y * pitch + x] == 255;
        }
        if (__ballot(!ok) == 0ull) bits |= (uint8_t)(1u << o);
    }
    if (lane == 0) out[(long long)b * kp_stride + k] = bits;
}

// ---- one-bit-per-pixel form of the instance masks -------------------------------------------------------------------------------------------
// bits[m][row][w] (uint64, WB = ceil(W / 64) words per row): bit i = pixel 64 w + i of mask m equals 255.  The keypoint test and the boundary lists read
// this 1/8-size image instead of the masks: ONE pass over the mask bytes per step (k_mask_bits), everything after it works on words.
__device__ __forceinline__ uint32_t ff_nibble(uint32_t x) {   // bit k = byte k of x == 0xFF
    const uint32_t t = ((x & 0x7f7f7f7fu) + 0x01010101u) & x & 0x80808080u;   // bit 7 of every byte that is 0xFF (no carry leaves a byte)
    return (((t >> 7) * 0x00204081u) >> 21) & 0xFu;
}

__global__ __launch_bounds__(256) void k_mask_bits(const uint8_t* const* ptrs, int H, int W, int pitch, int WB, unsigned long long* bits) {
    const int m = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;   // (row, word)
    if (i >= H * WB) return;
    const int row = i / WB, w = i - row * WB;
    const uint8_t* src = ptrs[m] + (long long)row * pitch + 64 * w;
    const int n = min(64, W - 64 * w);
    unsigned long long v = 0;
    if (n == 64 && (((uintptr_t)src) & 15) == 0) {
        const uint4* s4 = (const uint4*)src;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint4 x = s4[q];
            const unsigned long long nib = (unsigned long long)(ff_nibble(x.x) | (ff_nibble(x.y) << 4) | (ff_nibble(x.z) << 8) | (ff_nibble(x.w) << 12));
            v |= nib << (16 * q);
        }
    } else {
        for (int k = 0; k < n; k++) v |= (unsigned long long)(src[k] == 255) << k;
    }
    bits[((long long)m * H + row) * WB + w] = v;
}

// 20 consecutive bits of a row starting at column x (0 <= x, x + 20 <= W)
__device__ __forceinline__ bool row_ones20(const unsigned long long* r, int WB, int x) {
    const int w = x >> 6, sh = x & 63;
    unsigned long long v = r[w] >> sh;
    if (sh > 44 && w + 1 < WB) v |= r[w + 1] << (64 - sh);
    return (v & 0xFFFFFull) == 0xFFFFFull;
}

// Frame::BuildObject2DsRGBD keypoint test from the bitmaps: one thread per keypoint.  The reference evaluates (int)(kp.pt.y + row), (int)(kp.pt.x + col)
// for every (row, col) in [-10, 10)^2; the float sums are monotone in row / col, so when the first and the last of the 20 differ by 19 the coordinates
// are consecutive and a row of the window is 20 adjacent bits.  Anything else (a window that leaves the image, float sums that repeat an integer)
// goes through the literal per-pixel form.
__global__ __launch_bounds__(256) void k_object_kp_test_bits(const oslam_keypoint_t* keysUn, int kp_stride, const int* n_kps, const unsigned long long* bits,
                                                             const int* mask0, const int* n_masks, int H, int W, int WB, uint8_t* out) {
    const int b = blockIdx.y;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_kps[b]) return;
    const oslam_keypoint_t kp = keysUn[(long long)b * kp_stride + k];
    const int x_lo = (int)(kp.x + -10.0f), x_hi = (int)(kp.x + 9.0f), y_lo = (int)(kp.y + -10.0f), y_hi = (int)(kp.y + 9.0f);
    const bool fast = x_hi - x_lo == 19 && y_hi - y_lo == 19 && x_lo >= 0 && x_hi < W && y_lo >= 0 && y_hi < H && kp.x + -10.0f >= 0.0f && kp.y + -10.0f >= 0.0f;
    uint8_t res = 0;
    for (int o = 0; o < n_masks[b]; o++) {
        const unsigned long long* B = bits + (long long)(mask0[b] + o) * H * WB;
        bool ok = true;
        if (fast) {
            for (int r = 0; r < 20 && ok; r++) ok = row_ones20(B + (long long)(y_lo + r) * WB, WB, x_lo);
        } else {
            for (int i = 0; i < 400 && ok; i++) {
                const int row = i / 20 - 10, col = i % 20 - 10;
                const int y = (int)(kp.y + (float)row), x = (int)(kp.x + (float)col);
                ok = y >= 0 && y < H && x >= 0 && x < W && ((B[(long long)y * WB + (x >> 6)] >> (x & 63)) & 1ull);
            }
        }
        if (ok) res |= (uint8_t)(1u << o);
    }
    out[(long long)b * kp_stride + k] = res;
}

// boundary word of (mask, row, w): inside and with a 4-neighbour that is not inside (pixels outside the image count as not inside, which makes the
// image border a boundary exactly like mask_boundary)
__device__ __forceinline__ unsigned long long boundary_word(const unsigned long long* B, int H, int WB, int row, int w) {
    const unsigned long long* r = B + (long long)row * WB;
    const unsigned long long I = r[w];
    if (!I) return 0;
    const unsigned long long L = (I << 1) | (w > 0 ? r[w - 1] >> 63 : 0ull), R = (I >> 1) | (w + 1 < WB ? r[w + 1] << 63 : 0ull);
    const unsigned long long U = row > 0 ? r[w - WB] : 0ull, D = row + 1 < H ? r[w + WB] : 0ull;
    return I & ~(L & R & U & D);
}

__global__ __launch_bounds__(256) void k_mask_rowcount_bits(const unsigned long long* bits, const int* bits_index, int H, int WB, int* rowcnt) {
    const int o = blockIdx.y, row = blockIdx.x * 256 + threadIdx.x;
    if (row >= H) return;
    const unsigned long long* B = bits + (long long)bits_index[o] * H * WB;
    int n = 0;
    for (int w = 0; w < WB; w++) n += __popcll(boundary_word(B, H, WB, row, w));
    rowcnt[o * H + row] = n;
}

__global__ __launch_bounds__(256) void k_mask_fill_bits(const unsigned long long* bits, const int* bits_index, int H, int WB, const int* rowstart, const int* area_start,
                                                        short2* area) {
    const int o = blockIdx.y, row = blockIdx.x * 256 + threadIdx.x;
    if (row >= H) return;
    const unsigned long long* B = bits + (long long)bits_index[o] * H * WB;
    int at = area_start[o] + rowstart[o * H + row];
    for (int w = 0; w < WB; w++) {
        unsigned long long v = boundary_word(B, H, WB, row, w);
        while (v) {
            const int t = __ffsll((long long)v) - 1;
            area[at++] = make_short2((short)(64 * w + t), (short)row);
            v &= v - 1;
        }
    }
}

}  // namespace oslam

using namespace oslam;

// dynamic LDS of k_pose_optimize: chi2 (8 B) + level (1 B) + active-list entry (2 B) per edge slot, + the staged edge data (28 B) when STAGE
static size_t pose_lds_bytes(size_t stride, bool stage) { return stride * (stage ? 39 : 11) + 64; }
#ifndef OSLAM_POSE_LDS_BUDGET
#define OSLAM_POSE_LDS_BUDGET (22 * 1024)
#endif
constexpr size_t kPoseLdsBudget = OSLAM_POSE_LDS_BUDGET;   // FOUR workgroups of 128 threads per CU (what the kernel's ~230 registers allow): 40 KB each minus the kernel's static arrays (18 KB, 16 of them block_sum_wide's).
// Staging therefore only happens for frames of <= 570 edge slots; larger frames read their edges through the L2 in every pass, which measured faster than two staged frames per CU (0.85 against 1.37 ms per 1024 frames).

struct oslam_poseopt {
    int device = 0, max_batch = 0, max_points = 0;
    bool stage = false;   // edge data staged in LDS (fits for max_points <= 2700)
    float* d_Tout = nullptr; uint8_t* d_outlier = nullptr; int* d_ninl = nullptr; int* d_stats = nullptr;
    // staging for the host API
    float* d_T = nullptr; float* d_Xw = nullptr; float* d_obs = nullptr; float* d_inv = nullptr; uint8_t* d_has = nullptr;
    uint8_t* h_pin = nullptr; size_t pin_cap = 0;   // pinned staging of the single-frame host API (inputs, then results)
    double* d_trace = nullptr; int trace_cap = 0;   // LM trace of frame 0 (oslam_poseopt_trace), off by default
    const unsigned long long* bits = nullptr; const int* bits_index = nullptr;   // one-bit-per-pixel masks for the NEXT batch call (oslam_poseopt_use_mask_bits)
    // semantic variant: grow-only device buffers
    struct Buf { void* p = nullptr; size_t cap = 0; };
    Buf masks, rowcnt, objcnt, area, area_start, objmp_Xw, objmp_obj, joint_kp, joint_obj, kp_uv, eXw, eobs, elevel, echi2, eobj, eout, etmp, nsem;
};

static int ensure(oslam_poseopt::Buf& b, size_t bytes) {
    if (b.p && bytes <= b.cap) return OSLAM_OK;
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = bytes + bytes / 2 + 256;
    OSLAM_HIP_CHECK(hipMalloc(&b.p, b.cap));
    return OSLAM_OK;
}

extern "C" {

void oslam_poseopt_destroy(oslam_poseopt_t* h) {
    if (!h) return;
    void* ptrs[] = {h->d_Tout, h->d_outlier, h->d_ninl, h->d_stats, h->d_T, h->d_Xw, h->d_obs, h->d_inv, h->d_has,
                    h->masks.p, h->rowcnt.p, h->objcnt.p, h->area.p, h->area_start.p, h->objmp_Xw.p, h->objmp_obj.p, h->joint_kp.p, h->joint_obj.p, h->kp_uv.p,
                    h->eXw.p, h->eobs.p, h->elevel.p, h->echi2.p, h->eobj.p, h->eout.p, h->etmp.p, h->nsem.p};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (h->h_pin) (void)hipHostFree(h->h_pin);
    if (h->d_trace) (void)hipFree(h->d_trace);
    delete h;
}

int oslam_poseopt_create(oslam_poseopt_t** out, int max_batch, int max_points, int device) {
    if (!out) { set_error("out is NULL"); return OSLAM_E_INVALID; }
    *out = nullptr;
    if (max_batch < 1 || max_points < 1 || max_points > 11000) { set_error("oslam_poseopt_create: invalid argument (max_points <= 11000: chi2 + level per edge slot live in LDS)"); return OSLAM_E_INVALID; }
    int ndev = oslam_device_count();
    if (ndev <= 0) { set_error("no HIP device visible: the gfx950 pose optimiser has no CPU fallback"); return OSLAM_E_HIP; }
    if (device < 0 || device >= ndev) { set_error("device out of range"); return OSLAM_E_INVALID; }
    OSLAM_HIP_CHECK(hipSetDevice(device));
    oslam_poseopt* h = new oslam_poseopt();
    h->device = device; h->max_batch = max_batch; h->max_points = max_points;
    const size_t B = max_batch, NP = max_points;
#define ALLOC(ptr, bytes)                                                         \
    do {                                                                          \
        hipError_t e_ = hipMalloc((void**)&(ptr), (bytes));                       \
        if (e_ != hipSuccess) {                                                   \
            set_error("hipMalloc(%zu) failed: %s", (size_t)(bytes), hipGetErrorString(e_)); \
            oslam_poseopt_destroy(h);                                             \
            return OSLAM_E_HIP;                                                   \
        }                                                                         \
    } while (0)
    ALLOC(h->d_Tout, B * 64); ALLOC(h->d_outlier, B * NP); ALLOC(h->d_ninl, B * 4); ALLOC(h->d_stats, B * 8);
    ALLOC(h->d_T, 64); ALLOC(h->d_Xw, NP * 12); ALLOC(h->d_obs, NP * 12); ALLOC(h->d_inv, NP * 4); ALLOC(h->d_has, NP);
#undef ALLOC
    h->stage = pose_lds_bytes(NP, true) <= kPoseLdsBudget;
    OSLAM_HIP_CHECK(hipFuncSetAttribute((const void*)k_pose_optimize<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pose_lds_bytes(NP, false)));
    OSLAM_HIP_CHECK(hipFuncSetAttribute((const void*)k_pose_optimize<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pose_lds_bytes(NP, false)));
    if (h->stage) {
        OSLAM_HIP_CHECK(hipFuncSetAttribute((const void*)k_pose_optimize<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pose_lds_bytes(NP, true)));
        OSLAM_HIP_CHECK(hipFuncSetAttribute((const void*)k_pose_optimize<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pose_lds_bytes(NP, true)));
    }
    *out = h;
    return OSLAM_OK;
}

#ifdef OSLAM_POSE_PROFILE
int oslam_pose_debug_profile(unsigned long long out[8]) {
    OSLAM_HIP_CHECK(hipDeviceSynchronize());
    OSLAM_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pose_prof), 64));
    return OSLAM_OK;
}
#endif

// Test hook: record the Levenberg-Marquardt trials of frame 0 of every following call (cap > 0), or stop recording (cap == 0).
int oslam_poseopt_trace(oslam_poseopt_t* h, int cap) {
    if (!h || cap < 0) { set_error("oslam_poseopt_trace: bad argument"); return OSLAM_E_INVALID; }
    OSLAM_HIP_CHECK(hipSetDevice(h->device));
    OSLAM_HIP_CHECK(hipDeviceSynchronize());
    if (h->d_trace) (void)hipFree(h->d_trace);
    h->d_trace = nullptr; h->trace_cap = 0;
    if (cap == 0) return OSLAM_OK;
    OSLAM_HIP_CHECK(hipMalloc((void**)&h->d_trace, sizeof(double) * (1 + 6 * (size_t)cap)));
    OSLAM_HIP_CHECK(hipMemset(h->d_trace, 0, sizeof(double) * (1 + 6 * (size_t)cap)));
    h->trace_cap = cap;
    return OSLAM_OK;
}

// Copies the trace out ([cap][6] doubles: F before the trial, F of the trial, rho, lambda of the trial, accepted, first trial of a round), returns the number of trials seen
// since the last read in *n (it can exceed cap: only the first cap are kept) and clears the trace.
int oslam_poseopt_trace_read(oslam_poseopt_t* h, double* out, int32_t* n) {
    if (!h || !out || !n || !h->d_trace) { set_error("oslam_poseopt_trace_read: no trace"); return OSLAM_E_INVALID; }
    OSLAM_HIP_CHECK(hipSetDevice(h->device));
    OSLAM_HIP_CHECK(hipDeviceSynchronize());
    double cnt = 0;
    OSLAM_HIP_CHECK(hipMemcpy(&cnt, h->d_trace, sizeof(double), hipMemcpyDeviceToHost));
    OSLAM_HIP_CHECK(hipMemcpy(out, h->d_trace + 1, sizeof(double) * 6 * (size_t)h->trace_cap, hipMemcpyDeviceToHost));
    OSLAM_HIP_CHECK(hipMemset(h->d_trace, 0, sizeof(double)));
    *n = (int32_t)cnt;
    return OSLAM_OK;
}

int oslam_pose_optimize_batch_device(oslam_poseopt_t* h, int batch, int stride, const int32_t* d_n, int n_const, const float* d_Tcw,
                                     const float* d_Xw, const float* d_obs, const float* d_invSigma2, const uint8_t* d_has_mp,
                                     const float K5[5], void* stream) {
    if (!h || !d_Tcw || !d_Xw || !d_obs || !d_invSigma2 || !d_has_mp || !K5) { set_error("NULL argument"); return OSLAM_E_INVALID; }
    if (batch < 1 || batch > h->max_batch) { set_error("batch %d outside [1,%d]", batch, h->max_batch); return OSLAM_E_INVALID; }
    if (stride < 1 || stride > h->max_points || (!d_n && (n_const < 0 || n_const > stride))) { set_error("stride/n exceed max_points %d", h->max_points); return OSLAM_E_CAPACITY; }
    OSLAM_HIP_CHECK(hipSetDevice(h->device));
    PoseCtx c;
    c.Tcw = d_Tcw; c.Xw = d_Xw; c.obs = d_obs; c.invSigma2 = d_invSigma2; c.has_mp = d_has_mp;
    c.n = d_n; c.n_const = n_const; c.stride = stride;
    c.fx = K5[0]; c.fy = K5[1]; c.cx = K5[2]; c.cy = K5[3]; c.bf = K5[4];
    c.Tcw_out = h->d_Tout; c.outlier = h->d_outlier; c.n_inliers = h->d_ninl; c.stats = h->d_stats; c.trace = h->d_trace; c.trace_cap = h->trace_cap;
    memset(&c.sem, 0, sizeof(c.sem));
    if (h->stage) hipLaunchKernelGGL((k_pose_optimize<false, true>), dim3(batch), dim3(kPoseThreads), pose_lds_bytes(stride, true), (hipStream_t)stream, c);
    else hipLaunchKernelGGL((k_pose_optimize<false, false>), dim3(batch), dim3(kPoseThreads), pose_lds_bytes(stride, false), (hipStream_t)stream, c);
    OSLAM_HIP_CHECK(hipGetLastError());
    return OSLAM_OK;
}

int oslam_poseopt_results_device(const oslam_poseopt_t* h, const float** d_Tcw_out, const uint8_t** d_outlier, const int32_t** d_n_inliers,
                                 const int32_t** d_stats) {
    if (!h) { set_error("NULL handle"); return OSLAM_E_INVALID; }
    if (d_Tcw_out) *d_Tcw_out = h->d_Tout;
    if (d_outlier) *d_outlier = h->d_outlier;
    if (d_n_inliers) *d_n_inliers = h->d_ninl;
    if (d_stats) *d_stats = h->d_stats;
    return OSLAM_OK;
}

int oslam_pose_optimize(oslam_poseopt_t* h, int N, const float Tcw_in[16], const float* Xw, const float* obs, const float* invSigma2,
                        const uint8_t* has_mp, const float K5[5], float Tcw_out[16], uint8_t* outlier, int32_t* n_inliers,
                        int32_t stats[2]) {
    if (!h || !Tcw_in || !K5 || !Tcw_out || !n_inliers || (N > 0 && (!Xw || !obs || !invSigma2 || !has_mp || !outlier))) { set_error("NULL argument"); return OSLAM_E_INVALID; }
    if (N < 0 || N > h->max_points) { set_error("%d points > capacity %d", N, h->max_points); return OSLAM_E_CAPACITY; }
    OSLAM_HIP_CHECK(hipSetDevice(h->device));
    // inputs and results travel through one pinned buffer as async copies with a single synchronisation
    // (ten pageable hipMemcpy calls cost ~0.2 ms, a third of the kernel itself)
    const size_t a256 = 255;
    const size_t o_T = 0, o_X = 256, o_o = o_X + (((size_t)N * 12 + a256) & ~a256), o_i = o_o + (((size_t)N * 12 + a256) & ~a256),
                 o_h = o_i + (((size_t)N * 4 + a256) & ~a256), o_out = o_h + (((size_t)N + a256) & ~a256), total = o_out + 512 + (((size_t)N + a256) & ~a256);
    if (total > h->pin_cap) {
        if (h->h_pin) (void)hipHostFree(h->h_pin);
        h->h_pin = nullptr; h->pin_cap = 0;
        OSLAM_HIP_CHECK(hipHostMalloc((void**)&h->h_pin, total + total / 2, 0));
        h->pin_cap = total + total / 2;
    }
    uint8_t* pin = h->h_pin;
    memcpy(pin + o_T, Tcw_in, 64);
    OSLAM_HIP_CHECK(hipMemcpyAsync(h->d_T, pin + o_T, 64, hipMemcpyHostToDevice, nullptr));
    if (N > 0) {
        memcpy(pin + o_X, Xw, (size_t)N * 12); memcpy(pin + o_o, obs, (size_t)N * 12); memcpy(pin + o_i, invSigma2, (size_t)N * 4); memcpy(pin + o_h, has_mp, (size_t)N);
        OSLAM_HIP_CHECK(hipMemcpyAsync(h->d_Xw, pin + o_X, (size_t)N * 12, hipMemcpyHostToDevice, nullptr));
        OSLAM_HIP_CHECK(hipMemcpyAsync(h->d_obs, pin + o_o, (size_t)N * 12, hipMemcpyHostToDevice, nullptr));
        OSLAM_HIP_CHECK(hipMemcpyAsync(h->d_inv, pin + o_i, (size_t)N * 4, hipMemcpyHostToDevice, nullptr));
        OSLAM_HIP_CHECK(hipMemcpyAsync(h->d_has, pin + o_h, (size_t)N, hipMemcpyHostToDevice, nullptr));
    }
    int rc = oslam_pose_optimize_batch_device(h, 1, h->max_points, nullptr, N, h->d_T, h->d_Xw, h->d_obs, h->d_inv, h->d_has, K5, nullptr);
    if (rc) return rc;
    uint8_t* po = pin + o_out;   // [0,64) pose, [64,68) inliers, [128,136) stats, [512, 512+N) outlier flags
    OSLAM_HIP_CHECK(hipMemcpyAsync(po, h->d_Tout, 64, hipMemcpyDeviceToHost, nullptr));
    OSLAM_HIP_CHECK(hipMemcpyAsync(po + 64, h->d_ninl, 4, hipMemcpyDeviceToHost, nullptr));
    OSLAM_HIP_CHECK(hipMemcpyAsync(po + 128, h->d_stats, 8, hipMemcpyDeviceToHost, nullptr));
    if (N > 0) OSLAM_HIP_CHECK(hipMemcpyAsync(po + 512, h->d_outlier, (size_t)N, hipMemcpyDeviceToHost, nullptr));
    OSLAM_HIP_CHECK(hipStreamSynchronize(nullptr));
    memcpy(Tcw_out, po, 64);
    memcpy(n_inliers, po + 64, 4);
    if (N > 0) memcpy(outlier, po + 512, (size_t)N);
    if (stats) memcpy(stats, po + 128, 8);
    return OSLAM_OK;
}


int oslam_pose_optimize2(oslam_poseopt_t* h, int N, const float Tcw_in[16], const float* Xw, const float* obs, const float* invSigma2,
                         const uint8_t* has_mp, const float K5[5], const oslam_semantic_t* sem, float Tcw_out[16], uint8_t* outlier,
                         int32_t* n_inliers, int32_t* n_semantic) {
    if (!h || !Tcw_in || !K5 || !Tcw_out || !n_inliers || !sem || !n_semantic || (N > 0 && (!Xw || !obs || !invSigma2 || !has_mp || !outlier))) { set_error("NULL argument"); return OSLAM_E_INVALID; }
    if (N < 0 || N > h->max_points) { set_error("%d points > capacity %d", N, h->max_points); return OSLAM_E_CAPACITY; }
    if (sem->nObj < 0 || sem->H < 0 || sem->W < 0 || sem->W >= 32768 || sem->H >= 32768 || sem->nObjMp < 0 || sem->nJoint < 0) { set_error("bad semantic sizes"); return OSLAM_E_INVALID; }
    if (sem->nObj > 0 && (!sem->masks || (sem->nObjMp > 0 && (!sem->objmp_Xw || !sem->objmp_obj)) || (sem->nJoint > 0 && (!sem->joint_kp || !sem->joint_obj || !sem->kp_uv)))) { set_error("NULL semantic array"); return OSLAM_E_INVALID; }
    for (int i = 0; i < sem->nObjMp; i++) if (sem->objmp_obj[i] < 0 || sem->objmp_obj[i] >= sem->nObj) { set_error("objmp_obj out of range"); return OSLAM_E_INVALID; }
    for (int i = 0; i < sem->nJoint; i++) if (sem->joint_obj[i] < 0 || sem->joint_obj[i] >= sem->nObj || sem->joint_kp[i] < 0 || sem->joint_kp[i] >= N) { set_error("joint index out of range"); return OSLAM_E_INVALID; }
    OSLAM_HIP_CHECK(hipSetDevice(h->device));
    const size_t npx = (size_t)sem->nObj * sem->H * sem->W;
    const size_t nrows = (size_t)sem->nObj * sem->H;
    const size_t nsemcap = (size_t)sem->nJoint + sem->nObjMp;
    int rc;
    if ((rc = ensure(h->masks, npx + 1)) || (rc = ensure(h->area, (npx + 1) * sizeof(short2))) || (rc = ensure(h->rowcnt, (nrows + 1) * 4)) ||
        (rc = ensure(h->objcnt, ((size_t)sem->nObj + 1) * 4)) ||
        (rc = ensure(h->area_start, ((size_t)sem->nObj + 2) * 4)) || (rc = ensure(h->objmp_Xw, (size_t)sem->nObjMp * 12 + 12)) ||
        (rc = ensure(h->objmp_obj, (size_t)sem->nObjMp * 4 + 4)) || (rc = ensure(h->joint_kp, (size_t)sem->nJoint * 4 + 4)) ||
        (rc = ensure(h->joint_obj, (size_t)sem->nJoint * 4 + 4)) || (rc = ensure(h->kp_uv, (size_t)N * 8 + 8)) ||
        (rc = ensure(h->eXw, nsemcap * 12 + 12)) || (rc = ensure(h->eobs, nsemcap * 8 + 8)) || (rc = ensure(h->elevel, nsemcap + 1)) ||
        (rc = ensure(h->echi2, nsemcap * 8 + 8)) || (rc = ensure(h->eobj, nsemcap * 4 + 4)) || (rc = ensure(h->eout, nsemcap + 1)) ||
        (rc = ensure(h->etmp, nsemcap * 4 + 4)) || (rc = ensure(h->nsem, 4)))
        return rc;

    OSLAM_HIP_CHECK(hipMemcpy(h->d_T, Tcw_in, 64, hipMemcpyHostToDevice));
    if (N > 0) {
        OSLAM_HIP_CHECK(hipMemcpy(h->d_Xw, Xw, (size_t)N * 12, hipMemcpyHostToDevice));
        OSLAM_HIP_CHECK(hipMemcpy(h->d_obs, obs, (size_t)N * 12, hipMemcpyHostToDevice));
        OSLAM_HIP_CHECK(hipMemcpy(h->d_inv, invSigma2, (size_t)N * 4, hipMemcpyHostToDevice));
        OSLAM_HIP_CHECK(hipMemcpy(h->d_has, has_mp, (size_t)N, hipMemcpyHostToDevice));
        if (sem->kp_uv) OSLAM_HIP_CHECK(hipMemcpy((float*)h->kp_uv.p, sem->kp_uv, (size_t)N * 8, hipMemcpyHostToDevice));
    }
    if (npx) OSLAM_HIP_CHECK(hipMemcpy((uint8_t*)h->masks.p, sem->masks, npx, hipMemcpyHostToDevice));
    if (sem->nObjMp) {
        OSLAM_HIP_CHECK(hipMemcpy((float*)h->objmp_Xw.p, sem->objmp_Xw, (size_t)sem->nObjMp * 12, hipMemcpyHostToDevice));
        OSLAM_HIP_CHECK(hipMemcpy((int*)h->objmp_obj.p, sem->objmp_obj, (size_t)sem->nObjMp * 4, hipMemcpyHostToDevice));
    }
    if (sem->nJoint) {
        OSLAM_HIP_CHECK(hipMemcpy((int*)h->joint_kp.p, sem->joint_kp, (size_t)sem->nJoint * 4, hipMemcpyHostToDevice));
        OSLAM_HIP_CHECK(hipMemcpy((int*)h->joint_obj.p, sem->joint_obj, (size_t)sem->nJoint * 4, hipMemcpyHostToDevice));
    }
    if (sem->nObj > 0 && sem->H > 0 && sem->W > 0) {
        hipLaunchKernelGGL(k_mask_rowcount, dim3(sem->H, sem->nObj), dim3(64), 0, nullptr, (uint8_t*)h->masks.p, nullptr, sem->H, sem->W, sem->W, (int*)h->rowcnt.p);
        hipLaunchKernelGGL(k_mask_rowscan, dim3(sem->nObj), dim3(64), 0, nullptr, sem->H, (int*)h->rowcnt.p, (int*)h->objcnt.p);
        hipLaunchKernelGGL(k_mask_objscan, dim3(1), dim3(64), 0, nullptr, sem->nObj, (int*)h->objcnt.p, (int*)h->area_start.p);
        hipLaunchKernelGGL(k_mask_fill, dim3(sem->H, sem->nObj), dim3(64), 0, nullptr, (uint8_t*)h->masks.p, nullptr, sem->H, sem->W, sem->W, (int*)h->rowcnt.p, (int*)h->area_start.p, (short2*)h->area.p);
    } else {
        OSLAM_HIP_CHECK(hipMemset((int*)h->area_start.p, 0, 2 * sizeof(int)));
    }
    PoseCtx c;
    c.Tcw = h->d_T; c.Xw = h->d_Xw; c.obs = h->d_obs; c.invSigma2 = h->d_inv; c.has_mp = h->d_has;
    c.n = nullptr; c.n_const = N; c.stride = h->max_points;
    c.fx = K5[0]; c.fy = K5[1]; c.cx = K5[2]; c.cy = K5[3]; c.bf = K5[4];
    c.Tcw_out = h->d_Tout; c.outlier = h->d_outlier; c.n_inliers = h->d_ninl; c.stats = h->d_stats; c.trace = h->d_trace; c.trace_cap = h->trace_cap;
    SemCtx& sm = c.sem;
    memset(&sm, 0, sizeof(sm));   // every optional pointer (bitmaps, frame table, pointer table) off unless set below
    sm.nObj = sem->nObj; sm.area = (short2*)h->area.p; sm.area_start = (int*)h->area_start.p; sm.row_start = (int*)h->rowcnt.p;
    sm.masks = (const uint8_t*)h->masks.p; sm.mask_ptrs = nullptr; sm.H = sem->H; sm.W = sem->W; sm.pitch = sem->W;
    sm.nObjMp = sem->nObjMp; sm.objmp_Xw = (float*)h->objmp_Xw.p; sm.objmp_obj = (int*)h->objmp_obj.p;
    sm.nJoint = sem->nJoint; sm.joint_kp = (int*)h->joint_kp.p; sm.joint_obj = (int*)h->joint_obj.p; sm.kp_uv = (float*)h->kp_uv.p;
    sm.minX = sem->bounds[0]; sm.minY = sem->bounds[1]; sm.maxX = sem->bounds[2]; sm.maxY = sem->bounds[3]; sm.invSigma2_0 = sem->invSigma2_0;
    sm.e_Xw = (float*)h->eXw.p; sm.e_obs = (float*)h->eobs.p; sm.e_level = (uint8_t*)h->elevel.p; sm.e_chi2 = (double*)h->echi2.p; sm.e_obj = (int*)h->eobj.p; sm.e_out = (uint8_t*)h->eout.p; sm.e_tmp = (int*)h->etmp.p;
    sm.nSem = (int*)h->nsem.p;
    sm.frames = nullptr;
    if (h->stage) hipLaunchKernelGGL((k_pose_optimize<true, true>), dim3(1), dim3(kPoseThreads), pose_lds_bytes(h->max_points, true), nullptr, c);
    else hipLaunchKernelGGL((k_pose_optimize<true, false>), dim3(1), dim3(kPoseThreads), pose_lds_bytes(h->max_points, false), nullptr, c);
    OSLAM_HIP_CHECK(hipGetLastError());
    OSLAM_HIP_CHECK(hipDeviceSynchronize());
    OSLAM_HIP_CHECK(hipMemcpy(Tcw_out, h->d_Tout, 64, hipMemcpyDeviceToHost));
    OSLAM_HIP_CHECK(hipMemcpy(n_inliers, h->d_ninl, 4, hipMemcpyDeviceToHost));
    OSLAM_HIP_CHECK(hipMemcpy(n_semantic, (int*)h->nsem.p, 4, hipMemcpyDeviceToHost));
    if (N > 0) OSLAM_HIP_CHECK(hipMemcpy(outlier, h->d_outlier, (size_t)N, hipMemcpyDeviceToHost));
    return OSLAM_OK;
}


// Batch of frames with semantic constraints (Tracking::TrackLocalMap of several sequences): the base arrays as in
// oslam_pose_optimize_batch_device; the Object2D masks of all frames through a table of device pointers (no copy), the object map points and
// the M_joint sets as pools with one oslam_sem_frame_t per frame.  Asynchronous on `stream`; results through oslam_poseopt_results_device +
// oslam_poseopt_semantic_results_device.
int oslam_pose_optimize2_batch_device(oslam_poseopt_t* h, int batch, int stride, const int32_t* d_n, const float* d_Tcw, const float* d_Xw, const float* d_obs,
                                      const float* d_invSigma2, const uint8_t* d_has_mp, const float K5[5], const oslam_sem_frame_t* d_frames, int total_obj,
                                      const uint8_t* const* d_mask_ptrs, int H, int W, int mask_pitch, int total_objmp, const float* d_objmp_Xw,
                                      const int32_t* d_objmp_obj, int total_joint, const int32_t* d_joint_kp, const int32_t* d_joint_obj, const float bounds[4],
                                      float invSigma2_0, void* stream) {
    if (!h) { set_error("NULL argument"); return OSLAM_E_INVALID; }
    const unsigned long long* use_bits = h->bits; const int* use_index = h->bits_index;
    h->bits = nullptr; h->bits_index = nullptr;   // the bitmap hint (oslam_poseopt_use_mask_bits) holds for this call only, whatever its outcome
    if (!d_n || !d_Tcw || !d_Xw || !d_obs || !d_invSigma2 || !d_has_mp || !K5 || !d_frames || !bounds) { set_error("NULL argument"); return OSLAM_E_INVALID; }
    if (batch < 1 || batch > h->max_batch) { set_error("batch %d outside [1,%d]", batch, h->max_batch); return OSLAM_E_INVALID; }
    if (stride < 1 || stride > h->max_points) { set_error("stride exceeds max_points %d", h->max_points); return OSLAM_E_CAPACITY; }
    if (total_obj < 0 || total_objmp < 0 || total_joint < 0 || H < 1 || W < 1 || W >= 32768 || H >= 32768 || mask_pitch < W || (total_obj > 0 && !d_mask_ptrs && !(use_bits && use_index)) ||
        (total_objmp > 0 && (!d_objmp_Xw || !d_objmp_obj)) || (total_joint > 0 && (!d_joint_kp || !d_joint_obj))) { set_error("bad semantic sizes"); return OSLAM_E_INVALID; }
    OSLAM_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    const size_t npx = (size_t)total_obj * H * W, nrows = (size_t)total_obj * H, nsemcap = (size_t)total_joint + total_objmp;
    int rc;
    if ((rc = ensure(h->area, (npx + 1) * sizeof(short2))) || (rc = ensure(h->rowcnt, (nrows + 1) * 4)) || (rc = ensure(h->objcnt, ((size_t)total_obj + 1) * 4)) ||
        (rc = ensure(h->area_start, ((size_t)total_obj + 2) * 4)) || (rc = ensure(h->eXw, nsemcap * 12 + 12)) || (rc = ensure(h->eobs, nsemcap * 8 + 8)) ||
        (rc = ensure(h->elevel, nsemcap + 1)) || (rc = ensure(h->echi2, nsemcap * 8 + 8)) || (rc = ensure(h->eobj, nsemcap * 4 + 4)) || (rc = ensure(h->eout, nsemcap + 1)) ||
        (rc = ensure(h->etmp, nsemcap * 4 + 4)) || (rc = ensure(h->nsem, (size_t)h->max_batch * 4)))
        return rc;
    if (total_obj > 0 && use_bits && use_index) {   // boundary lists from the one-bit-per-pixel masks (oslam_poseopt_use_mask_bits)
        const int WB = (W + 63) / 64;
        hipLaunchKernelGGL(k_mask_rowcount_bits, dim3(div_up(H, 256), total_obj), dim3(256), 0, st, use_bits, use_index, H, WB, (int*)h->rowcnt.p);
        hipLaunchKernelGGL(k_mask_rowscan, dim3(total_obj), dim3(64), 0, st, H, (int*)h->rowcnt.p, (int*)h->objcnt.p);
        hipLaunchKernelGGL(k_mask_objscan, dim3(1), dim3(64), 0, st, total_obj, (int*)h->objcnt.p, (int*)h->area_start.p);
        hipLaunchKernelGGL(k_mask_fill_bits, dim3(div_up(H, 256), total_obj), dim3(256), 0, st, use_bits, use_index, H, WB, (int*)h->rowcnt.p, (int*)h->area_start.p,
                           (short2*)h->area.p);
    } else if (total_obj > 0) {
        hipLaunchKernelGGL(k_mask_rowcount, dim3(H, total_obj), dim3(64), 0, st, nullptr, d_mask_ptrs, H, W, mask_pitch, (int*)h->rowcnt.p);
        hipLaunchKernelGGL(k_mask_rowscan, dim3(total_obj), dim3(64), 0, st, H, (int*)h->rowcnt.p, (int*)h->objcnt.p);
        hipLaunchKernelGGL(k_mask_objscan, dim3(1), dim3(64), 0, st, total_obj, (int*)h->objcnt.p, (int*)h->area_start.p);
        hipLaunchKernelGGL(k_mask_fill, dim3(H, total_obj), dim3(64), 0, st, nullptr, d_mask_ptrs, H, W, mask_pitch, (int*)h->rowcnt.p, (int*)h->area_start.p, (short2*)h->area.p);
    } else {
        OSLAM_HIP_CHECK(hipMemsetAsync((int*)h->area_start.p, 0, 2 * sizeof(int), st));
    }
    PoseCtx c;
    c.Tcw = d_Tcw; c.Xw = d_Xw; c.obs = d_obs; c.invSigma2 = d_invSigma2; c.has_mp = d_has_mp;
    c.n = d_n; c.n_const = 0; c.stride = stride;
    c.fx = K5[0]; c.fy = K5[1]; c.cx = K5[2]; c.cy = K5[3]; c.bf = K5[4];
    c.Tcw_out = h->d_Tout; c.outlier = h->d_outlier; c.n_inliers = h->d_ninl; c.stats = h->d_stats; c.trace = h->d_trace; c.trace_cap = h->trace_cap;
    SemCtx& sm = c.sem;
    memset(&sm, 0, sizeof(sm));
    sm.area = (short2*)h->area.p; sm.area_start = (int*)h->area_start.p; sm.row_start = (int*)h->rowcnt.p;
    sm.masks = nullptr; sm.mask_ptrs = d_mask_ptrs; sm.H = H; sm.W = W; sm.pitch = mask_pitch;
    if (total_obj > 0 && use_bits && use_index) { sm.bits = use_bits; sm.bits_index = use_index; sm.WB = (W + 63) / 64; }
    sm.objmp_Xw = d_objmp_Xw; sm.objmp_obj = d_objmp_obj; sm.joint_kp = d_joint_kp; sm.joint_obj = d_joint_obj; sm.kp_uv = nullptr;
    sm.minX = bounds[0]; sm.minY = bounds[1]; sm.maxX = bounds[2]; sm.maxY = bounds[3]; sm.invSigma2_0 = invSigma2_0;
    sm.e_Xw = (float*)h->eXw.p; sm.e_obs = (float*)h->eobs.p; sm.e_level = (uint8_t*)h->elevel.p; sm.e_chi2 = (double*)h->echi2.p; sm.e_obj = (int*)h->eobj.p;
    sm.e_out = (uint8_t*)h->eout.p; sm.e_tmp = (int*)h->etmp.p; sm.nSem = (int*)h->nsem.p; sm.frames = d_frames;
    if (h->stage) hipLaunchKernelGGL((k_pose_optimize<true, true>), dim3(batch), dim3(kPoseThreads), pose_lds_bytes(stride, true), st, c);
    else hipLaunchKernelGGL((k_pose_optimize<true, false>), dim3(batch), dim3(kPoseThreads), pose_lds_bytes(stride, false), st, c);
    OSLAM_HIP_CHECK(hipGetLastError());
    return OSLAM_OK;
}

int oslam_poseopt_semantic_results_device(const oslam_poseopt_t* h, const int32_t** d_n_semantic) {
    if (!h || !d_n_semantic) { set_error("NULL argument"); return OSLAM_E_INVALID; }
    *d_n_semantic = (const int32_t*)h->nsem.p;
    return OSLAM_OK;
}

// One-bit-per-pixel form of n masks (device pointers, rows `pitch` bytes apart): d_bits [n][H][ceil(W / 64)] uint64, bit i of word w = pixel 64 w + i == 255.
int oslam_mask_bits_device(const uint8_t* const* d_mask_ptrs, int n, int H, int W, int pitch, uint64_t* d_bits, void* stream) {
    if (!d_mask_ptrs || !d_bits || n < 1 || H < 1 || W < 1 || pitch < W) { set_error("mask_bits: bad argument"); return OSLAM_E_INVALID; }
    const int WB = (W + 63) / 64;
    hipLaunchKernelGGL(k_mask_bits, dim3(div_up(H * WB, 256), n), dim3(256), 0, (hipStream_t)stream, d_mask_ptrs, H, W, pitch, WB, (unsigned long long*)d_bits);
    OSLAM_HIP_CHECK(hipGetLastError());
    return OSLAM_OK;
}

// The NEXT oslam_pose_optimize2_batch_device call of this handle builds its boundary lists from bitmaps made by oslam_mask_bits_device: object o of the call
// (pooled index) is bitmap d_bits_index[o].  The mask pointers are still needed (unit-cell test of the nearest-pixel search).
int oslam_poseopt_use_mask_bits(oslam_poseopt_t* h, const uint64_t* d_bits, const int32_t* d_bits_index) {
    if (!h) { set_error("NULL handle"); return OSLAM_E_INVALID; }
    h->bits = (const unsigned long long*)d_bits; h->bits_index = d_bits_index;
    return OSLAM_OK;
}

// The keypoint test from bitmaps: bit o of d_out[b][k] = the 20 x 20 window of keypoint k lies inside bitmap d_mask0[b] + o.
int oslam_frame_object_kp_test_bits_batch_device(const oslam_keypoint_t* d_keysUn, int kp_stride, const int32_t* d_n_kps, int batch, const uint64_t* d_bits,
                                                 const int32_t* d_mask0, const int32_t* d_n_masks, int H, int W, uint8_t* d_out, void* stream) {
    if (!d_keysUn || !d_n_kps || !d_bits || !d_mask0 || !d_n_masks || !d_out || batch < 1 || kp_stride < 1 || H < 1 || W < 1) { set_error("object_kp_test_bits: bad argument"); return OSLAM_E_INVALID; }
    hipLaunchKernelGGL(k_object_kp_test_bits, dim3(div_up(kp_stride, 256), batch), dim3(256), 0, (hipStream_t)stream, d_keysUn, kp_stride, d_n_kps,
                       (const unsigned long long*)d_bits, d_mask0, d_n_masks, H, W, (W + 63) / 64, d_out);
    OSLAM_HIP_CHECK(hipGetLastError());
    return OSLAM_OK;
}

// Frame::BuildObject2DsRGBD keypoint test for a batch of frames (see k_object_kp_test).  d_out [batch][kp_stride], bit o = mask mask0[b] + o passes.
int oslam_frame_object_kp_test_batch_device(const oslam_keypoint_t* d_keysUn, int kp_stride, const int32_t* d_n_kps, int batch, const uint8_t* const* d_mask_ptrs,
                                            const int32_t* d_mask0, const int32_t* d_n_masks, int H, int W, int mask_pitch, uint8_t* d_out, void* stream) {
    if (!d_keysUn || !d_n_kps || !d_mask_ptrs || !d_mask0 || !d_n_masks || !d_out || batch < 1 || kp_stride < 1 || H < 1 || W < 1 || mask_pitch < W) { set_error("object_kp_test: bad argument"); return OSLAM_E_INVALID; }
    hipLaunchKernelGGL(k_object_kp_test, dim3(div_up(kp_stride, 4), batch), dim3(256), 0, (hipStream_t)stream, d_keysUn, kp_stride, d_n_kps, d_mask_ptrs, d_mask0, d_n_masks, H, W,
                       mask_pitch, d_out);
    OSLAM_HIP_CHECK(hipGetLastError());
    return OSLAM_OK;
}

}  // extern "C"
