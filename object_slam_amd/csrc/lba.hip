// gfx950 local bundle adjustment: Optimizer::LocalBundleAdjustment (reference
// src/Optimizer.cc:453-778) after the graph has been gathered — g2o Levenberg-Marquardt with
// BlockSolver_6_3's Schur complement, optimize(5) with Huber, chi2 / depth
// gating, optimize(10) without, erase list.  One 1024-thread workgroup per BA problem runs the
// whole schedule in one launch (batch = independent keyframe windows).  fp64 throughout.
//
// Deterministic by construction (no floating-point atomics):
//   * point blocks (Hll, b_l, per-edge Hpl) : one thread per map point over its edge list;
//   * pose blocks (Hpp, b_p)                : one wavefront per keyframe, shuffle reduction;
//   * Schur complement                      : one wavefront per 6x6 block (a<=b) of the reduced system;
//     lanes stride over keyframe a's observations, find the partner observation of keyframe b in
//     the point's edge list, accumulate B_a Dinv B_b^T in registers, shuffle-tree reduction;
//   * reduced camera system                 : blocked (6-wide) Cholesky U^T U with the right-hand
//     side carried as an extra column, back-substitution by one wavefront.
#include <hip/hip_runtime.h>
#include <string.h>
#include <type_traits>
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <array>
#include <vector>

#include "common.h"
#include "se3_math.h"
#include "lane_ops.h"
#include "slam_pool.h"

namespace oslam {

#ifdef OSLAM_LBA_PROFILE
#define LBA_STAMP(i) do { if (tid == 0) { long long t_ = clock64(); prof[i] += t_ - tlast; tlast = t_; } } while (0)
#else
#define LBA_STAMP(i) do { } while (0)
#endif

constexpr int kLbaThreads = 512;
constexpr int kLbaWaves = kLbaThreads / 64;
constexpr int kLbaMaxKF = 128;           // FREE keyframes of a window (the reduced system's order is 6 x that); also every keyframe of the round-1 compact kernel
constexpr int kLbaMaxWindowKF = 1 << 16; // keyframes of a window, fixed cameras included (wide / window layouts keep the poses in memory sized by the window)
constexpr int kRowBufBytes = 8 * 1024;   // LDS scratch: back-substitution vector (n <= 768 doubles)

struct LbaProblem {
    int K, P, E;
    const float* poses;       // [K][16] Tcw
    const uint8_t* fixed;     // [K] 0 free, 1 fixed camera, 2 local keyframe with mnId==0 (fixed, written back)
    const float* points;      // [P][3]
    const int* e_kf;          // [E] point-major edge order
    const int* e_pt;          // [E]
    const float* e_obs;       // [E][3] u, v, uR (uR < 0: monocular)
    const float* e_info;      // [E] invSigma2
    const int* pt_start;      // [P+1]
    const int* pose_start;    // [K+1]
    const int* pose_edges;    // [E] edge ids grouped by keyframe, ascending
    // scratch
    double* Xa; double* Xb;   // [P][3] point state / trial state
    double* chi2;             // [E] _error chi2 as last computed
    uint8_t* level;           // [E]
    double* Hpl;              // [E][18]
    double* Hll; double* Dinv; double* bl; double* xl;   // [P][9],[P][9],[P][3],[P][3]
    double* Hpp; double* bp;  // [K][36],[K][6]
    double* Hs;               // [n][n+1] reduced system + rhs column
    double* xp;               // [n]
    // outputs
    float* poses_out; float* points_out; uint8_t* erase; int* stats;   // stats[4]: it1, trials1, it2, trials2
    const volatile int* stop; // may be NULL
    float K5[5];              // fx, fy, cx, cy, bf (one sensor per window)
    // schedule: LocalBundleAdjustment = {5, 10, 2 stages, robust, Huber sqrt(5.991)/sqrt(7.815)} (src/Optimizer.cc:569-707);
    // BundleAdjustment = {nIterations, -, 1 stage, bRobust, Huber sqrt(5.99)/sqrt(7.815)} (src/Optimizer.cc:85-188)
    int iters0, iters1, nstages, robust0;
    float delta_mono, delta_stereo;
};

struct LbaShared {
    SE3 T[kLbaMaxKF];      // current poses
    SE3 Tn[kLbaMaxKF];     // trial poses
    double R[kLbaMaxKF][9];
    int blk[kLbaMaxKF];    // block index among free poses or -1
    int free_pose[kLbaMaxKF];   // block index -> keyframe
    double red[2][kLbaWaves][4];
    int flag;
    int stopflag;
};

__device__ __forceinline__ double wsum(double v) { return wave_sum_xor(v); }   // (lane_ops.h: the xor butterfly without the LDS crossbar)
__device__ __forceinline__ double wmax(double v) {
    double a = v, b = v;
    swap32_f64(a, b);
    v = fmax(a, b);
    a = v; b = v;
    swap16_f64(a, b);
    v = fmax(a, b);
    v = fmax(v, lane_xor8(v)); v = fmax(v, lane_xor4(v)); v = fmax(v, lane_xor2(v)); v = fmax(v, lane_xor1(v));
    return v;
}
// Sums over the wavefront of N values at once: the first two butterfly steps pair the values, so that after them every row of 16 lanes carries ONE value's
// partial sums (a quarter of the registers); out[q] holds, in every lane of row (b5, b4) of the wavefront (b5 = lane >> 5, b4 = (lane >> 4) & 1), the sum of
// value 4 q + 2 b4 + b5.  Same operand pairs as wsum per value.
template <int N>
__device__ __forceinline__ void wsum_rows(const double (&in)[N], double (&out)[(N + 3) / 4]) {
    constexpr int N2 = (N + 1) / 2, N4 = (N + 3) / 4;
    double r1[2 * N4];
#pragma unroll
    for (int q = 0; q < 2 * N4; q++) {
        double a = 2 * q < N ? in[2 * q] : 0.0, b = 2 * q + 1 < N ? in[2 * q + 1] : 0.0;
        if (q < N2) { swap32_f64(a, b); r1[q] = a + b; }   // lanes < 32: value 2 q, lanes >= 32: value 2 q + 1
        else r1[q] = 0.0;
    }
#pragma unroll
    for (int q = 0; q < N4; q++) {
        double a = r1[2 * q], b = r1[2 * q + 1];
        swap16_f64(a, b);                                  // rows 0, 2: register 2 q; rows 1, 3: register 2 q + 1
        out[q] = row16_sum(a + b);
    }
}

// block-wide sum / max of up to 2 values; all threads get identical results
__device__ __forceinline__ void block_red2(double& a, double& b, bool is_max, LbaShared& S, int& phase) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const double ra = is_max ? wmax(a) : wsum(a), rb = is_max ? wmax(b) : wsum(b);
    if (lane == 0) { S.red[phase][wv][0] = ra; S.red[phase][wv][1] = rb; }
    __syncthreads();
    double sa = S.red[phase][0][0], sb = S.red[phase][0][1];
    for (int w = 1; w < kLbaWaves; w++) {
        if (is_max) { sa = fmax(sa, S.red[phase][w][0]); sb = fmax(sb, S.red[phase][w][1]); }
        else { sa += S.red[phase][w][0]; sb += S.red[phase][w][1]; }
    }
    a = sa; b = sb;
    phase ^= 1;
}

__global__ __launch_bounds__(kLbaThreads) void k_lba(const LbaProblem* probs) {
    const LbaProblem& pr = probs[blockIdx.x];   // uniform, read-only: fields come from scalar loads on use
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int K = pr.K, P = pr.P, E = pr.E;
    __shared__ LbaShared S;
    extern __shared__ __align__(16) uint8_t dyn[];
    double* rowbuf = (double*)dyn;   // scratch: solution vector during back-substitution
    int phase = 0;
#ifdef OSLAM_LBA_PROFILE
    long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = clock64();
#endif

    const Cam cam = {(double)pr.K5[0], (double)pr.K5[1], (double)pr.K5[2], (double)pr.K5[3], (double)pr.K5[4]};
    const double dMono = (double)pr.delta_mono, dStereo = (double)pr.delta_stereo;

    // ---- setup ----
    if (tid == 0) {
        int nb = 0;
        for (int a = 0; a < K; a++) {
            if (pr.fixed[a]) S.blk[a] = -1;
            else { S.free_pose[nb] = a; S.blk[a] = nb++; }
        }
        S.flag = nb;
    }
    for (int a = tid; a < K; a += kLbaThreads) {
        S.T[a] = se3_from_T(pr.poses + a * 16);
        se3_R(S.T[a], S.R[a]);
    }
    for (int i = tid; i < P * 3; i += kLbaThreads) pr.Xa[i] = (double)pr.points[i];
    for (int e = tid; e < E; e += kLbaThreads) { pr.level[e] = 0; pr.chi2[e] = 0; pr.erase[e] = 0; }
    __syncthreads();
    const int nfree = S.flag;
    const int n = 6 * nfree, ld = n + 1;

    double* X = pr.Xa;    // current points
    double* Xn = pr.Xb;   // trial points
    int st_its0 = 0, st_its1 = 0, st_trials0 = 0, st_trials1 = 0;
    // force-stop flag (g2o setForceStopFlag): one lane polls, the workgroup agrees on the value
    auto stopped = [&]() -> bool {
        if (!pr.stop) return false;
        __syncthreads();
        if (tid == 0) S.stopflag = *pr.stop;
        __syncthreads();
        return S.stopflag != 0;
    };

    bool early = stopped();   // reference :655-657
    bool robust = pr.robust0 != 0;

    // residual pass (thread per point) at (poses Tp, points Xp): stores chi2 per active edge, returns robust sum
    auto eval = [&](const SE3* Tp, const double* Xp) -> double {
        double F = 0;
        for (int p = tid; p < P; p += kLbaThreads) {
            const double Xw[3] = {Xp[p * 3], Xp[p * 3 + 1], Xp[p * 3 + 2]};
            for (int e = pr.pt_start[p]; e < pr.pt_start[p + 1]; e++) {
                if (pr.level[e] != 0) continue;
                const int a = pr.e_kf[e];
                const float ur = pr.e_obs[e * 3 + 2];
                const bool stereo = !(ur < 0);
                const double ob[3] = {(double)pr.e_obs[e * 3], (double)pr.e_obs[e * 3 + 1], (double)ur};
                double pc[3], er[3];
                se3_map(Tp[a], Xw, pc);
                const double c2 = edge_error(cam, pc, ob, stereo, (double)pr.e_info[e], er);
                pr.chi2[e] = c2;
                if (robust) { double r0, r1; huber(c2, stereo ? dStereo : dMono, r0, r1); F += r0; }
                else F += c2;
            }
        }
        return F;
    };

    for (int stage = 0; stage < pr.nstages && !early; stage++) {
        const int iters = stage == 0 ? pr.iters0 : pr.iters1;
        double lambda = 0, ni = 2;
        bool ok = true;
        for (int iter = 0; iter < iters && !stopped() && ok; iter++) {
            // ---------- linearise: point-major (errors, Hll, b_l, Hpl) ----------
            double F0 = 0, dmax = 0;
            for (int p = tid; p < P; p += kLbaThreads) {
                const double Xw[3] = {X[p * 3], X[p * 3 + 1], X[p * 3 + 2]};
                double hl[6] = {0, 0, 0, 0, 0, 0}, bl[3] = {0, 0, 0};
                for (int e = pr.pt_start[p]; e < pr.pt_start[p + 1]; e++) {
                    if (pr.level[e] != 0) continue;
                    const int a = pr.e_kf[e];
                    const float ur = pr.e_obs[e * 3 + 2];
                    const bool stereo = !(ur < 0);
                    const double ob[3] = {(double)pr.e_obs[e * 3], (double)pr.e_obs[e * 3 + 1], (double)ur};
                    const double info = (double)pr.e_info[e];
                    double pc[3], er[3], Jp[18], Jx[9];
                    se3_map(S.T[a], Xw, pc);
                    const double c2 = edge_error(cam, pc, ob, stereo, info, er);
                    pr.chi2[e] = c2;
                    double r0 = c2, w = 1.0;
                    if (robust) huber(c2, stereo ? dStereo : dMono, r0, w);
                    F0 += r0;
                    jac_binary(cam, pc, S.R[a], stereo, Jp, Jx);
                    const double wi = w * info;
                    // mono edges carry a zero third row (J, e): the 3-row loops add exact zeros
                    int k = 0;
#pragma unroll
                    for (int i = 0; i < 3; i++) {
                        double sb = 0;
                        _Pragma("unroll") for (int d = 0; d < 3; d++) sb += Jx[d * 3 + i] * (info * er[d]);
                        bl[i] -= w * sb;
#pragma unroll
                        for (int j = i; j < 3; j++) {
                            double sh = 0;
                            _Pragma("unroll") for (int d = 0; d < 3; d++) sh += Jx[d * 3 + i] * wi * Jx[d * 3 + j];
                            hl[k++] += sh;
                        }
                    }
                    if (S.blk[a] >= 0) {
                        double* B = pr.Hpl + (long long)e * 18;
#pragma unroll
                        for (int i = 0; i < 6; i++)
#pragma unroll
                            for (int j = 0; j < 3; j++) {
                                double sh = 0;
                                _Pragma("unroll") for (int d = 0; d < 3; d++) sh += Jp[d * 6 + i] * wi * Jx[d * 3 + j];
                                B[i * 3 + j] = sh;
                            }
                    }
                }
                double* H = pr.Hll + (long long)p * 9;
                H[0] = hl[0]; H[1] = hl[1]; H[2] = hl[2]; H[3] = hl[1]; H[4] = hl[3]; H[5] = hl[4]; H[6] = hl[2]; H[7] = hl[4]; H[8] = hl[5];
                pr.bl[p * 3] = bl[0]; pr.bl[p * 3 + 1] = bl[1]; pr.bl[p * 3 + 2] = bl[2];
                dmax = fmax(dmax, fmax(fabs(hl[0]), fmax(fabs(hl[3]), fabs(hl[5]))));
            }
            LBA_STAMP(0);
            // ---------- linearise: pose-major (Hpp, b_p), one wavefront per keyframe ----------
            for (int a = wv; a < K; a += kLbaWaves) {
                if (S.blk[a] < 0) continue;
                double acc[27];
#pragma unroll
                for (int k = 0; k < 27; k++) acc[k] = 0;
                for (int q = pr.pose_start[a] + lane; q < pr.pose_start[a + 1]; q += 64) {
                    const int e = pr.pose_edges[q];
                    if (pr.level[e] != 0) continue;
                    const int p = pr.e_pt[e];
                    const double Xw[3] = {X[p * 3], X[p * 3 + 1], X[p * 3 + 2]};
                    const float ur = pr.e_obs[e * 3 + 2];
                    const bool stereo = !(ur < 0);
                    const double ob[3] = {(double)pr.e_obs[e * 3], (double)pr.e_obs[e * 3 + 1], (double)ur};
                    const double info = (double)pr.e_info[e];
                    double pc[3], er[3], Jp[18], Jx[9];
                    se3_map(S.T[a], Xw, pc);
                    const double c2 = edge_error(cam, pc, ob, stereo, info, er);
                    double r0 = c2, w = 1.0;
                    if (robust) huber(c2, stereo ? dStereo : dMono, r0, w);
                    jac_binary(cam, pc, S.R[a], stereo, Jp, Jx);
                    const double wi = w * info;
                    // mono edges carry a zero third row (J, e): the 3-row loops add exact zeros
                    int k = 0;
#pragma unroll
                    for (int i = 0; i < 6; i++) {
                        double sb = 0;
                        _Pragma("unroll") for (int d = 0; d < 3; d++) sb += Jp[d * 6 + i] * (info * er[d]);
                        acc[21 + i] -= w * sb;
#pragma unroll
                        for (int j = i; j < 6; j++) {
                            double sh = 0;
                            _Pragma("unroll") for (int d = 0; d < 3; d++) sh += Jp[d * 6 + i] * wi * Jp[d * 6 + j];
                            acc[k++] += sh;
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < 27; k++) acc[k] = wsum(acc[k]);
                if (lane == 0) {
                    double* H = pr.Hpp + a * 36;
                    int k = 0;
                    for (int i = 0; i < 6; i++)
                        for (int j = i; j < 6; j++) { H[i * 6 + j] = acc[k]; H[j * 6 + i] = acc[k]; k++; }
                    for (int i = 0; i < 6; i++) pr.bp[a * 6 + i] = acc[21 + i];
                }
            }
            LBA_STAMP(1);
            block_red2(F0, dmax, false, S, phase);   // F0 summed (dmax unused: lambda init recomputes the maxima)
            double currentChi = F0;
            if (iter == 0) {   // computeLambdaInit over pose and point diagonals
                double m1 = 0, m2 = 0;
                for (int p = tid; p < P; p += kLbaThreads) {
                    const double* H = pr.Hll + (long long)p * 9;
                    m1 = fmax(m1, fmax(fabs(H[0]), fmax(fabs(H[4]), fabs(H[8]))));
                }
                for (int a = tid; a < K; a += kLbaThreads)
                    if (S.blk[a] >= 0)
                        for (int i = 0; i < 6; i++) m2 = fmax(m2, fabs(pr.Hpp[a * 36 + i * 7]));
                block_red2(m1, m2, true, S, phase);
                lambda = 1e-5 * fmax(m1, m2);
                ni = 2;
            }

            double rho = 0;
            int qmax = 0;
            do {
                // ---------- D^-1 and D^-1 b_l per point ----------
                for (int p = tid; p < P; p += kLbaThreads) {
                    double D[9], Di[9];
                    const double* H = pr.Hll + (long long)p * 9;
                    for (int i = 0; i < 9; i++) D[i] = H[i];
                    D[0] += lambda; D[4] += lambda; D[8] += lambda;
                    inv3(D, Di);
                    double* o = pr.Dinv + (long long)p * 9;
                    for (int i = 0; i < 9; i++) o[i] = Di[i];
                    const double* b = pr.bl + p * 3;
                    for (int i = 0; i < 3; i++) pr.xl[p * 3 + i] = Di[i * 3] * b[0] + Di[i * 3 + 1] * b[1] + Di[i * 3 + 2] * b[2];   // db
                }
                __syncthreads();
                LBA_STAMP(2);
                // ---------- Schur complement: one wavefront per (a <= b) block of the reduced system ----------
                // Hs(a,b) = [a==b](Hpp_a + lambda I) - sum_l B_al Dinv_l B_bl^T over points l seen by both;
                // lanes stride over keyframe a's observations and look the partner edge up in point l's
                // (short) edge list; fixed lane assignment + shuffle tree => deterministic sums.
                {
                    const int nblk = nfree * (nfree + 1) / 2;
                    for (int t = wv; t < nblk; t += kLbaWaves) {
                        // unrank t -> (ba <= bb) in row-major order of the upper triangle
                        int ba = 0, rem = t;
                        while (rem >= nfree - ba) { rem -= nfree - ba; ba++; }
                        const int bb = ba + rem;
                        const int a = S.free_pose[ba], b2 = S.free_pose[bb];
                        double acc[36];
#pragma unroll
                        for (int i = 0; i < 36; i++) acc[i] = 0;
                        double bsv[6] = {0, 0, 0, 0, 0, 0};
                        for (int q = pr.pose_start[a] + lane; q < pr.pose_start[a + 1]; q += 64) {
                            const int e = pr.pose_edges[q];
                            if (pr.level[e] != 0) continue;
                            const int p = pr.e_pt[e];
                            int e2 = -1;
                            if (a == b2) e2 = e;
                            else
                                for (int c2 = pr.pt_start[p]; c2 < pr.pt_start[p + 1]; c2++)
                                    if (pr.e_kf[c2] == b2) { e2 = c2; break; }
                            if (e2 < 0 || pr.level[e2] != 0) continue;
                            const double* Ba = pr.Hpl + (long long)e * 18;
                            const double* Bb = pr.Hpl + (long long)e2 * 18;
                            const double* Di = pr.Dinv + (long long)p * 9;
                            double BD[18];
#pragma unroll
                            for (int i = 0; i < 6; i++) {
                                const double b0 = Ba[i * 3], b1 = Ba[i * 3 + 1], b2v = Ba[i * 3 + 2];
                                BD[i * 3] = b0 * Di[0] + b1 * Di[3] + b2v * Di[6];
                                BD[i * 3 + 1] = b0 * Di[1] + b1 * Di[4] + b2v * Di[7];
                                BD[i * 3 + 2] = b0 * Di[2] + b1 * Di[5] + b2v * Di[8];
                            }
                            if (a == b2) {
                                const double* db = pr.xl + p * 3;
#pragma unroll
                                for (int i = 0; i < 6; i++) bsv[i] += Ba[i * 3] * db[0] + Ba[i * 3 + 1] * db[1] + Ba[i * 3 + 2] * db[2];
                            }
#pragma unroll
                            for (int i = 0; i < 6; i++)
#pragma unroll
                                for (int j = 0; j < 6; j++)
                                    acc[i * 6 + j] += BD[i * 3] * Bb[j * 3] + BD[i * 3 + 1] * Bb[j * 3 + 1] + BD[i * 3 + 2] * Bb[j * 3 + 2];
                        }
#pragma unroll
                        for (int i = 0; i < 36; i++) acc[i] = wsum(acc[i]);
                        if (a == b2) {
#pragma unroll
                            for (int i = 0; i < 6; i++) bsv[i] = wsum(bsv[i]);
                        }
                        if (lane == 0) {
                            for (int i = 0; i < 6; i++)
                                for (int j = 0; j < 6; j++) {
                                    double v = -acc[i * 6 + j];
                                    if (a == b2) v += pr.Hpp[a * 36 + i * 6 + j] + (i == j ? lambda : 0.0);
                                    pr.Hs[(size_t)(6 * ba + i) * ld + 6 * bb + j] = v;
                                }
                            if (a == b2)
                                for (int i = 0; i < 6; i++) pr.Hs[(size_t)(6 * ba + i) * ld + n] = pr.bp[a * 6 + i] - bsv[i];
                        }
                    }
                }
                __syncthreads();
                LBA_STAMP(3);
                // ---------- blocked Cholesky U^T U of the reduced system, rhs as column n ----------
                if (tid == 0) S.flag = 1;
                __syncthreads();
                for (int j0 = 0; j0 < n; j0 += 6) {
                    if (wv == 0) {   // panel: factor the 6x6 diagonal block, scale the 6 pivot rows
                        double Dg[36];
#pragma unroll
                        for (int i = 0; i < 6; i++)
#pragma unroll
                            for (int k = 0; k < 6; k++) Dg[i * 6 + k] = k >= i ? pr.Hs[(size_t)(j0 + i) * ld + j0 + k] : 0.0;
                        bool good = true;
#pragma unroll
                        for (int j = 0; j < 6; j++) {   // U^T U on the 6x6 (all lanes redundantly)
                            double d = Dg[j * 6 + j];
                            if (!(d > 0) || !(d < 1.7e308)) { good = false; d = 1; }
                            d = sqrt(d);
                            Dg[j * 6 + j] = d;
#pragma unroll
                            for (int k = j + 1; k < 6; k++) Dg[j * 6 + k] /= d;
#pragma unroll
                            for (int i = j + 1; i < 6; i++)
#pragma unroll
                                for (int k = i; k < 6; k++) Dg[i * 6 + k] -= Dg[j * 6 + i] * Dg[j * 6 + k];
                        }
                        if (!good && lane == 0) S.flag = 0;
                        for (int k = j0 + 6 + lane; k <= n; k += 64) {   // columns right of the block (incl. rhs)
                            double col[6];
#pragma unroll
                            for (int i = 0; i < 6; i++) col[i] = pr.Hs[(size_t)(j0 + i) * ld + k];
#pragma unroll
                            for (int j = 0; j < 6; j++) {   // forward substitution with U_diag^T
                                double s = col[j];
#pragma unroll
                                for (int i = 0; i < j; i++) s -= Dg[i * 6 + j] * col[i];
                                col[j] = s / Dg[j * 6 + j];
                            }
#pragma unroll
                            for (int i = 0; i < 6; i++) pr.Hs[(size_t)(j0 + i) * ld + k] = col[i];
                        }
                        if (lane < 36) {
                            const int r = lane / 6, cc = lane % 6;
                            if (cc >= r) pr.Hs[(size_t)(j0 + r) * ld + j0 + cc] = Dg[lane];
                        }
                    }
                    __syncthreads();
                    // trailing update: A[i][k] -= sum_r P[r][i] P[r][k], i >= j0+6, k >= i (and rhs column)
                    const int m = n - (j0 + 6);
                    for (int ii = wv; ii < m; ii += kLbaWaves) {
                        const int i = j0 + 6 + ii;
                        double pi[6];
#pragma unroll
                        for (int r = 0; r < 6; r++) pi[r] = pr.Hs[(size_t)(j0 + r) * ld + i];
                        for (int k = i + lane; k <= n; k += 64) {
                            double s = 0;
#pragma unroll
                            for (int r = 0; r < 6; r++) s += pi[r] * pr.Hs[(size_t)(j0 + r) * ld + k];
                            pr.Hs[(size_t)i * ld + k] -= s;
                        }
                    }
                    __syncthreads();
                }
                const bool ok2 = S.flag != 0;
                // back substitution U x = y (one wavefront)
                if (wv == 0 && ok2) {
                    double* xs = rowbuf;   // row buffers are idle here; LDS keeps the wave's stores/loads ordered
                    for (int i = n - 1; i >= 0; i--) {
                        double s = 0;
                        for (int k = i + 1 + lane; k < n; k += 64) s += pr.Hs[(size_t)i * ld + k] * xs[k];
                        s = wsum(s);
                        if (lane == 0) xs[i] = (pr.Hs[(size_t)i * ld + n] - s) / pr.Hs[(size_t)i * ld + i];
                    }
                    for (int i = lane; i < n; i += 64) pr.xp[i] = xs[i];
                }
                if (!ok2) for (int i = tid; i < n; i += kLbaThreads) pr.xp[i] = 0;
                __syncthreads();
                LBA_STAMP(4);
                // ---------- landmarks: x_l = Dinv (b_l - sum_a B_a^T x_a); trial state ----------
                double sc = 0;
                for (int p = tid; p < P; p += kLbaThreads) {
                    double cl[3] = {pr.bl[p * 3], pr.bl[p * 3 + 1], pr.bl[p * 3 + 2]};
                    double xo[3] = {0, 0, 0};
                    if (ok2) {
                        for (int e = pr.pt_start[p]; e < pr.pt_start[p + 1]; e++) {
                            if (pr.level[e] != 0) continue;
                            const int ba = S.blk[pr.e_kf[e]];
                            if (ba < 0) continue;
                            const double* B = pr.Hpl + (long long)e * 18;
                            const double* xa = pr.xp + 6 * ba;
                            for (int j = 0; j < 3; j++) {
                                double s = 0;
                                for (int i = 0; i < 6; i++) s += B[i * 3 + j] * xa[i];
                                cl[j] -= s;
                            }
                        }
                        const double* Di = pr.Dinv + (long long)p * 9;
                        for (int i = 0; i < 3; i++) xo[i] = Di[i * 3] * cl[0] + Di[i * 3 + 1] * cl[1] + Di[i * 3 + 2] * cl[2];
                    }
                    for (int i = 0; i < 3; i++) {
                        Xn[p * 3 + i] = X[p * 3 + i] + xo[i];
                        sc += xo[i] * (lambda * xo[i] + pr.bl[p * 3 + i]);   // computeScale, landmark part
                    }
                }
                for (int a = tid; a < K; a += kLbaThreads) {
                    const int ba = S.blk[a];
                    if (ba < 0) { S.Tn[a] = S.T[a]; continue; }
                    double xa[6];
                    for (int i = 0; i < 6; i++) { xa[i] = pr.xp[6 * ba + i]; sc += xa[i] * (lambda * xa[i] + pr.bp[a * 6 + i]); }
                    S.Tn[a] = se3_mul(se3_exp(xa), S.T[a]);
                }
                __syncthreads();
                LBA_STAMP(5);
                double F1 = eval(S.Tn, Xn);
                block_red2(F1, sc, false, S, phase);
                double tempChi = F1;
                if (!ok2) tempChi = 1.7976931348623157e308;
                rho = (currentChi - tempChi) / (sc + 1e-3);
                const bool finite = (tempChi - tempChi) == 0;
                if (rho > 0 && finite) {
                    double alpha = 1. - (2 * rho - 1) * (2 * rho - 1) * (2 * rho - 1);
                    alpha = fmin(alpha, 2. / 3.);
                    lambda *= fmax(1. / 3., alpha);
                    ni = 2;
                    currentChi = tempChi;
                    for (int a = tid; a < K; a += kLbaThreads) { S.T[a] = S.Tn[a]; se3_R(S.T[a], S.R[a]); }
                    { double* t = X; X = Xn; Xn = t; }
                } else {
                    lambda *= ni;
                    ni *= 2;
                }
                __syncthreads();
                LBA_STAMP(6);
                qmax++;
                if (stage == 0) st_trials0++; else st_trials1++;
            } while (rho < 0 && qmax < 10 && !stopped());
            if (stage == 0) st_its0++; else st_its1++;
            if (qmax == 10 || rho == 0) ok = false;
        }
        if (stage == 0 && pr.nstages > 1) {
            if (stopped()) break;   // bDoMore = false (:664-666)
            // gate observations, drop the robust kernel (:672-702)
            for (int e = tid; e < E; e += kLbaThreads) {
                const int a = pr.e_kf[e], p = pr.e_pt[e];
                const bool stereo = !(pr.e_obs[e * 3 + 2] < 0);
                const double Xw[3] = {X[p * 3], X[p * 3 + 1], X[p * 3 + 2]};
                double pc[3];
                se3_map(S.T[a], Xw, pc);
                if (pr.chi2[e] > (stereo ? 7.815 : 5.991) || !(pc[2] > 0.0)) pr.level[e] = 1;
            }
            robust = false;
            __syncthreads();
        }
    }
    __syncthreads();
    // ---- outputs (:711-777) ----
    if (!early) {
        for (int e = tid; e < E; e += kLbaThreads) {
            const int a = pr.e_kf[e], p = pr.e_pt[e];
            const bool stereo = !(pr.e_obs[e * 3 + 2] < 0);
            const double Xw[3] = {X[p * 3], X[p * 3 + 1], X[p * 3 + 2]};
            double pc[3];
            se3_map(S.T[a], Xw, pc);
            pr.erase[e] = (pr.chi2[e] > (stereo ? 7.815 : 5.991) || !(pc[2] > 0.0)) ? 1 : 0;
        }
    }
    for (int a = tid; a < K; a += kLbaThreads) {
        if (pr.fixed[a] != 1 && !early) se3_to_T(S.T[a], pr.poses_out + a * 16);
        else for (int i = 0; i < 16; i++) pr.poses_out[a * 16 + i] = pr.poses[a * 16 + i];
    }
    for (int i = tid; i < P * 3; i += kLbaThreads) pr.points_out[i] = early ? pr.points[i] : (float)X[i];
#ifdef OSLAM_LBA_PROFILE
    if (tid == 0) for (int i = 0; i < 8; i++) pr.stats[8 + i] = (int)(prof[i] / 1000);
#endif
    if (tid == 0) { pr.stats[0] = st_its0; pr.stats[1] = st_trials0; pr.stats[2] = st_its1; pr.stats[3] = st_trials1; }
}


// Jacobians of the binary edges (se3_math.h jac_binary) with ONE reciprocal: every x / z, y / z^2 ... of the g2o formulas becomes a product with iz = 1 / z,
// iz2 = iz * iz (an fp64 division is ~12 dependent instructions on this part and jac_binary has 28 of them; the one-workgroup layout recomputes the Jacobians of an edge in
// four places per LM trial, the multi-launch layout in two: k_w_lin's point and pose roles).  Entries differ from the divided form by a few ulp — far inside the 1e-4 bar; the residual (chi2, the erase decisions) keeps the
// reference's own arithmetic in edge_error.
__device__ __forceinline__ void jac_binary_rcp(const Cam& c, const double p[3], const double R[9], bool stereo, double Jp[18], double Jx[9]) {
#pragma clang fp contract(fast)
    const double x = p[0], y = p[1], iz = 1.0 / p[2], iz2 = iz * iz;
    const double fxiz = c.fx * iz, fyiz = c.fy * iz, xiz2 = x * iz2, yiz2 = y * iz2;
    Jp[0] = x * yiz2 * c.fx; Jp[1] = -(1 + (x * xiz2)) * c.fx; Jp[2] = y * fxiz;
    Jp[3] = -fxiz; Jp[4] = 0; Jp[5] = xiz2 * c.fx;
    Jp[6] = (1 + y * yiz2) * c.fy; Jp[7] = -x * yiz2 * c.fy; Jp[8] = -x * fyiz;
    Jp[9] = 0; Jp[10] = -fyiz; Jp[11] = yiz2 * c.fy;
    if (!stereo) {
        const double t2 = xiz2 * c.fx, t5 = yiz2 * c.fy;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            Jx[k] = -fxiz * R[k] + t2 * R[6 + k];
            Jx[3 + k] = -fyiz * R[3 + k] + t5 * R[6 + k];
            Jx[6 + k] = 0;
        }
        Jp[12] = Jp[13] = Jp[14] = Jp[15] = Jp[16] = Jp[17] = 0;
    } else {
        const double t2 = xiz2 * c.fx, t5 = yiz2 * c.fy, tb = c.bf * iz2;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            Jx[k] = -fxiz * R[k] + t2 * R[6 + k];
            Jx[3 + k] = -fyiz * R[3 + k] + t5 * R[6 + k];
            Jx[6 + k] = Jx[k] - tb * R[6 + k];
        }
        Jp[12] = Jp[0] - tb * y; Jp[13] = Jp[1] + tb * x; Jp[14] = Jp[2];
        Jp[15] = Jp[3]; Jp[16] = 0; Jp[17] = Jp[5] - tb;
    }
}


// ==========================================================================================
// Wide mode: the same LM schedule spread over the whole GPU as a fixed sequence of small kernels
// per LM trial, with the Levenberg-Marquardt control flow (accept / reject, lambda, iteration and
// stage transitions, force-stop polling) decided on the device in LbaCtrl.  Every kernel reads
// the control block and returns immediately when its phase is not due, so the host only enqueues
// "trial slots" and checks ctrl.done every few slots.  All reductions are two-level with a fixed
// order (per-block partials summed in block order), so results do not depend on scheduling.
// ==========================================================================================
struct LbaCtrl {
    int stage, iter, qmax, need_lin, gate, done, ok2, cur, robust, ok, early;
    int its[2], trials[2];
    int nfree, n;
    int ticket[2];   // blocks finished in k_w_lin / k_w_eval: the last one runs the LM control step
    double lambda, ni, currentChi, rho;
    double lambdaA, FA;   // LM control after a linearisation, pair-gather path: written by block 0 of k_w_edgeW, committed by the next k_w_ctrlB (see w_ctrlA_values)
};

struct LbaWide {
    LbaCtrl* ct;
    SE3* T;          // [2][K]
    double* R;       // [2][K][9]
    int* blk;        // [K]
    int* free_pose;  // [K]
    double* partF;   // [nblk_pt]
    double* partS;   // [nblk_pt + 1] scale partials (last = poses)
    double* partM;   // [nblk_pt] max |Hll diag|
    int nblk_pt;
    const int2* pairs;        // (edge in pose a, edge in pose b) of every point both poses see, grouped by Schur block, point order
    const int* pair_start;    // [nfree*(nfree+1)/2 + 1]
    uint16_t* pairM; int2* pairs_w; int* pair_start_w;   // pair lists built on the device (k_w_pair_*): then pairs / pair_start point at pairs_w / pair_start_w
    double* W;                // [E][18]: Hpl_e * (Hll_p + lambda I)^-1, written by k_w_edgeW for the current trial
    float4* eoi;              // [E] (u, v, u_R, invSigma2) of every edge in one 16-byte record (k_w_init_arrays packs e_obs / e_info once per call), or null
    double* rec;              // [E][4] compact edge records of the current linearisation (x, y, 1/z of the point in the camera frame, weight x information with the
                              // sign bit = monocular), or null: the per-edge blocks B_e are then materialised in pr.Hpl (see k_w_schur_rec)
    // Schur complement by tiles (k_w_schur_tiles / k_w_schur_sum, lba_win.inc): the structures of LbaWin on the point-major edge numbering
    const int* tile_p0; const int* tile_s0; const int* stg_edge; const int* thr_own; const int* blk_thr; const int* blk_slots;
    double* parts;            // [nwg][ngroup * kWinThreads][42] per-workgroup, per-slot block sums
    int ntile, ngroup, nwg, TE, TP, nblk;
};

constexpr int kWPt = 128;   // threads per block of the per-point kernels


__device__ __forceinline__ double* w_X(const LbaProblem& pr, int which) { return which ? pr.Xb : pr.Xa; }

__global__ __launch_bounds__(256) void k_w_init(const LbaProblem* probs, const LbaWide* ws) {
    const LbaProblem& pr = probs[blockIdx.y];
    const LbaWide& w = ws[blockIdx.y];
    const int tid = threadIdx.x;
    LbaCtrl* ct = w.ct;
    if (tid == 0) {
        int nb = 0;
        for (int a = 0; a < pr.K; a++) {
            if (pr.fixed[a]) w.blk[a] = -1;
            else { w.free_pose[nb] = a; w.blk[a] = nb++; }
        }
        ct->stage = 0; ct->iter = 0; ct->qmax = 0; ct->need_lin = 1; ct->gate = 0; ct->done = 0; ct->ok2 = 1; ct->cur = 0; ct->robust = pr.robust0; ct->ok = 1;
        ct->its[0] = ct->its[1] = ct->trials[0] = ct->trials[1] = 0;
        ct->ticket[0] = ct->ticket[1] = 0;
        ct->nfree = nb; ct->n = 6 * nb;
        ct->lambda = 0; ct->ni = 2; ct->currentChi = 0; ct->rho = 0; ct->lambdaA = 0; ct->FA = 0;
        ct->early = pr.stop ? (*pr.stop != 0) : 0;   // reference :655-657
        if (ct->early) ct->done = 1;
    }
    for (int a = tid; a < pr.K; a += 256) {
        w.T[a] = se3_from_T(pr.poses + a * 16);
        se3_R(w.T[a], w.R + a * 9);
    }
}

// state arrays of a fresh problem (grid-wide; k_w_init's single block only sets the control block and the poses)
__global__ __launch_bounds__(256) void k_w_init_arrays(const LbaProblem* probs, const LbaWide* ws) {
    const LbaProblem& pr = probs[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < pr.P * 3) pr.Xa[i] = (double)pr.points[i];
    if (i < pr.E) {
        pr.level[i] = 0; pr.chi2[i] = 0; pr.erase[i] = 0;
        if (ws && ws[blockIdx.y].eoi) ws[blockIdx.y].eoi[i] = make_float4(pr.e_obs[i * 3], pr.e_obs[i * 3 + 1], pr.e_obs[i * 3 + 2], pr.e_info[i]);
    }
}

// Between the two optimisation stages (reference src/Optimizer.cc:668-689): edges with chi2 above the gate or behind the
// camera leave the problem (level 1).  Runs once per LBA, inside the last block of the k_w_eval launch that ended stage 0.
__device__ void w_gate_block(const LbaProblem& pr, const LbaWide& w) {
    const LbaCtrl* ct = w.ct;
    const double* X = w_X(pr, ct->cur);
    const SE3* T = w.T + ct->cur * pr.K;
    for (int e = threadIdx.x; e < pr.E; e += blockDim.x) {
        const int a = pr.e_kf[e], p = pr.e_pt[e];
        const bool stereo = !(pr.e_obs[e * 3 + 2] < 0);
        const double Xw[3] = {X[p * 3], X[p * 3 + 1], X[p * 3 + 2]};
        double pc[3];
        se3_map(T[a], Xw, pc);
        if (pr.chi2[e] > (stereo ? 7.815 : 5.991) || !(pc[2] > 0.0)) pr.level[e] = 1;
    }
}

// Linearisation of one LM iteration in ONE launch: blocks [0, nblk_pt) own kWPt landmarks each (4 lanes per landmark
// split its edge list, combined by a fixed xor-shuffle tree: Hll, bl, Hpl blocks, robust chi2 partial), blocks
// [nblk_pt, nblk_pt + K) own one keyframe each (8 wavefronts split its edge list: Hpp, bp).  Both roles are independent,
// so they overlap instead of running back to back.
constexpr int kLinThreads = 4 * kWPt;   // 512
constexpr int kLinKfLds = 128;          // keyframes of a window whose rotation + translation k_w_lin's landmark blocks stage in LDS (12 KB)

__device__ void w_ctrlA(const LbaProblem& pr, const LbaWide& w);
__device__ void w_ctrlB(const LbaProblem& pr, const LbaWide& w, int win);

#ifndef OSLAM_LIN_MIN_WAVES
#define OSLAM_LIN_MIN_WAVES 4   // 128 VGPRs (96 B of scratch per lane) so that two 512-thread workgroups share a CU; 1 = the compiler's 166 VGPRs, one workgroup per CU
#endif
// Workgroups are dispatched to the 8 XCDs round robin in linear block order, and every XCD has its own 4 MB L2.  The per-window kernels below map
// (window, item) so that ALL workgroups of a window run on ONE XCD (window w -> XCD w mod 8; the windows of an XCD one after the other): the window's per-edge
// blocks (B_e, W_e: 288 B per edge, read once per pair they take part in) then come out of that XCD's L2 instead of being pulled into all eight.
// Launch grid: dim3(items per window, windows rounded up to a multiple of 8).
// Calls with fewer than 8 windows keep the plain mapping (window = blockIdx.y: a window's workgroups spread over all XCDs), signalled by gridDim.y == nwin.
__device__ __forceinline__ bool xcd_window_item(int nwin, int& win, int& item) {
    if ((int)gridDim.y == nwin && (nwin & 7)) { win = blockIdx.y; item = blockIdx.x; return true; }
    const int L = blockIdx.x + gridDim.x * blockIdx.y;
    const int xcd = L & 7, j = L >> 3;
    const int grp = j / (int)gridDim.x;
    item = j - grp * (int)gridDim.x;
    win = grp * 8 + xcd;
    return win < nwin;
}

template <bool REC>
__global__ __launch_bounds__(kLinThreads, OSLAM_LIN_MIN_WAVES) void k_w_lin(const LbaProblem* probs, const LbaWide* ws, int nwin) {
    int win_, item_;
    if (!xcd_window_item(nwin, win_, item_)) return;
    const LbaProblem& pr = probs[win_];
    const LbaWide& w = ws[win_];
    const LbaCtrl* ct = w.ct;
    if (ct->done || !ct->need_lin) return;
    const Cam cam = {(double)pr.K5[0], (double)pr.K5[1], (double)pr.K5[2], (double)pr.K5[3], (double)pr.K5[4]};
    const double dMono = (double)pr.delta_mono, dStereo = (double)pr.delta_stereo;
    const bool robust = ct->robust != 0;
    const double* __restrict__ X = w_X(pr, ct->cur);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // (bases of the arrays the edge loops walk, read once: the loops store through pr.chi2 / w.rec, so the compiler reloaded every base it took from pr / w with a
    // scalar load + wait per edge: ISA)
    const uint8_t* __restrict__ level_g = pr.level;
    const int* __restrict__ e_kf_g = pr.e_kf;
    const int* __restrict__ e_pt_g = pr.e_pt;
    const float* __restrict__ e_obs_g = pr.e_obs;
    const float* __restrict__ e_info_g = pr.e_info;
    const float4* __restrict__ eoi_g = REC ? w.eoi : nullptr;
    double* __restrict__ chi2_g = pr.chi2;
    double* __restrict__ rec_g = REC ? w.rec : nullptr;
    __shared__ double sF[kLinThreads / 64], sM[kLinThreads / 64], sAcc[kLinThreads / 64][27];
    // Point role: rotation matrix + translation of the window's keyframes staged in LDS (12 doubles each) when the window has at most kLinKfLds of them: a
    // landmark's edges name different keyframes, and the 12 doubles of an edge's keyframe came as twelve 8-byte gathers per edge from global memory (the
    // kernel's largest group of vector-memory instructions); larger windows (fixed cameras are unbounded, src/Optimizer.cc:489-504) read them where they are.
    __shared__ double sRt[REC ? kLinKfLds * 12 : 1];
    if (item_ < w.nblk_pt) {
        const SE3* T = w.T + ct->cur * pr.K;
        const double* Rm = w.R + (size_t)ct->cur * pr.K * 9;
        const bool staged = REC && pr.K <= kLinKfLds;
        if (staged) {
            for (int i = tid; i < pr.K * 12; i += kLinThreads) {
                const int a = i / 12, j = i - a * 12;
                sRt[i] = j < 9 ? Rm[a * 9 + j] : T[a].t[j - 9];
            }
            __syncthreads();
        }
        const int p = item_ * kWPt + (tid >> 2), sub = tid & 3;
        double F0 = 0, dmax = 0;
        double hl[6] = {0, 0, 0, 0, 0, 0}, bl[3] = {0, 0, 0};
        if (p < pr.P) {
            const double Xw[3] = {X[p * 3], X[p * 3 + 1], X[p * 3 + 2]};
            auto edge = [&](int e, uint8_t lv, int a, float o0, float o1, float ur, float inf, const double* Ra_, const double* ta_) {
                if (lv != 0) {
                    if (REC) {   // weight 0: the edge adds exact zeros wherever its record is read
                        double2* rc = (double2*)(rec_g + (long long)e * 4);
                        rc[0] = make_double2(0.0, 0.0); rc[1] = make_double2(1.0, 0.0);
                    }
                    return;
                }
                const bool stereo = !(ur < 0);
                const double ob[3] = {(double)o0, (double)o1, (double)ur};
                const double info = (double)inf;
                // (se3_math.h fast forms: rotation-matrix map, one Newton reciprocal, fused multiply-adds, products over the non-zero Jacobian entries only)
                double pc[3], er[3], iz, Jx[9];
                map_rt(Ra_, ta_, Xw, pc);
                const double c2 = residual_fast(cam, pc, ob, stereo, info, er, iz);
                chi2_g[e] = c2;
                double r0 = c2, wgt = 1.0;
                if (robust) huber_fast(c2, stereo ? dStereo : dMono, r0, wgt);
                F0 += r0;
                point_jac_rows(cam, pc, iz, Ra_, stereo, Jx);
                const double wi = wgt * info;
                {
#pragma clang fp contract(fast)
                    double wJx[9];
#pragma unroll
                    for (int q = 0; q < 9; q++) wJx[q] = wi * Jx[q];   // (row ur of a monocular edge is zero)
#pragma unroll
                    for (int d = 0; d < 3; d++) {
                        int k = 0;
#pragma unroll
                        for (int i = 0; i < 3; i++) {
                            bl[i] -= wJx[d * 3 + i] * er[d];
#pragma unroll
                            for (int j = i; j < 3; j++) hl[k++] += wJx[d * 3 + i] * Jx[d * 3 + j];
                        }
                    }
                    if (REC) {
                        double2* rc = (double2*)(rec_g + (long long)e * 4);
                        rc[0] = make_double2(pc[0], pc[1]); rc[1] = make_double2(iz, stereo ? wi : -wi);
                    } else if (w.blk[a] >= 0) {
                        double Ju[6], Jv[6], Jr[6];
                        pose_jac_rows(cam, pc, iz, stereo, Ju, Jv, Jr);
                        double* B = pr.Hpl + (long long)e * 18;
#pragma unroll
                        for (int i = 0; i < 6; i++)
#pragma unroll
                            for (int j = 0; j < 3; j++) {
                                double sh = 0;   // Ju[4] = Jv[3] = Jr[4] = 0
                                if (i != 4) sh += Ju[i] * wJx[j];
                                if (i != 3) sh += Jv[i] * wJx[3 + j];
                                if (i != 4) sh += Jr[i] * wJx[6 + j];
                                B[i * 3 + j] = sh;
                            }
                    }
                }
            };
            const int e_end = pr.pt_start[p + 1];
            for (int e = pr.pt_start[p] + sub; e < e_end; e += 4) {
                // (the edge's inputs are requested together with its level byte, not after it; requesting them one edge AHEAD of the arithmetic as well was measured:
                // 55 -> 61 us per launch of 40 windows — the kernel already spills at its 128-register cap)
                const uint8_t lv = level_g[e];
                const int a = e_kf_g[e];
                float o0, o1, ur, inf;
                if (REC && eoi_g) { const float4 oi = eoi_g[e]; o0 = oi.x; o1 = oi.y; ur = oi.z; inf = oi.w; }
                else { o0 = e_obs_g[e * 3]; o1 = e_obs_g[e * 3 + 1]; ur = e_obs_g[e * 3 + 2]; inf = e_info_g[e]; }
                if (staged) edge(e, lv, a, o0, o1, ur, inf, sRt + a * 12, sRt + a * 12 + 9);
                else edge(e, lv, a, o0, o1, ur, inf, Rm + a * 9, T[a].t);
            }
        }
        // the 4 lanes of a landmark: (s0 + s1) + (s2 + s3)
#pragma unroll
        for (int k = 0; k < 6; k++) { hl[k] += lane_xor1(hl[k]); hl[k] += lane_xor2(hl[k]); }
#pragma unroll
        for (int k = 0; k < 3; k++) { bl[k] += lane_xor1(bl[k]); bl[k] += lane_xor2(bl[k]); }
        F0 += lane_xor1(F0); F0 += lane_xor2(F0);
        if (p < pr.P && sub == 0) {
            double* H = pr.Hll + (long long)p * 9;
            H[0] = hl[0]; H[1] = hl[1]; H[2] = hl[2]; H[3] = hl[1]; H[4] = hl[3]; H[5] = hl[4]; H[6] = hl[2]; H[7] = hl[4]; H[8] = hl[5];
            pr.bl[p * 3] = bl[0]; pr.bl[p * 3 + 1] = bl[1]; pr.bl[p * 3 + 2] = bl[2];
            dmax = fmax(fabs(hl[0]), fmax(fabs(hl[3]), fabs(hl[5])));
        }
        if (sub != 0) F0 = 0;
        const double f = wsum(F0), m = wmax(dmax);
        if (lane == 0) { sF[wv] = f; sM[wv] = m; }
        __syncthreads();
        if (tid == 0) {
            double a = sF[0], bm = sM[0];
            for (int i = 1; i < kLinThreads / 64; i++) { a += sF[i]; bm = fmax(bm, sM[i]); }
            w.partF[item_] = a;
            w.partM[item_] = bm;
        }
        return;
    }
    const int a = item_ - w.nblk_pt;
    if (a >= pr.K || w.blk[a] < 0) return;   // fixed keyframe, or a padding block of a batched launch
    const SE3 Ta = w.T[ct->cur * pr.K + a];
    const double* Ra = w.R + ((size_t)ct->cur * pr.K + a) * 9;
    double acc[27];
#pragma unroll
    for (int k = 0; k < 27; k++) acc[k] = 0;
    const int* __restrict__ pose_edges_g = pr.pose_edges;
    const int q_end = pr.pose_start[a + 1];
    for (int q = pr.pose_start[a] + tid; q < q_end; q += kLinThreads) {
        const int e = pose_edges_g[q];
        const uint8_t lv = level_g[e];
        const int p = e_pt_g[e];
        float o0, o1, ur, inf;
        if (REC && eoi_g) { const float4 oi = eoi_g[e]; o0 = oi.x; o1 = oi.y; ur = oi.z; inf = oi.w; }   // (one 16-byte gather instead of four 4-byte ones)
        else { o0 = e_obs_g[e * 3]; o1 = e_obs_g[e * 3 + 1]; ur = e_obs_g[e * 3 + 2]; inf = e_info_g[e]; }
        const double Xw[3] = {X[p * 3], X[p * 3 + 1], X[p * 3 + 2]};
        if (lv != 0) continue;
        const bool stereo = !(ur < 0);
        const double ob[3] = {(double)o0, (double)o1, (double)ur};
        const double info = (double)inf;
        double pc[3], er[3], iz;
        map_rt(Ra, Ta.t, Xw, pc);
        const double c2 = residual_fast(cam, pc, ob, stereo, info, er, iz);
        double r0 = c2, wgt = 1.0;
        if (robust) huber_fast(c2, stereo ? dStereo : dMono, r0, wgt);
        const double wi = wgt * info;
        double Ju[6], Jv[6], Jr[6];
        pose_jac_rows(cam, pc, iz, stereo, Ju, Jv, Jr);
        accumulate_row<0>(Ju, er[0], wi, acc);
        accumulate_row<1>(Jv, er[1], wi, acc);
        if (stereo) accumulate_row<2>(Jr, er[2], wi, acc);
    }
    {
        double rs[7];
        wsum_rows<27>(acc, rs);   // (the 27 wavefront sums in 7 registers: row (b5, b4) of register q holds value 4 q + 2 b4 + b5)
        const int kq = 2 * ((lane >> 4) & 1) + (lane >> 5);
#pragma unroll
        for (int q = 0; q < 7; q++)
            if ((lane & 15) == 0 && 4 * q + kq < 27) sAcc[wv][4 * q + kq] = rs[q];
    }
    __syncthreads();
    if (tid < 27) {
        double sv = sAcc[0][tid];
        for (int i = 1; i < kLinThreads / 64; i++) sv += sAcc[i][tid];
        sAcc[0][tid] = sv;
    }
    __syncthreads();
    if (tid < 36) {
        const int i = tid / 6, j = tid - i * 6;
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        pr.Hpp[a * 36 + tid] = sAcc[0][lo * 6 - lo * (lo - 1) / 2 + (hi - lo)];
    }
    if (tid < 6) pr.bp[a * 6 + tid] = sAcc[0][21 + tid];
}

// LM control after the linearisation (k_w_ctrlA, one thread per window): chi2 of the current state, lambda init.
// The control steps are their own tiny launches: a kernel boundary makes every block's partial sums visible, whereas a "last block to
// finish" scheme needs an agent-scope release per block (an L2 write-back each): 185 us per k_w_lin launch of 28 windows x 20 blocks.
// The damping of the trial in flight.  In the pair-gather path the control step after a linearisation (w_ctrlA) has no launch of its own: k_w_edgeW computes its
// values — every workgroup for itself from the linearisation's partial sums, which no launch of the trial modifies — and workgroup 0 leaves them in
// lambdaA / FA, which k_w_edgeW itself never reads; the kernels behind it take lambdaA while need_lin still says "first trial after a linearisation", and the
// trial's k_w_ctrlB commits them (need_lin = 0, currentChi = FA, lambda = lambdaA on the first iteration of a stage) before it decides the trial.
__device__ __forceinline__ double w_lambda_eff(const LbaCtrl* ct) { return (ct->need_lin && ct->iter == 0) ? ct->lambdaA : ct->lambda; }

__device__ void w_ctrlA(const LbaProblem& pr, const LbaWide& w) {
    LbaCtrl* ct = w.ct;
    if (ct->gate) { ct->gate = 0; }
    if (!ct->need_lin) return;
    double F = 0, m = 0;
    for (int i = 0; i < w.nblk_pt; i++) { F += w.partF[i]; m = fmax(m, w.partM[i]); }
    ct->currentChi = F;
    if (ct->iter == 0) {
        for (int a = 0; a < pr.K; a++)
            if (w.blk[a] >= 0)
                for (int i = 0; i < 6; i++) m = fmax(m, fabs(pr.Hpp[a * 36 + i * 7]));
        ct->lambda = 1e-5 * m;
        ct->ni = 2;
    }
    ct->need_lin = 0;
    ct->qmax = 0;
}

// W_e = B_e (Hll_p + lambda I)^-1 for every active edge of a free keyframe: one inversion per edge instead of one per
// (block, edge) pair inside the Schur kernel
// (Writing W_e inside k_w_lin for the iterations whose lambda is known beforehand — the four lanes of a point hold its complete Hll — was measured: k_w_lin
// 107 -> 160 us, this kernel 73 -> 14 us per trial of 40 steady-state windows, bit-identical results: no gain, not kept.)
// The 144-byte block of an edge is computed by one lane, but a lane-per-block store (nine 16-byte pieces at a stride of 144 bytes across the lanes) reached
// the memory side as 2.5x the bytes (rocprofv3 WRITE_SIZE 185 MB per launch of 40 windows against 75 MB of blocks: profiles/r03_pmc_lba_traffic.json): the blocks
// of the workgroup's 256 consecutive edges are contiguous, so they go through LDS and out as 16 bytes per lane, consecutive lanes consecutive addresses.
// (Inactive edges — outliers, fixed keyframes — get zeros: nothing reads their W.)
// VINV (round 4, default): the kernel only inverts — (Hll_p + lambda I)^-1 of every landmark as 6 doubles (the cofactor inverse of a symmetric matrix is
// symmetric bit for bit) at w.W + 6 p — and k_w_schur forms W_e = B_e V^-1 for the operand it needs from the B block it fetched: the 144-byte W blocks (1.9 MB
// per steady-state window and trial written here, re-read ~8.5 times by the pair gather) no longer exist, both operands of a pair come from ONE array, and
// this launch shrinks from 145 to ~10 us per 128 windows.  Same products in the same order: the reduced system does not change by a bit.
template <bool VINV>
__global__ __launch_bounds__(256) void k_w_edgeW(const LbaProblem* probs, const LbaWide* ws, int nwin) {
    int win_, item_;
    if (!xcd_window_item(nwin, win_, item_)) return;
    const LbaProblem& pr = probs[win_];
    const LbaWide& w = ws[win_];
    const LbaCtrl* ct = w.ct;
    if (ct->done) return;
    const int e0 = item_ * 256;
    if (e0 >= (VINV ? max(pr.P, 1) : pr.E)) return;   // (VINV: 256 landmarks per workgroup; workgroup 0 always runs — it publishes the control values)
    __shared__ double sW[VINV ? 1 : 256 * 19];   // 19: one double of padding per block (18 would put the lanes of a wavefront on 16 of the 64 banks)
    __shared__ double s_lambda;
    // LM control after a linearisation (w_ctrlA's values, see w_lambda_eff): lambda of the first trial of a stage = 1e-5 x the largest diagonal entry of the
    // linearised system (computeLambdaInit), from the landmark blocks' partial maxima and the free keyframes' Hpp — by wavefront 0 of every workgroup;
    // workgroup 0 also sums the robust chi2 partials and publishes both for the later launches of the trial.
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        double lam = ct->lambda;
        if (ct->need_lin) {
            if (ct->iter == 0) {
                double m = 0;
                for (int i = lane; i < w.nblk_pt; i += 64) m = fmax(m, w.partM[i]);
                for (int i = lane; i < ct->nfree * 6; i += 64) m = fmax(m, fabs(pr.Hpp[w.free_pose[i / 6] * 36 + (i % 6) * 7]));
                lam = 1e-5 * wmax(m);
            }
            if (item_ == 0) {
                double F = 0;
                if (lane == 0) for (int i = 0; i < w.nblk_pt; i++) F += w.partF[i];   // (index order: the sum w_ctrlA forms)
                if (lane == 0) { w.ct->lambdaA = lam; w.ct->FA = F; }
            }
        }
        if (lane == 0) s_lambda = lam;
    }
    __syncthreads();
    if (VINV) {
        const int p = e0 + threadIdx.x;
        if (p >= pr.P) return;
        const double lambda = s_lambda;
        double D[9], Di[9];
        const double* H = pr.Hll + (long long)p * 9;
#pragma unroll
        for (int i = 0; i < 9; i++) D[i] = H[i];
        D[0] += lambda; D[4] += lambda; D[8] += lambda;
        inv3(D, Di);
        double2* out = (double2*)(w.W + (long long)p * 6);
        out[0] = make_double2(Di[0], Di[1]); out[1] = make_double2(Di[2], Di[4]); out[2] = make_double2(Di[5], Di[8]);
        return;
    }
    const int e = e0 + threadIdx.x;
    if (e < pr.E && pr.level[e] == 0 && w.blk[pr.e_kf[e]] >= 0) {
        const double lambda = s_lambda;
        const int p = pr.e_pt[e];
        double D[9], Di[9];
        const double* H = pr.Hll + (long long)p * 9;
#pragma unroll
        for (int i = 0; i < 9; i++) D[i] = H[i];
        D[0] += lambda; D[4] += lambda; D[8] += lambda;
        inv3(D, Di);
        const double* Ba = pr.Hpl + (long long)e * 18;
        double* We = sW + threadIdx.x * 19;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const double b0 = Ba[i * 3], b1 = Ba[i * 3 + 1], b2v = Ba[i * 3 + 2];
            We[i * 3] = b0 * Di[0] + b1 * Di[3] + b2v * Di[6];
            We[i * 3 + 1] = b0 * Di[1] + b1 * Di[4] + b2v * Di[7];
            We[i * 3 + 2] = b0 * Di[2] + b1 * Di[5] + b2v * Di[8];
        }
    } else {   // inactive edge (outlier, fixed keyframe): nothing reads its W, but it is written below — zeros, not whatever the LDS held
        double* We = sW + threadIdx.x * 19;
#pragma unroll
        for (int i = 0; i < 18; i++) We[i] = 0.0;
    }
    __syncthreads();
    const int nE = min(256, pr.E - e0);
    double2* out = (double2*)(w.W + (long long)e0 * 18);   // (e0 * 144 bytes: 16-byte aligned)
    for (int k = threadIdx.x; k < nE * 9; k += 256) {
        const int blk = k / 9, part = k - blk * 9;
        const double* src = sW + blk * 19 + part * 2;
        out[k] = make_double2(src[0], src[1]);
    }
}

// ---- Schur pair lists built on the device (once per call; the host used to spend ~430 us per steady-state window on them and upload 8 bytes per pair) ----
// Block t = (a, b), a <= b, of the reduced system lists the points both free keyframes see, ascending, as (edge in a, edge in b): exactly the order of the host
// builder in lba_build (a point has one observation per keyframe), so the sums of k_w_schur do not change by a bit.
//   k_w_pair_matrix : M[a][p] = 1 + position of the edge of point p in free keyframe a inside the point's edge range (0 = not seen); M is zeroed by the caller
//   k_w_pair_blocks : one wavefront per block walks the points 64 at a time — FILL = false counts, FILL = true writes the pairs at pair_start[t] + rank
//   k_w_pair_scan   : one wavefront per window, exclusive scan of the counts into pair_start[0 .. nblk]
__global__ __launch_bounds__(256) void k_w_pair_matrix(const LbaProblem* probs, const LbaWide* ws) {
    const LbaProblem& pr = probs[blockIdx.y];
    const LbaWide& w = ws[blockIdx.y];
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= pr.E) return;
    if (!w.pairM) return;
    const int a = w.blk[pr.e_kf[e]];
    if (a < 0) return;
    const int p = pr.e_pt[e];
    w.pairM[(size_t)a * pr.P + p] = (uint16_t)(e - pr.pt_start[p] + 1);   // keyframe-major: k_w_pair_blocks reads a keyframe's row 64 consecutive points at a time
}
template <bool FILL>
__global__ __launch_bounds__(64) void k_w_pair_blocks(const LbaProblem* probs, const LbaWide* ws) {
    const LbaProblem& pr = probs[blockIdx.y];
    const LbaWide& w = ws[blockIdx.y];
    if (!w.pairM) return;
    const int nfree = w.ct->nfree, t = blockIdx.x, lane = threadIdx.x;
    if (t >= nfree * (nfree + 1) / 2) return;
    int ba = 0, rem = t;
    while (rem >= nfree - ba) { rem -= nfree - ba; ba++; }
    const int bb = ba + rem;
    int base = FILL ? w.pair_start_w[t] : 0;
    const uint16_t* rowa = w.pairM + (size_t)ba * pr.P;
    const uint16_t* rowb = w.pairM + (size_t)bb * pr.P;
    for (int p0 = 0; p0 < pr.P; p0 += 256) {   // four chunks of 64 points per round: their eight loads travel together (the walk is a chain of load -> ballot -> count otherwise)
        int ma[4], mb[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int p = p0 + 64 * u + lane;
            ma[u] = 0; mb[u] = 0;
            if (p < pr.P) { ma[u] = rowa[p]; mb[u] = rowb[p]; }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int p = p0 + 64 * u + lane;
            const bool has = ma[u] != 0 && mb[u] != 0;
            const unsigned long long bal = __ballot(has);
            if (FILL && has) {
                const int s = pr.pt_start[p];
                w.pairs_w[base + __popcll(bal & ((1ull << lane) - 1ull))] = make_int2(s + ma[u] - 1, s + mb[u] - 1);
            }
            base += __popcll(bal);
        }
    }
    if (!FILL && lane == 0) w.pair_start_w[t + 1] = base;   // counts, shifted by one: the scan below turns them into starts in place
}
__global__ __launch_bounds__(64) void k_w_pair_scan(const LbaProblem* probs, const LbaWide* ws) {
    const LbaWide& w = ws[blockIdx.x];
    if (!w.pairM) return;
    const int nfree = w.ct->nfree, nblk = nfree * (nfree + 1) / 2, lane = threadIdx.x;
    int carry = 0;
    if (lane == 0) w.pair_start_w[0] = 0;
    for (int t0 = 0; t0 < nblk; t0 += 64) {
        const int t = t0 + lane;
        int v = t < nblk ? w.pair_start_w[t + 1] : 0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(v, d, 64); if (lane >= d) v += u; }
        if (t < nblk) w.pair_start_w[t + 1] = carry + v;
        carry += __shfl(v, 63, 64);
    }
}

// x + the value of another lane of its quad (DPP quad_perm control CTRL), for fp64 as two 32-bit moves
template <int CTRL>
__device__ __forceinline__ double dpp_quad_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_mov_dpp((int)(b & 0xffffffffll), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// One wavefront per 6x6 block (a, b) of the reduced system walks the block's pair list, 64 pairs per round, one pair per lane.
// Operand fetch (round 4): a pair's operands are two 144-byte blocks, W_a of edge pe.x and B_b of edge pe.y.  Fetched by the lane that owns the pair (nine
// 16-byte loads per block at a 144-byte stride ACROSS the lanes) every load instruction touched 64 different cache lines: rocprofv3 --pmc showed the texture
// addresser busy 70-77 % of the launch (TA_BUSY_avr 176 k of 256 k cycles, 46 cache accesses per load instruction) with the vector ALU at 22 %.  Now the
// wavefront fetches the 128 blocks of a round COOPERATIVELY — 16-byte piece g of the round's 18 KB goes to lane g mod 64, so nine consecutive lanes read one
// block and a load instruction touches ~21 lines — and hands them to their owners through LDS (half a round at a time: 9.2 KB per wavefront).
template <bool VINV>
__global__ __launch_bounds__(64) void k_w_schur(const LbaProblem* probs, const LbaWide* ws, int nwin) {
    int win, t;
    if (!xcd_window_item(nwin, win, t)) return;
    const LbaProblem& pr = probs[win];
    const LbaWide& w = ws[win];
    const LbaCtrl* ct = w.ct;
    if (ct->done) return;
    const int nfree = ct->nfree, n = ct->n, ld = n + 1;
    const int lane = threadIdx.x;
    if (t >= nfree * (nfree + 1) / 2) return;
    const double lambda = w_lambda_eff(ct);
    int ba = 0, rem = t;
    while (rem >= nfree - ba) { rem -= nfree - ba; ba++; }
    const int bb = ba + rem;
    const int a = w.free_pose[ba];
    const bool diag = ba == bb;
    double acc[36];
#pragma unroll
    for (int i = 0; i < 36; i++) acc[i] = 0;
    double bsv[6] = {0, 0, 0, 0, 0, 0};
    __shared__ __align__(16) double s_stage[64 * 18];   // 64 blocks of half a round; after the loop: the [42][17] reduction array
    __shared__ int s_idx[128];                          // edge of block b of the round (b = 2 x pair lane + operand), -1 = none
    const int q_end = w.pair_start[t + 1];
    int2 pe_next = make_int2(0, 0);
    if (w.pair_start[t] + lane < q_end) pe_next = w.pairs[w.pair_start[t] + lane];
    const double2* Wg = (const double2*)w.W;
    const double2* Bg = (const double2*)pr.Hpl;
    for (int q = w.pair_start[t] + lane; q - lane < q_end; q += 64) {   // (wave-uniform trip count: every lane takes part in the cooperative fetch)
        const bool mine = q < q_end;
        const int2 pe = pe_next;
        if (q + 64 < q_end) pe_next = w.pairs[q + 64];   // the next round's list entry travels while this round's operands do
        uint8_t la = 1, lb = 1;
        int pt = 0;
        if (mine) { la = pr.level[pe.x]; lb = pr.level[pe.y]; if (VINV) pt = pr.e_pt[pe.x]; }
        double2 vi0 = make_double2(0.0, 0.0), vi1 = vi0, vi2 = vi0;
        if (VINV && mine) { const double2* vp = (const double2*)(w.W + (long long)pt * 6); vi0 = vp[0]; vi1 = vp[1]; vi2 = vp[2]; }   // V^-1 of the pair's landmark (k_w_edgeW<true>)
        s_idx[2 * lane] = mine ? pe.x : -1;
        s_idx[2 * lane + 1] = mine ? pe.y : -1;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        double BD[18], B2[18];
#pragma unroll
        for (int half = 0; half < 2; half++) {
            double2 v[9];
#pragma unroll
            for (int i = 0; i < 9; i++) {
                const int g = 64 * i + lane, b = g / 9, piece = g - 9 * b;   // piece `piece` of block b of this half
                const int e = s_idx[64 * half + b];
                const double2* src = (VINV || (b & 1)) ? Bg : Wg;
                v[i] = e >= 0 ? src[(long long)e * 9 + piece] : make_double2(0.0, 0.0);
            }
            if (half == 1) {   // (the first half's blocks have been read by their owners)
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
#pragma unroll
            for (int i = 0; i < 9; i++) ((double2*)s_stage)[64 * i + lane] = v[i];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if ((lane >> 5) == half) {
                const double2* mb = (const double2*)s_stage + (lane & 31) * 18;   // blocks 2 (lane & 31) and + 1: W_a then B_b
#pragma unroll
                for (int i = 0; i < 9; i++) { const double2 x = mb[i]; BD[2 * i] = x.x; BD[2 * i + 1] = x.y; }
#pragma unroll
                for (int i = 0; i < 9; i++) { const double2 x = mb[9 + i]; B2[2 * i] = x.x; B2[2 * i + 1] = x.y; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();   // (the next round's fetch overwrites the staging area and the index table)
        if (la != 0 || lb != 0) continue;
        if (VINV) {   // W_a = B_a V^-1: k_w_edgeW's products, in its order
            const double d0 = vi0.x, d1 = vi0.y, d2 = vi1.x, d4 = vi1.y, d5 = vi2.x, d8 = vi2.y;   // Di[0], [1] = [3], [2] = [6], [4], [5] = [7], [8]
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const double b0 = BD[i * 3], b1 = BD[i * 3 + 1], b2v = BD[i * 3 + 2];
                BD[i * 3] = b0 * d0 + b1 * d1 + b2v * d2;
                BD[i * 3 + 1] = b0 * d1 + b1 * d4 + b2v * d5;
                BD[i * 3 + 2] = b0 * d2 + b1 * d5 + b2v * d8;
            }
        }
        if (diag) {
            const double* bl = pr.bl + (long long)(VINV ? pt : pr.e_pt[pe.x]) * 3;
            const double l0 = bl[0], l1 = bl[1], l2 = bl[2];
#pragma unroll
            for (int i = 0; i < 6; i++) bsv[i] += BD[i * 3] * l0 + BD[i * 3 + 1] * l1 + BD[i * 3 + 2] * l2;
        }
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = 0; j < 6; j++)
                acc[i * 6 + j] += BD[i * 3] * B2[j * 3] + BD[i * 3 + 1] * B2[j * 3 + 1] + BD[i * 3 + 2] * B2[j * 3 + 2];
    }
    // Sum of the 64 lanes' partial blocks: first inside every quad of lanes by two DPP quad-permute steps (full-rate cross-lane moves, no LDS), then the 16
    // quad sums of each of the 36 (+6) entries through a [42][17] LDS array (rows skewed by one double: conflict-free row reads), added in lane order by
    // lane i < 42 (the array lives in the staging area: the fetch loop is over).
    double* red = s_stage;
    static_assert(42 * 17 <= 64 * 18, "reduction array fits the staging area");
    auto quad_sum = [](double v) {
        v += dpp_quad_f64<0xB1>(v);   // quad_perm [1,0,3,2]: lane ^ 1
        v += dpp_quad_f64<0x4E>(v);   // quad_perm [2,3,0,1]: lane ^ 2
        return v;
    };
#pragma unroll
    for (int i = 0; i < 36; i++) { const double q4 = quad_sum(acc[i]); if ((lane & 3) == 0) red[i * 17 + (lane >> 2)] = q4; }
    if (diag) {
#pragma unroll
        for (int i = 0; i < 6; i++) { const double q4 = quad_sum(bsv[i]); if ((lane & 3) == 0) red[(36 + i) * 17 + (lane >> 2)] = q4; }
    }
    __syncthreads();
    double mine = 0;
    if (lane < (diag ? 42 : 36)) {
        const double* row = red + lane * 17;
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
        for (int k = 0; k < 16; k += 4) { s0 += row[k]; s1 += row[k + 1]; s2 += row[k + 2]; s3 += row[k + 3]; }
        mine = (s0 + s1) + (s2 + s3);
    }
    if (lane < 36) {   // lanes 0..35 write one entry each
        const int i = lane / 6, j = lane - i * 6;
        double v = -mine;
        if (diag) v += pr.Hpp[a * 36 + lane] + (i == j ? lambda : 0.0);
        pr.Hs[(size_t)(6 * ba + i) * ld + 6 * bb + j] = v;
    } else if (diag && lane < 42) {
        const int i = lane - 36;
        pr.Hs[(size_t)(6 * ba + i) * ld + n] = pr.bp[a * 6 + i] - mine;
    }
}

// The Schur complement from COMPACT EDGE RECORDS (round 5, the default of the pair-gather path).  The 144-byte block of an edge, B_e = Jp^T (w info) Jx, is a
// function of four doubles — the point in the edge's camera frame as (x, y, 1/z) and the edge's weight — and of the keyframe's rotation, which is the same
// for every pair of a 6x6 block (a, b).  So k_w_lin<true> stores 32 bytes per edge instead of 144, and the product of a pair is formed from the records:
//     W_a B_b^T = Jp_a^T [ (w_a Jx_a) V^-1 (w_b Jx_b)^T ] Jp_b = Jp_a^T M Jp_b,       M 3x3,
// (Jx = -d proj / d p_c * R: point_jac_rows; Jp: pose_jac_rows — the formulas k_w_lin used for B_e).  Per pair a lane fetches 64 + 48 bytes (two records and
// the landmark's V^-1) instead of two 144-byte blocks, straight into registers: the cooperative fetch through LDS of k_w_schur (two dependent hand-overs per
// round, the kernel's bound after round 4: profiles/r04_pmc_lba_wait_ta.json) is gone, and so are k_w_lin's 144-byte-per-lane stores (128 MB written per launch
// of 40 windows for 75 MB of blocks).  ~300 fused multiply-adds per pair instead of ~180: the vector ALU was at 16-22 % of the wave cycles.  An edge that has
// left the problem (level 1) carries weight 0 and adds exact zeros.  Same summation order over the pairs as k_w_schur (lane l takes pairs l, l + 64, ...; quad
// sums, then 16 partial sums in lane order), so a window's result does not depend on the batch it is solved in.
#ifndef OSLAM_SCHUR_REC_WAVES
#define OSLAM_SCHUR_REC_WAVES 2
#endif
__device__ __forceinline__ double uniform_f64(double v) {   // a wave-uniform value into scalar registers
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffll)), hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
#ifndef OSLAM_SCHUR_PRE
#define OSLAM_SCHUR_PRE 2   // rounds of a Schur block whose pair entries and point indices are requested together
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(OSLAM_SCHUR_REC_WAVES, OSLAM_SCHUR_REC_WAVES))) void k_w_schur_rec(const LbaProblem* probs, const LbaWide* ws, int nwin) {
#pragma clang fp contract(fast)
    int win, t;
    if (!xcd_window_item(nwin, win, t)) return;
    const LbaProblem& pr = probs[win];
    const LbaWide& w = ws[win];
    const LbaCtrl* ct = w.ct;
    if (ct->done) return;
    const int nfree = ct->nfree, n = ct->n, ld = n + 1;
    const int lane = threadIdx.x;
    if (t >= nfree * (nfree + 1) / 2) return;
    const double lambda = w_lambda_eff(ct);
    int ba = 0, rem = t;
    while (rem >= nfree - ba) { rem -= nfree - ba; ba++; }
    const int bb = ba + rem;
    const int a = w.free_pose[ba], b = w.free_pose[bb];
    const bool diag = ba == bb;
    const Cam cam = {(double)pr.K5[0], (double)pr.K5[1], (double)pr.K5[2], (double)pr.K5[3], (double)pr.K5[4]};
    double Ra[9], Rb[9];   // (wave-uniform: scalar loads)
    {
        const double* ra = w.R + ((size_t)ct->cur * pr.K + a) * 9;
        const double* rb = w.R + ((size_t)ct->cur * pr.K + b) * 9;
#pragma unroll
        for (int i = 0; i < 9; i++) { Ra[i] = uniform_f64(ra[i]); Rb[i] = uniform_f64(rb[i]); }
    }
    double acc[36];
#pragma unroll
    for (int i = 0; i < 36; i++) acc[i] = 0;
    double bsv[6] = {0, 0, 0, 0, 0, 0};
    __shared__ __align__(16) double red[42 * 17];
    const int q0 = w.pair_start[t], q_end = w.pair_start[t + 1];
    const double2* __restrict__ rec = (const double2*)w.rec;
    const double2* __restrict__ Vg = (const double2*)w.W;
    // (the bases the loop gathers from, read ONCE: through pr / w the compiler reloaded w.pairs, pr.e_pt and pr.bl with a scalar load + wait in every round: ISA)
    const int2* __restrict__ pairs_g = w.pairs;
    const int* __restrict__ e_pt_g = pr.e_pt;
    const double* __restrict__ bl_g = pr.bl;
    // Operands of a round are requested one round ahead of the arithmetic, in two register sets used in turn.  A pair's operands sit behind TWO dependent index
    // loads (pair entry -> edge records and the edge's point -> V_p): the index loads of kSchurPre rounds are issued up front, all pair entries first, then all
    // point indices (round 5, second pass: inside the one-round-ahead fetch each round still waited three dependent memory latencies — ISA — and a block has
    // only 2-3 rounds).
    struct Ops { double2 a0, a1, b0, b1, v0, v1, v2; int pt; };
    auto fetch = [&](Ops& o, int2 pe, int pt) {   // (32-bit offsets from wave-uniform bases)
        o.a0 = rec[(unsigned)pe.x * 2u]; o.a1 = rec[(unsigned)pe.x * 2u + 1u];
        if (!diag) { o.b0 = rec[(unsigned)pe.y * 2u]; o.b1 = rec[(unsigned)pe.y * 2u + 1u]; }
        o.pt = pt;
        o.v0 = Vg[(unsigned)pt * 3u]; o.v1 = Vg[(unsigned)pt * 3u + 1u]; o.v2 = Vg[(unsigned)pt * 3u + 2u];
    };
    auto compute = [&](const Ops& o) {
        const double2 cb0 = diag ? o.a0 : o.b0, cb1 = diag ? o.a1 : o.b1;
        const double wa = fabs(o.a1.y), wb = fabs(cb1.y);
        if (wa == 0.0 || wb == 0.0) return;
        // A monocular edge has no u_R row: its weight for that row is 0, so row r of (w Jx) — and with it row / column r of M — is zero and whatever the
        // stereo formulas put into Jx's and Jp's third rows is multiplied by zeros.
        const double wra = __builtin_signbit(o.a1.y) ? 0.0 : wa, wrb = __builtin_signbit(cb1.y) ? 0.0 : wb;
        const double pa[3] = {o.a0.x, o.a0.y, 0.0}, pb[3] = {cb0.x, cb0.y, 0.0};
        // N = (w_a Jx_a) V^-1
        double Jxa[9], N[9];
        point_jac_rows(cam, pa, o.a1.x, Ra, true, Jxa);
        {
            const double d0 = o.v0.x, d1 = o.v0.y, d2 = o.v1.x, d4 = o.v1.y, d5 = o.v2.x, d8 = o.v2.y;   // V^-1: [0], [1] = [3], [2] = [6], [4], [5] = [7], [8]
#pragma unroll
            for (int d = 0; d < 3; d++) {
                const double wd = d == 2 ? wra : wa;
                const double g0 = wd * Jxa[d * 3], g1 = wd * Jxa[d * 3 + 1], g2 = wd * Jxa[d * 3 + 2];
                N[d * 3] = g0 * d0 + g1 * d1 + g2 * d2;
                N[d * 3 + 1] = g0 * d1 + g1 * d4 + g2 * d5;
                N[d * 3 + 2] = g0 * d2 + g1 * d5 + g2 * d8;
            }
        }
        double Jua[6], Jva[6], Jra[6];
        pose_jac_rows(cam, pa, o.a1.x, true, Jua, Jva, Jra);
        if (diag) {   // rhs: W_a b_l = Jp_a^T (N b_l)
            const double* bl = bl_g + (unsigned)o.pt * 3u;
            const double l0 = bl[0], l1 = bl[1], l2 = bl[2];
            const double n0 = N[0] * l0 + N[1] * l1 + N[2] * l2, n1 = N[3] * l0 + N[4] * l1 + N[5] * l2, n2 = N[6] * l0 + N[7] * l1 + N[8] * l2;
#pragma unroll
            for (int i = 0; i < 6; i++) {   // (Ju[4] = Jv[3] = Jr[4] = 0; every term is accumulated by its own fused multiply-add)
                if (i != 4) bsv[i] = __builtin_fma(Jua[i], n0, bsv[i]);
                if (i != 3) bsv[i] = __builtin_fma(Jva[i], n1, bsv[i]);
                if (i != 4) bsv[i] = __builtin_fma(Jra[i], n2, bsv[i]);
            }
        }
        // M = N (w_b Jx_b)^T
        double Jxb[9], M[9];
        point_jac_rows(cam, pb, cb1.x, Rb, true, Jxb);
#pragma unroll
        for (int db = 0; db < 3; db++) {
            const double wd = db == 2 ? wrb : wb;
            const double g0 = wd * Jxb[db * 3], g1 = wd * Jxb[db * 3 + 1], g2 = wd * Jxb[db * 3 + 2];
#pragma unroll
            for (int da = 0; da < 3; da++) M[da * 3 + db] = N[da * 3] * g0 + N[da * 3 + 1] * g1 + N[da * 3 + 2] * g2;
        }
        double Jub[6], Jvb[6], Jrb[6];
        pose_jac_rows(cam, pb, cb1.x, true, Jub, Jvb, Jrb);
        // acc += Jp_a^T M Jp_b over the non-zero Jacobian entries
#pragma unroll
        for (int i = 0; i < 6; i++) {
            double T0, T1, T2;   // row i of Jp_a^T M
            if (i == 4) { T0 = Jva[i] * M[3]; T1 = Jva[i] * M[4]; T2 = Jva[i] * M[5]; }
            else {
                T0 = Jua[i] * M[0]; T1 = Jua[i] * M[1]; T2 = Jua[i] * M[2];
                if (i != 3) { T0 = __builtin_fma(Jva[i], M[3], T0); T1 = __builtin_fma(Jva[i], M[4], T1); T2 = __builtin_fma(Jva[i], M[5], T2); }
                T0 = __builtin_fma(Jra[i], M[6], T0); T1 = __builtin_fma(Jra[i], M[7], T1); T2 = __builtin_fma(Jra[i], M[8], T2);
            }
#pragma unroll
            for (int j = 0; j < 6; j++) {
                if (j != 4) acc[i * 6 + j] = __builtin_fma(T0, Jub[j], acc[i * 6 + j]);
                if (j != 3) acc[i * 6 + j] = __builtin_fma(T1, Jvb[j], acc[i * 6 + j]);
                if (j != 4) acc[i * 6 + j] = __builtin_fma(T2, Jrb[j], acc[i * 6 + j]);
            }
        }
    };
    {
        Ops oa, ob;
        oa.a0 = ob.a0 = oa.b0 = ob.b0 = make_double2(0.0, 0.0); oa.a1 = ob.a1 = oa.b1 = ob.b1 = make_double2(1.0, 0.0);
        oa.v0 = oa.v1 = oa.v2 = ob.v0 = ob.v1 = ob.v2 = make_double2(0.0, 0.0); oa.pt = ob.pt = 0;
        constexpr int kSchurPre = OSLAM_SCHUR_PRE;
        for (int q = q0 + lane; q < q_end; q += 64 * kSchurPre) {   // (per lane; a lane past its last pair idles)
            int2 pe[kSchurPre]; int pt[kSchurPre];
#pragma unroll
            for (int r = 0; r < kSchurPre; r++) pe[r] = pairs_g[(unsigned)min(q + 64 * r, q_end - 1)];
#pragma unroll
            for (int r = 0; r < kSchurPre; r++) pt[r] = e_pt_g[(unsigned)pe[r].x];
            fetch(oa, pe[0], pt[0]);
#pragma unroll
            for (int r = 0; r < kSchurPre; r++) {
                if (q + 64 * r >= q_end) break;
                Ops& cur = (r & 1) ? ob : oa;
                Ops& nxt = (r & 1) ? oa : ob;
                if (r + 1 < kSchurPre && q + 64 * (r + 1) < q_end) fetch(nxt, pe[r + 1], pt[r + 1]);
                compute(cur);
            }
        }
    }
    // reduction: quad sums by DPP, then the 16 quad sums of every entry in lane order through LDS
    auto quad_sum = [](double v) {
        v += dpp_quad_f64<0xB1>(v);
        v += dpp_quad_f64<0x4E>(v);
        return v;
    };
#pragma unroll
    for (int i = 0; i < 36; i++) acc[i] = quad_sum(acc[i]);
    if (diag) {
#pragma unroll
        for (int i = 0; i < 6; i++) bsv[i] = quad_sum(bsv[i]);
    }
    if ((lane & 3) == 0) {
        double* rq = red + (lane >> 2);
#pragma unroll
        for (int i = 0; i < 36; i++) rq[i * 17] = acc[i];
        if (diag) {
#pragma unroll
            for (int i = 0; i < 6; i++) rq[(36 + i) * 17] = bsv[i];
        }
    }
    __syncthreads();
    double mine = 0;
    if (lane < (diag ? 42 : 36)) {
        const double* row = red + lane * 17;
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
        for (int k = 0; k < 16; k += 4) { s0 += row[k]; s1 += row[k + 1]; s2 += row[k + 2]; s3 += row[k + 3]; }
        mine = (s0 + s1) + (s2 + s3);
    }
    if (lane < 36) {
        const int i = lane / 6, j = lane - i * 6;
        double v = -mine;
        if (diag) v += pr.Hpp[a * 36 + lane] + (i == j ? lambda : 0.0);
        pr.Hs[(size_t)(6 * ba + i) * ld + 6 * bb + j] = v;
    } else if (diag && lane < 42) {
        const int i = lane - 36;
        pr.Hs[(size_t)(6 * ba + i) * ld + n] = pr.bp[a * 6 + i] - mine;
    }
}

// Blocked Cholesky (U^T U, 6-wide panels) + back substitution of the reduced camera system, one workgroup.
// LDS_RESIDENT: the whole augmented system [n][n+1] fp64 is staged into LDS once (n <= kCholLdsN: 20 free keyframes
// = 116 KB of the CU's 160 KB), so the 2 barriers per panel and the n sequential back-substitution steps run at
// LDS latency instead of HBM latency; larger systems work in place in global memory (L2 resident).
// Pivot rows are scaled by one reciprocal per pivot (1 sqrt + 1 div on the critical path instead of 1 + 5).
constexpr int kCholLdsN = 132;   // 132*133*8 = 140 448 B

template <bool LDS_RESIDENT>
__global__ __launch_bounds__(1024) void k_w_chol(const LbaProblem* probs, const LbaWide* ws) {
    // dense solve of the reduced system: fused multiply-adds allowed here (the reference factors this matrix with a
    // different algorithm anyway, Eigen LDLT inside g2o; the rest of the library stays -ffp-contract=off)
#pragma clang fp contract(fast)
    const LbaProblem& pr = probs[blockIdx.y];
    const LbaWide& w = ws[blockIdx.y];
    LbaCtrl* ct = w.ct;
    if (ct->done) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = ct->n, ld = n + 1;
    extern __shared__ __align__(16) double s_A[];
    __shared__ int s_ok;
    __shared__ double xs[6 * kLbaMaxKF];
    __shared__ double rdiag[LDS_RESIDENT ? 1 : 6 * kLbaMaxKF];
    if (LDS_RESIDENT && n > kCholLdsN) {   // host launches <true> only when 6*nfree fits; never index past the LDS image
        if (tid == 0) ct->ok2 = 0;
        for (int i = tid; i < n; i += 1024) pr.xp[i] = 0;
        return;
    }
    // compile-time address space: ds_* for the LDS image, global_* otherwise (a runtime-selected generic pointer
    // would turn every access into a flat op that waits on both counters)
    auto A = [&](size_t idx) -> double& {
        if constexpr (LDS_RESIDENT) return s_A[idx];
        else return pr.Hs[idx];
    };
#ifdef OSLAM_LBA_PROFILE
    long long tw_ = wall_clock64();
#define CH_STAMP(i) do { if (tid == 0) { const long long t_ = wall_clock64(); pr.stats[8 + (i)] += (int)(t_ - tw_); tw_ = t_; } } while (0)
#else
#define CH_STAMP(i) do { } while (0)
#endif
    if (tid == 0) s_ok = 1;
    if (LDS_RESIDENT)
        for (int i = tid; i < n * ld; i += 1024) s_A[i] = pr.Hs[i];
    __syncthreads();
    CH_STAMP(0);
    __shared__ double s_Dg[36], s_rinv[6];
    // factor the 6x6 diagonal block at j0 (one wavefront, every lane redundantly): publishes the factor and the pivot
    // reciprocals for scale_cols
    auto factor_diag = [&](int j0) {
        double Dg[36], rinv[6];
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int k = 0; k < 6; k++) Dg[i * 6 + k] = k >= i ? A((size_t)(j0 + i) * ld + j0 + k) : 0.0;
        // validity is tested off the critical path: a non-positive or non-finite pivot poisons its own row with NaN/inf
        // and the flag turns the whole solve into "failed" (ok2 = 0) afterwards
        bool good = true;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const double d = Dg[j * 6 + j];
            good = good && (d > 0) && (d < 1.7e308);
            const double r = rsqrt(d);      // one transcendental on the critical path; pivot = d * d^-1/2
            rinv[j] = r;
            Dg[j * 6 + j] = d * r;
#pragma unroll
            for (int k = j + 1; k < 6; k++) Dg[j * 6 + k] *= r;
#pragma unroll
            for (int i = j + 1; i < 6; i++)
#pragma unroll
                for (int k = i; k < 6; k++) Dg[i * 6 + k] -= Dg[j * 6 + i] * Dg[j * 6 + k];
        }
        if (lane == 0) {   // every lane holds the same factor: one lane publishes it (independent stores, no select chain)
            if (!good) s_ok = 0;
#pragma unroll
            for (int i = 0; i < 6; i++) {
                s_rinv[i] = rinv[i];
#pragma unroll
                for (int k = i; k < 6; k++) { s_Dg[i * 6 + k] = Dg[i * 6 + k]; A((size_t)(j0 + i) * ld + j0 + k) = Dg[i * 6 + k]; }
            }
        }
    };
    // row panel of block j0: U(j0.., k) = L^-1 H(j0.., k) for the columns k > j0+5, one column per thread
    auto scale_cols = [&](int j0) {
        const int k = j0 + 6 + tid;
        if (k > n) return;
        double col[6];
#pragma unroll
        for (int i = 0; i < 6; i++) col[i] = A((size_t)(j0 + i) * ld + k);
#pragma unroll
        for (int j = 0; j < 6; j++) {
            double sv = col[j];
#pragma unroll
            for (int i = 0; i < j; i++) sv -= s_Dg[i * 6 + j] * col[i];
            col[j] = sv * s_rinv[j];
        }
#pragma unroll
        for (int i = 0; i < 6; i++) A((size_t)(j0 + i) * ld + k) = col[i];
    };
    // NR rows of the trailing matrix (i0 .. i0+NR-1, columns kmin .. kend) minus the contribution of panel j0, one
    // wavefront; the rows are independent, so their load -> fma -> store chains overlap
    auto update_rows = [&](int j0, int i0, int kmin, int kend, auto nr_tag) {
        constexpr int NR = decltype(nr_tag)::value;
        double P[NR][6];
#pragma unroll
        for (int ii = 0; ii < NR; ii++)
#pragma unroll
            for (int r = 0; r < 6; r++) P[ii][r] = A((size_t)(j0 + r) * ld + i0 + ii);
        for (int k = kmin + lane; k <= kend; k += 64) {
            double cpan[6], tv[NR];
#pragma unroll
            for (int r = 0; r < 6; r++) cpan[r] = A((size_t)(j0 + r) * ld + k);
#pragma unroll
            for (int ii = 0; ii < NR; ii++) tv[ii] = A((size_t)(i0 + ii) * ld + k);
#pragma unroll
            for (int ii = 0; ii < NR; ii++) {
                double sv = 0;
#pragma unroll
                for (int r = 0; r < 6; r++) sv += P[ii][r] * cpan[r];
                A((size_t)(i0 + ii) * ld + k) = tv[ii] - sv;
            }
        }
    };
    if (wv == 0 && n > 0) factor_diag(0);
    __syncthreads();
    scale_cols(0);
    __syncthreads();
    CH_STAMP(1);
    // look-ahead: while 15 wavefronts apply panel j0 to the trailing matrix, wavefront 0 updates only the next 6x6
    // diagonal block and factors it; the (cheap, column-parallel) scaling of the next row panel follows the barrier
    for (int j0 = 0; j0 + 6 < n; j0 += 6) {
        if (wv == 0) {
            update_rows(j0, j0 + 6, j0 + 6, j0 + 11, std::integral_constant<int, 6>());
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            factor_diag(j0 + 6);
        } else {
            for (int i0 = j0 + 6 + 3 * (wv - 1); i0 < n; i0 += 45) update_rows(j0, i0, max(i0, j0 + 12), n, std::integral_constant<int, 3>());
        }
        __syncthreads();
        scale_cols(j0 + 6);
        __syncthreads();
    }
    CH_STAMP(2);
    const bool ok2 = s_ok != 0;
    if (wv == 0 && ok2) {
        // U x = y, column oriented: x_i = y_i / U_ii, then y_k -= U_ki x_i for k < i (no reduction on the critical path)
        if (LDS_RESIDENT) {
            // y and 1/U_ii live in registers (lane l holds entries l, l+64, l+128); x_i is broadcast by readlane and the
            // next column of U is fetched while the current one is applied
            double y0 = 0, y1 = 0, y2 = 0, r0 = 0, r1 = 0, r2 = 0;
            if (lane < n) { y0 = A((size_t)lane * ld + n); r0 = 1.0 / A((size_t)lane * ld + lane); }
            if (lane + 64 < n) { y1 = A((size_t)(lane + 64) * ld + n); r1 = 1.0 / A((size_t)(lane + 64) * ld + lane + 64); }
            if (lane + 128 < n) { y2 = A((size_t)(lane + 128) * ld + n); r2 = 1.0 / A((size_t)(lane + 128) * ld + lane + 128); }
            auto bcast = [&](double v, int src) -> double {
                const unsigned long long u = (unsigned long long)__double_as_longlong(v);
                const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)u, src), hi = __builtin_amdgcn_readlane((int)(unsigned)(u >> 32), src);
                return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
            };
            // One loop per 64-entry segment so the segment is a compile-time constant (no selects between registers),
            // unconditional loads from clamped rows (no exec-masked branches), and the raw column loaded in step i is
            // masked and consumed in step i-1, a full iteration after its ds_read was issued.
            const int row0 = min(lane, n - 1), row1 = min(lane + 64, n - 1), row2 = min(lane + 128, n - 1);
            double v0 = 0, v1 = 0, v2 = 0;   // raw U(row, i) of the column about to be applied
            if (n > 0) { v0 = A((size_t)row0 * ld + n - 1); v1 = A((size_t)row1 * ld + n - 1); v2 = A((size_t)row2 * ld + n - 1); }
            auto run = [&](auto seg_tag, int ihi, int ilo) {
                constexpr int S = decltype(seg_tag)::value;
                for (int i = ihi; i >= ilo; i--) {
                    const int inext = i > 0 ? i - 1 : 0;
                    const double w0 = A((size_t)row0 * ld + inext);
                    double w1 = 0, w2 = 0;
                    if (S >= 1) w1 = A((size_t)row1 * ld + inext);
                    if (S >= 2) w2 = A((size_t)row2 * ld + inext);
                    const int src = i & 63;
                    const double xi = bcast(S == 0 ? y0 : S == 1 ? y1 : y2, src) * bcast(S == 0 ? r0 : S == 1 ? r1 : r2, src);
                    y0 -= (lane < i ? v0 : 0.0) * xi;
                    if (S >= 1) y1 -= (lane + 64 < i ? v1 : 0.0) * xi;
                    if (S >= 2) y2 -= (lane + 128 < i ? v2 : 0.0) * xi;
                    if (S == 0) y0 = lane == src ? xi : y0;
                    if (S == 1) y1 = lane == src ? xi : y1;
                    if (S == 2) y2 = lane == src ? xi : y2;
                    v0 = w0; v1 = w1; v2 = w2;
                }
            };
            if (n > 128) run(std::integral_constant<int, 2>(), n - 1, 128);
            if (n > 64) run(std::integral_constant<int, 1>(), min(n - 1, 127), 64);
            if (n > 0) run(std::integral_constant<int, 0>(), min(n - 1, 63), 0);
            if (lane < n) pr.xp[lane] = y0;
            if (lane + 64 < n) pr.xp[lane + 64] = y1;
            if (lane + 128 < n) pr.xp[lane + 128] = y2;
        } else {
            for (int i = lane; i < n; i += 64) { xs[i] = A((size_t)i * ld + n); rdiag[i] = 1.0 / A((size_t)i * ld + i); }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int i = n - 1; i >= 0; i--) {
                const double xi = xs[i] * rdiag[i];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                for (int k = lane; k < i; k += 64) xs[k] -= A((size_t)k * ld + i) * xi;
                if (lane == 0) xs[i] = xi;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            for (int i = lane; i < n; i += 64) pr.xp[i] = xs[i];
        }
    }
    CH_STAMP(3);
    if (!ok2) for (int i = tid; i < n; i += 1024) pr.xp[i] = 0;
    if (tid == 0) ct->ok2 = ok2 ? 1 : 0;
}

// Reduced camera system by MATRIX CORES: right-looking blocked Cholesky (U^T U, 16-wide panels) of the augmented system [n][n+1] in place in
// global memory (L2 resident: 463 KB at n = 240), one 512-thread workgroup per window.  Per panel: wavefront 0 factors the 16x16 diagonal
// block in LDS, one thread per column solves the row panel U12 = U11^-T A12 (kept in LDS, 16 x (n+1-j0) doubles), then the 8 wavefronts
// apply the rank-16 trailing update A22 -= U12^T U12 tile by tile with v_mfma_f64_16x16x4_f64: four MFMAs per 16x16 tile, the A operand
// (lane l: U12[4s + (l>>4)][i0 + (l&15)]) and the B operand (U12[4s + (l>>4)][k0 + (l&15)]) straight from the LDS row panel, C/D
// (col = l&15, row = (l>>4) + 4 reg) read-modify-written in place.  The right-hand side rides along as column n.  Then a blocked back
// substitution.  Used for systems that do not fit the LDS-resident kernel (6 nfree > kCholLdsN); OSLAM_LBA_CHOL_MFMA=1 forces it for all sizes.
constexpr int kMB = 16;                       // panel width = MFMA tile edge
constexpr int kMfmaMaxN = 6 * kLbaMaxKF;      // 768
typedef double v4f64 __attribute__((ext_vector_type(4)));

// 1 / sqrt(d) for the pivots of the 16x16 diagonal blocks: v_rsq_f64 (the hardware's ~26-bit estimate) + two Newton steps, a dependent chain of ~9
// instructions (16 of them sit on a panel's critical path).  Full double precision to an ulp or two; a non-positive or non-finite pivot is caught by the
// caller's test of d itself.
__device__ __forceinline__ double rsqrt_nr(double d) {
    double r = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
    r = r * (1.5 - h * r * r);
    r = r * (1.5 - h * r * r);
    return r;
}

// value of lane `src` (compile-time constant) of a double, as a wave-uniform scalar
__device__ __forceinline__ double rl_f64(double v, int src) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)u, src), hi = __builtin_amdgcn_readlane((int)(unsigned)(u >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Cholesky factor AND inverse of a 16x16 diagonal block in ONE pass of one wavefront (round 5).  The wavefront holds the 16 x 32 matrix [A | I]: lane
// 32 h + c keeps column c (c < 16: A, c >= 16: the identity) of the rows of parity h, row 2 s + h in R[s].  The right-looking factorisation A = U^T U applies
// to every column the row operations that turn A into U, so the identity half ends as U^-T: after the 16 steps R holds [U | U^-T] and V = U^-1 = (U^-T)^T —
// no second substitution pass.  Step j: the pivot comes from one lane (readlane: a scalar); row j reaches the other parity's lanes and the entries
// U(j, i) reach the lanes of the rows i below by ds_bpermute — all of a step's permutes read the one register that holds row j and are issued together, so a
// step costs one crossbar round trip plus 8 multiply / multiply-add pairs, with every lane busy.  The round-4 form (lane k = column k in 16 lanes, 120 + 120
// readlane pairs in two dependent chains) took 5.2 us per block and was the critical path of a panel.
// Rows and columns >= nb are padded with the identity by the caller.  Returns (wave-uniform) whether every pivot was positive and finite.
__device__ __forceinline__ double bperm_f64(int byte_addr, double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, (int)(b & 0xffffffffll)), hi = __builtin_amdgcn_ds_bpermute(byte_addr, (int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double rcp_nr2(double z) {
    double r = __builtin_amdgcn_rcp(z);
    r = __builtin_fma(__builtin_fma(-z, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-z, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ bool chol16_aug(double (&R)[8], int lane) {
#pragma clang fp contract(fast)
    // Software-pipelined over the 16 steps (round 5, second pass; the ISA of the straight form showed a step's eight row updates, its reciprocal chain AND the
    // reciprocal-square-root chain of the final row scaling all in front of the next step's permutes: 460 cycles per step, tools/chol_lds_phase_prof.py).  Step j
    // needs row j only, and row j is final as soon as step j - 1 has updated the ONE register that holds it: that update goes first, the next step's permutes and
    // pivot read are issued right behind it, and the other seven updates and the scaling of row j - 1 run while they are in flight.
    const int h = lane >> 5, c = lane & 31;
    bool good = true;
    double rowc = 0, dc = 0, uc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // (The lane exchanges stay on ds_bpermute.  A form that keeps them in the vector ALU — v_permlane32_swap for the row, v_permlane16_swap + DPP row_newbcast for the
    // U(j, i) — was built and measured: 25.4 -> 27.9 us of factor time per launch.  One wavefront issues an fp64 instruction every ~10 cycles (tools/mfma_f64_rate.py:
    // 31.6 of 78.6 TFLOP/s at one wavefront per SIMD), the factor is bound by exactly that, and the LDS crossbar works beside the vector ALU, not instead of it.)
    auto request = [&](int j, double& row, double (&u)[8], double& d) __attribute__((always_inline)) {
        const int sj = j >> 1, hj = j & 1;
        row = bperm_f64((hj * 32 + c) * 4, R[sj]);                                 // row j at my column
#pragma unroll
        for (int sI = 0; sI < 8; sI++)
            if (2 * sI + 1 > j) u[sI] = bperm_f64((hj * 32 + 2 * sI + h) * 4, R[sj]);   // U(j, i) for my rows i = 2 s + h > j
        d = rl_f64(R[sj], hj * 32 + j);
    };
    // (rows is the pivot row times -1 / d: one multiply per step instead of one per updated row)
    auto update = [&](int j, int sI, double rows) __attribute__((always_inline)) {
        if (2 * sI > j) R[sI] = __builtin_fma(uc[sI], rows, R[sI]);                 // both parities below row j
        else if (h == 1) R[sI] = __builtin_fma(uc[sI], rows, R[sI]);                // 2 s == j: only row 2 s + 1
    };
    request(0, rowc, uc, dc);
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const int sj = j >> 1, hj = j & 1;
        good = good && __builtin_amdgcn_class(dc, 0x180);   // positive and finite (normal or subnormal): one compare
        // the step's fp64 instructions are what bounds the factor (one wavefront): 1 / sqrt(d) by rsq + two Newton steps, 1 / d as its square (no second chain),
        // the pivot row scaled once
        const double r = rsqrt_nr(dc);
        const double rd = -(r * r) * rowc;
        const int sn = (j + 1) >> 1;   // the register of row j + 1
        double rown = 0, dn = 0, un[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (j + 1 < 16) {
            update(j, sn, rd);
            __builtin_amdgcn_sched_barrier(0);
            request(j + 1, rown, un, dn);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int sI = 0; sI < 8; sI++) {
            if (2 * sI + 1 < j + 1 || (j + 1 < 16 && sI == sn)) continue;   // rows <= j; the register done above
            update(j, sI, rd);
        }
        if (h == hj) R[sj] = rowc * r;   // row j is final: U(j, .) | U^-T(j, .)
        if (j + 1 < 16) {
            rowc = rown; dc = dn;
#pragma unroll
            for (int sI = 0; sI < 8; sI++) uc[sI] = un[sI];
        }
    }
    return good;
}

constexpr int kMfmaThreads = 512;   // 8 wavefronts: 256 registers each (the four-tile trailing step needs ~150; 1024 threads would cap them at 128 and spill)
// Body of the matrix-core solve for one workgroup of kMfmaThreads threads: A = augmented system [n][n + 1] in global memory (upper triangle + rhs column),
// s_P = LDS row panel of 16 x pw doubles (pw = ((n + 1 + 15) / 16 + 1) * 16), s_x = LDS vector of n + 16 doubles; the solution goes to x (global or LDS).
// Returns (to every thread) whether all pivots were positive.
__device__ __forceinline__ bool chol_mfma_dev(double* A, int n, double* x, double* s_P, double* s_x) {
#pragma clang fp contract(fast)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ld = n + 1;
    __shared__ double s_D[kMB][kMB + 1];
    __shared__ int s_ok;
    if (tid == 0) s_ok = 1;
    const int pw = ((n + 1 + kMB - 1) / kMB + 1) * kMB;   // pitch of the row panel in LDS
    __syncthreads();
    for (int j0 = 0; j0 < n; j0 += kMB) {
        const int nb = min(kMB, n - j0), c0 = j0 + nb, m = n - c0;   // trailing rows / columns m, plus the rhs column
        // ---- 1. diagonal block, wavefront 0, in registers (round 4; the same scheme as k_w_chol_lds_mfma): lane k (mod 16) holds COLUMN k of the block padded
        // to 16x16 with the identity; the pivot and the entries of row j reach the other lanes by readlane, no LDS round trip inside the 16 steps (the LDS form
        // took ~8 us per block).  Then V = U11^-1 by back substitution on the identity, column k in lane k, into s_D for the row-panel product. ----
        if (wv == 0) {
            const int hh = lane >> 5, cc = lane & 31;
            double R[8];
#pragma unroll
            for (int sI = 0; sI < 8; sI++) {
                const int i = 2 * sI + hh;
                if (cc < kMB) R[sI] = (i < nb && cc < nb && cc >= i) ? A[(size_t)(j0 + i) * ld + j0 + cc] : (i == cc ? 1.0 : 0.0);
                else R[sI] = (cc - kMB == i) ? 1.0 : 0.0;
            }
            const bool good = chol16_aug(R, lane);
            if (!good && lane == 0) s_ok = 0;
#pragma unroll
            for (int sI = 0; sI < 8; sI++) {
                const int i = 2 * sI + hh;
                if (cc < kMB) { if (i < nb && cc < nb && cc >= i) A[(size_t)(j0 + i) * ld + j0 + cc] = R[sI]; }   // the factor back into the matrix (back substitution reads it)
                else s_D[cc - kMB][i] = R[sI];                                                                      // V = U11^-1: V[k][i] = U^-T[i][k]
            }
        }
        __syncthreads();
        // ---- 2. row panel U12 = U11^-T A12 = V^T A12 by MFMA, one wavefront per 16-column tile: the result goes to the matrix AND to the LDS row panel the
        // trailing update reads its operands from (rows >= nb and the padding columns of the panel are zero) ----
        {
            const int Tp = pw / kMB;
            for (int tt = wv; tt < Tp; tt += kMfmaThreads / 64) {
                const int c = kMB * tt + (lane & 15), cg = c0 + c;
                const bool live = c <= m;
                v4f64 acc = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int sx = 0; sx < kMB / 4; sx++) {
                    const int kk = 4 * sx + (lane >> 4);
                    const double aop = s_D[kk][lane & 15];                                            // A[i][k] = V[k][i]
                    const double bop = (live && kk < nb) ? A[(size_t)(j0 + kk) * ld + cg] : 0.0;   // B[k][j] = A12[k][c]
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, acc, 0, 0, 0);
                }
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int r = (lane >> 4) + 4 * g;
                    const bool rv = live && r < nb;
                    s_P[r * pw + c] = rv ? acc[g] : 0.0;
                    if (rv) A[(size_t)(j0 + r) * ld + cg] = acc[g];
                }
            }
        }
        __syncthreads();
        // ---- 3. trailing update by MFMA: tiles (ti <= tk) of the m x (m+1) trailing block ----
        // A wavefront takes FOUR tiles at a time: their 16 accumulator loads (L2 latency each) are in flight while the four independent
        // MFMA chains run, instead of one load -> MFMA -> store round trip per tile.
        if (m > 0) {
            const int Tm = (m + kMB - 1) / kMB, Tc = (m + 1 + kMB - 1) / kMB;
            const int ntile = Tm * Tc - Tm * (Tm - 1) / 2;   // tiles with ti <= tk (Tc >= Tm)
            auto unrank = [&](int t, int& ti, int& tk) {      // row-major over the rows of the upper block triangle: row ti has Tc - ti tiles
                ti = 0;
                while (t >= Tc - ti) { t -= Tc - ti; ti++; }
                tk = ti + t;
            };
            for (int t0 = wv * 4; t0 < ntile; t0 += (kMfmaThreads / 64) * 4) {
                int ti[4], tk[4];
                double* cptr[4][4];
                double cval[4][4];
                bool live[4][4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int t = min(t0 + u, ntile - 1);
                    unrank(t, ti[u], tk[u]);
                    const int k = kMB * tk[u] + (lane & 15);
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const int i = kMB * ti[u] + (lane >> 4) + 4 * g;
                        live[u][g] = (t0 + u < ntile) && i < m && k <= m;
                        cptr[u][g] = A + (size_t)(c0 + min(i, m - 1)) * ld + c0 + min(k, m);
                        cval[u][g] = *cptr[u][g];
                    }
                }
                v4f64 acc[4];
#pragma unroll
                for (int u = 0; u < 4; u++) acc[u] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int sx = 0; sx < kMB / 4; sx++) {
                    const int r = 4 * sx + (lane >> 4);
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const double a = s_P[r * pw + kMB * ti[u] + (lane & 15)];
                        const double bb = s_P[r * pw + kMB * tk[u] + (lane & 15)];
                        acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc[u], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int g = 0; g < 4; g++)
                        if (live[u][g]) *cptr[u][g] = cval[u][g] - acc[u][g];
            }
        }
        __syncthreads();
    }
    const bool ok2 = s_ok != 0;
    // ---- blocked back substitution U x = y (y = column n) ----
    if (ok2) {
        for (int i = tid; i < n; i += kMfmaThreads) s_x[i] = A[(size_t)i * ld + n];
        __syncthreads();
        const int nblk = (n + kMB - 1) / kMB;
        for (int jb = nblk - 1; jb >= 0; jb--) {
            const int j0 = jb * kMB, nb = min(kMB, n - j0);
            if (tid < kMB * kMB) {
                const int i = tid >> 4, k = tid & 15;
                s_D[i][k] = (i < nb && k < nb && k >= i) ? A[(size_t)(j0 + i) * ld + j0 + k] : (i == k ? 1.0 : 0.0);
            }
            __syncthreads();
            if (wv == 0) {   // 16x16 upper-triangular solve: lane l < nb holds y_l
                double y = lane < nb ? s_x[j0 + lane] : 0.0;
                const double rd = lane < nb ? 1.0 / s_D[lane][lane] : 0.0;
                for (int i = nb - 1; i >= 0; i--) {
                    const double xi = __shfl(y * rd, i, 64);
                    if (lane < i) y -= s_D[lane][i] * xi;
                    if (lane == i) y = xi;
                }
                if (lane < nb) s_x[j0 + lane] = y;
            }
            __syncthreads();
            for (int r = tid; r < j0; r += kMfmaThreads) {   // rows above the block: y_r -= U[r][j0 .. j0+nb) x_blk
                double sv = 0;
                for (int c = 0; c < nb; c++) sv += A[(size_t)r * ld + j0 + c] * s_x[j0 + c];
                s_x[r] -= sv;
            }
            __syncthreads();
        }
        for (int i = tid; i < n; i += kMfmaThreads) x[i] = s_x[i];
    } else {
        for (int i = tid; i < n; i += kMfmaThreads) x[i] = 0;
    }
    __syncthreads();
    return ok2;
}

__global__ __launch_bounds__(kMfmaThreads) void k_w_chol_mfma(const LbaProblem* probs, const LbaWide* ws, int min_n) {
    const LbaProblem& pr = probs[blockIdx.y];
    const LbaWide& w = ws[blockIdx.y];
    LbaCtrl* ct = w.ct;
    if (ct->done || ct->n < min_n) return;   // (min_n: systems below it were factored by the LDS-resident kernel of the same trial)
    extern __shared__ __align__(16) double s_P[];          // [16][pw] row panel, pw = panel pitch (columns right of the block, incl. rhs, padded to 16)
    __shared__ double s_x[kMfmaMaxN + kMB];
    const bool ok2 = chol_mfma_dev(pr.Hs, ct->n, pr.xp, s_P, s_x);
    if (threadIdx.x == 0) ct->ok2 = ok2 ? 1 : 0;
}

// Landmark back-substitution, trial state, computeScale partials AND the trial's chi2 (round 4: one launch instead of k_w_update + k_w_eval).
// Blocks [0, nblk_pt): 128 landmarks each — x_l, the trial position, then the errors of the landmark's edges at the TRIAL state.  The trial poses of the free
// keyframes are what the pose block (item == nblk_pt) writes to w.T / w.R for the later launches; a landmark block cannot wait for another block, so it
// recomputes them itself into LDS (se3_exp + se3_mul per free keyframe: the same two calls on the same inputs, hence the same bits).
__device__ void w_gate_block(const LbaProblem& pr, const LbaWide& w);
// FOLD: the LM control step of the trial (k_w_ctrlB of rounds 1-4) runs in the LAST workgroup of the window to finish — every workgroup publishes its partial sums,
// then takes a ticket; the one that draws the last ticket sees all of them (release / acquire fences at device scope around the ticket) — one launch less per trial,
// and slower than the launch it replaces (see fold_ctrl at the launch site): an A/B knob, off by default.
__device__ __forceinline__ void w_update_tail(const LbaProblem& pr, const LbaWide& w, int win) {
    __shared__ int s_last, s_gate;
    __threadfence();        // every lane's chi2 / trial-state stores are out before the ticket is taken
    __syncthreads();
    if (threadIdx.x == 0) {
        LbaCtrl* ct = w.ct;
        const int t = atomicAdd(&ct->ticket[1], 1);
        s_last = t == w.nblk_pt;
        s_gate = 0;
        if (s_last) {
            ct->ticket[1] = 0;
            __threadfence();
            w_ctrlB(pr, w, win);
            s_gate = ct->gate && !ct->done;
        }
    }
    __syncthreads();
    if (s_last && s_gate) {
        __threadfence();
        w_gate_block(pr, w);
        __syncthreads();
        if (threadIdx.x == 0) w.ct->gate = 0;
    }
}

// (round 5: kUpdLanes = 1, 2 or 4 lanes per landmark split its edge list — stride kUpdLanes, partial sums combined by a fixed xor-shuffle tree — like k_w_lin's landmark
// role: the one-lane form walked ~8 edges twice in one dependent chain and the kernel spent its time waiting)
template <bool REC, bool FOLD, int kUpdLanes>
__global__ __launch_bounds__(kWPt * kUpdLanes) void k_w_update(const LbaProblem* probs, const LbaWide* ws, int nwin) {
    constexpr int kUpdThreads = kWPt * kUpdLanes;   // the workgroup still owns kWPt landmarks
    int win_, item_;
    if (!xcd_window_item(nwin, win_, item_)) return;
    const LbaProblem& pr = probs[win_];
    const LbaWide& w = ws[win_];
    const LbaCtrl* ct = w.ct;
    if (ct->done) return;
    const double lambda = w_lambda_eff(ct);
    const bool ok2 = ct->ok2 != 0;
    const int cur = ct->cur;
    const double* X = w_X(pr, cur);
    double* Xn = w_X(pr, cur ^ 1);
    double sc = 0;
    if (item_ > w.nblk_pt) return;   // padding block of a batched launch (the grid is sized for the largest window)
    __shared__ double sS[kUpdThreads / 64], sF[kUpdThreads / 64];
    __shared__ SE3 sTn[kLbaMaxKF];   // trial poses of the FREE keyframes (block index order)
    if (item_ == w.nblk_pt) {   // poses
        for (int a = threadIdx.x; a < pr.K; a += kUpdThreads) {
            const int ba = w.blk[a];
            SE3* Tn = w.T + (cur ^ 1) * pr.K + a;
            const SE3 Tc = w.T[cur * pr.K + a];
            if (ba < 0) { *Tn = Tc; }
            else {
                double xa[6];
                for (int i = 0; i < 6; i++) { xa[i] = pr.xp[6 * ba + i]; sc += xa[i] * (lambda * xa[i] + pr.bp[a * 6 + i]); }
                *Tn = se3_mul(se3_exp(xa), Tc);
            }
            se3_R(*Tn, w.R + ((size_t)(cur ^ 1) * pr.K + a) * 9);
        }
        const double s1 = wsum(sc);
        if ((threadIdx.x & 63) == 0) sS[threadIdx.x >> 6] = s1;
        __syncthreads();
        if (threadIdx.x == 0) {
            double a = sS[0];
            for (int i = 1; i < kUpdThreads / 64; i++) a += sS[i];
            w.partS[item_] = a;
        }
        if (FOLD) w_update_tail(pr, w, win_);
        return;
    }
    // ---- landmark block ----
    const Cam cam_ = {(double)pr.K5[0], (double)pr.K5[1], (double)pr.K5[2], (double)pr.K5[3], (double)pr.K5[4]};
    const int nfree = ct->nfree;
    for (int ba = threadIdx.x; ba < nfree; ba += kUpdThreads) {
        const int a = w.free_pose[ba];
        double xa[6];
        for (int i = 0; i < 6; i++) xa[i] = pr.xp[6 * ba + i];
        sTn[ba] = se3_mul(se3_exp(xa), w.T[cur * pr.K + a]);
    }
    const Cam cam = {(double)pr.K5[0], (double)pr.K5[1], (double)pr.K5[2], (double)pr.K5[3], (double)pr.K5[4]};
    const double dMono = (double)pr.delta_mono, dStereo = (double)pr.delta_stereo;
    const bool robust = ct->robust != 0;
    const SE3* Tc = w.T + cur * pr.K;   // (a fixed keyframe's trial pose is its current pose)
    const int p = item_ * kWPt + (int)(threadIdx.x / kUpdLanes), sub = (int)(threadIdx.x % kUpdLanes);
    double F = 0;
    double Xt[3] = {0, 0, 0};
    if (p < pr.P) {
        double cl[3] = {pr.bl[p * 3], pr.bl[p * 3 + 1], pr.bl[p * 3 + 2]};
        double xo[3] = {0, 0, 0};
        if (ok2 && REC) {
            // B_e^T x_a from the edge's compact record (k_w_schur_rec): (w Jx)^T (Jp x_a)
#pragma clang fp contract(fast)
            const double* Rc = w.R + (size_t)cur * pr.K * 9;
            const int e1 = pr.pt_start[p + 1];
            int e = pr.pt_start[p] + sub;
            double2 r0n = make_double2(0.0, 0.0), r1n = r0n; int a_n = 0;
            if (e < e1) { const double2* rc = (const double2*)(w.rec + (long long)e * 4); r0n = rc[0]; r1n = rc[1]; a_n = pr.e_kf[e]; }
            double dl[3] = {0, 0, 0};
            for (; e < e1; e += kUpdLanes) {
                const double2 r0 = r0n, r1 = r1n; const int a = a_n;
                if (e + kUpdLanes < e1) { const double2* rc = (const double2*)(w.rec + (long long)(e + kUpdLanes) * 4); r0n = rc[0]; r1n = rc[1]; a_n = pr.e_kf[e + kUpdLanes]; }
                const int ba = w.blk[a];
                const double wi = fabs(r1.y);
                if (wi == 0.0 || ba < 0) continue;
                const bool stereo = !__builtin_signbit(r1.y);
                const double pcv[3] = {r0.x, r0.y, 0.0};
                const double* xa = pr.xp + 6 * ba;
                double Ju[6], Jv[6], Jr[6], Jx[9];
                pose_jac_rows(cam_, pcv, r1.x, stereo, Ju, Jv, Jr);
                point_jac_rows(cam_, pcv, r1.x, Rc + a * 9, stereo, Jx);
                double v0 = 0, v1 = 0, v2 = 0;
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    const double x = xa[i];
                    if (i != 4) v0 += Ju[i] * x;
                    if (i != 3) v1 += Jv[i] * x;
                    if (i != 4) v2 += Jr[i] * x;
                }
                v0 *= wi; v1 *= wi; v2 *= wi;
#pragma unroll
                for (int j = 0; j < 3; j++) dl[j] += Jx[j] * v0 + Jx[3 + j] * v1 + Jx[6 + j] * v2;
            }
#pragma unroll
            for (int j = 0; j < 3; j++) {
#pragma unroll
                for (int d = 1; d < kUpdLanes; d <<= 1) dl[j] += d == 1 ? lane_xor1(dl[j]) : (d == 2 ? lane_xor2(dl[j]) : __shfl_xor(dl[j], d, 64));
                cl[j] -= dl[j];
            }
        } else if (ok2) {
            double dl[3] = {0, 0, 0};
            for (int e = pr.pt_start[p] + sub; e < pr.pt_start[p + 1]; e += kUpdLanes) {
                // (B_e is requested before the level byte and the keyframe's block index are known: its address depends on neither)
                const double* Bg = pr.Hpl + (long long)e * 18;
                double B[18];
#pragma unroll
                for (int i = 0; i < 18; i++) B[i] = Bg[i];
                const uint8_t lv = pr.level[e];
                const int ba = w.blk[pr.e_kf[e]];
                if (lv != 0 || ba < 0) continue;
                const double* xa = pr.xp + 6 * ba;
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    double sv = 0;
#pragma unroll
                    for (int i = 0; i < 6; i++) sv += B[i * 3 + j] * xa[i];
                    dl[j] += sv;
                }
            }
#pragma unroll
            for (int j = 0; j < 3; j++) {
#pragma unroll
                for (int d = 1; d < kUpdLanes; d <<= 1) dl[j] += d == 1 ? lane_xor1(dl[j]) : (d == 2 ? lane_xor2(dl[j]) : __shfl_xor(dl[j], d, 64));
                cl[j] -= dl[j];
            }
        }
        if (ok2 && REC) {   // (Hll_p + lambda I)^-1 as k_w_edgeW<true> left it for this trial
            const double2* vp = (const double2*)(w.W + (long long)p * 6);
            const double2 v0 = vp[0], v1 = vp[1], v2 = vp[2];
            xo[0] = v0.x * cl[0] + v0.y * cl[1] + v1.x * cl[2];
            xo[1] = v0.y * cl[0] + v1.y * cl[1] + v2.x * cl[2];
            xo[2] = v1.x * cl[0] + v2.x * cl[1] + v2.y * cl[2];
        } else if (ok2) {
            double D[9], Di[9];
            const double* H = pr.Hll + (long long)p * 9;
#pragma unroll
            for (int i = 0; i < 9; i++) D[i] = H[i];
            D[0] += lambda; D[4] += lambda; D[8] += lambda;
            inv3(D, Di);
#pragma unroll
            for (int i = 0; i < 3; i++) xo[i] = Di[i * 3] * cl[0] + Di[i * 3 + 1] * cl[1] + Di[i * 3 + 2] * cl[2];
        }
#pragma unroll
        for (int i = 0; i < 3; i++) {
            Xt[i] = X[p * 3 + i] + xo[i];
            if (sub == 0) {   // (the lanes of a landmark hold the same step)
                Xn[p * 3 + i] = Xt[i];
                sc += xo[i] * (lambda * xo[i] + pr.bl[p * 3 + i]);
            }
        }
    }
    __syncthreads();   // the trial poses are in LDS
    if (p < pr.P) {
        // (an edge's inputs are loaded one round ahead and before its level byte is tested: the loop is a chain of dependent loads per edge otherwise)
        const int e1 = pr.pt_start[p + 1];
        int e = pr.pt_start[p] + sub;
        uint8_t lv_n = 1; int a_n = 0; float o0_n = 0, o1_n = 0, o2_n = 0, inf_n = 0;
        if (e < e1) { lv_n = pr.level[e]; a_n = pr.e_kf[e]; o0_n = pr.e_obs[e * 3]; o1_n = pr.e_obs[e * 3 + 1]; o2_n = pr.e_obs[e * 3 + 2]; inf_n = pr.e_info[e]; }
        for (; e < e1; e += kUpdLanes) {
            const uint8_t lv = lv_n; const int a = a_n; const float o0 = o0_n, o1 = o1_n, ur = o2_n, inf = inf_n;
            if (e + kUpdLanes < e1) { const int en = e + kUpdLanes; lv_n = pr.level[en]; a_n = pr.e_kf[en]; o0_n = pr.e_obs[en * 3]; o1_n = pr.e_obs[en * 3 + 1]; o2_n = pr.e_obs[en * 3 + 2]; inf_n = pr.e_info[en]; }
            const int ba = w.blk[a];
            const SE3 Ta = ba >= 0 ? sTn[ba] : Tc[a];
            if (lv != 0) continue;
            const bool stereo = !(ur < 0);
            const double ob[3] = {(double)o0, (double)o1, (double)ur};
            double pc[3], er[3];
            se3_map(Ta, Xt, pc);
            const double c2 = edge_error(cam, pc, ob, stereo, (double)inf, er);
            pr.chi2[e] = c2;
            if (robust) { double r0, r1; huber(c2, stereo ? dStereo : dMono, r0, r1); F += r0; }
            else F += c2;
        }
    }
    const double s1 = wsum(sc), f = wsum(F);
    if ((threadIdx.x & 63) == 0) { sS[threadIdx.x >> 6] = s1; sF[threadIdx.x >> 6] = f; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = sS[0], b = sF[0];
        for (int i = 1; i < kUpdThreads / 64; i++) { a += sS[i]; b += sF[i]; }
        w.partS[item_] = a;
        w.partF[item_] = b;
    }
    if (FOLD) w_update_tail(pr, w, win_);
}

// Levenberg-Marquardt decision + schedule transitions (g2o OptimizationAlgorithmLevenberg::solve tail,
// SparseOptimizer::optimize loop, and the stage logic of reference src/Optimizer.cc:660-707)
// LM control after the trial evaluation (one thread: the last block of k_w_eval): gain ratio, accept / reject, lambda
// update, iteration / stage bookkeeping of g2o's OptimizationAlgorithmLevenberg + SparseOptimizer::optimize
// Test hook (oslam_lba_trace): LM trials of window 0 of the wide layout. [0] = records seen, then [cap][6] = (F before, F of the trial, rho, lambda,
// accepted, first trial of a stage).
__device__ double* g_lba_trace = nullptr;
__device__ int g_lba_trace_cap = 0;

// (ct: the window's control block or a register copy of it; pF / pS: the trial's partial sums, in global memory or staged)
__device__ __forceinline__ void w_ctrlB_impl(const LbaProblem& pr, const LbaWide& w, int win, LbaCtrl* ct, const double* pF, const double* pS) {
    if (ct->need_lin) {   // pair-gather path: w_ctrlA's step was left to k_w_edgeW (values) and to this commit (state); the tiles path ran k_w_ctrlA: need_lin is 0
        ct->currentChi = ct->FA;
        if (ct->iter == 0) { ct->lambda = ct->lambdaA; ct->ni = 2; }
        ct->need_lin = 0;
        ct->qmax = 0;
    }
    double F1 = 0, sc = 0;
    for (int i = 0; i < w.nblk_pt; i++) { F1 += pF[i]; sc += pS[i]; }
    sc += pS[w.nblk_pt];
    double tempChi = F1;
    if (!ct->ok2) tempChi = 1.7976931348623157e308;
    const double rho = (ct->currentChi - tempChi) / (sc + 1e-3);
    const bool finite = (tempChi - tempChi) == 0;
    if (g_lba_trace && win == 0) {
        const int r = (int)g_lba_trace[0];
        if (r < g_lba_trace_cap) {
            double* t = g_lba_trace + 1 + 6 * r;
            t[0] = ct->currentChi; t[1] = tempChi; t[2] = rho; t[3] = ct->lambda; t[4] = (rho > 0 && finite) ? 1.0 : 0.0; t[5] = (ct->iter == 0 && ct->qmax == 0) ? 1.0 : 0.0;
        }
        g_lba_trace[0] = r + 1;
    }
    if (rho > 0 && finite) {
        double alpha = 1. - (2 * rho - 1) * (2 * rho - 1) * (2 * rho - 1);
        alpha = fmin(alpha, 2. / 3.);
        ct->lambda *= fmax(1. / 3., alpha);
        ct->ni = 2;
        ct->currentChi = tempChi;
        ct->cur ^= 1;
    } else {
        ct->lambda *= ct->ni;
        ct->ni *= 2;
    }
    ct->rho = rho;
    ct->qmax++;
    if (ct->stage == 0) ct->trials[0]++; else ct->trials[1]++;   // (constant indices: a register copy of the block stays in registers)
    const bool stop = pr.stop ? (*pr.stop != 0) : false;
    if (rho < 0 && ct->qmax < 10 && !stop) return;   // another trial with the new lambda
    // iteration finished
    if (ct->stage == 0) ct->its[0]++; else ct->its[1]++;
    if (ct->qmax == 10 || rho == 0) ct->ok = 0;
    ct->iter++;
    const int iters = ct->stage == 0 ? pr.iters0 : pr.iters1;
    const bool stop2 = pr.stop ? (*pr.stop != 0) : false;
    if (ct->iter < iters && !stop2 && ct->ok) { ct->need_lin = 1; return; }
    // stage finished
    if (ct->stage == 0 && pr.nstages > 1) {
        const bool stop3 = pr.stop ? (*pr.stop != 0) : false;
        if (stop3) { ct->done = 1; return; }   // bDoMore = false (:664-666)
        ct->stage = 1; ct->iter = 0; ct->ok = 1; ct->robust = 0; ct->need_lin = 1; ct->gate = 1;
    } else {
        ct->done = 1;
    }
}

__device__ void w_ctrlB(const LbaProblem& pr, const LbaWide& w, int win) { w_ctrlB_impl(pr, w, win, w.ct, w.partF, w.partS); }

__global__ __launch_bounds__(64) void k_w_ctrlA(const LbaProblem* probs, const LbaWide* ws) {
    const LbaProblem& pr = probs[blockIdx.y];
    const LbaWide& w = ws[blockIdx.y];
    if (w.ct->done) return;
    if (threadIdx.x == 0) w_ctrlA(pr, w);
}

// gain ratio / accept / reject / stage transitions, then (once per LBA, when stage 0 has just ended) the observation gate
__global__ __launch_bounds__(256) void k_w_ctrlB(const LbaProblem* probs, const LbaWide* ws) {
    const LbaProblem& pr = probs[blockIdx.y];
    const LbaWide& w = ws[blockIdx.y];
    if (w.ct->done) return;
    __shared__ int s_gate;
    // (Measured and not kept: the trial's partial sums staged through LDS by the whole workgroup and the control block worked on in registers — 9.6 -> 12.3 us per
    // launch of 40 windows, 10.5 -> 18.5 us at 128: the step is short next to two more barriers and 256 busy threads per window.)
    if (threadIdx.x == 0) {
        w_ctrlB(pr, w, blockIdx.y);
        s_gate = w.ct->gate && !w.ct->done;
    }
    __syncthreads();
    if (s_gate) {
        w_gate_block(pr, w);
        __syncthreads();
        if (threadIdx.x == 0) w.ct->gate = 0;
    }
}

__global__ __launch_bounds__(256) void k_w_final(const LbaProblem* probs, const LbaWide* ws) {
    const LbaProblem& pr = probs[blockIdx.y];
    const LbaWide& w = ws[blockIdx.y];
    const LbaCtrl* ct = w.ct;
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const bool early = ct->early != 0;
    const double* X = w_X(pr, ct->cur);
    const SE3* T = w.T + ct->cur * pr.K;
    if (gid < pr.E && !early) {
        const int e = gid, a = pr.e_kf[e], p = pr.e_pt[e];
        const bool stereo = !(pr.e_obs[e * 3 + 2] < 0);
        const double Xw[3] = {X[p * 3], X[p * 3 + 1], X[p * 3 + 2]};
        double pc[3];
        se3_map(T[a], Xw, pc);
        pr.erase[e] = (pr.chi2[e] > (stereo ? 7.815 : 5.991) || !(pc[2] > 0.0)) ? 1 : 0;
    }
    if (gid < pr.K) {
        const int a = gid;
        if (pr.fixed[a] != 1 && !early) se3_to_T(T[a], pr.poses_out + a * 16);
        else for (int i = 0; i < 16; i++) pr.poses_out[a * 16 + i] = pr.poses[a * 16 + i];
    }
    if (gid < pr.P * 3) pr.points_out[gid] = early ? pr.points[gid] : (float)X[gid];
    if (gid == 0) { pr.stats[0] = ct->its[0]; pr.stats[1] = ct->trials[0]; pr.stats[2] = ct->its[1]; pr.stats[3] = ct->trials[1]; }
}

#include "lba_win.inc"

}  // namespace oslam

using namespace oslam;


struct oslam_lba {
    int device = 0, max_batch = 0, max_kf = 0;
    // Grow-only arenas, carved per launch for the windows of that launch (no per-window capacity: points and edges are limited by memory only):
    //   in   : LbaProblem[n], LbaWide[n], Schur pair lists, and every window's input arrays — filled in the pinned mirror, ONE upload per launch
    //   work : solver state of every window (never copied)
    //   out  : poses_out / points_out / erase / stats of every window — ONE download per launch
    struct Pool { void* p = nullptr; size_t cap = 0; };
    Pool in_d, work_d, out_d;
    uint8_t* in_h = nullptr; size_t in_h_cap = 0, in_off = 0;
    uint8_t* out_h = nullptr; size_t out_h_cap = 0;
    LbaCtrl* h_ctrl = nullptr; size_t h_ctrl_cap = 0;          // pinned copy of the control blocks (the host polls `done`)
    hipStream_t strm = nullptr;   // every copy and launch of this handle (non-blocking: handles driven by different host threads overlap on the GPU)
    bool owns_strm = true;        // false: the stream of the driver handle this solver belongs to (lba_use_stream)
    std::mutex* launch_gate = nullptr;   // held from the upload to the download of a call when set (lba_use_gate): solvers sharing a gate take turns on the device
    bool device_pairs = true;     // pair lists of the gather Schur built by k_w_pair_* (OSLAM_LBA_HOST_PAIRS=1: by lba_build on the host, the round-2 path)
    long long prof_pre_upload_ns = 0;   // host time of the last lba_launch before its upload (OSLAM_LBA_HOSTPROF)
    int wide = 1;                 // 1: every LM trial of all windows as whole-GPU launches, 0: one workgroup per window in one launch (k_lba, the round-1 kernel),
                                  // 2: one workgroup per window, LDS-resident reduced system (k_lba_win); windows that do not fit its LDS go through layout 1
    size_t win_lds_max = 0;       // dynamic LDS a k_lba_win workgroup may use
    int edge_rec = 1;             // pair gather with per-landmark inverses: compact 32-byte edge records instead of the 144-byte B_e blocks (k_w_schur_rec; round 5 default), 0 = materialised B_e (OSLAM_LBA_REC)
    int schur_vinv = 1;           // pair gather: W_e = B_e V^-1 formed inside k_w_schur from per-landmark inverses (1, default) or materialised per edge by k_w_edgeW (0: rounds 1-3)
    int schur_tiles = 0;          // wide layout, Schur complement (default 0: in the bench the gather is as fast or faster at every window size, see below): 1 = by LDS tiles (k_w_schur_tiles: one coalesced read of Hpl per trial), 0 = by the pair gather
                                  // (k_w_edgeW + k_w_schur: 288 bytes per pair from memory), 2 = per call: tiles when the windows average >= kSchurTilesMinEdges edges.
                                  // Measured (tools/lba_win_prof.py, kernels of one call): 50 windows of 10 keyframes / 4.4 k edges: gather 3.2 ms, tiles 5.7 ms; 40 windows
                                  // of 27 keyframes / 13 k edges: gather 8.7 ms, tiles 9.2 ms alone, but in the bench (8 handles, ~330 such windows in flight: the W / B
                                  // blocks no longer fit the 256 MB cache) the local-BA group takes 5.9 s with tiles against 6.9 s with the gather
    int chol_mode = 0;            // reduced-system solver: 0 auto (LDS-resident scalar kernel while it fits, matrix cores beyond), 1 always MFMA, 2 never
    struct Prep {                 // one prepared window: host-built arrays, then offsets into the `in` arena
        LbaProblem pr;            // scalar fields valid; pointers filled at launch
        const float* p_poses = nullptr; const uint8_t* p_fixed = nullptr; const float* p_points = nullptr;   // the caller's arrays (copied into the arena by lba_place)
        std::vector<int> ekf, ept, pt_start, pose_start, pose_edges, pstart, chunk_kf, chunk_e0, chunk_n, kf_chunk0, pt_edge, tile_p0, tile_s0, stg_edge;
        std::vector<int> thr_own, blk_thr, blk_slots;
        int TE = 0, TP = 0, region_doubles = 0, ngroup = 1, hs_global = 0;
        std::vector<float> eobs, einfo;
        std::vector<int2> pairs;
        size_t o_poses, o_fixed, o_points, o_ekf, o_ept, o_eobs, o_einfo, o_ptstart, o_posestart, o_poseedges, o_pairs, o_pstart;
        size_t o_chunk_kf, o_chunk_e0, o_chunk_n, o_kf_chunk0, o_pt_edge, o_tile_p0, o_tile_s0, o_stg_edge, o_thr_own, o_blk_thr, o_blk_slots;
        size_t o_out_poses, o_out_points, o_out_erase, o_out_stats;   // offsets into the `out` arena
        int nfree = 0, nblk = 1; size_t npairs = 0; bool dev_pairs = false;   // dev_pairs: the Schur pair lists are built on the device (npairs = their total)
        int layout = 0;           // 0 point-major edge numbering (compact / wide kernels), 1 keyframe-major (one workgroup per window, lba_win.inc)
        std::vector<int> order;   // edge permutation (kernel numbering -> caller order)
    };
    std::vector<Prep> prep;
    int n_prep = 0;               // windows of the current call (prep keeps its capacity)
    int* h_stop = nullptr;        // pinned, device-visible stop flag
    int* d_stop = nullptr;
    size_t lds = 0;
    // kernel timing (bench.py's roofline): HIP events on this handle's stream around the solve kernels
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int timing = 0;
    double kern_ms = 0;
    long long kern_launches = 0;
};

static void lba_time_begin(oslam_lba* h) { if (h->timing && h->ev0) (void)hipEventRecord(h->ev0, h->strm); }
static void lba_time_end(oslam_lba* h) { if (h->timing && h->ev1) (void)hipEventRecord(h->ev1, h->strm); }
static void lba_time_collect(oslam_lba* h, long long launches) {   // after the stream has been synchronised
    if (!h->timing || !h->ev0 || !h->ev1) return;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess) { h->kern_ms += ms; h->kern_launches += launches; }
}

static int pool_ensure(oslam_lba::Pool& q, size_t bytes) {
    if (bytes <= q.cap) return OSLAM_OK;
    OSLAM_HIP_CHECK(hipDeviceSynchronize());
    if (q.p) (void)hipFree(q.p);
    q.p = nullptr; q.cap = 0;
    const size_t cap = bytes + bytes / 2 + (1u << 20);
    hipError_t e = hipMalloc(&q.p, cap);
    if (e != hipSuccess) { set_error("local BA: hipMalloc(%zu) failed: %s", cap, hipGetErrorString(e)); return OSLAM_E_HIP; }
    q.cap = cap;
    return OSLAM_OK;
}

// reserves `bytes` (256-aligned) in the pinned mirror of the `in` arena and returns the offset; the mirror grows by copy
static int in_take(oslam_lba* h, size_t bytes, size_t* off) {
    const size_t need = h->in_off + ((bytes + 255) & ~(size_t)255);
    if (need > h->in_h_cap) {
        const size_t cap = need + need / 2 + (1u << 20);
        uint8_t* nb = nullptr;
        OSLAM_HIP_CHECK(hipStreamSynchronize(h->strm));   // an earlier launch's upload may still read the old block
        OSLAM_HIP_CHECK(hipHostMalloc((void**)&nb, cap, 0));
        if (h->in_h) { memcpy(nb, h->in_h, h->in_off); (void)hipHostFree(h->in_h); }
        h->in_h = nb; h->in_h_cap = cap;
    }
    *off = h->in_off;
    h->in_off = need;
    return OSLAM_OK;
}
static int in_put(oslam_lba* h, const void* src, size_t bytes, size_t* off) {
    const int rc = in_take(h, bytes, off);
    if (rc) return rc;
    if (bytes) memcpy(h->in_h + *off, src, bytes);
    return OSLAM_OK;
}

extern "C" {

int oslam_lba_kernel_time(oslam_lba_t* h, int enable, double* ms_out, long long* launches_out) {
    if (!h) { set_error("NULL handle"); return OSLAM_E_INVALID; }
    OSLAM_HIP_CHECK(hipSetDevice(h->device));
    if (enable && !h->ev0) { OSLAM_HIP_CHECK(hipEventCreate(&h->ev0)); OSLAM_HIP_CHECK(hipEventCreate(&h->ev1)); }
    if (ms_out) *ms_out = h->kern_ms;
    if (launches_out) *launches_out = h->kern_launches;
    h->kern_ms = 0; h->kern_launches = 0;
    h->timing = enable;
    return OSLAM_OK;
}

// Test hook: record the LM trials of window 0 of the wide-layout calls that follow (cap > 0; cap == 0 stops).  The buffer is process-wide: one
// handle traces at a time.
static double* s_lba_trace_buf = nullptr;
static int s_lba_trace_cap = 0;
int oslam_lba_trace(oslam_lba_t* h, int cap) {
    if (!h || cap < 0) { set_error("oslam_lba_trace: bad argument"); return OSLAM_E_INVALID; }
    OSLAM_HIP_CHECK(hipSetDevice(h->device));
    OSLAM_HIP_CHECK(hipDeviceSynchronize());
    double* none = nullptr; int zero = 0;
    OSLAM_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_lba_trace), &none, sizeof(none)));
    OSLAM_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_lba_trace_cap), &zero, sizeof(zero)));
    if (s_lba_trace_buf) (void)hipFree(s_lba_trace_buf);
    s_lba_trace_buf = nullptr; s_lba_trace_cap = 0;
    if (cap == 0) return OSLAM_OK;
    OSLAM_HIP_CHECK(hipMalloc((void**)&s_lba_trace_buf, sizeof(double) * (1 + 6 * (size_t)cap)));
    OSLAM_HIP_CHECK(hipMemset(s_lba_trace_buf, 0, sizeof(double) * (1 + 6 * (size_t)cap)));
    s_lba_trace_cap = cap;
    OSLAM_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_lba_trace), &s_lba_trace_buf, sizeof(s_lba_trace_buf)));
    OSLAM_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_lba_trace_cap), &cap, sizeof(cap)));
    return OSLAM_OK;
}

int oslam_lba_trace_read(oslam_lba_t* h, double* out, int32_t* n) {
    if (!h || !out || !n || !s_lba_trace_buf) { set_error("oslam_lba_trace_read: no trace"); return OSLAM_E_INVALID; }
    OSLAM_HIP_CHECK(hipSetDevice(h->device));
    OSLAM_HIP_CHECK(hipDeviceSynchronize());
    double cnt = 0;
    OSLAM_HIP_CHECK(hipMemcpy(&cnt, s_lba_trace_buf, sizeof(double), hipMemcpyDeviceToHost));
    OSLAM_HIP_CHECK(hipMemcpy(out, s_lba_trace_buf + 1, sizeof(double) * 6 * (size_t)s_lba_trace_cap, hipMemcpyDeviceToHost));
    OSLAM_HIP_CHECK(hipMemset(s_lba_trace_buf, 0, sizeof(double)));
    *n = (int32_t)cnt;
    return OSLAM_OK;
}

void oslam_lba_destroy(oslam_lba_t* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    for (oslam_lba::Pool* q : {&h->in_d, &h->work_d, &h->out_d})
        if (q->p) (void)hipFree(q->p);
    if (h->in_h) (void)hipHostFree(h->in_h);
    if (h->out_h) (void)hipHostFree(h->out_h);
    if (h->h_ctrl) (void)hipHostFree(h->h_ctrl);
    if (h->h_stop) (void)hipHostFree(h->h_stop);
    if (h->strm && h->owns_strm) (void)hipStreamDestroy(h->strm);
    delete h;
}

extern "C++" {
namespace oslam {
void lba_use_gate(oslam_lba* h, std::mutex* gate) { if (h) h->launch_gate = gate; }
void lba_use_stream(oslam_lba* h, hipStream_t s) {
    if (!h || !s) return;
    if (h->strm && h->owns_strm) { (void)hipStreamSynchronize(h->strm); (void)hipStreamDestroy(h->strm); }
    h->strm = s; h->owns_strm = false;
}
}  // namespace oslam
}  // extern "C++"

// max_batch = windows per call; max_keyframes (<= 128) = keyframes per window; max_points / max_edges only size the first reservation: the
// arenas grow with the problems (reference g2o has no such bounds).
int oslam_lba_create(oslam_lba_t** out, int max_batch, int max_keyframes, int max_points, int max_edges, int device) {
    if (!out) { set_error("out is NULL"); return OSLAM_E_INVALID; }
    *out = nullptr;
    if (max_batch < 1 || max_keyframes < 1 || max_keyframes > kLbaMaxWindowKF || max_points < 1 || max_edges < 1) {
        set_error("oslam_lba_create: invalid argument (max_keyframes <= %d)", kLbaMaxWindowKF);
        return OSLAM_E_INVALID;
    }
    int ndev = oslam_device_count();
    if (ndev <= 0) { set_error("no HIP device visible: the gfx950 local BA has no CPU fallback"); return OSLAM_E_HIP; }
    if (device < 0 || device >= ndev) { set_error("device out of range"); return OSLAM_E_INVALID; }
    OSLAM_HIP_CHECK(hipSetDevice(device));
    oslam_lba* h = new oslam_lba();
    if (hipStreamCreateWithFlags(&h->strm, hipStreamNonBlocking) != hipSuccess) { set_error("hipStreamCreate failed"); delete h; return OSLAM_E_HIP; }
    h->device = device; h->max_batch = max_batch; h->max_kf = max_keyframes;
    if (const char* e = getenv("OSLAM_LBA_CHOL_MFMA")) h->chol_mode = atoi(e) ? 1 : 2;   // kernel experiments: 1 = matrix cores for every size, 0 = never
    if (const char* e = getenv("OSLAM_LBA_SOLVER")) { const int v = atoi(e); if (v >= 0 && v <= 4) h->chol_mode = v; }   // A/B knob: oslam_lba_set_solver for every handle of the process
    if (getenv("OSLAM_LBA_HOST_PAIRS")) h->device_pairs = false;
    if (const char* e = getenv("OSLAM_LBA_SCHUR_VINV")) h->schur_vinv = atoi(e) != 0;
    if (const char* e = getenv("OSLAM_LBA_REC")) h->edge_rec = atoi(e) != 0;
    if (const char* e = getenv("OSLAM_LBA_SCHUR_TILES")) h->schur_tiles = atoi(e);   // 0 = always the pair gather, 1 = always tiles, 2 = per call (default)
    if (hipHostMalloc((void**)&h->h_stop, sizeof(int), hipHostMallocMapped) != hipSuccess) { set_error("LBA stop flag allocation failed"); oslam_lba_destroy(h); return OSLAM_E_HIP; }
    *h->h_stop = 0;
    if (hipHostGetDevicePointer((void**)&h->d_stop, h->h_stop, 0) != hipSuccess) { set_error("hipHostGetDevicePointer failed"); oslam_lba_destroy(h); return OSLAM_E_HIP; }
    h->lds = kRowBufBytes + 64;
    h->win_lds_max = kWinLdsMax;
    if (hipFuncSetAttribute((const void*)k_lba, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_w_chol<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kCholLdsN * (kCholLdsN + 1) * (int)sizeof(double)) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_w_chol_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, kMB * (kMfmaMaxN + 2 * kMB) * (int)sizeof(double)) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_lba_win, hipFuncAttributeMaxDynamicSharedMemorySize, kWinLdsMax) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_w_schur_tiles, hipFuncAttributeMaxDynamicSharedMemorySize, kWinLdsMax) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_w_chol_packed, hipFuncAttributeMaxDynamicSharedMemorySize, kCholPackedLds) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_w_chol_lds_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, (int)chol16_lds_bytes(kCholLdsMfmaN)) != hipSuccess) {
        set_error("hipFuncSetAttribute failed"); oslam_lba_destroy(h); return OSLAM_E_HIP;
    }
    *out = h;
    return OSLAM_OK;
}

int oslam_lba_debug_stats(oslam_lba_t* h, int32_t out[16]) {
    if (!h || h->n_prep < 1 || !h->out_h) return OSLAM_E_INVALID;
    memcpy(out, h->out_h + h->prep[0].o_out_stats, 64);   // stats of window 0 of the last call
    return OSLAM_OK;
}

int oslam_lba_set_mode(oslam_lba_t* h, int wide) {
    if (!h || wide < 0 || wide > 2) { set_error("oslam_lba_set_mode: mode must be 0 (compact), 1 (wide) or 2 (one workgroup per window, LDS-resident)"); return OSLAM_E_INVALID; }
    h->wide = wide;
    return OSLAM_OK;
}

int oslam_lba_set_schur(oslam_lba_t* h, int mode) {
    if (!h || mode < 0 || mode > 3) { set_error("oslam_lba_set_schur: mode must be 0 (pair gather), 1 (LDS tiles), 2 (per call) or 3 (pair gather, host-built lists)"); return OSLAM_E_INVALID; }
    h->schur_tiles = mode == 3 ? 0 : mode;
    h->device_pairs = mode != 3 && !getenv("OSLAM_LBA_HOST_PAIRS");
    return OSLAM_OK;
}

int oslam_lba_set_solver(oslam_lba_t* h, int mode) {
    if (!h || mode < 0 || mode > 4) { set_error("oslam_lba_set_solver: bad argument"); return OSLAM_E_INVALID; }
    h->chol_mode = mode;
    return OSLAM_OK;
}

volatile int32_t* oslam_lba_stop_flag(oslam_lba_t* h) { return h ? (volatile int32_t*)h->h_stop : nullptr; }

}  // extern "C"

// Schur tiles of a prepared window (lba_win.inc): points cut into tiles whose free-keyframe edges (<= TE) and pair list fit `region_bytes` of LDS, the
// (tile, block) pair lists on tile-local edge indices and the thread slots of the blocks.  q.pt_edge lists every point's edges by ascending keyframe.
static int lba_build_tiles(oslam_lba::Prep& q, const std::vector<int>& blk, int nfree, int nP, int nE, size_t region_bytes, char* err, size_t errn) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wformat-security"
    auto fail = [&](int rc, const char* fmt, auto... a) { snprintf(err, errn, fmt, a...); return rc; };
#pragma clang diagnostic pop
    const std::vector<int>& pt_start = q.pt_start;
    const std::vector<int>& ekf = q.ekf;
    const int nblk = q.nblk;
    auto tof = [&](int x, int y) { return x * nfree - x * (x - 1) / 2 + (y - x); };
    int TP = kWinTilePointsMax, TE = kWinTileEdgesMax;
    while (TE > 128 && win_tile_doubles(TE, TP) * 8 > region_bytes) TE -= 32;
    q.TE = TE; q.TP = TP;
    q.tile_p0.clear(); q.tile_s0.clear(); q.stg_edge.clear();
    std::vector<long long> load(nblk, 0);   // pairs of every block (for the thread slots below)
    {
        int np = 0, ns = 0;
        q.tile_p0.push_back(0); q.tile_s0.push_back(0);
        std::vector<int> fb;
        for (int p = 0; p < nP; p++) {
            fb.clear();
            for (int i = pt_start[p]; i < pt_start[p + 1]; i++) { const int b2 = blk[ekf[q.pt_edge[i]]]; if (b2 >= 0) fb.push_back(b2); }
            const int fe = (int)fb.size();
            if (np > 0 && (np + 1 > TP || ns + fe > TE)) { q.tile_p0.push_back(p); q.tile_s0.push_back((int)q.stg_edge.size()); np = 0; ns = 0; }
            for (int i = pt_start[p]; i < pt_start[p + 1]; i++) {
                const int e = q.pt_edge[i];
                if (blk[ekf[e]] >= 0) { q.stg_edge.push_back(e); ns++; }
            }
            for (int i = 0; i < fe; i++)   // ascending keyframe index inside a point: fb is ascending
                for (int j = i; j < fe; j++) load[tof(fb[i], fb[j])]++;
            np++;
        }
        q.tile_p0.push_back(nP); q.tile_s0.push_back((int)q.stg_edge.size());
    }
    q.npairs = 0;
    for (int b2 = 0; b2 < nblk; b2++) q.npairs += (size_t)load[b2];
    {   // slots of the Schur phase: every block with pairs gets one thread of one pass; spare threads go where the pairs per thread are highest; ordered by load
        std::vector<int> cnt(nblk, 0);
        int used = 0;
        for (int b2 = 0; b2 < nblk; b2++) if (load[b2] > 0) { cnt[b2] = 1; used++; }
        const int G = std::max(1, (used + kWinThreads - 1) / kWinThreads);
        if (G > kWinGroupsMax) return fail(OSLAM_E_CAPACITY, "%d blocks with pairs > %d", used, kWinGroupsMax * kWinThreads);
        q.ngroup = G;
        const int slots = G * kWinThreads;
        if (used > 0) {   // spare slots (of any pass) go to the blocks with the most pairs per thread
            std::vector<std::pair<double, int>> heap;   // (pairs per thread, block)
            for (int b2 = 0; b2 < nblk; b2++) if (cnt[b2]) heap.push_back(std::make_pair((double)load[b2], b2));
            std::make_heap(heap.begin(), heap.end());
            while (used < slots && !heap.empty()) {
                std::pop_heap(heap.begin(), heap.end());
                const int b2 = heap.back().second;
                heap.pop_back();
                if (cnt[b2] >= kWinSplitMax || load[b2] / cnt[b2] < 8) continue;   // (a thread with a handful of pairs gains nothing from help)
                cnt[b2]++; used++;
                heap.push_back(std::make_pair((double)load[b2] / cnt[b2], b2));
                std::push_heap(heap.begin(), heap.end());
            }
        }
        // one entry per (block, sub), ordered by load; dealt to the passes in turn so that every pass keeps the load order and a similar total
        std::vector<std::pair<double, int>> ord;   // (-pairs per thread, block * 64 + sub)
        for (int b2 = 0; b2 < nblk; b2++) for (int s2 = 0; s2 < cnt[b2]; s2++) ord.push_back(std::make_pair(-(double)load[b2] / cnt[b2], b2 * 64 + s2));
        std::sort(ord.begin(), ord.end());
        q.thr_own.assign((size_t)slots, -1); q.blk_thr.assign(nblk, 0); q.blk_slots.clear();
        std::vector<int> fill(G, 0), slot_of(ord.size());
        for (size_t i = 0; i < ord.size(); i++) {
            const int b2 = ord[i].second / 64, s2 = ord[i].second % 64, g = (int)(i % G), slot = g * kWinThreads + fill[g]++;
            q.thr_own[slot] = b2 | (s2 << 16) | (cnt[b2] << 24);
            slot_of[i] = slot;
        }
        // the slots of every block in sub order (the sums are added in this order)
        std::vector<std::vector<int>> per(nblk);
        for (size_t i = 0; i < ord.size(); i++) { const int b2 = ord[i].second / 64, s2 = ord[i].second % 64; if ((int)per[b2].size() < cnt[b2]) per[b2].resize(cnt[b2]); per[b2][s2] = slot_of[i]; }
        for (int b2 = 0; b2 < nblk; b2++) {
            q.blk_thr[b2] = (int)q.blk_slots.size() | (cnt[b2] << 16);
            for (int s2 = 0; s2 < cnt[b2]; s2++) q.blk_slots.push_back(per[b2][s2]);
        }
        if (q.blk_slots.empty()) q.blk_slots.push_back(0);
    }
    return OSLAM_OK;
}

// ---- preparation of one window (thread-safe: touches only its Prep): validation, the edge orders of the layout, Schur pair lists ----
static int lba_build(const oslam_lba_t* h, oslam_lba::Prep& q, int layout, bool want_pairs, bool use_tiles, int nKF, const float* poses, const uint8_t* fixed, int nP, const float* points,
                     int nE, const int32_t* edge_kf, const int32_t* edge_pt, const float* edge_obs, const float* edge_invSigma2, const float K5[5], int use_stop_flag,
                     const float* poses_out, const float* points_out, const uint8_t* erase, int iters0, int iters1, int nstages, int robust0, float delta_mono,
                     float delta_stereo, char* err, size_t errn) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wformat-security"
    auto fail = [&](int rc, const char* fmt, auto... a) { snprintf(err, errn, fmt, a...); return rc; };
#pragma clang diagnostic pop
    if (!h || !poses || !fixed || !points || !K5 || !poses_out || !points_out || (nE > 0 && (!edge_kf || !edge_pt || !edge_obs || !edge_invSigma2 || !erase)))
        return fail(OSLAM_E_INVALID, "NULL argument");
    if (nKF < 1 || nKF > h->max_kf || nP < 0 || nE < 0)
        return fail(OSLAM_E_CAPACITY, "problem (%d keyframes, %d points, %d edges) exceeds the handle capacity (%d keyframes per window)", nKF, nP, nE, h->max_kf);
    q.layout = layout;
    q.p_poses = poses; q.p_fixed = fixed; q.p_points = points;
    std::vector<int>& pt_start = q.pt_start; std::vector<int>& pose_start = q.pose_start;
    pt_start.assign(nP + 1, 0); pose_start.assign(nKF + 1, 0);
    std::vector<int>& order = q.order;
    order.resize(nE);
    for (int i = 0; i < nE; i++) {
        if (edge_kf[i] < 0 || edge_kf[i] >= nKF || edge_pt[i] < 0 || edge_pt[i] >= nP) return fail(OSLAM_E_INVALID, "edge %d references vertex out of range", i);
        pt_start[edge_pt[i] + 1]++;
        pose_start[edge_kf[i] + 1]++;
    }
    for (int p = 0; p < nP; p++) pt_start[p + 1] += pt_start[p];
    for (int k = 0; k < nKF; k++) pose_start[k + 1] += pose_start[k];
    {   // stable counting sort of the edges by point (O(E); the caller's order inside a point is kept)
        std::vector<int> cur(pt_start.begin(), pt_start.end() - 1);
        for (int i = 0; i < nE; i++) order[cur[edge_pt[i]]++] = i;
    }
    if (layout == 1) {   // keyframe-major: a second stable counting sort, by keyframe -> (keyframe, point) order
        std::vector<int> cur(pose_start.begin(), pose_start.end() - 1), o2(nE);
        for (int i = 0; i < nE; i++) o2[cur[edge_kf[order[i]]]++] = order[i];
        order.swap(o2);
    }
    std::vector<int>& ekf = q.ekf; std::vector<int>& ept = q.ept;
    ekf.resize(nE); ept.resize(nE); q.eobs.resize((size_t)nE * 3); q.einfo.resize(nE);
    for (int i = 0; i < nE; i++) {
        const int sidx = order[i];
        ekf[i] = edge_kf[sidx]; ept[i] = edge_pt[sidx]; q.einfo[i] = edge_invSigma2[sidx];
        for (int k = 0; k < 3; k++) q.eobs[(size_t)i * 3 + k] = edge_obs[(size_t)sidx * 3 + k];
    }
    int nfree = 0;
    for (int k = 0; k < nKF; k++) nfree += fixed[k] ? 0 : 1;
    if (nfree > kLbaMaxKF) return fail(OSLAM_E_CAPACITY, "%d free keyframes > %d", nfree, kLbaMaxKF);
    if (h->wide == 0 && nKF > kLbaMaxKF) return fail(OSLAM_E_CAPACITY, "%d keyframes > %d (compact layout; use mode 1 or 2)", nKF, kLbaMaxKF);
    q.nfree = nfree; q.nblk = std::max(1, nfree * (nfree + 1) / 2);
    std::vector<int> blk(nKF);
    for (int k = 0, nb = 0; k < nKF; k++) blk[k] = fixed[k] ? -1 : nb++;
    auto tof = [&](int x, int y) { return x * nfree - x * (x - 1) / 2 + (y - x); };
    q.npairs = 0; q.pairs.clear(); q.pstart.clear(); q.dev_pairs = false;
    if (layout == 0) {
        q.pose_edges.resize(nE);
        std::vector<int> cur(pose_start.begin(), pose_start.end() - 1);
        for (int i = 0; i < nE; i++) q.pose_edges[cur[ekf[i]]++] = i;
        std::vector<int> seen(nKF, -1);   // the reference has one observation of a point per keyframe
        for (int p = 0; p < nP; p++)
            for (int i = pt_start[p]; i < pt_start[p + 1]; i++) {
                if (seen[ekf[i]] == p) return fail(OSLAM_E_INVALID, "duplicate observation of point %d in keyframe %d", p, ekf[i]);
                seen[ekf[i]] = p;
            }
        q.tile_p0.clear();
        if (want_pairs && use_tiles) {
            // Schur complement by tiles (k_w_schur_tiles): every point's edges by ascending keyframe, then the tile / pair / slot structures of lba_win.inc
            q.pt_edge.resize(nE);
            for (int i = 0; i < nE; i++) q.pt_edge[i] = i;
            for (int p = 0; p < nP; p++) std::sort(q.pt_edge.begin() + pt_start[p], q.pt_edge.begin() + pt_start[p + 1], [&](int x, int y) { return ekf[x] < ekf[y]; });
            const int rc = lba_build_tiles(q, blk, nfree, nP, nE, (size_t)kWinLdsMax, err, errn);
            if (rc) return rc;
        } else if (want_pairs && h->device_pairs && h->wide == 1) {
            // the lists are built on the device (k_w_pair_*): the host only needs their total, sum over the points of k (k + 1) / 2 with k = edges in free keyframes
            size_t tot = 0;
            for (int p = 0; p < nP; p++) {
                if (pt_start[p + 1] - pt_start[p] > 65534) return fail(OSLAM_E_CAPACITY, "point %d has more than 65534 observations", p);
                size_t kf_ = 0;
                for (int i = pt_start[p]; i < pt_start[p + 1]; i++) kf_ += blk[ekf[i]] >= 0;
                tot += kf_ * (kf_ + 1) / 2;
            }
            q.npairs = tot; q.dev_pairs = true;
        } else if (want_pairs) {   // Schur pair lists: static over the LM iterations (edge levels are checked on the device)
            const int nblk = q.nblk;
            std::vector<int>& pstart = q.pstart;
            pstart.assign(nblk + 1, 0);
            for (int p = 0; p < nP; p++)
                for (int i = pt_start[p]; i < pt_start[p + 1]; i++) {
                    const int bi = blk[ekf[i]];
                    if (bi < 0) continue;
                    for (int j = pt_start[p]; j < pt_start[p + 1]; j++) {
                        const int bj = blk[ekf[j]];
                        if (bj >= bi) pstart[tof(bi, bj) + 1]++;
                    }
                }
            for (int t2 = 0; t2 < nblk; t2++) pstart[t2 + 1] += pstart[t2];
            q.npairs = (size_t)pstart[nblk];
            q.pairs.resize(q.npairs);
            std::vector<int> cur2(pstart.begin(), pstart.end() - 1);
            for (int p = 0; p < nP; p++)
                for (int i = pt_start[p]; i < pt_start[p + 1]; i++) {
                    const int bi = blk[ekf[i]];
                    if (bi < 0) continue;
                    for (int j = pt_start[p]; j < pt_start[p + 1]; j++) {
                        const int bj = blk[ekf[j]];
                        if (bj >= bi) q.pairs[cur2[tof(bi, bj)]++] = make_int2(i, j);
                    }
                }
        }
    } else {
        // a point's edges by ascending id (= ascending keyframe), the chunks of every keyframe, the pair lists on the keyframe-major ids
        q.pt_edge.resize(nE);
        {
            std::vector<int> cur(pt_start.begin(), pt_start.end() - 1);
            for (int i = 0; i < nE; i++) q.pt_edge[cur[ept[i]]++] = i;
        }
        for (int p = 0; p < nP; p++)
            for (int i = pt_start[p] + 1; i < pt_start[p + 1]; i++)
                if (ekf[q.pt_edge[i]] == ekf[q.pt_edge[i - 1]]) return fail(OSLAM_E_INVALID, "duplicate observation of point %d in keyframe %d", p, ekf[q.pt_edge[i]]);
        q.chunk_kf.clear(); q.chunk_e0.clear(); q.chunk_n.clear(); q.kf_chunk0.assign(nKF + 1, 0);
        for (int k = 0; k < nKF; k++) {
            q.kf_chunk0[k] = (int)q.chunk_kf.size();
            for (int e0 = pose_start[k]; e0 < pose_start[k + 1]; e0 += 64) { q.chunk_kf.push_back(k); q.chunk_e0.push_back(e0); q.chunk_n.push_back(std::min(64, pose_start[k + 1] - e0)); }
        }
        q.kf_chunk0[nKF] = (int)q.chunk_kf.size();
        // Schur tiles (lba_win.inc): the reduced system and the tile buffers share one LDS region
        const size_t region_bytes = h->win_lds_max - win_persist_bytes(nKF, nfree);
        size_t hs_d = (size_t)win_hs_doubles(nfree);
        q.hs_global = hs_d * 8 > region_bytes ? 1 : 0;
        if (q.hs_global) { const size_t n6 = 6 * (size_t)nfree, pw = ((n6 + 1 + kMB - 1) / kMB + 1) * kMB; hs_d = (size_t)kMB * pw + n6 + 2 * kMB; }   // the matrix-core solver's row panel + vector
        { const int rc = lba_build_tiles(q, blk, nfree, nP, nE, region_bytes, err, errn); if (rc) return rc; }
        q.region_doubles = (int)std::max(hs_d, win_tile_doubles(q.TE, q.TP));
    }
    LbaProblem& pr = q.pr;
    memset(&pr, 0, sizeof(pr));
    pr.K = nKF; pr.P = nP; pr.E = nE;
    pr.stop = use_stop_flag ? h->d_stop : nullptr;
    for (int i = 0; i < 5; i++) pr.K5[i] = K5[i];
    pr.iters0 = iters0; pr.iters1 = iters1; pr.nstages = nstages; pr.robust0 = robust0; pr.delta_mono = delta_mono; pr.delta_stereo = delta_stereo;
    return OSLAM_OK;
}

// copies a prepared window's arrays into the pinned mirror of the `in` arena (serial: the arena is a bump allocator)
static int lba_place(oslam_lba_t* h, oslam_lba::Prep& q) {
    const size_t nKF = q.pr.K, nP = q.pr.P, nE = q.pr.E;
    int rc;
    if ((rc = in_put(h, q.p_poses, nKF * 64, &q.o_poses)) || (rc = in_put(h, q.p_fixed, nKF, &q.o_fixed)) || (rc = in_put(h, q.p_points, nP * 12, &q.o_points)) ||
        (rc = in_put(h, q.ekf.data(), nE * 4, &q.o_ekf)) || (rc = in_put(h, q.ept.data(), nE * 4, &q.o_ept)) || (rc = in_put(h, q.eobs.data(), nE * 12, &q.o_eobs)) ||
        (rc = in_put(h, q.einfo.data(), nE * 4, &q.o_einfo)) || (rc = in_put(h, q.pt_start.data(), (nP + 1) * 4, &q.o_ptstart)))
        return rc;
    q.o_posestart = q.o_poseedges = q.o_pairs = q.o_pstart = q.o_chunk_kf = q.o_chunk_e0 = q.o_chunk_n = q.o_kf_chunk0 = q.o_pt_edge = 0;
    q.o_tile_p0 = q.o_tile_s0 = q.o_stg_edge = q.o_thr_own = q.o_blk_thr = q.o_blk_slots = 0;
    if (q.layout == 0) {
        if ((rc = in_put(h, q.pose_start.data(), (nKF + 1) * 4, &q.o_posestart)) || (rc = in_put(h, q.pose_edges.data(), nE * 4, &q.o_poseedges))) return rc;
        if (!q.tile_p0.empty() &&
            ((rc = in_put(h, q.tile_p0.data(), q.tile_p0.size() * 4, &q.o_tile_p0)) || (rc = in_put(h, q.tile_s0.data(), q.tile_s0.size() * 4, &q.o_tile_s0)) ||
             (rc = in_put(h, q.stg_edge.data(), q.stg_edge.size() * 4, &q.o_stg_edge)) || (rc = in_put(h, q.thr_own.data(), q.thr_own.size() * 4, &q.o_thr_own)) ||
             (rc = in_put(h, q.blk_thr.data(), q.blk_thr.size() * 4, &q.o_blk_thr)) || (rc = in_put(h, q.blk_slots.data(), q.blk_slots.size() * 4, &q.o_blk_slots))))
            return rc;
    } else {
        const size_t nc = q.chunk_kf.size();
        if ((rc = in_put(h, q.chunk_kf.data(), nc * 4, &q.o_chunk_kf)) || (rc = in_put(h, q.chunk_e0.data(), nc * 4, &q.o_chunk_e0)) || (rc = in_put(h, q.chunk_n.data(), nc * 4, &q.o_chunk_n)) ||
            (rc = in_put(h, q.kf_chunk0.data(), (nKF + 1) * 4, &q.o_kf_chunk0)) || (rc = in_put(h, q.pt_edge.data(), nE * 4, &q.o_pt_edge)) ||
            (rc = in_put(h, q.tile_p0.data(), q.tile_p0.size() * 4, &q.o_tile_p0)) || (rc = in_put(h, q.tile_s0.data(), q.tile_s0.size() * 4, &q.o_tile_s0)) ||
            (rc = in_put(h, q.stg_edge.data(), q.stg_edge.size() * 4, &q.o_stg_edge)) || (rc = in_put(h, q.thr_own.data(), q.thr_own.size() * 4, &q.o_thr_own)) ||
            (rc = in_put(h, q.blk_thr.data(), q.blk_thr.size() * 4, &q.o_blk_thr)) || (rc = in_put(h, q.blk_slots.data(), q.blk_slots.size() * 4, &q.o_blk_slots)))
            return rc;
    }
    if (q.layout == 0 && !q.pstart.empty()) {
        if ((rc = in_put(h, q.pairs.data(), q.npairs * sizeof(int2), &q.o_pairs)) || (rc = in_put(h, q.pstart.data(), q.pstart.size() * 4, &q.o_pstart))) return rc;
    }
    return OSLAM_OK;
}

constexpr int kSchurTilesMinEdges = 8000;   // wide layout: mean edges per window from which a call forms the Schur complement by tiles (oslam_lba::schur_tiles)

struct LbaArgs {   // one window as the entry points receive it
    int nKF; const float* poses; const uint8_t* fixed; int nP; const float* points; int nE; const int32_t* edge_kf; const int32_t* edge_pt; const float* edge_obs;
    const float* edge_invSigma2; float* poses_out; float* points_out; uint8_t* erase;
};

// does window (K keyframes, nfree of them free) fit the one-workgroup-per-window kernel?
static bool win_fits(const oslam_lba_t* h, int K, int nfree) {
    if (nfree < 1 || nfree * (nfree + 1) / 2 > kWinGroupsMax * kWinThreads || nfree > kLbaMaxKF) return false;
    const size_t persist = win_persist_bytes(K, nfree);
    if (persist >= h->win_lds_max) return false;
    const size_t n6 = 6 * (size_t)nfree, pw = ((n6 + 1 + kMB - 1) / kMB + 1) * kMB;
    // a tile must hold every point's edges (<= nfree <= 128 <= TE) and, in the worst case, their pairs; a system that does not fit LDS needs the solver's panel
    return persist + 8 * win_tile_doubles(128, kWinTilePointsMax) <= h->win_lds_max &&
           persist + 8 * std::min((size_t)win_hs_doubles(nfree), (size_t)kMB * pw + n6 + 2 * kMB) <= h->win_lds_max;
}

// Prepares all windows of a call (in parallel on the shared workers when there are several) and places them in the `in` arena.
static int lba_prepare_all(oslam_lba_t* h, int n, const LbaArgs* a, const float K5[5], int use_stop_flag, int iters0, int iters1, int nstages, int robust0, float delta_mono,
                           float delta_stereo, std::vector<int>* window_rc = nullptr) {
    h->in_off = 0;
    if ((int)h->prep.size() < n) h->prep.resize(n);   // (the windows' vectors keep their capacity from call to call)
    h->n_prep = n;
    std::vector<int> rcs(n, 0);
    std::vector<std::array<char, 192>> errs(n);
    // wide layout: the Schur complement by tiles needs every window's blocks to fit the thread slots (<= 63 free keyframes); otherwise the whole call gathers pairs
    long long sumE = 0;
    for (int i = 0; i < n; i++) sumE += a[i].nE;
    bool use_tiles = h->schur_tiles == 1 || (h->schur_tiles == 2 && sumE >= (long long)kSchurTilesMinEdges * n);
    std::vector<int> nfrees(n, 0);
    for (int i = 0; i < n; i++) {
        const LbaArgs& q = a[i];
        if (q.fixed && q.nKF > 0) for (int k = 0; k < q.nKF; k++) nfrees[i] += q.fixed[k] ? 0 : 1;
        if (nfrees[i] * (nfrees[i] + 1) / 2 > kWinGroupsMax * kWinThreads) use_tiles = false;
    }
    auto one = [&](int i) {
        const LbaArgs& q = a[i];
        const int nfree = nfrees[i];
        const int layout = (h->wide == 2 && win_fits(h, q.nKF, nfree)) ? 1 : 0;
        errs[i][0] = 0;
        rcs[i] = lba_build(h, h->prep[i], layout, layout == 1 || h->wide != 0, use_tiles, q.nKF, q.poses, q.fixed, q.nP, q.points, q.nE, q.edge_kf, q.edge_pt, q.edge_obs, q.edge_invSigma2, K5,
                           use_stop_flag, q.poses_out, q.points_out, q.erase, iters0, iters1, nstages, robust0, delta_mono, delta_stereo, errs[i].data(), errs[i].size());
    };
    if (n > 1) oslam_drv::shared_parallel_for(n, one);
    else one(0);
    if (window_rc) *window_rc = rcs;   // (which windows were refused: oslam_lba_optimize_batch solves the others)
    for (int i = 0; i < n; i++) if (rcs[i]) { set_error("%s", errs[i].data()); return rcs[i]; }
    if (window_rc) window_rc->assign(n, 0);
    for (int i = 0; i < n; i++) { const int rc = lba_place(h, h->prep[i]); if (rc) return rc; }
    return OSLAM_OK;
}

// Runs the prepared windows and brings their outputs into the pinned `out` mirror (the stream is drained on return).
// Layout-1 windows: ONE launch of k_lba_win, one workgroup per window (largest first).  Layout-0 windows: compact mode = one launch of k_lba; wide mode =
// every LM trial as eight launches whose grids cover all of them (blockIdx.y = window, blockIdx.x sized for the largest one; finished windows return at once).
static int lba_launch(oslam_lba_t* h) {
    const auto t_launch0 = std::chrono::steady_clock::now();
    const int n = h->n_prep;
    OSLAM_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t st = h->strm;
    int rc;
    std::vector<int> idx0, idx1;   // windows by layout
    for (int i = 0; i < n; i++) (h->prep[i].layout == 1 ? idx1 : idx0).push_back(i);
    const int n0 = (int)idx0.size(), n1 = (int)idx1.size();
    const bool wide = h->wide != 0;   // layout-0 windows of a wide / win handle run in wide mode
    // ---- carve the arenas ----
    size_t o_probs, o_ws, o_wins, o_order;
    if ((rc = in_take(h, sizeof(LbaProblem) * n, &o_probs)) || (rc = in_take(h, sizeof(LbaWide) * std::max(n0, 1), &o_ws)) || (rc = in_take(h, sizeof(LbaWin) * std::max(n1, 1), &o_wins)) ||
        (rc = in_take(h, sizeof(int) * std::max(n1, 1), &o_order)))
        return rc;
    size_t work = 0, outb = 0;
    auto takeW = [&](size_t bytes) { const size_t at = work; work += (bytes + 255) & ~(size_t)255; return at; };
    auto takeO = [&](size_t bytes) { const size_t at = outb; outb += (bytes + 255) & ~(size_t)255; return at; };
    struct WOff { size_t Xa, Xb, chi2, level, Hpl, Hll, Dinv, bl, xl, Hpp, bp, Hs, xp, ctrl, T, R, blk, free_pose, partF, partS, partM, W, rec, eoi, chunkC, pairPart, parts, pairM, pairsW, pstartW; };
    bool tiles = wide && n0 > 0;     // wide layout: Schur complement by tiles when every layout-0 window carries the structures
    for (int i : idx0) tiles = tiles && !h->prep[i].tile_p0.empty();
    const bool use_rec = wide && !tiles && h->schur_vinv && h->edge_rec;   // compact edge records (k_w_schur_rec): no B_e blocks at all
    const int nwg_call = std::max(1, std::min(16, (2 * 256 + std::max(n0, 1) - 1) / std::max(n0, 1)));   // workgroups per window: ~2 per CU over the call
    size_t tiles_lds = 0, packed_lds = 0, ldsm_lds = 0;
    int maxWg = 1, maxSum = 1, min_n6_big = 1 << 30;   // (min_n6_big: the smallest reduced system of the call)
    std::vector<WOff> wo(n);
    bool any_dev_pairs = false;
    int maxNbPt = 1, maxK = 1, maxE = 1, maxBlk = 1, maxFin = 1, maxInit = 1, max_slots = 0, min_group = 4, max_n6 = 0;
    bool all_lds = true;
    const size_t ctrl_base = takeW(sizeof(LbaCtrl) * std::max(n0, 1));   // contiguous: the host polls all of them with one copy
    size_t win_lds = 0;
    for (int i = 0; i < n; i++) {
        oslam_lba::Prep& q = h->prep[i];
        const size_t K = q.pr.K, P = q.pr.P, E = q.pr.E, n6 = 6 * (size_t)q.nfree;
        const int nbpt = div_up(std::max((int)P, 1), kWPt);
        WOff& o = wo[i];
        o.Xa = takeW(P * 24); o.Xb = takeW(P * 24); o.chi2 = takeW(E * 8); o.level = takeW(E); o.Hll = takeW(P * 72); o.bl = takeW(P * 24);
        if (q.layout == 1) {
            o.chunkC = takeW(q.chunk_kf.size() * 27 * 8); o.pairPart = takeW((size_t)q.ngroup * kWinThreads * 42 * 8);
            o.Hs = q.hs_global ? takeW(n6 * (n6 + 1) * 8) : 0;
            win_lds = std::max(win_lds, win_persist_bytes((int)K, q.nfree) + 8 * (size_t)q.region_doubles);
        } else {
            o.Hpl = use_rec ? 0 : takeW(E * 144); o.Dinv = takeW(P * 72); o.xl = takeW(P * 24); o.Hpp = takeW(K * 288); o.bp = takeW(K * 48); o.Hs = takeW(n6 * (n6 + 1) * 8); o.xp = takeW((n6 + 8) * 8);
            if (wide) {
                const int j = (int)(std::find(idx0.begin(), idx0.end(), i) - idx0.begin());
                o.ctrl = ctrl_base + sizeof(LbaCtrl) * j; o.T = takeW(sizeof(SE3) * 2 * K); o.R = takeW(144 * K); o.blk = takeW(4 * K); o.free_pose = takeW(4 * K);
                o.partF = takeW(8 * (nbpt + 2)); o.partS = takeW(8 * (nbpt + 2)); o.partM = takeW(8 * (nbpt + 2));
                if (tiles) {
                    const int ntile = (int)q.tile_p0.size() - 1, nwg = std::max(1, std::min(nwg_call, (ntile + 1) / 2));   // >= 2 tiles per workgroup (its block sums cost a pass over all slots)
                    o.parts = takeW((size_t)nwg * q.ngroup * kWinThreads * 42 * 8);
                    tiles_lds = std::max(tiles_lds, 8 * win_tile_doubles(q.TE, q.TP));
                    maxWg = std::max(maxWg, nwg); maxSum = std::max(maxSum, div_up(q.nblk * 42, 256));
                } else {
                    // (per-landmark inverses: 48 bytes per point, whatever the edge count — a window may hold points without edges)
                    o.W = takeW(use_rec ? P * 48 : std::max(E * 144, h->schur_vinv ? P * 48 : (size_t)0));
                    o.rec = use_rec ? takeW(E * 32) : 0;
                    o.eoi = use_rec ? takeW(E * 16) : 0;
                    if (q.dev_pairs) { o.pairsW = takeW(std::max<size_t>(q.npairs, 1) * 8); o.pstartW = takeW(((size_t)q.nblk + 1) * 4); any_dev_pairs = true; }   // (pairM: one block for the call, below)
                }
                if ((int)n6 <= kCholPackedN) packed_lds = std::max(packed_lds, ((size_t)win_hs_doubles(q.nfree) + n6 / 2 + 4 + n6 + 8) * 8);
                if ((int)n6 <= kCholLdsMfmaN) ldsm_lds = std::max(ldsm_lds, chol16_lds_bytes((int)n6));
                min_n6_big = std::min(min_n6_big, (int)n6);
            }
            maxNbPt = std::max(maxNbPt, nbpt); maxK = std::max(maxK, (int)K); maxE = std::max(maxE, (int)E); maxBlk = std::max(maxBlk, q.nblk);
            maxFin = std::max(maxFin, (int)std::max(std::max(E, K), P * 3)); maxInit = std::max(maxInit, (int)std::max(P * 3, E));
            const int its = q.pr.iters0 + (q.pr.nstages > 1 ? q.pr.iters1 : 0);
            max_slots = std::max(max_slots, its * 10 + 8); min_group = std::max(min_group, its);
            all_lds = all_lds && (int)n6 <= kCholLdsN; max_n6 = std::max(max_n6, (int)n6);
        }
        q.o_out_poses = takeO(K * 64); q.o_out_points = takeO(P * 12); q.o_out_erase = takeO(E); q.o_out_stats = takeO(64);
    }
    // the point x free-keyframe matrices of the device-built pair lists: one contiguous block (one memset per call)
    const size_t pairM_base = work;
    if (any_dev_pairs && !tiles)
        for (int i = 0; i < n; i++) {
            const oslam_lba::Prep& q = h->prep[i];
            if (q.layout == 0 && q.dev_pairs) wo[i].pairM = takeW(std::max<size_t>((size_t)q.pr.P * q.nfree, 1) * 2);
        }
    const size_t pairM_bytes = work - pairM_base;
    if ((rc = pool_ensure(h->in_d, h->in_off)) || (rc = pool_ensure(h->work_d, work)) || (rc = pool_ensure(h->out_d, outb))) return rc;
    if (outb > h->out_h_cap) {
        if (h->out_h) (void)hipHostFree(h->out_h);
        h->out_h = nullptr; h->out_h_cap = 0;
        OSLAM_HIP_CHECK(hipHostMalloc((void**)&h->out_h, outb + outb / 2, 0));
        h->out_h_cap = outb + outb / 2;
    }
    if (wide && n0 > 0 && sizeof(LbaCtrl) * n0 > h->h_ctrl_cap) {
        if (h->h_ctrl) (void)hipHostFree(h->h_ctrl);
        h->h_ctrl = nullptr; h->h_ctrl_cap = 0;
        OSLAM_HIP_CHECK(hipHostMalloc((void**)&h->h_ctrl, sizeof(LbaCtrl) * n0 * 2, 0));
        h->h_ctrl_cap = sizeof(LbaCtrl) * n0 * 2;
    }
    uint8_t* I = (uint8_t*)h->in_d.p; uint8_t* Wk = (uint8_t*)h->work_d.p; uint8_t* O = (uint8_t*)h->out_d.p;
    // the problem records: layout-0 windows first (slots 0 .. n0-1 of probs / ws), then the layout-1 windows (slots n0 .. n-1 of probs, 0 .. n1-1 of wins)
    LbaProblem* hp = (LbaProblem*)(h->in_h + o_probs);
    LbaWide* hw = (LbaWide*)(h->in_h + o_ws);
    LbaWin* hwin = (LbaWin*)(h->in_h + o_wins);
    int* horder = (int*)(h->in_h + o_order);
    auto fill_problem = [&](int i) {
        const oslam_lba::Prep& q = h->prep[i];
        const WOff& o = wo[i];
        LbaProblem pr = q.pr;
        pr.poses = (const float*)(I + q.o_poses); pr.fixed = I + q.o_fixed; pr.points = (const float*)(I + q.o_points); pr.e_kf = (const int*)(I + q.o_ekf);
        pr.e_pt = (const int*)(I + q.o_ept); pr.e_obs = (const float*)(I + q.o_eobs); pr.e_info = (const float*)(I + q.o_einfo); pr.pt_start = (const int*)(I + q.o_ptstart);
        pr.pose_start = (const int*)(I + q.o_posestart); pr.pose_edges = (const int*)(I + q.o_poseedges);
        pr.Xa = (double*)(Wk + o.Xa); pr.Xb = (double*)(Wk + o.Xb); pr.chi2 = (double*)(Wk + o.chi2); pr.level = Wk + o.level;
        pr.Hll = (double*)(Wk + o.Hll); pr.bl = (double*)(Wk + o.bl);
        if (q.layout == 0) {
            pr.Hpl = use_rec ? nullptr : (double*)(Wk + o.Hpl);
            pr.Dinv = (double*)(Wk + o.Dinv); pr.xl = (double*)(Wk + o.xl); pr.Hpp = (double*)(Wk + o.Hpp); pr.bp = (double*)(Wk + o.bp); pr.Hs = (double*)(Wk + o.Hs);
            pr.xp = (double*)(Wk + o.xp);
        }
        pr.poses_out = (float*)(O + q.o_out_poses); pr.points_out = (float*)(O + q.o_out_points); pr.erase = O + q.o_out_erase; pr.stats = (int*)(O + q.o_out_stats);
        return pr;
    };
    for (int j = 0; j < n0; j++) {
        const int i = idx0[j];
        const oslam_lba::Prep& q = h->prep[i];
        const WOff& o = wo[i];
        hp[j] = fill_problem(i);
        LbaWide w;
        memset(&w, 0, sizeof(w));
        if (wide) {
            w.ct = (LbaCtrl*)(Wk + o.ctrl); w.T = (SE3*)(Wk + o.T); w.R = (double*)(Wk + o.R); w.blk = (int*)(Wk + o.blk); w.free_pose = (int*)(Wk + o.free_pose);
            w.partF = (double*)(Wk + o.partF); w.partS = (double*)(Wk + o.partS); w.partM = (double*)(Wk + o.partM); w.W = (double*)(Wk + o.W);
            w.rec = use_rec ? (double*)(Wk + o.rec) : nullptr;
            w.eoi = use_rec ? (float4*)(Wk + o.eoi) : nullptr;
            w.nblk_pt = div_up(std::max(q.pr.P, 1), kWPt);
            w.pairs = (const int2*)(I + q.o_pairs); w.pair_start = (const int*)(I + q.o_pstart);
            if (q.dev_pairs && !tiles) {
                w.pairM = (uint16_t*)(Wk + o.pairM); w.pairs_w = (int2*)(Wk + o.pairsW); w.pair_start_w = (int*)(Wk + o.pstartW);
                w.pairs = w.pairs_w; w.pair_start = w.pair_start_w;
            }
            if (tiles) {
                w.W = nullptr; w.rec = nullptr; w.eoi = nullptr;
                w.tile_p0 = (const int*)(I + q.o_tile_p0); w.tile_s0 = (const int*)(I + q.o_tile_s0); w.stg_edge = (const int*)(I + q.o_stg_edge);
                w.thr_own = (const int*)(I + q.o_thr_own);
                w.blk_thr = (const int*)(I + q.o_blk_thr); w.blk_slots = (const int*)(I + q.o_blk_slots); w.parts = (double*)(Wk + o.parts);
                w.ntile = (int)q.tile_p0.size() - 1; w.ngroup = q.ngroup; w.nwg = std::max(1, std::min(nwg_call, (w.ntile + 1) / 2)); w.TE = q.TE; w.TP = q.TP; w.nblk = q.nblk;
            }
        }
        hw[j] = w;
    }
    std::vector<std::pair<long long, int>> cost(n1);
    for (int j = 0; j < n1; j++) {
        const int i = idx1[j];
        const oslam_lba::Prep& q = h->prep[i];
        const WOff& o = wo[i];
        hp[n0 + j] = fill_problem(i);
        LbaWin w;
        memset(&w, 0, sizeof(w));
        w.chunk_kf = (const int*)(I + q.o_chunk_kf); w.chunk_e0 = (const int*)(I + q.o_chunk_e0); w.chunk_n = (const int*)(I + q.o_chunk_n); w.kf_chunk0 = (const int*)(I + q.o_kf_chunk0);
        w.pt_edge = (const int*)(I + q.o_pt_edge); w.tile_p0 = (const int*)(I + q.o_tile_p0); w.tile_s0 = (const int*)(I + q.o_tile_s0); w.stg_edge = (const int*)(I + q.o_stg_edge);
        w.thr_own = (const int*)(I + q.o_thr_own); w.blk_thr = (const int*)(I + q.o_blk_thr); w.blk_slots = (const int*)(I + q.o_blk_slots);
        w.chunkC = (double*)(Wk + o.chunkC); w.pairPart = (double*)(Wk + o.pairPart); w.HsG = q.hs_global ? (double*)(Wk + o.Hs) : nullptr;
        w.ngroup = q.ngroup; w.hs_global = q.hs_global;
        w.nchunk = (int)q.chunk_kf.size(); w.npairs = (int)q.npairs; w.nfree = q.nfree; w.nblk = q.nblk; w.ntile = (int)q.tile_p0.size() - 1;
        w.TE = q.TE; w.TP = q.TP; w.hs_doubles = win_hs_doubles(q.nfree); w.region_doubles = q.region_doubles;
        hwin[j] = w;
        const long long n6 = 6LL * q.nfree;
        cost[j] = std::make_pair(-(700LL * q.pr.E + 216LL * (long long)q.npairs + n6 * n6 * n6 / 3), j);
    }
    std::sort(cost.begin(), cost.end());   // the most expensive windows are dispatched first
    for (int j = 0; j < n1; j++) horder[j] = cost[j].second;
    // OSLAM_LBA_CONCURRENCY=k: at most k local-BA calls of this process have device work in flight at a time (A/B knob: two overlapping calls both take about
    // twice as long, one after the other the first is back after half of that)
    struct Gate {
        std::mutex m; std::condition_variable cv; int free_slots;
        explicit Gate(int k) : free_slots(k) {}
        void enter() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return free_slots > 0; }); free_slots--; }
        void leave() { { std::lock_guard<std::mutex> lk(m); free_slots++; } cv.notify_one(); }
    };
    static Gate* gate = [] { const char* e = getenv("OSLAM_LBA_CONCURRENCY"); const int k = e ? atoi(e) : 0; return k > 0 ? new Gate(k) : (Gate*)nullptr; }();
    struct GateScope { Gate* g; explicit GateScope(Gate* g_) : g(g_) { if (g) g->enter(); } ~GateScope() { if (g) g->leave(); } } gate_scope(gate);
    h->prof_pre_upload_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_launch0).count();
    OSLAM_HIP_CHECK(hipMemcpyAsync(I, h->in_h, h->in_off, hipMemcpyHostToDevice, st));   // the ONE upload
    // Solvers that share a gate run their KERNELS in turns: the window preparation above and the upload (a DMA transfer on this solver's own stream: tens of
    // MB per call) proceed while the other solver has the device.
    std::unique_lock<std::mutex> handle_gate;
    if (h->launch_gate) {
        OSLAM_HIP_CHECK(stream_wait(st));   // (the upload has landed: what the gate then covers is kernel time only)
        handle_gate = std::unique_lock<std::mutex>(*h->launch_gate);
    }
    const LbaProblem* d_probs = (const LbaProblem*)(I + o_probs);
    const LbaWide* d_ws = (const LbaWide*)(I + o_ws);
    long long launches = 0;
    lba_time_begin(h);
    if (n1 > 0) {
        hipLaunchKernelGGL(k_lba_win, dim3(n1), dim3(kWinThreads), win_lds, st, d_probs + n0, (const LbaWin*)(I + o_wins), (const int*)(I + o_order));
        launches += 1;
    }
    if (n0 > 0 && !wide) {
        hipLaunchKernelGGL(k_lba, dim3(n0), dim3(kLbaThreads), h->lds, st, d_probs);
        launches += 1;
    } else if (n0 > 0) {
        const size_t chol_lds = all_lds ? (size_t)max_n6 * (max_n6 + 1) * sizeof(double) : 0;
        // systems beyond the LDS-resident kernels (full square up to 132 unknowns, packed upper triangle up to kCholPackedN) are factored by the matrix cores
        // (k_w_chol_mfma); h->chol_mode 1 forces them, 2 forbids them
        // per window: the packed LDS kernel up to kCholPackedN unknowns, the matrix cores beyond (both kernels are launched when a call mixes the two)
        // chol_mode 0 (auto): LDS-resident matrix-core kernel (k_w_chol_lds_mfma) up to kCholLdsMfmaN unknowns, global-memory matrix-core kernel beyond;
        // 3: the round-3 choice (scalar packed LDS kernel up to kCholPackedN, matrix cores beyond); 1: global-memory matrix cores for every size; 2: no matrix cores
        const int cm = h->chol_mode;
        // (4: the LDS-resident matrix-core kernel also for calls whose systems would all fit the scalar LDS kernel — A/B of the small-window regime)
        const bool chol_ldsm = (cm == 0 && !all_lds && min_n6_big <= kCholLdsMfmaN) || (cm == 4 && min_n6_big <= kCholLdsMfmaN);
        if (cm == 4) all_lds = false;
        const bool chol_packed = !chol_ldsm && cm != 1 && !all_lds && min_n6_big <= kCholPackedN && (cm == 0 || cm == 3 || max_n6 <= kCholPackedN);
        const bool chol_mfma = cm == 1 || ((cm == 0 || cm == 3 || cm == 4) && !all_lds && max_n6 > (chol_ldsm ? kCholLdsMfmaN : kCholPackedN));
        const int mfma_min_n = chol_ldsm ? kCholLdsMfmaN + 1 : (chol_packed ? kCholPackedN + 1 : 0);
        const size_t mfma_lds = (size_t)kMB * (((max_n6 + 1 + kMB - 1) / kMB + 1) * kMB) * sizeof(double);
        hipLaunchKernelGGL(k_w_init, dim3(1, n0), dim3(256), 0, st, d_probs, d_ws);
        hipLaunchKernelGGL(k_w_init_arrays, dim3(div_up(maxInit, 256), n0), dim3(256), 0, st, d_probs, d_ws);
        if (any_dev_pairs && !tiles) {   // the Schur pair lists of the call, once (k_w_init has numbered the free keyframes)
            OSLAM_HIP_CHECK(hipMemsetAsync(Wk + pairM_base, 0, pairM_bytes, st));
            hipLaunchKernelGGL(k_w_pair_matrix, dim3(div_up(maxE, 256), n0), dim3(256), 0, st, d_probs, d_ws);
            hipLaunchKernelGGL(k_w_pair_blocks<false>, dim3(maxBlk, n0), dim3(64), 0, st, d_probs, d_ws);
            hipLaunchKernelGGL(k_w_pair_scan, dim3(n0), dim3(64), 0, st, d_probs, d_ws);
            hipLaunchKernelGGL(k_w_pair_blocks<true>, dim3(maxBlk, n0), dim3(64), 0, st, d_probs, d_ws);
            launches += 4;
        }
        // worst case 15 iterations x 10 trials; slots past `done` return at once.  First group = the minimum number of LM trials (one per
        // iteration), so the common case needs a single host round trip; rejected steps add groups of 4.
        const int ny_xcd = n0 >= 8 ? (n0 + 7) / 8 * 8 : n0;   // grid rows of the kernels that map a window to one XCD (xcd_window_item)
        static const double gate_release_frac = [] { const char* e = getenv("OSLAM_LBA_GATE_RELEASE"); return e ? atof(e) : 0.0; }();
        static const bool call_stats = getenv("OSLAM_LBA_CALL_STATS") != nullptr;   // per host poll: windows of the call, trial slots enqueued so far, windows not done
        // LM control step in the last workgroup of k_w_update instead of its own launch (5 launches per trial instead of 6).  Default OFF, measured: the device-scope release
        // fence every workgroup needs before it takes its ticket (an L2 write-back on gfx950: the window's workgroups may sit on different XCDs) costs more than the
        // launch it saves — 314 against 294 us per trial at 40 windows, 750 against 640 us at 128 (same box, alternating runs; tests/test_lba_gpu.py green both ways).
        static const bool fold_ctrl = [] { const char* e = getenv("OSLAM_LBA_FOLD_CTRL"); return e && atoi(e) != 0; }();
        // lanes per landmark in k_w_update (OSLAM_LBA_UPD_LANES = 1 / 2 / 4 overrides).  us per LM trial of a call, 1 / 2 / 4 lanes, same box, two sweeps: 40 windows
        // 293 / 279 / 283, 96 windows 494 / 485 / 498, 128 windows 639 / 616 / 641, 256 windows 1223 / 1180 / 1232: the one-lane form waits (~8 edges walked twice in one
        // dependent chain), four lanes repeat the per-landmark part (inverse, step, stores) four times
        static const int upd_lanes_env = getenv("OSLAM_LBA_UPD_LANES") ? atoi(getenv("OSLAM_LBA_UPD_LANES")) : 0;
        const int upd_lanes = upd_lanes_env ? upd_lanes_env : 2;
        int slots_done = 0, group = min_group;
        while (slots_done < max_slots) {
            for (int sl = 0; sl < group; sl++, slots_done++) {
                if (use_rec) hipLaunchKernelGGL(k_w_lin<true>, dim3(maxNbPt + maxK, ny_xcd), dim3(kLinThreads), 0, st, d_probs, d_ws, n0);
                else hipLaunchKernelGGL(k_w_lin<false>, dim3(maxNbPt + maxK, ny_xcd), dim3(kLinThreads), 0, st, d_probs, d_ws, n0);
                if (tiles) {
                    hipLaunchKernelGGL(k_w_ctrlA, dim3(1, n0), dim3(64), 0, st, d_probs, d_ws);   // (the pair-gather path folds this step into k_w_edgeW / k_w_ctrlB)
                    hipLaunchKernelGGL(k_w_schur_tiles, dim3(maxWg, n0), dim3(kWinThreads), tiles_lds, st, d_probs, d_ws);
                    hipLaunchKernelGGL(k_w_schur_sum, dim3(maxSum, n0), dim3(256), 0, st, d_probs, d_ws);
                } else {
                    if (use_rec) {
                        hipLaunchKernelGGL(k_w_edgeW<true>, dim3(div_up(maxNbPt * kWPt, 256), ny_xcd), dim3(256), 0, st, d_probs, d_ws, n0);
                        hipLaunchKernelGGL(k_w_schur_rec, dim3(maxBlk, ny_xcd), dim3(64), 0, st, d_probs, d_ws, n0);
                    } else if (h->schur_vinv) {
                        hipLaunchKernelGGL(k_w_edgeW<true>, dim3(div_up(maxNbPt * kWPt, 256), ny_xcd), dim3(256), 0, st, d_probs, d_ws, n0);
                        hipLaunchKernelGGL(k_w_schur<true>, dim3(maxBlk, ny_xcd), dim3(64), 0, st, d_probs, d_ws, n0);
                    } else {
                        hipLaunchKernelGGL(k_w_edgeW<false>, dim3(div_up(maxE, 256), ny_xcd), dim3(256), 0, st, d_probs, d_ws, n0);
                        hipLaunchKernelGGL(k_w_schur<false>, dim3(maxBlk, ny_xcd), dim3(64), 0, st, d_probs, d_ws, n0);
                    }
                }
                if (chol_ldsm) hipLaunchKernelGGL(k_w_chol_lds_mfma, dim3(1, n0), dim3(kWinThreads), ldsm_lds, st, d_probs, d_ws, 0);
                if (chol_packed) hipLaunchKernelGGL(k_w_chol_packed, dim3(1, n0), dim3(kWinThreads), packed_lds, st, d_probs, d_ws);
                if (chol_mfma) hipLaunchKernelGGL(k_w_chol_mfma, dim3(1, n0), dim3(kMfmaThreads), mfma_lds, st, d_probs, d_ws, mfma_min_n);
                if (chol_ldsm || chol_packed || chol_mfma) { }
                else if (chol_lds) hipLaunchKernelGGL(k_w_chol<true>, dim3(1, n0), dim3(1024), chol_lds, st, d_probs, d_ws);
                else hipLaunchKernelGGL(k_w_chol<false>, dim3(1, n0), dim3(1024), 0, st, d_probs, d_ws);
                // (+ the trial's chi2: k_w_eval of rounds 1-3; OSLAM_LBA_FOLD_CTRL=1: + the LM control step in the window's last workgroup)
                {
                    const dim3 ug(maxNbPt + 1, ny_xcd);
#define OSLAM_UPD_LAUNCH(R, F, L) hipLaunchKernelGGL((k_w_update<R, F, L>), ug, dim3(kWPt * L), 0, st, d_probs, d_ws, n0)
#define OSLAM_UPD_LANES(R, F) do { if (upd_lanes == 4) OSLAM_UPD_LAUNCH(R, F, 4); else if (upd_lanes == 2) OSLAM_UPD_LAUNCH(R, F, 2); else OSLAM_UPD_LAUNCH(R, F, 1); } while (0)
                    if (fold_ctrl) { if (use_rec) OSLAM_UPD_LANES(true, true); else OSLAM_UPD_LANES(false, true); }
                    else { if (use_rec) OSLAM_UPD_LANES(true, false); else OSLAM_UPD_LANES(false, false); }
#undef OSLAM_UPD_LANES
#undef OSLAM_UPD_LAUNCH
                    if (!fold_ctrl) hipLaunchKernelGGL(k_w_ctrlB, dim3(1, n0), dim3(256), 0, st, d_probs, d_ws);
                }
            }
            OSLAM_HIP_CHECK(copy_to_host_async(h->h_ctrl, Wk + ctrl_base, sizeof(LbaCtrl) * n0, st));   // (a copy kernel, not the SDMA ring: common.h)
            OSLAM_HIP_CHECK(stream_wait(st));
            bool all_done = true;
            int n_active = 0;
            for (int i = 0; i < n0; i++) { all_done = all_done && h->h_ctrl[i].done != 0; n_active += h->h_ctrl[i].done == 0; }
            if (call_stats) fprintf(stderr, "[lba call] %d windows, %d slots done, %d active\n", n0, slots_done, n_active);
            if (all_done) break;
            // The tail of a call — the few windows whose LM rejects steps (after the first 15 slots ~16 % of the windows of a steady-state call are still active,
            // after 19 slots ~6 %) — is a chain of latency-bound launches that leaves the card almost idle: the solver that shares the gate may start its call
            // beside it when OSLAM_LBA_GATE_RELEASE=f is set (release once <= f x windows are active).  Default off: same-box A/B of the headline with f = 0.25,
            // alternating runs: 35.2 / 38.7 k frames/s against 37.7 / 38.5 k without — the overlapping call stretches this call's kernels (local-BA device time
            // 3.6 -> 3.9-4.4 s per 20 steps) by what the tail used to idle.
            if (handle_gate.owns_lock() && gate_release_frac > 0 && (double)n_active <= gate_release_frac * n0) handle_gate.unlock();
            group = 4;
        }
        hipLaunchKernelGGL(k_w_final, dim3(div_up(maxFin, 256), n0), dim3(256), 0, st, d_probs, d_ws);
        // launches per trial slot: linearisation, (control step +) Schur complement, the solver kernels this call mixes, update, control
        const int n_chol = (chol_ldsm ? 1 : 0) + (chol_packed ? 1 : 0) + (chol_mfma ? 1 : 0);
        launches += 3 + (1 + (tiles ? 3 : 2) + std::max(n_chol, 1) + (fold_ctrl ? 1 : 2)) * (long long)slots_done;
    }
    lba_time_end(h);
    OSLAM_HIP_CHECK(hipGetLastError());
    OSLAM_HIP_CHECK(copy_to_host_async(h->out_h, O, outb, st));   // the ONE download
    OSLAM_HIP_CHECK(stream_wait(st));
    lba_time_collect(h, launches);
    return OSLAM_OK;
}

// scatter of window i's outputs (erase flags back in the caller's edge order)
static void lba_fetch(oslam_lba_t* h, int i, float* poses_out, float* points_out, uint8_t* erase, int32_t* stats) {
    const oslam_lba::Prep& q = h->prep[i];
    memcpy(poses_out, h->out_h + q.o_out_poses, (size_t)q.pr.K * 64);
    if (q.pr.P > 0) memcpy(points_out, h->out_h + q.o_out_points, (size_t)q.pr.P * 12);
    const uint8_t* er = h->out_h + q.o_out_erase;
    for (int e = 0; e < q.pr.E; e++) erase[q.order[e]] = er[e];
    if (stats) { const int* st = (const int*)(h->out_h + q.o_out_stats); for (int k = 0; k < 4; k++) stats[k] = st[k]; }
}

extern "C" {

// Host drop-in: the caller flattens the graph (the covisibility gather stays with it).
int oslam_lba_optimize(oslam_lba_t* h, int nKF, const float* poses, const uint8_t* fixed, int nP, const float* points, int nE,
                       const int32_t* edge_kf, const int32_t* edge_pt, const float* edge_obs, const float* edge_invSigma2,
                       const float K5[5], int use_stop_flag, float* poses_out, float* points_out, uint8_t* erase, int32_t stats[4]) {
    if (!h) { set_error("NULL handle"); return OSLAM_E_INVALID; }
    const LbaArgs a = {nKF, poses, fixed, nP, points, nE, edge_kf, edge_pt, edge_obs, edge_invSigma2, poses_out, points_out, erase};
    int rc = lba_prepare_all(h, 1, &a, K5, use_stop_flag, 5, 10, 2, 1, (float)sqrt(5.991), (float)sqrt(7.815));
    if (!rc) rc = lba_launch(h);
    if (!rc) lba_fetch(h, 0, poses_out, points_out, erase, stats);
    return rc;
}

int oslam_ba_optimize(oslam_lba_t* h, int nKF, const float* poses, const uint8_t* fixed, int nP, const float* points, int nE,
                      const int32_t* edge_kf, const int32_t* edge_pt, const float* edge_obs, const float* edge_invSigma2,
                      const float K5[5], int nIterations, int bRobust, int use_stop_flag, float* poses_out, float* points_out) {
    if (!h) { set_error("NULL handle"); return OSLAM_E_INVALID; }
    if (nIterations < 0) { set_error("nIterations < 0"); return OSLAM_E_INVALID; }
    std::vector<uint8_t> erase(nE > 0 ? nE : 1);
    const LbaArgs a = {nKF, poses, fixed, nP, points, nE, edge_kf, edge_pt, edge_obs, edge_invSigma2, poses_out, points_out, erase.data()};
    int rc = lba_prepare_all(h, 1, &a, K5, use_stop_flag, nIterations, 0, 1, bRobust ? 1 : 0, (float)sqrt(5.99), (float)sqrt(7.815));
    if (!rc) rc = lba_launch(h);
    if (!rc) lba_fetch(h, 0, poses_out, points_out, erase.data(), nullptr);
    return rc;
}

// Batch of independent keyframe windows (the batch-of-sequences layout).
int oslam_lba_optimize_batch(oslam_lba_t* h, int n, const oslam_lba_problem_t* probs, const float K5[5]) {
    if (!h || !probs || !K5) { set_error("NULL argument"); return OSLAM_E_INVALID; }
    if (n < 1 || n > h->max_batch) { set_error("batch %d outside [1,%d]", n, h->max_batch); return OSLAM_E_INVALID; }
    std::vector<LbaArgs> a(n);
    for (int i = 0; i < n; i++) {
        const oslam_lba_problem_t& q = probs[i];
        a[i] = {q.nKF, q.poses, q.fixed, q.nP, q.points, q.nE, q.edge_kf, q.edge_pt, q.edge_obs, q.edge_invSigma2, q.poses_out, q.points_out, q.erase};
    }
    // OSLAM_LBA_HOSTPROF=1: wall-clock split of the batch call (preparation / launch incl. the wait for the device / scatter of the results), printed at exit
    struct HostProf {
        std::atomic<long long> ns[4]; std::atomic<long long> calls{0}, windows{0};
        HostProf() { for (auto& x : ns) x = 0; }
        ~HostProf() {
            if (calls.load()) fprintf(stderr, "[lba hostprof] %lld calls, %lld windows: prepare %.1f ms, launch+wait %.1f ms (of it before the upload %.1f ms), fetch %.1f ms\n", calls.load(),
                                      windows.load(), ns[0] * 1e-6, ns[1] * 1e-6, ns[3] * 1e-6, ns[2] * 1e-6);
        }
    };
    static HostProf* prof = getenv("OSLAM_LBA_HOSTPROF") ? new HostProf : nullptr;
    static struct AtExit { ~AtExit() { delete prof; } } at_exit;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t0 = now();
    std::vector<int> wrc;
    int rc = lba_prepare_all(h, n, a.data(), K5, 0, 5, 10, 2, 1, (float)sqrt(5.991), (float)sqrt(7.815), &wrc);
    std::vector<int> good;   // indices of the windows that are solved (empty: all of them)
    if (rc) {
        // A window the solver refuses (more free keyframes than its bound, an index out of range, a duplicate observation ...) fails ALONE when the caller
        // gave it a stats array: stats = {-1, error code, 0, 0}, outputs = inputs, nothing erased — and the other windows of the call are solved.  The batch
        // may carry the windows of a thousand independent sequences (and, through the local-BA service, of several driver handles).
        bool soft = (int)wrc.size() == n;
        for (int i = 0; soft && i < n; i++) if (wrc[i] && !probs[i].stats) soft = false;
        if (!soft) return rc;
        for (int i = 0; i < n; i++) {
            if (!wrc[i]) { good.push_back(i); continue; }
            const oslam_lba_problem_t& q = probs[i];
            if (q.poses_out && q.poses && q.nKF > 0) memcpy(q.poses_out, q.poses, (size_t)q.nKF * 64);
            if (q.points_out && q.points && q.nP > 0) memcpy(q.points_out, q.points, (size_t)q.nP * 12);
            if (q.erase && q.nE > 0) memset(q.erase, 0, (size_t)q.nE);
            q.stats[0] = -1; q.stats[1] = wrc[i]; q.stats[2] = q.stats[3] = 0;
        }
        if (good.empty()) return OSLAM_OK;
        std::vector<LbaArgs> a2(good.size());
        for (size_t j = 0; j < good.size(); j++) a2[j] = a[good[j]];
        a.swap(a2);
        if ((rc = lba_prepare_all(h, (int)a.size(), a.data(), K5, 0, 5, 10, 2, 1, (float)sqrt(5.991), (float)sqrt(7.815)))) return rc;
    }
    auto t1 = now();
    h->prof_pre_upload_ns = 0;
    rc = lba_launch(h);
    if (rc) return rc;
    auto t2 = now();
    for (int j = 0; j < (int)a.size(); j++) {
        const oslam_lba_problem_t& q = probs[good.empty() ? j : good[j]];
        lba_fetch(h, j, q.poses_out, q.points_out, q.erase, q.stats);
    }
    if (prof) {
        auto t3 = now();
        prof->ns[0] += std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
        prof->ns[1] += std::chrono::duration_cast<std::chrono::nanoseconds>(t2 - t1).count();
        prof->ns[2] += std::chrono::duration_cast<std::chrono::nanoseconds>(t3 - t2).count();
        prof->ns[3] += h->prof_pre_upload_ns;
        prof->calls++; prof->windows += n;
    }
    return OSLAM_OK;
}

}  // extern "C"
