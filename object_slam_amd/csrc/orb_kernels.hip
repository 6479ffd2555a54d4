// gfx950 kernels of the ORB extractor (pyramid, FAST+NMS+cell threshold, quad-tree distribution,
// Gaussian blur, orientation + steered BRIEF).  Compiled with -ffp-contract=off: float
// expressions that the reference evaluates on the CPU must round identically here.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/oslam_hip.h"
#include "orb_types.h"

namespace oslam {

struct OrbCtx {
    const OrbParams* P;       // device
    const uint8_t* img0;      // level-0 images (caller's buffer or staging)
    int img0_pitch;
    long long img0_stride;    // bytes between images
    uint8_t* pyr;             // [B][pyr_stride]   levels 1..n-1
    long long pyr_stride;
    uint8_t* blur;            // [B][blur_stride]  levels 0..n-1
    long long blur_stride;
    const int2* rtab;         // resize tables: (src index, a0 | a1 << 16)
    const int* qbase;         // k_resize_words: aligned source byte offset per dst quad
    const uint4* qpx;         // k_resize_words: per pixel (window offset | a0 << 4 | a1 << 16)
    const uint8_t* root_of_x; // per level: region x -> root index
    const short* root_x;      // per level: nIni+1 root boundaries
    int* cell_count;          // [B][total_cells]
    uint32_t* cand;           // [B][cand_per_image]
    const FastCellRec* fast_cells;   // [total_cells] per-cell geometry of k_fast_cells_wave
    int* ovf_count;           // cells whose quick-test worklist did not fit k_fast_cells_wave's LDS part (every-pixel path), cumulative
    int* oct_nodes;           // quad-tree node tables in HBM, [B][nlevels][oct_nodes_stride] (nullptr: they fit the LDS)
    long long oct_nodes_stride;
    uint32_t* ent_g;          // [B][cand_per_image] quad-tree spill (levels with > kCandCap candidates)
    uint16_t* knode_g;        // [B][cand_per_image]
    uint32_t* sel;            // [B][sel_per_image]
    int* sel_count;           // [B][nlevels]
    oslam_keypoint_t* out_kp; // [B][out_cap]
    uint8_t* out_desc;        // [B][out_cap][32]
    int* out_count;           // [B]
    int* status;              // [1]
    unsigned long long* dbg;  // [16] phase cycle counters (profiling builds only)
};

__device__ __forceinline__ const uint8_t* level_image(const OrbCtx& c, const OrbParams* P, int b, int l,
                                                      int& pitch) {
    if (l == 0) {
        pitch = c.img0_pitch;
        return c.img0 + (long long)b * c.img0_stride;
    }
    pitch = P->lv[l].pitch;
    return c.pyr + (long long)b * c.pyr_stride + P->lv[l].img_off;
}

// ------------------------------------------------------------------------------------------
// K1: pyramid level l from level l-1 (cv::resize INTER_LINEAR u8 fixed point, 11-bit taps).
// Replaces reference src/ORBextractor.cc:1107-1132.  4 dst pixels per thread.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resize(OrbCtx c, int level) {
    const OrbParams* P = c.P;
    const LevelGeom& g = P->lv[level];
    const LevelGeom& gs = P->lv[level - 1];
    const int b = blockIdx.z;
    const int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (y >= g.h || x4 >= g.w) return;
    int spitch;
    const uint8_t* src = level_image(c, P, b, level - 1, spitch);
    uint8_t* dst = c.pyr + (long long)b * c.pyr_stride + g.img_off;
    const int2 ty = c.rtab[g.ytab_off + y];
    const int sy0 = min(max(ty.x, 0), gs.h - 1), sy1 = min(max(ty.x + 1, 0), gs.h - 1);
    const int b0 = (short)(ty.y & 0xFFFF), b1 = (short)(ty.y >> 16);
    const uint8_t* S0 = src + (long long)sy0 * spitch;
    const uint8_t* S1 = src + (long long)sy1 * spitch;
    uint32_t outw = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int x = min(x4 + i, g.w - 1);
        const int2 tx = c.rtab[g.xtab_off + x];
        const int sx = tx.x, sx1 = min(sx + 1, gs.w - 1);
        const int a0 = (short)(tx.y & 0xFFFF), a1 = (short)(tx.y >> 16);
        const int r0 = S0[sx] * a0 + S0[sx1] * a1;
        const int r1 = S1[sx] * a0 + S1[sx1] * a1;
        int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
        v = min(max(v, 0), 255);
        outw |= (uint32_t)v << (8 * i);
    }
    // pitch is a multiple of 64 so the 4-byte store is aligned and stays inside the row's padding
    *(uint32_t*)(dst + (long long)y * g.pitch + x4) = outw;
}

// K1 (v2): word-load variant for 4-byte aligned source rows and scale factors <= 2.
// Per quad of 4 dst pixels the host tables give the aligned byte offset of a 12-byte source window (low 2 bits: offset o0 of the first
// pixel's left tap inside it) and, per pixel, the packed taps (a0 | a1 << 16) and a v_perm selector that picks the pixel's two source
// bytes out of the 8 bytes starting at o0 into two 16-bit fields: a row costs two v_alignbyte for the 8-byte window and, per pixel and
// source row, one v_perm + one v_dot2_u32_u16 (the kernel is VALU-issue bound: 40 -> 17 vector instructions per pixel).
constexpr int kResizeRows = 4;   // destination rows per thread: the per-quad x tables (36 B) are loaded once for all of them

typedef unsigned short oslam_u16x2 __attribute__((ext_vector_type(2)));
typedef short oslam_i16x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_resize_words(OrbCtx c, int level) {
    const OrbParams* P = c.P;
    const LevelGeom& g = P->lv[level];
    const LevelGeom& gs = P->lv[level - 1];
    const int b = blockIdx.z;
    const int q = blockIdx.x * 64 + (threadIdx.x & 63);   // quad index
    const int x4 = q * 4;
    const int y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * kResizeRows;
    if (y0 >= g.h || x4 >= g.w) return;
    int spitch;
    const uint8_t* src = level_image(c, P, b, level - 1, spitch);
    uint8_t* dst = c.pyr + (long long)b * c.pyr_stride + g.img_off;
    const int xq = c.qbase[g.qtab_off + q];
    const int xb = xq & ~3;
    const uint32_t o0 = (uint32_t)(xq & 3);
    const uint4 ta = c.qpx[2 * (g.qtab_off + q)], ts = c.qpx[2 * (g.qtab_off + q) + 1];
    const int maxoff = (gs.w - 1) & ~3;
    const int o1 = min(xb + 4, maxoff), o2 = min(xb + 8, maxoff);
    const uint32_t tap[4] = {ta.x, ta.y, ta.z, ta.w}, sel[4] = {ts.x, ts.y, ts.z, ts.w};
    // all source words of the kResizeRows rows are requested before the first one is used
    uint32_t u0[kResizeRows], u1[kResizeRows], u2[kResizeRows], v0[kResizeRows], v1[kResizeRows], v2[kResizeRows];
    int b0[kResizeRows], b1[kResizeRows];
#pragma unroll
    for (int r = 0; r < kResizeRows; r++) {
        const int y = min(y0 + r, g.h - 1);
        const int2 ty = c.rtab[g.ytab_off + y];
        const int sy0 = min(max(ty.x, 0), gs.h - 1), sy1 = min(max(ty.x + 1, 0), gs.h - 1);
        b0[r] = (short)(ty.y & 0xFFFF); b1[r] = (short)(ty.y >> 16);
        const uint8_t* S0 = src + (long long)sy0 * spitch;
        const uint8_t* S1 = src + (long long)sy1 * spitch;
        u0[r] = *(const uint32_t*)(S0 + xb); u1[r] = *(const uint32_t*)(S0 + o1); u2[r] = *(const uint32_t*)(S0 + o2);
        v0[r] = *(const uint32_t*)(S1 + xb); v1[r] = *(const uint32_t*)(S1 + o1); v2[r] = *(const uint32_t*)(S1 + o2);
    }
#pragma unroll
    for (int r = 0; r < kResizeRows; r++) {
        // bytes o0 .. o0+7 of the 12-byte windows of the two source rows
        const uint32_t ul = __builtin_amdgcn_alignbyte(u1[r], u0[r], o0), uh = __builtin_amdgcn_alignbyte(u2[r], u1[r], o0);
        const uint32_t vl = __builtin_amdgcn_alignbyte(v1[r], v0[r], o0), vh = __builtin_amdgcn_alignbyte(v2[r], v1[r], o0);
        uint32_t outw = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const oslam_u16x2 a = __builtin_bit_cast(oslam_u16x2, tap[i]);
            const int r0 = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(oslam_u16x2, __builtin_amdgcn_perm(uh, ul, sel[i])), a, 0u, false);
            const int r1 = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(oslam_u16x2, __builtin_amdgcn_perm(vh, vl, sel[i])), a, 0u, false);
            int v = (((b0[r] * (r0 >> 4)) >> 16) + ((b1[r] * (r1 >> 4)) >> 16) + 2) >> 2;
            v = min(max(v, 0), 255);
            outw |= (uint32_t)v << (8 * i);
        }
        if (y0 + r < g.h) *(uint32_t*)(dst + (long long)(y0 + r) * g.pitch + x4) = outw;
    }
}

// K1 (v3): same arithmetic and tables as k_resize_words, but the source window of a 256 x 16 destination tile is first staged into LDS
// with 16-byte coalesced loads (a dword-per-lane kernel keeps too few bytes in flight per CU: 1.4 TB/s), then every thread cuts its
// 12-byte windows out of LDS.  Requires 16-byte aligned source rows (pyramid levels always; level 0 if the caller's buffer is).
constexpr int kRzTW = 256, kRzTH = 16;           // destination tile
constexpr int kRzLdsPitch = 352, kRzLdsRows = 24;   // source window capacity (scale factors up to ~1.3; checked on the host per level)

__global__ __launch_bounds__(256) void k_resize_lds(OrbCtx c, int level) {
    const OrbParams* P = c.P;
    const LevelGeom& g = P->lv[level];
    const LevelGeom& gs = P->lv[level - 1];
    const int b = blockIdx.z, tid = threadIdx.x;
    __shared__ __align__(16) uint8_t s_src[kRzLdsRows * kRzLdsPitch];
    int spitch;
    const uint8_t* src = level_image(c, P, b, level - 1, spitch);
    uint8_t* dst = c.pyr + (long long)b * c.pyr_stride + g.img_off;
    const int nq = (g.w + 3) >> 2;
    const int q0 = blockIdx.x * (kRzTW / 4), q1 = min(q0 + kRzTW / 4, nq) - 1;      // quads of this tile
    const int ty0 = blockIdx.y * kRzTH, ty1 = min(ty0 + kRzTH, g.h) - 1;             // destination rows of this tile
    const int maxoff = (gs.w - 1) & ~3;
    // source window (wave-uniform): bytes [sx0, sx1) of rows [sy0, sy1]
    const int sx0 = (c.qbase[g.qtab_off + q0] & ~3) & ~15;
    const int sx1 = min((c.qbase[g.qtab_off + q1] & ~3) + 12, maxoff + 4);
    const int sy0 = min(max(c.rtab[g.ytab_off + ty0].x, 0), gs.h - 1);
    const int sy1 = min(max(c.rtab[g.ytab_off + ty1].x + 1, 0), gs.h - 1);
    const int nch = (sx1 - sx0 + 15) >> 4, nrows = sy1 - sy0 + 1;
    // this thread's table entries are requested before the tile so that both travel together
    const int q = min(q0 + (tid & 63), q1);
    const int y0 = ty0 + (tid >> 6) * kResizeRows;
    const int xq = c.qbase[g.qtab_off + q];
    const uint4 ta = c.qpx[2 * (g.qtab_off + q)], ts = c.qpx[2 * (g.qtab_off + q) + 1];
    int2 tyr[kResizeRows];
#pragma unroll
    for (int r = 0; r < kResizeRows; r++) tyr[r] = c.rtab[g.ytab_off + min(y0 + r, g.h - 1)];
    for (int i = tid; i < nch * nrows; i += 256) {
        const int r = i / nch, ch = i - r * nch;
        const uint4 v = *(const uint4*)(src + (long long)(sy0 + r) * spitch + sx0 + ch * 16);
        *(uint4*)(s_src + r * kRzLdsPitch + ch * 16) = v;
    }
    __syncthreads();
    if (q0 + (tid & 63) > q1 || y0 > ty1) return;
    const int x4 = q * 4;
    const int xb = xq & ~3;
    const uint32_t o0 = (uint32_t)(xq & 3);
    const int l0 = xb - sx0, l1 = min(xb + 4, maxoff) - sx0, l2 = min(xb + 8, maxoff) - sx0;   // window words inside the LDS rows
    const uint32_t tap[4] = {ta.x, ta.y, ta.z, ta.w}, sel[4] = {ts.x, ts.y, ts.z, ts.w};
#pragma unroll
    for (int r = 0; r < kResizeRows; r++) {
        const int2 ty = tyr[r];
        const int ra = min(max(ty.x, 0), gs.h - 1) - sy0, rb = min(max(ty.x + 1, 0), gs.h - 1) - sy0;
        const int b0 = (short)(ty.y & 0xFFFF), b1 = (short)(ty.y >> 16);
        const uint8_t* S0 = s_src + ra * kRzLdsPitch;
        const uint8_t* S1 = s_src + rb * kRzLdsPitch;
        const uint32_t u0 = *(const uint32_t*)(S0 + l0), u1 = *(const uint32_t*)(S0 + l1), u2 = *(const uint32_t*)(S0 + l2);
        const uint32_t v0 = *(const uint32_t*)(S1 + l0), v1 = *(const uint32_t*)(S1 + l1), v2 = *(const uint32_t*)(S1 + l2);
        const uint32_t ul = __builtin_amdgcn_alignbyte(u1, u0, o0), uh = __builtin_amdgcn_alignbyte(u2, u1, o0);
        const uint32_t vl = __builtin_amdgcn_alignbyte(v1, v0, o0), vh = __builtin_amdgcn_alignbyte(v2, v1, o0);
        uint32_t outw = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const oslam_u16x2 a = __builtin_bit_cast(oslam_u16x2, tap[i]);
            const int r0 = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(oslam_u16x2, __builtin_amdgcn_perm(uh, ul, sel[i])), a, 0u, false);
            const int r1 = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(oslam_u16x2, __builtin_amdgcn_perm(vh, vl, sel[i])), a, 0u, false);
            int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
            v = min(max(v, 0), 255);
            outw |= (uint32_t)v << (8 * i);
        }
        if (y0 + r <= ty1) *(uint32_t*)(dst + (long long)(y0 + r) * g.pitch + x4) = outw;
    }
}

// ------------------------------------------------------------------------------------------
// K6: 7x7 Gaussian (sigma 2) u8 fixed point, BORDER_REFLECT_101 about the image edge.
// Replaces reference src/ORBextractor.cc:1085-1086.  Tile 64x16 outputs per workgroup.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect101(int p, int len) {
    if (p < 0) p = -p;
    if (p >= len) p = 2 * (len - 1) - p;
    return p;
}

constexpr int kBlurTW = 64, kBlurTH = 16;

__global__ __launch_bounds__(256) void k_blur(OrbCtx c, int level, int sse2_rounding) {
    const OrbParams* P = c.P;
    const LevelGeom& g = P->lv[level];
    const int b = blockIdx.z;
    const int tx0 = blockIdx.x * kBlurTW, ty0 = blockIdx.y * kBlurTH;
    int spitch;
    const uint8_t* src = level_image(c, P, b, level, spitch);
    uint8_t* dst = c.blur + (long long)b * c.blur_stride + g.img_off;

    __shared__ uint8_t tile[kBlurTH + 6][kBlurTW + 8];   // 22 x 72
    __shared__ uint16_t rowp[kBlurTH + 6][kBlurTW];      // horizontal pass (<= 255*257 fits u16)
    const int tid = threadIdx.x;
    for (int i = tid; i < (kBlurTH + 6) * (kBlurTW + 6); i += 256) {
        const int ry = i / (kBlurTW + 6), rx = i - ry * (kBlurTW + 6);
        const int sx = reflect101(min(tx0 + rx - 3, g.w + 2), g.w);
        const int sy = reflect101(min(ty0 + ry - 3, g.h + 2), g.h);
        tile[ry][rx] = src[(long long)sy * spitch + sx];
    }
    __syncthreads();
    int k0 = P->gk[0], k1 = P->gk[1], k2 = P->gk[2], k3 = P->gk[3];
    for (int i = tid; i < (kBlurTH + 6) * kBlurTW; i += 256) {
        const int ry = i >> 6, rx = i & 63;
        const uint8_t* t = &tile[ry][rx];
        int s = k0 * (t[0] + t[6]) + k1 * (t[1] + t[5]) + k2 * (t[2] + t[4]) + k3 * t[3];
        rowp[ry][rx] = (uint16_t)s;
    }
    __syncthreads();
    for (int i = tid; i < kBlurTH * kBlurTW; i += 256) {
        const int oy = i >> 6, ox = i & 63;
        const int x = tx0 + ox, y = ty0 + oy;
        if (x >= g.w || y >= g.h) continue;
        int s = k0 * ((int)rowp[oy][ox] + rowp[oy + 6][ox]) + k1 * ((int)rowp[oy + 1][ox] + rowp[oy + 5][ox]) +
                k2 * ((int)rowp[oy + 2][ox] + rowp[oy + 4][ox]) + k3 * (int)rowp[oy + 3][ox];
        int v;
        if (sse2_rounding && x < (g.w & ~3)) {
            // OpenCV 3.2 SymmColumnVec_32s8u: exact fp32 sum, cvtps2dq (half-to-even)
            int q = s >> 16, r = s & 0xFFFF;
            if (r > 0x8000) q++;
            else if (r == 0x8000) q += (q & 1);
            v = q;
        } else {
            v = (s + (1 << 15)) >> 16;
        }
        dst[(long long)y * g.pitch + x] = (uint8_t)min(max(v, 0), 255);
    }
}

// ------------------------------------------------------------------------------------------
// K2+K3: per-cell FAST-9/16 score, 3x3 NMS inside the cell, ini/min threshold fallback and
// ordered compaction.  One workgroup per (image, level, cell).
// Replaces the per-cell cv::FAST calls of reference src/ORBextractor.cc:789-829.
// score = max over the 16 arcs of 9 of min |v - p| (signed), minus 1: threshold independent,
// so "corner at t" <=> score >= t and one score map serves iniThFAST and minThFAST.
// ------------------------------------------------------------------------------------------
template <int TP>
__device__ __forceinline__ int fast_score(const uint8_t* t) {
    // Bresenham circle r=3, OpenCV order
    constexpr int off[16] = {3 * TP,      3 * TP + 1,  2 * TP + 2,  TP + 3,  3,        -TP + 3, -2 * TP + 2, -3 * TP + 1,
                             -3 * TP,     -3 * TP - 1, -2 * TP - 2, -TP - 3, -3,       TP - 3,  2 * TP - 2,  3 * TP - 1};
    const int v = t[0];
    int d[16];
#pragma unroll
    for (int k = 0; k < 16; k++) d[k] = v - (int)t[off[k]];
    int lo2[16], hi2[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        lo2[k] = min(d[k], d[(k + 1) & 15]);
        hi2[k] = max(d[k], d[(k + 1) & 15]);
    }
    int lo4[16], hi4[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        lo4[k] = min(lo2[k], lo2[(k + 2) & 15]);
        hi4[k] = max(hi2[k], hi2[(k + 2) & 15]);
    }
    int A = -256, Bm = 256;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int lo9 = min(min(lo4[k], lo4[(k + 4) & 15]), d[(k + 8) & 15]);
        const int hi9 = max(max(hi4[k], hi4[(k + 4) & 15]), d[(k + 8) & 15]);
        A = max(A, lo9);
        Bm = min(Bm, hi9);
    }
    const int s = max(A, -Bm) - 1;
    return max(s, 0);
}

// One FAST cell by a whole workgroup: exact scores for every pixel of the cell, NMS, ordered output.  BIG_ONLY: cells wider than one wavefront
// handles (k_fast_cells); otherwise any cell.  All threads of the workgroup must call.
template <bool BIG_ONLY>
__device__ __forceinline__ void fast_cell_workgroup(const OrbCtx& c, const OrbParams* P, const int b, int cell) {
    int level = 0;
    for (int l = 1; l < P->nlevels; l++)
        if (cell >= P->lv[l].cell_base) level = l;
    const LevelGeom& g = P->lv[level];
    if (BIG_ONLY && g.wCell <= 40 && g.hCell <= 40) return;   // handled by k_fast_cells_wave
    cell -= g.cell_base;
    const int ci = cell / g.nCols, cj = cell - ci * g.nCols;
    int* count_out = c.cell_count + (long long)b * P->total_cells + g.cell_base + cell;

    // cell ROI in level coords (reference :789-806)
    const int minBX = kRegionBorder, minBY = kRegionBorder;
    const int maxBX = g.w - kRegionBorder, maxBY = g.h - kRegionBorder;
    const int iniY = minBY + ci * g.hCell, iniX = minBX + cj * g.wCell;
    int maxY = iniY + g.hCell + 6, maxX = iniX + g.wCell + 6;
    const bool skip = (iniY >= maxBY - 3) || (iniX >= maxBX - 6);
    if (maxY > maxBY) maxY = maxBY;
    if (maxX > maxBX) maxX = maxBX;
    const int cw = maxX - iniX - 6, ch = maxY - iniY - 6;   // interior (pixels FAST can report)
    if (skip || cw <= 0 || ch <= 0) {
        if (threadIdx.x == 0) *count_out = 0;
        return;
    }

    __shared__ uint8_t tile[(kMaxCell + 6) * kTilePitch];
    __shared__ uint8_t sc[kMaxCell * kMaxCell];
    __shared__ int wave_tot[4];
    __shared__ int any_ini;
    const int tid = threadIdx.x;
    if (tid == 0) any_ini = 0;

    int pitch;
    const uint8_t* img = level_image(c, P, b, level, pitch);
    const int rw = cw + 6, rh = ch + 6;
    for (int i = tid; i < rw * rh; i += 256) {
        const int ry = i / rw, rx = i - ry * rw;
        tile[ry * kTilePitch + rx] = img[(long long)(iniY + ry) * pitch + iniX + rx];
    }
    __syncthreads();

    // scores: 32 x 8 thread layout
    const int lx = tid & 31, ly = tid >> 5;
    for (int y = ly; y < ch; y += 8)
        for (int x = lx; x < cw; x += 32)
            sc[y * kMaxCell + x] = (uint8_t)fast_score<kTilePitch>(&tile[(y + 3) * kTilePitch + x + 3]);
    __syncthreads();

    // NMS inside the cell: strictly greater than the 8 neighbours, outside-of-cell counts 0
    const int minTh = P->minTh, iniTh = P->iniTh;
    uint32_t keepmask = 0;   // bit i: pixel of iteration i survives NMS with score >= minTh
    int hit_ini = 0;
    {
        int it = 0;
        for (int y = ly; y < ch; y += 8)
            for (int x = lx; x < cw; x += 32, it++) {
                const int s = sc[y * kMaxCell + x];
                bool keep = s >= minTh;
                if (keep) {
#pragma unroll
                    for (int dy = -1; dy <= 1; dy++)
#pragma unroll
                        for (int dx = -1; dx <= 1; dx++) {
                            if (dx == 0 && dy == 0) continue;
                            const int xx = x + dx, yy = y + dy;
                            const int n = (xx >= 0 && xx < cw && yy >= 0 && yy < ch) ? sc[yy * kMaxCell + xx] : 0;
                            keep = keep && (s > n);
                        }
                }
                if (keep) {
                    keepmask |= 1u << it;
                    if (s >= iniTh) hit_ini = 1;
                }
            }
    }
    __syncthreads();   // all NMS reads of sc done
    {
        int it = 0;
        for (int y = ly; y < ch; y += 8)
            for (int x = lx; x < cw; x += 32, it++)
                if (!((keepmask >> it) & 1)) sc[y * kMaxCell + x] = 0;
    }
    if (hit_ini) any_ini = 1;   // benign race: all writers store 1
    __syncthreads();
    const int th = any_ini ? iniTh : minTh;

    // ordered compaction, row-major over the cell interior (reference pushes FAST output in
    // row-major order, :820-825)
    uint32_t* out = c.cand + (long long)b * P->cand_per_image + g.cand_base + (long long)cell * g.cell_cap;
    const int npx = cw * ch;
    const int lane = tid & 63, wv = tid >> 6;
    int running = 0;
    for (int base = 0; base < npx; base += 256) {
        const int p = base + tid;
        int s = 0, x = 0, y = 0;
        if (p < npx) {
            y = p / cw;
            x = p - y * cw;
            s = sc[y * kMaxCell + x];
        }
        const bool flag = s >= th;   // s == 0 for suppressed pixels, th >= 1
        const unsigned long long m = __ballot(flag);
        const int pre = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wv] = __popcll(m);
        __syncthreads();
        int wbase = running;
        for (int i = 0; i < wv; i++) wbase += wave_tot[i];
        if (flag) {
            const int slot = wbase + pre;
            if (slot < g.cell_cap) out[slot] = pack_xys(cj * g.wCell + x + 3, ci * g.hCell + y + 3, s);
        }
        running += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        __syncthreads();
    }
    if (tid == 0) {
        if (running > g.cell_cap) { atomicOr(c.status, 1); running = g.cell_cap; }
        *count_out = running;
    }
}

__global__ __launch_bounds__(256) void k_fast_cells(OrbCtx c) { fast_cell_workgroup<true>(c, c.P, blockIdx.y, blockIdx.x); }

// ------------------------------------------------------------------------------------------
// K6 (v2): 7x7 Gaussian, register sliding window.  One thread produces a 4-pixel-wide, kBlurRows
// tall strip: per input row it loads three aligned u32 words (x-4 .. x+7), forms four horizontal
// sums and keeps the last seven rows of sums in registers; no LDS, no barriers.  All levels of
// all images in one launch (blockIdx.x -> level through OrbParams::blur_block_base).
// ------------------------------------------------------------------------------------------
constexpr int kBlurRows = 8;   // measured at B=256 with the up-front loads: 4 -> 1.49, 6 -> 1.30, 8 -> 1.25, 12 -> 2.3, 16 -> 1.43 us/frame

// horizontal 7-tap sums of 4 adjacent pixels.  Interior: three aligned words, byte windows by
// v_alignbyte, taps by two v_dot4_u32_u8 per pixel (taps 18,34,49,55 fit a byte).
template <bool BORDER>
__device__ __forceinline__ void blur_hsum4(const uint8_t* row, int x4, int w, uint32_t KA, uint32_t KB, int k0, int k1, int k2, int k3,
                                           int out[4]) {
    if (!BORDER) {
        const uint32_t w0 = *(const uint32_t*)(row + x4 - 4), w1 = *(const uint32_t*)(row + x4), w2 = *(const uint32_t*)(row + x4 + 4);
        out[0] = (int)__builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 1), KA, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 1), KB, 0u, false), false);
        out[1] = (int)__builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 2), KA, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 2), KB, 0u, false), false);
        out[2] = (int)__builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 3), KA, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 3), KB, 0u, false), false);
        out[3] = (int)__builtin_amdgcn_udot4(w1, KA, __builtin_amdgcn_udot4(w2, KB, 0u, false), false);
    } else {
        int px[10];
#pragma unroll
        for (int i = 0; i < 10; i++) px[i] = row[reflect101(min(x4 - 3 + i, w + 2), w)];
#pragma unroll
        for (int i = 0; i < 4; i++)
            out[i] = k0 * (px[i] + px[i + 6]) + k1 * (px[i + 1] + px[i + 5]) + k2 * (px[i + 2] + px[i + 4]) + k3 * px[i + 3];
    }
}

// BORDER=false: strips x4 = 4, 8, ... with all taps inside the row (word loads, no divergence).
// BORDER=true : the left strip (x4 = 0) and the 1-2 right strips whose taps reflect (byte loads).
template <bool BORDER>
__global__ __launch_bounds__(256) void k_blur_strip(OrbCtx c, int sse2_rounding) {
    const OrbParams* P = c.P;
    const int* base = BORDER ? P->blurb_block_base : P->blur_block_base;
    int level = 0;
    for (int l = 1; l < P->nlevels; l++)
        if ((int)blockIdx.x >= base[l]) level = l;
    const LevelGeom& g = P->lv[level];
    const int b = blockIdx.y;
    const int bl = blockIdx.x - base[level];
    int spitch;
    const uint8_t* src = level_image(c, P, b, level, spitch);
    const bool aligned = ((spitch & 3) == 0) && ((((unsigned long long)src) & 3ull) == 0);
    // interior strips: x4 = 4 .. last (last + 8 <= w); when rows are not word aligned every strip is "border"
    const int last = aligned ? ((g.w - 8) / 4) * 4 : 0;
    const int nint = aligned && g.w >= 12 ? last / 4 : 0;
    int x4, y0;
    if (!BORDER) {
        const int nbx = (nint + 63) / 64;
        if (nbx == 0) return;
        const int bx = bl % nbx, by = bl / nbx;
        const int si = bx * 64 + (threadIdx.x & 63);
        if (si >= nint) return;
        x4 = 4 + si * 4;
        y0 = (by * 4 + (threadIdx.x >> 6)) * kBlurRows;
    } else {
        // border strips: index 0 -> x4 = 0; index k>=1 -> x4 = first strip after the interior + 4*(k-1)
        const int first_right = nint ? last + 4 : 4;
        const int nright = (g.w - first_right + 3) / 4;
        const int nb = 1 + (nright > 0 ? nright : 0);
        const int t = bl * 256 + threadIdx.x;
        const int sidx = t % nb, rg = t / nb;
        x4 = sidx == 0 ? 0 : first_right + (sidx - 1) * 4;
        y0 = rg * kBlurRows;
    }
    if (x4 >= g.w || y0 >= g.h) return;
    uint8_t* dst = c.blur + (long long)b * c.blur_stride + g.img_off;
    const int k0 = P->gk[0], k1 = P->gk[1], k2 = P->gk[2], k3 = P->gk[3];
    const int wvec = g.w & ~3;
    const uint32_t KA = (uint32_t)k0 | ((uint32_t)k1 << 8) | ((uint32_t)k2 << 16) | ((uint32_t)k3 << 24);
    const uint32_t KB = (uint32_t)k2 | ((uint32_t)k1 << 8) | ((uint32_t)k0 << 16);
    int win[7][4];
    // Interior strips: every source word of the strip (kBlurRows + 6 rows x 3 words) is requested before the first
    // one is used.  Left to itself the compiler issued the three loads of a row right before that row's sums, so a
    // wavefront paid one full memory latency per row (22 per strip) and the SIMDs idled ~85 % of the time.
    uint32_t pw0[BORDER ? 1 : kBlurRows + 6], pw1[BORDER ? 1 : kBlurRows + 6], pw2[BORDER ? 1 : kBlurRows + 6];
    if (!BORDER) {
#pragma unroll
        for (int r = 0; r < kBlurRows + 6; r++) {
            const uint8_t* row = src + (long long)reflect101(min(y0 - 3 + r, g.h + 2), g.h) * spitch;
            pw0[r] = *(const uint32_t*)(row + x4 - 4); pw1[r] = *(const uint32_t*)(row + x4); pw2[r] = *(const uint32_t*)(row + x4 + 4);
        }
    }
    auto hsum_pre = [&](int r, int (&out)[4]) {   // the four horizontal sums of prefetched row r
        const uint32_t w0 = pw0[r], w1 = pw1[r], w2 = pw2[r];
        out[0] = (int)__builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 1), KA, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 1), KB, 0u, false), false);
        out[1] = (int)__builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 2), KA, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 2), KB, 0u, false), false);
        out[2] = (int)__builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 3), KA, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 3), KB, 0u, false), false);
        out[3] = (int)__builtin_amdgcn_udot4(w1, KA, __builtin_amdgcn_udot4(w2, KB, 0u, false), false);
    };
#pragma unroll
    for (int r = 0; r < 6; r++) {
        if (!BORDER) hsum_pre(r, win[r]);
        else blur_hsum4<BORDER>(src + (long long)reflect101(min(y0 - 3 + r, g.h + 2), g.h) * spitch, x4, g.w, KA, KB, k0, k1, k2, k3, win[r]);
    }
#pragma unroll
    for (int iy = 0; iy < kBlurRows; iy++) {
        const int y = y0 + iy;
        // no early exit: rows past the image are computed from clamped/reflected addresses and not stored, so
        // the unrolled loop has no control dependence
        if (!BORDER) hsum_pre(iy + 6, win[6]);
        else blur_hsum4<BORDER>(src + (long long)reflect101(min(y + 3, g.h + 2), g.h) * spitch, x4, g.w, KA, KB, k0, k1, k2, k3, win[6]);
        uint32_t outw = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int s = k0 * (win[0][i] + win[6][i]) + k1 * (win[1][i] + win[5][i]) + k2 * (win[2][i] + win[4][i]) + k3 * win[3][i];
            // s <= 257*65535 < 2^24; half-to-even = (s + 0x7FFF + bit16(s)) >> 16, half-up = (s + 0x8000) >> 16
            const bool rne = sse2_rounding && (BORDER ? (x4 + i) < wvec : true);
            const int v = rne ? (s + 0x7FFF + ((s >> 16) & 1)) >> 16 : (s + 0x8000) >> 16;
            outw |= (uint32_t)min(v, 255) << (8 * i);
        }
        if (y < g.h) *(uint32_t*)(dst + (long long)y * g.pitch + x4) = outw;
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int i = 0; i < 4; i++) win[r][i] = win[r + 1][i];
    }
}

// ------------------------------------------------------------------------------------------
// K2+K3 (v2): one WAVEFRONT per FAST cell (cells up to 40x40 interior): no workgroup barriers.
// Tile rows are fetched as aligned u32 words, scores use min3/max3 trees
// (arc minimum of 9 = min3 of three min3-of-3), the ini/min threshold vote and the ordered
// compaction are ballots.  Larger cells fall back to k_fast_cells.
// ------------------------------------------------------------------------------------------
constexpr int kWCell = 40;            // max interior edge handled per wavefront
constexpr int kWTileP = 52;           // tile pitch in bytes (>= kWCell + 6 + 3 alignment slack, multiple of 4)
constexpr int kWTileRows = kWCell + 6;
// Worklist of pixels that pass the quick test: kFastWorkLds entries in LDS (a workgroup then needs 23 KB instead of 32 KB: 7 instead of 5
// workgroups per CU; 3.33 -> 2.82 us/frame beside the blur).  A cell with more survivors takes the every-pixel path of the same wavefront.
constexpr int kFastWorkLds = 880;
constexpr int kScP = 40;              // pitch of the score array (worklist entries keep the y * 64 + x encoding)
// Score array of one wavefront: rows -1 .. kWCell of kScP bytes behind 4 leading bytes, all zeroed before the quick test, so the NMS reads the 8
// neighbours of a pixel without bounds checks when the cell is narrower than the pitch (column cw of row y and column -1 of row y + 1 are then zeros).
constexpr int kScBytes = 4 + (kWCell + 2) * kScP;

template <int TP>
__device__ __forceinline__ int fast_score3(const uint8_t* t) {
    constexpr int off[16] = {3 * TP,      3 * TP + 1,  2 * TP + 2,  TP + 3,  3,        -TP + 3, -2 * TP + 2, -3 * TP + 1,
                             -3 * TP,     -3 * TP - 1, -2 * TP - 2, -TP - 3, -3,       TP - 3,  2 * TP - 2,  3 * TP - 1};
    // min over an arc of (v - p) = v - max over the arc of p: the trees run on the raw circle pixels, v enters once at the end
    const int v = t[0];
    int p[16];
#pragma unroll
    for (int k = 0; k < 16; k++) p[k] = (int)t[off[k]];
    int mx3[16], mn3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        mx3[k] = max(max(p[k], p[(k + 1) & 15]), p[(k + 2) & 15]);
        mn3[k] = min(min(p[k], p[(k + 1) & 15]), p[(k + 2) & 15]);
    }
    int minmax = 255, maxmin = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int mx9 = max(max(mx3[k], mx3[(k + 3) & 15]), mx3[(k + 6) & 15]);
        const int mn9 = min(min(mn3[k], mn3[(k + 3) & 15]), mn3[(k + 6) & 15]);
        minmax = min(minmax, mx9);
        maxmin = max(maxmin, mn9);
    }
    return max(max(v - minmax, maxmin - v) - 1, 0);
}

#ifdef OSLAM_FAST_PROFILE
#define FSTAMP(i) do { const long long t_ = clock64(); if (lane == 0) atomicAdd(&c.dbg[i], (unsigned long long)(t_ - tl_)); tl_ = clock64(); } while (0)
#else
#define FSTAMP(i) do { } while (0)
#endif

// Geometry of one FAST cell as a wavefront sees it: everything is wave-uniform and comes from the cell's FastCellRec through one scalar load.
struct FastCell {
    int valid;             // 0: nothing to do (out of range / handled by k_fast_cells); 1: process; 2: empty cell, count = 0
    int cw, ch, iniX, iniY, rw, rh, pitch, gbase, gsh, nwt, aligned, cell_cap;
    uint32_t cand_ofs;
    const uint8_t* img;
};

__device__ __forceinline__ void fast_cell_geom(const OrbCtx& c, int cell_end, int b, int cell /* wave-uniform */, FastCell& G) {
    G.valid = 0;
    if (cell >= cell_end) return;
    const FastCellRec r = c.fast_cells[cell];
    G.valid = (int)(r.dims >> 24);
    G.cand_ofs = r.cand_ofs; G.cell_cap = (int)r.cell_cap;
    if (G.valid != 1) return;
    G.iniX = (int)(r.xy & 0xFFFFu); G.iniY = (int)(r.xy >> 16);
    G.cw = (int)(r.dims & 0xFFu); G.ch = (int)((r.dims >> 8) & 0xFFu);
    if (((r.dims >> 16) & 0xFFu) == 0u) { G.pitch = c.img0_pitch; G.img = c.img0 + (long long)b * c.img0_stride; }
    else { G.pitch = (int)r.pitch; G.img = c.pyr + (long long)b * c.pyr_stride + r.img_off; }
    G.rw = G.cw + 6; G.rh = G.ch + 6;
    // Tile layout: ROI column cc (0..rw-1) at tile byte cc+1, i.e. interior pixel x at byte x+4 (word aligned
    // for x % 4 == 0).  Tile word j = global bytes base+s+4j.., base = (iniX-1) & ~3, s = (iniX-1) & 3
    // (iniX >= 16, so the window never starts before the row; it ends before column w).
    G.gbase = (G.iniX - 1) & ~3; G.gsh = (G.iniX - 1) & 3;
    G.nwt = (G.rw + 1 + 3) >> 2;   // tile words per row (<= 12)
    G.aligned = ((G.pitch & 3) == 0) && ((((unsigned long long)G.img) & 3ull) == 0);
}

constexpr int kFastCellsPerWave = 1;   // cells per wavefront (measured: 1 -> 2.64 us/frame, 2 -> 3.2, 4 -> 3.15 at B=64: the kernel is VALU bound, fewer and longer waves only add tail)

// Cells [cell_begin, cell_end) of every image: level 0's cells need only the caller's image and are launched beside the pyramid kernels.
__global__ __launch_bounds__(256) void k_fast_cells_wave(OrbCtx c, int cell_begin, int cell_end) {
    const OrbParams* P = c.P;
#ifdef OSLAM_FAST_PROFILE
    long long tl_ = clock64();
#endif
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // the cell geometry lives in scalar registers
    const int cell_first = cell_begin + (blockIdx.x * 4 + wv) * kFastCellsPerWave;
    __shared__ __align__(4) uint8_t s_tile[4][kWTileRows * kWTileP];
    __shared__ __align__(4) uint8_t s_sc[4][kScBytes];   // scores, zero ring around the cell (see kScBytes)
    __shared__ __align__(8) uint16_t s_work[4][kFastWorkLds];      // worklist: y*64 + x of pixels passing the quick test, row-major
    static_assert((kFastWorkLds * 2) % 8 == 0 && kWCell * 8 <= kFastWorkLds * 2, "the overflow path parks 8-byte row masks in the worklist");
    uint8_t* tile = s_tile[wv];
    uint8_t* sc = s_sc[wv] + kScP + 4;   // sc[y * kScP + x], rows -1 .. kWCell and column -1 exist
    uint16_t* work = s_work[wv];
    const int minTh = P->minTh, iniTh = P->iniTh;
    // lanes 0..15 = word in row, lane >> 4 = row inside a group of 4: no divisions; neighbour word by a row shift
    const int wx = lane & 15, rsub = lane >> 4;
    constexpr int kLoadIters = (kWTileRows + 3) / 4;
    uint32_t gw[kLoadIters];                           // raw aligned words of the NEXT cell's tile, in flight
    FastCell N;                                        // geometry of the next cell
    auto issue_loads = [&](const FastCell& G) {
        const uint32_t lane_off = (uint32_t)(rsub * G.pitch + wx * 4);
        const bool col_ok = G.valid == 1 && G.aligned && wx <= G.nwt;
#pragma unroll
        for (int k = 0; k < kLoadIters; k++) {
            gw[k] = 0;
            if (col_ok && 4 * k + rsub < G.rh) gw[k] = *(const uint32_t*)(G.img + ((long long)(G.iniY + 4 * k) * G.pitch + G.gbase) + lane_off);
        }
    };
    fast_cell_geom(c, cell_end, b, cell_first, N);
    issue_loads(N);
    for (int jc = 0; jc < kFastCellsPerWave; jc++) {
    const FastCell G = N;
    const int cell = cell_first + jc;
    if (G.valid == 0) { if (jc + 1 < kFastCellsPerWave) { fast_cell_geom(c, cell_end, b, cell_first + jc + 1, N); issue_loads(N); } continue; }
    const int cw = G.cw, ch = G.ch;
    int* count_out = c.cell_count + (long long)b * P->total_cells + cell;
    uint32_t* out = c.cand + (long long)b * P->cand_per_image + G.cand_ofs;
    const int cell_cap = G.cell_cap;
    const int xout0 = G.iniX - kRegionBorder + 3, yout0 = G.iniY - kRegionBorder + 3;   // cj * wCell + 3, ci * hCell + 3
    if (G.valid == 2) {
        if (lane == 0) *count_out = 0;
        if (jc + 1 < kFastCellsPerWave) { fast_cell_geom(c, cell_end, b, cell_first + jc + 1, N); issue_loads(N); }
        continue;
    }
    const int rw = G.rw, rh = G.rh;
    __builtin_amdgcn_wave_barrier();                   // the previous cell no longer reads the LDS tile
    for (int i = lane; i < kScBytes / 4; i += 64) ((uint32_t*)s_sc[wv])[i] = 0u;   // while the tile words travel
    if (G.aligned) {
#pragma unroll
        for (int k = 0; k < kLoadIters; k++) {
            const int ry = 4 * k + rsub;
            const uint32_t gn = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)gw[k], 0x101 /* row_shl:1: lane i reads lane i + 1 of its 16-lane row */, 0xf, 0xf, true);
            if (ry < rh && wx < G.nwt) *(uint32_t*)(tile + ry * kWTileP + wx * 4) = __builtin_amdgcn_alignbyte(gn, gw[k], G.gsh);
        }
    } else {   // caller's level-0 buffer with an unaligned pitch: byte loads
        for (int i = lane; i < rw * rh; i += 64) {
            const int ry = i / rw, rx = i - ry * rw;
            tile[ry * kWTileP + 1 + rx] = G.img[(long long)(G.iniY + ry) * G.pitch + G.iniX + rx];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // the next cell's tile words travel while this cell is processed
    if (jc + 1 < kFastCellsPerWave) { fast_cell_geom(c, cell_end, b, cell_first + jc + 1, N); issue_loads(N); }
    else N.valid = 0;
    FSTAMP(0);
    const uint8_t* t0 = tile + 3 * kWTileP + 4;   // t0[y*kWTileP + x] = interior pixel (x, y)

    // A. quick rejection at minTh: an arc of 9 of 16 contains one pixel of every opposite pair, so a
    //    corner needs (p0|p8) and (p4|p12) brighter than v+t, or both darker than v-t.  Survivors ->
    //    ordered worklist.
    int nwork = 0;
    if (cw <= 32) {
        // 4 pixels per lane (one aligned word), 8 rows per pass; two pixels per register in 16-bit fields (`v_perm` extraction, packed
        // 16-bit min / max / sub): brighter needs min(max(p0,p8), max(p4,p12)) > v + t, darker max(min(p0,p8), min(p4,p12)) < v - t.
        // Flags of the word's pixels 0..3 sit at bits 0, 1, 16, 17 of f.
        const int x4 = (lane & 7) * 4;
        const oslam_i16x2 tt = {(short)minTh, (short)minTh};
        const int nvalid = cw - x4;   // pixels of this lane's word inside the cell
        const uint32_t fmask = nvalid >= 4 ? 0x00030003u : nvalid == 3 ? 0x00010003u : nvalid == 2 ? 0x00000003u : nvalid == 1 ? 1u : 0u;
        for (int yb = 0; yb < ch; yb += 8) {
            const int y = yb + (lane >> 3);
            uint32_t f = 0;
            if (y < ch) {
                const uint8_t* tr = t0 + y * kWTileP + x4;
                const uint32_t wv4 = *(const uint32_t*)tr;
                const uint32_t wl = *(const uint32_t*)(tr - 4), wr = *(const uint32_t*)(tr + 4);
                const uint32_t w0 = *(const uint32_t*)(tr + 3 * kWTileP), w8 = *(const uint32_t*)(tr - 3 * kWTileP);
                const uint32_t w4 = __builtin_amdgcn_alignbyte(wr, wv4, 3), w12 = __builtin_amdgcn_alignbyte(wv4, wl, 1);
                uint32_t pass[2];
#pragma unroll
                for (int par = 0; par < 2; par++) {
                    const uint32_t sel = par ? 0x0c030c01u : 0x0c020c00u;
                    const oslam_i16x2 v2 = __builtin_bit_cast(oslam_i16x2, __builtin_amdgcn_perm(0u, wv4, sel));
                    const oslam_i16x2 a0 = __builtin_bit_cast(oslam_i16x2, __builtin_amdgcn_perm(0u, w0, sel));
                    const oslam_i16x2 a8 = __builtin_bit_cast(oslam_i16x2, __builtin_amdgcn_perm(0u, w8, sel));
                    const oslam_i16x2 a4 = __builtin_bit_cast(oslam_i16x2, __builtin_amdgcn_perm(0u, w4, sel));
                    const oslam_i16x2 a12 = __builtin_bit_cast(oslam_i16x2, __builtin_amdgcn_perm(0u, w12, sel));
                    const oslam_i16x2 mB = __builtin_elementwise_min(__builtin_elementwise_max(a0, a8), __builtin_elementwise_max(a4, a12));
                    const oslam_i16x2 mD = __builtin_elementwise_max(__builtin_elementwise_min(a0, a8), __builtin_elementwise_min(a4, a12));
                    const oslam_i16x2 s1 = (v2 + tt) - mB;      // negative iff mB > v + t
                    const oslam_i16x2 s2 = mD - (v2 - tt);      // negative iff mD < v - t
                    pass[par] = __builtin_bit_cast(uint32_t, s1) | __builtin_bit_cast(uint32_t, s2);
                }
                f = (((pass[0] >> 15) & 0x00010001u) | ((pass[1] >> 14) & 0x00020002u)) & fmask;
            }
            const unsigned long long m0 = __ballot(f & 1u), m1 = __ballot(f & 2u), m2 = __ballot(f & 0x10000u), m3 = __ballot(f & 0x20000u);
            int pos = nwork;
            pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, pos));
            pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, pos));
            pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m2, pos));
            pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(m3 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m3, pos));
            const int q0 = y * 64 + x4;
            if (f & 1u) { if (pos < kFastWorkLds) work[pos] = (uint16_t)q0; pos++; }
            if (f & 2u) { if (pos < kFastWorkLds) work[pos] = (uint16_t)(q0 + 1); pos++; }
            if (f & 0x10000u) { if (pos < kFastWorkLds) work[pos] = (uint16_t)(q0 + 2); pos++; }
            if (f & 0x20000u) { if (pos < kFastWorkLds) work[pos] = (uint16_t)(q0 + 3); pos++; }
            nwork += __popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3);
        }
    } else {
        const int npx = cw * ch;
        for (int base = 0; base < npx; base += 64) {
            const int p = base + lane;
            bool pass = false;
            int x = 0, y = 0;
            if (p < npx) {
                y = p / cw; x = p - y * cw;
                const uint8_t* t = t0 + y * kWTileP + x;
                const int v = t[0];
                const int hi = v + minTh, lo = v - minTh;
                const int p0 = t[3 * kWTileP], p4 = t[3], p8 = t[-3 * kWTileP], p12 = t[-3];
                pass = (((p0 > hi) || (p8 > hi)) && ((p4 > hi) || (p12 > hi))) || (((p0 < lo) || (p8 < lo)) && ((p4 < lo) || (p12 < lo)));
            }
            const unsigned long long m = __ballot(pass);
            if (pass) { const int pos = nwork + __popcll(m & ((1ull << lane) - 1ull)); if (pos < kFastWorkLds) work[pos] = (uint16_t)(y * 64 + x); }
            nwork += __popcll(m);
        }
    }
    if (nwork > kFastWorkLds) {
        // (wave-uniform, rare) more survivors than the LDS worklist holds: exact scores for every pixel of the cell, one row of
        // the cell per pass (lane = column), then the same NMS / threshold vote / row-major output from per-row keep masks that
        // are parked in the worklist's LDS.  "corner at t" <=> score >= t, so skipping the quick test changes nothing.
        __builtin_amdgcn_wave_barrier();
        for (int y = 0; y < ch; y++)
            if (lane < cw) sc[y * kScP + lane] = (uint8_t)fast_score3<kWTileP>(t0 + y * kWTileP + lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        unsigned long long* rowmask = (unsigned long long*)s_work[wv];   // ch <= kWCell masks of 8 bytes: 320 B of the 1760 B worklist (8-byte aligned: see the static_assert)
        bool hit = false;
        for (int y = 0; y < ch; y++) {
            bool keep = false;
            if (lane < cw) {
                const int sv = sc[y * kScP + lane];
                keep = sv >= minTh;
#pragma unroll
                for (int dy = -1; dy <= 1; dy++)
#pragma unroll
                    for (int dx = -1; dx <= 1; dx++) {
                        if (dx == 0 && dy == 0) continue;
                        const int xx = lane + dx, yy = y + dy;
                        const int nb = (xx >= 0 && xx < cw && yy >= 0 && yy < ch) ? sc[yy * kScP + xx] : 0;
                        keep = keep && (sv > nb);
                    }
                hit = hit || (keep && sv >= iniTh);
            }
            const unsigned long long km = __ballot(keep);
            if (lane == 0) rowmask[y] = km;
        }
        const int th_s = __any(hit) ? iniTh : minTh;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int run = 0;
        for (int y = 0; y < ch; y++) {
            const unsigned long long km = rowmask[y];
            const int sv = lane < cw ? sc[y * kScP + lane] : 0;
            const bool flag = ((km >> lane) & 1ull) && sv >= th_s;
            const unsigned long long m = __ballot(flag);
            if (flag) {
                const int slot = run + __popcll(m & ((1ull << lane) - 1ull));
                if (slot < cell_cap) out[slot] = pack_xys(xout0 + lane, yout0 + y, sv);
            }
            run += __popcll(m);
        }
        if (lane == 0) {
            if (run > cell_cap) { atomicOr(c.status, 1); run = cell_cap; }
            *count_out = run;
            atomicAdd(c.ovf_count, 1);
        }
        if (jc + 1 < kFastCellsPerWave) { fast_cell_geom(c, cell_end, b, cell_first + jc + 1, N); issue_loads(N); }
        continue;
    }
    __builtin_amdgcn_wave_barrier();
    FSTAMP(1);
    // B. exact scores of the survivors (others stay 0 = "not a corner at minTh")
    for (int i = lane; i < nwork; i += 64) {
        const int q = work[i], y = q >> 6, x = q & 63;
        sc[y * kScP + x] = (uint8_t)fast_score3<kWTileP>(t0 + y * kWTileP + x);
    }
    __builtin_amdgcn_wave_barrier();
    FSTAMP(2);
    // C. NMS over the worklist (strictly greater than the 8 in-cell neighbours), ini/min vote
    unsigned long long keepmask = 0;   // bit it: this lane's worklist entry of pass `it` survives
    bool hit_ini = false;
    if (cw < kScP) {   // zero ring: no bounds checks
        int it = 0;
        for (int i = lane; i < nwork; i += 64, it++) {
            const int q = work[i], y = q >> 6, x = q & 63;
            const uint8_t* n = sc + (y - 1) * kScP + x - 1;
            const int s = n[kScP + 1];
            const int m = max(max(max((int)n[0], (int)n[1]), max((int)n[2], (int)n[kScP])),
                              max(max((int)n[kScP + 2], (int)n[2 * kScP]), max((int)n[2 * kScP + 1], (int)n[2 * kScP + 2])));
            if (s >= minTh && s > m) { keepmask |= 1ull << it; hit_ini = hit_ini || (s >= iniTh); }
        }
    } else {
        int it = 0;
        for (int i = lane; i < nwork; i += 64, it++) {
            const int q = work[i], y = q >> 6, x = q & 63;
            const int s = sc[y * kScP + x];
            bool keep = s >= minTh;
            if (keep) {
#pragma unroll
                for (int dy = -1; dy <= 1; dy++)
#pragma unroll
                    for (int dx = -1; dx <= 1; dx++) {
                        if (dx == 0 && dy == 0) continue;
                        const int xx = x + dx, yy = y + dy;
                        const int n = (xx >= 0 && xx < cw && yy >= 0 && yy < ch) ? sc[yy * kScP + xx] : 0;
                        keep = keep && (s > n);
                    }
            }
            if (keep) { keepmask |= 1ull << it; hit_ini = hit_ini || (s >= iniTh); }
        }
    }
    const int th = __any(hit_ini) ? iniTh : minTh;
    FSTAMP(3);
    // D. ordered output (worklist order is row-major)
    int running = 0;
    {
        int it = 0;
        for (int base = 0; base < nwork; base += 64, it++) {
            const int i = base + lane;
            bool flag = false;
            int s = 0, x = 0, y = 0;
            if (i < nwork && ((keepmask >> it) & 1ull)) {
                const int q = work[i];
                y = q >> 6; x = q & 63;
                s = sc[y * kScP + x];
                flag = s >= th;
            }
            const unsigned long long m = __ballot(flag);
            if (flag) {
                const int slot = running + __popcll(m & ((1ull << lane) - 1ull));
                if (slot < cell_cap) out[slot] = pack_xys(xout0 + x, yout0 + y, s);
            }
            running += __popcll(m);
        }
    }
    if (lane == 0) {
        if (running > cell_cap) { atomicOr(c.status, 1); running = cell_cap; }
        *count_out = running;
    }
    FSTAMP(4);
#ifdef OSLAM_FAST_PROFILE
    if (lane == 0) { atomicAdd(&c.dbg[5], (unsigned long long)nwork); atomicAdd(&c.dbg[6], 1ull); atomicAdd(&c.dbg[7], (unsigned long long)(cw * ch)); }
#endif
    }   // cells of this wavefront
}

// ------------------------------------------------------------------------------------------
// K4: quad-tree distribution (reference DistributeOctTree, src/ORBextractor.cc:539-763), one
// workgroup per (image, level).  The std::list algorithm is restated over arrays kept in list
// order: each pass is described by the set of "processed" nodes and their processing order;
// children of processed nodes go to the front in reverse processing order (push_front), all
// other nodes keep their relative order behind them.  Keypoints carry the list position of
// their node; the per-node survivor is the first maximum response in candidate order.
// Tie-break of the (size, pointer) sort (:684) is normalised to node creation order.
// ------------------------------------------------------------------------------------------
constexpr int kOctThreads = 512;

// exclusive scan of a[0..n) in place; returns the total.  All threads must call.
__device__ int block_exclusive_scan(int* a, int n, int* scratch /* >= 16 ints */) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int per = (n + kOctThreads - 1) / kOctThreads;
    const int beg = min(tid * per, n), end = min(beg + per, n);
    int local = 0;
    for (int i = beg; i < end; i++) local += a[i];
    int incl = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    if (lane == 63) scratch[wv] = incl;
    __syncthreads();
    int wbase = 0, total = 0;
    for (int i = 0; i < kOctThreads / 64; i++) {
        if (i < wv) wbase += scratch[i];
        total += scratch[i];
    }
    int run = wbase + incl - local;
    for (int i = beg; i < end; i++) {
        const int v = a[i];
        a[i] = run;
        run += v;
    }
    __syncthreads();
    return total;
}

struct NodeTab {
    short* x0; short* x1; short* y0; short* y1;
    int* cnt;
    int* seq;
};

__device__ __forceinline__ int child_of(int kx, int ky, int x0, int x1, int y0, int y1) {
    const int halfX = (x1 - x0 + 1) >> 1;   // ceil((x1-x0)/2), :483
    const int halfY = (y1 - y0 + 1) >> 1;
    const int cx = kx < x0 + halfX ? 0 : 1;
    const int cy = ky < y0 + halfY ? 0 : 1;
    return cx + 2 * cy;   // n1=0 (UL), n2=1 (UR), n3=2 (BL), n4=3 (BR)
}

constexpr int kIdxBits = 20;   // candidate index bits in the survivor key (score << 20 | ~index)

template <typename EntT, typename NodeT>
__device__ __forceinline__ void octree_body(const OrbCtx& c, EntT ent, NodeT knode, int* ibase, const int* cellofs,
                                            const int n) {
    const OrbParams* P = c.P;
    const int level = blockIdx.x, b = blockIdx.y;
    const LevelGeom& g = P->lv[level];
    const int tid = threadIdx.x;
    const int NC = P->node_cap;
    const int N = g.quota;
    int* childcnt = ibase;            ibase += 4 * NC;     // [4*NC]
    int* childpos = ibase;            ibase += 4 * NC;     // [4*NC] new list position of child
    int* staypos = ibase;             ibase += NC;         // new list position of an unprocessed node
    int* prank = ibase;               ibase += NC;         // processing rank or -1
    int* byrank = ibase;              ibase += NC;         // rank -> node
    int* scanA = ibase;               ibase += NC + 1;
    int* scanB = ibase;               ibase += NC + 1;
    int* cntA = ibase;                ibase += NC;
    int* cntB = ibase;                ibase += NC;
    int* seqA = ibase;                ibase += NC;
    int* seqB = ibase;                ibase += NC;
    int* best = ibase;                ibase += NC;
    int* scratch = ibase;             ibase += 32;
    short* sbase = (short*)ibase;
    short* bx0A = sbase; sbase += NC; short* bx1A = sbase; sbase += NC;
    short* by0A = sbase; sbase += NC; short* by1A = sbase; sbase += NC;
    short* bx0B = sbase; sbase += NC; short* bx1B = sbase; sbase += NC;
    short* by0B = sbase; sbase += NC; short* by1B = sbase; sbase += NC;
    NodeTab cur = {bx0A, bx1A, by0A, by1A, cntA, seqA};
    NodeTab nxt = {bx0B, bx1B, by0B, by1B, cntB, seqB};
    __shared__ int sh_flag, sh_J;

    uint32_t* sel = c.sel + (long long)b * P->sel_per_image + g.sel_base;
    int* sel_count = c.sel_count + (long long)b * P->nlevels + level;

    // ---- gather the level's candidates in reference order: cells row-major, pixels row-major ----
    const int ncells = g.nCols * g.nRows;
    const int* cell_count = c.cell_count + (long long)b * P->total_cells + g.cell_base;
    const uint32_t* cand = c.cand + (long long)b * P->cand_per_image + g.cand_base;
    for (int ce = tid >> 4; ce < ncells; ce += kOctThreads / 16) {
        const int cnt = cell_count[ce], o = cellofs[ce];
        for (int i = tid & 15; i < cnt; i += 16) ent[o + i] = cand[(long long)ce * g.cell_cap + i];
    }
    __syncthreads();

    // ---- roots (:543-585) ----
    const int nIni = g.nIni;
    const short* rootx = c.root_x + g.root_off;                        // nIni+1 boundaries
    const uint8_t* root_of_x = c.root_of_x + (long long)level * 4096;  // region x -> root, (int)(x/hX) :569
    for (int i = tid; i < nIni; i += kOctThreads) childcnt[i] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += kOctThreads) {
        const int r = root_of_x[ent_x(ent[i])];
        knode[i] = (uint16_t)r;
        atomicAdd(&childcnt[r], 1);
    }
    __syncthreads();
    for (int i = tid; i < nIni; i += kOctThreads) scanA[i] = childcnt[i] > 0 ? 1 : 0;
    __syncthreads();
    int S = block_exclusive_scan(scanA, nIni, scratch);
    for (int i = tid; i < nIni; i += kOctThreads) {
        if (childcnt[i] > 0) {
            const int p = scanA[i];
            cur.x0[p] = rootx[i]; cur.x1[p] = rootx[i + 1];
            cur.y0[p] = 0; cur.y1[p] = (short)g.region_h;
            cur.cnt[p] = childcnt[i]; cur.seq[p] = i;
            staypos[i] = p;
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += kOctThreads) knode[i] = (uint16_t)staypos[knode[i]];
    __syncthreads();

    // ---- passes ----
    bool careful = false;
    for (int iter = 0; iter < 64; iter++) {
        // child occupancy of every divisible node
        for (int i = tid; i < 4 * S; i += kOctThreads) childcnt[i] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += kOctThreads) {
            const int p = knode[i];
            if (cur.cnt[p] > 1) {
                const uint32_t e = ent[i];
                const int ch = child_of(ent_x(e), ent_y(e), cur.x0[p], cur.x1[p], cur.y0[p], cur.y1[p]);
                atomicAdd(&childcnt[4 * p + ch], 1);
            }
        }
        __syncthreads();

        // processing order
        int M;   // number of processed nodes
        if (!careful) {
            for (int p = tid; p < S; p += kOctThreads) scanA[p] = cur.cnt[p] > 1 ? 1 : 0;
            __syncthreads();
            M = block_exclusive_scan(scanA, S, scratch);
            for (int p = tid; p < S; p += kOctThreads) {
                if (cur.cnt[p] > 1) { prank[p] = scanA[p]; byrank[scanA[p]] = p; }
                else prank[p] = -1;
            }
            __syncthreads();
        } else {
            // sort divisible nodes by (size, creation seq) descending (:684-686)
            for (int p = tid; p < S; p += kOctThreads) {
                int r = -1;
                const int cp = cur.cnt[p];
                if (cp > 1) {
                    const int sp = cur.seq[p];
                    r = 0;
                    for (int q = 0; q < S; q++) {
                        const int cq = cur.cnt[q];
                        if (cq > 1 && (cq > cp || (cq == cp && cur.seq[q] > sp))) r++;
                    }
                    byrank[r] = p;
                }
                prank[p] = r;
            }
            if (tid == 0) sh_J = 0x7fffffff;
            for (int p = tid; p < S; p += kOctThreads) scanA[p] = cur.cnt[p] > 1 ? 1 : 0;
            __syncthreads();
            const int V = block_exclusive_scan(scanA, S, scratch);
            // list growth per processed node, in rank order; stop at the first size >= N (:729-730)
            for (int j = tid; j < V; j += kOctThreads) {
                const int p = byrank[j];
                const int k = (childcnt[4 * p] > 0) + (childcnt[4 * p + 1] > 0) + (childcnt[4 * p + 2] > 0) +
                              (childcnt[4 * p + 3] > 0);
                scanB[j] = k - 1;
            }
            __syncthreads();
            block_exclusive_scan(scanB, V, scratch);
            for (int j = tid; j < V; j += kOctThreads) {
                const int p = byrank[j];
                const int k = (childcnt[4 * p] > 0) + (childcnt[4 * p + 1] > 0) + (childcnt[4 * p + 2] > 0) +
                              (childcnt[4 * p + 3] > 0);
                if (S + scanB[j] + k - 1 >= N) atomicMin(&sh_J, j);
            }
            __syncthreads();
            M = min(V, sh_J == 0x7fffffff ? V : sh_J + 1);
            for (int p = tid; p < S; p += kOctThreads)
                if (prank[p] >= M) prank[p] = -1;
            __syncthreads();
        }

        // children of processed nodes: positions at the front, reverse processing order
        for (int j = tid; j < M; j += kOctThreads) {
            const int p = byrank[j];
            scanA[j] = (childcnt[4 * p] > 0) + (childcnt[4 * p + 1] > 0) + (childcnt[4 * p + 2] > 0) +
                       (childcnt[4 * p + 3] > 0);
        }
        __syncthreads();
        const int Ktot = block_exclusive_scan(scanA, M, scratch);   // scanA[j] = children before rank j
        for (int p = tid; p < S; p += kOctThreads) scanB[p] = prank[p] < 0 ? 1 : 0;
        __syncthreads();
        const int R = block_exclusive_scan(scanB, S, scratch);      // scanB[p] = unprocessed before p
        const int newS = Ktot + R;
        if (newS > NC) {   // cannot happen for quota+3 sized tables; fail loudly
            if (tid == 0) { atomicOr(c.status, 4); *sel_count = 0; }
            return;
        }
        if (tid == 0) sh_flag = 0;
        __syncthreads();
        int nexp_local = 0;
        for (int p = tid; p < S; p += kOctThreads) {
            const int j = prank[p];
            if (j < 0) {
                const int pos = Ktot + scanB[p];
                staypos[p] = pos;
                nxt.x0[pos] = cur.x0[p]; nxt.x1[pos] = cur.x1[p];
                nxt.y0[pos] = cur.y0[p]; nxt.y1[pos] = cur.y1[p];
                nxt.cnt[pos] = cur.cnt[p]; nxt.seq[pos] = cur.seq[p];
            } else {
                const int k = (childcnt[4 * p] > 0) + (childcnt[4 * p + 1] > 0) + (childcnt[4 * p + 2] > 0) +
                              (childcnt[4 * p + 3] > 0);
                const int basepos = Ktot - scanA[j] - k;   // children of later-processed nodes come first
                const int x0 = cur.x0[p], x1 = cur.x1[p], y0 = cur.y0[p], y1 = cur.y1[p];
                const int hx = x0 + ((x1 - x0 + 1) >> 1), hy = y0 + ((y1 - y0 + 1) >> 1);
                int after = 0;   // non-empty children with a larger index (pushed later => in front)
                for (int ch = 3; ch >= 0; ch--) {
                    const int cc = childcnt[4 * p + ch];
                    if (cc > 0) {
                        const int pos = basepos + after;
                        after++;
                        childpos[4 * p + ch] = pos;
                        nxt.x0[pos] = (ch & 1) ? hx : x0; nxt.x1[pos] = (ch & 1) ? x1 : hx;
                        nxt.y0[pos] = (ch & 2) ? hy : y0; nxt.y1[pos] = (ch & 2) ? y1 : hy;
                        nxt.cnt[pos] = cc;
                        nxt.seq[pos] = j * 4 + ch;
                        if (cc > 1) nexp_local++;
                    }
                }
            }
        }
        if (nexp_local) atomicAdd(&sh_flag, nexp_local);
        __syncthreads();
        for (int i = tid; i < n; i += kOctThreads) {
            const int p = knode[i];
            if (prank[p] < 0) knode[i] = (uint16_t)staypos[p];
            else {
                const uint32_t e = ent[i];
                const int ch = child_of(ent_x(e), ent_y(e), cur.x0[p], cur.x1[p], cur.y0[p], cur.y1[p]);
                knode[i] = (uint16_t)childpos[4 * p + ch];
            }
        }
        __syncthreads();
        const int nToExpand = sh_flag;
        { NodeTab t = cur; cur = nxt; nxt = t; }
        const int prevS = S;
        S = newS;
        if (S >= N || S == prevS) break;                      // :669-672, :733-734
        if (!careful && (S + nToExpand * 3) > N) careful = true;   // :673
    }

    // ---- survivor per node: first maximum response in candidate order (:744-760) ----
    for (int p = tid; p < S; p += kOctThreads) best[p] = -1;
    __syncthreads();
    for (int i = tid; i < n; i += kOctThreads)
        atomicMax(&best[knode[i]], (ent_s(ent[i]) << kIdxBits) | ((1 << kIdxBits) - 1 - i));
    __syncthreads();
    if (S > g.sel_cap) {
        if (tid == 0) { atomicOr(c.status, 8); *sel_count = 0; }
        return;
    }
    for (int p = tid; p < S; p += kOctThreads) {
        const int i = (1 << kIdxBits) - 1 - (best[p] & ((1 << kIdxBits) - 1));
        const uint32_t e = ent[i];
        sel[p] = pack_xys(ent_x(e) + kRegionBorder, ent_y(e) + kRegionBorder, ent_s(e));   // level coords, :839-840
    }
    if (tid == 0) *sel_count = S;
}

// One (image, level) of the quad-tree.
// MODE 0: node tables and candidates in LDS; a level with more than kCandCap candidates is only flagged (sel_count = -1) for k_octree_spill.
// MODE 1: node tables in LDS, candidates in the HBM spill arrays, flagged levels only.  MODE 2: everything in HBM (node tables beyond the LDS).
template <int MODE>
__device__ __forceinline__ void octree_block(const OrbCtx& c) {
    extern __shared__ __align__(16) uint8_t smem[];
    uint32_t* ent = (uint32_t*)smem;                       // [kCandCap]   (NODES_LDS only)
    uint16_t* knode = (uint16_t*)(ent + kCandCap);         // [kCandCap]
    int* ibase = MODE != 2 ? (int*)(knode + kCandCap) : c.oct_nodes + ((long long)blockIdx.y * c.P->nlevels + blockIdx.x) * c.oct_nodes_stride;
    const OrbParams* P = c.P;
    if (MODE == 1 && c.sel_count[(long long)blockIdx.y * P->nlevels + blockIdx.x] != -1) return;   // not flagged by k_octree (uniform)
    const int level = blockIdx.x, b = blockIdx.y;
    const LevelGeom& g = P->lv[level];
    const int tid = threadIdx.x;
    int* scratch = ibase + 18 * P->node_cap + 2;           // the body's 32-int scratch slot
    int* sel_count = c.sel_count + (long long)b * P->nlevels + level;

    // prefix of the per-cell counts, kept in the node-table area: the gather reads it while writing
    // only `ent`; the body first writes its tables after the gather's barrier
    const int ncells = g.nCols * g.nRows;
    const int* cell_count = c.cell_count + (long long)b * P->total_cells + g.cell_base;
    int* cellofs = ibase;         // ncells <= 18*node_cap ints (checked at create time); the body re-initialises this area after the gather
    for (int i = tid; i < ncells; i += kOctThreads) cellofs[i] = cell_count[i];
    __syncthreads();
    const int n = block_exclusive_scan(cellofs, ncells, scratch);
    if (n == 0) {
        if (tid == 0) *sel_count = 0;
        return;
    }
    if (n >= (1 << kIdxBits)) {
        if (tid == 0) { atomicOr(c.status, 2); *sel_count = 0; }
        return;
    }
    if (MODE == 0) {
        if (n > kCandCap) {   // the spill kernel redoes this level with the candidates in HBM
            if (tid == 0) *sel_count = -1;
            return;
        }
        octree_body(c, ent, knode, ibase, cellofs, n);
    } else {
        // spill variant: candidate list and node ids in HBM (L2-resident), same algorithm
        uint32_t* ent_g = c.ent_g + (long long)b * P->cand_per_image + g.cand_base;
        uint16_t* knode_g = c.knode_g + (long long)b * P->cand_per_image + g.cand_base;
        octree_body(c, ent_g, knode_g, ibase, cellofs, n);
    }
}

// Three kernels instead of one with three copies of the body: with the spill copy inside, the common variant ran 13 % slower, with the HBM
// copy as well 30 % slower (registers 91 -> 107, code size).
// (dbg[8 + level] += the workgroup's duration in 100 MHz ticks, one atomic per workgroup: tools/octree_levels_prof.py reads how the launch's slot time splits over
// the pyramid levels — VERDICT r4 asked whether levels 4-7 should share a workgroup)
__global__ __launch_bounds__(kOctThreads) void k_octree(OrbCtx c) {
    const long long t0 = wall_clock64();
    octree_block<0>(c);
    if (threadIdx.x == 0 && c.dbg && blockIdx.x < 8) atomicAdd(&c.dbg[8 + blockIdx.x], (unsigned long long)(wall_clock64() - t0));
}
// levels that k_octree flagged (more than kCandCap candidates: noise images); every other block returns after one load
__global__ __launch_bounds__(kOctThreads) void k_octree_spill(OrbCtx c) { octree_block<1>(c); }
// quotas so large that the node tables exceed the LDS (e.g. 2000+ features on one or two levels): tables and candidates in HBM
__global__ __launch_bounds__(kOctThreads) void k_octree_hbm(OrbCtx c) { octree_block<2>(c); }

// ------------------------------------------------------------------------------------------
// K5+K7: intensity-centroid orientation on the level image and steered BRIEF on the blurred
// level; one wavefront per keypoint.  Replaces reference src/ORBextractor.cc:77-147,472-479,
// 1090-1104 (IC_Angle, computeOrbDescriptor, the level concat and the pt *= scale).
// ------------------------------------------------------------------------------------------
__constant__ __attribute__((aligned(16))) int8_t c_pattern[1024] = {
#include "brief_pattern.inc"
};

__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    // cv::fastAtan2 (OpenCV 3.2 atanImpl<float>), fp32 without contraction
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float eps = (float)2.2204460492503131e-16;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, cc, c2;
    if (ax >= ay) {
        cc = __fdiv_rn(ay, ax + eps);
        c2 = cc * cc;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * cc;
    } else {
        cc = __fdiv_rn(ax, ay + eps);
        c2 = cc * cc;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * cc;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// 64 keypoint slots per 256-thread workgroup, three phases:
//  1. each wavefront computes the integer moments of its 16 keypoints (two patch rows per pass,
//     lanes 0-30 / 32-62 = columns -15..15, no divisions);
//  2. 64 lanes in parallel: fastAtan2 polynomial and the fp64 cos/sin of 64 keypoints at once
//     (the reference's (float)cos((double)angle)), instead of once per wavefront;
//  3. each wavefront writes the descriptors of its 16 keypoints (4 tests per lane).
// IC_Angle weights (reference :77-104) for a patch fetched as aligned words: entry (sh, i) belongs to word i = row * 9 + wi of the 31 x 9-word
// window whose first word holds column kx - 15 at byte sh.  w = (u + 15) per byte for the columns u inside the circular patch (|u| <= umax[|v|]), 0
// elsewhere; o = 1 / 0 likewise.  m10 = sum dot4(pixels, w) - 15 * dot4(pixels, o), m01 = sum v * dot4(pixels, o): two v_dot4_u32_u8 per word.
constexpr int kMomW = 9, kMomR = 31, kMomN = kMomR * kMomW, kMomStride = 280;
struct MomentLut { uint32_t w[4 * kMomStride]; uint32_t o[4 * kMomStride]; };
constexpr MomentLut make_moment_lut() {
    MomentLut L{};
    const int umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};   // HALF_PATCH_SIZE = 15, reference :454-469
    for (int sh = 0; sh < 4; sh++)
        for (int i = 0; i < kMomN; i++) {
            const int r = i / kMomW, wi = i - r * kMomW;
            const int v = r - 15, av = v < 0 ? -v : v;
            uint32_t W = 0, O = 0;
            for (int j = 0; j < 4; j++) {
                const int u15 = 4 * wi + j - sh, u = u15 - 15, au = u < 0 ? -u : u;
                if (u15 >= 0 && u15 <= 30 && au <= umax[av]) { W |= (uint32_t)u15 << (8 * j); O |= 1u << (8 * j); }
            }
            L.w[sh * kMomStride + i] = W; L.o[sh * kMomStride + i] = O;
        }
    return L;
}
__constant__ MomentLut c_mlut = make_moment_lut();

__global__ __launch_bounds__(256) void k_orient_describe(OrbCtx c, int kpw /* keypoints per wavefront: 16 for big batches, 1 for latency */) {
    const OrbParams* P = c.P;
    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nslot = 4 * kpw;
    const int slot0 = blockIdx.x * nslot;
    const int* sel_count = c.sel_count + (long long)b * P->nlevels;
    __shared__ int s_level[64], s_kx[64], s_ky[64], s_score[64], s_m01[64], s_m10[64];
    __shared__ float s_angle[64], s_a[64], s_b[64];

    // the moment weight table travels to LDS: requested here, before the dependent loads of phase 0, stored after them
    __shared__ uint2 s_mlut[4 * kMomStride];
    constexpr int kLutPerThread = (4 * kMomStride + 255) / 256;
    uint2 lutreg[kLutPerThread];
#pragma unroll
    for (int k = 0; k < kLutPerThread; k++) {
        const int i = min(tid + 256 * k, 4 * kMomStride - 1);
        lutreg[k] = make_uint2(c_mlut.w[i], c_mlut.o[i]);
    }
    // phase 0: slot -> (level, keypoint)
    if (tid < nslot) {
        const int slot = slot0 + tid;
        int level = -1, idx = slot, total = 0;
        for (int l = 0; l < P->nlevels; l++) {
            const int cnt = sel_count[l];
            if (level < 0 && idx < cnt) level = l;
            if (level < 0) idx -= cnt;
            total += cnt;
        }
        if (slot == 0) {
            c.out_count[b] = total;
            if (total > P->out_cap) atomicOr(c.status, 16);
        }
        if (slot >= P->out_cap) level = -1;
        s_level[tid] = level;
        if (level >= 0) {
            const uint32_t e = c.sel[(long long)b * P->sel_per_image + P->lv[level].sel_base + idx];
            s_kx[tid] = ent_x(e); s_ky[tid] = ent_y(e); s_score[tid] = ent_s(e);
        }
    }
#pragma unroll
    for (int k = 0; k < kLutPerThread; k++) {
        const int i = tid + 256 * k;
        if (i < 4 * kMomStride) s_mlut[i] = lutreg[k];
    }
    __syncthreads();
    // phase 1: IC_Angle moments (reference :77-104), integer exact
    // The 31 x 31 patch is fetched as 31 rows x 9 aligned words (279 words = 5 wave loads instead of 16 byte gathers); lane l owns the words
    // l, l + 64, ... and multiplies each with its two table words.  The words of the NEXT keypoint are in flight while the current one is reduced.
    uint32_t ow[5];
    int osh = 0;
    int vrow[5];
#pragma unroll
    for (int k = 0; k < 5; k++) vrow[k] = (lane + 64 * k) / kMomW - 15;
    auto fetch_orient = [&](int q) {
        const int s = wv * kpw + q;
        const int level = q < kpw ? s_level[s] : -1;
#pragma unroll
        for (int k = 0; k < 5; k++) ow[k] = 0;
        if (level < 0) return;
        int pitch;
        const uint8_t* img = level_image(c, P, b, level, pitch);
        const int x0 = s_kx[s] - 15, y0 = s_ky[s] - 15;
        osh = x0 & 3;
        const bool al = ((pitch & 3) == 0) && ((((unsigned long long)img) & 3ull) == 0);
        if (al) {
            const int wbase = x0 >> 2, wmax = (pitch >> 2) - 1;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const int i = lane + 64 * k;
                const int row = i / kMomW, wi = i - row * kMomW;
                if (i < kMomN) ow[k] = *(const uint32_t*)(img + (long long)(y0 + row) * pitch + 4 * min(wbase + wi, wmax));
            }
        } else {   // caller's level-0 buffer with an unaligned pitch: the same words from byte loads (bytes past the patch carry zero weights)
            const int w = P->lv[level].w;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const int i = lane + 64 * k;
                const int row = i / kMomW, wi = i - row * kMomW;
                if (i < kMomN) {
                    const uint8_t* rp = img + (long long)(y0 + row) * pitch;
                    const int cb = (x0 & ~3) + 4 * wi;
                    uint32_t v = 0;
#pragma unroll
                    for (int j = 0; j < 4; j++) v |= (uint32_t)rp[min(cb + j, w - 1)] << (8 * j);
                    ow[k] = v;
                }
            }
        }
    };
    fetch_orient(0);
    for (int q = 0; q < kpw; q++) {
        const int s = wv * kpw + q;
        const int level = s_level[s];
        int m10 = 0, m01 = 0;
        const uint2* lut = s_mlut + osh * kMomStride;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int i = lane + 64 * k;
            if (i < kMomN) {
                const uint2 t = lut[i];
                const int d1 = (int)__builtin_amdgcn_udot4(ow[k], t.x, 0u, false), d0 = (int)__builtin_amdgcn_udot4(ow[k], t.y, 0u, false);
                m10 += d1 - 15 * d0;
                m01 += vrow[k] * d0;
            }
        }
        fetch_orient(q + 1);
        if (level < 0) continue;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            m10 += __shfl_xor(m10, d, 64);
            m01 += __shfl_xor(m01, d, 64);
        }
        if (lane == 0) { s_m10[s] = m10; s_m01[s] = m01; }
    }
    __syncthreads();
    // phase 2: orientation and rotation coefficients, one lane per keypoint
    if (tid < nslot && s_level[tid] >= 0) {
        const float angle = fast_atan2_deg((float)s_m01[tid], (float)s_m10[tid]);
        const float factorPI = (float)(3.14159265358979323846 / 180.f);
        const float ang = angle * factorPI;
        s_angle[tid] = angle;
        s_a[tid] = (float)cos((double)ang);
        s_b[tid] = (float)sin((double)ang);
    }
    __syncthreads();
    // phase 3: steered BRIEF on the blurred level (:108-147) + keypoint record (:1090-1104).
    // The 37x37 neighbourhood (pattern radius 18.38 -> |row|,|col| <= 18 after rounding) is staged into LDS with
    // row-coalesced aligned word loads (6 wave loads of ~9 cache lines each instead of 8 byte gathers that touch up
    // to 37 lines each); the next keypoint's patch is fetched into registers while the current one is sampled.
    constexpr int kPW = 10, kPR = 37;                 // words per patch row, rows
    __shared__ uint32_t s_patch[4][kPR * kPW];
    uint32_t* patch = s_patch[wv];
    uint32_t nxt[6];
    auto fetch_patch = [&](int q) {                   // global -> registers for this wavefront's q-th keypoint
        const int s = wv * kpw + q;
        const int level = q < kpw ? s_level[s] : -1;
#pragma unroll
        for (int k = 0; k < 6; k++) nxt[k] = 0;
        if (level < 0) return;
        const LevelGeom& g = P->lv[level];
        const uint8_t* bimg = c.blur + (long long)b * c.blur_stride + g.img_off;
        const int x0 = s_kx[s] - 18, y0 = s_ky[s] - 18;
        const int wbase = x0 >> 2, wmax = (g.pitch >> 2) - 1;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const int i = lane + 64 * k;
            const int row = i / kPW, wi = i - row * kPW;
            if (i < kPR * kPW) nxt[k] = *(const uint32_t*)(bimg + (long long)(y0 + row) * g.pitch + 4 * min(wbase + wi, wmax));
        }
    };
    // this lane's four point pairs of the pattern (16 int8, the same for every keypoint): loaded once
    const int4 patw = *(const int4*)&c_pattern[lane * 16];
    const int patv[4] = {patw.x, patw.y, patw.z, patw.w};
    fetch_patch(0);
    for (int q = 0; q < kpw; q++) {
        const int s = wv * kpw + q;
        const int level = s_level[s];
        __builtin_amdgcn_wave_barrier();              // the previous keypoint's samples are done with the LDS patch
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const int i = lane + 64 * k;
            if (i < kPR * kPW) patch[i] = nxt[k];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        fetch_patch(q + 1);                            // in flight while this keypoint is described
        if (level < 0) continue;
        const LevelGeom& g = P->lv[level];
        const int kx = s_kx[s], ky = s_ky[s];
        const float a = s_a[s], bb = s_b[s];
        const uint8_t* pb = (const uint8_t*)patch + 18 * (kPW * 4) + ((kx - 18) & 3) + 18;   // pb[r*40 + c] = blurred(ky + r, kx + c)
        int nib = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int pw = patv[t];   // bytes: x0, y0, x1, y1 (signed)
            const float x0 = (float)(int8_t)(pw & 0xff), y0 = (float)(int8_t)((pw >> 8) & 0xff), x1 = (float)(int8_t)((pw >> 16) & 0xff), y1 = (float)(int8_t)(pw >> 24);
            const int r0 = __float2int_rn(x0 * bb + y0 * a), c0 = __float2int_rn(x0 * a - y0 * bb);
            const int r1 = __float2int_rn(x1 * bb + y1 * a), c1 = __float2int_rn(x1 * a - y1 * bb);
            const int t0 = pb[r0 * (kPW * 4) + c0], t1 = pb[r1 * (kPW * 4) + c1];
            nib |= (t0 < t1) << t;
        }
        const int other = __shfl_xor(nib, 1, 64);
        const long long o = (long long)b * P->out_cap + slot0 + s;
        if ((lane & 1) == 0) c.out_desc[o * 32 + (lane >> 1)] = (uint8_t)(nib | (other << 4));
        if (lane == 0) {
            oslam_keypoint_t kp;
            const float fx = (float)kx, fy = (float)ky;
            kp.x = level ? fx * g.scale : fx;
            kp.y = level ? fy * g.scale : fy;
            kp.size = g.kp_size;
            kp.angle = s_angle[s];
            kp.response = (float)s_score[s];
            kp.octave = level;
            kp.class_id = -1;
            c.out_kp[o] = kp;
        }
    }
}

}  // namespace oslam
