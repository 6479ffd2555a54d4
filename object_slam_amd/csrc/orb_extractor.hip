// Host side of the gfx950 ORB extractor: table construction (reference arithmetic of
// src/ORBextractor.cc:410-470 and the OpenCV-3.2 resize / Gaussian coefficient rules), HBM arena
// layout, kernel launches and the C ABI of include/oslam_hip.h.
#include "orb_kernels.hip"

#include <algorithm>
#include <cmath>
#include <vector>

#include "common.h"

#include <time.h>
#include <stdlib.h>

namespace oslam {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// see common.h: poll briefly, then sleep on a blocking event (one event per host thread and device)
static thread_local int t_wait_mode = -3;   // -3: the process default (OSLAM_WAIT_SPIN_US), else this thread's own mode (stream_wait_thread_mode)
void stream_wait_thread_mode(int spin_us) { t_wait_mode = spin_us; }
hipError_t stream_wait(hipStream_t s) {
    static const int spin_env = [] { const char* e = getenv("OSLAM_WAIT_SPIN_US"); return e ? atoi(e) : -1; }();
    const int spin_us = t_wait_mode != -3 ? t_wait_mode : spin_env;
    if (spin_us == -2) {   // pure poll of the stream (never sleeps on the interrupt path)
        for (;;) {
            const hipError_t q = hipStreamQuery(s);
            if (q != hipErrorNotReady) return q;
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
    }
    if (spin_us < 0) return hipStreamSynchronize(s);
    struct Ev { hipEvent_t ev = nullptr; int dev = -1; };
    static thread_local Ev t;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (!t.ev || t.dev != dev) {
        if (t.ev) (void)hipEventDestroy(t.ev);
        t.ev = nullptr;
        if ((e = hipEventCreateWithFlags(&t.ev, hipEventBlockingSync | hipEventDisableTiming)) != hipSuccess) return e;
        t.dev = dev;
    }
    if ((e = hipEventRecord(t.ev, s)) != hipSuccess) return e;
    timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (;;) {
        e = hipEventQuery(t.ev);
        if (e != hipErrorNotReady) return e;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if ((t1.tv_sec - t0.tv_sec) * 1000000L + (t1.tv_nsec - t0.tv_nsec) / 1000 > spin_us) break;
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    return hipEventSynchronize(t.ev);
}

__global__ void k_copy_to_host(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, size_t n16, size_t bytes) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t i = t0; i < n16; i += stride) ((uint4*)dst)[i] = ((const uint4*)src)[i];
    for (size_t i = n16 * 16 + t0; i < bytes; i += stride) dst[i] = src[i];
}
hipError_t copy_to_host_async(void* dst_pinned, const void* src_dev, size_t bytes, hipStream_t s) {
    if (!bytes) return hipSuccess;
    static const bool sdma = getenv("OSLAM_D2H_SDMA") != nullptr;   // A/B knob: the runtime's copy path
    if (sdma) return hipMemcpyAsync(dst_pinned, src_dev, bytes, hipMemcpyDeviceToHost, s);
    const bool aligned = ((((uintptr_t)dst_pinned) | ((uintptr_t)src_dev)) & 15) == 0;
    const size_t n16 = aligned ? bytes / 16 : 0, units = n16 + (bytes - n16 * 16);
    const int blocks = (int)((units + 255) / 256 < 1024 ? (units + 255) / 256 : 1024);
    hipLaunchKernelGGL(k_copy_to_host, dim3(blocks), dim3(256), 0, s, (uint8_t*)dst_pinned, (const uint8_t*)src_dev, n16, bytes);
    return hipGetLastError();
}

static inline int cv_round(float v) { return (int)lrintf(v); }     // cvRound: half-to-even
static inline int cv_round(double v) { return (int)lrint(v); }
static inline short sat_short(float v) {
    int i = cv_round(v);
    return (short)std::min(std::max(i, -32768), 32767);
}

}  // namespace oslam

using namespace oslam;

struct oslam_orb {
    int device = 0;
    int nfeatures = 0, nlevels = 0, iniTh = 0, minTh = 0;
    double scaleFactor = 0;
    int width = 0, height = 0, max_batch = 0;
    int blur_sse2 = 1;
    OrbParams P;
    std::vector<float> scale, invScale, sigma2, invSigma2;
    std::vector<int> quota;

    // device
    OrbParams* dP = nullptr;
    int2* d_rtab = nullptr;
    int* d_qbase = nullptr; uint4* d_qpx = nullptr;
    uint8_t* d_root_of_x = nullptr;
    short* d_root_x = nullptr;
    uint8_t* d_stage = nullptr;   // [B][H][pitch0] staging for host images
    int stage_pitch = 0;
    uint8_t* d_pyr = nullptr;
    long long pyr_stride = 0;
    uint8_t* d_blur = nullptr;
    long long blur_stride = 0;
    int* d_cell_count = nullptr;
    uint32_t* d_cand = nullptr;
    FastCellRec* d_fast_cells = nullptr;   // per-cell geometry of k_fast_cells_wave
    int* d_ovf_count = nullptr;   // FAST cells that took k_fast_cells_wave's every-pixel path since creation (diagnostics)
    uint32_t* d_ent_g = nullptr;
    uint16_t* d_knode_g = nullptr;
    uint32_t* d_sel = nullptr;
    int* d_sel_count = nullptr;
    oslam_keypoint_t* d_out_kp = nullptr;
    uint8_t* d_out_desc = nullptr;
    int* d_out_count = nullptr;
    int* d_status = nullptr;
    unsigned long long* d_dbg = nullptr;
    size_t oct_lds = 0;
    int* d_oct_nodes = nullptr; long long oct_nodes_stride = 0;   // quad-tree node tables in HBM when they exceed the LDS
    hipStream_t side_stream = nullptr;             // blur runs here, concurrently with FAST + quad-tree
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipStream_t fast0_stream = nullptr;            // FAST of level 0 (needs only the caller's image) runs here, beside the pyramid kernels
    hipEvent_t ev_fork0 = nullptr, ev_join0 = nullptr;
    hipStream_t aux_stream = nullptr, aux_side_stream = nullptr;   // second half of a large batch (see launch_batch)
    hipEvent_t ev_fork2 = nullptr, ev_join2 = nullptr, ev_fork3 = nullptr, ev_join3 = nullptr;
    int split_min = 1 << 30;                       // batches of at least this many images are cut in two halves (off by default: measured +4 % frames/s at
                                                   // B = 256, but the halves share the chip, so every kernel's own duration doubles; OSLAM_ORB_SPLIT_MIN enables it)
    int prof_batch_images = 0;
    bool no_lds_resize = false;                    // OSLAM_ORB_NO_LDS_RESIZE: kernel experiments

    // per-kernel-group timing (HIP events on the launch stream), enabled by oslam_orb_set_profiling
    int profiling = 0;
    hipEvent_t ev[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // [6], [7]: blur begin / end on its own stream; [8], [9]: level-0 FAST on its stream
    double prof_ms[5] = {0, 0, 0, 0, 0};
    long long prof_batches = 0, prof_images = 0;
    bool prof_pending = false, prof_fast0 = false;

    // last batch
    OrbCtx ctx;
    int last_batch = 0;
    hipStream_t last_stream = nullptr;
};

static void build_resize_tab(int ssize, int dsize, bool is_x, std::vector<int2>& tab) {
    // OpenCV 3.2 resize(): INTER_LINEAR coefficient tables (imgwarp.cpp)
    const double inv_scale = (double)dsize / ssize;
    const double scale = 1. / inv_scale;
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)std::floor(f);
        f -= s;
        if (is_x) {
            if (s < 0) { f = 0; s = 0; }
            if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        }
        const float c0 = 1.f - f, c1 = f;
        const int a0 = (unsigned short)sat_short(c0 * 2048), a1 = (unsigned short)sat_short(c1 * 2048);
        tab.push_back(make_int2(s, a0 | (a1 << 16)));
    }
}

extern "C" {

const char* oslam_last_error(void) { return g_err; }

int oslam_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void oslam_orb_destroy(oslam_orb_t* h) {
    if (!h) return;
    void* ptrs[] = {h->d_qbase, h->d_qpx, h->dP, h->d_rtab, h->d_root_of_x, h->d_root_x, h->d_stage, h->d_pyr, h->d_blur,
                    h->d_cell_count, h->d_cand, h->d_oct_nodes, h->d_fast_cells, h->d_ovf_count, h->d_ent_g, h->d_knode_g, h->d_sel, h->d_sel_count, h->d_out_kp, h->d_out_desc,
                    h->d_out_count, h->d_status, h->d_dbg};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (hipEvent_t e : h->ev)
        if (e) (void)hipEventDestroy(e);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->side_stream) (void)hipStreamDestroy(h->side_stream);
    if (h->fast0_stream) (void)hipStreamDestroy(h->fast0_stream);
    if (h->ev_fork0) (void)hipEventDestroy(h->ev_fork0);
    if (h->ev_join0) (void)hipEventDestroy(h->ev_join0);
    if (h->aux_stream) (void)hipStreamDestroy(h->aux_stream);
    if (h->aux_side_stream) (void)hipStreamDestroy(h->aux_side_stream);
    for (hipEvent_t e : {h->ev_fork2, h->ev_join2, h->ev_fork3, h->ev_join3}) if (e) (void)hipEventDestroy(e);
    delete h;
}

int oslam_orb_create(oslam_orb_t** out, int nfeatures, float scaleFactor_, int nlevels, int iniTh, int minTh,
                     int width, int height, int max_batch, int device) {
    if (!out) { set_error("out is NULL"); return OSLAM_E_INVALID; }
    *out = nullptr;
    if (nfeatures <= 0 || nlevels < 1 || nlevels > OSLAM_MAX_LEVELS || !(scaleFactor_ > 1.0f) || width <= 0 ||
        height <= 0 || max_batch < 1 || iniTh < 1 || minTh < 1 || iniTh > 255 || minTh > 255) {
        set_error("oslam_orb_create: invalid argument");
        return OSLAM_E_INVALID;
    }
    int ndev = oslam_device_count();
    if (ndev <= 0) { set_error("no HIP device visible: the gfx950 extractor has no CPU fallback"); return OSLAM_E_HIP; }
    if (device < 0 || device >= ndev) { set_error("device %d out of range (%d visible)", device, ndev); return OSLAM_E_INVALID; }
    OSLAM_HIP_CHECK(hipSetDevice(device));

    oslam_orb* h = new oslam_orb();
    // a HIP failure below releases what was allocated so far (oslam_orb_destroy tolerates a half-built handle)
#define ORB_CREATE_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); oslam_orb_destroy(h); return OSLAM_E_HIP; } } while (0)
    h->device = device; h->nfeatures = nfeatures; h->nlevels = nlevels; h->iniTh = iniTh; h->minTh = minTh;
    h->scaleFactor = scaleFactor_;   // float -> double member, reference include/ORBextractor.h:96
    h->width = width; h->height = height; h->max_batch = max_batch;
    const double scaleFactor = h->scaleFactor;

    // scale tables, reference src/ORBextractor.cc:415-432
    h->scale.resize(nlevels); h->sigma2.resize(nlevels); h->invScale.resize(nlevels); h->invSigma2.resize(nlevels);
    h->scale[0] = 1.0f; h->sigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
        h->scale[i] = (float)(h->scale[i - 1] * scaleFactor);
        h->sigma2[i] = h->scale[i] * h->scale[i];
    }
    for (int i = 0; i < nlevels; i++) {
        h->invScale[i] = 1.0f / h->scale[i];
        h->invSigma2[i] = 1.0f / h->sigma2[i];
    }
    // per-level quotas, :436-446
    h->quota.resize(nlevels);
    {
        float factor = (float)(1.0f / scaleFactor);
        float nDesired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
        int sum = 0;
        for (int l = 0; l < nlevels - 1; l++) {
            h->quota[l] = cv_round(nDesired);
            sum += h->quota[l];
            nDesired *= factor;
        }
        h->quota[nlevels - 1] = std::max(nfeatures - sum, 0);
    }

    OrbParams& P = h->P;
    memset(&P, 0, sizeof(P));
    P.nlevels = nlevels; P.iniTh = iniTh; P.minTh = minTh;
    // umax, :454-469
    {
        int v, v0, vmax = (int)std::floor(kHalfPatch * sqrt(2.f) / 2 + 1);
        int vmin = (int)std::ceil(kHalfPatch * sqrt(2.f) / 2);
        const double hp2 = kHalfPatch * kHalfPatch;
        for (v = 0; v <= vmax; ++v) P.umax[v] = cv_round(sqrt(hp2 - v * v));
        for (v = kHalfPatch, v0 = 0; v >= vmin; --v) {
            while (P.umax[v0] == P.umax[v0 + 1]) ++v0;
            P.umax[v] = v0;
            ++v0;
        }
    }
    {
        static const int kUmax15[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};   // packed in k_orient_describe
        for (int i = 0; i < 16; i++)
            if (P.umax[i] != kUmax15[i]) { set_error("umax table mismatch"); delete h; return OSLAM_E_INVALID; }
    }
    // Gaussian 7 taps sigma 2 -> 8-bit fixed point (OpenCV 3.2 getGaussianKernel + convertTo(CV_32S, 256))
    {
        float cf[7];
        double sum = 0;
        for (int i = 0; i < 7; i++) {
            double x = i - 3.0;
            cf[i] = (float)std::exp(-0.5 / 4.0 * x * x);
            sum += cf[i];
        }
        sum = 1. / sum;
        for (int i = 0; i < 7; i++) {
            cf[i] = (float)(cf[i] * sum);
            P.gk[i] = cv_round(cf[i] * 256.f);
        }
    }

    std::vector<int2> rtab;
    std::vector<int> qbase;
    std::vector<uint4> qpx;
    std::vector<uint8_t> root_of_x((size_t)nlevels * 4096, 0);
    std::vector<short> root_x;
    long long pyr_off = 0, blur_off = 0;
    int cell_base = 0, cand_base = 0, sel_base = 0, node_cap = 0;
    for (int l = 0; l < nlevels; l++) {
        LevelGeom& g = P.lv[l];
        const float sc = h->invScale[l];
        g.w = cv_round((float)width * sc);     // :1112
        g.h = cv_round((float)height * sc);
        g.pitch = (int)align_up(g.w, 64);
        g.scale = h->scale[l];
        g.kp_size = (float)(int)(kPatchSize * h->scale[l]);   // :837
        g.region_w = g.w - 2 * kRegionBorder;
        g.region_h = g.h - 2 * kRegionBorder;
        if (g.region_w < 30 || g.region_h < 30 || g.w >= 4096 || g.h >= 4096) {
            set_error("level %d is %dx%d: the FAST cell grid needs 62 <= dim < 4096 (reference divides by width/30)", l, g.w, g.h);
            delete h;
            return OSLAM_E_INVALID;
        }
        const float W = 30;
        const float fw = (float)g.region_w, fh = (float)g.region_h;
        g.nCols = (int)(fw / W);
        g.nRows = (int)(fh / W);
        g.wCell = (int)std::ceil(fw / g.nCols);
        g.hCell = (int)std::ceil(fh / g.nRows);
        if (g.wCell > kMaxCell || g.hCell > kMaxCell) { set_error("cell too large"); delete h; return OSLAM_E_INVALID; }
        g.cell_base = cell_base;
        g.cell_cap = ((g.wCell + 1) / 2) * ((g.hCell + 1) / 2);
        g.cand_base = cand_base;
        cell_base += g.nCols * g.nRows;
        cand_base += g.nCols * g.nRows * g.cell_cap;
        g.quota = h->quota[l];
        // quad-tree roots, :543-563
        g.nIni = (int)std::round((float)g.region_w / g.region_h);
        if (g.nIni < 1 || g.nIni > kMaxRoots) {
            set_error("level %d aspect %dx%d gives %d quad-tree roots (supported 1..%d)", l, g.region_w, g.region_h, g.nIni, kMaxRoots);
            delete h;
            return OSLAM_E_INVALID;
        }
        const float hX = (float)g.region_w / g.nIni;
        g.root_off = (int)root_x.size();
        for (int i = 0; i <= g.nIni; i++) root_x.push_back((short)(int)(hX * (float)i));
        for (int x = 0; x < g.region_w && x < 4096; x++) {
            int r = (int)((float)x / hX);   // vpIniNodes[kp.pt.x/hX], :569
            root_of_x[(size_t)l * 4096 + x] = (uint8_t)std::min(r, g.nIni - 1);
        }
        g.sel_cap = std::max(g.quota + 3, 4 * g.nIni) + 1;
        g.sel_base = sel_base;
        sel_base += g.sel_cap;
        node_cap = std::max(node_cap, g.sel_cap);
        g.img_off = blur_off;   // same offsets used in both arenas (level 0 slot unused in the pyramid arena)
        blur_off += (long long)g.pitch * g.h;
        if (l > 0) {
            g.xtab_off = (int)rtab.size();
            build_resize_tab(P.lv[l - 1].w, g.w, true, rtab);
            g.ytab_off = (int)rtab.size();
            build_resize_tab(P.lv[l - 1].h, g.h, false, rtab);
            // quad tables for the word-load kernel: every 4-pixel group must fit a 12-byte aligned window
            g.qtab_off = (int)qbase.size();
            bool fits = true;
            const int nq = div_up(g.w, 4);
            for (int q = 0; q < nq && fits; q++) {
                const int x0s = rtab[g.xtab_off + q * 4].x;
                const int xb = x0s & ~3, o0 = x0s & 3;
                uint32_t t[4], sl[4];
                for (int i = 0; i < 4; i++) {
                    const int x = std::min(q * 4 + i, g.w - 1);
                    const int2 e = rtab[g.xtab_off + x];
                    const int d = e.x - x0s;   // left tap of pixel i inside the 8 bytes that start at the first pixel's left tap
                    const int a0 = (short)(e.y & 0xFFFF), a1 = (short)(e.y >> 16);
                    if (x0s < 0 || d < 0 || d > 6 || a0 < 0 || a0 > 2048 || a1 < 0 || a1 > 2048) { fits = false; break; }
                    t[i] = (uint32_t)a0 | ((uint32_t)a1 << 16);
                    sl[i] = (uint32_t)d | (0x0cu << 8) | ((uint32_t)(d + 1) << 16) | (0x0cu << 24);   // v_perm: bytes d, d+1 -> two zero-extended 16-bit fields
                }
                qbase.push_back(xb | o0);
                qpx.push_back(make_uint4(t[0], t[1], t[2], t[3]));
                qpx.push_back(make_uint4(sl[0], sl[1], sl[2], sl[3]));
            }
            if (!fits) { qbase.resize(g.qtab_off); qpx.resize((size_t)2 * g.qtab_off); g.qtab_off = -1; }
            // can every 256 x 16 destination tile stage its source window in the LDS tile of k_resize_lds?
            g.lds_tile_ok = 0;
            if (fits) {
                const LevelGeom& gsrc = P.lv[l - 1];
                bool ok = true;
                const int maxoff = (gsrc.w - 1) & ~3;
                for (int q0 = 0; q0 < nq && ok; q0 += kRzTW / 4) {
                    const int q1 = std::min(q0 + kRzTW / 4, nq) - 1;
                    const int sx0 = (qbase[g.qtab_off + q0] & ~3) & ~15, sx1 = std::min((qbase[g.qtab_off + q1] & ~3) + 12, maxoff + 4);
                    if (sx1 - sx0 > kRzLdsPitch || sx0 < 0) ok = false;
                }
                for (int y0 = 0; y0 < g.h && ok; y0 += kRzTH) {
                    const int y1 = std::min(y0 + kRzTH, g.h) - 1;
                    const int sy0 = std::min(std::max(rtab[g.ytab_off + y0].x, 0), gsrc.h - 1), sy1 = std::min(std::max(rtab[g.ytab_off + y1].x + 1, 0), gsrc.h - 1);
                    if (sy1 - sy0 + 1 > kRzLdsRows) ok = false;
                }
                g.lds_tile_ok = ok ? 1 : 0;
            }
        }
    }
    (void)pyr_off;
    P.total_cells = cell_base;
    {
        // blur launch geometry (k_blur_strip): interior strips x4 = 4..last, border strips = left + 1-2 right;
        // sized for word-aligned rows, and for the all-border layout of an unaligned caller image at level 0
        int bi = 0, bbord = 0;
        P.any_big_cell = 0;
        for (int l = 0; l < nlevels; l++) {
            const int w = P.lv[l].w, hh = P.lv[l].h;
            const int last = ((w - 8) / 4) * 4, nint = w >= 12 ? last / 4 : 0;
            P.blur_block_base[l] = bi;
            bi += div_up(nint, 64) * div_up(hh, 4 * kBlurRows);
            const int nb_aligned = 1 + div_up(w - (nint ? last + 4 : 4), 4);
            const int nb_all = 1 + div_up(w - 4, 4);   // unaligned rows: every strip goes through the byte path
            const int nb = l == 0 ? nb_all : nb_aligned;
            P.blurb_block_base[l] = bbord;
            bbord += div_up(nb * div_up(hh, kBlurRows), 256);
            if (P.lv[l].wCell > kWCell || P.lv[l].hCell > kWCell) P.any_big_cell = 1;
        }
        P.blur_block_base[nlevels] = bi;
        P.blurb_block_base[nlevels] = bbord;
    }
    P.cand_per_image = cand_base;
    P.sel_per_image = sel_base;
    P.out_cap = sel_base;
    P.node_cap = node_cap + 8;
    for (int l = 0; l < nlevels; l++)
        if (P.lv[l].nCols * P.lv[l].nRows > 18 * P.node_cap) { set_error("too many FAST cells at level %d for the quad-tree kernel", l); delete h; return OSLAM_E_INVALID; }
    h->pyr_stride = blur_off;
    h->blur_stride = blur_off;
    {
        const size_t NC = P.node_cap;
        h->oct_lds = (size_t)kCandCap * 6 + (4 * NC + 4 * NC + NC * 3 + 2 * (NC + 1) + NC * 5 + 32) * 4 + 8 * NC * 2 + 64;
        if (h->oct_lds > 160 * 1024 - 512) {   // the node tables go to HBM (one slice per image and level), the kernel keeps 64 B of LDS
            h->oct_nodes_stride = (long long)align_up((4 * NC + 4 * NC + NC * 3 + 2 * (NC + 1) + NC * 5 + 32) + 4 * NC + 16, 64);
            h->oct_lds = 64;
        }
    }

    const size_t B = max_batch;
#define ALLOC(ptr, bytes)                                                         \
    do {                                                                          \
        hipError_t e_ = hipMalloc((void**)&(ptr), (bytes));                       \
        if (e_ != hipSuccess) {                                                   \
            set_error("hipMalloc(%zu) failed: %s", (size_t)(bytes), hipGetErrorString(e_)); \
            oslam_orb_destroy(h);                                                 \
            return OSLAM_E_HIP;                                                   \
        }                                                                         \
    } while (0)
    ALLOC(h->dP, sizeof(OrbParams));
    ALLOC(h->d_rtab, std::max<size_t>(rtab.size(), 1) * sizeof(int2));
    ALLOC(h->d_qbase, std::max<size_t>(qbase.size(), 1) * sizeof(int));
    ALLOC(h->d_qpx, std::max<size_t>(qpx.size(), 1) * sizeof(uint4));
    ALLOC(h->d_root_of_x, root_of_x.size());
    ALLOC(h->d_root_x, root_x.size() * sizeof(short));
    h->stage_pitch = (int)align_up(width, 64);
    ALLOC(h->d_stage, B * (size_t)h->stage_pitch * height);
    ALLOC(h->d_pyr, B * (size_t)h->pyr_stride);
    ALLOC(h->d_blur, B * (size_t)h->blur_stride);
    ALLOC(h->d_cell_count, B * (size_t)P.total_cells * sizeof(int));
    ALLOC(h->d_cand, B * (size_t)P.cand_per_image * sizeof(uint32_t));
    ALLOC(h->d_ovf_count, 64);
    if (h->oct_nodes_stride) ALLOC(h->d_oct_nodes, B * (size_t)nlevels * h->oct_nodes_stride * sizeof(int));
    ORB_CREATE_CHECK(hipMemset(h->d_ovf_count, 0, 64));
    ALLOC(h->d_ent_g, B * (size_t)P.cand_per_image * sizeof(uint32_t));
    ALLOC(h->d_knode_g, B * (size_t)P.cand_per_image * sizeof(uint16_t));
    ALLOC(h->d_sel, B * (size_t)P.sel_per_image * sizeof(uint32_t));
    ALLOC(h->d_sel_count, B * (size_t)nlevels * sizeof(int));
    ALLOC(h->d_out_kp, B * (size_t)P.out_cap * sizeof(oslam_keypoint_t));
    ALLOC(h->d_out_desc, B * (size_t)P.out_cap * 32);
    ALLOC(h->d_out_count, B * sizeof(int));
    ALLOC(h->d_status, sizeof(int));
    ALLOC(h->d_dbg, 16 * sizeof(unsigned long long));
    ORB_CREATE_CHECK(hipMemset(h->d_dbg, 0, 16 * sizeof(unsigned long long)));
#undef ALLOC
    ORB_CREATE_CHECK(hipMemcpy(h->dP, &P, sizeof(P), hipMemcpyHostToDevice));
    {   // FAST cell records (the cell grid of reference src/ORBextractor.cc:784-808, one record per cell of every level)
        std::vector<FastCellRec> cells((size_t)P.total_cells);
        for (int l = 0; l < nlevels; l++) {
            const LevelGeom& g = P.lv[l];
            const int ncell = g.nCols * g.nRows;
            for (int cell = 0; cell < ncell; cell++) {
                FastCellRec r{};
                const int ci = cell / g.nCols, cj = cell - ci * g.nCols;
                const int maxBX = g.w - kRegionBorder, maxBY = g.h - kRegionBorder;
                const int iniY = kRegionBorder + ci * g.hCell, iniX = kRegionBorder + cj * g.wCell;
                int maxY = iniY + g.hCell + 6, maxX = iniX + g.wCell + 6;
                const bool skip = (iniY >= maxBY - 3) || (iniX >= maxBX - 6);
                if (maxY > maxBY) maxY = maxBY;
                if (maxX > maxBX) maxX = maxBX;
                const int cw = maxX - iniX - 6, ch = maxY - iniY - 6;
                int valid = 1;
                if (g.wCell > kWCell || g.hCell > kWCell) valid = 0;          // k_fast_cells handles the level
                else if (skip || cw <= 0 || ch <= 0) valid = 2;
                if (iniX > 0xFFFF || iniY > 0xFFFF || g.img_off > 0xFFFFFFFFll) { set_error("oslam_orb_create: image too large for the FAST cell records"); oslam_orb_destroy(h); return OSLAM_E_INVALID; }
                r.xy = (uint32_t)iniX | ((uint32_t)iniY << 16);
                r.dims = (uint32_t)(valid == 1 ? cw : 0) | ((uint32_t)(valid == 1 ? ch : 0) << 8) | ((uint32_t)l << 16) | ((uint32_t)valid << 24);
                r.cand_ofs = (uint32_t)(g.cand_base + (long long)cell * g.cell_cap);
                r.img_off = (uint32_t)g.img_off;
                r.cell_cap = (uint32_t)g.cell_cap;
                r.pitch = (uint32_t)g.pitch;
                cells[(size_t)g.cell_base + cell] = r;
            }
        }
        if (hipMalloc((void**)&h->d_fast_cells, cells.size() * sizeof(FastCellRec)) != hipSuccess) { set_error("hipMalloc of the FAST cell records failed"); oslam_orb_destroy(h); return OSLAM_E_HIP; }
        ORB_CREATE_CHECK(hipMemcpy(h->d_fast_cells, cells.data(), cells.size() * sizeof(FastCellRec), hipMemcpyHostToDevice));
    }
    if (!rtab.empty()) ORB_CREATE_CHECK(hipMemcpy(h->d_rtab, rtab.data(), rtab.size() * sizeof(int2), hipMemcpyHostToDevice));
    if (!qbase.empty()) {
        ORB_CREATE_CHECK(hipMemcpy(h->d_qbase, qbase.data(), qbase.size() * sizeof(int), hipMemcpyHostToDevice));
        ORB_CREATE_CHECK(hipMemcpy(h->d_qpx, qpx.data(), qpx.size() * sizeof(uint4), hipMemcpyHostToDevice));
    }
    ORB_CREATE_CHECK(hipMemcpy(h->d_root_of_x, root_of_x.data(), root_of_x.size(), hipMemcpyHostToDevice));
    ORB_CREATE_CHECK(hipMemcpy(h->d_root_x, root_x.data(), root_x.size() * sizeof(short), hipMemcpyHostToDevice));
    ORB_CREATE_CHECK(hipMemset(h->d_status, 0, sizeof(int)));
    ORB_CREATE_CHECK(hipMemset(h->d_out_count, 0, B * sizeof(int)));
    ORB_CREATE_CHECK(hipFuncSetAttribute((const void*)k_octree, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->oct_lds));
    ORB_CREATE_CHECK(hipFuncSetAttribute((const void*)k_octree_spill, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->oct_lds));
    ORB_CREATE_CHECK(hipFuncSetAttribute((const void*)k_octree_hbm, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->oct_lds));
    {   // The blur runs on a side stream of ORDINARY priority.  Rounds 2-5 created it with the lowest priority (FAST, the overflow cells and the quad-tree on the
        // caller's stream dispatched first).  In the batch-of-sequences driver that starves it: with eight handles on the card there is almost always ordinary-priority
        // work of another handle to dispatch, the blur — and with it the frame's descriptor kernel and the whole Frame::Frame stage — waits for gaps, and the stereo
        // workload (two extractions per frame) fell into phases of 10-30x its Frame::Frame time in 6 of 9 runs on fresh boxes (2.3-4.0 k instead of 22-24 k frames/s;
        // tools/gpu/r5b_stereo_queues.sh).  Ordinary priority: 3 of 3 stereo runs at 22.4-23.3 k, the RGB-D headline 41.2-41.9 k against 41.0 k (same box).
        // OSLAM_ORB_SIDE_PRIORITY=low restores the lowest priority (A/B knob).
        int least = 0, greatest = 0;
        const char* spe = getenv("OSLAM_ORB_SIDE_PRIORITY");
        const bool low = spe && !strcmp(spe, "low");
        if (!low || hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || hipStreamCreateWithPriority(&h->side_stream, hipStreamNonBlocking, least) != hipSuccess) {
            (void)hipGetLastError();
            h->side_stream = nullptr;
            ORB_CREATE_CHECK(hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking));
        }
    }
    ORB_CREATE_CHECK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    if (getenv("OSLAM_ORB_FAST0_STREAM")) {   // kernel experiments (off: measured no gain at B = 512, the pyramid kernels slow down by what FAST gains: 2.60 ms per batch either way)
        ORB_CREATE_CHECK(hipStreamCreateWithFlags(&h->fast0_stream, hipStreamNonBlocking));
        ORB_CREATE_CHECK(hipEventCreateWithFlags(&h->ev_fork0, hipEventDisableTiming));
        ORB_CREATE_CHECK(hipEventCreateWithFlags(&h->ev_join0, hipEventDisableTiming));
    }
    ORB_CREATE_CHECK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    if (getenv("OSLAM_ORB_SPLIT_MIN")) h->split_min = atoi(getenv("OSLAM_ORB_SPLIT_MIN"));   // kernel experiments
    h->no_lds_resize = getenv("OSLAM_ORB_NO_LDS_RESIZE") != nullptr;
    if (max_batch >= h->split_min) {
        ORB_CREATE_CHECK(hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking));
        ORB_CREATE_CHECK(hipStreamCreateWithFlags(&h->aux_side_stream, hipStreamNonBlocking));
        for (hipEvent_t* e : {&h->ev_fork2, &h->ev_join2, &h->ev_fork3, &h->ev_join3}) ORB_CREATE_CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
#undef ORB_CREATE_CHECK
    *out = h;
    return OSLAM_OK;
}

int oslam_orb_get_scale_tables(const oslam_orb_t* h, float* sf, float* isf, float* s2, float* is2, int* nf) {
    if (!h) { set_error("NULL handle"); return OSLAM_E_INVALID; }
    for (int i = 0; i < h->nlevels; i++) {
        if (sf) sf[i] = h->scale[i];
        if (isf) isf[i] = h->invScale[i];
        if (s2) s2[i] = h->sigma2[i];
        if (is2) is2[i] = h->invSigma2[i];
        if (nf) nf[i] = h->quota[i];
    }
    return OSLAM_OK;
}

int oslam_orb_max_keypoints(const oslam_orb_t* h) { return h ? h->P.out_cap : OSLAM_E_INVALID; }

int oslam_orb_set_blur_rounding(oslam_orb_t* h, int sse2) {
    if (!h) return OSLAM_E_INVALID;
    h->blur_sse2 = sse2 != 0;
    return OSLAM_OK;
}

static int collect_profile(oslam_orb* h) {
    if (!h->prof_pending) return OSLAM_OK;
    OSLAM_HIP_CHECK(hipEventSynchronize(h->ev[5]));
    // groups: 0 pyramid, 1 FAST, 2 blur (on its own stream, overlapping FAST + quad-tree exactly as in un-profiled runs), 3 quad-tree, 4 orientation + descriptors
    const int a[5] = {0, 1, 6, 2, 4}, b[5] = {1, 2, 7, 3, 5};
    OSLAM_HIP_CHECK(hipEventSynchronize(h->ev[7]));
    for (int i = 0; i < 5; i++) {
        float ms = 0;
        OSLAM_HIP_CHECK(hipEventElapsedTime(&ms, h->ev[a[i]], h->ev[b[i]]));
        h->prof_ms[i] += ms;
    }
    if (h->prof_fast0) {   // level 0's FAST cells ran on their own stream: their launch counts to the FAST group
        float ms = 0;
        OSLAM_HIP_CHECK(hipEventSynchronize(h->ev[9]));
        OSLAM_HIP_CHECK(hipEventElapsedTime(&ms, h->ev[8], h->ev[9]));
        h->prof_ms[1] += ms;
    }
    h->prof_batches++;
    h->prof_images += h->prof_batch_images;
    h->prof_pending = false;
    return OSLAM_OK;
}

static int launch_batch(oslam_orb* h, const uint8_t* d_gray, int batch, int stride, size_t image_stride, hipStream_t st) {
    const OrbParams& P = h->P;
    OrbCtx c;
    c.P = h->dP;
    c.img0 = d_gray; c.img0_pitch = stride; c.img0_stride = (long long)image_stride;
    c.pyr = h->d_pyr; c.pyr_stride = h->pyr_stride;
    c.blur = h->d_blur; c.blur_stride = h->blur_stride;
    c.rtab = h->d_rtab; c.qbase = h->d_qbase; c.qpx = h->d_qpx; c.root_of_x = h->d_root_of_x; c.root_x = h->d_root_x;
    c.cell_count = h->d_cell_count; c.cand = h->d_cand; c.fast_cells = h->d_fast_cells; c.oct_nodes = h->d_oct_nodes; c.oct_nodes_stride = h->oct_nodes_stride; c.ovf_count = h->d_ovf_count; c.ent_g = h->d_ent_g; c.knode_g = h->d_knode_g; c.sel = h->d_sel; c.sel_count = h->d_sel_count;
    c.out_kp = h->d_out_kp; c.out_desc = h->d_out_desc; c.out_count = h->d_out_count; c.status = h->d_status; c.dbg = h->d_dbg;
    const bool prof = h->profiling != 0;
    if (prof) {
        int rc = collect_profile(h);   // previous batch, if not collected yet
        if (rc) return rc;
    }
    h->ctx = c; h->last_batch = batch; h->last_stream = st;
    // One sub-batch = the whole kernel sequence for images [b0, b0 + nb) on a (main, blur) stream pair.  The blur needs only the pyramid and
    // so do FAST + quad-tree: the blur goes to the pair's second stream and overlaps the (VALU-bound) FAST kernel and the
    // (barrier-latency-bound) quad-tree kernel.  Large batches are cut in two halves on two stream pairs so that the kernels of one half
    // (each bound by a different resource) overlap the kernels of the other; profiling events bracket the kernels of the first half.
    auto issue = [&](const OrbCtx& cs, int nb, hipStream_t sm, hipStream_t sb, hipStream_t f0, hipEvent_t fork, hipEvent_t join, bool pr) -> int {
#define PROF_MARK(i) do { if (pr) OSLAM_HIP_CHECK(hipEventRecord(h->ev[i], sm)); } while (0)
        if (pr) h->prof_fast0 = false;
        PROF_MARK(0);
        // level 0's FAST cells need only the caller's image: they run on their own stream beside the (HBM / latency bound) pyramid kernels
        const int cells0 = (f0 && P.nlevels > 1) ? P.lv[1].cell_base : 0;
        if (cells0 > 0) {
            OSLAM_HIP_CHECK(hipEventRecord(h->ev_fork0, sm));
            OSLAM_HIP_CHECK(hipStreamWaitEvent(f0, h->ev_fork0, 0));
            if (pr) OSLAM_HIP_CHECK(hipEventRecord(h->ev[8], f0));
            hipLaunchKernelGGL(k_fast_cells_wave, dim3(div_up(cells0, 4 * kFastCellsPerWave), nb), dim3(256), 0, f0, cs, 0, cells0);
            if (pr) { OSLAM_HIP_CHECK(hipEventRecord(h->ev[9], f0)); h->prof_fast0 = true; }
            OSLAM_HIP_CHECK(hipEventRecord(h->ev_join0, f0));
        }
        for (int l = 1; l < P.nlevels; l++) {
            const LevelGeom& g = P.lv[l];
            dim3 grid(div_up(g.w, 256), div_up(g.h, 4), nb);
            dim3 gridw(div_up(g.w, 256), div_up(g.h, 4 * kResizeRows), nb);
            // source rows 4-byte aligned? (levels >= 1 always; level 0 is the caller's buffer)
            const bool src_aligned = l > 1 || (((stride & 3) == 0) && ((((uintptr_t)cs.img0) & 3) == 0) && ((image_stride & 3) == 0));
            // source rows 16-byte aligned (LDS-staged variant)?
            const bool src_aligned16 = l > 1 || (((stride & 15) == 0) && ((((uintptr_t)cs.img0) & 15) == 0) && ((image_stride & 15) == 0));
            if (g.qtab_off >= 0 && g.lds_tile_ok && src_aligned16 && !h->no_lds_resize)
                hipLaunchKernelGGL(k_resize_lds, dim3(div_up(g.w, kRzTW), div_up(g.h, kRzTH), nb), dim3(256), 0, sm, cs, l);
            else if (g.qtab_off >= 0 && src_aligned) hipLaunchKernelGGL(k_resize_words, gridw, dim3(256), 0, sm, cs, l);
            else hipLaunchKernelGGL(k_resize, grid, dim3(256), 0, sm, cs, l);
        }
        PROF_MARK(1);
        const bool overlap = sb != sm;
        if (overlap) {
            OSLAM_HIP_CHECK(hipEventRecord(fork, sm));
            OSLAM_HIP_CHECK(hipStreamWaitEvent(sb, fork, 0));
        }
        hipLaunchKernelGGL(k_fast_cells_wave, dim3(div_up(P.total_cells - cells0, 4 * kFastCellsPerWave), nb), dim3(256), 0, sm, cs, cells0, P.total_cells);
        if (cells0 > 0) OSLAM_HIP_CHECK(hipStreamWaitEvent(sm, h->ev_join0, 0));
        if (getenv("OSLAM_ORB_DEBUG_OVF")) { int n = 0; (void)hipStreamSynchronize(sm); (void)hipMemcpy(&n, cs.ovf_count, 4, hipMemcpyDeviceToHost); fprintf(stderr, "fast overflow cells since creation: %d (this batch has %d cells)\n", n, P.total_cells * nb); }
        if (P.any_big_cell) hipLaunchKernelGGL(k_fast_cells, dim3(P.total_cells, nb), dim3(256), 0, sm, cs);
        PROF_MARK(2);
        if (pr) OSLAM_HIP_CHECK(hipEventRecord(h->ev[6], sb));
        hipLaunchKernelGGL(k_blur_strip<false>, dim3(P.blur_block_base[P.nlevels], nb), dim3(256), 0, sb, cs, h->blur_sse2);
        hipLaunchKernelGGL(k_blur_strip<true>, dim3(P.blurb_block_base[P.nlevels], nb), dim3(256), 0, sb, cs, h->blur_sse2);
        if (pr) OSLAM_HIP_CHECK(hipEventRecord(h->ev[7], sb));
        if (cs.oct_nodes) hipLaunchKernelGGL(k_octree_hbm, dim3(P.nlevels, nb), dim3(kOctThreads), h->oct_lds, sm, cs);
        else {
            hipLaunchKernelGGL(k_octree, dim3(P.nlevels, nb), dim3(kOctThreads), h->oct_lds, sm, cs);
            hipLaunchKernelGGL(k_octree_spill, dim3(P.nlevels, nb), dim3(kOctThreads), h->oct_lds, sm, cs);
        }
        PROF_MARK(3);
        if (overlap) {
            OSLAM_HIP_CHECK(hipEventRecord(join, sb));
            OSLAM_HIP_CHECK(hipStreamWaitEvent(sm, join, 0));
        }
        PROF_MARK(4);
        {
            const int kpw = nb >= 32 ? 16 : (nb >= 8 ? 4 : 1);
            hipLaunchKernelGGL(k_orient_describe, dim3(div_up(P.out_cap, 4 * kpw), nb), dim3(256), 0, sm, cs, kpw);
        }
        PROF_MARK(5);
#undef PROF_MARK
        return OSLAM_OK;
    };
    auto sub_ctx = [&](int b0) {
        OrbCtx cs = c;
        const long long o = b0;
        cs.img0 += o * c.img0_stride; cs.pyr += o * c.pyr_stride; cs.blur += o * c.blur_stride;
        cs.cell_count += o * P.total_cells; if (cs.oct_nodes) cs.oct_nodes += o * P.nlevels * c.oct_nodes_stride; cs.cand += o * P.cand_per_image; cs.ent_g += o * P.cand_per_image; cs.knode_g += o * P.cand_per_image;
        cs.sel += o * P.sel_per_image; cs.sel_count += o * P.nlevels;
        cs.out_kp += o * P.out_cap; cs.out_desc += o * P.out_cap * 32; cs.out_count += o;
        return cs;
    };
    int rc;
    const bool split = batch >= h->split_min && h->aux_stream != nullptr;
    if (!split) {
        rc = issue(c, batch, st, h->side_stream ? h->side_stream : st, h->fast0_stream, h->ev_fork, h->ev_join, prof);
        if (rc) return rc;
        h->prof_batch_images = batch;
    } else {
        const int nb0 = batch / 2, nb1 = batch - nb0;
        OSLAM_HIP_CHECK(hipEventRecord(h->ev_fork2, st));
        OSLAM_HIP_CHECK(hipStreamWaitEvent(h->aux_stream, h->ev_fork2, 0));
        if ((rc = issue(c, nb0, st, h->side_stream, h->fast0_stream, h->ev_fork, h->ev_join, prof))) return rc;
        if ((rc = issue(sub_ctx(nb0), nb1, h->aux_stream, h->aux_side_stream, nullptr, h->ev_fork3, h->ev_join3, false))) return rc;
        OSLAM_HIP_CHECK(hipEventRecord(h->ev_join2, h->aux_stream));
        OSLAM_HIP_CHECK(hipStreamWaitEvent(st, h->ev_join2, 0));
        h->prof_batch_images = nb0;
    }
    if (prof) h->prof_pending = true;
    OSLAM_HIP_CHECK(hipGetLastError());
    return OSLAM_OK;
}

int oslam_orb_extract_batch_device(oslam_orb_t* h, const uint8_t* d_gray, int batch, int stride, size_t image_stride,
                                   void* stream) {
    if (!h || !d_gray) { set_error("NULL argument"); return OSLAM_E_INVALID; }
    if (batch < 1 || batch > h->max_batch) { set_error("batch %d outside [1,%d]", batch, h->max_batch); return OSLAM_E_INVALID; }
    if (stride < h->width || (batch > 1 && image_stride < (size_t)stride * h->height)) { set_error("bad strides"); return OSLAM_E_INVALID; }
    OSLAM_HIP_CHECK(hipSetDevice(h->device));
    return launch_batch(h, d_gray, batch, stride, image_stride, (hipStream_t)stream);
}

int oslam_orb_results_device(const oslam_orb_t* h, const oslam_keypoint_t** kp, const uint8_t** desc, const int32_t** counts,
                             const int32_t** status) {
    if (!h) { set_error("NULL handle"); return OSLAM_E_INVALID; }
    if (kp) *kp = h->d_out_kp;
    if (desc) *desc = h->d_out_desc;
    if (counts) *counts = h->d_out_count;
    if (status) *status = h->d_status;
    return OSLAM_OK;
}

static int check_status(oslam_orb* h) {
    int st = 0;
    OSLAM_HIP_CHECK(hipMemcpy(&st, h->d_status, sizeof(int), hipMemcpyDeviceToHost));
    if (st) {
        OSLAM_HIP_CHECK(hipMemset(h->d_status, 0, sizeof(int)));
        set_error("extractor arena overflow, status bits 0x%x (1 cell, 2 candidates per level, 4/8 quad-tree nodes, 16 outputs)", st);
        return OSLAM_E_CAPACITY;
    }
    return OSLAM_OK;
}

int oslam_orb_fetch(oslam_orb_t* h, int b, oslam_keypoint_t* kps, uint8_t* desc, int cap, int* n_out) {
    if (!h || !n_out) { set_error("NULL argument"); return OSLAM_E_INVALID; }
    if (b < 0 || b >= h->last_batch) { set_error("image %d not in the last batch (%d)", b, h->last_batch); return OSLAM_E_INVALID; }
    OSLAM_HIP_CHECK(hipSetDevice(h->device));
    OSLAM_HIP_CHECK(hipStreamSynchronize(h->last_stream));
    int rc = check_status(h);
    if (rc) return rc;
    int n = 0;
    OSLAM_HIP_CHECK(hipMemcpy(&n, h->d_out_count + b, sizeof(int), hipMemcpyDeviceToHost));
    *n_out = n;
    if (n > cap) { set_error("caller capacity %d < %d keypoints", cap, n); return OSLAM_E_CAPACITY; }
    if (n > 0) {
        if (!kps || !desc) { set_error("NULL output"); return OSLAM_E_INVALID; }
        OSLAM_HIP_CHECK(hipMemcpy(kps, h->d_out_kp + (size_t)b * h->P.out_cap, (size_t)n * sizeof(oslam_keypoint_t), hipMemcpyDeviceToHost));
        OSLAM_HIP_CHECK(hipMemcpy(desc, h->d_out_desc + (size_t)b * h->P.out_cap * 32, (size_t)n * 32, hipMemcpyDeviceToHost));
    }
    return OSLAM_OK;
}

int oslam_orb_extract(oslam_orb_t* h, const uint8_t* gray, int width, int height, int stride, oslam_keypoint_t* kps,
                      uint8_t* desc, int cap, int* n_out) {
    if (!h || !n_out) { set_error("NULL argument"); return OSLAM_E_INVALID; }
    *n_out = 0;
    if (!gray || width == 0 || height == 0) return OSLAM_OK;   // reference: empty image -> silent return, :1046
    if (width != h->width || height != h->height || stride < width) {
        set_error("image %dx%d (stride %d) does not match the handle geometry %dx%d", width, height, stride, h->width, h->height);
        return OSLAM_E_INVALID;
    }
    OSLAM_HIP_CHECK(hipSetDevice(h->device));
    OSLAM_HIP_CHECK(hipMemcpy2D(h->d_stage, h->stage_pitch, gray, stride, width, height, hipMemcpyHostToDevice));
    int rc = launch_batch(h, h->d_stage, 1, h->stage_pitch, (size_t)h->stage_pitch * height, nullptr);
    if (rc) return rc;
    return oslam_orb_fetch(h, 0, kps, desc, cap, n_out);
}

int oslam_orb_set_profiling(oslam_orb_t* h, int on) {
    if (!h) { set_error("NULL handle"); return OSLAM_E_INVALID; }
    OSLAM_HIP_CHECK(hipSetDevice(h->device));
    if (on && !h->ev[0])
        for (int i = 0; i < 10; i++) OSLAM_HIP_CHECK(hipEventCreate(&h->ev[i]));
    h->profiling = on != 0;
    for (int i = 0; i < 5; i++) h->prof_ms[i] = 0;
    h->prof_batches = 0; h->prof_images = 0; h->prof_pending = false;
    return OSLAM_OK;
}

int oslam_orb_get_profile(oslam_orb_t* h, double ms[5], long long* batches, long long* images) {
    if (!h) { set_error("NULL handle"); return OSLAM_E_INVALID; }
    int rc = collect_profile(h);
    if (rc) return rc;
    for (int i = 0; i < 5; i++) ms[i] = h->prof_ms[i];
    if (batches) *batches = h->prof_batches;
    if (images) *images = h->prof_images;
    return OSLAM_OK;
}

int oslam_orb_debug_counters(oslam_orb_t* h, unsigned long long out[16], int reset) {
    if (!h) return OSLAM_E_INVALID;
    OSLAM_HIP_CHECK(hipDeviceSynchronize());
    OSLAM_HIP_CHECK(hipMemcpy(out, h->d_dbg, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (reset) OSLAM_HIP_CHECK(hipMemset(h->d_dbg, 0, 16 * sizeof(unsigned long long)));
    return OSLAM_OK;
}

int oslam_orb_level_size(const oslam_orb_t* h, int level, int* w, int* hh) {
    if (!h || level < 0 || level >= h->nlevels) { set_error("bad level"); return OSLAM_E_INVALID; }
    if (w) *w = h->P.lv[level].w;
    if (hh) *hh = h->P.lv[level].h;
    return OSLAM_OK;
}

int oslam_orb_pyramid_level_device(const oslam_orb_t* h, int b, int level, const uint8_t** d_ptr, int* pitch) {
    if (!h || level < 0 || level >= h->nlevels || b < 0 || b >= h->last_batch) { set_error("bad level/image"); return OSLAM_E_INVALID; }
    if (level == 0) {
        *d_ptr = h->ctx.img0 + (long long)b * h->ctx.img0_stride;
        *pitch = h->ctx.img0_pitch;
    } else {
        *d_ptr = h->d_pyr + (long long)b * h->pyr_stride + h->P.lv[level].img_off;
        *pitch = h->P.lv[level].pitch;
    }
    return OSLAM_OK;
}

int oslam_orb_get_pyramid_level(oslam_orb_t* h, int b, int level, uint8_t* out) {
    const uint8_t* p; int pitch;
    int rc = oslam_orb_pyramid_level_device(h, b, level, &p, &pitch);
    if (rc) return rc;
    OSLAM_HIP_CHECK(hipStreamSynchronize(h->last_stream));
    const LevelGeom& g = h->P.lv[level];
    OSLAM_HIP_CHECK(hipMemcpy2D(out, g.w, p, pitch, g.w, g.h, hipMemcpyDeviceToHost));
    return OSLAM_OK;
}

int oslam_orb_debug_get_blurred(oslam_orb_t* h, int b, int level, uint8_t* out) {
    if (!h || level < 0 || level >= h->nlevels || b < 0 || b >= h->last_batch) { set_error("bad level/image"); return OSLAM_E_INVALID; }
    OSLAM_HIP_CHECK(hipStreamSynchronize(h->last_stream));
    const LevelGeom& g = h->P.lv[level];
    OSLAM_HIP_CHECK(hipMemcpy2D(out, g.w, h->d_blur + (long long)b * h->blur_stride + g.img_off, g.pitch, g.w, g.h, hipMemcpyDeviceToHost));
    return OSLAM_OK;
}

int oslam_orb_debug_get_candidates(oslam_orb_t* h, int b, int level, int32_t* out, int cap, int* n_out) {
    if (!h || level < 0 || level >= h->nlevels || b < 0 || b >= h->last_batch || !n_out) { set_error("bad argument"); return OSLAM_E_INVALID; }
    OSLAM_HIP_CHECK(hipStreamSynchronize(h->last_stream));
    const LevelGeom& g = h->P.lv[level];
    const int ncells = g.nCols * g.nRows;
    std::vector<int> cnt(ncells);
    std::vector<uint32_t> ent((size_t)ncells * g.cell_cap);
    OSLAM_HIP_CHECK(hipMemcpy(cnt.data(), h->d_cell_count + (size_t)b * h->P.total_cells + g.cell_base, ncells * sizeof(int), hipMemcpyDeviceToHost));
    OSLAM_HIP_CHECK(hipMemcpy(ent.data(), h->d_cand + (size_t)b * h->P.cand_per_image + g.cand_base, ent.size() * 4, hipMemcpyDeviceToHost));
    int n = 0;
    for (int ce = 0; ce < ncells; ce++)
        for (int i = 0; i < cnt[ce]; i++, n++)
            if (n < cap) {
                uint32_t e = ent[(size_t)ce * g.cell_cap + i];
                out[3 * n] = ent_x(e); out[3 * n + 1] = ent_y(e); out[3 * n + 2] = ent_s(e);
            }
    *n_out = n;
    if (n > cap) { set_error("capacity"); return OSLAM_E_CAPACITY; }
    return OSLAM_OK;
}

int oslam_orb_debug_get_level_keys(oslam_orb_t* h, int b, int level, int32_t* out, int cap, int* n_out) {
    if (!h || level < 0 || level >= h->nlevels || b < 0 || b >= h->last_batch || !n_out) { set_error("bad argument"); return OSLAM_E_INVALID; }
    OSLAM_HIP_CHECK(hipStreamSynchronize(h->last_stream));
    const LevelGeom& g = h->P.lv[level];
    int n = 0;
    OSLAM_HIP_CHECK(hipMemcpy(&n, h->d_sel_count + (size_t)b * h->nlevels + level, sizeof(int), hipMemcpyDeviceToHost));
    *n_out = n;
    if (n > cap) { set_error("capacity"); return OSLAM_E_CAPACITY; }
    std::vector<uint32_t> ent(std::max(n, 1));
    OSLAM_HIP_CHECK(hipMemcpy(ent.data(), h->d_sel + (size_t)b * h->P.sel_per_image + g.sel_base, (size_t)n * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) { out[3 * i] = ent_x(ent[i]); out[3 * i + 1] = ent_y(ent[i]); out[3 * i + 2] = ent_s(ent[i]); }
    return OSLAM_OK;
}

int oslam_memcpy_from_device(void* dst, const void* d_src, size_t bytes) {
    OSLAM_HIP_CHECK(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
    return OSLAM_OK;
}

int64_t oslam_orb_algorithmic_bytes(const oslam_orb_t* h, int n_kp) {
    // W*H + (P-p_last)+(P-p0) + P + 2P + 749 N + 512 N + 60 N
    if (!h) return 0;
    int64_t Ptot = 0;
    for (int l = 0; l < h->nlevels; l++) Ptot += (int64_t)h->P.lv[l].w * h->P.lv[l].h;
    const int64_t p0 = (int64_t)h->P.lv[0].w * h->P.lv[0].h;
    const int64_t pl = (int64_t)h->P.lv[h->nlevels - 1].w * h->P.lv[h->nlevels - 1].h;
    return p0 + (Ptot - pl) + (Ptot - p0) + Ptot + 2 * Ptot + (int64_t)(749 + 512 + 60) * n_kp;
}

}  // extern "C"
