// Device/host shared geometry of the ORB extractor (product code).
#pragma once
#include <stdint.h>

namespace oslam {

constexpr int kEdgeThreshold = 19;   // reference src/ORBextractor.cc:74
constexpr int kHalfPatch = 15;       // :73
constexpr int kPatchSize = 31;       // :72
constexpr int kRegionBorder = 16;    // EDGE_THRESHOLD-3, :773
constexpr int kMaxCell = 64;         // max FAST cell interior edge handled by one workgroup
constexpr int kTilePitch = 72;       // LDS tile pitch (>= kMaxCell + 6)
constexpr int kCandCap = 4096;       // FAST candidates per (image, level) the quad-tree kernel holds in LDS (more: HBM spill path)
constexpr int kMaxRoots = 64;        // nIni upper bound

// candidate / survivor entry: x (12 bit) | y (12 bit) << 12 | score (8 bit) << 24
__host__ __device__ inline uint32_t pack_xys(int x, int y, int s) {
    return (uint32_t)x | ((uint32_t)y << 12) | ((uint32_t)s << 24);
}
__host__ __device__ inline int ent_x(uint32_t e) { return e & 0xFFF; }
__host__ __device__ inline int ent_y(uint32_t e) { return (e >> 12) & 0xFFF; }
__host__ __device__ inline int ent_s(uint32_t e) { return e >> 24; }

struct LevelGeom {
    int w, h, pitch;          // level image (level 0: the caller's image / pitch is passed separately)
    int region_w, region_h;   // [16, w-16) x [16, h-16): reference :773-782
    int nCols, nRows, wCell, hCell;  // :784-787
    int cell_base;            // first cell of this level in the per-image cell arrays
    int cell_cap;             // candidate slots per cell (= NMS upper bound)
    int cand_base;            // first candidate slot of this level in the per-image arena
    int quota;                // mnFeaturesPerLevel, :436-446
    int nIni;                 // quad-tree roots, :543
    int root_off;             // offset into root tables
    int sel_cap, sel_base;    // survivor slots of this level
    int xtab_off, ytab_off;   // resize coefficient tables
    int qtab_off;             // quad tables of k_resize_words (-1: level cannot use it)
    int lds_tile_ok;          // every 256 x 16 destination tile's source window fits the LDS tile of k_resize_lds
    long long img_off;        // byte offset inside the per-image pyramid / blur arenas
    float scale;              // mvScaleFactor[level]
    float kp_size;            // (int)(PATCH_SIZE*scale), :837
};

// One FAST cell as k_fast_cells_wave needs it (built on the host once per extractor: the geometry does not depend on the image of the
// batch).  32 bytes, read with one scalar load.
struct FastCellRec {
    uint32_t xy;         // iniX | iniY << 16 (first ROI column / row of the cell in its level image)
    uint32_t dims;       // cw | ch << 8 | level << 16 | valid << 24  (valid 0: not handled here, 1: process, 2: empty cell)
    uint32_t cand_ofs;   // first candidate slot of the cell in the per-image arena
    uint32_t img_off;    // byte offset of the level image in the per-image pyramid arena (levels >= 1)
    uint32_t cell_cap;   // candidate slots of the cell
    uint32_t pitch;      // level image pitch (levels >= 1)
    uint32_t pad0, pad1;
};

struct OrbParams {
    LevelGeom lv[OSLAM_MAX_LEVELS];
    int nlevels;
    int iniTh, minTh;
    int total_cells;
    int cand_per_image;   // u32 slots
    int sel_per_image;    // u32 slots
    int out_cap;          // keypoints per image in the output arrays
    int node_cap;         // quad-tree node table size
    int gk[7];            // 7-tap Gaussian, 8 fractional bits
    int umax[16];         // IC_Angle row extents, :454-469
    int blur_block_base[OSLAM_MAX_LEVELS + 1];   // first blockIdx.x of each level in k_blur_strip<false> (interior strips)
    int blurb_block_base[OSLAM_MAX_LEVELS + 1];  // ... in k_blur_strip<true> (border strips)
    int any_big_cell;     // some level has FAST cells larger than the per-wavefront kernel handles
};

}  // namespace oslam
