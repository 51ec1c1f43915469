// orb_plan.h — geometry of one ORB extraction configuration (image size x extractor params).
// Host computes it once per (rows, cols); kernels receive it by value.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace myslam_hip {

constexpr int MAXL = 12;              // max pyramid levels supported
constexpr int EDGE_THRESHOLD = 19;    // reference ORBextractor.cpp:25
constexpr int MIN_BORDER = 16;        // EDGE_THRESHOLD-3, ORBextractor.cpp:822
constexpr int PATCH_SIZE = 31;
constexpr int HALF_PATCH = 15;
constexpr int MAX_CELL = 59;          // FAST grid cell interior is < 60 px for any image (W=30)
constexpr int MAX_DEPTH = 13;         // quad-tree digits stored in the 32-bit path code
constexpr int ROOT_SHIFT = 26;        // code = root<<26 | d1<<24 | ... | d13

struct LevelGeom {
    int w, h, pitch;                  // level image size and internal row pitch (bytes, multiple of 64)
    int nCols, nRows, wCell, hCell;   // FAST grid, ORBextractor.cpp:830-836
    int maxBX, maxBY;                 // maxBorderX/Y, :824-825
    int cellBase;                     // first cell index of this level in the per-image cell list
    int stripBase;                    // first strip (4 horizontally adjacent cells) of this level
    int N;                            // oct-tree budget mnFeaturesPerLevel[l] (:410-421) or nfeatures (Detect)
    int nIni;                         // root nodes, :590
    float hX;                         // :592
    int sortDepth;                    // D: quad-tree levels covered by the LDS counting sort (nIni*4^D <= 1024 buckets)
    int ndepth;                       // quad-tree depth digits that can ever matter (<= MAX_DEPTH)
    int keyCap;                       // candidate capacity of the level
    int nodeCap;                      // node-list capacity (>= N+3)
    int outBase;                      // first slot of this level in the per-image selected-key array
    float scale;                      // mvScaleFactor[l]
    float scaledPatch;                // (int)(31*scale) as float, :891
    size_t imgOff;                    // byte offset of the level plane inside one image's pyramid block
    size_t keyOff;                    // element offset of the level's candidate list inside one image's block
    int tabOff, tabX, tabY;           // oct-tree lookup tables of the level: [xcode tabX][ycode tabY][xcell tabX][ycell tabY] (u32 each)
};

struct OrbPlan {
    int nlevels;
    int rows, cols;
    int ncells;                       // total FAST cells per image
    int nstrips;                      // total 4-cell strips per image
    int iniTh, minTh;
    int totalKeyCap;                  // sum keyCap
    int totalOut;                     // sum nodeCap
    size_t pyrBytes;                  // bytes of one image's pyramid block
    // level 0 read in place: images b < ext0N take their level-0 plane from the caller's buffer (ext0 + b * ext0Stride, row pitch
    // ext0Pitch) instead of a copy inside the pyramid block.  The kernels' unaligned 8 / 16-byte loads may run a few bytes past a row:
    // harmless inside the caller's batch, which is why the LAST image of a batch is always copied (ext0N <= batch - 1).
    const uint8_t* ext0; size_t ext0Stride; int ext0Pitch, ext0N;
    // the levels' first strips once more, side by side (round 6): the grid-FAST kernel finds a strip's level with ONE scalar load of this array instead of one
    // dependent load per level out of lv[] (8 cache lines); levels >= nlevels hold INT_MAX
    int stripBaseOf[MAXL];
    // what the grid-FAST kernel needs before it can load a strip's tile, precomputed per strip (round 6: a block spent 1.9 of its 9.9 us deriving this on the scalar
    // unit, each of its four waves for itself): 8 dwords per strip of the FULL plan in device memory (the level-0 strips come first, so Detect()'s one-level plan reads
    // the same entries).  [0] level | ci << 8 | cj0 << 16 | ncell << 24 (ncell = 0: nothing to do, ORBextractor.cpp:843 / :852)   [1] iniY | hr << 16
    // [2] iniX0 | wCell << 16   [3] the four cells' interior widths, a byte each   [4] plane pitch   [5] pixel pairs of the strip   [6..7] plane offset in the pyramid block
    const uint32_t* stripTab;
    LevelGeom lv[MAXL];
};

struct ResizeArgs {
    const uint8_t* src; int sw, sh, spitch; size_t sstride;
    const uint8_t* src0 = nullptr; int spitch0 = 0, n0 = 0; size_t sstride0 = 0;      // images b < n0 read their source plane here (level 0 in place)
    uint8_t* dst; int dw, dh, dpitch; size_t dstride;
    double scale_x, scale_y;                     // 1 / ((double)dsize / ssize), as cv::resize computes it
};

struct BlurArgs {
    const uint8_t* src; uint8_t* dst; int w, h, spitch, dpitch; size_t sstride, dstride;
    const uint8_t* src0 = nullptr; int spitch0 = 0, n0 = 0; size_t sstride0 = 0;      // images b < n0 read their source plane here (level 0 in place; any alignment)
    int q[7];                                    // Q8 taps, sum <= 257
    int dtiled = 0;                              // destination in 16 x 8-pixel tiles (tiled_off): the extractor's blurred planes
    // operand tables of the matrix-core form (k_blur7_mfma; orb_kernels.hip blur_mfma_tables): banded Toeplitz blocks of the horizontal pass
    // per 32-column strip, of the vertical pass per 32-row tile, the 0/1 transposition matrix; vconst = 128 sum(q)^2 + 32768.  Null =
    // the register-strip kernel
    const uint4* tabH = nullptr; const uint4* tabV = nullptr; const uint4* ident = nullptr; int vconst = 0;
};

// The blurred planes of the extractor are stored in tiles of 16 x 8 pixels = one 128-byte cache line each (tile rows of pitch / 16 tiles,
// pitch a multiple of 64; plane heights rounded up to 8 rows): their only reader is the descriptor kernel, which fetches a 37 x 37
// window per key-point — 20-24 lines as tiles against ~55 (37 rows x 1.5) in row-major order, and the L1 line fills are what that
// kernel waits for (tools/ta_probe.hip).  Byte offset of pixel (x, y) in a plane:
__host__ __device__ inline size_t tiled_off(int x, int y, int pitch) {
    return (size_t)(y >> 3) * (size_t)pitch * 8 + (size_t)((x >> 4) << 7) + (size_t)(((y & 7) << 4) | (x & 15));
}

}  // namespace myslam_hip
