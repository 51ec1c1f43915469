// orb_engine.hip — host side of the ORB extractor: geometry plan, HBM buffers, kernel sequencing and
// the C-ABI entry points that replace class ORBextractor (include/myslam/ORBextractor.h:52-110).
//
// Data layout in HBM (per image b of a batch, all planes 64-byte pitched, 256-byte aligned):
//   pyramid block  : levels 0..L-1 back to back (level 0 is ingested from the caller's buffer)
//   blurred block  : same geometry, 7x7 sigma=2 Gaussian of every level
//   candidates     : per level a u32 list  py<<20 | px<<8 | score   (+ count)
//   sort buffer    : per level the candidates' u32 selection keys in quad-tree bucket order
//   selected keys  : per level <= N+3 payloads in oct-tree list order (+ count)
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "common.h"
#include "orb_plan.h"

namespace myslam_hip {

void launch_resize(const ResizeArgs& a, int batch, hipStream_t s);
void launch_resize_chain(const ResizeArgs* lv, int n, int batch, hipStream_t s);
bool resize_is_little(const ResizeArgs& a, int batch);
int resize_chain_max();
int pyr_head_levels();
void launch_pyr_head(const ResizeArgs* lv, int n, int rows, int cols, uint8_t* dst0, int dpitch0, size_t dstride0, int b0, int batch,
                     uint32_t* p0, int n0, uint32_t* p1, int n1, uint32_t* p2, int n2, uint32_t* p3, int n3, hipStream_t s);
void launch_blur(const BlurArgs& a, int batch, hipStream_t s);
bool blur_uses_strips(const BlurArgs& a);
bool resize_uses_strips(const ResizeArgs& a);
void launch_fast(const OrbPlan& P, const uint8_t* pyr, size_t pyrStride, const uint8_t* maskPyr, uint32_t* cand,
                 int32_t* candCount, const uint32_t* statPrev, uint32_t* statCur, int forceMode, int batch, hipStream_t s);
bool launch_octree(const OrbPlan& P, const uint32_t* cand, const int32_t* candCount, uint32_t* sortbuf, const uint32_t* octTab, uint32_t* selOut,
                   int32_t* selCount, int32_t* status, int batch, uint16_t* order, hipStream_t s, const BlurArgs* blurLv, int nBlur);
bool describe_uses_tile_order(bool have_order, int detectOnly, int batch);
void launch_describe(const OrbPlan& P, const uint8_t* pyr, const uint8_t* blur, size_t pyrStride, const uint32_t* selOut,
                     const int32_t* selCount, myslam_keypoint* kps, uint8_t* desc, int32_t* counts, int32_t* status,
                     int cap, int detectOnly, int batch, uint16_t* order, bool order_ready, int blocks_per_cu, hipStream_t s);
void launch_screen(const OrbPlan& P, const uint8_t* pyr, myslam_keypoint* kin, int n, myslam_keypoint* kout, uint8_t* keep,
                   hipStream_t s);
void launch_calc_desc(const OrbPlan& P, const uint8_t* blur, const myslam_keypoint* kps, int n, uint8_t* desc, hipStream_t s);
void launch_unpack_cands(const uint32_t* cand, int n, int32_t* xs, int32_t* ys, int32_t* sc, hipStream_t s);
void launch_blur_levels(const BlurArgs* lv, int n, int batch, hipStream_t s);
bool blur_mfma_tables(int w, int h, const int q[7], std::vector<uint4>& tab, size_t& offH, size_t& offV);
void blur_mfma_ident(std::vector<uint4>& tab, size_t& offI);
void launch_zero_u32(uint32_t* p0, int n0, uint32_t* p1, int n1, uint32_t* p2, int n2, uint32_t* p3, int n3, hipStream_t s);
void launch_ingest_clear(const uint8_t* src, int rows, int cols, int step, size_t sstride, uint8_t* dst, int dpitch, size_t dstride, int batch,
                         uint32_t* p0, int n0, uint32_t* p1, int n1, uint32_t* p2, int n2, uint32_t* p3, int n3, hipStream_t s);
void launch_ingest(const uint8_t* src, int rows, int cols, int step, size_t sstride, uint8_t* dst, int dpitch,
                   size_t dstride, int batch, hipStream_t s);

static inline int cv_round(float v) { return (int)lrintf(v); }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Gaussian taps in Q8.  kind 0: sigma = 2 (ORBextractor.cpp:966) as OpenCV 3.4.8's getFixedpointGaussianKernel builds it: every
// normalised tap rounded to Q8 on its own (ufixedpoint16(softdouble) = cvRound(v * 256)) -> [18,34,49,55,49,34,18], sum 257 — the
// error-diffusion construction that forces the sum to 256 came with later releases.  kind 1: OpenCV's fixed 7-tap table used when
// sigma <= 0 (deeplcd.cpp:46), sum 256.
void gauss_q8(int kind, int q[7]) {
    if (kind == 1) { const int t[7] = {8, 28, 56, 72, 56, 28, 8}; memcpy(q, t, sizeof(t)); return; }
    double g[7], sum = 0;
    for (int i = 0; i < 7; i++) { double x = i - 3; g[i] = exp(-(x * x) / 8.0); sum += g[i]; }
    for (int i = 0; i < 7; i++) q[i] = (int)lrint(g[i] / sum * 256.0);
}

// cv::resize INTER_LINEAR coefficient tables (OpenCV 3.4 resize.cpp: fx = (dx+0.5)*scale-0.5, 11-bit weights)
[[maybe_unused]] static void resize_tables(int ssize, int dsize, bool is_x, std::vector<int32_t>& ofs, std::vector<int16_t>& coef) {
    const double inv_scale = (double)dsize / ssize;
    const double scale = 1. / inv_scale;
    ofs.resize(dsize); coef.resize(2 * dsize);
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= s;
        if (is_x) {
            if (s < 0) { f = 0; s = 0; }
            if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        }
        ofs[d] = s;
        coef[2 * d] = (int16_t)cv_round((1.f - f) * 2048.f);
        coef[2 * d + 1] = (int16_t)cv_round(f * 2048.f);
    }
}

template <typename T>
static int dev_alloc(T*& p, size_t n) {
    if (p) { (void)hipFree(p); p = nullptr; }
    if (n == 0) return MYSLAM_OK;
    MYSLAM_HIP_CHECK(hipMalloc((void**)&p, n * sizeof(T)));
    return MYSLAM_OK;
}

}  // namespace myslam_hip

using namespace myslam_hip;

struct myslam_orb {
    int nfeatures, nlevels, iniTh, minTh;
    float scaleFactor;
    std::vector<float> scale, invScale;
    std::vector<int> nPerLevel;
    int umax[16];
    hipStream_t stream = nullptr;
    // the Gaussian pyramid only depends on the image pyramid: it runs on an internal stream beside the latency-bound oct-tree
    // kernel, fenced by events against the caller's stream (run_batch)
    hipStream_t aux = nullptr;
    hipEvent_t evFork = nullptr, evJoin = nullptr;
    hipEvent_t evUserFast = nullptr;     // caller's event, recorded after the FAST stage (myslam_orb_set_fast_event)
    hipEvent_t evUserGate = nullptr;     // caller's event, waited for before the FAST stage (myslam_orb_set_fast_gate)

    // plan for the current image size
    int rows = 0, cols = 0;
    OrbPlan full{}, det{};

    // batch buffers
    int batchCap = 0;
    bool maskAlloc = false;
    uint8_t *d_pyr = nullptr, *d_blur = nullptr, *d_mask = nullptr;
    uint32_t* d_cand = nullptr;
    uint32_t* d_sort = nullptr;        // per level the candidates' 32-bit sort entries in bucket order + their path codes
    int32_t *d_candCount = nullptr, *d_selCount = nullptr, *d_status = nullptr;
    uint32_t* d_sel = nullptr;
    uint16_t* d_order = nullptr;       // processing order of the descriptor kernel (tile order of the selected keys, orb_kernels.hip k_sel_order)
    uint32_t* d_octTab = nullptr;      // per-level oct-tree path-code / cell-index tables (see make_plan)
    uint32_t* d_stripTab = nullptr;    // per-strip head of the grid-FAST kernel (see make_plan, orb_plan.h)
    // FAST path selection (orb_kernels.hip FastCtl): two [MAXL][4] counter blocks, the launch accumulates into one and reads the other
    uint32_t* d_fastStat = nullptr; int fastFlip = 0;
    // options (myslam_orb_set_option)
    int optFastMode = -1;              // -1 = chosen per level from the previous launch's statistics, 0 = two-phase, 1 = dense
    int optInternalStream = 1;         // 0 = everything on the caller's stream, 1 = Gaussian pyramid forked after FAST (default), 2 = after the image pyramid
    int optCopyInput = 0;              // 1 = copy every input image into the pyramid block (default 0: level 0 is read in place, see run_batch)
    int optStopAfter = 0;              // debug: stop a batched call after stage 1 ingest / 2 pyramid / 3 oct-tree / 4 blur (0 = run all)
    int tapsSet = 0, taps[7] = {0};    // myslam_orb_set_gauss_taps: replacement of the sigma = 2 Q8 taps
    int optSideBlocksPerCu = 0;        // > 0: the descriptor kernel runs as a limited grid of this many blocks per CU (each walks several work items)
    int optBlurMfma = 0;               // 1 = the Gaussian pyramid on the int8 matrix cores (k_blur7_mfma) for every level that can take it
    uint4* d_blurTab = nullptr; bool blurTabValid = false; size_t blurOffI = 0, blurOffH[MAXL] = {0}, blurOffV[MAXL] = {0}; bool blurLvOk[MAXL] = {false}; int blurVconst = 0;
    int ensure_blur_tables();

    // Host-pointer calls (one frame per call: the drop-ins) replay a HIP graph: the ~35 launches, memsets and the copies of a call are
    // captured once per (image shape, mask, Detect / DetectAndCompute, FAST statistics parity) and replayed with ONE hipGraphLaunch.
    // `gen` changes whenever something a captured graph depends on does (buffers, plan, options, taps, streams): stale graphs are dropped.
    // (the captured hipGraph_t is kept alive beside its executable: on ROCm 7.2 an executable whose source graph has been destroyed reads
    // freed kernel-argument memory as soon as the heap reuses it — found as pixel bytes landing in the candidate counters)
    struct HostGraph { hipGraphExec_t exec = nullptr; hipGraph_t graph = nullptr; uint64_t gen = 0; int rows = 0, cols = 0, step = 0, dcap = 0, calls = 0; bool mask = false; };
    HostGraph hostGraph[2][2];           // [detectOnly][fastFlip]
    uint64_t gen = 1;
    struct Clear { uint32_t* p[4]; int n[4]; } clr{{nullptr, nullptr, nullptr, nullptr}, {0, 0, 0, 0}};      // counters the next level-0 ingest clears (run_batch -> build_pyramids)
    hipStream_t hostStream = nullptr;    // the host-pointer calls' stream when the handle has none (the legacy NULL stream cannot be captured)
    uint8_t* h_pin = nullptr; size_t pinBytes = 0;      // pinned staging: image, mask, counts, key-points, descriptors
    int ensure_pin(size_t bytes);
    void drop_host_graphs();
    // staging for the host-buffer entry points
    uint8_t *d_stageImg = nullptr, *d_stageMask = nullptr; size_t stageImgBytes = 0, stageMaskBytes = 0;
    myslam_keypoint *d_stageKps = nullptr, *d_stageKps2 = nullptr; uint8_t* d_stageDesc = nullptr; uint8_t* d_stageKeep = nullptr;
    int32_t* d_stageCounts = nullptr; int stageCap = 0;
    uint8_t* d_stageOut = nullptr; size_t stageOutBytes = 0, stageDescOff = 0;      // the block d_stageCounts / d_stageKps / d_stageDesc point into

    int make_tables();
    int make_plan(int r, int c);
    int ensure(int batch, int r, int c, bool needMask);
    int ensure_stage(size_t imgBytes, size_t maskBytes, int cap);
    int build_pyramids(const uint8_t* d_imgs, int batch, int step, size_t stride, const uint8_t* d_masks, int nlev);
    ResizeArgs level_resize_args(uint8_t* base, int l) const;
    BlurArgs level_blur_args(int l) const;
    int blur_levels(int batch, int nlev, hipStream_t s);
    void fill_blur_args(BlurArgs* lv, int nlev) const;
    int run_fast(const OrbPlan& P, const uint8_t* maskPyr, int batch);
    int run_batch(const uint8_t* d_imgs, int batch, int r, int c, int step, size_t stride, const uint8_t* d_masks,
                  myslam_keypoint* d_kps, uint8_t* d_desc, int32_t* d_counts, int32_t* d_stat, int cap, bool detectOnly);
    void free_all();
};

// ORBextractor::ORBextractor, src/ORBextractor.cpp:384-445
int myslam_orb::make_tables() {
    scale.assign(nlevels, 1.f); invScale.assign(nlevels, 1.f); nPerLevel.assign(nlevels, 0);
    for (int i = 1; i < nlevels; i++) scale[i] = scale[i - 1] * scaleFactor;
    for (int i = 0; i < nlevels; i++) invScale[i] = 1.0f / scale[i];
    float factor = 1.0f / scaleFactor;
    float nDesired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) { nPerLevel[l] = cv_round(nDesired); sum += nPerLevel[l]; nDesired *= factor; }
    nPerLevel[nlevels - 1] = std::max(nfeatures - sum, 0);
    int v, v0, vmax = (int)floor(HALF_PATCH * sqrtf(2.f) / 2 + 1), vmin = (int)ceil(HALF_PATCH * sqrtf(2.f) / 2);
    const double hp2 = HALF_PATCH * HALF_PATCH;
    for (v = 0; v < 16; v++) umax[v] = 0;
    for (v = 0; v <= vmax; ++v) umax[v] = (int)lrint(sqrt(hp2 - v * v));
    for (v = HALF_PATCH, v0 = 0; v >= vmin; --v) { while (umax[v0] == umax[v0 + 1]) ++v0; umax[v] = v0; ++v0; }
    const int expect[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};   // baked into c_umax
    for (v = 0; v < 16; v++) if (umax[v] != expect[v]) return MYSLAM_ERR_INVALID;
    return MYSLAM_OK;
}

static int ceil_log2(int v) { int b = 0; while ((1 << b) < v) b++; return b; }

int myslam_orb::make_plan(int r, int c) {
    OrbPlan P{};
    P.nlevels = nlevels; P.rows = r; P.cols = c; P.iniTh = iniTh; P.minTh = minTh;
    size_t imgOff = 0, keyOff = 0; int cellBase = 0, outBase = 0, stripBase = 0;
    for (int l = 0; l < nlevels; l++) {
        LevelGeom& g = P.lv[l];
        g.w = cv_round((float)c * invScale[l]);                 // ORBextractor.cpp:1237-1238
        g.h = cv_round((float)r * invScale[l]);
        if (g.w < 1 || g.h < 1) return MYSLAM_ERR_UNSUPPORTED;
        g.pitch = (int)align_up(g.w, 64);
        g.maxBX = g.w - EDGE_THRESHOLD + 3; g.maxBY = g.h - EDGE_THRESHOLD + 3;
        const float width = (float)(g.maxBX - MIN_BORDER), height = (float)(g.maxBY - MIN_BORDER);
        const float W = 30;
        g.nCols = (int)(width / W); g.nRows = (int)(height / W);  // :833-834
        if (g.nCols < 1 || g.nRows < 1) return MYSLAM_ERR_UNSUPPORTED;      // reference divides by zero
        g.wCell = (int)ceilf(width / g.nCols); g.hCell = (int)ceilf(height / g.nRows);
        if (g.wCell > MAX_CELL || g.hCell > MAX_CELL) return MYSLAM_ERR_UNSUPPORTED;
        if (g.maxBX + 3 > 4095 + MIN_BORDER || g.maxBY + 3 > 4095 + MIN_BORDER) return MYSLAM_ERR_UNSUPPORTED;
        g.cellBase = cellBase; cellBase += g.nCols * g.nRows;
        g.stripBase = stripBase; stripBase += g.nRows * ((g.nCols + 3) / 4);
        g.N = nPerLevel[l];
        g.nIni = (int)roundf((float)(g.maxBX - MIN_BORDER) / (g.maxBY - MIN_BORDER));   // :590
        if (g.nIni < 1 || g.nIni > 64) return MYSLAM_ERR_UNSUPPORTED;
        g.hX = (float)(g.maxBX - MIN_BORDER) / g.nIni;                                  // :592
        const int rootW = (int)ceilf(g.hX) + 2, H = g.maxBY - MIN_BORDER;
        g.ndepth = std::min(MAX_DEPTH, ceil_log2(std::max(rootW, H)) + 2);
        g.sortDepth = 0;
        while (g.sortDepth + 1 <= g.ndepth && (g.nIni << (2 * (g.sortDepth + 1))) <= 1024) g.sortDepth++;
        // a cell interior of a x b pixels holds at most ceil(a/2)*ceil(b/2) strict 8-neighbour maxima: no overflow possible
        g.keyCap = (int)std::min<size_t>(262143, std::max<size_t>(256, (size_t)((g.w + 1) / 2) * ((g.h + 1) / 2)));
        g.nodeCap = (std::max(g.N + 4, 4 * g.nIni + 4) + 3) & ~3;
        if (g.nodeCap > 4092) return MYSLAM_ERR_UNSUPPORTED;          // oct-tree node list lives in LDS
        g.outBase = outBase; outBase += g.nodeCap;
        g.scale = scale[l];
        g.scaledPatch = (float)(int)(PATCH_SIZE * scale[l]);                            // :891
        g.imgOff = imgOff; imgOff += align_up((size_t)g.pitch * align_up((size_t)g.h, 8), 256);      // whole 8-row tiles: the blurred planes are tiled (orb_plan.h)
        g.keyOff = keyOff; keyOff += g.keyCap;
    }
    for (int l = 0; l < MAXL; l++) P.stripBaseOf[l] = l < P.nlevels ? P.lv[l].stripBase : INT32_MAX;
    P.ncells = cellBase; P.nstrips = stripBase; P.totalKeyCap = (int)keyOff; P.totalOut = outBase; P.pyrBytes = imgOff;
    // Oct-tree lookup tables.  A key's quad-tree path splits x and y independently (ExtractorNode::DivideNode halves each
    // axis with ceil, ORBextractor.cpp:526-582), so its path code is xcode[px] | ycode[py]: root index (:616) and the x
    // digits in one table, the y digits in the other, both already spread to their bit positions.  xcell/ycell give the
    // FAST grid cell of a key ((px-3)/wCell, ((py-3)/hCell)*nCols): the reference's candidate order is cell-major.
    {
        std::vector<uint32_t> tab;
        for (int l = 0; l < nlevels; l++) {
            LevelGeom& g = P.lv[l];
            g.tabX = g.maxBX - MIN_BORDER + 8; g.tabY = g.maxBY - MIN_BORDER + 8; g.tabOff = (int)tab.size();
            tab.resize(tab.size() + 2 * (size_t)(g.tabX + g.tabY), 0u);
            uint32_t* xcode = tab.data() + g.tabOff; uint32_t* ycode = xcode + g.tabX;
            uint32_t* xcell = ycode + g.tabY; uint32_t* ycell = xcell + g.tabX;
            for (int px = 0; px < g.tabX; px++) {
                int r = (int)((float)px / g.hX);                                         // :616
                r = std::min(std::max(r, 0), g.nIni - 1);
                int UL = (int)(g.hX * (float)r), UR = (int)(g.hX * (float)(r + 1));      // :602-603
                uint32_t code = (uint32_t)r << ROOT_SHIFT;
                for (int k = 1; k <= g.ndepth; k++) {
                    const int mid = UL + ((UR - UL + 1) >> 1);                           // ceil(w/2), :528-529
                    const uint32_t d = px >= mid;
                    code |= d << (ROOT_SHIFT - 2 * k);
                    if (d) UL = mid; else UR = mid;
                }
                xcode[px] = code;
                xcell[px] = (uint32_t)(std::max(px - 3, 0) / g.wCell);
            }
            for (int py = 0; py < g.tabY; py++) {
                int UL = 0, BR = g.maxBY - MIN_BORDER;
                uint32_t code = 0;
                for (int k = 1; k <= g.ndepth; k++) {
                    const int mid = UL + ((BR - UL + 1) >> 1);
                    const uint32_t d = py >= mid;
                    code |= (d << 1) << (ROOT_SHIFT - 2 * k);
                    if (d) UL = mid; else BR = mid;
                }
                ycode[py] = code;
                ycell[py] = (uint32_t)((std::max(py - 3, 0) / g.hCell) * g.nCols);
            }
        }
        int rc = dev_alloc(d_octTab, tab.size());
        if (rc) return rc;
        if ((rc = upload_table(d_octTab, tab.data(), tab.size() * sizeof(uint32_t)))) return rc;
    }
    {   // per-strip head of the grid-FAST kernel (orb_plan.h stripTab): the arithmetic k_fast_strip did per block, ORBextractor.cpp:838-852 for strips of 4 cells
        std::vector<uint32_t> st((size_t)P.nstrips * 8, 0u);
        for (int l = 0; l < nlevels; l++) {
            const LevelGeom& g = P.lv[l];
            const int spr = (g.nCols + 3) / 4;
            for (int strip = 0; strip < g.nRows * spr; strip++) {
                uint32_t* e = st.data() + (size_t)(g.stripBase + strip) * 8;
                const int ci = strip / spr, cj0 = (strip - ci * spr) * 4;
                const int iniY = MIN_BORDER + ci * g.hCell;
                int ncell = 0, pairs = 0; uint32_t wcs = 0; int hr = 0;
                if (iniY < g.maxBY - 3) {
                    hr = std::min(iniY + g.hCell + 6, g.maxBY) - iniY;
                    const int hc = hr - 6;
                    if (hc > 0)
                        for (int c = 0; c < 4; c++) {
                            const int cj = cj0 + c, iniX = MIN_BORDER + cj * g.wCell;
                            int wc = 0;
                            if (cj < g.nCols && iniX < g.maxBX - 6) wc = std::max(0, std::min(iniX + g.wCell + 6, g.maxBX) - iniX - 6);
                            if (wc > 0) ncell = c + 1;
                            wcs |= (uint32_t)wc << (8 * c);
                            pairs += ((wc + 1) >> 1) * hc;
                        }
                }
                if (ncell == 0) wcs = 0;
                e[0] = (uint32_t)l | ((uint32_t)ci << 8) | ((uint32_t)cj0 << 16) | ((uint32_t)ncell << 24);
                e[1] = (uint32_t)iniY | ((uint32_t)hr << 16);
                e[2] = (uint32_t)(MIN_BORDER + cj0 * g.wCell) | ((uint32_t)g.wCell << 16);
                e[3] = wcs; e[4] = (uint32_t)g.pitch; e[5] = (uint32_t)pairs;
                e[6] = (uint32_t)(g.imgOff & 0xffffffffu); e[7] = (uint32_t)(g.imgOff >> 32);
            }
        }
        int rc = dev_alloc(d_stripTab, st.size());
        if (rc) return rc;
        if ((rc = upload_table(d_stripTab, st.data(), st.size() * sizeof(uint32_t)))) return rc;
        P.stripTab = d_stripTab;
    }
    full = P;
    // Detect(): level 0 only, budget = nfeatures (ORBextractor.cpp:1064-1065)
    det = P;
    det.nlevels = 1;
    det.ncells = P.lv[0].nCols * P.lv[0].nRows;
    det.nstrips = P.lv[0].nRows * ((P.lv[0].nCols + 3) / 4);
    det.lv[0].N = nfeatures;
    det.lv[0].nodeCap = (std::max(nfeatures + 4, 4 * P.lv[0].nIni + 4) + 3) & ~3;
    if (det.lv[0].nodeCap > 4092) return MYSLAM_ERR_UNSUPPORTED;
    det.lv[0].outBase = 0;
    det.totalOut = det.lv[0].nodeCap;
    rows = r; cols = c;
    return MYSLAM_OK;
}

int myslam_orb::ensure(int batch, int r, int c, bool needMask) {
    if (r != rows || c != cols) {
        MYSLAM_HIP_CHECK(hipStreamSynchronize(stream));
        int rc = make_plan(r, c);
        if (rc) { rows = cols = 0; return rc; }
        batchCap = 0; gen++; blurTabValid = false;
    }
    const int detOut = det.totalOut;
    const size_t selPer = (size_t)std::max(full.totalOut, detOut);
    if (batch > batchCap) {
        MYSLAM_HIP_CHECK(hipStreamSynchronize(stream));
        gen++;
        int rc;
        if ((rc = dev_alloc(d_pyr, (size_t)batch * full.pyrBytes + 64))) return rc;      // + 64: the resize kernel's 8-byte row loads may run 7 bytes past a row
        if ((rc = dev_alloc(d_blur, (size_t)batch * full.pyrBytes + 4096))) return rc;     // + 4096: a descriptor window's fourth tile column may lie past the last plane
        if ((rc = dev_alloc(d_cand, (size_t)batch * full.totalKeyCap))) return rc;
        if ((rc = dev_alloc(d_sort, (size_t)batch * full.totalKeyCap * 2))) return rc;
        if ((rc = dev_alloc(d_candCount, (size_t)batch * MAXL))) return rc;
        if ((rc = dev_alloc(d_selCount, (size_t)batch * MAXL))) return rc;
        if ((rc = dev_alloc(d_status, (size_t)batch))) return rc;
        if ((rc = dev_alloc(d_sel, (size_t)batch * selPer))) return rc;
        if ((rc = dev_alloc(d_order, (size_t)batch * selPer))) return rc;
        if ((rc = dev_alloc(d_mask, 0))) return rc;
        maskAlloc = false;
        batchCap = batch;
    }
    if (needMask && !maskAlloc) {
        int rc = dev_alloc(d_mask, (size_t)batchCap * full.pyrBytes + 64);
        if (rc) return rc;
        maskAlloc = true;
    }
    return MYSLAM_OK;
}

int myslam_orb::build_pyramids(const uint8_t* d_imgs, int batch, int step, size_t stride, const uint8_t* d_masks, int nlev) {
    const OrbPlan& P = full;
    for (int pass = 0; pass < (d_masks ? 2 : 1); pass++) {
        uint8_t* base = pass ? d_mask : d_pyr;
        const uint8_t* src = pass ? d_masks : d_imgs;
        const int n0 = pass ? 0 : P.ext0N;                     // images read in place have no level-0 copy (masks are always copied)
        int l = 1;
        // a launch with little work (a live stream's frame): ingest, counters and the first levels of all images in ONE launch (k_pyr_head)
        const int nhead = std::min(pyr_head_levels(), nlev - 1);
        if (nhead >= 1 && batch > n0 && (size_t)batch * P.lv[1].w * P.lv[1].h < (size_t)1500000) {
            ScopedProf sp(P_RESIZE, stream);
            ResizeArgs grp[3];
            for (int j = 0; j < nhead; j++) grp[j] = level_resize_args(base, 1 + j);
            grp[0].src = src; grp[0].spitch = step; grp[0].sstride = stride;      // level 1 (and the levels above, through it) from the caller's buffers
            const bool c0 = pass == 0 && clr.n[0] > 0;
            launch_pyr_head(grp, nhead, P.rows, P.cols, base + P.lv[0].imgOff, P.lv[0].pitch, P.pyrBytes, n0, batch,
                            c0 ? clr.p[0] : nullptr, c0 ? clr.n[0] : 0, c0 ? clr.p[1] : nullptr, c0 ? clr.n[1] : 0, c0 ? clr.p[2] : nullptr,
                            c0 ? clr.n[2] : 0, c0 ? clr.p[3] : nullptr, c0 ? clr.n[3] : 0, stream);
            if (c0) clr.n[0] = 0;
            l = 1 + nhead;
        } else if (batch > n0) {
            uint8_t* dst0 = base + (size_t)n0 * P.pyrBytes + P.lv[0].imgOff;
            if (pass == 0 && clr.n[0] > 0) {       // the call's first launch also clears the per-call counters (run_batch)
                launch_ingest_clear(src + (size_t)n0 * stride, P.rows, P.cols, step, stride, dst0, P.lv[0].pitch, P.pyrBytes, batch - n0,
                                    clr.p[0], clr.n[0], clr.p[1], clr.n[1], clr.p[2], clr.n[2], clr.p[3], clr.n[3], stream);
                clr.n[0] = 0;
            } else {
                launch_ingest(src + (size_t)n0 * stride, P.rows, P.cols, step, stride, dst0, P.lv[0].pitch, P.pyrBytes, batch - n0, stream);
            }
        }
        for (; l < nlev;) {                                    // ComputePyramid, ORBextractor.cpp:1235-1246
            ScopedProf sp(P_RESIZE, stream);
            ResizeArgs a = level_resize_args(base, l);
            if (l == 1 && n0 > 0) { a.src0 = d_imgs; a.spitch0 = step; a.sstride0 = stride; a.n0 = n0; }
            // launches with little work (a live stream's frame): up to three consecutive levels share one launch (k_resize_chain) — such a
            // step is bound by the number of its dependent launches
            ResizeArgs grp[4]; int ng = 0;
            const int gmax = resize_chain_max();
            if (gmax > 1 && resize_is_little(a, batch)) {
                // ... as long as the recomputation stays small: 1 / 5 / 21 / 85 interpolations per pixel of the first .. fourth level of a chain
                // (8 images: levels 3 - 5 in one launch are 18 M interpolations and cost more than the two nodes they save)
                static const double kEvals[4] = {1, 5, 21, 85};
                double evals = (double)batch * a.dw * a.dh;
                grp[ng++] = a;
                while (ng < gmax && l + ng < nlev) {
                    const ResizeArgs nx = level_resize_args(base, l + ng);
                    evals += (double)batch * nx.dw * nx.dh * kEvals[ng];
                    if (!resize_is_little(nx, batch) || evals > 6e6) break;
                    grp[ng++] = nx;
                }
            }
            if (ng > 1) { launch_resize_chain(grp, ng, batch, stream); l += ng; }
            else { launch_resize(a, batch, stream); l++; }
        }
    }
    return MYSLAM_OK;
}
ResizeArgs myslam_orb::level_resize_args(uint8_t* base, int l) const {
    const OrbPlan& P = full;
    ResizeArgs a;
    a.src = base + P.lv[l - 1].imgOff; a.sw = P.lv[l - 1].w; a.sh = P.lv[l - 1].h; a.spitch = P.lv[l - 1].pitch; a.sstride = P.pyrBytes;
    a.src0 = nullptr; a.spitch0 = 0; a.n0 = 0; a.sstride0 = 0;
    a.dst = base + P.lv[l].imgOff; a.dw = P.lv[l].w; a.dh = P.lv[l].h; a.dpitch = P.lv[l].pitch; a.dstride = P.pyrBytes;
    a.scale_x = 1. / ((double)a.dw / a.sw); a.scale_y = 1. / ((double)a.dh / a.sh);
    return a;
}
// operand tables of the matrix-core Gaussian: per level of the current plan and tap table (rebuilt when either changes)
int myslam_orb::ensure_blur_tables() {
    if (blurTabValid) return MYSLAM_OK;
    std::vector<uint4> tab;
    int q[7];
    if (tapsSet) memcpy(q, taps, sizeof(q)); else gauss_q8(0, q);
    int sum = 0;
    for (int t = 0; t < 7; t++) sum += q[t];
    blurVconst = sum * 128 * sum + 32768;          // H = 256 hi + lo + 128 sum(q)  (H' = H - 128 sum + 128, lo carries another -128): vertical sum of that constant + the rounding half
    blur_mfma_ident(tab, blurOffI);
    for (int l = 0; l < full.nlevels; l++) blurLvOk[l] = blur_mfma_tables(full.lv[l].w, full.lv[l].h, q, tab, blurOffH[l], blurOffV[l]);
    MYSLAM_HIP_CHECK(hipStreamSynchronize(stream));
    int rc = dev_alloc(d_blurTab, tab.size());
    if (rc) return rc;
    if ((rc = upload_table(d_blurTab, tab.data(), tab.size() * sizeof(uint4)))) return rc;
    blurTabValid = true; gen++;
    return MYSLAM_OK;
}

BlurArgs myslam_orb::level_blur_args(int l) const {
    const OrbPlan& P = full;
    BlurArgs a;
    a.src = d_pyr + P.lv[l].imgOff; a.dst = d_blur + P.lv[l].imgOff;
    a.w = P.lv[l].w; a.h = P.lv[l].h; a.spitch = a.dpitch = P.lv[l].pitch; a.sstride = a.dstride = P.pyrBytes;
    a.src0 = nullptr; a.spitch0 = 0; a.n0 = 0; a.sstride0 = 0;
    a.dtiled = 1;
    if (tapsSet) memcpy(a.q, taps, sizeof(taps)); else gauss_q8(0, a.q);
    if (optBlurMfma && blurTabValid && blurLvOk[l]) { a.tabH = d_blurTab + blurOffH[l]; a.tabV = d_blurTab + blurOffV[l]; a.ident = d_blurTab + blurOffI; a.vconst = blurVconst; }
    return a;
}
void myslam_orb::fill_blur_args(BlurArgs* lv, int nlev) const {
    const OrbPlan& P = full;
    for (int l = 0; l < nlev; l++) {                           // ORBextractor.cpp:965-966 / :1194-1199
        lv[l] = level_blur_args(l);
        if (l == 0 && P.ext0N > 0) { lv[l].src0 = P.ext0; lv[l].spitch0 = P.ext0Pitch; lv[l].sstride0 = P.ext0Stride; lv[l].n0 = P.ext0N; }
    }
}
int myslam_orb::blur_levels(int batch, int nlev, hipStream_t stream) {
    BlurArgs lv[MAXL];
    fill_blur_args(lv, nlev);
    ScopedProf sp(P_BLUR, stream);
    launch_blur_levels(lv, nlev, batch, stream);               // every level in one launch
    return MYSLAM_OK;
}

int myslam_orb::run_batch(const uint8_t* d_imgs, int batch, int r, int c, int step, size_t stride, const uint8_t* d_masks,
                          myslam_keypoint* d_kps, uint8_t* d_desc, int32_t* d_counts, int32_t* d_stat, int cap, bool detectOnly) {
    if (!d_imgs || batch <= 0 || r <= 0 || c <= 0 || step < c || cap <= 0 || !d_kps || !d_counts) return MYSLAM_ERR_INVALID;
    if (!detectOnly && !d_desc) return MYSLAM_ERR_INVALID;
    int rc = ensure(batch, r, c, d_masks != nullptr);
    if (rc) return rc;
    if (optBlurMfma && !detectOnly && (rc = ensure_blur_tables())) return rc;
    const OrbPlan& P = detectOnly ? det : full;
    int32_t* stat = d_stat ? d_stat : d_status;
    if (!d_fastStat) {
        MYSLAM_HIP_CHECK(hipMalloc((void**)&d_fastStat, sizeof(uint32_t) * 2 * MAXL * 4));
        MYSLAM_HIP_CHECK(hipMemsetAsync(d_fastStat, 0, sizeof(uint32_t) * 2 * MAXL * 4, stream));
    }
    // candidate / selection counters, status words and the FAST statistics block this call accumulates into (run_fast) are cleared by the
    // call's first launch, the level-0 ingest (build_pyramids)
    clr = {{reinterpret_cast<uint32_t*>(d_candCount), reinterpret_cast<uint32_t*>(d_selCount), reinterpret_cast<uint32_t*>(stat), d_fastStat + (size_t)fastFlip * MAXL * 4},
           {batch * MAXL, batch * MAXL, batch, MAXL * 4}};
    const int stop = optStopAfter;
    // Level 0 in place: every image but the last of the batch is read where the caller put it (no copy into the pyramid block: 0.94 MB
    // of HBM traffic per 1241 x 376 image saved).  The gather kernels' unaligned loads may run a few bytes past a row, which stays
    // inside the caller's batch for all images but the last — that one is copied.  Needs the register-strip forms of the level-1
    // resize and the level-0 blur (the fallbacks want aligned rows) and a complete call (the debug stops read the copy).
    {
        int n0 = 0;
        // ... and only when the images do not overlap (stride >= rows x pitch): the over-reads of an in-place image then land in the next
        // image or in the gap before it; any other layout (stride 0 = one image repeated, interleaved images) is copied as before
        if (!optCopyInput && stop == 0 && batch > 1 && full.nlevels >= 1 && stride >= (size_t)r * (size_t)step) {
            bool ok = blur_uses_strips(level_blur_args(0));
            if (full.nlevels > 1) ok = ok && resize_uses_strips(level_resize_args(d_pyr, 1));
            if (ok) n0 = batch - 1;
        }
        full.ext0 = det.ext0 = d_imgs; full.ext0Stride = det.ext0Stride = stride; full.ext0Pitch = det.ext0Pitch = step; full.ext0N = det.ext0N = n0;
    }
    if (stop == 1) {
        launch_ingest_clear(d_imgs, full.rows, full.cols, step, stride, d_pyr + full.lv[0].imgOff, full.lv[0].pitch, full.pyrBytes, batch,
                            clr.p[0], clr.n[0], clr.p[1], clr.n[1], clr.p[2], clr.n[2], clr.p[3], clr.n[3], stream);
        clr.n[0] = 0;                      // the list holds the caller's d_stat pointer: it must not outlive the call (build_pyramids resets it too)
        return MYSLAM_OK;
    }
    if ((rc = build_pyramids(d_imgs, batch, step, stride, d_masks, P.nlevels))) return rc;
    if (stop == 2) return MYSLAM_OK;
    // (a call that is being recorded into a HIP graph stays on its stream: on ROCm 7.2 a fork onto the internal stream from a stream that
    // itself joined the capture as a side stream crashes hipStreamEndCapture — found with two handles recorded into one graph)
    hipStreamCaptureStatus capst = hipStreamCaptureStatusNone;
    if (stream && hipStreamIsCapturing(stream, &capst) != hipSuccess) { (void)hipGetLastError(); capst = hipStreamCaptureStatusNone; }
    const int aux_mode = capst == hipStreamCaptureStatusActive ? 0 : optInternalStream;
    const bool fork = aux_mode > 0 && !detectOnly && stop == 0;
    if (fork && !aux) {
        MYSLAM_HIP_CHECK(hipStreamCreateWithFlags(&aux, hipStreamNonBlocking));
        MYSLAM_HIP_CHECK(hipEventCreateWithFlags(&evFork, hipEventDisableTiming));
        MYSLAM_HIP_CHECK(hipEventCreateWithFlags(&evJoin, hipEventDisableTiming));
    }
    auto fork_blur = [&]() -> int {                          // blur on the internal stream, ordered after everything enqueued so far
        MYSLAM_HIP_CHECK(hipEventRecord(evFork, stream));
        MYSLAM_HIP_CHECK(hipStreamWaitEvent(aux, evFork, 0));
        int rc2 = blur_levels(batch, P.nlevels, aux);
        if (rc2) return rc2;
        MYSLAM_HIP_CHECK(hipEventRecord(evJoin, aux));
        return MYSLAM_OK;
    };
    if (fork && aux_mode == 2 && (rc = fork_blur())) return rc;
    if (evUserGate) MYSLAM_HIP_CHECK(hipStreamWaitEvent(stream, evUserGate, 0));
    if ((rc = run_fast(P, d_masks ? d_mask : nullptr, batch))) return rc;
    if (evUserFast) MYSLAM_HIP_CHECK(hipEventRecord(evUserFast, stream));
    if (fork && aux_mode != 2 && (rc = fork_blur())) return rc;
    bool blurred = false;
    {
        ScopedProf sp(P_OCTREE, stream);
        // the descriptor kernel's processing order is written by the oct-tree blocks themselves (one launch less on the chain); for small
        // batches the Gaussian of all levels rides in the same launch too (it depends on the pyramid only)
        BlurArgs blv[MAXL];
        const bool tryBlur = !fork && !detectOnly && stop == 0 && !optBlurMfma;
        if (tryBlur) fill_blur_args(blv, P.nlevels);
        blurred = launch_octree(P, d_cand, d_candCount, d_sort, d_octTab, d_sel, d_selCount, stat, batch,
                                describe_uses_tile_order(d_order != nullptr, detectOnly ? 1 : 0, batch) ? d_order : nullptr, stream,
                                tryBlur ? blv : nullptr, tryBlur ? P.nlevels : 0);
    }
    if (stop == 3) return MYSLAM_OK;
    if (fork) MYSLAM_HIP_CHECK(hipStreamWaitEvent(stream, evJoin, 0));
    else if (!detectOnly && !blurred && (rc = blur_levels(batch, P.nlevels, stream))) return rc;
    if (stop == 4) return MYSLAM_OK;
    {
        ScopedProf sp(P_DESC, stream);
        launch_describe(P, d_pyr, d_blur, full.pyrBytes, d_sel, d_selCount, d_kps, d_desc, d_counts, stat, cap,
                        detectOnly ? 1 : 0, batch, d_order, true, optSideBlocksPerCu, stream);
    }
    MYSLAM_HIP_CHECK(hipGetLastError());
    return MYSLAM_OK;
}

// grid FAST on the handle's stream; the launch reads the statistics block of the previous launch and fills the other one
int myslam_orb::run_fast(const OrbPlan& P, const uint8_t* maskPyr, int batch) {
    uint32_t* cur = d_fastStat + (size_t)fastFlip * MAXL * 4;             // cleared by run_batch's counter launch
    const uint32_t* prev = d_fastStat + (size_t)(fastFlip ^ 1) * MAXL * 4;
    fastFlip ^= 1;
    ScopedProf sp(P_FAST, stream);
    launch_fast(P, d_pyr, full.pyrBytes, maskPyr, d_cand, d_candCount, prev, cur, optFastMode, batch, stream);
    return MYSLAM_OK;
}

int myslam_orb::ensure_pin(size_t bytes) {
    if (bytes <= pinBytes) return MYSLAM_OK;
    if (h_pin) { (void)hipHostFree(h_pin); h_pin = nullptr; pinBytes = 0; }
    gen++;
    const size_t want = (bytes + (bytes >> 2) + 4095) & ~(size_t)4095;
    MYSLAM_HIP_CHECK(hipHostMalloc((void**)&h_pin, want));
    pinBytes = want;
    return MYSLAM_OK;
}

void myslam_orb::drop_host_graphs() {
    for (auto& row : hostGraph)
        for (auto& g : row) { if (g.exec) (void)hipGraphExecDestroy(g.exec); if (g.graph) (void)hipGraphDestroy(g.graph); g = HostGraph(); }
}

int myslam_orb::ensure_stage(size_t imgBytes, size_t maskBytes, int cap) {
    if (imgBytes > stageImgBytes) { gen++; int rc = dev_alloc(d_stageImg, imgBytes); if (rc) return rc; stageImgBytes = imgBytes; }
    if (maskBytes > stageMaskBytes) { gen++; int rc = dev_alloc(d_stageMask, maskBytes); if (rc) return rc; stageMaskBytes = maskBytes; }
    if (cap > stageCap) {
        gen++;
        int rc;
        // counts, key-points and descriptors of a one-frame call share ONE block ([256 bytes][cap key-points][cap descriptors]): the
        // call brings all of it back in one device -> host copy
        const size_t oK = 256, oD = oK + (((size_t)cap * sizeof(myslam_keypoint) + 255) & ~(size_t)255), total = oD + (size_t)cap * 32;
        if ((rc = dev_alloc(d_stageOut, total))) return rc;
        stageOutBytes = total; stageDescOff = oD;
        d_stageCounts = reinterpret_cast<int32_t*>(d_stageOut);
        d_stageKps = reinterpret_cast<myslam_keypoint*>(d_stageOut + oK);
        d_stageDesc = d_stageOut + oD;
        if ((rc = dev_alloc(d_stageKps2, (size_t)cap))) return rc;
        if ((rc = dev_alloc(d_stageKeep, (size_t)cap))) return rc;
        stageCap = cap;
    }
    return MYSLAM_OK;
}

void myslam_orb::free_all() {
    void* ptrs[] = {d_blurTab, d_fastStat, d_octTab, d_stripTab, d_pyr, d_blur, d_mask, d_cand, d_sort, d_candCount, d_selCount, d_status, d_sel, d_order, d_stageImg, d_stageMask,
                    d_stageOut, d_stageKps2, d_stageKeep};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (aux) { (void)hipStreamSynchronize(aux); (void)hipStreamDestroy(aux); (void)hipEventDestroy(evFork); (void)hipEventDestroy(evJoin); aux = nullptr; }
    drop_host_graphs();
    if (hostStream) { (void)hipStreamSynchronize(hostStream); (void)hipStreamDestroy(hostStream); hostStream = nullptr; }
    if (h_pin) { (void)hipHostFree(h_pin); h_pin = nullptr; pinBytes = 0; }
}

// =================================================================================================
extern "C" {

int myslam_orb_create(myslam_orb** out, int nfeatures, float scale_factor, int nlevels, int ini_th_fast, int min_th_fast) {
    if (!out || nfeatures < 1 || nlevels < 1 || nlevels > MAXL || !(scale_factor > 1.0f) || ini_th_fast < 0 ||
        min_th_fast < 1 || min_th_fast > ini_th_fast || ini_th_fast > 255 || scale_factor > 2.5f)
        return MYSLAM_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return MYSLAM_ERR_HIP;   // no CPU fallback
    myslam_orb* h = new myslam_orb();
    h->nfeatures = nfeatures; h->scaleFactor = scale_factor; h->nlevels = nlevels; h->iniTh = ini_th_fast; h->minTh = min_th_fast;
    int rc = h->make_tables();
    if (rc) { delete h; return rc; }
    *out = h;
    return MYSLAM_OK;
}

int myslam_orb_destroy(myslam_orb* h) {
    if (!h) return MYSLAM_ERR_INVALID;
    (void)hipStreamSynchronize(h->stream);
    h->free_all();
    delete h;
    return MYSLAM_OK;
}

int myslam_orb_set_stream(myslam_orb* h, void* s) {
    if (!h) return MYSLAM_ERR_INVALID;
    (void)hipStreamSynchronize(h->stream);
    h->stream = (hipStream_t)s; h->gen++;
    return MYSLAM_OK;
}

int myslam_orb_set_fast_event(myslam_orb* h, void* ev) {
    if (!h) return MYSLAM_ERR_INVALID;
    h->evUserFast = (hipEvent_t)ev; h->gen++;
    return MYSLAM_OK;
}

int myslam_orb_set_fast_gate(myslam_orb* h, void* ev) {
    if (!h) return MYSLAM_ERR_INVALID;
    h->evUserGate = (hipEvent_t)ev; h->gen++;
    return MYSLAM_OK;
}

int myslam_orb_set_option(myslam_orb* h, int option, int value) {
    if (!h) return MYSLAM_ERR_INVALID;
    h->gen++;
    switch (option) {
        case MYSLAM_ORB_OPT_FAST_MODE: if (value < -1 || value > 1) return MYSLAM_ERR_INVALID; h->optFastMode = value; return MYSLAM_OK;
        case MYSLAM_ORB_OPT_INTERNAL_STREAM: if (value < 0 || value > 2) return MYSLAM_ERR_INVALID; h->optInternalStream = value; return MYSLAM_OK;
        case MYSLAM_ORB_OPT_COPY_INPUT: if (value < 0 || value > 1) return MYSLAM_ERR_INVALID; h->optCopyInput = value; return MYSLAM_OK;
        case MYSLAM_ORB_OPT_STOP_AFTER: if (value < 0 || value > 4) return MYSLAM_ERR_INVALID; h->optStopAfter = value; return MYSLAM_OK;
        case MYSLAM_ORB_OPT_BLUR_MFMA: if (value < 0 || value > 1) return MYSLAM_ERR_INVALID; h->optBlurMfma = value; return MYSLAM_OK;
        case MYSLAM_ORB_OPT_SIDE_BLOCKS_PER_CU: if (value < 0 || value > 64) return MYSLAM_ERR_INVALID; h->optSideBlocksPerCu = value; return MYSLAM_OK;
    }
    return MYSLAM_ERR_INVALID;
}

int myslam_orb_set_gauss_taps(myslam_orb* h, const int32_t* q7) {
    if (!h) return MYSLAM_ERR_INVALID;
    h->blurTabValid = false;
    if (!q7) { h->tapsSet = 0; h->gen++; return MYSLAM_OK; }
    // any table whose Q8.8 row sums fit the 16-bit horizontal accumulator (255 * sum <= 65535); the u8 result saturates
    int sum = 0;
    for (int i = 0; i < 7; i++) { if (q7[i] < 0 || q7[i] > 255) return MYSLAM_ERR_INVALID; sum += q7[i]; }
    if (sum < 1 || sum > 257) return MYSLAM_ERR_INVALID;
    for (int i = 0; i < 7; i++) h->taps[i] = q7[i];
    h->tapsSet = 1; h->gen++;
    return MYSLAM_OK;
}

int myslam_orb_get_tables(const myslam_orb* h, float* scale, float* inv_scale, int* fpl, int* umax16) {
    if (!h) return MYSLAM_ERR_INVALID;
    for (int i = 0; i < h->nlevels; i++) {
        if (scale) scale[i] = h->scale[i];
        if (inv_scale) inv_scale[i] = h->invScale[i];
        if (fpl) fpl[i] = h->nPerLevel[i];
    }
    if (umax16) for (int i = 0; i < 16; i++) umax16[i] = h->umax[i];
    return MYSLAM_OK;
}

int myslam_orb_max_keypoints(const myslam_orb* h) {
    if (!h) return MYSLAM_ERR_INVALID;
    // per level the oct-tree returns < N + 3 nodes, except that the very first split of the nIni = round(w/h) root
    // nodes is unconditional (ORBextractor.cpp:645-716) and can already yield 4*nIni; aspect ratios up to 8 are covered
    int s = 0;
    for (int l = 0; l < h->nlevels; l++) s += std::max(h->nPerLevel[l] + 3, 32);
    return std::max(s, std::max(h->nfeatures + 3, 32));
}

int myslam_orb_max_keypoints_for(const myslam_orb* h, int rows, int cols) {
    if (!h || rows < 1 || cols < 1) return MYSLAM_ERR_INVALID;
    // exact bound for this image size: level l returns at most max(N_l + 3, 4 nIni_l) nodes, nIni_l = round of the level's aspect ratio
    int s = 0, first = 0;
    for (int l = 0; l < h->nlevels; l++) {
        const int w = cv_round((float)cols * h->invScale[l]), hh = cv_round((float)rows * h->invScale[l]);
        const int bx = w - EDGE_THRESHOLD + 3 - MIN_BORDER, by = hh - EDGE_THRESHOLD + 3 - MIN_BORDER;
        const int nIni = (bx > 0 && by > 0) ? std::max(1, (int)roundf((float)bx / (float)by)) : 1;
        s += std::max(h->nPerLevel[l] + 3, 4 * nIni);
        if (l == 0) first = std::max(h->nfeatures + 3, 4 * nIni);
    }
    return std::max(std::max(s, first), 32);
}

int myslam_orb_detect_and_compute_batch(myslam_orb* h, const uint8_t* d_imgs, int batch, int rows, int cols, int step,
                                        size_t img_stride, const uint8_t* d_masks, myslam_keypoint* d_kps, uint8_t* d_desc,
                                        int32_t* d_counts, int32_t* d_status, int cap) {
    if (!h) return MYSLAM_ERR_INVALID;
    return h->run_batch(d_imgs, batch, rows, cols, step, img_stride, d_masks, d_kps, d_desc, d_counts, d_status, cap, false);
}

int myslam_orb_detect_batch(myslam_orb* h, const uint8_t* d_imgs, int batch, int rows, int cols, int step, size_t img_stride,
                            const uint8_t* d_masks, myslam_keypoint* d_kps, int32_t* d_counts, int32_t* d_status, int cap) {
    if (!h) return MYSLAM_ERR_INVALID;
    return h->run_batch(d_imgs, batch, rows, cols, step, img_stride, d_masks, d_kps, nullptr, d_counts, d_status, cap, true);
}

static int host_extract(myslam_orb* h, const uint8_t* img, int rows, int cols, int step, const uint8_t* mask, int mask_step,
                        myslam_keypoint* kps, uint8_t* desc, int cap, int* n, bool detectOnly) {
    if (!h || !n) return MYSLAM_ERR_INVALID;
    *n = 0;
    if (!img || rows <= 0 || cols <= 0) return MYSLAM_OK;            // reference: silent return on empty input (:924, :990)
    if (detectOnly && !mask) { /* Detect() returns on empty mask (:990); NULL here means "all 255" */ }
    if (step < cols || cap <= 0 || !kps || (!detectOnly && !desc)) return MYSLAM_ERR_INVALID;
    if (mask && mask_step < cols) return MYSLAM_ERR_INVALID;
    int rc = h->ensure(1, rows, cols, mask != nullptr);          // plan first: the exact slot count depends on the image shape
    if (rc) return rc;
    if (mask && mask_step != step) return MYSLAM_ERR_INVALID;    // masks share the image's pitch inside the engine
    const int dcap = std::max(cap, std::max(h->full.totalOut, h->det.totalOut));
    const size_t imgBytes = (size_t)rows * step, maskBytes = mask ? imgBytes : 0;
    if ((rc = h->ensure_stage(imgBytes, maskBytes, dcap))) return rc;
    // the matrix-core Gaussian's tables are built (allocation, upload, synchronisation) BEFORE a capture can begin: inside one they would
    // break it and pin this key to the eager path (advisor, round 4); building them bumps `gen`, so it also precedes the graph-key test
    if (h->optBlurMfma && !detectOnly && (rc = h->ensure_blur_tables())) return rc;
    // pinned staging: [image][mask][a mirror of the device output block: counts (256 bytes) | key-points | descriptors]
    const size_t oOut = (imgBytes + maskBytes + 255) & ~(size_t)255, oCnt = oOut, oKps = oOut + 256, oDesc = oOut + h->stageDescOff;
    // what a call copies back: everything up to the last slot it can fill (Detect: no descriptors)
    const size_t outBytes = detectOnly ? 256 + sizeof(myslam_keypoint) * (size_t)dcap : h->stageDescOff + (size_t)32 * dcap;
    if ((rc = h->ensure_pin(oOut + h->stageOutBytes))) return rc;
    // the caller's stream, or a private one: host-pointer calls complete before they return, so the stream they run on is invisible —
    // work the caller queued on the handle's own stream that uses the handle's buffers (a _batch call) is waited for first
    hipStream_t caller = h->stream;
    if (!caller) {
        if (!h->hostStream) { MYSLAM_HIP_CHECK(hipStreamCreateWithFlags(&h->hostStream, hipStreamNonBlocking)); h->gen++; }
        MYSLAM_HIP_CHECK(hipStreamSynchronize(nullptr));
    }
    hipStream_t hs = caller ? caller : h->hostStream;
    struct Swap { myslam_orb* h; hipStream_t keep; ~Swap() { h->stream = keep; } } swap{h, caller};
    h->stream = hs;
    memcpy(h->h_pin, img, imgBytes);
    if (mask) memcpy(h->h_pin + imgBytes, mask, maskBytes);
    auto enqueue = [&]() -> int {                               // everything a call puts on the stream (what a graph captures)
        MYSLAM_HIP_CHECK(hipMemcpyAsync(h->d_stageImg, h->h_pin, imgBytes, hipMemcpyHostToDevice, hs));
        if (mask) MYSLAM_HIP_CHECK(hipMemcpyAsync(h->d_stageMask, h->h_pin + imgBytes, maskBytes, hipMemcpyHostToDevice, hs));
        int r2 = h->run_batch(h->d_stageImg, 1, rows, cols, step, imgBytes, mask ? h->d_stageMask : nullptr,
                              h->d_stageKps, h->d_stageDesc, h->d_stageCounts, h->d_stageCounts + 1, dcap, detectOnly);
        if (r2) return r2;
        MYSLAM_HIP_CHECK(hipMemcpyAsync(h->h_pin + oOut, h->d_stageOut, outBytes, hipMemcpyDeviceToHost, hs));      // counts + key-points (+ descriptors): one copy
        return MYSLAM_OK;
    };
    // a graph is replayable when nothing outside the capture takes part: no profiling events, no caller events, a complete call
    const bool graphable = !prof_is_on() && !h->evUserGate && !h->evUserFast && h->optStopAfter == 0;
    myslam_orb::HostGraph& G = h->hostGraph[detectOnly ? 1 : 0][h->fastFlip & 1];
    const bool same = G.gen == h->gen && G.rows == rows && G.cols == cols && G.step == step && G.dcap == dcap && G.mask == (mask != nullptr);
    if (!same) { if (G.exec) (void)hipGraphExecDestroy(G.exec); if (G.graph) (void)hipGraphDestroy(G.graph); G = myslam_orb::HostGraph(); G.gen = h->gen; G.rows = rows; G.cols = cols; G.step = step; G.dcap = dcap; G.mask = mask != nullptr; }
    bool done = false;
    if (graphable && G.exec) {
        if (hipGraphLaunch(G.exec, hs) == hipSuccess) { h->fastFlip ^= 1; done = true; }        // the replay stands for run_fast's ping-pong step too
        else { (void)hipGetLastError(); (void)hipGraphExecDestroy(G.exec); G.exec = nullptr; if (G.graph) { (void)hipGraphDestroy(G.graph); G.graph = nullptr; } G.calls = -1000000; }
    } else if (graphable && G.calls >= 1) {
        // second call with this key (the first ran eagerly: lazy allocations, stream / event creation): capture, instantiate, launch
        const int flip0 = h->fastFlip;
        if (hipStreamBeginCapture(hs, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            const int r2 = enqueue();
            hipGraph_t graph = nullptr;
            const hipError_t e = hipStreamEndCapture(hs, &graph);
            if (r2 == MYSLAM_OK && e == hipSuccess && graph && hipGraphInstantiate(&G.exec, graph, nullptr, nullptr, 0) == hipSuccess &&
                hipGraphLaunch(G.exec, hs) == hipSuccess) {
                done = true; G.graph = graph;
            } else {                       // not capturable here: stay eager for this key
                (void)hipGetLastError();
                if (G.exec) { (void)hipGraphExecDestroy(G.exec); G.exec = nullptr; }
                if (graph) (void)hipGraphDestroy(graph);
                G.calls = -1000000; h->fastFlip = flip0;
            }
        } else {
            (void)hipGetLastError(); G.calls = -1000000;
        }
    }
    if (!done && (rc = enqueue())) return rc;
    G.calls++;
    MYSLAM_HIP_CHECK(hipStreamSynchronize(hs));
    int32_t res[2];
    memcpy(res, h->h_pin + oCnt, sizeof(res));
    if (res[1] != 0) return res[1];
    if (res[0] > cap) { *n = res[0]; return MYSLAM_ERR_CAPACITY; }
    memcpy(kps, h->h_pin + oKps, sizeof(myslam_keypoint) * (size_t)res[0]);
    if (!detectOnly) memcpy(desc, h->h_pin + oDesc, (size_t)32 * res[0]);
    *n = res[0];
    return MYSLAM_OK;
}

int myslam_orb_detect_and_compute(myslam_orb* h, const uint8_t* img, int rows, int cols, int step, const uint8_t* mask,
                                  int mask_step, myslam_keypoint* kps, uint8_t* desc, int cap, int* n) {
    return host_extract(h, img, rows, cols, step, mask, mask_step, kps, desc, cap, n, false);
}

int myslam_orb_detect(myslam_orb* h, const uint8_t* img, int rows, int cols, int step, const uint8_t* mask, int mask_step,
                      myslam_keypoint* kps, int cap, int* n) {
    return host_extract(h, img, rows, cols, step, mask, mask_step, kps, nullptr, cap, n, true);
}

// shared front half of Screen / CalcDescriptors: upload, ComputePyramid (ORBextractor.cpp:1096, :1192)
static int host_pyramid(myslam_orb* h, const uint8_t* img, int rows, int cols, int step, int ncap) {
    int rc = h->ensure(1, rows, cols, false);
    if (rc) return rc;
    if ((rc = h->ensure_stage((size_t)rows * step, 0, ncap))) return rc;
    MYSLAM_HIP_CHECK(hipMemcpyAsync(h->d_stageImg, img, (size_t)rows * step, hipMemcpyHostToDevice, h->stream));
    h->full.ext0N = h->det.ext0N = 0;                          // single staged image: nothing is read in place
    return h->build_pyramids(h->d_stageImg, 1, step, (size_t)rows * step, nullptr, h->nlevels);
}

int myslam_orb_screen_and_compute_params(myslam_orb* h, const uint8_t* img, int rows, int cols, int step,
                                         myslam_keypoint* kps_in, int n_in, myslam_keypoint* kps_out, int cap, int* n_out) {
    if (!h || !n_out) return MYSLAM_ERR_INVALID;
    *n_out = 0;
    if (!img || rows <= 0 || cols <= 0 || n_in <= 0) return MYSLAM_OK;       // :1085-1088 (logs + returns)
    if (!kps_in || !kps_out || step < cols) return MYSLAM_ERR_INVALID;
    for (int i = 0; i < n_in; i++) if (kps_in[i].octave < 0 || kps_in[i].octave >= h->nlevels) return MYSLAM_ERR_INVALID;
    int rc = host_pyramid(h, img, rows, cols, step, n_in);
    if (rc) return rc;
    MYSLAM_HIP_CHECK(hipMemcpyAsync(h->d_stageKps, kps_in, sizeof(myslam_keypoint) * n_in, hipMemcpyHostToDevice, h->stream));
    {
        ScopedProf sp(P_SCREEN, h->stream);
        launch_screen(h->full, h->d_pyr, h->d_stageKps, n_in, h->d_stageKps2, h->d_stageKeep, h->stream);
    }
    std::vector<myslam_keypoint> tmp(n_in);
    std::vector<uint8_t> keep(n_in);
    MYSLAM_HIP_CHECK(hipMemcpyAsync(kps_in, h->d_stageKps, sizeof(myslam_keypoint) * n_in, hipMemcpyDeviceToHost, h->stream));
    MYSLAM_HIP_CHECK(hipMemcpyAsync(tmp.data(), h->d_stageKps2, sizeof(myslam_keypoint) * n_in, hipMemcpyDeviceToHost, h->stream));
    MYSLAM_HIP_CHECK(hipMemcpyAsync(keep.data(), h->d_stageKeep, n_in, hipMemcpyDeviceToHost, h->stream));
    MYSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));
    int m = 0;
    for (int i = 0; i < n_in; i++)                      // out_keypoints.push_back in input order (:1125)
        if (keep[i]) { if (m >= cap) return MYSLAM_ERR_CAPACITY; kps_out[m++] = tmp[i]; }
    *n_out = m;
    return MYSLAM_OK;
}

int myslam_orb_calc_descriptors(myslam_orb* h, const uint8_t* img, int rows, int cols, int step, const myslam_keypoint* kps,
                                int n, uint8_t* desc) {
    if (!h) return MYSLAM_ERR_INVALID;
    if (!img || rows <= 0 || cols <= 0 || n <= 0) return MYSLAM_OK;          // :1183-1186
    if (!kps || !desc || step < cols) return MYSLAM_ERR_INVALID;
    for (int i = 0; i < n; i++) if (kps[i].octave < 0 || kps[i].octave >= h->nlevels) return MYSLAM_ERR_INVALID;
    int rc = host_pyramid(h, img, rows, cols, step, n);
    if (rc) return rc;
    if ((rc = h->blur_levels(1, h->nlevels, h->stream))) return rc;
    MYSLAM_HIP_CHECK(hipMemcpyAsync(h->d_stageKps, kps, sizeof(myslam_keypoint) * n, hipMemcpyHostToDevice, h->stream));
    {
        ScopedProf sp(P_DESC, h->stream);
        launch_calc_desc(h->full, h->d_blur, h->d_stageKps, n, h->d_stageDesc, h->stream);
    }
    MYSLAM_HIP_CHECK(hipMemcpyAsync(desc, h->d_stageDesc, (size_t)32 * n, hipMemcpyDeviceToHost, h->stream));
    MYSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));
    return MYSLAM_OK;
}

// debug taps: a blurred plane (16 x 8-pixel tiles on the device, orb_plan.h) as rows of out_step bytes
static int download_tiled_plane(const uint8_t* d_plane, const LevelGeom& g, uint8_t* out, int out_step, hipStream_t stream) {
    std::vector<uint8_t> tmp((size_t)g.pitch * align_up((size_t)g.h, 8));
    MYSLAM_HIP_CHECK(hipMemcpyAsync(tmp.data(), d_plane, tmp.size(), hipMemcpyDeviceToHost, stream));
    MYSLAM_HIP_CHECK(hipStreamSynchronize(stream));
    for (int y = 0; y < g.h; y++)
        for (int x = 0; x < g.w; x++) out[(size_t)y * out_step + x] = tmp[tiled_off(x, y, g.pitch)];
    return MYSLAM_OK;
}

int myslam_orb_debug_pyramid(myslam_orb* h, const uint8_t* img, int rows, int cols, int step, int level, int blurred,
                             uint8_t* out, int out_step, int* w, int* hgt) {
    if (!h || !img || level < 0 || level >= h->nlevels) return MYSLAM_ERR_INVALID;
    int rc = host_pyramid(h, img, rows, cols, step, 16);
    if (rc) return rc;
    if (blurred && (rc = h->blur_levels(1, h->nlevels, h->stream))) return rc;
    const LevelGeom& g = h->full.lv[level];
    if (w) *w = g.w;
    if (hgt) *hgt = g.h;
    if (out) {
        if (out_step < g.w) return MYSLAM_ERR_INVALID;
        if (blurred) return download_tiled_plane(h->d_blur + g.imgOff, g, out, out_step, h->stream);
        MYSLAM_HIP_CHECK(hipMemcpy2DAsync(out, out_step, h->d_pyr + g.imgOff, g.pitch, g.w, g.h, hipMemcpyDeviceToHost, h->stream));
    }
    MYSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));
    return MYSLAM_OK;
}

int myslam_orb_debug_candidates(myslam_orb* h, const uint8_t* img, int rows, int cols, int step, const uint8_t* mask,
                                int mask_step, int level, int32_t* xs, int32_t* ys, int32_t* scores, int cap, int* n) {
    if (!h || !img || !n || level < 0 || level >= h->nlevels) return MYSLAM_ERR_INVALID;
    if (mask && mask_step != step) return MYSLAM_ERR_INVALID;
    int rc = h->ensure(1, rows, cols, mask != nullptr);
    if (rc) return rc;
    if ((rc = h->ensure_stage((size_t)rows * step, mask ? (size_t)rows * step : 0, 16))) return rc;
    MYSLAM_HIP_CHECK(hipMemcpyAsync(h->d_stageImg, img, (size_t)rows * step, hipMemcpyHostToDevice, h->stream));
    if (mask) MYSLAM_HIP_CHECK(hipMemcpyAsync(h->d_stageMask, mask, (size_t)rows * step, hipMemcpyHostToDevice, h->stream));
    if (!h->d_fastStat) {
        MYSLAM_HIP_CHECK(hipMalloc((void**)&h->d_fastStat, sizeof(uint32_t) * 2 * MAXL * 4));
        MYSLAM_HIP_CHECK(hipMemsetAsync(h->d_fastStat, 0, sizeof(uint32_t) * 2 * MAXL * 4, h->stream));
    }
    launch_zero_u32(reinterpret_cast<uint32_t*>(h->d_candCount), MAXL, h->d_fastStat + (size_t)h->fastFlip * MAXL * 4, MAXL * 4, nullptr, 0, nullptr, 0, h->stream);
    h->full.ext0N = h->det.ext0N = 0;                          // single staged image: nothing is read in place
    if ((rc = h->build_pyramids(h->d_stageImg, 1, step, (size_t)rows * step, mask ? h->d_stageMask : nullptr, h->nlevels))) return rc;
    if ((rc = h->run_fast(h->full, mask ? h->d_mask : nullptr, 1))) return rc;
    int32_t counts[MAXL];
    MYSLAM_HIP_CHECK(hipMemcpyAsync(counts, h->d_candCount, sizeof(counts), hipMemcpyDeviceToHost, h->stream));
    MYSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));
    const LevelGeom& g = h->full.lv[level];
    *n = counts[level];
    if (counts[level] > g.keyCap || counts[level] > cap) return MYSLAM_ERR_CAPACITY;
    int32_t* d_tmp = nullptr;
    const int m = counts[level];
    if (m > 0) {
        MYSLAM_HIP_CHECK(hipMalloc((void**)&d_tmp, sizeof(int32_t) * 3 * m));
        launch_unpack_cands(h->d_cand + g.keyOff, m, d_tmp, d_tmp + m, d_tmp + 2 * m, h->stream);
        MYSLAM_HIP_CHECK(hipMemcpyAsync(xs, d_tmp, 4 * m, hipMemcpyDeviceToHost, h->stream));
        MYSLAM_HIP_CHECK(hipMemcpyAsync(ys, d_tmp + m, 4 * m, hipMemcpyDeviceToHost, h->stream));
        MYSLAM_HIP_CHECK(hipMemcpyAsync(scores, d_tmp + 2 * m, 4 * m, hipMemcpyDeviceToHost, h->stream));
        MYSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));
        (void)hipFree(d_tmp);
    }
    return MYSLAM_OK;
}

// raw readback of the engine's batch buffers after a *_batch call (stage-level parity tests)
//   what: 0 pyramid plane (w*h bytes, tight), 1 blurred plane, 2 candidate count, 3 candidate payloads (u32),
//         4 selected count, 5 selected payloads (u32, oct-tree list order)
int myslam_orb_debug_readback(myslam_orb* h, int what, int b, int level, void* out, size_t cap_bytes, int detect_plan) {
    if (!h || !out || h->rows == 0 || b < 0 || b >= h->batchCap || level < 0 || level >= h->nlevels) return MYSLAM_ERR_INVALID;
    MYSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));
    const OrbPlan& P = detect_plan ? h->det : h->full;
    const LevelGeom& g = P.lv[level];
    switch (what) {
        case 0: case 1: {
            if (cap_bytes < (size_t)g.w * g.h) return MYSLAM_ERR_CAPACITY;
            const uint8_t* base = (what ? h->d_blur : h->d_pyr) + (size_t)b * h->full.pyrBytes + g.imgOff;
            if (what) return download_tiled_plane(base, g, static_cast<uint8_t*>(out), g.w, h->stream);
            MYSLAM_HIP_CHECK(hipMemcpy2D(out, g.w, base, g.pitch, g.w, g.h, hipMemcpyDeviceToHost));
            return MYSLAM_OK;
        }
        case 2: case 4: {
            if (cap_bytes < 4) return MYSLAM_ERR_CAPACITY;
            MYSLAM_HIP_CHECK(hipMemcpy(out, (what == 2 ? h->d_candCount : h->d_selCount) + b * MAXL + level, 4, hipMemcpyDeviceToHost));
            return MYSLAM_OK;
        }
        case 3: {
            if (cap_bytes < (size_t)g.keyCap * 4) return MYSLAM_ERR_CAPACITY;
            MYSLAM_HIP_CHECK(hipMemcpy(out, h->d_cand + (size_t)b * P.totalKeyCap + g.keyOff, (size_t)g.keyCap * 4, hipMemcpyDeviceToHost));
            return MYSLAM_OK;
        }
        case 6: {     // what the last grid-FAST launch measured on this level (FastCtl, orb_kernels.hip): {statistic, pixel pairs, path, 0}
            if (cap_bytes < 16) return MYSLAM_ERR_CAPACITY;
            if (!h->d_fastStat) return MYSLAM_ERR_INVALID;
            MYSLAM_HIP_CHECK(hipMemcpy(out, h->d_fastStat + ((size_t)(h->fastFlip ^ 1) * MAXL + level) * 4, 16, hipMemcpyDeviceToHost));
            return MYSLAM_OK;
        }
        case 5: {
            if (cap_bytes < (size_t)g.nodeCap * 4) return MYSLAM_ERR_CAPACITY;
            MYSLAM_HIP_CHECK(hipMemcpy(out, h->d_sel + (size_t)b * P.totalOut + g.outBase, (size_t)g.nodeCap * 4, hipMemcpyDeviceToHost));
            return MYSLAM_OK;
        }
    }
    return MYSLAM_ERR_INVALID;
}

}  // extern "C"
