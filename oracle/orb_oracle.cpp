/*
 * orb_oracle.cpp — CPU ORACLE (test infrastructure only; never linked into the product).
 *
 * Restates the ORB part of the hot path.  In-tree reference
 * arithmetic is followed line by line (file:line cited per function, all relative to
 * /root/reference); OpenCV 3.4.8 primitives follow SURVEY.md Appendix A.
 * PARITY UNPINNED for the OpenCV-derived pieces (resize, GaussianBlur, FAST score/NMS,
 * fastAtan2): no reference golden vectors exist and OpenCV is not available.
 *
 * Build: g++ -O2 -ffp-contract=off (contraction must stay off: float-derived integers).
 */
#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <list>
#include <utility>
#include <vector>

namespace {

const int PATCH_SIZE = 31;        // ORBextractor.cpp:23
const int HALF_PATCH_SIZE = 15;   // :24
const int EDGE_THRESHOLD = 19;    // :25

const int8_t kPattern[1024] = {
#include "orb_pattern.inc"
};

inline int cv_round_f(float v) { return (int)lrintf(v); }     // cvRound: round-half-even
inline int cv_round_d(double v) { return (int)lrint(v); }

struct Tables {
    std::vector<float> scale, inv_scale;
    std::vector<int> n_per_level;
    int umax[HALF_PATCH_SIZE + 1];
};

// ORBextractor::ORBextractor, ORBextractor.cpp:384-445
void make_tables(const orc_orb_params& p, Tables& t) {
    const int nl = p.nlevels;
    t.scale.assign(nl, 1.0f);
    t.inv_scale.assign(nl, 1.0f);
    for (int i = 1; i < nl; i++) t.scale[i] = t.scale[i - 1] * p.scale_factor;       // :395 (f32*f32)
    for (int i = 0; i < nl; i++) t.inv_scale[i] = 1.0f / t.scale[i];                   // :403
    t.n_per_level.assign(nl, 0);
    float factor = 1.0f / p.scale_factor;                                             // :411
    float nDesired = p.nfeatures * (1 - factor) /
                     (1 - (float)pow((double)factor, (double)nl));                    // :412
    int sum = 0;
    for (int level = 0; level < nl - 1; level++) {
        t.n_per_level[level] = cv_round_f(nDesired);                                  // :417
        sum += t.n_per_level[level];
        nDesired *= factor;
    }
    t.n_per_level[nl - 1] = std::max(p.nfeatures - sum, 0);                           // :421
    // umax, :429-444
    int v, v0;
    int vmax = (int)floor(HALF_PATCH_SIZE * sqrtf(2.f) / 2 + 1);
    int vmin = (int)ceil(HALF_PATCH_SIZE * sqrtf(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= HALF_PATCH_SIZE; ++v) t.umax[v] = 0;
    for (v = 0; v <= vmax; ++v) t.umax[v] = cv_round_d(sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (t.umax[v0] == t.umax[v0 + 1]) ++v0;
        t.umax[v] = v0;
        ++v0;
    }
}

const int* umax_default() {
    static Tables t;
    static bool init = false;
    if (!init) {
        orc_orb_params p = {1000, 1.2f, 8, 20, 7};
        make_tables(p, t);
        init = true;
    }
    return t.umax;
}

// ---------------------------------------------------------------------------------------
// cv::resize(INTER_LINEAR) 8UC1, Appendix A.2 (OpenCV 3.4.8 imgproc/resize.cpp, generic path)
// ---------------------------------------------------------------------------------------
void resize_tables(int ssize, int dsize, std::vector<int>& ofs, std::vector<short>& coef, bool is_x) {
    // scale = 1/inv_scale with inv_scale = (double)dsize/ssize  (cv::resize computes it this way)
    double inv_scale = (double)dsize / ssize;
    double scale = 1. / inv_scale;
    ofs.resize(dsize);
    coef.resize(dsize * 2);
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= s;
        if (is_x) {
            if (s < 0) { f = 0; s = 0; }
            if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        }
        ofs[d] = s;
        coef[2 * d + 0] = (short)cv_round_f((1.f - f) * 2048.f);   // saturate_cast<short>
        coef[2 * d + 1] = (short)cv_round_f(f * 2048.f);
    }
}

int resize_linear(const uint8_t* src, int sw, int sh, int sstep, uint8_t* dst, int dw, int dh, int dstep) {
    if (sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0) return -1;
    std::vector<int> xofs, yofs;
    std::vector<short> alpha, beta;
    resize_tables(sw, dw, xofs, alpha, true);
    resize_tables(sh, dh, yofs, beta, false);
    std::vector<int> row0(dw), row1(dw);
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = std::min(std::max(yofs[dy], 0), sh - 1);        // clip(sy + k, 0, ssize.height)
        int sy1 = std::min(std::max(yofs[dy] + 1, 0), sh - 1);
        const uint8_t* S0 = src + (size_t)sy0 * sstep;
        const uint8_t* S1 = src + (size_t)sy1 * sstep;
        for (int dx = 0; dx < dw; dx++) {
            int sx = xofs[dx];
            int sx1 = std::min(sx + 1, sw - 1);
            int a0 = alpha[2 * dx], a1 = alpha[2 * dx + 1];
            row0[dx] = S0[sx] * a0 + S0[sx1] * a1;
            row1[dx] = S1[sx] * a0 + S1[sx1] * a1;
        }
        int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
        uint8_t* D = dst + (size_t)dy * dstep;
        for (int dx = 0; dx < dw; dx++) {
            int v = (((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2;
            D[dx] = (uint8_t)std::min(std::max(v, 0), 255);
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------
// cv::GaussianBlur 8U 7x7 fixed point, Appendix A.3
// ---------------------------------------------------------------------------------------
// The sigma = 2 taps are one of the definitions that could not be pinned against OpenCV here (the 8U fixed-point kernel construction
// is version specific): they can be replaced at run time (orc_set_gauss_taps) so that a maintainer who finds another rounding with
// tools/dump_opencv_goldens.py changes ONE table on each side.  g_taps_set == 0: the default below.
int g_taps[7]; int g_taps_set = 0;

void gauss_coeffs(int kind, int q[7]) {
    if (kind == 0 && g_taps_set) { for (int i = 0; i < 7; i++) q[i] = g_taps[i]; return; }
    if (kind == 1) {   // sigma<=0, ksize 7: OpenCV small_gaussian_tab[3] = {1,3.5,7,9,7,3.5,1}/32
        const int t[7] = {8, 28, 56, 72, 56, 28, 8};
        for (int i = 0; i < 7; i++) q[i] = t[i];
        return;
    }
    const double sigma = 2.0;
    double g[7], sum = 0;
    for (int i = 0; i < 7; i++) { double x = i - 3; g[i] = exp(-(x * x) / (2 * sigma * sigma)); sum += g[i]; }
    // OpenCV 3.4.8 getFixedpointGaussianKernel: kernel[i] = ufixedpoint16(values[i] / sum) = cvRound(v * 256) per tap, no fix-up of the
    // sum -> [18,34,49,55,49,34,18] (sum 257); the error-diffusion construction with sum 256 belongs to later releases.  (SURVEY
    // Appendix A.3 chose "residue on the centre tap", [..,54,..]; orc_set_gauss_taps still accepts that table.)
    for (int i = 0; i < 7; i++) q[i] = (int)lrint(g[i] / sum * 256.0);
}

inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

int gaussian_blur7(const uint8_t* src, int w, int h, int sstep, uint8_t* dst, int dstep, int kind) {
    if (w <= 0 || h <= 0) return -1;
    int q[7];
    gauss_coeffs(kind, q);
    std::vector<uint16_t> hbuf((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t* S = src + (size_t)y * sstep;
        for (int x = 0; x < w; x++) {
            int acc = 0;
            for (int i = 0; i < 7; i++) acc += q[i] * S[reflect101(x + i - 3, w)];
            hbuf[(size_t)y * w + x] = (uint16_t)std::min(acc, 65535);   // Q8.8 ufixedpoint16 (saturating; 255 * 257 = 65535 still fits)
        }
    }
    for (int y = 0; y < h; y++) {
        uint8_t* D = dst + (size_t)y * dstep;
        for (int x = 0; x < w; x++) {
            uint32_t acc = 0;
            for (int j = 0; j < 7; j++) acc += (uint32_t)q[j] * hbuf[(size_t)reflect101(y + j - 3, h) * w + x];
            D[x] = (uint8_t)std::min<uint32_t>((acc + 32768u) >> 16, 255u);
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------
// FAST-9/16, Appendix A.1; ring = ORBextractor.cpp:365-369
// ---------------------------------------------------------------------------------------
const int kRing[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                          {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

// score = (largest t at which the pixel is still a FAST-9 corner); cornerScore<16> of OpenCV
// without the threshold seed.  corner@t  <=>  score >= t.
inline int fast_score_px(const uint8_t* p, int step) {
    int v = p[0];
    int d[25];
    for (int k = 0; k < 16; k++) d[k] = v - p[kRing[k][0] + kRing[k][1] * step];
    for (int k = 16; k < 25; k++) d[k] = d[k - 16];
    int best_dark = -1000, best_bright = -1000;
    for (int k = 0; k < 16; k++) {
        int mn = d[k], mx = d[k];
        for (int j = 1; j < 9; j++) { mn = std::min(mn, d[k + j]); mx = std::max(mx, d[k + j]); }
        best_dark = std::max(best_dark, mn);       // all 9 have d > t  <=> min d > t
        best_bright = std::max(best_bright, -mx);  // all 9 have -d > t
    }
    return std::max(best_dark, best_bright) - 1;
}

// cv::cornerScore<16> as OpenCV 3.4 writes it (fast_score.cpp, scalar path), INCLUDING the threshold seed of the running maxima
// that the oracle's definition above leaves out.  Used only to show that the seed never matters where the score is used: for a
// pixel that is a corner at `threshold` both return the same value (tests/test_oracle_kat.py).
inline int fast_score_px_seeded(const uint8_t* p, int step, int threshold) {
    const int K = 8, N = K * 3 + 1;
    int v = p[0], d[N];
    for (int k = 0; k < N; k++) d[k] = v - p[kRing[k % 16][0] + kRing[k % 16][1] * step];
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min(d[k + 1], d[k + 2]);
        a = std::min(a, d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, d[k + 4]); a = std::min(a, d[k + 5]); a = std::min(a, d[k + 6]); a = std::min(a, d[k + 7]); a = std::min(a, d[k + 8]);
        a0 = std::max(a0, std::min(a, d[k]));
        a0 = std::max(a0, std::min(a, d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max(d[k + 1], d[k + 2]);
        b = std::max(b, d[k + 3]); b = std::max(b, d[k + 4]); b = std::max(b, d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, d[k + 6]); b = std::max(b, d[k + 7]); b = std::max(b, d[k + 8]);
        b0 = std::min(b0, std::max(b, d[k]));
        b0 = std::min(b0, std::max(b, d[k + 9]));
    }
    return -b0 - 1;
}

// scores of the detection interior [3,w-3)x[3,h-3) of a ROI; zero elsewhere (OpenCV row buffers)
void fast_scores(const uint8_t* img, int w, int h, int step, int th, std::vector<uint8_t>& sc) {
    sc.assign((size_t)w * h, 0);
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            int s = fast_score_px(img + (size_t)y * step + x, step);
            if (s >= th) sc[(size_t)y * w + x] = (uint8_t)s;
        }
}

struct Cand { int x, y, score; };

void fast_detect(const uint8_t* img, int w, int h, int step, int th, std::vector<Cand>& out) {
    out.clear();
    if (w < 7 || h < 7) return;
    std::vector<uint8_t> sc;
    fast_scores(img, w, h, step, th, sc);
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            int s = sc[(size_t)y * w + x];
            if (!s) continue;
            bool ok = true;
            for (int dy = -1; dy <= 1 && ok; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    if (!dx && !dy) continue;
                    if (s <= sc[(size_t)(y + dy) * w + x + dx]) { ok = false; break; }   // strict >
                }
            if (ok) out.push_back({x, y, s});
        }
}

// isFastCorner, ORBextractor.cpp:449-511 (in-tree)
bool is_fast_corner(const uint8_t* img, int step, int x, int y, int threshold) {
    const int K = 8, N = 25;
    int pixel[25];
    for (int k = 0; k < 16; k++) pixel[k] = kRing[k][0] + kRing[k][1] * step;     // :363-380
    for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
    threshold = std::min(std::max(threshold, 0), 255);
    uint8_t tab[512];
    for (int i = -255; i <= 255; i++) tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);
    const uint8_t* ptr = img + (size_t)y * step + x;
    int v = ptr[0];
    const uint8_t* t = &tab[0] - v + 255;
    int d = t[ptr[pixel[0]]] | t[ptr[pixel[8]]];
    if (d == 0) return false;
    d &= t[ptr[pixel[2]]] | t[ptr[pixel[10]]];
    d &= t[ptr[pixel[4]]] | t[ptr[pixel[12]]];
    d &= t[ptr[pixel[6]]] | t[ptr[pixel[14]]];
    if (d == 0) return false;
    d &= t[ptr[pixel[1]]] | t[ptr[pixel[9]]];
    d &= t[ptr[pixel[3]]] | t[ptr[pixel[11]]];
    d &= t[ptr[pixel[5]]] | t[ptr[pixel[13]]];
    d &= t[ptr[pixel[7]]] | t[ptr[pixel[15]]];
    if (d & 1) {
        int vt = v - threshold, count = 0;
        for (int k = 0; k < N; k++) {
            if (ptr[pixel[k]] < vt) { if (++count > K) return true; }
            else count = 0;
        }
    }
    if (d & 2) {
        int vt = v + threshold, count = 0;
        for (int k = 0; k < N; k++) {
            if (ptr[pixel[k]] > vt) { if (++count > K) return true; }
            else count = 0;
        }
    }
    return false;
}

// grid FAST of one level: ORBextractor.cpp:814-883 (== :998-1060 for Detect)
int grid_fast(const uint8_t* img, int cols, int rows, int step, const uint8_t* mask, int mstep,
              int iniTh, int minTh, std::vector<Cand>& out) {
    out.clear();
    const float W = 30;
    const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
    const int maxBorderX = cols - EDGE_THRESHOLD + 3;
    const int maxBorderY = rows - EDGE_THRESHOLD + 3;
    const float width = (float)(maxBorderX - minBorderX);
    const float height = (float)(maxBorderY - minBorderY);
    const int nCols = (int)(width / W);
    const int nRows = (int)(height / W);
    if (nCols < 1 || nRows < 1) return -2;   // reference divides by zero here
    const int wCell = (int)ceil(width / nCols);
    const int hCell = (int)ceil(height / nRows);
    std::vector<Cand> cell;
    for (int i = 0; i < nRows; i++) {
        const int iniY = minBorderY + i * hCell;
        int maxY = iniY + hCell + 6;
        if (iniY >= maxBorderY - 3) continue;
        if (maxY > maxBorderY) maxY = maxBorderY;
        for (int j = 0; j < nCols; j++) {
            const int iniX = minBorderX + j * wCell;
            int maxX = iniX + wCell + 6;
            if (iniX >= maxBorderX - 6) continue;
            if (maxX > maxBorderX) maxX = maxBorderX;
            const uint8_t* roi = img + (size_t)iniY * step + iniX;
            fast_detect(roi, maxX - iniX, maxY - iniY, step, iniTh, cell);
            if (cell.empty()) fast_detect(roi, maxX - iniX, maxY - iniY, step, minTh, cell);
            for (const Cand& c : cell) {
                int px = c.x + j * wCell, py = c.y + i * hCell;     // border-relative (:871-872)
                if (mask && mask[(size_t)py * mstep + px] == 0) continue;   // :873-877 (quirk: no +16)
                out.push_back({px, py, c.score});
            }
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------
// Oct-tree, ORBextractor.cpp:526-810.  Literal std::list restatement.
// ---------------------------------------------------------------------------------------
struct Node {
    int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
    std::vector<int> keys;
    bool noMore = false;
    int serial = 0;                         // creation order (stands in for the heap address, :731)
    std::list<Node>::iterator lit;
};

struct OctCtx {
    const int* xs; const int* ys;
    int serial = 0;
};

void divide_node(const OctCtx& c, const Node& n, Node& n1, Node& n2, Node& n3, Node& n4) {   // :526-582
    const int halfX = (int)ceilf((float)(n.URx - n.ULx) / 2);
    const int halfY = (int)ceilf((float)(n.BRy - n.ULy) / 2);
    n1.ULx = n.ULx; n1.ULy = n.ULy;
    n1.URx = n.ULx + halfX; n1.URy = n.ULy;
    n1.BLx = n.ULx; n1.BLy = n.ULy + halfY;
    n1.BRx = n.ULx + halfX; n1.BRy = n.ULy + halfY;
    n2.ULx = n1.URx; n2.ULy = n1.URy;
    n2.URx = n.URx; n2.URy = n.URy;
    n2.BLx = n1.BRx; n2.BLy = n1.BRy;
    n2.BRx = n.URx; n2.BRy = n.ULy + halfY;
    n3.ULx = n1.BLx; n3.ULy = n1.BLy;
    n3.URx = n1.BRx; n3.URy = n1.BRy;
    n3.BLx = n.BLx; n3.BLy = n.BLy;
    n3.BRx = n1.BRx; n3.BRy = n.BLy;
    n4.ULx = n3.URx; n4.ULy = n3.URy;
    n4.URx = n2.BRx; n4.URy = n2.BRy;
    n4.BLx = n3.BRx; n4.BLy = n3.BRy;
    n4.BRx = n.BRx; n4.BRy = n.BRy;
    for (int k : n.keys) {
        float x = (float)c.xs[k], y = (float)c.ys[k];
        if (x < n1.URx) { if (y < n1.BRy) n1.keys.push_back(k); else n3.keys.push_back(k); }
        else if (y < n1.BRy) n2.keys.push_back(k);
        else n4.keys.push_back(k);
    }
    if (n1.keys.size() == 1) n1.noMore = true;
    if (n2.keys.size() == 1) n2.noMore = true;
    if (n3.keys.size() == 1) n3.noMore = true;
    if (n4.keys.size() == 1) n4.noMore = true;
}

int distribute_octree(const int* xs, const int* ys, const int* scores, int n,
                      int minX, int maxX, int minY, int maxY, int N, std::vector<int>& result) {
    result.clear();
    OctCtx c; c.xs = xs; c.ys = ys;
    if (maxY - minY <= 0) return -2;
    const int nIni = (int)roundf((float)(maxX - minX) / (maxY - minY));    // :590
    if (nIni < 1) return -2;                                               // reference: UB
    const float hX = (float)(maxX - minX) / nIni;                          // :592
    std::list<Node> L;
    std::vector<Node*> ini(nIni);
    for (int i = 0; i < nIni; i++) {                                       // :599-610
        Node ni;
        ni.ULx = (int)(hX * (float)i); ni.ULy = 0;
        ni.URx = (int)(hX * (float)(i + 1)); ni.URy = 0;
        ni.BLx = ni.ULx; ni.BLy = maxY - minY;
        ni.BRx = ni.URx; ni.BRy = maxY - minY;
        ni.serial = c.serial++;
        L.push_back(ni);
        ini[i] = &L.back();
    }
    for (int k = 0; k < n; k++) {                                          // :613-617
        int idx = (int)((float)xs[k] / hX);
        if (idx < 0 || idx >= nIni) return -3;                             // reference: out-of-bounds write
        ini[idx]->keys.push_back(k);
    }
    for (auto lit = L.begin(); lit != L.end();) {                          // :619-632
        if (lit->keys.size() == 1) { lit->noMore = true; ++lit; }
        else if (lit->keys.empty()) lit = L.erase(lit);
        else ++lit;
    }
    bool finish = false;
    typedef std::pair<int, Node*> SP;
    auto sp_less = [](const SP& a, const SP& b) {
        if (a.first != b.first) return a.first < b.first;
        return a.second->serial < b.second->serial;                        // deterministic stand-in for ptr '<'
    };
    std::vector<SP> vSize;
    auto add_children = [&](Node* kids[4], int& nToExpand, bool count) {
        for (int q = 0; q < 4; q++) {
            Node& ch = *kids[q];
            if (ch.keys.size() > 0) {
                ch.serial = c.serial++;
                L.push_front(ch);
                if (ch.keys.size() > 1) {
                    if (count) nToExpand++;
                    vSize.push_back(std::make_pair((int)ch.keys.size(), &L.front()));
                    L.front().lit = L.begin();
                }
            }
        }
    };
    while (!finish) {                                                      // :641-786
        int prevSize = (int)L.size();
        auto lit = L.begin();
        int nToExpand = 0;
        vSize.clear();
        while (lit != L.end()) {
            if (lit->noMore) { ++lit; continue; }
            Node n1, n2, n3, n4;
            divide_node(c, *lit, n1, n2, n3, n4);
            Node* kids[4] = {&n1, &n2, &n3, &n4};
            add_children(kids, nToExpand, true);
            lit = L.erase(lit);
        }
        if ((int)L.size() >= N || (int)L.size() == prevSize) {
            finish = true;
        } else if ((int)L.size() + nToExpand * 3 > N) {
            while (!finish) {
                prevSize = (int)L.size();
                std::vector<SP> prev = vSize;
                vSize.clear();
                std::sort(prev.begin(), prev.end(), sp_less);              // :731
                for (int j = (int)prev.size() - 1; j >= 0; j--) {
                    Node n1, n2, n3, n4;
                    divide_node(c, *prev[j].second, n1, n2, n3, n4);
                    Node* kids[4] = {&n1, &n2, &n3, &n4};
                    int dummy = 0;
                    add_children(kids, dummy, false);
                    L.erase(prev[j].second->lit);
                    if ((int)L.size() >= N) break;
                }
                if ((int)L.size() >= N || (int)L.size() == prevSize) finish = true;
            }
        }
    }
    for (auto& nd : L) {                                                   // :788-807
        int best = nd.keys[0];
        int maxR = scores[best];
        for (size_t k = 1; k < nd.keys.size(); k++)
            if (scores[nd.keys[k]] > maxR) { best = nd.keys[k]; maxR = scores[best]; }
        result.push_back(best);
    }
    return 0;
}

// ---------------------------------------------------------------------------------------
// orientation + descriptor
// ---------------------------------------------------------------------------------------
// cv::fastAtan2 scalar path, Appendix A.4 (degrees)
float fast_atan2(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
    const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
    const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
    const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
    const float eps = (float)2.2204460492503131e-16;
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// IC_Angle, ORBextractor.cpp:27-55
float ic_angle(const uint8_t* img, int step, int x, int y, const int* umax) {
    int m_01 = 0, m_10 = 0;
    const uint8_t* center = img + (size_t)y * step + x;
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0;
        int d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * step], val_minus = center[u - v * step];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return fast_atan2((float)m_01, (float)m_10);
}

// Deterministic sin/cos.  The reference calls cosf/sinf of libm (ORBextractor.cpp:64); libm and the
// GPU math library are not bit-identical, so oracle and HIP path share ONE definition: double
// Cody-Waite reduction + Taylor polynomials evaluated with plain IEEE mul/add in this exact order
// (no FMA), rounded once to float.  |error| < 1e-15 before the final rounding, i.e. the result
// equals the correctly rounded cosf/sinf except in ~2^-29 of the cases.
void det_sincos(float rad, float* s_out, float* c_out) {
    const double x = (double)rad;
    const double kd = rint(x * 0.6366197723675814);
    const int k = (int)kd;
    const double y = (x - kd * 1.5707963267948966) - kd * 6.123233995736766e-17;
    const double y2 = y * y;
    double ps = -7.647163731819816e-13;              // -1/15!
    ps = ps * y2 + 1.6059043836821613e-10;            //  1/13!
    ps = ps * y2 + -2.505210838544172e-08;           // -1/11!
    ps = ps * y2 + 2.7557319223985893e-06;            //  1/9!
    ps = ps * y2 + -0.0001984126984126984;           // -1/7!
    ps = ps * y2 + 0.008333333333333333;            //  1/5!
    ps = ps * y2 + -0.16666666666666666;           // -1/3!
    const double sn = y + y * (y2 * ps);
    double pc = 4.779477332387385e-14;               //  1/16!
    pc = pc * y2 + -1.1470745597729725e-11;           // -1/14!
    pc = pc * y2 + 2.08767569878681e-09;            //  1/12!
    pc = pc * y2 + -2.755731922398589e-07;           // -1/10!
    pc = pc * y2 + 2.48015873015873e-05;            //  1/8!
    pc = pc * y2 + -0.001388888888888889;           // -1/6!
    pc = pc * y2 + 0.041666666666666664;            //  1/4!
    pc = pc * y2 + -0.5;           // -1/2!
    const double cs = 1.0 + y2 * pc;
    double s, c;
    switch (k & 3) {
        case 0: s = sn; c = cs; break;
        case 1: s = cs; c = -sn; break;
        case 2: s = -sn; c = -cs; break;
        default: s = -cs; c = sn; break;
    }
    *s_out = (float)s;
    *c_out = (float)c;
}

// computeOrbDescriptor, ORBextractor.cpp:58-98
void brief(const uint8_t* img, int step, int x, int y, float angle_deg, uint8_t* desc) {
    const float factorPI = (float)(M_PI / 180.f);
    float angle = angle_deg * factorPI;
    float a, b;
    det_sincos(angle, &b, &a);
    const uint8_t* center = img + (size_t)y * step + x;
    const int8_t* pat = kPattern;
    for (int i = 0; i < 32; ++i, pat += 32) {
        int val = 0;
        for (int bit = 0; bit < 8; bit++) {
            const int8_t* pp = pat + 4 * bit;
            float x0 = (float)pp[0], y0 = (float)pp[1], x1 = (float)pp[2], y1 = (float)pp[3];
            int t0 = center[cv_round_f(x0 * b + y0 * a) * step + cv_round_f(x0 * a - y0 * b)];
            int t1 = center[cv_round_f(x1 * b + y1 * a) * step + cv_round_f(x1 * a - y1 * b)];
            val |= (t0 < t1) << bit;
        }
        desc[i] = (uint8_t)val;
    }
}

// ---------------------------------------------------------------------------------------
// pyramid (ORBextractor.cpp:1229-1265)
// ---------------------------------------------------------------------------------------
struct Pyr {
    std::vector<std::vector<uint8_t>> img;
    std::vector<int> w, h;
};

int build_pyramid(const Tables& t, int nlevels, const uint8_t* src, int rows, int cols, int step, Pyr& P) {
    P.img.resize(nlevels); P.w.resize(nlevels); P.h.resize(nlevels);
    for (int l = 0; l < nlevels; l++) {
        float sc = t.inv_scale[l];
        P.w[l] = cv_round_f((float)cols * sc);
        P.h[l] = cv_round_f((float)rows * sc);
        if (P.w[l] < 1 || P.h[l] < 1) return -2;
        P.img[l].resize((size_t)P.w[l] * P.h[l]);
        if (l == 0) {
            for (int y = 0; y < rows; y++) memcpy(&P.img[0][(size_t)y * cols], src + (size_t)y * step, cols);
        } else {
            int rc = resize_linear(P.img[l - 1].data(), P.w[l - 1], P.h[l - 1], P.w[l - 1],
                                   P.img[l].data(), P.w[l], P.h[l], P.w[l]);
            if (rc) return rc;
        }
    }
    return 0;
}

}  // namespace

// =======================================================================================
// C interface
// =======================================================================================
extern "C" {

int orc_orb_tables(const orc_orb_params* p, float* scale, float* inv_scale, int* n_per_level, int* umax16) {
    if (!p || p->nlevels < 1 || p->nlevels > 32) return -1;
    Tables t;
    make_tables(*p, t);
    for (int i = 0; i < p->nlevels; i++) {
        if (scale) scale[i] = t.scale[i];
        if (inv_scale) inv_scale[i] = t.inv_scale[i];
        if (n_per_level) n_per_level[i] = t.n_per_level[i];
    }
    if (umax16) for (int i = 0; i < 16; i++) umax16[i] = t.umax[i];
    return 0;
}

int orc_level_size(int cols, int rows, float inv_scale, int* w, int* h) {
    *w = cv_round_f((float)cols * inv_scale);
    *h = cv_round_f((float)rows * inv_scale);
    return 0;
}

const int8_t* orc_orb_pattern(void) { return kPattern; }

int orc_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstep, uint8_t* dst, int dw, int dh, int dstep) {
    return resize_linear(src, sw, sh, sstep, dst, dw, dh, dstep);
}

int orc_set_gauss_taps(const int* q7) {           // NULL = back to the default; taps must be 0..255 and sum to 1..257 (Q8.8 row sums fit 16 bits)
    if (!q7) { g_taps_set = 0; return 0; }
    int sum = 0;
    for (int i = 0; i < 7; i++) { if (q7[i] < 0 || q7[i] > 255) return -1; sum += q7[i]; }
    if (sum < 1 || sum > 257) return -1;
    for (int i = 0; i < 7; i++) g_taps[i] = q7[i];
    g_taps_set = 1;
    return 0;
}

int orc_fast_score_seeded(const uint8_t* img, int step, int x, int y, int threshold) { return fast_score_px_seeded(img + (size_t)y * step + x, step, threshold); }
int orc_fast_score_px(const uint8_t* img, int step, int x, int y) { return fast_score_px(img + (size_t)y * step + x, step); }

int orc_gaussian_blur7_u8(const uint8_t* src, int w, int h, int sstep, uint8_t* dst, int dstep, int kind) {
    return gaussian_blur7(src, w, h, sstep, dst, dstep, kind);
}

int orc_build_pyramid(const orc_orb_params* p, const uint8_t* img, int rows, int cols, int step, uint8_t** levels) {
    Tables t; make_tables(*p, t);
    Pyr P;
    int rc = build_pyramid(t, p->nlevels, img, rows, cols, step, P);
    if (rc) return rc;
    for (int l = 0; l < p->nlevels; l++) memcpy(levels[l], P.img[l].data(), P.img[l].size());
    return 0;
}

int orc_fast_score_map(const uint8_t* img, int w, int h, int step, int th, uint8_t* out) {
    std::vector<uint8_t> sc;
    fast_scores(img, w, h, step, th, sc);
    memcpy(out, sc.data(), sc.size());
    return 0;
}

int orc_fast_detect(const uint8_t* img, int w, int h, int step, int th,
                    int* xs, int* ys, int* scores, int cap, int* n) {
    std::vector<Cand> c;
    fast_detect(img, w, h, step, th, c);
    *n = (int)c.size();
    if ((int)c.size() > cap) return -4;
    for (size_t i = 0; i < c.size(); i++) { xs[i] = c[i].x; ys[i] = c[i].y; scores[i] = c[i].score; }
    return 0;
}

int orc_is_fast_corner(const uint8_t* img, int step, int x, int y, int th) {
    return is_fast_corner(img, step, x, y, th) ? 1 : 0;
}

int orc_grid_fast(const uint8_t* img, int w, int h, int step, const uint8_t* mask, int mstep,
                  int ini_th, int min_th, int* xs, int* ys, int* scores, int cap, int* n) {
    std::vector<Cand> c;
    int rc = grid_fast(img, w, h, step, mask, mstep, ini_th, min_th, c);
    if (rc) return rc;
    *n = (int)c.size();
    if ((int)c.size() > cap) return -4;
    for (size_t i = 0; i < c.size(); i++) { xs[i] = c[i].x; ys[i] = c[i].y; scores[i] = c[i].score; }
    return 0;
}

int orc_distribute_octree(const int* xs, const int* ys, const int* scores, int n,
                          int minX, int maxX, int minY, int maxY, int N, int* out_idx, int cap, int* nout) {
    std::vector<int> r;
    int rc = distribute_octree(xs, ys, scores, n, minX, maxX, minY, maxY, N, r);
    if (rc) return rc;
    *nout = (int)r.size();
    if ((int)r.size() > cap) return -4;
    for (size_t i = 0; i < r.size(); i++) out_idx[i] = r[i];
    return 0;
}

float orc_fast_atan2(float y, float x) { return fast_atan2(y, x); }
float orc_ic_angle(const uint8_t* img, int step, int x, int y) { return ic_angle(img, step, x, y, umax_default()); }
void orc_sincos(float rad, float* s, float* c) { det_sincos(rad, s, c); }
int orc_brief(const uint8_t* blurred, int step, int x, int y, float angle_deg, uint8_t* desc32) {
    brief(blurred, step, x, y, angle_deg, desc32);
    return 0;
}

// ORBextractor::DetectAndCompute, ORBextractor.cpp:922-985 (+ ComputeKeyPointsOctTree :814-907)
int orc_detect_and_compute(const orc_orb_params* p, const uint8_t* img, int rows, int cols, int step,
                           const uint8_t* mask, int mstep, orc_keypoint* kps, uint8_t* desc, int cap, int* n) {
    *n = 0;
    if (!img || rows <= 0 || cols <= 0) return 0;      // :924 silent return on empty
    Tables t; make_tables(*p, t);
    Pyr P, M;
    int rc = build_pyramid(t, p->nlevels, img, rows, cols, step, P);
    if (rc) return rc;
    if (mask) { rc = build_pyramid(t, p->nlevels, mask, rows, cols, mstep, M); if (rc) return rc; }
    int total = 0;
    std::vector<Cand> cand;
    std::vector<int> xs, ys, sc, sel;
    std::vector<uint8_t> blurred;
    for (int level = 0; level < p->nlevels; level++) {
        const int w = P.w[level], h = P.h[level];
        const uint8_t* L = P.img[level].data();
        rc = grid_fast(L, w, h, w, mask ? M.img[level].data() : nullptr, w, p->ini_th_fast, p->min_th_fast, cand);
        if (rc) return rc;
        const int minB = EDGE_THRESHOLD - 3;
        const int maxBX = w - EDGE_THRESHOLD + 3, maxBY = h - EDGE_THRESHOLD + 3;
        xs.resize(cand.size()); ys.resize(cand.size()); sc.resize(cand.size());
        for (size_t i = 0; i < cand.size(); i++) { xs[i] = cand[i].x; ys[i] = cand[i].y; sc[i] = cand[i].score; }
        rc = distribute_octree(xs.data(), ys.data(), sc.data(), (int)cand.size(), minB, maxBX, minB, maxBY,
                               t.n_per_level[level], sel);
        if (rc) return rc;
        if (sel.empty()) continue;
        if (total + (int)sel.size() > cap) return -4;
        const int scaledPatchSize = (int)(PATCH_SIZE * t.scale[level]);           // :891
        blurred.resize((size_t)w * h);
        gaussian_blur7(L, w, h, w, blurred.data(), w, 0);                         // :965-966
        for (size_t i = 0; i < sel.size(); i++) {
            const Cand& c = cand[sel[i]];
            orc_keypoint& k = kps[total + i];
            int px = c.x + minB, py = c.y + minB;                                 // :897-898
            k.x = (float)px; k.y = (float)py;
            k.size = (float)scaledPatchSize;
            k.response = (float)c.score;
            k.octave = level;
            k.class_id = -1;
            k.angle = ic_angle(L, w, px, py, t.umax);                             // :905-906
            brief(blurred.data(), w, px, py, k.angle, desc + (size_t)(total + i) * 32);   // :970
            if (level != 0) { k.x *= t.scale[level]; k.y *= t.scale[level]; }     // :975-981
        }
        total += (int)sel.size();
    }
    *n = total;
    return 0;
}

// ORBextractor::Detect, ORBextractor.cpp:989-1074
int orc_detect(const orc_orb_params* p, const uint8_t* img, int rows, int cols, int step,
               const uint8_t* mask, int mstep, orc_keypoint* kps, int cap, int* n) {
    *n = 0;
    if (!img || rows <= 0 || cols <= 0) return 0;
    std::vector<Cand> cand;
    int rc = grid_fast(img, cols, rows, step, mask, mstep, p->ini_th_fast, p->min_th_fast, cand);
    if (rc) return rc;
    const int minB = EDGE_THRESHOLD - 3;
    std::vector<int> xs(cand.size()), ys(cand.size()), sc(cand.size()), sel;
    for (size_t i = 0; i < cand.size(); i++) { xs[i] = cand[i].x; ys[i] = cand[i].y; sc[i] = cand[i].score; }
    rc = distribute_octree(xs.data(), ys.data(), sc.data(), (int)cand.size(), minB, cols - EDGE_THRESHOLD + 3,
                           minB, rows - EDGE_THRESHOLD + 3, p->nfeatures, sel);
    if (rc) return rc;
    if ((int)sel.size() > cap) return -4;
    for (size_t i = 0; i < sel.size(); i++) {
        const Cand& c = cand[sel[i]];
        orc_keypoint& k = kps[i];
        k.x = (float)(c.x + minB); k.y = (float)(c.y + minB);
        k.size = 7.f; k.angle = -1.f; k.response = (float)c.score; k.octave = 0; k.class_id = -1;   // cv::FAST KeyPoint
    }
    *n = (int)sel.size();
    return 0;
}

// ORBextractor::ScreenAndComputeKPsParams, ORBextractor.cpp:1083-1129
int orc_screen(const orc_orb_params* p, const uint8_t* img, int rows, int cols, int step,
               orc_keypoint* kin, int n_in, orc_keypoint* kout, int cap, int* n_out) {
    *n_out = 0;
    if (!img || n_in <= 0) return 0;
    Tables t; make_tables(*p, t);
    Pyr P;
    int rc = build_pyramid(t, p->nlevels, img, rows, cols, step, P);
    if (rc) return rc;
    int m = 0;
    for (int i = 0; i < n_in; i++) {
        orc_keypoint& k = kin[i];
        int level = k.octave;
        if (level < 0 || level >= p->nlevels) return -1;
        float scale = t.scale[level];
        k.x = k.x / scale; k.y = k.y / scale;                                       // :1104
        const int w = P.w[level], h = P.h[level];
        if (!(k.y - EDGE_THRESHOLD >= 0 && k.y + EDGE_THRESHOLD < h &&
              k.x - EDGE_THRESHOLD >= 0 && k.x + EDGE_THRESHOLD < w)) {
            k.x *= scale; k.y *= scale; continue;
        }
        int px = cv_round_f(k.x), py = cv_round_f(k.y);
        if (!is_fast_corner(P.img[level].data(), w, px, py, p->min_th_fast)) {
            k.x *= scale; k.y *= scale; continue;
        }
        k.angle = ic_angle(P.img[level].data(), w, px, py, t.umax);                 // :1118
        k.size = PATCH_SIZE * t.scale[level];                                       // :1121
        k.x *= scale; k.y *= scale;                                                 // :1123
        if (m >= cap) return -4;
        kout[m++] = k;
    }
    *n_out = m;
    return 0;
}

// ORBextractor::CalcDescriptors, ORBextractor.cpp:1180-1226
int orc_calc_descriptors(const orc_orb_params* p, const uint8_t* img, int rows, int cols, int step,
                         const orc_keypoint* kps, int n, uint8_t* desc) {
    if (!img || n <= 0) return 0;
    Tables t; make_tables(*p, t);
    Pyr P;
    int rc = build_pyramid(t, p->nlevels, img, rows, cols, step, P);
    if (rc) return rc;
    std::vector<std::vector<uint8_t>> work(p->nlevels);
    for (int l = 0; l < p->nlevels; l++) {
        work[l].resize(P.img[l].size());
        gaussian_blur7(P.img[l].data(), P.w[l], P.h[l], P.w[l], work[l].data(), P.w[l], 0);
    }
    for (int i = 0; i < n; i++) {
        orc_keypoint k = kps[i];
        int level = k.octave;
        if (level < 0 || level >= p->nlevels) return -1;
        float scale = t.scale[level];
        k.x = k.x / scale; k.y = k.y / scale;                                       // :1218
        brief(work[level].data(), P.w[level], cv_round_f(k.x), cv_round_f(k.y), k.angle, desc + (size_t)i * 32);
    }
    return 0;
}

}  // extern "C"
