/*
 * calc_oracle.cpp — CPU ORACLE (test infrastructure only).
 *
 * DeepLCD / CALC descriptor path: src/deeplcd.cpp:35-91 (in-tree: blur 7x7 sigma=0 in place,
 * resize to 160x120, u8->f32 /255, forward, 1064 floats, L2 normalise, dot-product score) and the
 * linear database scan src/loopclosing.cpp:124-161.
 *
 * The network itself is Caffe + the CALC deploy.prototxt/caffemodel, neither of which is in the
 * reference tree nor downloadable here => PARITY UNPINNED.  Architecture = SURVEY.md Appendix A.6:
 *   in 1x120x160
 *   conv1 64@5x5 s2 p4 -> 64x62x82, ReLU, maxpool 3x3 s2 (Caffe ceil) -> 64x31x41, LRN(5,1e-4,0.75,k=1)
 *   conv2 128@4x4 s1 p2 -> 128x32x42, ReLU, maxpool 3x3 s2 -> 128x16x21, LRN
 *   conv3 4@3x3 s1 p0 -> 4x14x19, ReLU, flatten -> 1064
 * Layer arithmetic is checked against torch (conv2d / max_pool2d(ceil_mode) / LocalResponseNorm)
 * in tests/test_oracle_kat.py.
 *
 * Weight blob (flat f32): conv1.w[64][1][5][5], conv1.b[64], conv2.w[128][64][4][4], conv2.b[128],
 *                         conv3.w[4][128][3][3], conv3.b[4]   (137 476 floats).
 */
#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

const int IN_H = 120, IN_W = 160;
const size_t NW = 64 * 25 + 64 + 128 * 64 * 16 + 128 + 4 * 128 * 9 + 4;

void conv2d(const float* in, int C, int H, int W, const float* w, const float* b, int OC, int K, int S, int P,
            bool relu, std::vector<float>& out, int& OH, int& OW) {
    OH = (H + 2 * P - K) / S + 1;
    OW = (W + 2 * P - K) / S + 1;
    out.assign((size_t)OC * OH * OW, 0.f);
    for (int oc = 0; oc < OC; oc++) {
        float* o = &out[(size_t)oc * OH * OW];
        for (int ic = 0; ic < C; ic++)
            for (int ky = 0; ky < K; ky++)
                for (int kx = 0; kx < K; kx++) {
                    const float wv = w[(((size_t)oc * C + ic) * K + ky) * K + kx];
                    for (int oy = 0; oy < OH; oy++) {
                        int iy = oy * S + ky - P;
                        if (iy < 0 || iy >= H) continue;
                        const float* irow = in + ((size_t)ic * H + iy) * W;
                        float* orow = o + (size_t)oy * OW;
                        for (int ox = 0; ox < OW; ox++) {
                            int ix = ox * S + kx - P;
                            if (ix < 0 || ix >= W) continue;
                            orow[ox] += wv * irow[ix];
                        }
                    }
                }
        for (int i = 0; i < OH * OW; i++) {
            float v = o[i] + b[oc];
            o[i] = relu ? std::max(v, 0.f) : v;
        }
    }
}

// Caffe max pooling, ceil mode, windows clipped to the input
void maxpool(const std::vector<float>& in, int C, int H, int W, int K, int S, std::vector<float>& out, int& OH, int& OW) {
    OH = (int)ceil((float)(H - K) / S) + 1;
    OW = (int)ceil((float)(W - K) / S) + 1;
    if ((OH - 1) * S >= H) OH--;
    if ((OW - 1) * S >= W) OW--;
    out.assign((size_t)C * OH * OW, 0.f);
    for (int c = 0; c < C; c++)
        for (int oy = 0; oy < OH; oy++)
            for (int ox = 0; ox < OW; ox++) {
                int y0 = oy * S, x0 = ox * S;
                int y1 = std::min(y0 + K, H), x1 = std::min(x0 + K, W);
                float m = -INFINITY;
                for (int y = y0; y < y1; y++)
                    for (int x = x0; x < x1; x++) m = std::max(m, in[((size_t)c * H + y) * W + x]);
                out[((size_t)c * OH + oy) * OW + ox] = m;
            }
}

// Caffe LRN across channels: y = x * (k + alpha/n * sum_{window} x^2)^-beta
void lrn(std::vector<float>& x, int C, int H, int W, int n, float alpha, float beta, float k) {
    std::vector<float> out(x.size());
    const int half = n / 2;
    for (int c = 0; c < C; c++)
        for (int i = 0; i < H * W; i++) {
            float s = 0;
            for (int j = std::max(0, c - half); j <= std::min(C - 1, c + half); j++) {
                float v = x[(size_t)j * H * W + i];
                s += v * v;
            }
            float scale = k + (alpha / n) * s;
            out[(size_t)c * H * W + i] = x[(size_t)c * H * W + i] * powf(scale, -beta);
        }
    x.swap(out);
}

}  // namespace

extern "C" {

size_t orc_calc_nweights(void) { return NW; }

// deeplcd.cpp:43-52 + :55-66 (blur in place when asked — reference quirk 7 — then resize, /255)
int orc_calc_preproc(uint8_t* img, int rows, int cols, int step, int blur_in_place, float* out) {
    std::vector<uint8_t> tmp;
    const uint8_t* src = img;
    int sstep = step;
    if (blur_in_place) {
        int rc = orc_gaussian_blur7_u8(img, cols, rows, step, img, step, 1);
        if (rc) return rc;
    } else {
        tmp.resize((size_t)rows * cols);
        int rc = orc_gaussian_blur7_u8(img, cols, rows, step, tmp.data(), cols, 1);
        if (rc) return rc;
        src = tmp.data(); sstep = cols;
    }
    std::vector<uint8_t> small((size_t)IN_H * IN_W);
    int rc = orc_resize_linear_u8(src, cols, rows, sstep, small.data(), IN_W, IN_H, IN_W);
    if (rc) return rc;
    const float inv255 = (float)(1.0 / 255.0);       // convertTo(CV_32F, 1/255.): f32 multiply (deeplcd.cpp:64)
    for (int i = 0; i < IN_H * IN_W; i++) out[i] = (float)small[i] * inv255;
    return 0;
}

// The layer list as data (what deploy.prototxt says): Caffe semantics per layer, NCHW f32.
int orc_calc_forward_net(const orc_calc_layer* L, int nlayers, const float* weights, size_t nweights, const float* in, float* out1064) {
    std::vector<float> a(in, in + (size_t)IN_H * IN_W), t;
    int C = 1, H = IN_H, W = IN_W;
    const float* w = weights; size_t used = 0;
    for (int i = 0; i < nlayers; i++) {
        const orc_calc_layer& l = L[i];
        int OH, OW;
        if (l.type == 1) {                                                     // Convolution
            const size_t nw = (size_t)l.num_output * C * l.kernel * l.kernel;
            if (used + nw + l.num_output > nweights) return -1;
            conv2d(a.data(), C, H, W, w, w + nw, l.num_output, l.kernel, l.stride, l.pad, false, t, OH, OW);
            w += nw + l.num_output; used += nw + l.num_output;
            a.swap(t); C = l.num_output; H = OH; W = OW;
        } else if (l.type == 2) {                                              // ReLU
            for (auto& v : a) v = std::max(v, 0.f);
        } else if (l.type == 3) {                                              // Pooling MAX (ceil mode)
            maxpool(a, C, H, W, l.kernel, l.stride, t, OH, OW);
            a.swap(t); H = OH; W = OW;
        } else if (l.type == 4) {                                              // LRN across channels
            lrn(a, C, H, W, l.local_size, l.alpha, l.beta, l.k);
        } else {
            return -3;
        }
    }
    if (used != nweights) return -1;
    if ((size_t)C * H * W != 1064) return -2;                                 // deeplcd.cpp:80 assert
    // deeplcd.cpp:88 descriptor /= descriptor.norm()   (f32)
    float ss = 0;
    for (int i = 0; i < 1064; i++) ss += a[i] * a[i];
    float nrm = sqrtf(ss);
    for (int i = 0; i < 1064; i++) out1064[i] = a[i] / nrm;
    return 0;
}

// the SURVEY A.6 list
int orc_calc_forward(const float* weights, size_t nweights, const float* in, float* out1064) {
    if (nweights != NW) return -1;
    const orc_calc_layer L[10] = {{1, 64, 5, 2, 4, 0, 0, 0, 0},  {2, 0, 0, 0, 0, 0, 0, 0, 0}, {3, 0, 3, 2, 0, 0, 0, 0, 0}, {4, 0, 0, 0, 0, 5, 1e-4f, 0.75f, 1.f},
                                  {1, 128, 4, 1, 2, 0, 0, 0, 0}, {2, 0, 0, 0, 0, 0, 0, 0, 0}, {3, 0, 3, 2, 0, 0, 0, 0, 0}, {4, 0, 0, 0, 0, 5, 1e-4f, 0.75f, 1.f},
                                  {1, 4, 3, 1, 0, 0, 0, 0, 0},   {2, 0, 0, 0, 0, 0, 0, 0, 0}};
    return orc_calc_forward_net(L, 10, weights, nweights, in, out1064);
}

float orc_lcd_score(const float* a, const float* b) {      // deeplcd.cpp:35-39
    float s = 0;
    for (int i = 0; i < 1064; i++) s += a[i] * b[i];
    return s;
}

// LoopClosing::DetectLoop, loopclosing.cpp:124-161.  ids ascending (std::map order).
int orc_lcddb_query(const float* db, const uint64_t* ids, int n, const float* q, uint64_t cur_id,
                    float thr_low, uint64_t* best_id, float* max_score, int* cnt) {
    float mx = 0;
    int c = 0;
    uint64_t best = 0;
    for (int i = 0; i < n; i++) {
        if (cur_id - ids[i] < 20) break;                   // :133 (unsigned arithmetic as in the reference)
        float s = orc_lcd_score(q, db + (size_t)i * 1064);
        if (s > mx) { mx = s; best = ids[i]; }             // :136 strict >
        if (s > thr_low) c++;                              // :140
    }
    *best_id = best; *max_score = mx; *cnt = c;
    return 0;
}

}  // extern "C"
