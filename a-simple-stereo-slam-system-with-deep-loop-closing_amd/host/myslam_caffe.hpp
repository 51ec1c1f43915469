// myslam_caffe.hpp — dependency-free reader of the two files DeepLCD::DeepLCD hands to Caffe (reference src/deeplcd.cpp:10-31,
// include/myslam/deeplcd.h:33: "calc_model/deploy.prototxt" + "calc_model/calc.caffemodel", fetched by get_model.sh):
//   * deploy.prototxt  — protobuf TEXT format: the layer list with its hyper-parameters
//   * calc.caffemodel  — protobuf WIRE format: NetParameter.layer[].blobs[].data (and the V1 `layers` field of older files)
// No protobuf library, no Caffe: a text-format tree parser and a varint / length-delimited walker, restricted to the messages
// and fields named below (caffe.proto field numbers are part of Caffe's on-disk format and never change).
// Result: the layer list as myslam_calc_layer records + the convolution weights as one flat f32 blob in layer order
// (w[OC][IC][K][K], b[OC] per convolution) — what myslam_lcd_create_from_layers takes.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/myslam_hip.h"

namespace myslam_caffe {

// ---------------------------------------------------------------------------------------------- text format
struct TextNode {
    std::vector<std::pair<std::string, std::string>> scalars;       // field: value (quotes stripped)
    std::vector<std::pair<std::string, TextNode>> children;         // field { ... }
    const std::string* get(const std::string& k) const {
        for (auto& s : scalars) if (s.first == k) return &s.second;
        return nullptr;
    }
    std::vector<std::string> all(const std::string& k) const {
        std::vector<std::string> v;
        for (auto& s : scalars) if (s.first == k) v.push_back(s.second);
        return v;
    }
    const TextNode* child(const std::string& k) const {
        for (auto& c : children) if (c.first == k) return &c.second;
        return nullptr;
    }
};

class TextParser {
  public:
    explicit TextParser(const std::string& s) : t(s) {}
    bool parse(TextNode& root) { return body(root, false, 0) && (skip(), p >= t.size()); }

  private:
    const std::string& t; size_t p = 0;
    void skip() {
        for (;;) {
            while (p < t.size() && (t[p] == ' ' || t[p] == '\t' || t[p] == '\n' || t[p] == '\r' || t[p] == ',' || t[p] == ';')) p++;
            if (p < t.size() && t[p] == '#') { while (p < t.size() && t[p] != '\n') p++; continue; }
            return;
        }
    }
    bool ident(std::string& out) {
        skip();
        const size_t b = p;
        while (p < t.size() && (isalnum((unsigned char)t[p]) || t[p] == '_' || t[p] == '.')) p++;
        out = t.substr(b, p - b);
        return p > b;
    }
    bool value(std::string& out) {
        skip();
        if (p < t.size() && (t[p] == '"' || t[p] == '\'')) {
            const char q = t[p++];
            out.clear();
            while (p < t.size() && t[p] != q) { if (t[p] == '\\' && p + 1 < t.size()) p++; out += t[p++]; }
            if (p >= t.size()) return false;
            p++;
            return true;
        }
        const size_t b = p;
        while (p < t.size() && !isspace((unsigned char)t[p]) && t[p] != '}' && t[p] != '{' && t[p] != ',' && t[p] != ';' && t[p] != '#' && t[p] != ']' && t[p] != '[') p++;
        out = t.substr(b, p - b);
        return p > b;
    }
    static constexpr int kMaxDepth = 64;          // a prototxt nests 3-4 levels; user-supplied files must not be able to exhaust the stack
    bool body(TextNode& n, bool braced, int depth) {
        if (depth > kMaxDepth) return false;
        for (;;) {
            skip();
            if (p >= t.size()) return !braced;
            if (t[p] == '}') { if (!braced) return false; p++; return true; }
            std::string key;
            if (!ident(key)) return false;
            skip();
            if (p < t.size() && t[p] == ':') {
                p++; skip();
                if (p < t.size() && t[p] == '{') { p++; TextNode c; if (!body(c, true, depth + 1)) return false; n.children.emplace_back(key, std::move(c)); continue; }
                if (p < t.size() && t[p] == '[') {                  // the short form of a repeated field: `dim: [1, 1, 120, 160]` = four `dim:` entries
                    p++;                                            // (protobuf's TextFormat prints it with use_short_repeated_primitives and always parses it)
                    for (;;) {
                        skip();                                     // (skip() also passes the separating commas)
                        if (p >= t.size()) return false;
                        if (t[p] == ']') { p++; break; }
                        if (t[p] == '{') { p++; TextNode c; if (!body(c, true, depth + 1)) return false; n.children.emplace_back(key, std::move(c)); continue; }
                        std::string v;
                        if (!value(v)) return false;
                        n.scalars.emplace_back(key, v);
                    }
                    continue;
                }
                std::string v;
                if (!value(v)) return false;
                n.scalars.emplace_back(key, v);
            } else if (p < t.size() && t[p] == '{') {
                p++; TextNode c;
                if (!body(c, true, depth + 1)) return false;
                n.children.emplace_back(key, std::move(c));
            } else {
                return false;
            }
        }
    }
};

// ---------------------------------------------------------------------------------------------- wire format
struct Blob { std::vector<int64_t> shape; std::vector<float> data; };
struct WireLayer { std::string name, type; std::vector<Blob> blobs; };

class Wire {
  public:
    Wire(const uint8_t* b, size_t n) : p(b), e(b + n) {}
    bool done() const { return p >= e; }
    bool varint(uint64_t& v) {
        v = 0;
        for (int s = 0; s < 64 && p < e; s += 7) { const uint8_t c = *p++; v |= (uint64_t)(c & 0x7f) << s; if (!(c & 0x80)) return true; }
        return false;
    }
    bool key(uint32_t& field, uint32_t& wt) { uint64_t k; if (!varint(k)) return false; field = (uint32_t)(k >> 3); wt = (uint32_t)(k & 7); return true; }
    bool bytes(const uint8_t*& b, size_t& n) { uint64_t l; if (!varint(l) || l > (uint64_t)(e - p)) return false; b = p; n = (size_t)l; p += l; return true; }
    bool fixed32(uint32_t& v) { if (e - p < 4) return false; memcpy(&v, p, 4); p += 4; return true; }
    bool skip(uint32_t wt) {
        uint64_t v; const uint8_t* b; size_t n; uint32_t f;
        switch (wt) {
            case 0: return varint(v);
            case 1: if (e - p < 8) return false; p += 8; return true;
            case 2: return bytes(b, n);
            case 5: return fixed32(f);
        }
        return false;
    }

  private:
    const uint8_t* p; const uint8_t* e;
};

inline bool parse_blob(const uint8_t* b, size_t n, Blob& out) {
    Wire w(b, n);
    int64_t legacy[4] = {0, 0, 0, 0}; bool has_legacy = false;
    std::vector<double> dd;
    while (!w.done()) {
        uint32_t f, wt;
        if (!w.key(f, wt)) return false;
        if (f == 5 && wt == 2) {                                   // repeated float data = 5 [packed]
            const uint8_t* q; size_t m;
            if (!w.bytes(q, m) || (m & 3)) return false;
            const size_t old = out.data.size();
            out.data.resize(old + m / 4);
            memcpy(out.data.data() + old, q, m);
        } else if (f == 5 && wt == 5) {                            // unpacked float
            uint32_t v; if (!w.fixed32(v)) return false;
            float x; memcpy(&x, &v, 4); out.data.push_back(x);
        } else if (f == 8 && wt == 2) {                            // repeated double double_data = 8 [packed]
            const uint8_t* q; size_t m;
            if (!w.bytes(q, m) || (m & 7)) return false;
            for (size_t i = 0; i < m; i += 8) { double x; memcpy(&x, q + i, 8); dd.push_back(x); }
        } else if (f == 7 && wt == 2) {                            // BlobShape shape = 7 { repeated int64 dim = 1 [packed] }
            const uint8_t* q; size_t m;
            if (!w.bytes(q, m)) return false;
            Wire s(q, m);
            while (!s.done()) {
                uint32_t sf, swt;
                if (!s.key(sf, swt)) return false;
                if (sf == 1 && swt == 2) { const uint8_t* r; size_t k; if (!s.bytes(r, k)) return false; Wire d(r, k); while (!d.done()) { uint64_t v; if (!d.varint(v)) return false; out.shape.push_back((int64_t)v); } }
                else if (sf == 1 && swt == 0) { uint64_t v; if (!s.varint(v)) return false; out.shape.push_back((int64_t)v); }
                else if (!s.skip(swt)) return false;
            }
        } else if (f >= 1 && f <= 4 && wt == 0) {                  // legacy num / channels / height / width
            uint64_t v; if (!w.varint(v)) return false;
            legacy[f - 1] = (int64_t)v; has_legacy = true;
        } else if (!w.skip(wt)) {
            return false;
        }
    }
    if (out.data.empty() && !dd.empty()) for (double x : dd) out.data.push_back((float)x);
    if (out.shape.empty() && has_legacy) out.shape.assign(legacy, legacy + 4);
    return true;
}

// LayerParameter (NetParameter.layer = 100): name = 1, type = 2 (string), blobs = 7
// V1LayerParameter (NetParameter.layers = 2): name = 4, type = 5 (enum: 4 CONVOLUTION, 18 RELU, 17 POOLING, 15 LRN), blobs = 6
inline bool parse_layer(const uint8_t* b, size_t n, bool v1, WireLayer& out) {
    Wire w(b, n);
    const uint32_t f_name = v1 ? 4 : 1, f_type = v1 ? 5 : 2, f_blobs = v1 ? 6 : 7;
    while (!w.done()) {
        uint32_t f, wt;
        if (!w.key(f, wt)) return false;
        if (f == f_name && wt == 2) { const uint8_t* q; size_t m; if (!w.bytes(q, m)) return false; out.name.assign((const char*)q, m); }
        else if (f == f_type && wt == 2) { const uint8_t* q; size_t m; if (!w.bytes(q, m)) return false; out.type.assign((const char*)q, m); }
        else if (f == f_type && wt == 0) { uint64_t v; if (!w.varint(v)) return false; out.type = v == 4 ? "Convolution" : v == 18 ? "ReLU" : v == 17 ? "Pooling" : v == 15 ? "LRN" : "V1:" + std::to_string(v); }
        else if (f == f_blobs && wt == 2) { const uint8_t* q; size_t m; if (!w.bytes(q, m)) return false; Blob bl; if (!parse_blob(q, m, bl)) return false; out.blobs.push_back(std::move(bl)); }
        else if (!w.skip(wt)) return false;
    }
    return true;
}

inline bool parse_caffemodel(const std::vector<uint8_t>& buf, std::vector<WireLayer>& layers) {
    Wire w(buf.data(), buf.size());
    while (!w.done()) {
        uint32_t f, wt;
        if (!w.key(f, wt)) return false;
        if ((f == 100 || f == 2) && wt == 2) {
            const uint8_t* q; size_t m;
            if (!w.bytes(q, m)) return false;
            WireLayer l;
            if (!parse_layer(q, m, f == 2, l)) return false;
            layers.push_back(std::move(l));
        } else if (!w.skip(wt)) {
            return false;
        }
    }
    return true;
}

inline bool read_file(const char* path, std::vector<uint8_t>& out) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (n < 0) { fclose(f); return false; }
    out.resize((size_t)n);
    const size_t got = n ? fread(out.data(), 1, (size_t)n, f) : 0;
    fclose(f);
    return got == (size_t)n;
}

// ---------------------------------------------------------------------------------------------- the CALC net
struct Model {
    int in_c = 0, in_h = 0, in_w = 0;
    std::vector<myslam_calc_layer> layers;
    std::vector<std::string> conv_names;         // names of the convolution layers, in order
    std::vector<float> weights;                  // w[OC][IC][K][K], b[OC] per convolution, in order
};

inline int geti(const TextNode& n, const char* k, int def) { const std::string* s = n.get(k); return s ? atoi(s->c_str()) : def; }
inline float getf(const TextNode& n, const char* k, float def) { const std::string* s = n.get(k); return s ? strtof(s->c_str(), nullptr) : def; }

// returns MYSLAM_OK / MYSLAM_ERR_INVALID (unreadable or malformed) / MYSLAM_ERR_UNSUPPORTED (a layer or parameter outside
// Convolution / ReLU / Pooling MAX / LRN ACROSS_CHANNELS)
inline int parse_prototxt(const std::string& text, Model& m) {
    TextNode root;
    if (!TextParser(text).parse(root)) return MYSLAM_ERR_INVALID;
    std::vector<int> dims;
    for (auto& s : root.all("input_dim")) dims.push_back(atoi(s.c_str()));
    if (const TextNode* sh = root.child("input_shape")) for (auto& s : sh->all("dim")) dims.push_back(atoi(s.c_str()));
    int in_ch = 0;
    // the net is run as a CHAIN: every layer must consume exactly the blob the previous layer produced (in-place layers included); a
    // prototxt with branches, skips or several inputs is refused instead of being silently flattened
    std::string cur_top = root.get("input") ? *root.get("input") : "";
    for (auto& c : root.children) {
        if (c.first != "layer" && c.first != "layers") continue;
        const TextNode& L = c.second;
        std::string type = L.get("type") ? *L.get("type") : "";
        const std::string name = L.get("name") ? *L.get("name") : "";
        for (auto& ch : type) ch = (char)tolower((unsigned char)ch);
        {
            const std::vector<std::string> bottoms = L.all("bottom"), tops = L.all("top");
            if (bottoms.size() > 1 || tops.size() > 1) return MYSLAM_ERR_UNSUPPORTED;
            if (type != "input" && !bottoms.empty() && !cur_top.empty() && bottoms[0] != cur_top) return MYSLAM_ERR_UNSUPPORTED;
            if (!tops.empty()) cur_top = tops[0];
        }
        myslam_calc_layer rec;
        memset(&rec, 0, sizeof(rec));
        if (type == "input") {
            if (const TextNode* ip = L.child("input_param")) if (const TextNode* sh = ip->child("shape")) for (auto& s : sh->all("dim")) dims.push_back(atoi(s.c_str()));
            continue;
        } else if (type == "convolution") {
            const TextNode* cp = L.child("convolution_param");
            if (!cp) return MYSLAM_ERR_INVALID;
            if (cp->get("kernel_h") || cp->get("kernel_w") || cp->get("stride_h") || cp->get("pad_h") || geti(*cp, "group", 1) != 1 || geti(*cp, "dilation", 1) != 1)
                return MYSLAM_ERR_UNSUPPORTED;
            rec.type = MYSLAM_CALC_CONV; rec.num_output = geti(*cp, "num_output", 0); rec.kernel = geti(*cp, "kernel_size", 0);
            rec.stride = geti(*cp, "stride", 1); rec.pad = geti(*cp, "pad", 0);
            if (rec.num_output < 1 || rec.kernel < 1 || rec.stride < 1 || rec.pad < 0) return MYSLAM_ERR_INVALID;
            const std::string* bt = cp->get("bias_term");
            if (bt && (*bt == "false" || *bt == "0")) return MYSLAM_ERR_UNSUPPORTED;
            m.conv_names.push_back(name);
        } else if (type == "relu") {
            if (const TextNode* rp = L.child("relu_param")) if (getf(*rp, "negative_slope", 0.f) != 0.f) return MYSLAM_ERR_UNSUPPORTED;
            rec.type = MYSLAM_CALC_RELU;
        } else if (type == "pooling") {
            const TextNode* pp = L.child("pooling_param");
            if (!pp) return MYSLAM_ERR_INVALID;
            const std::string* pool = pp->get("pool");
            if (pool && *pool != "MAX" && *pool != "0") return MYSLAM_ERR_UNSUPPORTED;
            if (pp->get("global_pooling") || pp->get("kernel_h") || pp->get("stride_h")) return MYSLAM_ERR_UNSUPPORTED;
            rec.type = MYSLAM_CALC_POOL_MAX; rec.kernel = geti(*pp, "kernel_size", 0); rec.stride = geti(*pp, "stride", 1); rec.pad = geti(*pp, "pad", 0);
            if (rec.kernel < 1 || rec.stride < 1) return MYSLAM_ERR_INVALID;
            if (rec.pad != 0) return MYSLAM_ERR_UNSUPPORTED;
        } else if (type == "lrn") {
            const TextNode* lp = L.child("lrn_param");
            rec.type = MYSLAM_CALC_LRN; rec.local_size = 5; rec.alpha = 1.f; rec.beta = 0.75f; rec.k = 1.f;       // caffe.proto defaults
            if (lp) {
                rec.local_size = geti(*lp, "local_size", 5); rec.alpha = getf(*lp, "alpha", 1.f); rec.beta = getf(*lp, "beta", 0.75f); rec.k = getf(*lp, "k", 1.f);
                const std::string* nr = lp->get("norm_region");
                if (nr && *nr != "ACROSS_CHANNELS" && *nr != "0") return MYSLAM_ERR_UNSUPPORTED;
            }
            if (rec.local_size < 1 || !(rec.local_size & 1)) return MYSLAM_ERR_INVALID;
        } else if (type == "flatten" || type == "dropout") {
            continue;                                             // identities at inference (the descriptor is read flat, deeplcd.cpp:80-86)
        } else {
            return MYSLAM_ERR_UNSUPPORTED;
        }
        m.layers.push_back(rec);
        (void)in_ch;
    }
    if (dims.size() != 4 || dims[0] != 1) return MYSLAM_ERR_INVALID;
    m.in_c = dims[1]; m.in_h = dims[2]; m.in_w = dims[3];
    return m.layers.empty() ? MYSLAM_ERR_INVALID : MYSLAM_OK;
}

// convolution weights out of the caffemodel, matched to the prototxt's convolution layers by NAME; shapes are validated
inline int attach_weights(const std::vector<WireLayer>& wl, Model& m) {
    std::map<std::string, const WireLayer*> by_name;
    for (auto& l : wl) if (!l.blobs.empty()) by_name[l.name] = &l;
    int ic = m.in_c;
    size_t ci = 0;
    for (auto& L : m.layers) {
        if (L.type != MYSLAM_CALC_CONV) continue;
        auto it = by_name.find(m.conv_names[ci++]);
        if (it == by_name.end() || it->second->blobs.size() < 2) return MYSLAM_ERR_INVALID;
        const Blob& w = it->second->blobs[0]; const Blob& b = it->second->blobs[1];
        const size_t nw = (size_t)L.num_output * ic * L.kernel * L.kernel;
        if (w.data.size() != nw || b.data.size() != (size_t)L.num_output) return MYSLAM_ERR_INVALID;
        if (w.shape.size() == 4 && (w.shape[0] != L.num_output || w.shape[1] != ic || w.shape[2] != L.kernel || w.shape[3] != L.kernel)) return MYSLAM_ERR_INVALID;
        m.weights.insert(m.weights.end(), w.data.begin(), w.data.end());
        m.weights.insert(m.weights.end(), b.data.begin(), b.data.end());
        ic = L.num_output;
    }
    return MYSLAM_OK;
}

inline int load(const char* prototxt_path, const char* caffemodel_path, Model& m) {
    std::vector<uint8_t> pt, cm;
    if (!prototxt_path || !caffemodel_path || !read_file(prototxt_path, pt) || !read_file(caffemodel_path, cm)) return MYSLAM_ERR_INVALID;
    int rc = parse_prototxt(std::string(pt.begin(), pt.end()), m);
    if (rc) return rc;
    std::vector<WireLayer> wl;
    if (!parse_caffemodel(cm, wl)) return MYSLAM_ERR_INVALID;
    return attach_weights(wl, m);
}

}  // namespace myslam_caffe
