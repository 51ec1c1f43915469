// myslam_png.hpp — dependency-free PNG reader for the KITTI grey images, header only.
//   myslam::io::ReadPngGray(path, pixels, rows, cols)   what the reference gets from cv::imread(file, cv::IMREAD_GRAYSCALE) for the
//                                                       8-bit single-channel PNGs of KITTI image_0 / image_1   app/run_kitti_stereo.cpp:66-67
// Supported: every non-interlaced PNG — grey (1 / 2 / 4 / 8 / 16 bit), grey + alpha, RGB, RGBA (8 / 16 bit), palette (1 / 2 / 4 / 8 bit) —
// reduced to 8-bit grey the way cv::imread(..., IMREAD_GRAYSCALE) of OpenCV 3.4.8 asks libpng 1.6 to (modules/imgcodecs/src/grfmt_png.cpp:
// png_set_strip_16, png_set_strip_alpha, png_set_palette_to_rgb, png_set_expand_gray_1_2_4_to_8, png_set_rgb_to_gray(1, 0.299, 0.587)):
//   low-bit grey is scaled to 0..255, a palette index becomes its PLTE colour, alpha (and tRNS) is dropped, colour becomes
//   grey = (9797 R + 19234 G + 3737 B) >> 15 for 8-bit samples (libpng's coefficients 0.299 / 0.587 * 32768 TRUNCATED, blue = the rest, the
//   "historical approach which simply truncates"; R = G = B passes through) and (… + 16384) >> 15 for 16-bit samples, and 16-bit results
//   keep their high byte.  These colour rules are a recollection of libpng's pngrtran.c (not checkable here: parity unpinned); KITTI's
//   image_0 / image_1 are 8-bit grey, where none of them applies.  Interlaced (Adam7) files return false.
// zlib stream: stored, fixed and dynamic Huffman blocks (RFC 1950 / 1951); CRC-32 of every chunk and the Adler-32 are verified.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>
#include <new>
#include <stdexcept>

namespace myslam {
namespace io {
namespace png_detail {

struct BitReader {                                           // LSB-first bit stream with a 64-bit window
    const uint8_t* p; size_t n, pos = 0; uint64_t buf = 0; int cnt = 0; bool bad = false;
    BitReader(const uint8_t* p_, size_t n_) : p(p_), n(n_) {}
    void refill() {                                          // afterwards cnt >= 56 unless the input is exhausted
        if (pos + 8 <= n) {
            uint64_t w;
            std::memcpy(&w, p + pos, 8);                     // little-endian host (x86-64 / the MI355X boxes)
            buf |= w << cnt;
            const int take = (63 - cnt) >> 3;
            pos += (size_t)take; cnt += take * 8;
        } else {
            while (cnt <= 56 && pos < n) { buf |= (uint64_t)p[pos++] << cnt; cnt += 8; }
        }
    }
    uint32_t bits(int k) {                                   // k <= 16
        if (cnt < k) { refill(); if (cnt < k) { bad = true; return 0; } }
        const uint32_t v = (uint32_t)(buf & ((1u << k) - 1));
        buf >>= k; cnt -= k;
        return v;
    }
    void align() {                                           // drop the rest of the current byte, give whole unread bytes back
        const int drop = cnt & 7;
        buf >>= drop; cnt -= drop;
        pos -= (size_t)(cnt >> 3); buf = 0; cnt = 0;
    }
};

struct Huffman {                                             // canonical code; decode = one 10-bit table look-up, long codes bit by bit
    enum { FAST = 10 };
    uint16_t count[16] = {0}; std::vector<uint16_t> sym;
    uint16_t fast[1 << FAST];                                // symbol << 4 | length; 0 = code longer than FAST bits (or invalid)
    bool build(const uint8_t* len, int n) {
        for (int i = 0; i < 16; i++) count[i] = 0;
        for (int i = 0; i < n; i++) count[len[i]]++;
        count[0] = 0;
        int left = 1;
        for (int l = 1; l < 16; l++) { left = (left << 1) - count[l]; if (left < 0) return false; }
        uint16_t offs[16]; offs[1] = 0;
        for (int l = 1; l < 15; l++) offs[l + 1] = offs[l] + count[l];
        sym.assign(n, 0);
        for (int i = 0; i < n; i++) if (len[i]) sym[offs[len[i]]++] = (uint16_t)i;
        for (auto& f : fast) f = 0;
        int code = 0, index = 0;
        for (int l = 1; l <= FAST; l++) {                    // canonical codes are MSB first, the stream delivers them LSB first: reverse
            for (int k = 0; k < count[l]; k++, code++, index++) {
                int rev = 0;
                for (int b = 0; b < l; b++) rev |= ((code >> b) & 1) << (l - 1 - b);
                for (int fill = rev; fill < (1 << FAST); fill += 1 << l) fast[fill] = (uint16_t)((sym[index] << 4) | l);
            }
            code <<= 1;
        }
        return true;
    }
    int decode(BitReader& br) const {
        if (br.cnt < 16) br.refill();
        const uint16_t f = fast[br.buf & ((1u << FAST) - 1)];
        if (f) {
            const int l = f & 15;
            if (l > br.cnt) { br.bad = true; return -1; }
            br.buf >>= l; br.cnt -= l;
            return f >> 4;
        }
        int code = 0, first = 0, index = 0;
        for (int l = 1; l < 16; l++) {
            code |= (int)br.bits(1);
            const int c = count[l];
            if (code - c < first) return sym[index + (code - first)];
            index += c; first += c; first <<= 1; code <<= 1;
            if (br.bad) return -1;
        }
        return -1;
    }
};

inline bool inflate(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t expect) {
    if (n < 6 || (src[0] & 0x0f) != 8 || ((src[0] << 8) | src[1]) % 31 != 0 || (src[1] & 0x20)) return false;
    BitReader br(src + 2, n - 2);
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    // the caller knows the size of the unfiltered image: the output is one pre-sized block, anything longer is an error
    out.assign(expect, 0);
    uint8_t* const o = out.data();
    size_t op = 0;
    for (bool last = false; !last;) {
        last = br.bits(1) != 0;
        const uint32_t type = br.bits(2);
        if (br.bad) return false;
        if (type == 0) {
            br.align();
            if (br.pos + 4 > br.n) return false;
            const uint32_t len = br.p[br.pos] | (br.p[br.pos + 1] << 8), nlen = br.p[br.pos + 2] | (br.p[br.pos + 3] << 8);
            br.pos += 4;
            if ((len ^ 0xffffu) != nlen || br.pos + len > br.n || op + len > expect) return false;
            std::memcpy(o + op, br.p + br.pos, len);
            op += len; br.pos += len;
            continue;
        }
        if (type == 3) return false;
        Huffman lit, dist;
        uint8_t lens[320];
        if (type == 1) {
            for (int i = 0; i < 144; i++) lens[i] = 8;
            for (int i = 144; i < 256; i++) lens[i] = 9;
            for (int i = 256; i < 280; i++) lens[i] = 7;
            for (int i = 280; i < 288; i++) lens[i] = 8;
            lit.build(lens, 288);
            for (int i = 0; i < 30; i++) lens[i] = 5;
            dist.build(lens, 30);
        } else {
            const int nlen = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncode = (int)br.bits(4) + 4;
            if (nlen > 286 || ndist > 30) return false;
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t cl[19] = {0};
            for (int i = 0; i < ncode; i++) cl[order[i]] = (uint8_t)br.bits(3);
            Huffman clen;
            if (!clen.build(cl, 19)) return false;
            for (int i = 0; i < nlen + ndist;) {
                const int s = clen.decode(br);
                if (s < 0) return false;
                if (s < 16) { lens[i++] = (uint8_t)s; continue; }
                int rep, val = 0;
                if (s == 16) { if (i == 0) return false; val = lens[i - 1]; rep = 3 + (int)br.bits(2); }
                else if (s == 17) rep = 3 + (int)br.bits(3);
                else rep = 11 + (int)br.bits(7);
                if (i + rep > nlen + ndist) return false;
                while (rep--) lens[i++] = (uint8_t)val;
            }
            if (lens[256] == 0 || !lit.build(lens, nlen) || !dist.build(lens + nlen, ndist)) return false;
        }
        // the symbol loop works on a LOCAL copy of the bit window: the byte stores into `o` may alias the reader's fields, which would
        // otherwise be re-loaded after every literal
        BitReader lr = br;
        const uint16_t* const lfast = lit.fast;
        for (;;) {
            if (lr.cnt < 32) lr.refill();
            int s;
            const uint16_t f = lfast[lr.buf & ((1u << Huffman::FAST) - 1)];
            if (f && (f & 15) <= lr.cnt) { lr.buf >>= (f & 15); lr.cnt -= (f & 15); s = f >> 4; }
            else s = lit.decode(lr);
            if (s < 256) { if (s < 0 || op >= expect) return false; o[op++] = (uint8_t)s; continue; }
            if (s == 256) break;
            if (s > 285) return false;
            const size_t len = (size_t)lbase[s - 257] + lr.bits(lext[s - 257]);
            const int ds = dist.decode(lr);
            if (ds < 0 || ds > 29) return false;
            const size_t d = dbase[ds] + lr.bits(dext[ds]);
            if (lr.bad || d > op || op + len > expect) return false;
            const uint8_t* from = o + op - d;
            if (d >= len) std::memcpy(o + op, from, len);
            else for (size_t k = 0; k < len; k++) o[op + k] = from[k];           // overlapping run: byte by byte, as the format defines it
            op += len;
        }
        if (lr.bad) return false;
        br = lr;
    }
    out.resize(op);
    br.align();
    if (br.pos + 4 > br.n) return false;
    uint32_t a = 1, b = 0;                                   // Adler-32, the modulo taken once per 5552 bytes (the largest run that cannot overflow)
    for (size_t i = 0; i < op;) {
        const size_t e = std::min(op, i + 5552);
        for (; i < e; i++) { a += o[i]; b += a; }
        a %= 65521u; b %= 65521u;
    }
    const uint32_t want = ((uint32_t)br.p[br.pos] << 24) | (br.p[br.pos + 1] << 16) | (br.p[br.pos + 2] << 8) | br.p[br.pos + 3];
    return ((b << 16) | a) == want;
}

struct CrcTable {
    uint32_t t[4][256];
    CrcTable() {
        for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1; t[0][i] = c; }
        for (uint32_t i = 0; i < 256; i++) for (int k = 1; k < 4; k++) t[k][i] = (t[k - 1][i] >> 8) ^ t[0][t[k - 1][i] & 0xff];
    }
};

inline uint32_t crc32(const uint8_t* p, size_t n) {             // slicing-by-4
    static const CrcTable T;                                    // function-local static: initialised once, thread-safe (images are read on several threads)
    const uint32_t (*table)[256] = T.t;
    uint32_t c = 0xffffffffu;
    size_t i = 0;
    for (; i + 4 <= n; i += 4) {
        c ^= (uint32_t)p[i] | ((uint32_t)p[i + 1] << 8) | ((uint32_t)p[i + 2] << 16) | ((uint32_t)p[i + 3] << 24);
        c = table[3][c & 0xff] ^ table[2][(c >> 8) & 0xff] ^ table[1][(c >> 16) & 0xff] ^ table[0][c >> 24];
    }
    for (; i < n; i++) c = table[0][(c ^ p[i]) & 0xff] ^ (c >> 8);
    return c ^ 0xffffffffu;
}

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }

}  // namespace png_detail

inline bool DecodePngGrayUnguarded(const uint8_t* data, size_t size, std::vector<uint8_t>& pixels, int& rows, int& cols) {
    using namespace png_detail;
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (size < 8 + 25 || std::char_traits<char>::compare((const char*)data, (const char*)sig, 8) != 0) return false;
    size_t pos = 8;
    uint32_t w = 0, h = 0; int depth = 0, ctype = -1;
    std::vector<uint8_t> z, plte;
    bool end = false;
    while (!end && pos + 12 <= size) {
        const uint32_t len = be32(data + pos);
        if (pos + 12 + (size_t)len > size) return false;
        const uint8_t* type = data + pos + 4;
        const uint8_t* body = data + pos + 8;
        if (crc32(type, 4 + (size_t)len) != be32(body + len)) return false;
        const std::string t((const char*)type, 4);
        if (t == "IHDR") {
            if (len != 13) return false;
            w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9];
            if (body[10] != 0 || body[11] != 0 || body[12] != 0) return false;          // compression, filter, no interlace
        } else if (t == "PLTE") { if (len % 3 != 0 || len > 768) return false; plte.assign(body, body + len); }
        else if (t == "IDAT") z.insert(z.end(), body, body + len);
        else if (t == "IEND") end = true;
        pos += 12 + (size_t)len;
    }
    if (!end || w == 0 || h == 0 || w > 65535 || h > 65535) return false;
    int ch;
    const bool d816 = depth == 8 || depth == 16, dlow = depth == 1 || depth == 2 || depth == 4;
    if (ctype == 0 && (d816 || dlow)) ch = 1;
    else if (ctype == 2 && d816) ch = 3;
    else if (ctype == 3 && (dlow || depth == 8) && !plte.empty()) ch = 1;
    else if (ctype == 4 && d816) ch = 2;
    else if (ctype == 6 && d816) ch = 4;
    else return false;
    const size_t bits = (size_t)ch * depth, bpp = bits >= 8 ? bits / 8 : 1, stride = (bits * w + 7) / 8;
    std::vector<uint8_t> raw;
    // deflate expands by at most 1032 : 1 (a 258-byte match for 2 bits): an IHDR that promises more than the IDAT bytes could ever hold is
    // refused BEFORE anything of its size is allocated (65535 x 65535 RGBA16 would be 34 GB from a 60-byte file)
    const size_t expect = (stride + 1) * (size_t)h;
    if (expect > 1032 * z.size() + 64) return false;
    if (!inflate(z.data(), z.size(), raw, expect) || raw.size() != expect) return false;
    auto grey8 = [](uint32_t r, uint32_t g, uint32_t b) -> uint8_t { return (r == g && r == b) ? (uint8_t)r : (uint8_t)((9797u * r + 19234u * g + 3737u * b) >> 15); };
    // undo the per-row filters in place (row r occupies raw[r*(stride+1)+1 ...])
    const std::vector<uint8_t> zero(stride, 0);
    pixels.assign((size_t)w * h, 0);
    for (uint32_t r = 0; r < h; r++) {
        uint8_t* cur = &raw[(size_t)r * (stride + 1) + 1];
        const int f = cur[-1];
        if (f > 4) return false;
        const uint8_t* up = r ? cur - (stride + 1) : zero.data();              // the row above, already unfiltered in place
        // the left neighbour is carried in a register (a store-to-load round trip per byte otherwise); one loop per byte distance for the
        // grey formats KITTI uses (bpp 1), the general loop for the rest
        if (f == 1) {
            if (bpp == 1) { unsigned a = 0; for (size_t i = 0; i < stride; i++) { a = (cur[i] + a) & 255u; cur[i] = (uint8_t)a; } }
            else for (size_t i = bpp; i < stride; i++) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]);
        } else if (f == 2) { for (size_t i = 0; i < stride; i++) cur[i] = (uint8_t)(cur[i] + up[i]); }
        else if (f == 3) {
            if (bpp == 1) { unsigned a = 0; for (size_t i = 0; i < stride; i++) { a = (cur[i] + ((a + up[i]) >> 1)) & 255u; cur[i] = (uint8_t)a; } }
            else {
                for (size_t i = 0; i < bpp && i < stride; i++) cur[i] = (uint8_t)(cur[i] + (up[i] >> 1));
                for (size_t i = bpp; i < stride; i++) cur[i] = (uint8_t)(cur[i] + ((cur[i - bpp] + up[i]) >> 1));
            }
        } else if (f == 4) {                                                                     // |p - a| = |b - c|, |p - b| = |a - c|, |p - c| = |a + b - 2c|
            if (bpp == 1) {
                int a = 0, c = 0;
                for (size_t i = 0; i < stride; i++) {
                    const int b = up[i];
                    const int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - 2 * c);
                    const int bc = pb <= pc ? b : c;                                             // selects, no branches: noise-like rows mispredict every other byte
                    a = (cur[i] + (((pa <= pb) & (pa <= pc)) ? a : bc)) & 255;
                    cur[i] = (uint8_t)a; c = b;
                }
            } else {
                for (size_t i = 0; i < bpp && i < stride; i++) cur[i] = (uint8_t)(cur[i] + up[i]);      // a = c = 0: the predictor is b
                for (size_t i = bpp; i < stride; i++) {
                    const int a = cur[i - bpp], b = up[i], c = up[i - bpp];
                    const int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - 2 * c);
                    const int bc = pb <= pc ? b : c;
                    cur[i] = (uint8_t)(cur[i] + (((pa <= pb) & (pa <= pc)) ? a : bc));
                }
            }
        }
        uint8_t* dst = &pixels[(size_t)r * w];
        if (depth < 8) {                                                     // packed samples, most significant bits first
            const int mask = (1 << depth) - 1, scale = 255 / mask;
            for (uint32_t x = 0; x < w; x++) {
                const size_t bit = (size_t)x * depth;
                const int v = (cur[bit >> 3] >> (8 - depth - (int)(bit & 7))) & mask;
                if (ctype == 3) { if ((size_t)v * 3 + 2 >= plte.size()) return false; dst[x] = grey8(plte[v * 3], plte[v * 3 + 1], plte[v * 3 + 2]); }
                else dst[x] = (uint8_t)(v * scale);
            }
        } else if (ctype == 3) {
            for (uint32_t x = 0; x < w; x++) { const size_t v = cur[x]; if (v * 3 + 2 >= plte.size()) return false; dst[x] = grey8(plte[v * 3], plte[v * 3 + 1], plte[v * 3 + 2]); }
        } else if (ch <= 2) {
            for (uint32_t x = 0; x < w; x++) dst[x] = cur[x * bpp];                                        // grey (+ alpha): 16-bit keeps the high byte
        } else if (depth == 8) {
            for (uint32_t x = 0; x < w; x++) { const uint8_t* q = cur + x * bpp; dst[x] = grey8(q[0], q[1], q[2]); }
        } else {                                                             // 16-bit colour: converted at 16 bits (with rounding), then the high byte
            for (uint32_t x = 0; x < w; x++) {
                const uint8_t* q = cur + x * bpp;
                const uint32_t R = (q[0] << 8) | q[1], G = (q[2] << 8) | q[3], B = (q[4] << 8) | q[5];
                const uint32_t g16 = (R == G && R == B) ? R : ((9797u * R + 19234u * G + 3737u * B + 16384u) >> 15);
                dst[x] = (uint8_t)(g16 >> 8);
            }
        }
    }
    rows = (int)h; cols = (int)w;
    return true;
}

// Decodes PNG bytes into a tight rows x cols u8 grey plane.  false: not a PNG this reader supports, a corrupt one, or one the process
// has no memory for (the documented contract is "returns false": no exception leaves the reader).
inline bool DecodePngGray(const uint8_t* data, size_t size, std::vector<uint8_t>& pixels, int& rows, int& cols) {
    try { return DecodePngGrayUnguarded(data, size, pixels, rows, cols); }
    catch (const std::bad_alloc&) { return false; }
    catch (const std::length_error&) { return false; }
}

inline bool ReadPngGray(const std::string& path, std::vector<uint8_t>& pixels, int& rows, int& cols) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f.is_open()) return false;
    const std::streamoff size = f.tellg();
    if (size <= 0) return false;
    std::vector<uint8_t> buf((size_t)size);
    f.seekg(0);
    if (!f.read((char*)buf.data(), size)) return false;
    return DecodePngGray(buf.data(), buf.size(), pixels, rows, cols);
}

}  // namespace io
}  // namespace myslam
