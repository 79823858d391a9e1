// Minimal PNG -> 8UC3 BGR decoder for the C++ host mirror (page ingest).
//
// The reference reads every page image with OpenCV's imread and (intends to) hand ORB an 8UC3 BGR matrix
// (crates/matching-opencv/src/lib.rs:98-104).  poppler's `pdftocairo -png` writes 8-bit
// non-interlaced RGB; for completeness grey, grey+alpha, palette and RGBA are accepted too (alpha is dropped,
// as imread's default IMREAD_COLOR does).  Anything else (16-bit, Adam7) fails loudly.  Host side only; the
// decoded page goes to slideo_matcher_add_pages_bgr8.  Needs zlib (-lz) for inflate.
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace slideo_host {

struct PngImage { int w = 0, h = 0; std::vector<uint8_t> bgr; };

namespace png_detail {
inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
}  // namespace png_detail

inline PngImage decode_png_bgr(const std::string& path) {
    using namespace png_detail;
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("File '" + path + "' must exist");                    // lib.rs:95-97 (panic)
    std::vector<uint8_t> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (d.size() < 8 || std::memcmp(d.data(), sig, 8) != 0) throw std::runtime_error(path + ": not a PNG file");
    int w = 0, h = 0, depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte;
    for (size_t pos = 8; pos + 12 <= d.size();) {
        const uint32_t len = be32(&d[pos]);
        const char* type = reinterpret_cast<const char*>(&d[pos + 4]);
        if (pos + 12 + (size_t)len > d.size()) throw std::runtime_error(path + ": truncated PNG chunk");
        const uint8_t* body = &d[pos + 8];
        if (!std::memcmp(type, "IHDR", 4) && len >= 13) {
            w = (int)be32(body); h = (int)be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
        } else if (!std::memcmp(type, "PLTE", 4)) plte.assign(body, body + len);
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!std::memcmp(type, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    if (w <= 0 || h <= 0) throw std::runtime_error(path + ": PNG without IHDR");
    if (depth != 8 || interlace != 0) throw std::runtime_error(path + ": only 8-bit non-interlaced PNGs are supported");
    int ch;
    switch (ctype) { case 0: ch = 1; break; case 2: ch = 3; break; case 3: ch = 1; break; case 4: ch = 2; break; case 6: ch = 4; break;
                     default: throw std::runtime_error(path + ": unknown PNG colour type"); }
    if (ctype == 3 && plte.empty()) throw std::runtime_error(path + ": palette PNG without PLTE");
    const size_t rowb = (size_t)w * ch;
    std::vector<uint8_t> raw((rowb + 1) * (size_t)h);
    uLongf outlen = (uLongf)raw.size();
    const int zr = uncompress(raw.data(), &outlen, idat.data(), (uLong)idat.size());
    if (zr != Z_OK || outlen != raw.size()) throw std::runtime_error(path + ": PNG inflate failed");
    // undo the per-row filters in place (prev = previous reconstructed row, zeros above the first)
    std::vector<uint8_t> zero(rowb, 0);
    for (int y = 0; y < h; ++y) {
        uint8_t* cur = &raw[(rowb + 1) * (size_t)y + 1];
        const uint8_t* prev = y ? &raw[(rowb + 1) * (size_t)(y - 1) + 1] : zero.data();
        const int ft = cur[-1];
        for (size_t i = 0; i < rowb; ++i) {
            const int a = i >= (size_t)ch ? cur[i - ch] : 0, b = prev[i], c = i >= (size_t)ch ? prev[i - ch] : 0;
            int add;
            switch (ft) { case 0: add = 0; break; case 1: add = a; break; case 2: add = b; break; case 3: add = (a + b) >> 1; break;
                          case 4: add = paeth(a, b, c); break; default: throw std::runtime_error(path + ": bad PNG filter type"); }
            cur[i] = (uint8_t)(cur[i] + add);
        }
    }
    PngImage img; img.w = w; img.h = h; img.bgr.resize((size_t)w * h * 3);
    for (int y = 0; y < h; ++y) {
        const uint8_t* s = &raw[(rowb + 1) * (size_t)y + 1];
        uint8_t* o = &img.bgr[(size_t)y * w * 3];
        for (int x = 0; x < w; ++x, s += ch, o += 3) {
            switch (ctype) {
                case 0: case 4: o[0] = o[1] = o[2] = s[0]; break;                                   // grey (+alpha dropped)
                case 2: case 6: o[0] = s[2]; o[1] = s[1]; o[2] = s[0]; break;                       // RGB(A) -> BGR
                default: {                                                                          // palette
                    const size_t k = (size_t)s[0] * 3;
                    if (k + 2 >= plte.size() + 0 && k + 3 > plte.size()) throw std::runtime_error(path + ": palette index out of range");
                    o[0] = plte[k + 2]; o[1] = plte[k + 1]; o[2] = plte[k];
                }
            }
        }
    }
    return img;
}

}  // namespace slideo_host
