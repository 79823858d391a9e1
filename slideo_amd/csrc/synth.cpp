/*
 * synth.cpp — seeded synthetic slide pages and video frames with ground truth
 * (SURVEY.md §8d "Synthetic inputs"; seeds: pages 0x511DE0, frames 0xF4A3E5).
 *
 * Host-only utility (g++), used by tests and bench.py to make inputs; it is not
 * on the product path and contains no matching logic.  Integer/IEEE-only
 * arithmetic (no libm transcendental in the pixel path) so that the same seed
 * gives the same bytes on every machine.
 *
 * Pages: white 16:9 canvas, coloured header bar, 3-8 text-like rows of random
 * glyphs (random 3x5 cell patterns, dark intensities), 0-3 outlined rectangles,
 * 0-1 random-noise "photo" block, page-number glyphs bottom-right; a page
 * shares its predecessor's layout with probability 0.3 (incremental builds).
 * Frames: a page (or, p = 0.1, no slide) under a similarity transform (scale
 * U[0.85,1], rotation U[-1,1] deg, slide kept >= 90 % visible), bilinear
 * sampled, optional occluder (<= 10 % area), additive noise sigma ~ 2 and a
 * per-8x8-block offset that mimics codec quantisation.
 * Perspective frames (slideo_synth_frames_persp, persp > 0; BASELINE configs[4] "RANSAC homography
 * verify"): the slide is first seen through a projective map w = 1 + g u / pw + h v / ph in
 * centred slide coordinates, g, h ~ U[-persp, persp] (a keystone of about persp / 2 across the
 * slide), then the similarity above; the ground truth is the 3x3 slide -> frame homography.
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace {

struct Rng {  // splitmix64
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    }
    int range(int lo, int hi) { return lo + (int)(next() % (uint64_t)(hi - lo + 1)); }  // inclusive
    double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    // approx N(0,1): Irwin-Hall sum of 4 uniforms, IEEE-only
    float gauss() {
        uint64_t r = next();
        int a = (int)(r & 0xffff) + (int)((r >> 16) & 0xffff) + (int)((r >> 32) & 0xffff) + (int)(r >> 48);
        return (float)(a - 131070) * (1.0f / 37837.0f);  // var of sum = 4*65536^2/12 -> sd 37837
    }
};

struct Rect { int x, y, w, h; uint8_t b, g, r; };
struct Glyph { int x, y, cw, ch; uint16_t bits; uint8_t v; };
struct Layout {
    Rect header;
    std::vector<Glyph> glyphs;
    std::vector<Rect> outlines;
    bool photo = false; Rect photo_rect{}; uint64_t photo_seed = 0;
    int rows_used_y = 0;
};

static void fill(uint8_t* img, int W, int H, int x, int y, int w, int h, uint8_t b, uint8_t g, uint8_t r) {
    int x0 = std::max(x, 0), y0 = std::max(y, 0), x1 = std::min(x + w, W), y1 = std::min(y + h, H);
    for (int yy = y0; yy < y1; ++yy) {
        uint8_t* p = img + ((size_t)yy * W + x0) * 3;
        for (int xx = x0; xx < x1; ++xx) { p[0] = b; p[1] = g; p[2] = r; p += 3; }
    }
}

static void add_text_row(Layout& L, Rng& rng, int W, int y, double sc) {
    int gh = (int)(rng.range(14, 28) * sc);          // glyph height
    int x = (int)(rng.range(60, 160) * sc);
    int xend = W - (int)(rng.range(60, 500) * sc);
    uint8_t v = (uint8_t)rng.range(0, 80);
    while (x < xend) {
        int gw = (int)(rng.range(6, 24) * sc);
        if (rng.range(0, 6) == 0) { x += gw; continue; }   // word gap
        Glyph g;
        g.x = x; g.y = y; g.cw = std::max(gw / 3, 1); g.ch = std::max(gh / 5, 1);
        g.bits = (uint16_t)(rng.next() & 0x7fff);
        if (!g.bits) g.bits = 0x7fff;
        g.v = v;
        L.glyphs.push_back(g);
        x += g.cw * 3 + (int)(rng.range(3, 8) * sc);
    }
    L.rows_used_y = y + gh + (int)(rng.range(18, 40) * sc);
}

static void make_layout(Layout& L, Rng& rng, int W, int H, const Layout* prev) {
    double sc = (double)W / 2001.0;
    if (prev && rng.unit() < 0.3 && prev->rows_used_y < H - (int)(160 * sc)) {
        L = *prev;                                   // incremental build: same template + more rows
        int extra = rng.range(1, 2);
        for (int i = 0; i < extra && L.rows_used_y < H - (int)(160 * sc); ++i)
            add_text_row(L, rng, W, L.rows_used_y, sc);
        return;
    }
    L = Layout();
    L.header = {0, 0, W, (int)(rng.range(70, 120) * sc), (uint8_t)rng.range(40, 220),
                (uint8_t)rng.range(40, 220), (uint8_t)rng.range(40, 220)};
    int y = L.header.h + (int)(rng.range(30, 70) * sc);
    L.rows_used_y = y;
    int rows = rng.range(3, 8);
    for (int i = 0; i < rows && L.rows_used_y < H - (int)(160 * sc); ++i) add_text_row(L, rng, W, L.rows_used_y, sc);
    int nrect = rng.range(0, 3);
    for (int i = 0; i < nrect; ++i) {
        Rect r;
        r.w = (int)(rng.range(120, 600) * sc); r.h = (int)(rng.range(80, 300) * sc);
        r.x = rng.range(20, std::max(21, W - r.w - 20)); r.y = rng.range(L.header.h + 10, std::max(L.header.h + 11, H - r.h - 20));
        r.b = (uint8_t)rng.range(0, 120); r.g = (uint8_t)rng.range(0, 120); r.r = (uint8_t)rng.range(0, 120);
        L.outlines.push_back(r);
    }
    if (rng.range(0, 1)) {
        L.photo = true;
        Rect r;
        r.w = (int)(rng.range(200, 900) * sc); r.h = (int)(rng.range(150, 620) * sc);   // <= 30 % area
        r.x = rng.range(20, std::max(21, W - r.w - 20)); r.y = rng.range(L.header.h + 10, std::max(L.header.h + 11, H - r.h - 20));
        r.b = r.g = r.r = 0;
        L.photo_rect = r; L.photo_seed = rng.next();
    }
}

static void render_page(const Layout& L, int page_nr, int W, int H, uint8_t* img) {
    double sc = (double)W / 2001.0;
    std::memset(img, 255, (size_t)W * H * 3);
    fill(img, W, H, L.header.x, L.header.y, L.header.w, L.header.h, L.header.b, L.header.g, L.header.r);
    if (L.photo) {   // smooth-ish random texture: 8x8 px cells of random colour
        Rng pr(L.photo_seed);
        int cell = std::max((int)(8 * sc), 2);
        for (int y = 0; y < L.photo_rect.h; y += cell)
            for (int x = 0; x < L.photo_rect.w; x += cell) {
                uint64_t r = pr.next();
                fill(img, W, H, L.photo_rect.x + x, L.photo_rect.y + y, std::min(cell, L.photo_rect.w - x),
                     std::min(cell, L.photo_rect.h - y), (uint8_t)r, (uint8_t)(r >> 8), (uint8_t)(r >> 16));
            }
    }
    int t = std::max((int)(3 * sc), 1);
    for (const Rect& r : L.outlines) {
        fill(img, W, H, r.x, r.y, r.w, t, r.b, r.g, r.r);
        fill(img, W, H, r.x, r.y + r.h - t, r.w, t, r.b, r.g, r.r);
        fill(img, W, H, r.x, r.y, t, r.h, r.b, r.g, r.r);
        fill(img, W, H, r.x + r.w - t, r.y, t, r.h, r.b, r.g, r.r);
    }
    for (const Glyph& g : L.glyphs)
        for (int cy = 0; cy < 5; ++cy)
            for (int cx = 0; cx < 3; ++cx)
                if (g.bits >> (cy * 3 + cx) & 1)
                    fill(img, W, H, g.x + cx * g.cw, g.y + cy * g.ch, g.cw, g.ch, g.v, g.v, g.v);
    // page number, bottom-right: binary digits of page_nr as glyph cells
    int cw = std::max((int)(8 * sc), 2), ch = std::max((int)(20 * sc), 3);
    int x = W - (int)(60 * sc) - 12 * (cw + 2), y = H - (int)(50 * sc);
    for (int b = 0; b < 12; ++b)
        if ((page_nr + 1) >> b & 1) fill(img, W, H, x + (11 - b) * (cw + 2), y, cw, ch, 30, 30, 30);
}

struct FrameTruth { int32_t page; double M[9]; };  // M: slide -> frame (3x3; rows 0-1 = the 2x3 of a similarity frame)

static void render_frame(uint64_t seed, int64_t frame_idx, const uint8_t* pages, int n_pages, int pw, int ph,
                         int fw, int fh, uint8_t* out, FrameTruth* truth, double persp = 0.0) {
    Rng rng(seed ^ (0x9E3779B97F4A7C15ULL * (uint64_t)(frame_idx + 1)));
    bool none = n_pages == 0 || rng.unit() < 0.1;
    int page = none ? -1 : rng.range(0, n_pages - 1);
    truth->page = page;
    std::fill(truth->M, truth->M + 9, 0.0);
    if (none) {   // dark noisy scene (speaker shot): smooth blobs + noise
        int cell = 64;
        int gw = fw / cell + 2, gh = fh / cell + 2;
        std::vector<uint8_t> grid((size_t)gw * gh * 3);
        for (auto& v : grid) v = (uint8_t)rng.range(10, 90);
        for (int y = 0; y < fh; ++y)
            for (int x = 0; x < fw; ++x) {
                int gx = x / cell, gy = y / cell;
                int fx = x % cell, fy = y % cell;
                for (int c = 0; c < 3; ++c) {
                    int a = grid[((size_t)gy * gw + gx) * 3 + c], b = grid[((size_t)gy * gw + gx + 1) * 3 + c];
                    int d = grid[((size_t)(gy + 1) * gw + gx) * 3 + c], e = grid[((size_t)(gy + 1) * gw + gx + 1) * 3 + c];
                    int top = a * (cell - fx) + b * fx, bot = d * (cell - fx) + e * fx;
                    out[((size_t)y * fw + x) * 3 + c] = (uint8_t)((top * (cell - fy) + bot * fy) / (cell * cell));
                }
            }
    } else {
        const uint8_t* pg = pages + (size_t)page * pw * ph * 3;
        double s = (0.85 + 0.15 * rng.unit()) * (double)fw / (double)pw;
        double ang = (rng.unit() * 2.0 - 1.0) * (3.14159265358979323846 / 180.0);
        double ca = std::cos(ang) * s, sa = std::sin(ang) * s;
        // slide centre lands near the frame centre; keep >= 90 % visible: shift <= 5 % of frame
        double tx0 = fw * 0.5 + (rng.unit() * 2 - 1) * 0.05 * fw, ty0 = fh * 0.5 + (rng.unit() * 2 - 1) * 0.05 * fh;
        double cxp = pw * 0.5, cyp = ph * 0.5;
        double M0 = ca, M1 = -sa, M2 = tx0 - (ca * cxp - sa * cyp);
        double M3 = sa, M4 = ca, M5 = ty0 - (sa * cxp + ca * cyp);
        // (persp == 0 keeps the similarity frames of earlier rounds bit for bit: no extra draw)
        const double g = persp > 0 ? (rng.unit() * 2 - 1) * persp / pw : 0.0, hh = persp > 0 ? (rng.unit() * 2 - 1) * persp / ph : 0.0;
        // H = [M0 M1 M2; M3 M4 M5; 0 0 1] * P^-1-free form: centred projective P = [1 0 -cx; 0 1 -cy; g h 1 - g cx - h cy],
        // S = [ca -sa tx0; sa ca ty0; 0 0 1] on centred coordinates
        double H[9] = {ca + tx0 * g, -sa + tx0 * hh, -ca * cxp + sa * cyp + tx0 * (1 - g * cxp - hh * cyp),
                       sa + ty0 * g, ca + ty0 * hh, -sa * cxp - ca * cyp + ty0 * (1 - g * cxp - hh * cyp),
                       g, hh, 1 - g * cxp - hh * cyp};
        if (persp > 0) { for (int i = 0; i < 9; ++i) H[i] /= H[8]; }
        else { H[0] = M0; H[1] = M1; H[2] = M2; H[3] = M3; H[4] = M4; H[5] = M5; H[6] = 0; H[7] = 0; H[8] = 1; }
        std::memcpy(truth->M, H, sizeof(H));
        // inverse: frame -> slide (adjugate; for a similarity frame the projective row is 0 0 1 and W == 1)
        double I0, I1, I2, I3, I4, I5, I6 = 0, I7 = 0, I8 = 1;
        if (persp > 0) {
            I0 = H[4] * H[8] - H[5] * H[7]; I1 = H[2] * H[7] - H[1] * H[8]; I2 = H[1] * H[5] - H[2] * H[4];
            I3 = H[5] * H[6] - H[3] * H[8]; I4 = H[0] * H[8] - H[2] * H[6]; I5 = H[2] * H[3] - H[0] * H[5];
            I6 = H[3] * H[7] - H[4] * H[6]; I7 = H[1] * H[6] - H[0] * H[7]; I8 = H[0] * H[4] - H[1] * H[3];
        } else {
            double det = M0 * M4 - M1 * M3;
            I0 = M4 / det; I1 = -M1 / det; I3 = -M3 / det; I4 = M0 / det;
            I2 = -(I0 * M2 + I1 * M5); I5 = -(I3 * M2 + I4 * M5);
        }
        uint8_t bg = (uint8_t)rng.range(15, 45);
        for (int y = 0; y < fh; ++y) {
            for (int x = 0; x < fw; ++x) {
                double sx = I0 * x + I1 * y + I2, sy = I3 * x + I4 * y + I5;
                if (persp > 0) { const double w = I6 * x + I7 * y + I8; sx /= w; sy /= w; }
                uint8_t* o = out + ((size_t)y * fw + x) * 3;
                int ix = (int)std::floor(sx), iy = (int)std::floor(sy);
                if (ix < 0 || iy < 0 || ix >= pw - 1 || iy >= ph - 1) { o[0] = o[1] = o[2] = bg; continue; }
                int fx = (int)((sx - ix) * 256.0), fy = (int)((sy - iy) * 256.0);
                const uint8_t* p00 = pg + ((size_t)iy * pw + ix) * 3;
                const uint8_t* p10 = p00 + (size_t)pw * 3;
                for (int c = 0; c < 3; ++c) {
                    int top = p00[c] * (256 - fx) + p00[3 + c] * fx;
                    int bot = p10[c] * (256 - fx) + p10[3 + c] * fx;
                    o[c] = (uint8_t)((top * (256 - fy) + bot * fy + 32768) >> 16);
                }
            }
        }
        if (rng.range(0, 1)) {   // occluder (speaker inset), <= 10 % area
            int ow = rng.range(fw / 10, fw / 4), oh = rng.range(fh / 8, (int)(fh * 0.4));
            int ox = rng.range(0, fw - ow), oy = fh - oh;
            uint8_t v = (uint8_t)rng.range(40, 110);
            fill(out, fw, fh, ox, oy, ow, oh, v, (uint8_t)(v + 10), (uint8_t)(v + 25));
        }
    }
    // sensor noise + block offsets
    int bw = (fw + 7) / 8;
    std::vector<int8_t> blk((size_t)bw * 3);
    for (int y = 0; y < fh; ++y) {
        if ((y & 7) == 0) for (auto& v : blk) v = (int8_t)rng.range(-1, 1);
        uint8_t* o = out + (size_t)y * fw * 3;
        for (int x = 0; x < fw; ++x)
            for (int c = 0; c < 3; ++c) {
                float n = rng.gauss() * 2.0f;
                int v = (int)o[3 * x + c] + (int)std::lrintf(n) + blk[(size_t)(x >> 3) * 3 + c];
                o[3 * x + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
    }
}

}  // namespace

extern "C" {

// n pages of w x h (BGR, packed) into out[n][h][w][3]; sequential (template sharing).
void slideo_synth_pages(uint64_t seed, int n, int w, int h, uint8_t* out, int threads) {
    Rng rng(seed);
    std::vector<Layout> layouts(n);
    for (int i = 0; i < n; ++i) make_layout(layouts[i], rng, w, h, i ? &layouts[i - 1] : nullptr);   // sequential: template sharing
    threads = std::max(1, std::min(threads, n));
    auto work = [&](int t) { for (int i = t; i < n; i += threads) render_page(layouts[i], i, w, h, out + (size_t)i * w * h * 3); };
    if (threads == 1) { work(0); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t) th.emplace_back(work, t);
    for (auto& x : th) x.join();
}

// frames [first, first+n) of fw x fh; truth_page[n], truth_M[n][6] (slide -> frame).
void slideo_synth_frames(uint64_t seed, int64_t first, int n, const uint8_t* pages, int n_pages, int pw, int ph,
                         int fw, int fh, uint8_t* out, int32_t* truth_page, double* truth_M, int threads) {
    threads = std::max(1, std::min(threads, n));
    auto work = [=](int t) {
        for (int i = t; i < n; i += threads) {
            FrameTruth tr;
            render_frame(seed, first + i, pages, n_pages, pw, ph, fw, fh, out + (size_t)i * fw * fh * 3, &tr);
            if (truth_page) truth_page[i] = tr.page;
            if (truth_M) std::memcpy(truth_M + (size_t)i * 6, tr.M, 6 * sizeof(double));
        }
    };
    if (threads == 1) { work(0); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t) th.emplace_back(work, t);
    for (auto& x : th) x.join();
}

// The same with a projective component (persp > 0, see the header comment); truth_H[n][9] (slide -> frame, 3x3).
void slideo_synth_frames_persp(uint64_t seed, int64_t first, int n, const uint8_t* pages, int n_pages, int pw, int ph,
                               int fw, int fh, double persp, uint8_t* out, int32_t* truth_page, double* truth_H, int threads) {
    threads = std::max(1, std::min(threads, n));
    auto work = [=](int t) {
        for (int i = t; i < n; i += threads) {
            FrameTruth tr;
            render_frame(seed, first + i, pages, n_pages, pw, ph, fw, fh, out + (size_t)i * fw * fh * 3, &tr, persp);
            if (truth_page) truth_page[i] = tr.page;
            if (truth_H) std::memcpy(truth_H + (size_t)i * 9, tr.M, sizeof(tr.M));
        }
    };
    if (threads == 1) { work(0); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t) th.emplace_back(work, t);
    for (auto& x : th) x.join();
}

}  // extern "C"
