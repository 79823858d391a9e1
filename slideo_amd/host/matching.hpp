// matching.hpp — C++ host-side mirror of the reference's `matching` trait surface over the C ABI.
//
// The reference is compiled code (Rust); its toolchain is absent from this image, so the compiled host side
// above include/slideo_amd.h is C++.  Same names, argument meaning and behaviour as
// crates/matching/src/lib.rs:7-40 and progress.rs:3-17, implemented the way
// crates/matching-opencv/src/lib.rs implements them over OpenCV (cited below as mo/lib.rs:<line>).
//
//   HipImageVideoMatcher matcher;                                   // OpenCVImageVideoMatcher::default()   main.rs:69
//   auto vm   = matcher.create_video_matcher(pages, reporter);      // mo/lib.rs:37-64
//   auto task = vm->match_images_with_video(video_path, reporter);  // mo/lib.rs:140-158
//   auto out  = task->process();                                    // mo/lib.rs:168-246
//
// Errors: the reference panics (unwrap()/panic! throughout); here every non-zero status of the C ABI throws
// std::runtime_error.  Video decode and PNG decode are outside the hot path: frames come from a raw
// container (RawVideo), page images through a caller-supplied loader (default: binary PPM "P6").
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <filesystem>
#include <fstream>
#include <functional>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "slideo_amd.h"
#include "png.hpp"

namespace slideo_host {

// matching::ProgressReporter (crates/matching/src/progress.rs:3-17)
class ProgressReporter {
public:
    using Handler = std::function<void(uint64_t, uint64_t, const std::string&)>;
    explicit ProgressReporter(Handler h = nullptr) : handler_(std::move(h)) {}
    void report(uint64_t processed_count, uint64_t total_count, const std::string& message) const {
        if (handler_) handler_(processed_count, total_count, message);
    }
private:
    Handler handler_;
};

// matching::Matching<I> (crates/matching/src/lib.rs:35-40)
// ---- timeline output contract of crates/app -----------------------------------------
// videos_mapping rows (db.rs:162-191) and PdfVideoMatching records (db.rs:194-201, 209-260); host logic only.
struct VideoMappingRow { uint32_t video_ms; bool has_pdf; std::string pdf_hash; uint32_t page; };
struct PdfVideoMatching { uint32_t video_offset_ms; std::string pdf_hash, video_hash; uint32_t page_idx, duration_ms; };

// I must expose .pdf_hash (std::string) and .page_nr (1-based), like PdfPage (pdf_to_images.rs:19-31)
template <class MatchingT>
inline std::vector<VideoMappingRow> videos_mapping_rows(const std::vector<MatchingT>& matchings) {
    std::vector<VideoMappingRow> rows;
    for (const auto& m : matchings) {
        VideoMappingRow r;
        r.video_ms = (uint32_t)(uint64_t)(m.video_time_s * 1000.0);           // as_millis() as u32   (db.rs:176)
        r.has_pdf = (bool)m.image;
        r.pdf_hash = m.image ? m.image->pdf_hash : std::string();             // db.rs:175
        r.page = m.image ? (uint32_t)(m.image->page_nr - 1) : 0u;             // db.rs:177
        rows.push_back(r);
    }
    return rows;
}

inline std::vector<PdfVideoMatching> pdf_video_matchings(std::vector<VideoMappingRow> rows, const std::string& pdf_hash,
                                                        const std::string& video_hash) {
    std::stable_sort(rows.begin(), rows.end(), [](const VideoMappingRow& a, const VideoMappingRow& b) { return a.video_ms < b.video_ms; });
    std::vector<PdfVideoMatching> out;
    for (size_t i = 0; i < rows.size(); ++i) {
        const uint32_t duration = i + 1 < rows.size() ? rows[i + 1].video_ms - rows[i].video_ms : 5000u;   // db.rs:239-245
        if (rows[i].has_pdf && rows[i].pdf_hash == pdf_hash)
            out.push_back({rows[i].video_ms, rows[i].pdf_hash, video_hash, rows[i].page, duration});
    }
    return out;
}

template <class I>
struct Matching {
    double video_time_s;
    size_t video_frame_idx;
    std::optional<I> image;
};

struct Image8 { int w = 0, h = 0; std::vector<uint8_t> bgr; };
using ImageLoader = std::function<Image8(const std::string& path)>;

inline Image8 load_ppm_bgr(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("File '" + path + "' must exist");                       // mo/lib.rs:95-97
    std::string magic; int w, h, maxv;
    f >> magic >> w >> h >> maxv;
    f.get();
    if (magic != "P6" || maxv != 255) throw std::runtime_error("Could not read file '" + path + "'");   // mo/lib.rs:99-101
    Image8 im; im.w = w; im.h = h; im.bgr.resize((size_t)w * h * 3);
    f.read(reinterpret_cast<char*>(im.bgr.data()), (std::streamsize)im.bgr.size());
    for (size_t i = 0; i < im.bgr.size(); i += 3) std::swap(im.bgr[i], im.bgr[i + 2]);        // RGB -> BGR
    return im;
}

// imread stand-in of the page ingest (mo/lib.rs:98-104): PNG (what pdftocairo writes) or binary PPM, by extension
inline Image8 load_image_bgr(const std::string& path) {
    const auto dot = path.rfind('.');
    std::string ext = dot == std::string::npos ? "" : path.substr(dot + 1);
    for (auto& c : ext) c = (char)std::tolower((unsigned char)c);
    if (ext == "png") { PngImage p = decode_png_bgr(path); Image8 im; im.w = p.w; im.h = p.h; im.bgr = std::move(p.bgr); return im; }
    return load_ppm_bgr(path);
}

// Page list of one PDF as the app builds it: every entry of the pdftocairo target directory is
// named "p-<nr>.png" (zero padded to the page count's width), page_nr = the number after "p-", pages sorted by it
// (crates/pdftocairo/src/pdftocairo.rs:216-231); PdfPage{page_nr (1-based), image_path, pdf_path, pdf_hash}
// (crates/app/src/pdf_to_images.rs:19-31,138-146).  A foreign file name is a panic there and an exception here.
struct PdfPage {
    size_t page_nr = 0; std::string image_path, pdf_path, pdf_hash;
    std::string get_path() const { return image_path; }                                     // MatchableImage, pdf_to_images.rs:33-37
    bool operator==(const PdfPage& o) const { return page_nr == o.page_nr && pdf_hash == o.pdf_hash; }
};

inline std::vector<PdfPage> scan_page_dir(const std::string& target_dir, const std::string& pdf_hash, const std::string& pdf_path = "") {
    std::vector<PdfPage> pages;
    for (const auto& item : std::filesystem::directory_iterator(target_dir)) {
        const std::string file_name = item.path().filename().string();                     // e.g. p-01.png
        const std::string stem = file_name.substr(0, file_name.find('.'));
        size_t used = 0; unsigned long nr = 0;
        bool ok = stem.size() > 2;
        if (ok) { try { nr = std::stoul(stem.substr(2), &used); } catch (const std::exception&) { ok = false; } }
        if (!ok || used != stem.size() - 2) throw std::runtime_error("unexpected file '" + file_name + "' in page directory '" + target_dir + "'");
        pages.push_back({(size_t)nr, item.path().string(), pdf_path, pdf_hash});
    }
    std::stable_sort(pages.begin(), pages.end(), [](const PdfPage& a, const PdfPage& b) { return a.page_nr < b.page_nr; });
    return pages;
}

// Raw BGR frame container standing in for VideoCapture (mo/video_capture.rs:16-40):
// "SLVF" u32 w u32 h f64 fps u64 n, then n * h*w*3 bytes (same layout as slideo_amd/matching.py RawVideo).
class RawVideo {
public:
    explicit RawVideo(const std::string& path) : path_(path), f_(path, std::ios::binary) {
        char magic[4];
        if (!f_ || !f_.read(magic, 4) || std::string(magic, 4) != "SLVF") throw std::runtime_error("not a raw frame container: " + path);
        uint32_t w, h; double fps; uint64_t n;
        f_.read(reinterpret_cast<char*>(&w), 4); f_.read(reinterpret_cast<char*>(&h), 4);
        f_.read(reinterpret_cast<char*>(&fps), 8); f_.read(reinterpret_cast<char*>(&n), 8);
        width = (int)w; height = (int)h; this->fps = fps; n_frames = n;
    }
    double total_frames() const { return (double)n_frames; }
    double total_time() const { return (double)n_frames / fps; }                                 // video_capture.rs:34-36
    void read(uint64_t idx, uint8_t* dst) {
        const size_t fb = (size_t)width * height * 3;
        f_.seekg((std::streamoff)(28 + idx * fb));
        f_.read(reinterpret_cast<char*>(dst), (std::streamsize)fb);
    }
    const std::string& path() const { return path_; }
    int width = 0, height = 0; double fps = 0; uint64_t n_frames = 0;
private:
    std::string path_;
    std::ifstream f_;
};

namespace detail {
// one slideo_group = one matcher per GPU of the node behind one handle (include/slideo_amd.h, "N-device group"): the N-device
// counterpart of the reference's fan-out over the global rayon pool (mo/lib.rs:45,174,213)
struct Handle {
    slideo_group* g = nullptr;
    int n_devices = 1;
    ~Handle() { if (g) slideo_group_destroy(g); }
    void check(int32_t rc) const {
        if (rc != SLIDEO_OK) throw std::runtime_error(std::string("slideo_amd error ") + std::to_string(rc) + ": " + slideo_group_last_error(g));
    }
};
inline void tramp(void* user, uint64_t d, uint64_t t, const char* msg) {
    static_cast<const ProgressReporter*>(user)->report(d, t, msg ? msg : "");
}
}  // namespace detail

template <class I>
class VideoMatcherTask {                                       // matching::VideoMatcherTask (lib.rs:26-29)
public:
    virtual ~VideoMatcherTask() = default;
    virtual std::vector<Matching<I>> process() = 0;
};

template <class I>
class VideoMatcher {                                           // matching::VideoMatcher (lib.rs:16-24)
public:
    virtual ~VideoMatcher() = default;
    virtual std::unique_ptr<VideoMatcherTask<I>> match_images_with_video(const std::string& video_path, ProgressReporter reporter) = 0;
};

template <class I>
class HipVideoMatcherTask : public VideoMatcherTask<I> {       // OpenCVVideoMatcherTask (mo/lib.rs:161-246)
public:
    HipVideoMatcherTask(std::shared_ptr<detail::Handle> h, std::shared_ptr<std::vector<I>> images, std::string video_path,
                        ProgressReporter rep, float changed_similarity)
        : h_(std::move(h)), images_(std::move(images)), video_path_(std::move(video_path)), rep_(std::move(rep)) { (void)changed_similarity; }

    std::vector<Matching<I>> process() override {
        RawVideo video(video_path_);
        const double interval = 5.0;
        const double total_time = video.total_time();
        const uint64_t frames_to_process = (uint64_t)(total_time / interval);                       // mo/lib.rs:179
        std::vector<Matching<I>> results;
        results.push_back({total_time, (size_t)video.total_frames(), std::nullopt});                // sentinel, mo/lib.rs:185-189
        const std::string name = video_path_.substr(video_path_.find_last_of("/\\") + 1);
        uint64_t progress = 0;
        const double step = std::floor(video.fps * interval);
        const size_t fb = (size_t)video.width * video.height * 3;
        const int batch = 64 * h_->n_devices;             // one shard of 64 sampled frames per device and call
        std::vector<uint8_t> frames, prev_small, last_small;
        std::vector<std::pair<double, size_t>> meta;
        int sw = 0, sh = 0;
        auto flush = [&]() {
            if (meta.empty()) return;
            const int n = (int)meta.size();
            std::vector<uint8_t> changed(n);
            if (last_small.empty()) {       // size of the small image: ask once
                std::vector<uint8_t> tmp(fb);
                int32_t a, b;
                h_->check(slideo_small_image_bgr8(slideo_group_member(h_->g, 0), frames.data(), video.width, video.height, video.width * 3, tmp.data(), (int64_t)tmp.size(), &a, &b));
                sw = a; sh = b; last_small.resize((size_t)sw * sh * 3);
            }
            h_->check(slideo_group_changed_mask_bgr8(h_->g, n, frames.data(), video.width, video.height, video.width * 3, (int64_t)fb,
                                               prev_small.empty() ? nullptr : prev_small.data(), last_small.data(), changed.data(), nullptr));   // video_capture.rs:86-98
            prev_small = last_small;
            std::vector<int32_t> idx;
            for (int i = 0; i < n; ++i) if (changed[i]) idx.push_back(i);
            if (!idx.empty()) {
                std::vector<slideo_verdict> v(idx.size());
                // mo/lib.rs:213-214 on the copy of the frames the mask call left on the devices (no second upload)
                h_->check(slideo_group_match_kept_frames(h_->g, (int32_t)idx.size(), idx.data(), v.data()));
                for (size_t k = 0; k < idx.size(); ++k) {
                    std::optional<I> img;
                    if (v[k].page_idx >= 0) img = (*images_)[(size_t)v[k].page_idx];
                    results.push_back({meta[idx[k]].first, meta[idx[k]].second, img});
                }
            }
            for (int i = 0; i < n; ++i) rep_.report(++progress, frames_to_process, "Processing frames of '" + name + "'...");   // mo/lib.rs:192-203
            meta.clear(); frames.clear();
        };
        for (uint64_t idx = 0; idx < video.n_frames; ++idx) {                                      // VideoCaptureIter, video_capture.rs:42-57
            if (step <= 0 || !(std::fmod((double)idx, step) < 1.0)) continue;   // step 0 (fps < 0.2): `idx % 0.0` is NaN in the reference, nothing is retrieved (video_capture.rs:53)
            frames.resize(frames.size() + fb);
            video.read(idx, frames.data() + frames.size() - fb);
            meta.push_back({(double)idx / video.fps, (size_t)idx});
            if ((int)meta.size() >= batch) flush();
        }
        flush();
        rep_.report(frames_to_process, frames_to_process, "Finished!");                              // mo/lib.rs:223-227
        // mo/lib.rs:229-244: stable sort by time, drop consecutive mappings with the same image
        std::stable_sort(results.begin(), results.end(), [](const Matching<I>& a, const Matching<I>& b) { return a.video_time_s < b.video_time_s; });
        std::vector<Matching<I>> cleaned;
        for (auto& mm : results) {
            if (!cleaned.empty() && cleaned.back().image == mm.image) continue;
            cleaned.push_back(mm);
        }
        return cleaned;
    }
private:
    std::shared_ptr<detail::Handle> h_;
    std::shared_ptr<std::vector<I>> images_;
    std::string video_path_;
    ProgressReporter rep_;
};

template <class I>
class HipVideoMatcher : public VideoMatcher<I> {               // OpenCVVideoMatcher (mo/lib.rs:134-158)
public:
    HipVideoMatcher(std::shared_ptr<detail::Handle> h, std::shared_ptr<std::vector<I>> images) : h_(std::move(h)), images_(std::move(images)) {}
    std::unique_ptr<VideoMatcherTask<I>> match_images_with_video(const std::string& video_path, ProgressReporter reporter) override {
        RawVideo video(video_path);
        reporter.report(0, (uint64_t)(video.total_time() / 5.0), "");                               // mo/lib.rs:148-150
        return std::make_unique<HipVideoMatcherTask<I>>(h_, images_, video_path, std::move(reporter), 0.98f);
    }
private:
    std::shared_ptr<detail::Handle> h_;
    std::shared_ptr<std::vector<I>> images_;
};

// Drop-in for OpenCVImageVideoMatcher behind matching::ImageVideoMatcher (mo/lib.rs:34-73).
// I must provide `std::string get_path() const` (matching::MatchableImage, lib.rs:31-33) and operator==.
class HipImageVideoMatcher {
public:
    // device >= 0: that one device; -1 (default): every gfx950 device of the node (with_devices names them explicitly)
    explicit HipImageVideoMatcher(int device = -1, const slideo_config* cfg = nullptr, ImageLoader loader = load_image_bgr)
        : loader_(std::move(loader)) {
        if (device >= 0) devices_.push_back(device);
        slideo_config_default(&cfg_);
        if (cfg) cfg_ = *cfg;
    }
    HipImageVideoMatcher& with_devices(std::vector<int32_t> devices) { devices_ = std::move(devices); return *this; }
    // SIFT features + squared-L2 search instead of the reference's ORB + Hamming (slideo_group_use_sift).  ratio 0 (default): the
    // path's own 5 % tolerance vote on the L2 distances — the robust choice on decks whose pages share a template; ratio in
    // (0, 1]: Lowe's ratio test on the two nearest rows (the north_star's wording)
    HipImageVideoMatcher& with_sift(float ratio = 0.f) { sift_on_ = true; sift_ratio_ = ratio; return *this; }
    template <class I>
    std::unique_ptr<VideoMatcher<I>> create_video_matcher(std::vector<I> images, ProgressReporter reporter) const {
        auto h = std::make_shared<detail::Handle>();
        // no explicit list: every gfx950 device of the node, enumerated by the library (n_devices 0; no device at all: create reports it)
        int32_t rc = slideo_group_create(&cfg_, (int32_t)devices_.size(), devices_.empty() ? nullptr : devices_.data(), &h->g);
        if (rc != SLIDEO_OK) throw std::runtime_error(std::string("slideo_amd error ") + std::to_string(rc) + ": " + slideo_group_last_error(nullptr));
        h->n_devices = (int)slideo_group_device_count(h->g);
        if (sift_on_) {                                                                             // the north-star's SIFT + L2 front end
            slideo_sift_config sc;
            slideo_sift_config_default(&sc);
            h->check(slideo_group_use_sift(h->g, &sc, sift_ratio_));
        }
        h->check(slideo_group_set_progress(h->g, detail::tramp, &reporter));                         // "Analyzing PDF pages..." protocol, mo/lib.rs:43-58
        const size_t CH = 32 * (size_t)h->n_devices;
        for (size_t i = 0; i < images.size(); i += CH) {
            std::vector<Image8> dec;
            for (size_t j = i; j < std::min(images.size(), i + CH); ++j) dec.push_back(loader_(images[j].get_path()));
            std::vector<const uint8_t*> ptrs; std::vector<int32_t> w, hh, st;
            for (auto& d : dec) { ptrs.push_back(d.bgr.data()); w.push_back(d.w); hh.push_back(d.h); st.push_back(d.w * 3); }
            h->check(slideo_group_add_pages_bgr8(h->g, (int32_t)dec.size(), ptrs.data(), w.data(), hh.data(), st.data()));
        }
        h->check(slideo_group_set_progress(h->g, nullptr, nullptr));
        h->check(slideo_group_finalize_pages(h->g));                                              // FlannMatcher::new, mo/flann.rs:65-71
        return std::make_unique<HipVideoMatcher<I>>(h, std::make_shared<std::vector<I>>(std::move(images)));
    }
private:
    std::vector<int32_t> devices_;
    bool sift_on_ = false;
    float sift_ratio_ = 0.f;
    slideo_config cfg_;
    ImageLoader loader_;
};

}  // namespace slideo_host
