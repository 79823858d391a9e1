// host_demo — drives the C++ trait-surface mirror the way crates/app/src/main.rs:69-93 drives the reference:
//   host_demo <pages.txt | page dir> <video.slvf> [nfeatures] [min_rating] [video_hash]      (SLIDEO_DEMO_DEVICES=0,0: the group's devices)
//   host_demo --dump-image <file.png|.ppm>
// pages.txt: one PPM/PNG path per line (page order); a directory is scanned like a pdftocairo target dir (p-<nr>.png).  Prints "time_ms page_nr" per timeline entry (page_nr 0 = None).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "matching.hpp"

using slideo_host::PdfPage;          // crates/app/src/pdf_to_images.rs:19-31

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: host_demo <pages.txt> <video.slvf> [nfeatures] [min_rating]\n"); return 2; }
    try {
        if (std::string(argv[1]) == "--dump-image") {          // raw BGR of one image to stdout (decoder test)
            auto im = slideo_host::load_image_bgr(argv[2]);
            std::printf("%d %d\n", im.w, im.h);
            std::fwrite(im.bgr.data(), 1, im.bgr.size(), stdout);
            return 0;
        }
        std::vector<PdfPage> pages;
        if (std::filesystem::is_directory(argv[1])) pages = slideo_host::scan_page_dir(argv[1], "pdfhash");   // pdftocairo target dir
        else {
            std::ifstream lst(argv[1]);
            std::string line;
            while (std::getline(lst, line)) if (!line.empty()) pages.push_back({pages.size() + 1, line, "", "pdfhash"});
        }
        slideo_config cfg; slideo_config_default(&cfg);
        if (argc > 3) cfg.nfeatures = std::atoi(argv[3]);
        if (argc > 4) cfg.min_rating = std::atof(argv[4]);
        uint64_t last = 0;
        slideo_host::ProgressReporter rep([&](uint64_t d, uint64_t t, const std::string& msg) { last = d; (void)t; (void)msg; });
        slideo_host::HipImageVideoMatcher matcher(-1, &cfg);             // every gfx950 device of the node, as the app would
        if (const char* e = std::getenv("SLIDEO_DEMO_DEVICES")) {          // e.g. "0,0": two members on one device (tests)
            std::vector<int32_t> devs;
            for (const char* p = e; *p;) { devs.push_back((int32_t)std::strtol(p, const_cast<char**>(&p), 10)); if (*p == ',') ++p; }
            matcher.with_devices(devs);
        }
        auto vm = matcher.create_video_matcher(pages, rep);
        auto task = vm->match_images_with_video(argv[2], rep);
        auto out = task->process();
        for (auto& m : out) std::printf("%lld %d\n", (long long)std::llround(m.video_time_s * 1000.0), m.image ? (int)m.image->page_nr : 0);
        if (argc > 5) {      // the app's output contract (db.rs:162-260): rows, then the viewer records of this pdf
            const auto rows = slideo_host::videos_mapping_rows(out);
            for (auto& r : rows) std::printf("row %u %s %u\n", r.video_ms, r.has_pdf ? r.pdf_hash.c_str() : "-", r.page);
            for (auto& pm : slideo_host::pdf_video_matchings(rows, "pdfhash", argv[5]))
                std::printf("pvm %u %s %s %u %u\n", pm.video_offset_ms, pm.pdf_hash.c_str(), pm.video_hash.c_str(), pm.page_idx, pm.duration_ms);
        }
        std::fprintf(stderr, "progress callbacks ended at %llu\n", (unsigned long long)last);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "host_demo: %s\n", e.what());
        return 1;
    }
    return 0;
}
