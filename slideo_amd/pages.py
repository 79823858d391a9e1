"""Page ingest as the app crate does it: the page list of one PDF from a pdftocairo target dir.

`pdftocairo <pdf> <dir>/p -png` names its outputs `p-<nr>.png` (zero padded); the reference lists the directory,
takes the number after "p-" as the 1-based page_nr, sorts by it (crates/pdftocairo/src/pdftocairo.rs:216-231) and
wraps each entry as PdfPage{page_nr, image_path, pdf_path, pdf_hash} (crates/app/src/pdf_to_images.rs:19-31,
138-146), which is the `I: MatchableImage` handed to create_video_matcher (crates/app/src/main.rs:71-72).
PNG decode stays on the host (PIL here, slideo_amd/host/png.hpp in the C++ mirror); pixels reach the GPU through
slideo_matcher_add_pages_bgr8.
"""
import os
from dataclasses import dataclass
from typing import List


@dataclass(frozen=True)
class PdfPage:
    page_nr: int            # 1-based (pdf_to_images.rs:20-21)
    image_path: str
    pdf_path: str = ""
    pdf_hash: str = ""

    def get_path(self):     # MatchableImage (pdf_to_images.rs:33-37)
        return self.image_path


def scan_page_dir(target_dir: str, pdf_hash: str = "", pdf_path: str = "") -> List[PdfPage]:
    pages = []
    for file_name in os.listdir(target_dir):
        stem = file_name.split(".")[0]                     # pdftocairo.rs:222
        digits = stem[2:]                                  # pdftocairo.rs:223
        if not digits.isdigit():                           # `.parse().unwrap()` panics in the reference
            raise ValueError("unexpected file %r in page directory %r" % (file_name, target_dir))
        pages.append(PdfPage(int(digits), os.path.join(target_dir, file_name), pdf_path, pdf_hash))
    pages.sort(key=lambda p: p.page_nr)                    # pdftocairo.rs:231
    return pages
