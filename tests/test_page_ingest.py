"""page ingest — pdftocairo directory naming and the C++ mirror's PNG decoder (host only)."""
import os
import subprocess

import numpy as np
import pytest
from PIL import Image

from slideo_amd import pages as pg

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_scan_page_dir_orders_by_page_number(tmp_path):
    for name in ["p-10.png", "p-02.png", "p-1.png", "p-003.png"]:
        open(os.path.join(tmp_path, name), "wb").close()
    got = pg.scan_page_dir(str(tmp_path), pdf_hash="h", pdf_path="deck.pdf")
    assert [p.page_nr for p in got] == [1, 2, 3, 10]                       # pdftocairo.rs:223-231
    assert [os.path.basename(p.get_path()) for p in got] == ["p-1.png", "p-02.png", "p-003.png", "p-10.png"]
    assert all(p.pdf_hash == "h" and p.pdf_path == "deck.pdf" for p in got)


def test_scan_page_dir_rejects_foreign_files(tmp_path):
    open(os.path.join(tmp_path, "p-01.png"), "wb").close()
    open(os.path.join(tmp_path, "notes.txt"), "wb").close()                # `"tes".parse::<u32>().unwrap()` panics
    with pytest.raises(ValueError):
        pg.scan_page_dir(str(tmp_path))


@pytest.fixture(scope="module")
def host_demo():
    from slideo_amd import build
    if not os.path.exists(build.HIP_LIB):
        pytest.skip("libslideo_amd.so not built")
    return build.build_host_demo()


def _dump(exe, path):
    out = subprocess.run([exe, "--dump-image", path], capture_output=True, timeout=120)
    assert out.returncode == 0, out.stderr.decode()
    head, _, body = out.stdout.partition(b"\n")
    w, h = (int(x) for x in head.split())
    return np.frombuffer(body, np.uint8).reshape(h, w, 3)


def _pil_bgr(path):
    return np.ascontiguousarray(np.array(Image.open(path).convert("RGB"))[:, :, ::-1])


def test_cpp_png_decoder_matches_pil_on_reference_fixtures(host_demo):
    names = [n for n in sorted(os.listdir(GOLDEN)) if n.endswith(".png")]
    assert names
    for n in names:
        p = os.path.join(GOLDEN, n)
        assert np.array_equal(_dump(host_demo, p), _pil_bgr(p)), n


@pytest.mark.parametrize("mode", ["L", "RGB", "RGBA", "LA", "P"])
def test_cpp_png_decoder_colour_types(host_demo, tmp_path, mode):
    rng = np.random.default_rng(7)
    h, w = 37, 53
    smooth = (np.add.outer(np.arange(h) * 3, np.arange(w) * 2) % 256).astype(np.uint8)     # exercises Sub/Up/Paeth filters
    if mode == "L":
        im = Image.fromarray(smooth, "L")
    elif mode == "LA":
        im = Image.fromarray(np.dstack([smooth, rng.integers(0, 256, (h, w), dtype=np.uint8)]), "LA")
    elif mode == "P":
        im = Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8), "RGB").quantize(17)
    else:
        arr = np.dstack([smooth, smooth[::-1], rng.integers(0, 256, (h, w), dtype=np.uint8)] + ([smooth.T[:h, :w] if False else smooth] if mode == "RGBA" else []))
        im = Image.fromarray(arr, mode)
    path = os.path.join(tmp_path, "t.png")
    im.save(path, optimize=True)
    assert np.array_equal(_dump(host_demo, path), _pil_bgr(path))


def test_cpp_decoder_fails_loudly(host_demo, tmp_path):
    bad = os.path.join(tmp_path, "bad.png")
    open(bad, "wb").write(b"not a png at all")
    out = subprocess.run([host_demo, "--dump-image", bad], capture_output=True, timeout=60)
    assert out.returncode != 0 and b"not a PNG" in out.stderr
    Image.fromarray(np.zeros((4, 4), np.uint16)).save(os.path.join(tmp_path, "d16.png"))
    out = subprocess.run([host_demo, "--dump-image", os.path.join(tmp_path, "d16.png")], capture_output=True, timeout=60)
    assert out.returncode != 0 and b"8-bit" in out.stderr
