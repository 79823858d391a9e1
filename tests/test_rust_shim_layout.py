"""crates/matching-hip (the Rust shim a maintainer adds; uncompiled here — no cargo/rustc in the image) is held to the C
ABI mechanically: its #[repr(C)] structs must list the fields of include/slideo_amd.h in order with matching types (via the
ctypes mirror, whose layout tests/test_capi_load.py pins against the compiled library), every function it declares must be
exported by libslideo_amd.so, and its ABI constant must be the header's."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FFI = os.path.join(ROOT, "crates", "matching-hip", "src", "ffi.rs")
RUST2C = {"i32": C.c_int32, "u32": C.c_uint32, "f32": C.c_float, "f64": C.c_double, "i64": C.c_int64}


def _struct_fields(src, name):
    body = re.search(r"pub struct %s \{(.*?)\n\}" % name, src, re.S).group(1)
    body = re.sub(r"//[^\n]*", "", body)
    return re.findall(r"pub (\w+): (\w+)", body)


def test_ffi_structs_follow_the_header(capi):
    src = open(FFI).read()
    for rust_name, ct in (("slideo_ocv_variants", capi.OcvVariants), ("slideo_config", capi.Config)):
        got = _struct_fields(src, rust_name)
        want = list(ct._fields_)
        assert [n for n, _ in got] == [n for n, _ in want], rust_name
        for (n, rt), (_, cty) in zip(got, want):
            if rt in RUST2C:
                assert RUST2C[rt] is cty, (rust_name, n, rt, cty)
            else:
                assert rt == "slideo_ocv_variants" and cty is capi.OcvVariants
    assert [n for n, _ in _struct_fields(src, "slideo_verdict")] == list(capi.VERDICT_DTYPE.names)
    assert "168" in src and C.sizeof(capi.Config) == 168                      # assert_abi()'s size check matches


def test_ffi_functions_exist_and_abi_constant_matches():
    src = open(FFI).read()
    hdr = open(os.path.join(ROOT, "include", "slideo_amd.h")).read()
    fns = re.findall(r"pub fn (slideo_\w+)\(", src)
    assert len(fns) >= 9 and len(set(fns)) == len(fns)
    for f in fns:
        assert re.search(r"\b%s\(" % f, hdr), f
    from slideo_amd import _capi
    assert set(fns) <= set(_capi.EXPORTS)
    abi = int(re.search(r"SLIDEO_ABI_VERSION: u32 = (\d+)", src).group(1))
    assert abi == int(re.search(r"#define SLIDEO_ABI_VERSION (\d+)", hdr).group(1))


def test_crate_tree_is_complete():
    d = os.path.join(ROOT, "crates", "matching-hip")
    for f in ("Cargo.toml", "build.rs", "src/ffi.rs", "src/lib.rs", "src/decode.rs"):
        assert os.path.getsize(os.path.join(d, f)) > 200, f
    lib = open(os.path.join(d, "src", "lib.rs")).read()
    for needle in ("impl<'i> ImageVideoMatcher<'i> for HipImageVideoMatcher", "VideoMatcher<'i, I> for HipVideoMatcher<I>",
                   "VideoMatcherTask<I> for HipVideoMatcherTask<I>", "Analyzing PDF pages...", "PDF page analysis successful.",
                   "Processing frames of '{}'...", "Finished!", "Mutex<RawHandle>", "assert_abi()"):
        assert needle in lib, needle
    assert "unsafe impl Sync" not in lib                                        # the handle is not re-entrant (slideo_amd.h)
