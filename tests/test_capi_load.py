"""The C-ABI library loads on a CPU-only box and exports every symbol that
include/slideo_amd.h declares; compute calls fail loudly without a GPU."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "slideo_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(slideo_[a-z0-9_]+)\s*\(", src)) - {"slideo_progress_fn"})


def test_library_exports_every_declared_symbol(capi):
    L = capi.lib()
    names = _declared_symbols()
    assert len(names) >= 19
    for n in names:
        assert hasattr(L, n), "missing export " + n
    assert sorted(capi.EXPORTS) == names
    assert L.slideo_abi_version() == 7        # ABI 7: + slideo_matcher_read_shader_clock (6: verdict_rule, slideo_device_list, n_devices 0; ocv.hdlt defaults to 1)


def test_config_struct_matches_oracle_layout(capi, oracle):
    import ctypes as C
    a, b = capi.default_config(), oracle.default_config()
    assert C.sizeof(a) == C.sizeof(b) == 168        # ABI 6: + verdict_rule (160 since ABI 4: verify_model, matcher, lsh_*, ocv.hdlt)
    assert bytes(a) == bytes(b)


def test_unsupported_config_is_rejected(capi):
    from conftest import HAS_GPU
    with pytest.raises(capi.SlideoError) as e:
        capi.Matcher(capi.default_config(patch_size=31))
    assert e.value.code in (5,)      # UNSUPPORTED is checked before the device probe


def test_no_silent_cpu_fallback(capi):
    from conftest import HAS_GPU
    if HAS_GPU:
        pytest.skip("GPU present")
    with pytest.raises(capi.SlideoError) as e:
        capi.Matcher()
    assert e.value.code == 2          # SLIDEO_ERR_NO_DEVICE


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "slideo_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                assert "pyoracle" not in txt and "liboracle" not in txt and "slideo_oracle" not in txt, f
