"""CPU-side checks of the C-ABI library: it loads, exports every symbol the header declares, the
pure helpers agree with the oracle, and the product refuses to run without a GPU (no fallback)."""
import re

import pytest

import pnghelp as ph
import swift_png_amd as spng


def test_exports_match_header():
    header = (ph.ROOT / "include" / "spng_mi355.h").read_text()
    declared = set(re.findall(r"\b(spng_[a-z0-9_]+)\s*\(", header))
    declared -= {"spng_ctx", "spng_result", "spng_stream_desc", "spng_image_desc"}
    lib = spng.load_library()
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert declared == set(spng.EXPORTS)
    assert lib.spng_version() == 0x000100


def test_status_codes_agree_with_oracle_header():
    h1 = (ph.ROOT / "include" / "spng_mi355.h").read_text()
    h2 = (ph.ROOT / "oracle" / "spng_oracle.h").read_text()
    a = dict(re.findall(r"SPNG_(E?_?[A-Z_]+)\s*=\s*(\d+)", h1))
    b = dict(re.findall(r"ORC_(E?_?[A-Z_]+)\s*=\s*(\d+)", h2))
    assert b and all(a[k] == v for k, v in b.items())


@pytest.mark.parametrize("w,h,depth,ch,il", [(4096, 4096, 8, 4, 0), (8192, 8192, 16, 4, 1), (256, 256, 8, 4, 0),
                                             (1, 1, 1, 1, 1), (7, 3, 2, 1, 1), (33, 9, 4, 1, 0), (5, 5, 16, 3, 1)])
def test_geometry_matches_oracle(w, h, depth, ch, il):
    lib, orc = spng.load_library(), ph.oracle()
    assert lib.spng_inflated_size(w, h, depth, ch, il) == orc.orc_inflated_size(w, h, depth, ch, il)
    assert lib.spng_storage_size(w, h, depth, ch) == orc.orc_storage_size(w, h, depth, ch)


def test_survey_sizes():
    lib = spng.load_library()
    assert lib.spng_inflated_size(4096, 4096, 8, 4, 0) == 67112960
    assert lib.spng_inflated_size(8192, 8192, 16, 4, 1) == 536886272
    assert lib.spng_inflated_size(256, 256, 8, 4, 0) == 262400


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        spng.load()
    with pytest.raises(RuntimeError):
        spng.LZ77.Inflator()


def test_product_never_touches_oracle():
    for p in (ph.ROOT / "swift_png_amd").rglob("*"):
        if p.suffix in (".py", ".hip", ".hpp", ".cpp", ".sh") and p.is_file():
            assert "oracle" not in p.read_text().replace("no oracle", ""), p
