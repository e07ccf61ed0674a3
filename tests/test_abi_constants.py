"""The Python host's copy of the C ABI's vocabulary (status codes, formats, kernel ids, knobs) is the header's:
every SPNG_* enumerator the binding mirrors has the value include/spng_mi355.h gives it."""
import re
from pathlib import Path

import swift_png_amd as spng

HEADER = (Path(__file__).resolve().parent.parent / "include" / "spng_mi355.h").read_text()


def enumerators():
    return {m.group(1): int(m.group(2), 0) for m in re.finditer(r"\bSPNG_([A-Z0-9_]+)\s*=\s*(0x[0-9a-fA-F]+|\d+)", HEADER)}


def test_python_constants_match_header():
    enums = enumerators()
    assert len(enums) >= 50
    checked = 0
    for name, value in enums.items():
        for candidate in (name, name.replace("INFLATE_", "INFLATE_", 1)):
            if hasattr(spng, candidate):
                assert getattr(spng, candidate) == value, (name, getattr(spng, candidate), value)
                checked += 1
                break
    assert checked >= 48, checked
    # every status the header defines has a name in the error mirror (or is one of the two non-errors)
    for name, value in enums.items():
        if name.startswith("E_"):
            assert value in spng._NAMES, name


def test_status_ranges_route_to_the_mirrored_error_types():
    for status, kind in ((spng.E_CHECK_BITS, spng.StreamHeaderError), (spng.E_GZIP_FLAG_BITS, spng.GzipStreamHeaderError),
                         (spng.E_HUFFMAN_TABLE, spng.DecompressionError), (spng.E_EXTRANEOUS_IMAGE_DATA, spng.DecodingError),
                         (spng.E_CHUNK_CHECKSUM, spng.LexingError), (spng.E_DEVICE, spng.SpngError)):
        try:
            spng.raise_for(status, (1, 2))
        except spng.SpngError as e:
            assert type(e) is kind and e.status == status and e.aux == (1, 2)
        else:
            raise AssertionError(status)
    spng.raise_for(spng.DONE); spng.raise_for(spng.NEED_MORE_INPUT)
