#!/usr/bin/env python3
"""Regenerates tests/golden/ from the reference checkout (run in the build container only).

  pngsuite/{common,ios}/*.png   the reference's own decode inputs
                                (Sources/PNGIntegrationTests/Inputs/{Common,iOS}), copied verbatim:
                                they are PngSuite test images, i.e. data, not source code.
  pngsuite.json                 per file: sha256 of the RGBA16 pixels the reference's golden
                                (Sources/PNGIntegrationTests/RGBA/<name>.png.rgba, premultiplied in
                                8-bit for the iOS set exactly as Roundtripping.swift:206-215 does)
                                says the decode must unpack to, plus sha256 of PNG.Image.storage as
                                produced by the CPU oracle *after* it matched that golden here.

  invalid/*.png                 the reference's negative lexing inputs (Inputs/Invalid, 14 PngSuite "x" files):
                                bad signatures, the two bad chunk checksums pinned in
                                PNGIntegrationTests/ErrorHandling.swift:30,42, bad IHDR codes, missing IDAT.

The GPU box has no /root/reference; its tests read only what this script wrote.
"""
import hashlib, json, shutil, sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
from pnghelp import REFERENCE, orc_decode, parse_png, premultiply8, unpack_rgba16  # noqa: E402


def main():
    base = REFERENCE / "Sources" / "PNGIntegrationTests"
    table = {}
    for sub in ("Common", "iOS"):
        dst = HERE / "pngsuite" / sub.lower()
        dst.mkdir(parents=True, exist_ok=True)
        for f in sorted((base / "Inputs" / sub).glob("*.png")):
            png = parse_png(f.read_bytes())
            gold = np.frombuffer((base / "RGBA" / (f.name + ".rgba")).read_bytes(), dtype="<u2").reshape(-1, 4)
            if png.ios:
                gold = premultiply8(gold)
            st, storage, _ = orc_decode(png)
            got = unpack_rgba16(storage, png)
            assert st == 0 and got.shape == gold.shape and (got == gold).all(), f.name
            shutil.copyfile(f, dst / f.name)
            table[f"{sub.lower()}/{f.name}"] = {
                "rgba16_sha256": hashlib.sha256(np.ascontiguousarray(gold).astype("<u2").tobytes()).hexdigest(),
                "storage_sha256": hashlib.sha256(storage.tobytes()).hexdigest(),
            }
    (HERE / "pngsuite.json").write_text(json.dumps(table, indent=1, sort_keys=True) + "\n")
    inv = HERE / "invalid"
    inv.mkdir(exist_ok=True)
    for f in sorted((base / "Inputs" / "Invalid").glob("*.png")):
        shutil.copyfile(f, inv / f.name)
    print(len(table), "fixtures written")

    # encoder goldens: swift-png's own committed level-9 outputs (Tests/Outputs, written by
    # PNGCompressionTests/Compression.swift:56) next to the inputs they were made from
    # (Tests/Baselines).  All 28 inputs are copied (5 MB) and all 28 outputs digested, so that the GPU box --
    # which has no reference checkout -- runs the device's graph search on every one of them, the 16-bit
    # photographic ones included; five small outputs are copied too, for a byte-wise diff when a digest fails.
    enc = HERE / "encode"
    enc.mkdir(exist_ok=True)
    keep = {"indexed8-color-nonphotographic", "v8-monochrome-nonphotographic", "va8-monochrome-nonphotographic",
            "rgba8-monochrome-nonphotographic", "indexed8-monochrome-nonphotographic"}
    digests = {}
    for f in sorted((REFERENCE / "Tests" / "Baselines").glob("*.png")):
        gold = parse_png((REFERENCE / "Tests" / "Outputs" / f.name).read_bytes())
        digests[f.stem] = {"idat_sha256": hashlib.sha256(gold.idat).hexdigest(), "idat_len": len(gold.idat)}
        shutil.copyfile(f, enc / (f.stem + ".baseline.png"))
        if f.stem in keep:
            shutil.copyfile(REFERENCE / "Tests" / "Outputs" / f.name, enc / (f.stem + ".swiftpng9.png"))
    (HERE / "encode.json").write_text(json.dumps(digests, indent=1, sort_keys=True) + "\n")
    print(len(digests), "encoder goldens digested,", len(keep), "copied")


if __name__ == "__main__":
    main()
