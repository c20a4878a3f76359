#!/usr/bin/env python3
"""Generates the golden fixtures in tests/golden/ with the CPU oracle, IN THIS CONTAINER.

The reference (Rust) cannot be built or run here, so these vectors come from the oracle
AFTER it reproduced every known-answer test of the reference (tests/test_oracle_kat.py) and
after the independent C++ host stage agreed with it bit for bit (tests/test_host_vs_oracle.py).
They pin the oracle against regressions and give the GPU tests reference bytes that do not
need the oracle at run time.  Fixtures are data only: inputs and expected outputs.

  glyphs_<set>.csv     debug-style rows (commands/debug.rs:56-64 shape + two columns):
                       codepoint,width,height,left,top,advance,bitmap_size,n_segments,sha256(bitmap)
  pbf_sha256.json      {set: {block_start: sha256(id-sorted PBF bytes)}}, precise renderer
  samples.npz          full segment lists + rects + bitmaps of a few dozen glyphs
  synthetic64.npz      bitmaps of the first 64 synthetic outlines (seed 0x5DF61F95)
  synthetic8192_sha256.json  SHA-256 of the oracle's output for outlines [0, 8192): one rank's benchmark batch

usage: python tests/golden/make_golden.py
"""
import hashlib
import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from conftest import FIRA, NOTO, load_product, noto_files  # noqa: E402
from oracle import oracle as O  # noqa: E402

SETS = {
    "fira": ("fira_sans_regular", [FIRA]),
    "noto_regular": ("noto_sans_regular", [NOTO]),
    "noto_all": ("noto_sans_regular", noto_files()),
}


def main():
    pbf = {}
    samples = {}
    rng = np.random.default_rng(20260101)
    for name, (fid, paths) in SETS.items():
        fonts = [O.Font(p) for p in paths]
        prov = {}
        for fi, f in enumerate(fonts):
            for cp in f.codepoints():
                if cp <= 0xFFFF:
                    prov.setdefault(int(cp), fi)
        rows = ["codepoint,width,height,left,top,advance,bitmap_size,n_segments,sha256"]
        rendered = []
        for cp in sorted(prov):
            r = fonts[prov[cp]].render_glyph(cp, O.PRECISE)
            if r is None:
                continue
            info, bm = r
            sha = hashlib.sha256(bm.tobytes()).hexdigest() if bm is not None else ""
            rows.append(f"{cp},{info.width},{info.height},{info.left},{info.top},{info.advance},"
                        f"{0 if bm is None else bm.size},{info.n_segments},{sha}")
            if bm is not None:
                rendered.append((cp, prov[cp], info.n_segments, bm.size))
        (HERE / f"glyphs_{name}.csv").write_text("\n".join(rows) + "\n")
        pbf[name] = {}
        for blk in range(256):
            data, _, _ = O.render_block(fonts, fid, blk * 256, O.PRECISE)
            pbf[name][str(blk * 256)] = hashlib.sha256(data).hexdigest()
        # samples: min / median / max N and w*h, plus random ones
        if name != "noto_regular":
            by_n = sorted(rendered, key=lambda t: t[2])
            by_px = sorted(rendered, key=lambda t: t[3])
            pick = {by_n[0][0], by_n[len(by_n) // 2][0], by_n[-1][0], by_px[0][0], by_px[-1][0]}
            pick |= {int(rendered[i][0]) for i in rng.choice(len(rendered), 10, replace=False)}
            for cp in sorted(pick):
                f = fonts[prov[cp]]
                info, segs = f.prepare_glyph(cp)
                _, bm = f.render_glyph(cp, O.PRECISE)
                k = f"{name}_{cp}"
                samples[k + "_segs"] = segs
                samples[k + "_rect"] = np.array([info.x0, info.y0, info.w, info.h], dtype=np.int64)
                samples[k + "_bitmap"] = bm
        print(name, len(rows) - 1, "glyph rows")
    (HERE / "pbf_sha256.json").write_text(json.dumps(pbf, indent=0, sort_keys=True) + "\n")
    np.savez_compressed(HERE / "samples.npz", **samples)

    vg = load_product()
    from versatiles_glyphs_rs_amd import synthetic as S
    batch = S.make_batch(0, 64)
    out, _ = O.sdf_render_batch(batch, O.PRECISE, 8)
    np.savez_compressed(HERE / "synthetic64.npz", bitmaps=out.reshape(64, S.H, S.W))
    print("samples:", len(samples) // 3, "synthetic64 sha", hashlib.sha256(out.tobytes()).hexdigest()[:16])
    big, _ = O.sdf_render_batch(S.make_batch(0, 8192), O.PRECISE, 8)
    (HERE / "synthetic8192_sha256.json").write_text(json.dumps({
        "outlines": [0, 8192], "bytes": int(len(big)), "sha256": hashlib.sha256(big.tobytes()).hexdigest(),
        "generator": "versatiles-glyphs-rs_amd/synthetic.py make_batch(0, 8192); oracle PRECISE"}, indent=1) + "\n")


if __name__ == "__main__":
    main()
