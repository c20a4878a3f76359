"""Whole-font parity of one kernel variant against the oracle + timing (development aid)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import load_product, NOTO
from oracle import oracle as O
import numpy as np

vg = load_product()
variants = [int(a) for a in sys.argv[1:]] or [30]
f = O.Font(NOTO)
cps = f.codepoints(); cps = cps[cps <= 0xFFFF]
jobs = []
for cp in cps:
    r = f.prepare_glyph(int(cp))
    if r and r[0].has_bitmap:
        jobs.append(r)
batch = vg.make_batch((s, i.x0, i.y0, i.w, i.h) for i, s in jobs)
want = O.sdf_render_batch(batch)[0]
ctx = vg.SdfContext(0)
for v in variants:
    ctx.set_variant(v)
    db = ctx.upload(batch)
    db.launch()
    got = db.download()
    msg = ""
    if v < 31:
        diff = np.flatnonzero(np.asarray(got) != np.asarray(want))
        msg = f"parity: {diff.size} bytes differ of {len(got)}"
    db.time(3)
    ms = min(db.time(20) / 20 for _ in range(3))
    st = db.stats()
    print(f"variant {v}: {ms:.4f} ms/launch {st['n_glyphs']/ms*1e3:.3e} glyphs/s {msg}", flush=True)
