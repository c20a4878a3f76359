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
    if v == 44:
        a = np.asarray(got)
        w = np.array([(a[i:i+64] == 2).any() for i in range(0, len(a), 64)])
        msg = f"rescan pixels: {(a == 2).mean():.4%}; 64-byte runs with one: {w.mean():.2%}"
    elif v == 43:
        a = np.asarray(got)
        w = [a[i:i+64].min() for i in range(0, len(a), 64)]
        msg = f"exact-path pixels: {(a == 0).mean():.4%}; 64-byte runs with one: {np.mean(np.array(w) == 0):.2%}"
    elif v < 31 or v in (45, 50):
        diff = np.flatnonzero(np.asarray(got) != np.asarray(want))
        msg = f"parity: {diff.size} bytes differ of {len(got)}"
    db.time(3)
    ms = min(db.time(20) / 20 for _ in range(3))
    st = db.stats()
    print(f"variant {v}: {ms:.4f} ms/launch {st['n_glyphs']/ms*1e3:.3e} glyphs/s {msg}", flush=True)
