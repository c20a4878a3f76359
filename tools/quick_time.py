"""Ad-hoc kernel timing on the GPU box (development aid; bench.py is the contract)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import load_product, NOTO
from oracle import oracle as O
import numpy as np

vg = load_product()
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
f = O.Font(NOTO)
cps = f.codepoints(); cps = cps[cps <= 0xFFFF]
jobs = []
for cp in cps:
    r = f.prepare_glyph(int(cp))
    if r and r[0].has_bitmap:
        jobs.append(r)
batch = vg.make_batch((s, i.x0, i.y0, i.w, i.h) for i, s in jobs)
ctx = vg.SdfContext(0)
ctx.set_variant(variant)
db = ctx.upload(batch)
st = db.stats()
db.time(3)
for it in (1, 10, 50):
    ms = db.time(it) / it
    print(f"variant {variant} iters {it}: {ms:.4f} ms/launch  {st['n_glyphs']/ms*1e3:.3e} glyphs/s  "
          f"{st['n_pairs']/ms*1e-6:.1f} Gpair/s  {st['alg_bytes']/ms*1e-6:.2f} GB/s alg", flush=True)
print(st)
