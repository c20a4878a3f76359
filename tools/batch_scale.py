"""kernel time vs batch size (the Noto Sans Regular batch replicated k times): how much of a
launch is tail / under-filled chip (development aid)"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import load_product, NOTO
from oracle import oracle as O
vg = load_product()
f = O.Font(NOTO)
cps = f.codepoints(); cps = cps[cps <= 0xFFFF]
jobs = []
for cp in cps:
    r = f.prepare_glyph(int(cp))
    if r and r[0].has_bitmap:
        jobs.append(r)
ctx = vg.SdfContext(0)
ctx.set_variant(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
for k in ((1, 2, 4, 8) if len(sys.argv) < 3 else (int(sys.argv[2]),)):
    batch = vg.make_batch([(s, i.x0, i.y0, i.w, i.h) for i, s in jobs] * k)
    db = ctx.upload(batch)
    db.time(3)
    ms = min(db.time(20) / 20 for _ in range(3))
    print(f"x{k}: {ms:.4f} ms/launch, {ms/k:.4f} ms per replica, {len(jobs)*k/ms*1e3:.3e} glyphs/s", flush=True)
