"""Does the heaviest workgroup set the makespan of a single launch?  Times the Noto Sans Regular batch with and
without its glyphs of more than N segments (development aid)."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import NOTO, load_product
vg = load_product()
m = vg.FontManager(True); fid = m.add_font_with_name("Noto Sans Regular", [NOTO]); hb = m.build_batch(fid); b = hb.batch
ctx = vg.SdfContext(0)
nseg = np.diff(b.seg_off.astype(np.int64))
def sub(keep):
    gl = []
    for g in np.nonzero(keep)[0]:
        a, e = int(b.seg_off[g]), int(b.seg_off[g + 1])
        gl.append((np.stack([b.seg_sx[a:e], b.seg_sy[a:e], b.seg_ex[a:e], b.seg_ey[a:e]], axis=1), int(b.x0[g]), int(b.y0[g]), int(b.w[g]), int(b.h[g])))
    return vg.make_batch(gl)
for thr in (10**9, 2048, 1024, 768):
    keep = nseg <= thr
    bb = sub(keep)
    db = ctx.upload(bb); db.time(5)
    ms = min(db.time(50) / 50 for _ in range(3)); st = db.stats(); db.free()
    print(f"glyphs with <= {thr} segments: {int(keep.sum())} glyphs, {st['n_tiles']} workgroups, pairs {st['n_pairs']/1e6:.0f} M: {ms*1e3:.2f} us", flush=True)
only = sub(nseg > 1024); db = ctx.upload(only); db.time(5); print("only > 1024 segments:", only.n_glyphs, "glyphs", f"{min(db.time(50)/50 for _ in range(3))*1e3:.2f} us")
