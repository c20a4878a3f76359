"""front-end timing of the same glyphs as glyf (quadratics) and as CFF (cubics) outlines (development aid)"""
import io, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import load_product, NOTO
from fontTools.ttLib import TTFont
from fontTools.pens.t2CharStringPen import T2CharStringPen
import test_cff_outlines as T
vg = load_product()
src = TTFont(NOTO); gs = src.getGlyphSet(); order = src.getGlyphOrder()[:3000]
cmap = {cp: g for cp, g in src.getBestCmap().items() if g in order}
cs = {}
for g in order:
    pen = T2CharStringPen(gs[g].width, gs); gs[g].draw(pen); cs[g] = pen.getCharString()
cff = T._build(order, cmap, cs, {g: gs[g].width for g in order}, src["head"].unitsPerEm)
ctx = vg.SdfContext(0)
for name, data in (("glyf", Path(NOTO).read_bytes()), ("cff", cff)):
    mgr = vg.FontManager(parallel=False); fid = mgr.add_font_data("X", data); o = mgr.record_outlines(fid)
    if name == "glyf":
        keep = np.isin(o["ids"], np.array(sorted(cmap)))
    best = 1e9
    for _ in range(6):
        t = time.perf_counter(); rects, ob, ns = ctx.outlines_prepare(o["cmd_off"], o["cmds"], o["scale"], o["shift_x"]); best = min(best, time.perf_counter() - t)
    kinds = np.bincount(o["cmds"]["kind"], minlength=5)
    print(f"{name}: {len(o['ids'])} glyphs, {len(o['cmds'])} commands (quad {kinds[2]}, cubic {kinds[3]}), {ns} segments: prepare {best*1e3:.3f} ms")
