"""timing-only ablation sweep of the filtered kernel (development aid)"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import load_product, NOTO
vg = load_product()
m = vg.FontManager(True); fid = m.add_font_with_name("Noto Sans Regular", [NOTO]); hb = m.build_batch(fid)
ctx = vg.SdfContext(0)
for v in [0, 45, 30, 22, 12, 0, 45, 30, 22, 12]:
    ctx.set_variant(v); db = ctx.upload(hb.batch); db.time(3)
    print(f"variant {v:4d}: {db.time(30)/30:.4f} ms", flush=True)
