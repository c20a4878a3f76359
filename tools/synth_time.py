"""kernel timing on the synthetic stress workload (development aid)"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import load_product
vg = load_product()
from versatiles_glyphs_rs_amd import synthetic as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
variants = [int(v) for v in sys.argv[2:]] or [0, 1]
b = S.make_batch(0, n)
ctx = vg.SdfContext(0)
for v in variants:
    ctx.set_variant(v)
    db = ctx.upload(b)  # the tile list layout depends on the variant
    st = db.stats(); db.time(1)
    ms = db.time(3) / 3
    print(f"synthetic n={n} variant {v}: {ms:.3f} ms  {st['n_pairs']/ms*1e-6:.1f} Gpair/s  {n/ms*1e3:.3e} glyphs/s  "
          f"{st['n_pixels']/ms*1e-3:.1f} Mpx/s  cycles/pair/SIMD@2.4GHz {ms*1e-3*2.4e9*1024/st['n_pairs']:.3f}", flush=True)
