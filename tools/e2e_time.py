"""end-to-end (fonts -> PBF bytes) timing of the host pipeline + GPU (development aid)"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import load_product, NOTO, noto_files
vg = load_product()
which = sys.argv[1] if len(sys.argv) > 1 else "noto_regular"
paths = [NOTO] if which == "noto_regular" else noto_files()
many = which == "many"  # every fixture file as its own font: 21 groups through the pipelined dispatcher
r = vg.Renderer.new_precise(0) if vg.device_count() else vg.Renderer.new_dummy()
fe = len(sys.argv) > 2 and sys.argv[2] == "fe"
for th in (4, 16):
    m = vg.FontManager(True); m.set_threads(th, int(sys.argv[3]) if len(sys.argv) > 3 else 0); m.set_device_front_end(fe)
    if many:
        from conftest import FIRA
        for i, p in enumerate([FIRA] + list(noto_files())):
            m.add_font_with_name(f"Font {i:02d}", [p])
    else:
        fid = m.add_font_with_name("Noto Sans Regular", paths)
    best = None
    for i in range(4):
        w = None if many else vg.DummyWriter(); t = time.perf_counter(); m.render_glyphs(w, r); dt = time.perf_counter() - t
        if i == 0: cold = dt
        best = dt if best is None else min(best, dt)
    tm = m.timings()
    print(f"{which} threads {th}: cold {cold*1e3:.1f} ms, best {best*1e3:.1f} ms -> {tm['glyphs']/best:.0f} glyphs/s | " +
          " ".join(f"{k[:-2]}={v*1e3:.2f}" for k, v in tm.items() if k.endswith('_s')), flush=True)
