"""end-to-end (fonts -> PBF bytes, native NULL sink) against the size of the host pool and the group policy of the
pipelined dispatcher (development aid).   python tools/e2e_sweep.py {noto_regular,noto_all,many} threads...
VG_FE_MIN_GROUP (glyphs a group keeps at least) is read by the library."""
import os
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import FIRA, NOTO, load_product, noto_files  # noqa: E402
vg = load_product()
which = sys.argv[1]
r = vg.Renderer.new_precise(0)
for th in [int(a) for a in sys.argv[2:]] or [16]:
    m = vg.FontManager(True)
    m.set_threads(th, 0)
    if which == "many":
        for i, p in enumerate([FIRA] + list(noto_files())):
            m.add_font_with_name(f"Font {i:02d}", [p])
    else:
        m.add_font_with_name("Noto Sans Regular", [NOTO] if which == "noto_regular" else noto_files())
    m.render_glyphs(None, r)
    best, tm = None, None
    for i in range(20):
        t = time.perf_counter(); m.render_glyphs(None, r); dt = time.perf_counter() - t
        if best is None or dt < best:
            best, tm = dt, m.timings()
    print(f"{which} threads {th} min_group {os.environ.get('VG_FE_MIN_GROUP', '-')}: best {best*1e3:.3f} ms -> {tm['glyphs']/best/1e6:.2f} M glyphs/s | " +
          " ".join(f"{k[:-2]}={v*1e3:.3f}" for k, v in tm.items() if k.endswith('_s')), flush=True)
