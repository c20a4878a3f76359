"""fonts -> PBF bytes on N device lanes of ONE process in the three lane forms (0 glyph shards + merge of every block, 1 whole
(font, block) tasks, 2 hybrid: whole tasks + the heaviest blocks split), best of K warm runs, beside one device.  On a one-GPU
box all lanes share the device (and its 16-CPU quota): a test of the host side and of "the hybrid costs nothing where it
cannot help", not a scaling measurement.   python tools/lane_forms.py [lanes ...]"""
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import FIRA, NOTO, load_product, noto_files  # noqa: E402
vg = load_product()
lanes_list = [int(a) for a in sys.argv[1:]] or [2, 4, 8]
K = 12


def best_of(m, r):
    m.render_glyphs(None, r)
    m.render_glyphs(None, r)
    best = 1e9
    for _ in range(K):
        t0 = time.perf_counter()
        m.render_glyphs(None, r)
        best = min(best, time.perf_counter() - t0)
    return best, m.timings()


sets = {"noto_regular": [("Noto Sans Regular", [NOTO])], "noto_all": [("Noto Sans Regular", noto_files())],
        "fira": [("Fira Sans Regular", [FIRA])],
        "21_fonts": [(f"Font {i:02d}", [p]) for i, p in enumerate([FIRA] + noto_files())]}
single = vg.Renderer.new_precise(0)
for name, fonts in sets.items():
    m = vg.FontManager(True)
    for disp, files in fonts:
        m.add_font_with_name(disp, files)
    b1, t1 = best_of(m, single)
    print(f"{name}: one device {b1 * 1e3:.3f} ms = {t1['glyphs'] / b1 / 1e6:.2f} M glyphs/s", flush=True)
    for n in lanes_list:
        lanes = vg.Renderer.new_multi([0] * n)
        row = []
        for form in (0, 1, 2):
            m.set_lane_form(form)
            t0 = time.perf_counter()
            m.render_glyphs(None, lanes)
            first = time.perf_counter() - t0
            b, t = best_of(m, lanes)
            fid = m.font_ids()[0]
            split = m.plan_lanes(fid, n)[1] if form else "-"
            row.append(f"form {form}: {b * 1e3:.3f} ms = {t['glyphs'] / b / 1e6:.2f} M/s (first run {first * 1e3:.1f} ms, split blocks of the first font {split})")
        print(f"  {n} lanes sharing the GPU | " + " | ".join(row), flush=True)
        lanes.close()
