"""The calling thread's time line of end-to-end runs (VG_TRACE_PHASES=1: one stderr line per run with the start of every phase of
every group, us from the start of the run; VG_TRACE_PACK=1 adds the pack breakdown).  Prints the last three warm runs.
    python tools/phase_trace.py [noto_regular|noto_all|fira|many] [runs]"""
import os
import subprocess
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
if os.environ.get("_VG_PH_CHILD") != "1":
    env = dict(os.environ, VG_TRACE_PHASES="1", VG_TRACE_PACK="1", _VG_PH_CHILD="1")
    cp = subprocess.run([sys.executable, __file__] + sys.argv[1:], env=env, capture_output=True, text=True)
    lines = [ln for ln in cp.stderr.splitlines() if ln.startswith("[phases]") or ln.startswith("[pack]")]
    k = [i for i, ln in enumerate(lines) if ln.startswith("[phases]")]
    start = k[-4] + 1 if len(k) >= 4 else 0
    print(cp.stdout, end="")
    print("\n".join(lines[start:]))
    sys.exit(cp.returncode)
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import FIRA, NOTO, load_product, noto_files  # noqa: E402
vg = load_product()
which = sys.argv[1] if len(sys.argv) > 1 else "many"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 30
m = vg.FontManager(True)
if which == "many":
    for i, p in enumerate([FIRA] + list(noto_files())):
        m.add_font_with_name(f"Font {i:02d}", [p])
else:
    m.add_font_with_name("Some Font", {"noto_regular": [NOTO], "fira": [FIRA], "noto_all": noto_files()}[which])
r = vg.Renderer.new_precise(0)
for i in range(runs):
    m.render_glyphs(None, r)
tm = m.timings()
print(f"{which}: {tm['glyphs']} glyphs, last run {tm['total_s'] * 1e6:.0f} us")
