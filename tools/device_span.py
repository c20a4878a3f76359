"""Device span of one submission (HIP events around upload ... last kernel, VGSDF_TRACE) and end-to-end time of the same run,
config 2 (Noto Sans Regular) or another workload, over N warm runs: median / min of both.  Development aid.
    python tools/device_span.py [noto_regular|noto_all|fira|many] [runs]"""
import os
import re
import subprocess
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
if os.environ.get("_VG_SPAN_CHILD") != "1":
    env = dict(os.environ, VGSDF_TRACE="1", _VG_SPAN_CHILD="1")
    cp = subprocess.run([sys.executable, __file__] + sys.argv[1:], env=env, capture_output=True, text=True)
    spans = [float(m) for m in re.findall(r"device span of the submission.*?: ([0-9.]+) us", cp.stderr)]
    warm = spans[len(spans) // 3:]
    warm.sort()
    sys.stdout.write(cp.stdout)
    if warm:
        print(f"device span over {len(warm)} warm submissions: min {warm[0]:.1f} us, median {warm[len(warm) // 2]:.1f} us, p90 {warm[int(len(warm) * 0.9)]:.1f} us")
    else:
        print(cp.stderr[-2000:])
    sys.exit(cp.returncode)
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import FIRA, NOTO, load_product, noto_files  # noqa: E402
vg = load_product()
which = sys.argv[1] if len(sys.argv) > 1 else "noto_regular"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 60
m = vg.FontManager(True)
if which == "many":
    for i, p in enumerate([FIRA] + list(noto_files())):
        m.add_font_with_name(f"Font {i:02d}", [p])
else:
    m.add_font_with_name("Some Font", {"noto_regular": [NOTO], "fira": [FIRA], "noto_all": noto_files()}[which])
r = vg.Renderer.new_precise(0)
ts = []
for i in range(runs):
    t0 = time.perf_counter()
    m.render_glyphs(None, r)
    ts.append(time.perf_counter() - t0)
tm = m.timings()
w = sorted(ts[runs // 3:])
print(f"{which}: {tm['glyphs']} glyphs, end to end min {w[0] * 1e6:.0f} us ({tm['glyphs'] / w[0] / 1e6:.2f} M glyphs/s), median {w[len(w) // 2] * 1e6:.0f} us; "
      f"phases of the last run: " + " ".join(f"{k[:-2]}={v * 1e6:.0f}" for k, v in tm.items() if k.endswith("_s")))
