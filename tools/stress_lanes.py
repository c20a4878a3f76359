"""Repeated multi-lane renders (one process, N device lanes on the box's GPU) compared with the first one: races between the
lanes' host threads, the pools, the early read-back and the in-place assembly would show as differing bytes or a hang
(development aid).   python tools/stress_lanes.py [lanes] [iterations]"""
import hashlib
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import FIRA, load_product, noto_files  # noqa: E402
vg = load_product()
lanes_n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
m = vg.FontManager(True)
m.add_font_with_name("Fira Sans Regular", [FIRA])
m.add_font_with_name("Noto Sans Regular", noto_files())
single = vg.Renderer.new_precise(0)
w0 = vg.DummyWriter()
m.render_glyphs(w0, single)
want = hashlib.sha256(b"".join(w0.files[k] for k in sorted(w0.files))).hexdigest()
lanes = vg.Renderer.new_multi([0] * lanes_n)
t0 = time.perf_counter()
for i in range(iters):
    r = lanes if i % 3 else single
    m.set_in_place_pbf(i % 5 != 4)
    m.set_glyf_on_device(i % 7 != 6)   # (mostly the device's glyf decoder, now and then the host's reader)
    w = vg.DummyWriter()
    m.render_glyphs(w, r)
    got = hashlib.sha256(b"".join(w.files[k] for k in sorted(w.files))).hexdigest()
    assert got == want, f"iteration {i}: output differs"
    if r is lanes:
        assert m.reduced_counters()[1] == m.timings()["glyphs"]
print(f"{iters} renders ({lanes_n} lanes / single device alternating, in-place assembly and device glyf decoder on and off): all equal, {time.perf_counter() - t0:.1f} s")
