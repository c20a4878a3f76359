#!/usr/bin/env python3
"""Sustained end-to-end run (fonts -> PBF bytes, NULL sink) for several seconds: glyph rate per window and what the
container's CPU controller did meanwhile (cpu.stat: nr_throttled / throttled_usec).  bench.py's e2e legs last a few
milliseconds — shorter than one period of the CPU quota — so they cannot show whether a pool of 2 x quota threads that
spin between fork/joins runs into the quota when the run is long.

usage: sustained_e2e.py [seconds=3] [workload=21fonts|noto_regular|noto_all] [lanes=1]   (VG_THREADS / VG_POOL_SPIN_US vary the pool)
"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from conftest import load_product  # noqa: E402


def cpu_stat():
    try:
        return {k: int(v) for k, v in (l.split() for l in Path("/sys/fs/cgroup/cpu.stat").read_text().splitlines())}
    except OSError:
        return {}


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
    workload = sys.argv[2] if len(sys.argv) > 2 else "21fonts"
    vg = load_product()
    td = ROOT / "testdata"
    mgr = vg.FontManager(True)
    if workload == "21fonts":
        for i, p in enumerate([td / "Fira Sans - Regular.ttf"] + sorted((td / "Noto Sans").glob("*.ttf"), key=lambda q: q.name)):
            mgr.add_font_with_name(f"Font {i:02d}", [p])
    elif workload == "noto_regular":
        mgr.add_font_with_name("Noto Sans Regular", [td / "Noto Sans" / "Noto Sans - Regular.ttf"])
    else:
        mgr.add_font_with_name("Noto Sans", sorted((td / "Noto Sans").glob("*.ttf"), key=lambda q: q.name))
    lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 1   # > 1: that many device lanes of the library on the one GPU
    r = vg.Renderer.new_precise(0) if lanes == 1 else vg.Renderer.new_multi([0] * lanes)
    for _ in range(3):
        mgr.render_glyphs(None, r)
    glyphs = mgr.timings()["glyphs"]
    def thread_times():
        out = {}
        for t in Path("/proc/self/task").iterdir():
            try:
                f = (t / "stat").read_text()
                name = f[f.index("(") + 1:f.rindex(")")]
                rest = f[f.rindex(")") + 2:].split()
                out[int(t.name)] = (name, (int(rest[11]) + int(rest[12])) / 100.0)   # utime + stime, clock ticks of 10 ms
            except OSError:
                pass
        return out
    th0 = thread_times()
    c0 = cpu_stat()
    t0 = time.perf_counter()
    window, n_win, t_win, rates, runs = 0.25, 0, t0, [], 0
    best = None
    while True:
        t1 = time.perf_counter()
        mgr.render_glyphs(None, r)
        t2 = time.perf_counter()
        best = t2 - t1 if best is None else min(best, t2 - t1)
        runs += 1
        n_win += 1
        if t2 - t_win >= window:
            rates.append(n_win * glyphs / (t2 - t_win))
            n_win, t_win = 0, t2
        if t2 - t0 >= seconds:
            break
    wall = time.perf_counter() - t0
    c1 = cpu_stat()
    d = {k: c1.get(k, 0) - c0.get(k, 0) for k in ("usage_usec", "nr_periods", "nr_throttled", "throttled_usec")}
    print(f"{workload}: {glyphs} glyphs per run, {runs} runs in {wall:.2f} s -> {runs * glyphs / wall / 1e6:.2f} M glyphs/s sustained "
          f"(best single run {glyphs / best / 1e6:.2f} M); per 0.25 s window: "
          + " ".join(f"{x / 1e6:.2f}" for x in rates))
    print(f"  cpu.stat: {d.get('usage_usec', 0) / 1e6 / wall:.1f} CPUs used on average, {d.get('nr_throttled', 0)} of {d.get('nr_periods', 0)} periods throttled, "
          f"{d.get('throttled_usec', 0) / 1e3:.0f} ms throttled (summed over threads)")

    tm = mgr.timings()
    print("  phases of the last run (ms): " + " ".join(f"{k[:-2]}={v * 1e3:.3f}" for k, v in tm.items() if k.endswith("_s")))
    th1 = thread_times()
    by = {}
    for tid, (name, secs) in th1.items():
        key = "vg-pool" if name.startswith("vg-pool") else name
        by[key] = by.get(key, 0.0) + secs - th0.get(tid, (name, 0.0))[1]
    print("  CPU seconds by thread name: " + ", ".join(f"{k} {v:.2f}" for k, v in sorted(by.items(), key=lambda kv: -kv[1]) if v >= 0.01))


if __name__ == "__main__":
    main()
