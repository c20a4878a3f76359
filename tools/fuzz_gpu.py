"""Differential fuzzing of the default kernel against the oracle (development aid / evidence):
random batches of polygons over a wide range of sizes, segment counts, offsets, snapped coordinates
(ties, samples on vertices and edges), degenerate and duplicated points.  Prints progress every few
batches; exits non-zero at the first differing byte."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import load_product  # noqa: E402
from oracle import oracle as O  # noqa: E402

vg = load_product()
budget_s = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = vg.SdfContext(0)


def ring(points):
    p = np.asarray(points, dtype=np.float64)
    return np.concatenate([p, np.roll(p, -1, axis=0)], axis=1)


def glyph():
    size = float(rng.choice([6, 12, 24, 40, 80, 160, 300]))
    n_rings = int(rng.integers(1, 6))
    snap = rng.choice([0, 0, 0, 2, 4, 64])          # 0: none; k: coordinates on multiples of 1/k
    off = rng.choice([0.0, 0.0, 0.0, 1.0e3, -7.0e4, 3.0e6]) * rng.choice([1.0, 1.0, 0.5])
    segs = []
    for k in range(n_rings):
        n_pts = int(rng.choice([3, 4, 7, 20, 60, 200, 700, 1500]))
        c = rng.uniform(size * 0.2, size * 0.8, 2)
        r0 = rng.uniform(size * 0.05, size * 0.4)
        a = np.sort(rng.uniform(0, 2 * np.pi, n_pts))
        r = r0 * (1 + rng.choice([0.0, 0.05, 0.35, 0.9]) * rng.uniform(-1, 1, n_pts))
        pts = np.stack([c[0] + r * np.cos(a), c[1] + r * np.sin(a)], 1)
        if rng.random() < 0.3:  # duplicated / zero-length runs
            idx = rng.integers(0, n_pts, max(1, n_pts // 10))
            pts = np.insert(pts, idx, pts[idx], axis=0)
        if snap:
            pts = np.round(pts * snap) / snap
        if k % 2:
            pts = pts[::-1]
        segs.append(ring(pts))
    segs = np.concatenate(segs) + off
    lo = np.floor(segs[:, [0, 1]].min(0)).astype(np.int64) - 3
    hi = np.ceil(segs[:, [0, 1]].max(0)).astype(np.int64) + 3
    if rng.random() < 0.15:  # rect not matching the outline: clipped or shifted window
        lo += rng.integers(-4, 5, 2)
        hi += rng.integers(-4, 5, 2)
        hi = np.maximum(hi, lo + 1)
    return segs, int(lo[0]), int(lo[1]), int(hi[0] - lo[0]), int(hi[1] - lo[1])


t0 = time.time()
n_batches = n_glyphs = n_px = 0
while time.time() - t0 < budget_s:
    gl = [glyph() for _ in range(int(rng.integers(1, 40)))]
    batch = vg.make_batch(gl)
    want, _ = O.sdf_render_batch(batch, O.BRUTE, 16)
    got = ctx.render_batch(batch)
    diff = np.flatnonzero(got != want)
    if diff.size:
        g = int(np.searchsorted(batch.out_off, diff[0], side="right") - 1)
        np.savez(ROOT / "gpurun_out" / f"fuzz_fail_{seed}_{n_batches}.npz", segs=gl[g][0], rect=np.array(gl[g][1:]))
        print(f"MISMATCH seed {seed} batch {n_batches}: {diff.size} bytes, first in glyph {g} rect {gl[g][1:]} "
              f"(got {got[diff[0]]} want {want[diff[0]]})", flush=True)
        sys.exit(1)
    n_batches += 1; n_glyphs += len(gl); n_px += int(batch.out_off[-1])
    if n_batches % 20 == 0:
        print(f"[{time.time() - t0:6.0f} s] {n_batches} batches, {n_glyphs} glyphs, {n_px / 1e6:.1f} Mpx: all equal", flush=True)
print(f"done: {n_batches} batches, {n_glyphs} glyphs, {n_px / 1e6:.1f} Mpx, 0 differing bytes (seed {seed})", flush=True)
