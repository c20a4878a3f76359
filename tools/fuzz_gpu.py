"""Differential fuzzing of the default kernel against the oracle (development aid / evidence):
random batches of polygons over a wide range of sizes, segment counts, offsets, snapped coordinates
(ties, samples on vertices and edges), degenerate and duplicated points.  Prints progress every few
batches; exits non-zero at the first differing byte.

    python tools/fuzz_gpu.py seconds seed [boundary|guards]
`boundary`: outlines made of long straight edges on multiples of 1/64 px, axis-parallel or nearly so, so that whole rows and
columns of pixels sit at distances (2 m + 1) / 64 px — exactly on the rounding boundaries of the byte (32 d + 1/2 an integer,
renderer_precise.rs:75-79), where the f32 filter must hand over to the exact evaluation.
`guards`: outlines whose far vertices put the coordinate bound M of a chunk just below / above 4096 px (group bounds switch off)
and 10^6 px (the filter switches off), seen through a small window on the near part of the outline."""
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
mode = sys.argv[3] if len(sys.argv) > 3 else ""
ctx = vg.SdfContext(0)


def ring(points):
    p = np.asarray(points, dtype=np.float64)
    return np.concatenate([p, np.roll(p, -1, axis=0)], axis=1)


def boundary_glyph():
    """rectilinear / nearly rectilinear polygons on the 1/64 px grid"""
    size = int(rng.choice([12, 24, 40, 80]))
    segs = []
    for k in range(int(rng.integers(1, 4))):
        x0, y0 = rng.integers(0, size * 64 // 2, 2)
        w, h = rng.integers(64, size * 64 // 2 + 65, 2)
        # odd multiples of 1/64 put pixel centres (k + 1/2) at distances (2 m + 1) / 64 from the edge: byte boundaries
        x0, y0, w, h = (int(v) | 1 for v in (x0, y0, w, h))
        pts = np.array([(x0, y0), (x0 + w, y0), (x0 + w, y0 + h), (x0, y0 + h)], dtype=np.float64) / 64.0
        if rng.random() < 0.5:  # steps along one side: many collinear segments, ties between neighbours
            n = int(rng.integers(2, 200))
            xs = np.linspace(pts[0, 0], pts[1, 0], n + 1)[1:-1]
            xs = np.round(xs * 64) / 64
            pts = np.concatenate([pts[:1], np.stack([xs, np.full_like(xs, pts[0, 1])], 1), pts[1:]])
        if rng.random() < 0.3:  # a slight tilt: distances drift across the boundary along the edge
            pts[:, 1] += (pts[:, 0] - pts[0, 0]) * float(rng.choice([1, 2, 3])) / 4096.0
        if k % 2:
            pts = pts[::-1]
        segs.append(ring(pts))
    segs = np.concatenate(segs) + float(rng.choice([0.0, 0.0, 17.0, -300.0]))
    lo = np.floor(segs[:, [0, 1]].min(0)).astype(np.int64) - 3
    hi = np.ceil(segs[:, [0, 1]].max(0)).astype(np.int64) + 3
    return segs, int(lo[0]), int(lo[1]), int(hi[0] - lo[0]), int(hi[1] - lo[1])


def guard_glyph():
    """a window of ~40 px on the near corner of an outline whose far vertices lie ~4096 px or ~10^6 px away"""
    far = float(rng.choice([4096.0, 4096.0, 1.0e6])) * float(rng.choice([0.97, 0.995, 0.9995, 1.0, 1.0005, 1.005, 1.03]))
    n_near = int(rng.choice([3, 8, 40, 300, 1200]))
    a = np.sort(rng.uniform(0, 0.5 * np.pi, n_near))
    r = rng.uniform(8, 30) * (1 + 0.3 * rng.uniform(-1, 1, n_near))
    near = np.stack([20 + r * np.cos(a), 20 + r * np.sin(a)], 1)
    # the far part: a few vertices out at `far` (relative to the window's middle, the filter's origin), on either axis or both
    k = int(rng.integers(1, 4))
    sgn = rng.choice([-1.0, 1.0], 2)
    farp = np.stack([20 + sgn[0] * far * rng.uniform(0.2, 1.0, k), 20 + sgn[1] * far * rng.uniform(0.2, 1.0, k)], 1)
    farp[int(rng.integers(0, k)), int(rng.integers(0, 2))] = 20 + float(rng.choice([-1.0, 1.0])) * far   # one coordinate AT the bound
    pts = np.concatenate([near, farp])
    if rng.random() < 0.5:
        pts = np.round(pts * 64) / 64
    segs = ring(pts)
    if rng.random() < 0.5:  # the far vertices in a chunk of their own: duplicate the near part up to a chunk boundary
        pad = ring(near[::-1] * 0.5 + 10)
        segs = np.concatenate([pad, segs])
    w = int(rng.integers(30, 60))
    return segs, 0, 0, w, w


def glyph():
    if mode == "boundary":
        return boundary_glyph()
    if mode == "guards":
        return guard_glyph()
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
print(f"done: {n_batches} batches, {n_glyphs} glyphs, {n_px / 1e6:.1f} Mpx, 0 differing bytes (seed {seed}{', ' + mode if mode else ''})", flush=True)
