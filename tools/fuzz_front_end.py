"""Differential fuzzing of the device front-end (outline commands -> flattened, closed, scaled
segments + rects) against the oracle's RingBuilder restatement: random command streams, including
ones ttf-parser never emits (curves on an empty ring, missing / repeated closes, cubic segments,
degenerate and repeated points, tiny and huge coordinates).  Exits non-zero at the first difference.

    python tools/fuzz_front_end.py seconds seed [curves|render]
`curves`: outline-like streams of mostly cubics and quadratics whose control points stay near the chord (the depths the
parallel flattening rounds take: 0..6), also with coordinates at the flatness threshold.
`render`: the same streams at 24 .. 200 px per EM, and the BITMAPS of the whole device path (front-end + raster, with the chunk
boxes the front-end derives from the commands' boxes: they decide which chunks a span skips) against the oracle's raster of
the oracle's segments — outlines of several hundred to several thousand segments, where those boxes matter."""
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
render = len(sys.argv) > 3 and sys.argv[3] == "render"
curvy = render or (len(sys.argv) > 3 and sys.argv[3] == "curves")
ctx = vg.SdfContext(0)
M, L, Q, C, Z = 0, 1, 2, 3, 4


def f32(v):
    return float(np.float32(v))


def curvy_stream():
    """closed contours of chained curves: end points on a rough circle, control points within a fraction of the chord"""
    cmds = []
    upem = float(rng.choice([256, 1000, 2048]))
    for _ in range(int(rng.integers(1, 5))):
        cx, cy = rng.uniform(0, upem, 2)
        rad = rng.uniform(0.02, 0.5) * upem
        k = int(rng.integers(3, 24))
        ang = np.sort(rng.uniform(0, 2 * np.pi, k))
        pts = [(f32(cx + rad * np.cos(a) * rng.uniform(0.7, 1.3)), f32(cy + rad * np.sin(a) * rng.uniform(0.7, 1.3))) for a in ang]
        cmds.append((M, 0, 0, 0, 0, pts[0][0], pts[0][1]))
        last = pts[0]
        for p in pts[1:] + [pts[0]]:
            kind = rng.choice([L, Q, C], p=[.15, .35, .5])
            ch = (p[0] - last[0], p[1] - last[1])
            bulge = rng.choice([0.0, 0.02, 0.2, 0.6]) * float(np.hypot(*ch))
            def ctl(t):
                return (f32(last[0] + t * ch[0] + rng.normal(0, 1) * bulge), f32(last[1] + t * ch[1] + rng.normal(0, 1) * bulge))
            if kind == L:
                cmds.append((L, 0, 0, 0, 0, p[0], p[1]))
            elif kind == Q:
                c1 = ctl(0.5)
                cmds.append((Q, c1[0], c1[1], 0, 0, p[0], p[1]))
            else:
                c1, c2 = ctl(0.33), ctl(0.67)
                cmds.append((C, c1[0], c1[1], c2[0], c2[1], p[0], p[1]))
            last = p
        cmds.append((Z, 0, 0, 0, 0, 0, 0))
    px = float(rng.choice([24.0, 24.0, 60.0, 200.0])) if render else 24.0
    return cmds, px / upem, float(rng.uniform(-0.5, 0.5))


def stream():
    if curvy:
        return curvy_stream()
    n = int(rng.choice([0, 1, 3, 8, 30, 120]))
    span = float(rng.choice([50, 1000, 2048, 16000]))
    snap = rng.choice([0, 0, 1, 4])
    cmds = []
    def pt():
        p = rng.uniform(-0.1 * span, span, 2)
        if snap:
            p = np.round(p / snap) * snap
        return f32(p[0]), f32(p[1])
    last = pt()
    for _ in range(n):
        k = rng.choice([M, L, L, L, Q, Q, Q, C, Z], p=[.08, .12, .12, .12, .14, .14, .14, .08, .06])
        if k == M or k == L:
            x, y = pt() if rng.random() > 0.1 else last
            cmds.append((int(k), 0, 0, 0, 0, x, y)); last = (x, y)
        elif k == Q:
            (x1, y1), (x, y) = pt(), pt()
            if rng.random() < 0.1:
                x1, y1 = last  # degenerate control point
            cmds.append((Q, x1, y1, 0, 0, x, y)); last = (x, y)
        elif k == C:
            (x1, y1), (x2, y2), (x, y) = pt(), pt(), pt()
            cmds.append((C, x1, y1, x2, y2, x, y)); last = (x, y)
        else:
            cmds.append((Z, 0, 0, 0, 0, 0, 0))
    return cmds, 24.0 / float(rng.choice([1000, 2048, 256, 16384])), float(rng.uniform(-0.5, 0.5))


t0 = time.time()
n_batches = n_streams = n_segs = n_px = n_big = 0
while time.time() - t0 < budget_s:
    sts = [stream() for _ in range(int(rng.integers(1, 24 if render else 60)))]
    cmds, cmd_off = [], [0]
    for st, _, _ in sts:
        cmds += [(c[1], c[2], c[3], c[4], c[5], c[6], c[0]) for c in st]
        cmd_off.append(len(cmds))
    scale = np.array([s for _, s, _ in sts]); shift = np.array([d for _, _, d in sts])
    rects, _, _ = ctx.outlines_prepare(np.array(cmd_off, np.uint32), np.array(cmds, dtype=vg.OUTLINE_CMD_DTYPE).reshape(-1), scale, shift)
    seg_off, segs = ctx.outlines_segments()
    bitmaps = ctx.outlines_render() if render else None
    gl = []
    for g, (st, sc, dx) in enumerate(sts):
        want = []
        for r in O.build_rings(st):
            p = r * sc
            p[:, 0] += dx
            p[:, 1] += 0.0
            want.append(np.concatenate([p[:-1], p[1:]], axis=1))
        want = np.concatenate(want) if want else np.zeros((0, 4))
        got = segs[seg_off[g]:seg_off[g + 1]]
        ok = got.tobytes() == want.tobytes() if len(want) else True
        if len(want):
            allp = np.concatenate([want[:, :2], want[:, 2:]])
            empty = allp[:, 0].max() <= allp[:, 0].min() and allp[:, 1].max() <= allp[:, 1].min()  # bbox.rs:56-58
            if empty:
                ok = int(rects[g]["has_raster"]) == 0
            else:
                ok = ok and int(rects[g]["has_raster"]) == 1 and int(rects[g]["x0"]) == int(np.floor(allp[:, 0].min())) - 3 \
                    and int(rects[g]["y0"]) == int(np.floor(allp[:, 1].min())) - 3 \
                    and int(rects[g]["w"]) == int(np.ceil(allp[:, 0].max())) + 3 - (int(np.floor(allp[:, 0].min())) - 3) \
                    and int(rects[g]["h"]) == int(np.ceil(allp[:, 1].max())) + 3 - (int(np.floor(allp[:, 1].min())) - 3)
        else:
            ok = int(rects[g]["has_raster"]) == 0 and len(got) == 0
        if not ok:
            print(f"MISMATCH seed {seed} batch {n_batches} stream {g}: {len(got)} vs {len(want)} segments, rect {rects[g]}\n{st}", flush=True)
            sys.exit(1)
        n_segs += len(want)
        if render and int(rects[g]["has_raster"]):
            gl.append((want, int(rects[g]["x0"]), int(rects[g]["y0"]), int(rects[g]["w"]), int(rects[g]["h"])))
            n_big += len(want) > 512
    if render and gl:
        ob = vg.make_batch(gl)
        ref, _ = O.sdf_render_batch(ob, O.BRUTE, 16)
        if not np.array_equal(ref, bitmaps[:len(ref)]) or len(bitmaps) != len(ref):
            bad = np.flatnonzero(ref != bitmaps[:len(ref)])
            print(f"BITMAP MISMATCH seed {seed} batch {n_batches}: {bad.size} bytes differ (first at {bad[:1]}), {len(bitmaps)} vs {len(ref)} bytes", flush=True)
            np.savez(ROOT / "gpurun_out" / f"fuzz_render_fail_{seed}_{n_batches}.npz", cmds=np.array(cmds, dtype=vg.OUTLINE_CMD_DTYPE).reshape(-1),
                     cmd_off=np.array(cmd_off, np.uint32), scale=scale, shift=shift)
            sys.exit(1)
        n_px += len(ref)
    n_batches += 1; n_streams += len(sts)
    if n_batches % 50 == 0:
        print(f"[{time.time() - t0:6.0f} s] {n_batches} batches, {n_streams} streams, {n_segs / 1e6:.2f} M segments: all equal", flush=True)
print(f"done: {n_batches} batches, {n_streams} command streams, {n_segs / 1e6:.2f} M segments, 0 differences (seed {seed})" +
      (f"; bitmaps of the whole device path: {n_px / 1e6:.1f} Mpx, {n_big} outlines of more than 512 segments, 0 differing bytes" if render else ""), flush=True)
