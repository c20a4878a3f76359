"""A second, independent restatement of the raster (renderer_precise.rs:8-84, rtree_segments.rs:40-68,
segment.rs:54-72, point.rs:38-42) in numpy float64 — every operation a separate IEEE-rounded ufunc, no
FMA, same operand order — must give the oracle's bytes exactly, on real glyphs and on adversarial
polygons.  (The oracle itself is pinned by the reference's KATs; this guards the restatement against
a slip that those KATs do not reach, e.g. in the clamp / tie handling of the projection.)"""
import numpy as np
import pytest

from conftest import FIRA, NOTO


def numpy_sdf(segs, x0, y0, w, h):
    segs = np.asarray(segs, dtype=np.float64).reshape(-1, 4)
    vx, vy, wx, wy = (segs[:, i][None, None, :] for i in range(4))
    px = (np.arange(w, dtype=np.float64) + (np.float64(x0) + 0.5))[None, :, None]   # renderer_precise.rs:27,62
    py = (np.arange(h, dtype=np.float64) + (np.float64(y0) + 0.5))[:, None, None]   # :28,34
    out = np.zeros((h, w), dtype=np.uint8)
    if segs.shape[0] == 0:
        return out  # no candidate: +inf, outside -> 0 (rtree_segments.rs:57)
    with np.errstate(all="ignore"):
        dx, dy = wx - vx, wy - vy                                 # segment.rs:63
        l2 = dx * dx + dy * dy                                    # point.rs:38-42
        t = ((px - vx) * dx + (py - vy) * dy) / l2                # segment.rs:63-64
        qx, qy = vx + t * dx, vy + t * dy
        at_w = t > 1.0
        at_v = (l2 == 0.0) | (t < 0.0)
        qx = np.where(at_v, vx, np.where(at_w, wx, qx))
        qy = np.where(at_v, vy, np.where(at_w, wy, qy))
        ex, ey = qx - px, qy - py                                 # point.rs:39-40 (other - self)
        best = (ex * ex + ey * ey).min(axis=2)                    # rtree_segments.rs:57-62
        # winding: renderer_precise.rs:41-51, 58-66
        up = (vy <= py) & (wy > py)
        down = (vy > py) & (wy <= py)
        xc = vx + ((py - vy) / (wy - vy)) * (wx - vx)
        left = xc <= px
        wn = -((up & left).sum(axis=2).astype(np.int64) - (down & left).sum(axis=2).astype(np.int64))
        d = np.sqrt(best)
        d = np.where(wn != 0, -d, d)                              # :71-73
        d = d * 32.0 + 64.0                                       # :75
        n = np.clip(255.0 - d, 0.0, 255.0)                        # :76
        out = np.floor(n + 0.5).astype(np.uint8)                  # :79 round half away (n >= 0)
    return out[::-1]                                               # :78 top row first


def ring(points):
    p = np.asarray(points, dtype=np.float64)
    return np.concatenate([p, np.roll(p, -1, axis=0)], axis=1)


@pytest.mark.parametrize("path", [FIRA, NOTO], ids=["fira", "noto"])
def test_numpy_restatement_equals_oracle_on_glyphs(oracle, path):
    f = oracle.Font(path)
    cps = f.codepoints()
    cps = cps[cps <= 0xFFFF]
    rng = np.random.default_rng(3)
    pick = list(rng.choice(cps, 60, replace=False)) + [32, 65, 96, 230]  # + the glyphs of the reference's KATs
    n = 0
    for cp in pick:
        r = f.prepare_glyph(int(cp))
        if not r or not r[0].has_bitmap:
            continue
        i, segs = r
        want = np.asarray(oracle.sdf_render(segs, i.x0, i.y0, i.w, i.h, oracle.BRUTE)).reshape(i.h, i.w)
        got = numpy_sdf(segs, i.x0, i.y0, i.w, i.h)
        assert np.array_equal(got, want), f"U+{int(cp):04X}: {np.argwhere(got != want)[:4]}"
        n += 1
    assert n >= 50


def test_numpy_restatement_equals_oracle_on_adversarial_polygons(oracle):
    rng = np.random.default_rng(4)
    cases = []
    # overlapping rings (winding 2 / 0), degenerate segments, samples exactly on vertices and edges
    cases.append((np.concatenate([ring([(2, 2), (12, 2), (12, 12), (2, 12)]), ring([(6, 6), (16, 6), (16, 16), (6, 16)])]), -1, -1, 20, 20))
    cases.append((np.concatenate([ring([(2, 2), (12, 2), (12, 12), (2, 12)]), ring([(4, 10), (10, 10), (10, 4), (4, 4)])]), -1, -1, 16, 16))
    cases.append((np.concatenate([ring([(1.5, 1.5), (9.5, 1.5), (9.5, 9.5), (1.5, 9.5)]), ring([(5.5, 5.5), (5.5, 5.5), (5.5, 5.5)])]), -2, -2, 14, 14))
    for _ in range(12):
        k = int(rng.integers(3, 40))
        pts = rng.uniform(0, 24, (k, 2))
        if rng.random() < 0.5:
            pts = np.round(pts * 2) / 2  # vertices on half-integers = pixel centres
        cases.append((ring(pts), -3, -3, 30, 30))
    for segs, x0, y0, w, h in cases:
        want = np.asarray(oracle.sdf_render(segs, x0, y0, w, h, oracle.BRUTE)).reshape(h, w)
        assert np.array_equal(numpy_sdf(segs, x0, y0, w, h), want)
