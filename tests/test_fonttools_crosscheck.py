"""Independent cross-check of the TrueType reading the oracle restates from ttf-parser 0.25.1 (absent
here): fontTools — a third implementation that shares no code with the oracle's C reader
or the product's C++ reader — must see the same code points, glyph ids, advances and, per glyph, the
same set of outline primitives (lines and quadratic segments with their control points), including
composite glyphs with offsets and 2x2 transforms.  Primitives are compared as multisets per glyph:
where a contour starts is a convention (the reference's bits depend on it; those are pinned by the
reference's own KATs in test_oracle_kat.py), the geometry is not."""
from collections import Counter

import numpy as np
import pytest

from conftest import FIRA, NOTO, noto_files

ft = pytest.importorskip("fontTools.ttLib")


def fonts_under_test():
    return sorted(set(str(p) for p in [FIRA, NOTO] + list(noto_files())))  # all 21 fixture fonts


def contour_primitives(pts, on):
    """TrueType quadratic contour -> [(p0, ctrl or None, p1)], implied on-curve midpoints included"""
    n = len(pts)
    if n < 2:
        return []
    pts = [np.asarray(p, dtype=np.float64) for p in pts]
    # expand implied midpoints between consecutive off-curve points
    ex = []
    for i in range(n):
        a, b = i, (i + 1) % n
        ex.append((pts[a], bool(on[a])))
        if not on[a] and not on[b]:
            ex.append(((pts[a] + pts[b]) / 2.0, True))
    m = len(ex)
    first_on = next((i for i, (_, o) in enumerate(ex) if o), None)
    if first_on is None:
        return []
    out = []
    i = first_on
    for _ in range(m):
        p0 = ex[i % m][0]
        nxt = ex[(i + 1) % m]
        if nxt[1]:
            out.append((p0, None, nxt[0]))
            i += 1
        else:
            out.append((p0, nxt[0], ex[(i + 2) % m][0]))
            i += 2
        if i - first_on >= m:
            break
    return out


def key(prim, nd):
    p0, c, p1 = prim
    r = lambda v: tuple(np.round(np.asarray(v, dtype=np.float64), nd) + 0.0)  # noqa: E731  (+0.0: no -0)
    return (r(p0), None if c is None else r(c), r(p1))


def degenerate(prim):
    p0, c, p1 = prim
    return c is None and np.array_equal(np.asarray(p0), np.asarray(p1))


@pytest.mark.parametrize("path", fonts_under_test(), ids=lambda p: p.split("/")[-1][:-4])
def test_same_cmap_advances_and_outline_primitives(oracle, path):
    font = ft.TTFont(path, lazy=False)
    f = oracle.Font(path)
    glyf, hmtx, order = font["glyf"], font["hmtx"], font.getGlyphOrder()
    # ---- code points: union of the Unicode subtables, entries that map to a glyph other than .notdef ----
    want_cps = {}
    for t in font["cmap"].tables:
        if t.isUnicode():
            for cp, name in t.cmap.items():
                if font.getGlyphID(name) != 0:
                    want_cps.setdefault(cp, name)
    got = [int(c) for c in f.codepoints()]
    missing = sorted(set(want_cps) - set(got))
    extra = sorted(set(got) - set(want_cps))
    # a code point mapped to glyph 0 is "present" for ttf-parser (glyph_index is Some(0)); fontTools lists it too
    extra = [c for c in extra if f.glyph_index(c) != 0]
    assert not missing and not extra, (missing[:5], extra[:5])
    assert f.units_per_em == font["head"].unitsPerEm

    n_transformed = n_composite = n_checked = 0
    for cp in got:
        gid = f.glyph_index(cp)
        if gid is None or gid == 0 and cp not in want_cps:
            continue
        name = order[gid]
        if cp in want_cps:
            assert font.getGlyphID(want_cps[cp]) == gid, (cp, gid, want_cps[cp])
        assert f.hor_advance(gid) == hmtx[name][0], (cp, name)
        g = glyf[name]
        coords, ends, flags = g.getCoordinates(glyf)
        transformed = g.isComposite() and any(hasattr(c, "transform") for c in g.components)
        n_composite += g.isComposite()
        n_transformed += transformed
        want = []
        start = 0
        for e in ends:
            pts = [coords[i] for i in range(start, e + 1)]
            on = [flags[i] & 1 for i in range(start, e + 1)]
            want += contour_primitives(pts, on)
            start = e + 1
        # oracle: the command stream ttf-parser would emit
        have = []
        cur = first = None
        for kind, x1, y1, x2, y2, x, y in f.outline(gid):
            p = np.array([x, y], dtype=np.float64)
            if kind == 0:
                cur = first = p
            elif kind == 1:
                have.append((cur, None, p))
                cur = p
            elif kind == 2:
                have.append((cur, np.array([x1, y1], dtype=np.float64), p))
                cur = p
            elif kind == 3:
                pytest.fail("cubic in a glyf font")
            else:
                cur = first
        nd = 2 if transformed else 6  # f32 (ttf-parser) vs f64 (fontTools) affine arithmetic
        a = Counter(key(p, nd) for p in want if not degenerate(p))
        b = Counter(key(p, nd) for p in have if not degenerate(p))
        if a != b:
            only_a, only_b = list((a - b).items())[:3], list((b - a).items())[:3]
            pytest.fail(f"U+{cp:04X} gid {gid} ({name}): fontTools-only {only_a} oracle-only {only_b}")
        n_checked += 1
    assert n_checked > 100
    print(f"{path.split('/')[-1]}: {n_checked} glyphs, {n_composite} composite, {n_transformed} with a 2x2 transform")
