"""The device's glyf decoder on shapes the fixture fonts hardly hold: contours that begin off the curve, contours without
any on-curve point, single- and two-point contours, flag runs longer than a wave, end points that do not ascend, a glyph
with more points than the decoder takes (the batch falls back to the host's reader), composites with 2 x 2 transforms.
Fonts are built here with fontTools (raw `glyf` entries); the yardstick is the host's reader of the same bytes
(csrc/host/ttf_face.cpp — ttf-parser's rules, pinned by the oracle on the fixture fonts): same segments, rects and
bitmaps, bit for bit, and the same PBF files through FontManager.
"""
import io
from array import array

import numpy as np
import pytest

fontTools = pytest.importorskip("fontTools")
from fontTools.fontBuilder import FontBuilder  # noqa: E402
from fontTools.ttLib.tables import ttProgram  # noqa: E402
from fontTools.ttLib.tables._g_l_y_f import Glyph, GlyphComponent, GlyphCoordinates  # noqa: E402

pytestmark = pytest.mark.gpu


def _simple(contours):
    """contours: lists of (x, y, on_curve)"""
    g = Glyph()
    g.numberOfContours = len(contours)
    pts, flags, ends = [], [], []
    for c in contours:
        for x, y, on in c:
            pts.append((x, y))
            flags.append(1 if on else 0)
        ends.append(len(pts) - 1)
    g.coordinates = GlyphCoordinates(pts)
    g.flags = array("B", flags)
    g.endPtsOfContours = ends
    g.program = ttProgram.Program()
    g.program.fromBytecode(b"")
    return g


def _composite(parts):
    """parts: (glyph name, dx, dy, 2x2 or None)"""
    g = Glyph()
    g.numberOfContours = -1
    g.components = []
    for name, dx, dy, xf in parts:
        c = GlyphComponent()
        c.glyphName = name
        c.x, c.y = dx, dy
        c.flags = 0x0002 | 0x0001   # ARGS_ARE_XY_VALUES | ARG_1_AND_2_ARE_WORDS
        if xf is not None:
            c.transform = [[xf[0], xf[1]], [xf[2], xf[3]]]
        g.components.append(c)
    return g


def _circle(n, r, cx, cy, on_every):
    import math
    return [(int(cx + r * math.cos(2 * math.pi * i / n)), int(cy + r * math.sin(2 * math.pi * i / n)), i % on_every == 0 if on_every else False)
            for i in range(n)]


def _font(glyphs, cmap):
    order = list(glyphs)
    fb = FontBuilder(1000, isTTF=True)
    fb.setupGlyphOrder(order)
    fb.setupCharacterMap(cmap)
    fb.setupGlyf(glyphs)
    fb.setupHorizontalMetrics({g: (700, 0) for g in order})
    fb.setupHorizontalHeader(ascent=935, descent=-265)
    fb.setupNameTable({"familyName": "Synth Shapes", "styleName": "Regular"})
    fb.setupOS2()
    fb.setupPost()
    buf = io.BytesIO()
    fb.save(buf)
    return buf.getvalue()


def _shapes(huge_points=0):
    square = [(100, 100, 1), (600, 100, 1), (600, 600, 1), (100, 600, 1)]
    glyphs = {
        ".notdef": _simple([square]),
        "off_start": _simple([[(100, 0, 0), (300, 0, 1), (300, 400, 0), (100, 400, 1)]]),
        "two_off_start": _simple([[(100, 0, 0), (400, 0, 0), (400, 400, 1), (100, 400, 0)]]),
        "all_off": _simple([_circle(8, 250, 350, 350, 0)]),
        "tiny_contours": _simple([[(50, 50, 1)], [(60, 60, 0)], [(100, 100, 1), (300, 300, 1)], [(100, 500, 0), (300, 500, 0)], square]),
        # 300 points with the same flags and deltas: repeat counts of 255 + a rest, crossing the 64-byte windows of the decoder
        "long_runs": _simple([[(100 + i, 100, 1) for i in range(300)] + [(400, 600, 1), (100, 600, 1)],
                              [(150 + 2 * i, 200 + (i % 2), i % 2 == 0) for i in range(130)] + [(420, 500, 1), (150, 500, 1)]]),
        "many_contours": _simple([[(20 * k, 20 * (k % 7), 1), (20 * k + 15, 20 * (k % 7), k % 3 != 0), (20 * k + 15, 20 * (k % 7) + 15, 1),
                                   (20 * k, 20 * (k % 7) + 15, k % 2 == 0)] for k in range(100)]),
        "mixed": _simple([_circle(90, 300, 350, 350, 3), _circle(40, 120, 350, 350, 2)]),
    }
    glyphs["composite"] = _composite([("off_start", 0, 0, None), ("all_off", 120, -80, (0.5, 0.25, -0.25, 0.75)), ("mixed", -30, 40, (1.0, 0.0, 0.0, -1.0))])
    glyphs["nested"] = _composite([("composite", 10, 20, (0.75, 0.0, 0.0, 0.75)), ("tiny_contours", 0, 300, None)])
    if huge_points:
        glyphs["huge"] = _simple([[(100 + (i % 700), 100 + 5 * (i // 700) + (i % 2), i % 4 != 1) for i in range(huge_points)]])
    names = [n for n in glyphs if n != ".notdef"]
    return _font(glyphs, {0x41 + i: n for i, n in enumerate(names)}), names


def _compare(vg, font):
    mgr = vg.FontManager(False)
    fid = mgr.add_font_data("Shapes", font)
    o, g = mgr.record_outlines(fid), mgr.record_glyf_parts(fid)
    assert list(o["ids"]) == list(g["ids"])
    ctx = vg.SdfContext(0)
    try:
        rects_h, out_bytes_h, n_seg_h = ctx.outlines_prepare(o["cmd_off"], o["cmds"], o["scale"], o["shift_x"])
        bitmaps_h = ctx.outlines_render()
        seg_off_h, segs_h = ctx.outlines_segments()
        ctx.outlines_submit_glyf(g["cmd_off"], g["parts"], g["bytes"], g["scale"], g["shift_x"], capacity=int(out_bytes_h) + 64)
        rects_d, bitmaps_d, out_bytes_d, n_seg_d = ctx.outlines_wait()
        seg_off_d, segs_d = ctx.outlines_segments()
    finally:
        ctx.close()
    assert np.array_equal(rects_d, rects_h) and np.array_equal(seg_off_d, seg_off_h)
    assert segs_d.tobytes() == segs_h.tobytes() and n_seg_d == n_seg_h
    assert bitmaps_d is not None and np.array_equal(bitmaps_d, bitmaps_h)
    return o, g, rects_h


def test_odd_contours_long_runs_and_composites(vg):
    font, names = _shapes()
    o, g, rects = _compare(vg, font)
    assert len(o["ids"]) == len(names) and int(rects["has_raster"].sum()) >= len(names) - 1
    by_name = dict(zip(names, range(len(names))))
    n_cmds = np.diff(o["cmd_off"])
    assert n_cmds[by_name["all_off"]] == 10           # move + 8 implied-midpoint quads + close
    assert n_cmds[by_name["long_runs"]] > 350
    assert (g["parts"]["plain"] == 0).sum() >= 4       # the components of the two composites


def test_end_points_that_do_not_ascend(vg):
    """ttf-parser's EndpointsIter: a contour whose end point is not behind its predecessor's still takes one point, and
    points behind the last contour are contours of their own — bytes patched by hand, both readers agree"""
    font, names = _shapes()
    b = bytearray(font)
    n_tables = int.from_bytes(b[4:6], "big")
    for i in range(n_tables):
        rec = 12 + 16 * i
        if bytes(b[rec:rec + 4]) == b"glyf":
            off = int.from_bytes(b[rec + 8:rec + 12], "big")
    # .notdef is the first entry: one contour, end point 3 -> make "tiny_contours" (5 contours) non-ascending instead
    from fontTools.ttLib import TTFont
    f = TTFont(io.BytesIO(font))
    glyf, loca = f["glyf"], f["loca"]
    gid = f.getGlyphID("tiny_contours")
    at = off + loca[gid] + 10                       # endPtsOfContours of the entry: 0, 1, 3, 5, 9
    assert [int.from_bytes(b[at + 2 * k:at + 2 * k + 2], "big") for k in range(5)] == [0, 1, 3, 5, 9]
    b[at + 4:at + 6] = (1).to_bytes(2, "big")        # 0, 1, 1, 5, 9: the third contour does not ascend
    o, g, _ = _compare(vg, bytes(b))
    b[at + 4:at + 6] = (3).to_bytes(2, "big")
    b[at + 6:at + 8] = (2).to_bytes(2, "big")        # 0, 1, 3, 2, 9: the fourth goes backwards
    _compare(vg, bytes(b))


def test_a_glyph_beyond_the_decoders_limits_falls_back_to_the_host_reader(vg):
    font, names = _shapes(huge_points=7000)
    r = vg.Renderer.new_precise(0)
    files = {}
    for on in (True, False):
        mgr = vg.FontManager(True)
        mgr.set_glyf_on_device(on)
        mgr.add_font_data("Shapes", font)
        w = vg.DummyWriter()
        mgr.render_glyphs(w, r)
        files[on] = w.files
        t = mgr.timings()
        assert (t["glyf_groups"], t["glyf_fallbacks"]) == ((1, 1) if on else (0, 0))
        # the manager remembers that the device refused this font: the next run records it with the host's reader at once
        w2 = vg.DummyWriter()
        mgr.render_glyphs(w2, r)
        t = mgr.timings()
        assert w2.files == w.files and (t["glyf_groups"], t["glyf_fallbacks"]) == (0, 0)
    assert files[True] == files[False] and len(files[True]) == 256


def test_parts_past_the_batch_bounds_fall_back_to_the_host_reader():
    """Composite fan-out (tests/test_composite_fanout.py) sends a group to the host's reader BEFORE anything is uploaded.  With
    the byte bound turned down to 4 KB (VG_GLYF_PARTS_LIMIT, read once per process: a child) every group of Fira Sans takes
    that path: the files still carry the golden SHA-256s, one fallback is counted, the second run skips the glyf form."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    code = r'''
import hashlib, json, sys
sys.path.insert(0, "tests")
from conftest import load_product, FIRA
vg = load_product()
golden = json.load(open("tests/golden/pbf_sha256.json"))["fira"]
m = vg.FontManager(True)
fid = m.add_font_with_name("Fira Sans Regular", [FIRA])
r = vg.Renderer.new_precise(0)
out = []
for run in range(2):
    w = vg.DummyWriter()
    m.render_glyphs(w, r)
    bad = [s for s, h in golden.items() if hashlib.sha256(w.files[f"{fid}/{s}-{int(s) + 255}.pbf"]).hexdigest() != h]
    t = m.timings()
    out.append([len(bad), t["glyf_groups"], t["glyf_fallbacks"]])
print(json.dumps(out))
'''
    env = dict(os.environ, VG_GLYF_PARTS_LIMIT="4096")
    cp = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert cp.returncode == 0, cp.stderr[-2000:]
    assert json.loads(cp.stdout.strip().splitlines()[-1]) == [[0, 0, 1], [0, 0, 0]]
