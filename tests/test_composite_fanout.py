"""Composite fan-out (round-3 advice): the glyf PARTS of a glyph copy every simple glyph once per LEAF of the composite
tree, so a font of a few hundred KB can ask for gigabytes — one 1500-point glyph under 300 x 300 nested components is
90 000 leaves.  The recorder's offsets are 32 bits wide; before the fix the sums wrapped and the merge copied `b1 - b0`
of wrapped values (a heap overflow on an untrusted font).  Now the recorder stops at 2^26 bytes / slots per batch and
says so, and FontManager drops the glyf form for the group (GPU half: tests/test_gpu_glyf_shapes.py).
ttf-parser has no such bound (it streams callbacks, renderer.rs:110): a font like this is a denial of service there too;
nothing in the reference's fixtures comes near (the largest glyph of testdata has 6 components)."""
import io
import resource
from array import array

import pytest

fontTools = pytest.importorskip("fontTools")
from fontTools.fontBuilder import FontBuilder  # noqa: E402
from fontTools.ttLib.tables import ttProgram  # noqa: E402
from fontTools.ttLib.tables._g_l_y_f import Glyph, GlyphComponent, GlyphCoordinates  # noqa: E402


def _leaf(n_points):
    g = Glyph()
    g.numberOfContours = 1
    g.coordinates = GlyphCoordinates([(100 + 7 * (i % 97) + 300 * (i % 2), 50 + 11 * (i % 61)) for i in range(n_points)])
    g.flags = array("B", [1] * n_points)
    g.endPtsOfContours = [n_points - 1]
    g.program = ttProgram.Program()
    g.program.fromBytecode(b"")
    return g


def _fan(child, n):
    g = Glyph()
    g.numberOfContours = -1
    g.components = []
    for i in range(n):
        c = GlyphComponent()
        c.glyphName = child
        c.x, c.y = i % 50, i // 50
        c.flags = 0x0002 | 0x0001
        g.components.append(c)
    return g


def fan_out_font(points=1500, fan=300):
    glyphs = {".notdef": _leaf(4), "leaf": _leaf(points), "mid": _fan("leaf", fan), "top": _fan("mid", fan)}
    order = list(glyphs)
    fb = FontBuilder(1000, isTTF=True)
    fb.setupGlyphOrder(order)
    fb.setupCharacterMap({0x41: "leaf", 0x42: "mid", 0x43: "top"})
    fb.setupGlyf(glyphs)
    fb.setupHorizontalMetrics({g: (700, 0) for g in order})
    fb.setupHorizontalHeader(ascent=935, descent=-265)
    fb.setupNameTable({"familyName": "Fan Out", "styleName": "Regular"})
    fb.setupOS2()
    fb.setupPost()
    # (fontTools would sum the points of the whole tree into maxp.maxCompositePoints, a u16: no recalculation — the boxes
    # and maxp values are written by hand; neither reader here consults them)
    fb.font.recalcBBoxes = False
    for g in glyphs.values():
        g.xMin, g.yMin, g.xMax, g.yMax = 0, 0, 1000, 1000
    mx = fb.font["maxp"]
    mx.numGlyphs = len(order)
    mx.maxPoints, mx.maxContours, mx.maxCompositePoints, mx.maxCompositeContours = points, 1, 0xFFFF, 0xFFFF
    mx.maxComponentElements, mx.maxComponentDepth = fan, 2
    buf = io.BytesIO()
    fb.save(buf)
    return buf.getvalue()


def test_parts_recorder_stops_at_the_batch_bounds(vg):
    font = fan_out_font()
    assert len(font) < 200_000
    mgr = vg.FontManager(False)
    fid = mgr.add_font_data("Fan Out", font)
    before = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    with pytest.raises(RuntimeError, match="fan-out"):
        mgr.record_glyf_parts(fid)
    grown_mb = (resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - before) / 1024
    assert grown_mb < 400, grown_mb      # 90 000 leaves x ~3 KB would be ~270 MB of parts alone, gigabytes with bigger leaves
    # a modest fan-out (300 leaves) is an ordinary batch
    small = vg.FontManager(False)
    fid2 = small.add_font_data("Fan Small", fan_out_font(points=200, fan=3))
    g = small.record_glyf_parts(fid2)
    assert len(g["parts"]) == 1 + 3 + 9 and int(g["cmd_off"][-1]) == 13 * (200 + 2)
