"""In-place PBF assembly at the varint boundaries of the wire format (glyph.rs:10-41): a TrueType font synthesised with
fontTools whose glyphs are boxes of chosen sizes and positions at 24 units per em (font units = pixels), so that bitmap
sizes cross 127 / 128 and 16 383 / 16 384 bytes, width / height cross 127 / 128, left / top go negative and beyond +-64
(zig-zag varints of 1 and 2 bytes), advances cross 127 / 128 and ids sit in every varint class of the BMP.  Expected
bytes: the oracle's PBF of the same font.  CPU: host tessellation + dummy raster; GPU: both dispatchers, in place and
encoded afterwards, and three device lanes."""
import io

import pytest

pytest.importorskip("fontTools")
from fontTools.fontBuilder import FontBuilder  # noqa: E402
from fontTools.pens.ttGlyphPen import TTGlyphPen  # noqa: E402


def _box_font():
    boxes = {}
    # (x0, y0, w, h, advance) in pixels; bitmap = (w + 6) * (h + 6) bytes
    sizes = [(0, 0, 1, 1, 3), (0, 0, 5, 5, 8), (0, 0, 5, 6, 120), (0, 0, 6, 5, 140), (0, 0, 121, 1, 127), (0, 0, 122, 1, 128),
             (0, 0, 1, 121, 134), (0, 0, 1, 122, 135), (0, 0, 121, 122, 200), (0, 0, 122, 122, 300), (0, 0, 122, 123, 16383),
             (-70, -70, 10, 10, 1), (-67, -40, 3, 200, 2), (60, 60, 4, 4, 70), (61, 62, 4, 4, 71), (-3, -3, 7, 7, 0), (-4, 24, 2, 2, 9),
             (100, -200, 30, 17, 500), (0, 21, 2, 6, 4), (0, 20, 2, 6, 4)]
    cps = [0x21, 0x7F, 0x80, 0xFF, 0x100, 0x3FFF, 0x4000, 0x4001, 0x7FFF, 0x8000, 0xFFFD, 0x22, 0x23, 0x24, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2A]
    order, cmap, glyphs, metrics = [".notdef"], {}, {}, {".notdef": (10, 0)}
    glyphs[".notdef"] = TTGlyphPen(None).glyph()
    for i, ((x0, y0, w, h, adv), cp) in enumerate(zip(sizes, cps)):
        name = f"box{i}"
        pen = TTGlyphPen(None)
        pen.moveTo((x0, y0))
        pen.lineTo((x0 + w, y0))
        pen.lineTo((x0 + w, y0 + h))
        pen.lineTo((x0, y0 + h))
        pen.closePath()
        glyphs[name] = pen.glyph()
        order.append(name)
        cmap[cp] = name
        metrics[name] = (adv, x0)
        boxes[cp] = (x0, y0, w, h, adv)
    glyphs["empty"] = TTGlyphPen(None).glyph()  # a glyph without an outline (PbfGlyph::empty) between them
    order.append("empty")
    cmap[0x2B] = "empty"
    metrics["empty"] = (7, 0)
    fb = FontBuilder(24, isTTF=True)
    fb.setupGlyphOrder(order)
    fb.setupCharacterMap(cmap)
    fb.setupGlyf(glyphs)
    fb.setupHorizontalMetrics(metrics)
    fb.setupHorizontalHeader(ascent=20, descent=-4)
    fb.setupNameTable({"familyName": "Boxes", "styleName": "Regular"})
    fb.setupOS2()
    fb.setupPost()
    buf = io.BytesIO()
    fb.save(buf)
    return buf.getvalue(), boxes


def _varint_len(v):
    n = 1
    while v >= 0x80:
        v >>= 7
        n += 1
    return n


def test_the_font_reaches_the_boundaries(oracle):
    data, boxes = _box_font()
    sizes = sorted((w + 6) * (h + 6) for (_, _, w, h, _) in boxes.values())
    assert {_varint_len(s) for s in sizes} == {1, 2, 3} and 127 * 128 in sizes and 128 * 128 in sizes and 121 in sizes
    font = oracle.Font(data)
    assert len(font.codepoints()) == len(boxes) + 1


def test_boxes_on_the_cpu_dummy(vg, oracle):
    data, _ = _box_font()
    m = vg.FontManager(True)
    fid = m.add_font_data("Boxes Regular", data)
    font = oracle.Font(data)
    for in_place in (True, False):
        m.set_in_place_pbf(in_place)
        w = vg.DummyWriter()
        m.render_glyphs(w, vg.Renderer.new_dummy())
        for b in range(256):
            assert w.files[f"{fid}/{b * 256}-{b * 256 + 255}.pbf"] == oracle.render_block([font], fid, b * 256, oracle.DUMMY)[0], (in_place, b)


@pytest.mark.gpu
def test_boxes_on_the_gpu(vg, oracle):
    data, _ = _box_font()
    m = vg.FontManager(True)
    fid = m.add_font_data("Boxes Regular", data)
    font = oracle.Font(data)
    want = {f"{fid}/{b * 256}-{b * 256 + 255}.pbf": oracle.render_block([font], fid, b * 256, oracle.PRECISE)[0] for b in range(256)}
    r = vg.Renderer.new_precise(0)
    for fe in (True, False):
        for in_place in (True, False):
            m.set_device_front_end(fe)
            m.set_in_place_pbf(in_place)
            w = vg.DummyWriter()
            m.render_glyphs(w, r)
            bad = [n for n in want if w.files[n] != want[n]]
            assert not bad, (fe, in_place, bad[:4])
    lanes = vg.Renderer.new_multi([0, 0, 0])
    m.set_device_front_end(True)
    m.set_in_place_pbf(True)
    w = vg.DummyWriter()
    m.render_glyphs(w, lanes)
    assert w.files == want
