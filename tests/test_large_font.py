"""A font of CJK scale through the whole pipeline: 20 000 glyphs (random closed contours of lines and quadratics) on the code
points U+4E00.., synthesised with fontTools (the reference's testdata lacks its Noto Sans JP / KR / SC files:
.MISSING_LARGE_BLOBS).  Four groups of >= 5000 glyphs in flight two at a time, capacities that grow from one group to the
next, block files far from the first ones.  Expected bytes: the oracle's.  CPU: dummy raster; GPU: HIP raster through both
dispatchers and through three device lanes."""
import io

import numpy as np
import pytest

pytest.importorskip("fontTools")
from fontTools.fontBuilder import FontBuilder  # noqa: E402
from fontTools.pens.ttGlyphPen import TTGlyphPen  # noqa: E402

N_GLYPHS = 20000
FIRST_CP = 0x4E00


@pytest.fixture(scope="module")
def big_font():
    rng = np.random.default_rng(20261004)
    order, cmap, glyphs, metrics = [".notdef"], {}, {".notdef": TTGlyphPen(None).glyph()}, {".notdef": (500, 0)}
    for i in range(N_GLYPHS):
        name = f"g{i}"
        pen = TTGlyphPen(None)
        for _ in range(int(rng.integers(1, 4))):
            cx, cy, r = rng.integers(150, 850), rng.integers(0, 700), rng.integers(40, 300)
            k = int(rng.integers(3, 9))
            ang = np.sort(rng.uniform(0, 2 * np.pi, k))
            pts = [(int(cx + r * np.cos(a)), int(cy + r * np.sin(a))) for a in ang]
            pen.moveTo(pts[0])
            for j in range(1, k):
                if rng.random() < 0.6:
                    mid = ((pts[j - 1][0] + pts[j][0]) // 2 + int(rng.integers(-60, 60)), (pts[j - 1][1] + pts[j][1]) // 2 + int(rng.integers(-60, 60)))
                    pen.qCurveTo(mid, pts[j])
                else:
                    pen.lineTo(pts[j])
            pen.closePath()
        glyphs[name] = pen.glyph()
        order.append(name)
        cmap[FIRST_CP + i] = name
        metrics[name] = (int(rng.integers(400, 1100)), 0)
    fb = FontBuilder(1000, isTTF=True)
    fb.setupGlyphOrder(order)
    fb.setupCharacterMap(cmap)
    fb.setupGlyf(glyphs)
    fb.setupHorizontalMetrics(metrics)
    fb.setupHorizontalHeader(ascent=900, descent=-100)
    fb.setupNameTable({"familyName": "Big Synthetic", "styleName": "Regular"})
    fb.setupOS2()
    fb.setupPost()
    buf = io.BytesIO()
    fb.save(buf)
    return buf.getvalue()


def _oracle_files(oracle, data, fid, mode):
    font = oracle.Font(data)
    return {f"{fid}/{b * 256}-{b * 256 + 255}.pbf": oracle.render_block([font], fid, b * 256, mode)[0] for b in range(256)}


def test_large_font_dummy(vg, oracle, big_font):
    m = vg.FontManager(True)
    fid = m.add_font_data("Big Synthetic Regular", big_font)
    assert int(m.block_counts(fid).sum()) == N_GLYPHS
    w = vg.DummyWriter()
    m.render_glyphs(w, vg.Renderer.new_dummy())
    assert w.files == _oracle_files(oracle, big_font, fid, oracle.DUMMY)


@pytest.mark.gpu
def test_large_font_on_the_gpu(vg, oracle, big_font):
    m = vg.FontManager(True)
    fid = m.add_font_data("Big Synthetic Regular", big_font)
    want = _oracle_files(oracle, big_font, fid, oracle.PRECISE)
    r = vg.Renderer.new_precise(0)
    for fe in (True, False):
        m.set_device_front_end(fe)
        w = vg.DummyWriter()
        m.render_glyphs(w, r)
        bad = [n for n in want if w.files[n] != want[n]]
        assert not bad, (fe, len(bad), bad[:3])
        assert m.timings()["glyphs"] == N_GLYPHS
    m.set_device_front_end(True)
    w = vg.DummyWriter()
    m.render_glyphs(w, vg.Renderer.new_multi([0, 0, 0]))
    assert w.files == want
    # a second font in the same manager: groups now span fonts
    from conftest import FIRA
    m.add_font_with_name("Fira Sans Regular", [FIRA])
    w = vg.DummyWriter()
    m.render_glyphs(w, r)
    assert all(w.files[n] == want[n] for n in want) and len(w.files) == 512
