"""Product host façade (C++: FontManager / GlyphBlock / Renderer / PBF) on the CPU: mirrors the
reference's own tests for this layer, with the dummy renderer where pixels are not the
point (exactly as the reference does), and checks the HIP mode fails loudly without a GPU."""
import re

import numpy as np
import pytest

from conftest import FIRA, NOTO, NOTO_DIR, noto_files
from test_oracle_kat import FIRA_BLOCK_COUNTS, FIRA_PBF_SIZES


@pytest.fixture(scope="module")
def fira_mgr(vg):
    m = vg.FontManager(False)
    assert m.add_font_with_name("Fira Sans - Regular", [FIRA]) == "fira_sans_regular"
    return m


def test_name_to_id(vg):
    # manager.rs:141-147
    assert vg.name_to_id("Fira Sans - Regular") == "fira_sans_regular"
    assert vg.name_to_id("  Noto_Sans--Arabic  Bold ") == "noto_sans_arabic_bold"
    assert vg.name_to_id("A\tB") == "a_b"
    assert vg.name_to_id("---") == ""


def test_add_path_names_the_font_from_its_name_table(vg):
    # manager.rs:39-53, metadata.rs:136-144; more in tests/test_ingestion_and_sinks.py
    m = vg.FontManager(False)
    m.add_path(FIRA)
    assert m.font_ids() == ["fira_sans_regular"]
    with pytest.raises(RuntimeError):
        m.add_path(FIRA.parent / "no such file.ttf")


def test_get_blocks_counts(fira_mgr):
    # wrapper.rs:177-227 test_get_blocks
    counts = fira_mgr.block_counts("fira_sans_regular")
    assert len(counts) == 256
    assert {i * 256: int(c) for i, c in enumerate(counts) if c} == FIRA_BLOCK_COUNTS
    assert counts[3840 // 256] == 0  # Tibetan range present but empty


def test_render_glyphs_dummy_pbf_sizes(vg, fira_mgr):
    # recurse.rs:324-371 / merge.rs:141-186: sizes of every Fira block under the dummy renderer
    w = vg.DummyWriter()
    fira_mgr.render_glyphs(w, vg.Renderer.new_dummy())
    assert w.inner[0] == "fira_sans_regular/"
    sizes = {}
    for e in w.inner[1:]:
        m = re.fullmatch(r"fira_sans_regular/(\d+)-(\d+)\.pbf \((\d+)\)", e)
        assert m and int(m.group(2)) == int(m.group(1)) + 255
        sizes[int(m.group(1))] = int(m.group(3))
    assert sorted(sizes) == [i * 256 for i in range(256)]
    assert {k: v for k, v in sizes.items() if not 32 <= v <= 34} == FIRA_PBF_SIZES
    t = fira_mgr.timings()
    assert (t["blocks"], t["glyphs"], t["rasters"]) == (256, 1686, 1679)
    assert (t["pixels"], t["segments"]) == (758736, 600952)


def test_render_glyphs_two_fonts(vg):
    # manager.rs:163-231 test_render_glyphs (add_paths replaced by explicit names)
    m = vg.FontManager(False)
    m.add_font_with_name("Fira Sans Regular", [FIRA])
    m.add_font_with_name("Noto Sans Regular", [NOTO, NOTO_DIR / "Noto Sans Arabic - Regular.ttf",
                                               NOTO_DIR / "Noto Sans Tamil - Regular.ttf"])
    w = vg.DummyWriter()
    m.render_glyphs(w, vg.Renderer.new_dummy())
    assert "fira_sans_regular/" in w.inner and "noto_sans_regular/" in w.inner
    for font in ("fira_sans_regular", "noto_sans_regular"):
        starts = sorted(int(p.split("/")[1].split("-")[0]) for p in w.files if p.startswith(font + "/"))
        assert starts == [i * 256 for i in range(256)]
    assert len(w.files["noto_sans_regular/256-511.pbf"]) > 1000
    assert len(w.files["noto_sans_regular/3840-4095.pbf"]) < 100


def test_render_glyph_metrics_dummy(vg, fira_mgr):
    # renderer.rs:176-270: metrics do not depend on the raster back-end
    r = vg.Renderer.new_dummy()
    g = r.render_glyph(fira_mgr, "fira_sans_regular", 32)
    assert g.metrics() == (0, 0, 0, 0, 6) and not g.has_bitmap
    for cp, want in ((65, (14, 17, 0, -7, 13)), (230, (19, 14, 0, -11, 19)), (96, (7, 5, 0, -4, 7))):
        g = r.render_glyph(fira_mgr, "fira_sans_regular", cp)
        assert g.metrics() == want
        assert g.bitmap.shape == (g.height + 6, g.width + 6) and not g.bitmap.any()
    assert r.render_glyph(fira_mgr, "fira_sans_regular", 0xD800) is None  # char::from_u32 fails
    assert r.render_glyph(fira_mgr, "fira_sans_regular", 0x0F00) is None  # not in cmap


def test_pbf_roundtrip_layout(vg):
    # protobuf/glyphs.rs:91-139: decode what we encode with an independent mini-decoder
    g1 = vg.PbfGlyph(id=66, has_bitmap=1, width=1, height=2, left=-3, top=4, advance=5)
    g1.bitmap = np.arange(7 * 8, dtype=np.uint8).reshape(8, 7)
    g0 = vg.PbfGlyph(id=65, has_bitmap=0, advance=12)
    data = vg.pbf_encode("MyFont", "0-255", [g1, g0])

    def varint(b, i):
        v = s = 0
        while True:
            v |= (b[i] & 0x7F) << s
            s += 7
            i += 1
            if not b[i - 1] & 0x80:
                return v, i
    assert data[0] == 0x0A
    n, i = varint(data, 1)
    assert i + n == len(data)
    fields = []
    while i < len(data):
        tag = data[i]
        n, i = varint(data, i + 1)
        fields.append((tag, data[i:i + n]))
        i += n
    assert fields[0] == (0x0A, b"MyFont") and fields[1] == (0x12, b"0-255")
    assert [t for t, _ in fields[2:]] == [0x1A, 0x1A]
    # ascending id: 65 first; required scalars always present, zig-zag for left/top
    assert fields[2][1] == bytes([0x08, 65, 0x18, 0, 0x20, 0, 0x28, 0, 0x30, 0, 0x38, 12])
    body = fields[3][1]
    assert body[:2] == bytes([0x08, 66]) and body[2] == 0x12 and body[3] == 56
    assert body[4:60] == bytes(range(56))
    assert body[60:] == bytes([0x18, 1, 0x20, 2, 0x28, 5, 0x30, 8, 0x38, 5])


def test_hip_renderer_fails_loudly_without_gpu(vg):
    if vg.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(vg.VgsdfError) as e:
        vg.Renderer.new_precise(0)
    assert "no CPU fallback" in str(e.value)
    with pytest.raises(vg.VgsdfError):
        vg.SdfContext(0)


@pytest.mark.timeout(300)
def test_worker_pool_survives_many_short_forks(vg):
    """the host pool polls briefly before it sleeps: hundreds of back-to-back fork/joins (three per render), with the
    gaps between renders longer and shorter than the polling window, must neither hang nor lose a block"""
    import time
    m = vg.FontManager(True)
    m.set_threads(8, 16)
    fid = m.add_font_with_name("Fira Sans Regular", [FIRA])
    r = vg.Renderer.new_dummy()
    w = vg.DummyWriter()
    m.render_glyphs(w, r)
    want = {k: len(v) for k, v in w.files.items()}
    for i in range(120):
        w = vg.DummyWriter()
        m.render_glyphs(w, r)
        assert {k: len(v) for k, v in w.files.items()} == want
        if i % 10 == 0:
            time.sleep(0.002)   # let the workers fall asleep on the condition variable


def test_in_place_assembly_equals_encoding_afterwards_on_the_cpu(vg):
    """host tessellation + dummy raster: blocks assembled in place in the output arena (bitmaps where the finished PBF has
    them, headers written around them) equal blocks encoded from packed bitmaps"""
    from conftest import NOTO
    m = vg.FontManager(True)
    m.add_font_with_name("Fira Sans Regular", [FIRA])
    m.add_font_with_name("Noto Sans Regular", [NOTO])
    r = vg.Renderer.new_dummy()
    files = []
    for in_place in (True, False):
        m.set_in_place_pbf(in_place)
        w = vg.DummyWriter()
        m.render_glyphs(w, r)
        files.append(w.files)
    assert files[0] == files[1] and len(files[0]) == 512
    assert len(files[0]["fira_sans_regular/0-255.pbf"]) == 80022  # recurse.rs:344
