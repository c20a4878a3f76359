"""GPU: the product end to end (C++ FontManager / GlyphBlock / Renderer over the HIP raster)
against the oracle: PBF files byte for byte, and the reference's pixel KATs."""
import numpy as np
import pytest

from conftest import FIRA, NOTO, noto_files
from test_oracle_kat import GLYPH_KATS, ascii_art, FIRA_PBF_SIZES

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip(vg):
    return vg.Renderer.new_precise(0)


def test_render_glyph_kats(vg, hip):
    # renderer.rs:176-287 with the HIP back-end in place of renderer_precise
    m = vg.FontManager(False)
    fid = m.add_font_with_name("Fira Sans Regular", [FIRA])
    g = hip.render_glyph(m, fid, 32)
    assert g.metrics() == (0, 0, 0, 0, 6) and not g.has_bitmap
    for cp, (metrics, art) in GLYPH_KATS.items():
        g = hip.render_glyph(m, fid, cp)
        assert g.metrics() == metrics
        assert g.bitmap.size == (g.width + 6) * (g.height + 6)
        assert ascii_art(g.bitmap) == art


def render_all_files(vg, hip, name, paths, threads=0, blocks_per_batch=0, device_front_end=True):
    m = vg.FontManager(True)
    m.set_threads(threads, blocks_per_batch)
    m.set_device_front_end(device_front_end)
    fid = m.add_font_with_name(name, paths)
    w = vg.DummyWriter()
    m.render_glyphs(w, hip)
    return fid, w, m


def check_files(oracle, fid, w, paths, mode):
    fonts = [oracle.Font(p) for p in paths]
    bad = []
    for blk in range(256):
        want, _, _ = oracle.render_block(fonts, fid, blk * 256, mode)
        if w.files[f"{fid}/{blk * 256}-{blk * 256 + 255}.pbf"] != want:
            bad.append(blk * 256)
    assert not bad, bad


@pytest.mark.parametrize("fe", [True, False], ids=["device_front_end", "host_tessellation"])
def test_fira_pbf_bytes(vg, oracle, hip, fe):
    fid, w, m = render_all_files(vg, hip, "Fira Sans Regular", [FIRA], device_front_end=fe)
    for start, size in FIRA_PBF_SIZES.items():  # recurse.rs:341-367 sizes hold for real pixels too
        assert len(w.files[f"{fid}/{start}-{start + 255}.pbf"]) == size
    check_files(oracle, fid, w, [FIRA], oracle.PRECISE)


@pytest.mark.parametrize("fe", [True, False], ids=["device_front_end", "host_tessellation"])
def test_noto_regular_pbf_bytes(vg, oracle, hip, fe):
    fid, w, m = render_all_files(vg, hip, "Noto Sans Regular", [NOTO], blocks_per_batch=7, device_front_end=fe)
    check_files(oracle, fid, w, [NOTO], oracle.BRUTE)


@pytest.mark.parametrize("fe", [True, False], ids=["device_front_end", "host_tessellation"])
def test_noto_all_languages_pbf_bytes(vg, oracle, hip, fe):
    # config 3: 20 files merged, 6480 code points, 45 non-empty blocks
    fid, w, m = render_all_files(vg, hip, "Noto Sans Regular", noto_files(), device_front_end=fe)
    t = m.timings()
    assert (t["glyphs"], t["rasters"], t["pixels"], t["segments"]) == (6480, 6445, 3295280, 3956999)
    check_files(oracle, fid, w, noto_files(), oracle.BRUTE)


def test_single_threaded_and_batched_agree(vg, hip):
    a = render_all_files(vg, hip, "Fira Sans Regular", [FIRA], threads=1, blocks_per_batch=1)[1].files
    b = render_all_files(vg, hip, "Fira Sans Regular", [FIRA], threads=4, blocks_per_batch=256)[1].files
    c = render_all_files(vg, hip, "Fira Sans Regular", [FIRA], threads=3, blocks_per_batch=5, device_front_end=False)[1].files
    assert a == b == c


def test_several_fonts_in_groups(vg, hip):
    """several fonts in one manager go through the device front-end group by group (reused buffers).
    Every file must equal what the same font gives when it is rendered alone, and arrive in task order
    (fonts sorted by id, blocks ascending) — also with groups smaller than a font and with the null sink."""
    fonts = [("Fira Sans Regular", [FIRA]), ("Noto Sans Regular", [NOTO])] + \
            [(f"Noto {i}", [p]) for i, p in enumerate(noto_files()[:5])]
    alone = {}
    for name, paths in fonts:
        fid, w, _ = render_all_files(vg, hip, name, paths)
        alone.update(w.files)
    for blocks_per_batch in (0, 100, 7):
        m = vg.FontManager(True)
        m.set_threads(0, blocks_per_batch)
        for name, paths in fonts:
            m.add_font_with_name(name, paths)
        w = vg.DummyWriter()
        m.render_glyphs(w, hip)
        assert set(w.files) == set(alone)
        assert all(w.files[k] == alone[k] for k in alone), blocks_per_batch
        order = [s.split(" (")[0] for s in w.inner if s.endswith(")")]
        assert order == sorted(order, key=lambda p: (p.split("/")[0], int(p.split("/")[1].split("-")[0])))
        m.render_glyphs(None, hip)
        assert m.timings()["pbf_bytes"] == sum(len(v) for v in alone.values())
