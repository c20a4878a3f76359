"""The product's host stage (C++ TTF reader + tessellation + bbox, and the PBF encoder) against
the CPU oracle: segment lists, rects and PBF bytes must be identical bit for bit, for every
glyph of every fixture font.  Two independent implementations (C oracle / C++ product) of
the same reference rules.  No GPU."""
import numpy as np
import pytest

from conftest import FIRA, NOTO, noto_files


def compare_font(vg, oracle, name, paths):
    m = vg.FontManager(True)
    fid = m.add_font_with_name(name, paths)
    hb = m.build_batch(fid)
    b = hb.batch
    fonts = [oracle.Font(p) for p in paths]
    prov = {}
    for fi, f in enumerate(fonts):
        for cp in f.codepoints():
            if cp <= 0xFFFF:
                prov.setdefault(int(cp), fi)
    want_ids, n_some = [], 0
    g = 0
    for cp in sorted(prov):
        r = fonts[prov[cp]].prepare_glyph(cp)
        if r is None:
            continue
        n_some += 1
        info, segs = r
        if not info.has_bitmap:
            continue
        assert int(hb.ids[g]) == cp
        a, e = int(b.seg_off[g]), int(b.seg_off[g + 1])
        assert e - a == info.n_segments, cp
        got = np.stack([b.seg_sx[a:e], b.seg_sy[a:e], b.seg_ex[a:e], b.seg_ey[a:e]], axis=1)
        assert got.tobytes() == segs.tobytes(), f"segments differ for U+{cp:04X}"
        assert (int(b.x0[g]), int(b.y0[g]), int(b.w[g]), int(b.h[g])) == (info.x0, info.y0, info.w, info.h)
        g += 1
    assert g == b.n_glyphs
    assert n_some == hb.n_jobs
    return m, fid, fonts


def test_fira_segments_identical(vg, oracle):
    compare_font(vg, oracle, "Fira Sans Regular", [FIRA])


def test_noto_regular_segments_identical(vg, oracle):
    compare_font(vg, oracle, "Noto Sans Regular", [NOTO])


def test_noto_all_languages_merge_identical(vg, oracle):
    # config 3: 20 files merged under one id, sorted file order, first provider wins
    m, fid, fonts = compare_font(vg, oracle, "Noto Sans Regular", noto_files())
    assert int(m.block_counts(fid).sum()) == 6480


def test_pbf_bytes_identical_dummy(vg, oracle):
    # every block of the merged Noto family + Fira: PBF bytes (canonical id order) equal
    for name, paths in (("Fira Sans Regular", [FIRA]), ("Noto Sans Regular", noto_files())):
        m = vg.FontManager(True)
        fid = m.add_font_with_name(name, paths)
        w = vg.DummyWriter()
        m.render_glyphs(w, vg.Renderer.new_dummy())
        fonts = [oracle.Font(p) for p in paths]
        for blk in range(256):
            want, _, _ = oracle.render_block(fonts, fid, blk * 256, oracle.DUMMY)
            assert w.files[f"{fid}/{blk * 256}-{blk * 256 + 255}.pbf"] == want, (name, blk)


def test_single_block_render_matches(vg, oracle):
    m = vg.FontManager(False)
    fid = m.add_font_with_name("Fira Sans Regular", [FIRA])
    got = m.render_block(vg.Renderer.new_dummy(), fid, 7424)
    want, _, _ = oracle.render_block([oracle.Font(FIRA)], fid, 7424, oracle.DUMMY)
    assert got == want and len(got) == 7260


def test_recorded_outlines_match_oracle(vg, oracle):
    """host half of the device front-end: the recorded OutlineBuilder callbacks, scale, shift and
    advance of every glyph equal the oracle's (all 20 merged Noto files + Fira)"""
    for name, paths in (("Fira Sans Regular", [FIRA]), ("Noto Sans Regular", noto_files())):
        m = vg.FontManager(False)
        rec = m.record_outlines(m.add_font_with_name(name, paths))
        fonts = [oracle.Font(p) for p in paths]
        prov = {}
        for fi, f in enumerate(fonts):
            for cp in f.codepoints():
                if cp <= 0xFFFF:
                    prov.setdefault(int(cp), fi)
        g = 0
        for cp in sorted(prov):
            f = fonts[prov[cp]]
            gid = f.glyph_index(cp)
            if gid is None:
                continue
            assert int(rec["ids"][g]) == cp
            want = f.outline(gid)
            got = rec["cmds"][rec["cmd_off"][g]:rec["cmd_off"][g + 1]]
            assert len(got) == len(want), cp
            for c, (kind, x1, y1, x2, y2, x, y) in zip(got, want):
                assert (int(c["kind"]), float(c["x1"]), float(c["y1"]), float(c["x2"]), float(c["y2"]), float(c["x"]),
                        float(c["y"])) == (kind, x1, y1, x2, y2, x, y), cp
            sc = 24.0 / f.units_per_em
            adv = (f.hor_advance(gid) or 0) * sc * 0.95
            a = float(np.floor(adv + 0.5))
            assert rec["scale"][g] == sc and rec["shift_x"][g] == (a - adv) / 2.0 and int(rec["advances"][g]) == int(a)
            g += 1
        assert g == len(rec["ids"])
