"""The device's glyf decoder (vgsdf_outlines_submit_glyf): glyphs that arrive as their `glyf` arrays.

With a HIP renderer the host no longer decodes TrueType outlines: it looks every glyph up (cmap, loca, the component records
of composite glyphs), copies the arrays of the simple glyphs it is drawn from as they stand, and the device replays
ttf-parser's walk (glyf.rs: parse_simple_outline + Builder; the call site is /root/reference/src/render/renderer.rs:110).
Checked here against the host's own reader (csrc/host/ttf_face.cpp, itself pinned by the oracle and the golden SHAs):
the same segments in f64, the same rects and the same bitmaps for every glyph of the fixture fonts — composites with
2 x 2 transforms included —, the same PBF files through FontManager, entries that need more command slots than they
were given, and fonts with damaged `glyf` tables, where a batch falls back to the host's reader.
"""
import hashlib
import json

import numpy as np
import pytest

from conftest import FIRA, GOLDEN, NOTO, noto_files

pytestmark = pytest.mark.gpu


def _both_ways(vg, paths, name="Font"):
    mgr = vg.FontManager(True)
    fid = mgr.add_font_with_name(name, paths)
    return mgr, fid, mgr.record_outlines(fid), mgr.record_glyf_parts(fid)


@pytest.mark.parametrize("which", ["fira", "noto_regular", "noto_all"])
def test_decoded_on_the_device_equals_the_host_reader(vg, which):
    paths = {"fira": [FIRA], "noto_regular": [NOTO], "noto_all": noto_files()}[which]
    _, _, o, g = _both_ways(vg, paths)
    assert list(o["ids"]) == list(g["ids"]) and list(o["advances"]) == list(g["advances"])
    assert np.array_equal(o["scale"], g["scale"]) and np.array_equal(o["shift_x"], g["shift_x"])
    n = len(o["ids"])
    # the slots a glyph gets cover the commands the host reader records for it
    n_host = np.diff(o["cmd_off"]).astype(np.int64)
    n_slots = np.diff(g["cmd_off"]).astype(np.int64)
    assert (n_slots >= n_host).all() and n_slots.sum() < 2.2 * n_host.sum()
    assert (g["parts"]["byte_off"] % 4 == 0).all()
    n_composite_parts = int((g["parts"]["plain"] == 0).sum())
    if which == "fira":
        assert n_composite_parts > 100          # accented letters moved by their component offsets
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
    assert n > 1000 and n_seg_d == n_seg_h and out_bytes_d == out_bytes_h
    assert np.array_equal(rects_d, rects_h)
    assert np.array_equal(seg_off_d, seg_off_h)
    assert segs_d.tobytes() == segs_h.tobytes()            # every segment, bit for bit
    assert bitmaps_d is not None and np.array_equal(bitmaps_d, bitmaps_h)


def test_font_manager_takes_the_device_decoder_and_writes_the_same_files(vg):
    """FontManager::render_glyphs with and without the device decoder, against the golden SHA-256s of the PBF files"""
    golden = json.loads((GOLDEN / "pbf_sha256.json").read_text())
    r = vg.Renderer.new_precise(0)
    from test_golden_cpu import set_paths
    for key in ("fira", "noto_all"):
        name, paths = set_paths(key)
        files = {}
        for on in (True, False):
            mgr = vg.FontManager(True)
            mgr.set_glyf_on_device(on)
            fid = mgr.add_font_with_name(name, paths)
            w = vg.DummyWriter()
            mgr.render_glyphs(w, r)
            files[on] = w.files
            t = mgr.timings()
            # (one group per >= 3000 glyphs of a font: Fira's 1686 glyphs are one, the 6480 of the 20 Noto files two)
            assert t["glyphs"] > 1000 and t["glyf_groups"] == ((2 if key == "noto_all" else 1) if on else 0) and t["glyf_fallbacks"] == 0
        assert files[True] == files[False]
        want = golden[key]
        got = {k.split("/", 1)[1].split("-")[0]: hashlib.sha256(v).hexdigest() for k, v in files[True].items()}
        assert got == want, key


def test_an_entry_that_needs_more_slots_than_it_was_given_fails_the_batch(vg):
    """the slot count is part of the ABI: too few -> VGSDF_E_GLYF (nothing is written past a part's slots)"""
    _, _, _, g = _both_ways(vg, [FIRA])
    parts = g["parts"].copy()
    k = int(np.argmax(parts["cmd_cap"]))
    # give the largest part 5 slots fewer and hand them to the next one: the tiling still holds
    assert k + 1 < len(parts) and parts["cmd_cap"][k] > 8
    parts["cmd_cap"][k] -= 5
    parts["cmd_at"][k + 1] -= 5
    parts["cmd_cap"][k + 1] += 5
    ctx = vg.SdfContext(0)
    try:
        ctx.outlines_submit_glyf(g["cmd_off"], parts, g["bytes"], g["scale"], g["shift_x"], capacity=1 << 20)
        with pytest.raises(RuntimeError, match="glyf"):
            ctx.outlines_wait()
        # parts that do not tile the slots are refused before anything runs
        bad = g["parts"].copy()
        bad["cmd_at"][3] += 1
        with pytest.raises(RuntimeError, match="tile"):
            ctx.outlines_submit_glyf(g["cmd_off"], bad, g["bytes"], g["scale"], g["shift_x"], capacity=1 << 20)
        # the context is usable afterwards
        ctx.outlines_submit_glyf(g["cmd_off"], g["parts"], g["bytes"], g["scale"], g["shift_x"], capacity=1 << 20)
        rects, _, _, _ = ctx.outlines_wait()
        assert int(rects["has_raster"].sum()) > 1000
    finally:
        ctx.close()


def _damage_glyf(font: bytes, rng, n_hits: int) -> bytes:
    """random bytes inside the glyf table (flags, coordinates, end points, component records)"""
    n_tables = int.from_bytes(font[4:6], "big")
    for i in range(n_tables):
        rec = 12 + 16 * i
        if font[rec:rec + 4] == b"glyf":
            off, ln = int.from_bytes(font[rec + 8:rec + 12], "big"), int.from_bytes(font[rec + 12:rec + 16], "big")
            b = bytearray(font)
            for pos in rng.integers(0, ln, n_hits):
                b[off + int(pos)] = int(rng.integers(0, 256))
            return bytes(b)
    raise AssertionError("no glyf table")


def test_damaged_glyf_tables_render_like_the_host_reader(vg):
    """Where an entry's arrays do not fit (ttf-parser: no outline for that glyph, the rest of a composite skipped) the device
    flags the batch and FontManager records it with the host's reader; everywhere else the two decoders read the same
    damaged bytes the same way.  Either way: the files of the run with the device decoder equal those without."""
    from pathlib import Path
    rng = np.random.default_rng(7)
    font = Path(FIRA).read_bytes()
    r = vg.Renderer.new_precise(0)
    n_checked = n_fallbacks = 0
    for i in range(12):
        mutant = _damage_glyf(font, rng, n_hits=(1, 3, 40, 400)[i % 4])
        files = {}
        try:
            for on in (True, False):
                mgr = vg.FontManager(True)
                mgr.set_glyf_on_device(on)
                mgr.add_font_data(f"Mutant {i}", mutant)
                w = vg.DummyWriter()
                mgr.render_glyphs(w, r)
                files[on] = w.files
                if on:
                    n_fallbacks += mgr.timings()["glyf_fallbacks"]
        except RuntimeError as e:
            # absurd coordinates can make the front-end refuse a batch ("a glyph flattens to more than 2^28 points ..."):
            # then both ways refuse it
            assert on is True or "glyf" not in str(e), str(e)
            continue
        assert files[True] == files[False], i
        n_checked += 1
    assert n_checked >= 6 and 1 <= n_fallbacks < n_checked   # both ways of reading damaged bytes were taken
