"""`CFF ` outlines (OpenType fonts with Type 2 charstrings) through the host reader.

The reference gets them from ttf-parser (`face.outline_glyph`, src/render/renderer.rs:110, then `curve_to` in
src/render/ring_builder.rs:98-110); none of its fixtures holds a CFF font, so PARITY WITH THE CRATE IS UNPINNED for
this file.  What is checked instead: the reader's callbacks against fontTools' own charstring interpreter
(an independent implementation) on fonts synthesised here with fontTools — the outlines of Fira Sans converted to
charstrings, hand-written programs covering every path operator, hints, widths and local / global subroutines, a
CID-keyed variant — and, on the GPU, the rendered bitmaps of those outlines against the oracle's raster.
"""
import io

import numpy as np
import pytest

from conftest import FIRA

fontTools = pytest.importorskip("fontTools")
from fontTools.cffLib import SubrsIndex  # noqa: E402
from fontTools.fontBuilder import FontBuilder  # noqa: E402
from fontTools.misc.psCharStrings import T2CharString  # noqa: E402
from fontTools.pens.recordingPen import RecordingPen  # noqa: E402
from fontTools.pens.t2CharStringPen import T2CharStringPen  # noqa: E402
from fontTools.ttLib import TTFont  # noqa: E402

M, L, Q, C, Z = 0, 1, 2, 3, 4


def _build(order, cmap, charstrings, widths, upem=1000, local_subrs=(), global_subrs=()):
    fb = FontBuilder(upem, isTTF=False)
    fb.setupGlyphOrder(order)
    fb.setupCharacterMap(cmap)
    fb.setupCFF("SynthCFF-Regular", {"FullName": "Synth CFF Regular"}, charstrings, {})
    fb.setupHorizontalMetrics({g: (widths[g], 0) for g in order})
    fb.setupHorizontalHeader(ascent=935, descent=-265)
    fb.setupNameTable({"familyName": "Synth CFF", "styleName": "Regular"})
    fb.setupOS2()
    fb.setupPost()
    cff = fb.font["CFF "].cff
    top = cff.topDictIndex[0]
    if local_subrs:
        top.Private.Subrs = SubrsIndex()
        for prog in local_subrs:
            top.Private.Subrs.append(T2CharString(program=list(prog)))
    for prog in global_subrs:
        cff.GlobalSubrs.append(T2CharString(program=list(prog)))
    buf = io.BytesIO()
    fb.save(buf)
    return buf.getvalue()


@pytest.fixture(scope="module")
def fira_cff():
    """the first 400 glyphs of Fira Sans (composites decomposed, quadratics raised to cubics by the pen)"""
    src = TTFont(FIRA)
    gs = src.getGlyphSet()
    order = src.getGlyphOrder()[:400]
    cmap = {cp: g for cp, g in src.getBestCmap().items() if g in order}
    cs = {}
    for g in order:
        pen = T2CharStringPen(gs[g].width, gs)
        gs[g].draw(pen)
        cs[g] = pen.getCharString()
    return _build(order, cmap, cs, {g: gs[g].width for g in order}, src["head"].unitsPerEm)


# every path operator of Technical Note #5177, widths, hints, subroutines (bias 107 for < 1240 subroutines)
_OPS = [600, 50, 100, "rmoveto", 10, 20, 30, 40, "rlineto", 15, "hlineto", 5, 6, 7, "vlineto", 8, 9, "hlineto",
        1, 2, 3, 4, 5, 6, "rrcurveto", 1, 2, 3, 4, 5, 6, 7, 8, "rcurveline", 9, 10, 11, 12, 1, 2, 3, 4, 5, 6, "rlinecurve",
        3, 10, 20, 30, 40, "vvcurveto", 10, 20, 30, 40, 11, 21, 31, 41, "vvcurveto",
        4, 10, 20, 30, 40, "hhcurveto", 10, 20, 30, 40, "hhcurveto",
        10, 20, 30, 40, 11, 21, 31, 41, 5, "hvcurveto", 10, 20, 30, 40, "hvcurveto",
        10, 20, 30, 40, "vhcurveto", 10, 20, 30, 40, 11, 21, 31, 41, 12, 22, 32, 42, 7, "vhcurveto",
        1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 50, "flex", 10, 20, 30, 40, 50, 60, 70, "hflex",
        10, 20, 30, 40, 50, 60, 70, 80, 90, "hflex1", 10, 20, 30, 40, 50, 60, 70, 80, 90, 100, 110, "flex1",
        100, 1, 2, 3, 4, 5, 6, 7, 8, 9, -300, "flex1",
        -400, "hmoveto", 20, 20, -20, 20, "rlineto", -30, "vmoveto", 10, -10, 10, 10, "rlineto", "endchar"]
_HINTS = [10, 20, "hstemhm", 5, 10, "hintmask", b"\xc0", 100, 100, "rmoveto", 50, "hlineto", 50, "vlineto",
          "cntrmask", b"\x40", -50, "hlineto", "endchar"]
_HINTS_W = [555, 10, 20, 30, 40, "hstem", 5, 10, 15, 20, "vstem", 7, "vmoveto", 50, 60, 70, "hlineto", "endchar"]
_SUBRS = [100, 200, "rmoveto", -107, "callsubr", -106, "callsubr", 5, 5, -107, "callgsubr", -105, "callgsubr"]
_LOCAL = [[10, 20, "rlineto", "return"], [30, 0, "rlineto", -107, "callgsubr", "return"]]
_GLOBAL = [[-5, 40, "rlineto", "return"], [1, 2, 3, 4, 5, 6, "rrcurveto", -106, "callgsubr", "return"], [9, 9, "rlineto", "endchar"]]
_FIXED = [100.5, 200.25, "rmoveto", 10.5, 0.125, "rlineto", -3.75, 8, "rlineto", 1.5, 2.5, 3.5, 4.5, 5.5, 6.5, "rrcurveto", "endchar"]
_EMPTY = [500, "endchar"]


@pytest.fixture(scope="module")
def ops_cff():
    names = [".notdef", "ops", "hints", "hintsw", "subrs", "fixed", "empty"]
    progs = [[0, "hmoveto", "endchar"], _OPS, _HINTS, _HINTS_W, _SUBRS, _FIXED, _EMPTY]
    cs = {n: T2CharString(program=list(p)) for n, p in zip(names, progs)}
    cmap = {0x41 + i: n for i, n in enumerate(names[1:])}
    return _build(names, cmap, cs, {n: 600 for n in names}, local_subrs=_LOCAL, global_subrs=_GLOBAL)


def _fonttools_callbacks(font_bytes):
    f = TTFont(io.BytesIO(font_bytes))
    gs, cmap = f.getGlyphSet(), f.getBestCmap()
    out = {}
    for cp, name in cmap.items():
        rp = RecordingPen()
        gs[name].draw(rp)
        seq = []
        for op, a in rp.value:
            if op == "moveTo":
                seq.append((M, 0, 0, 0, 0) + tuple(a[0]))
            elif op == "lineTo":
                seq.append((L, 0, 0, 0, 0) + tuple(a[0]))
            elif op == "curveTo":
                seq.append((C,) + tuple(a[0]) + tuple(a[1]) + tuple(a[2]))
            elif op in ("closePath", "endPath"):
                seq.append((Z, 0, 0, 0, 0, 0, 0))
            else:
                raise AssertionError(op)
        out[cp] = [(t[0],) + tuple(float(np.float32(v)) for v in t[1:]) for t in seq]
    return out


def _product_callbacks(vg, font_bytes):
    mgr = vg.FontManager(parallel=False)
    fid = mgr.add_font_data("Synth", font_bytes)
    o = mgr.record_outlines(fid)
    out = {}
    for i, cp in enumerate(o["ids"]):
        got = o["cmds"][o["cmd_off"][i]:o["cmd_off"][i + 1]]
        out[int(cp)] = [(int(c["kind"]),) + ((0.0,) * 6 if c["kind"] == Z else
                                             tuple(float(c[k]) for k in ("x1", "y1", "x2", "y2", "x", "y"))) for c in got]
    return out, o


def test_fira_outlines_as_cff_match_fonttools(vg, fira_cff):
    want = _fonttools_callbacks(fira_cff)
    got, _ = _product_callbacks(vg, fira_cff)
    assert len(got) > 250
    n_curves = 0
    for cp, seq in got.items():
        assert seq == want[cp], hex(cp)
        n_curves += sum(1 for t in seq if t[0] == C)
    assert n_curves > 2000
    assert set(got) == set(want)   # glyphs without points (space) are jobs without commands: PbfGlyph::empty, renderer.rs:118-120


def test_every_operator_hints_widths_and_subroutines(vg, ops_cff):
    want = _fonttools_callbacks(ops_cff)
    got, _ = _product_callbacks(vg, ops_cff)
    for cp, name in zip(range(0x41, 0x46), ["ops", "hints", "hintsw", "subrs", "fixed"]):
        assert got[cp] == want[cp], name
        assert any(t[0] == Z for t in got[cp])
    assert sum(1 for t in got[0x41] if t[0] == C) == 25   # "ops": every curve operator ran
    assert got[0x46] == []                               # "empty": no callbacks -> PbfGlyph::empty (renderer.rs:118-120)


def test_seac_accented_glyphs(oracle, vg):
    """`adx ady bchar achar endchar`: the base glyph and the accent (codes of the StandardEncoding, looked up through the
    charset) drawn one after the other, the accent moved by (adx, ady) — against fontTools' decomposition"""
    from fontTools.pens.recordingPen import DecomposingRecordingPen
    names = [".notdef", "A", "acute", "Aacute", "dieresis", "Adieresis", "o", "oacute"]
    progs = [
        [0, "hmoveto", "endchar"],
        [600, 100, 0, "rmoveto", 200, 700, "rlineto", 200, -700, "rlineto", "endchar"],
        [300, 10, 20, "hstem", 50, 60, "rmoveto", 80, 120, "rlineto", -40, 0, "rlineto", "endchar"],
        [640, 150, 700, 65, 194, "endchar"],          # width 640, A + acute at (150, 700)
        [250, 0, "rmoveto", 60, "hlineto", 60, "vlineto", -60, "hlineto", 100, 0, "rmoveto", 60, "hlineto", 60, "vlineto", -60, "hlineto", "endchar"],
        [100, 720, 65, 200, "endchar"],                # no width operand: A + dieresis
        [550, 100, 100, "rmoveto", 100, 0, 100, 100, 0, 100, "rrcurveto", -100, 0, -100, -100, 0, -100, "rrcurveto", "endchar"],
        [-20, 520, 111, 194, "endchar"],               # o + acute
    ]
    cs = {n: T2CharString(program=list(p)) for n, p in zip(names, progs)}
    cmap = {0x41: "A", 0xB4: "acute", 0xC1: "Aacute", 0xA8: "dieresis", 0xC4: "Adieresis", 0x6F: "o", 0xF3: "oacute"}
    font = _build(names, cmap, cs, {n: 600 for n in names})
    f = TTFont(io.BytesIO(font))
    gs = f.getGlyphSet()
    got, _ = _product_callbacks(vg, font)
    for cp, name in cmap.items():
        rp = DecomposingRecordingPen(gs)
        gs[name].draw(rp)
        want = []
        for op, a in rp.value:
            if op == "moveTo":
                want.append((M, 0, 0, 0, 0) + tuple(float(v) for v in a[0]))
            elif op == "lineTo":
                want.append((L, 0, 0, 0, 0) + tuple(float(v) for v in a[0]))
            elif op == "curveTo":
                want.append((C,) + tuple(float(v) for pt in a for v in pt))
            elif op in ("closePath", "endPath"):
                want.append((Z, 0, 0, 0, 0, 0, 0))
        assert got[cp] == want, name
    assert sum(1 for t in got[0xC1] if t[0] == M) == 2 and sum(1 for t in got[0xC4] if t[0] == M) == 3
    assert _oracle_callbacks(oracle, font) == got


def _oracle_callbacks(oracle, font_bytes):
    f = oracle.Font(font_bytes)
    out = {}
    for cp in f.codepoints():
        seq = f.outline(f.glyph_index(int(cp)))
        out[int(cp)] = [(k,) + ((0.0,) * 6 if k == Z else (x1, y1, x2, y2, x, y)) for k, x1, y1, x2, y2, x, y in seq]
    return out


def test_three_readers_agree(oracle, vg, fira_cff, ops_cff):
    """the oracle's C reader (written on its own from the same technical notes), the product's C++ reader and fontTools
    emit the same callbacks for every mapped glyph of the synthesised fonts"""
    for font in (fira_cff, ops_cff):
        want = _fonttools_callbacks(font)
        got, _ = _product_callbacks(vg, font)
        orc = _oracle_callbacks(oracle, font)
        assert set(orc) == set(want) == set(got)
        for cp in want:
            assert orc[cp] == want[cp] == got[cp], hex(cp)


def test_the_width_operand_exists_once_per_charstring(oracle, vg):
    """ttf-parser's cff1 takes a moveto's extra leading operand as the width only while no width was parsed
    (`stack.len() == N && !width_parsed`); afterwards the stack-length check of the move fails the charstring, BEFORE its
    close(): the callbacks emitted so far stay (renderer.rs:110 ignores outline_glyph's result).  ADVICE r2.
    No reference fixture: parity with the crate unpinned; product and oracle are kept to the crate's published rule."""
    names = [".notdef", "rmove", "hmove", "vmove", "hintw"]
    progs = [[0, "hmoveto", "endchar"],
             [500, 10, 20, "rmoveto", 30, 0, "rlineto", 0, 30, "rlineto", 7, 5, 5, "rmoveto", 10, 10, "rlineto", "endchar"],
             [500, 10, "hmoveto", 30, 40, "rlineto", 7, 5, "hmoveto", 10, 10, "rlineto", "endchar"],
             [500, 10, "vmoveto", 30, 40, "rlineto", 7, 5, "vmoveto", 10, 10, "rlineto", "endchar"],
             [500, 10, 20, "hstem", 10, 20, "rmoveto", 30, 0, "rlineto", 7, 5, 5, "rmoveto", 10, 10, "rlineto", "endchar"]]
    cs = {n: T2CharString(program=list(p)) for n, p in zip(names, progs)}
    font = _build(names, {0x41 + i: n for i, n in enumerate(names[1:])}, cs, {n: 600 for n in names})
    got, _ = _product_callbacks(vg, font)
    z = (0.0,) * 4
    assert got[0x41] == [(M,) + z + (10.0, 20.0), (L,) + z + (40.0, 20.0), (L,) + z + (40.0, 50.0)]
    assert got[0x42] == [(M,) + z + (10.0, 0.0), (L,) + z + (40.0, 40.0)]
    assert got[0x43] == [(M,) + z + (0.0, 10.0), (L,) + z + (30.0, 50.0)]
    assert got[0x44] == [(M,) + z + (10.0, 20.0), (L,) + z + (40.0, 20.0)]
    assert _oracle_callbacks(oracle, font) == got


def test_cff_font_to_pbf_product_equals_oracle(oracle, vg, fira_cff):
    """whole path on a CFF font with the dummy raster (CPU): every PBF file of the product equals the oracle's"""
    mgr = vg.FontManager(True)
    fid = mgr.add_font_data("Fira CFF", fira_cff)
    w = vg.DummyWriter()
    mgr.render_glyphs(w, vg.Renderer.new_dummy())
    font = oracle.Font(fira_cff)
    n_glyphs = 0
    for blk in range(256):
        want, n, _ = oracle.render_block([font], fid, blk * 256, oracle.DUMMY)
        assert w.files[f"{fid}/{blk * 256}-{blk * 256 + 255}.pbf"] == want, blk
        n_glyphs += n
    assert n_glyphs > 250


def test_scan_picks_up_otf_files(vg, tmp_path, fira_cff):
    """recurse.rs:104-133 takes `.ttf` and `.otf`: an OpenType/CFF file in a scanned directory becomes a font named by its
    own name table and renders (index.json lists it)"""
    import json
    (tmp_path / "fonts").mkdir()
    (tmp_path / "fonts" / "Synth CFF - Regular.otf").write_bytes(fira_cff)
    mgr = vg.FontManager(True)
    mgr.scan(tmp_path / "fonts")
    assert mgr.font_ids() == ["synth_cff_regular"]
    w = vg.DummyWriter()
    mgr.render_glyphs(w, vg.Renderer.new_dummy())
    assert len(w.files) == 256
    assert json.loads(mgr.index_json()) == ["synth_cff_regular"]


def test_cid_keyed_font(oracle, vg, ops_cff):
    """the same charstrings behind ROS / FDArray / FDSelect (local subroutines come from the glyph's font dict)"""
    from fontTools.cffLib import FDArrayIndex, FDSelect, FontDict
    f = TTFont(io.BytesIO(ops_cff))
    cff = f["CFF "].cff
    top = cff.topDictIndex[0]
    top.ROS = ("Adobe", "Identity", 0)
    fd = FontDict()
    fd.setCFF2(False)
    fd.Private = top.Private
    fda = FDArrayIndex()
    fda.append(fd)
    top.FDArray = fda
    sel = FDSelect()
    sel.format = 3
    sel.gidArray = [0] * len(f.getGlyphOrder())
    top.FDSelect = sel
    top.rawDict.pop("Private", None)
    del top.Private
    for cs in top.CharStrings.values():
        cs.private = fd.Private
    top.CharStrings.charStringsIndex.fdArray = fda
    top.CharStrings.charStringsIndex.fdSelect = sel
    top.CharStrings.fdArray = fda
    top.CharStrings.fdSelect = sel
    buf = io.BytesIO()
    f.save(buf)
    cid = buf.getvalue()
    check = TTFont(io.BytesIO(cid))
    assert hasattr(check["CFF "].cff.topDictIndex[0], "ROS")
    want = _fonttools_callbacks(ops_cff)
    got, _ = _product_callbacks(vg, cid)
    orc = _oracle_callbacks(oracle, cid)
    for cp in range(0x41, 0x46):
        assert got[cp] == want[cp] == orc[cp], hex(cp)


def test_unreadable_outline_tables_are_refused(vg, fira_cff):
    data = bytearray(fira_cff)
    at = bytes(data).index(b"CFF ")
    data[at:at + 4] = b"CFF2"
    with pytest.raises(RuntimeError, match="CFF2"):
        vg.FontManager(False).add_font_data("Fake CFF2", bytes(data))


_CHILD = r"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(sys.argv[1]); font = Path(sys.argv[2]).read_bytes(); seed = int(sys.argv[3])
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import load_product
from oracle import oracle as O
vg = load_product()
rng = np.random.default_rng(seed)
at = font.index(b"CFF ")
off, ln = int.from_bytes(font[at + 8:at + 12], "big"), int.from_bytes(font[at + 12:at + 16], "big")
r = vg.Renderer.new_dummy()
ok = bad = 0
for i in range(150):
    b = bytearray(font)
    if i:  # (0: control) damage inside the CFF table: header / INDEX offsets / DICTs at its start, charstrings further in
        hi = (64, 600, ln)[i % 3]
        for pos in rng.integers(0, hi, int(rng.integers(1, 16))):
            b[off + int(pos)] = int(rng.integers(0, 256))
    mgr = vg.FontManager(False)
    try:
        fid = mgr.add_font_data(f"Mutant {i}", bytes(b))
    except RuntimeError:
        bad += 1
        continue
    rec = mgr.record_outlines(fid)
    assert len(rec["cmd_off"]) == len(rec["ids"]) + 1
    try:
        mgr.render_glyphs(vg.DummyWriter(), r)
    except RuntimeError:
        pass
    ok += 1
    try:  # the oracle's own reader on the same bytes
        f = O.Font(bytes(b))
    except Exception:
        continue
    for cp in f.codepoints()[:300]:
        f.prepare_glyph(int(cp))
assert ok >= 1
print(f"{ok} loaded, {bad} rejected")
"""


@pytest.mark.parametrize("seed", [1, 2])
def test_damaged_cff_tables_never_crash(tmp_path, fira_cff, seed):
    """random byte damage inside the CFF table (INDEX offsets, DICT operands, charstring programs), in a child
    process whose exit status is checked: an error or missing glyphs, never a crash (as ttf-parser)"""
    import subprocess
    import sys
    from conftest import ROOT
    path = tmp_path / "fira_cff.otf"
    path.write_bytes(fira_cff)
    p = subprocess.run([sys.executable, "-c", _CHILD, str(ROOT), str(path), str(seed)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, f"child died with {p.returncode}\n{p.stdout[-2000:]}\n{p.stderr[-4000:]}"
    assert "loaded" in p.stdout


@pytest.mark.gpu
def test_cff_font_to_pbf_on_the_gpu_equals_oracle(oracle, vg, fira_cff):
    """fonts -> PBF bytes through the HIP renderer (device front-end, cubics flattened on the GPU) = the oracle's files"""
    mgr = vg.FontManager(True)
    fid = mgr.add_font_data("Fira CFF", fira_cff)
    w = vg.DummyWriter()
    mgr.render_glyphs(w, vg.Renderer.new_precise(0))
    font = oracle.Font(fira_cff)
    for blk in range(256):
        want, _, _ = oracle.render_block([font], fid, blk * 256, oracle.PRECISE)
        assert w.files[f"{fid}/{blk * 256}-{blk * 256 + 255}.pbf"] == want, blk


@pytest.mark.gpu
def test_cff_glyphs_render_like_the_oracle(oracle, vg, fira_cff, ops_cff):
    """commands of the CFF reader -> device front-end (cubic flattening included) -> raster, against the oracle's
    RingBuilder + renderer_precise on the same commands"""
    ctx = vg.SdfContext(0)
    for font in (fira_cff, ops_cff):
        _, o = _product_callbacks(vg, font)
        rects, out_bytes, _ = ctx.outlines_prepare(o["cmd_off"], o["cmds"], o["scale"], o["shift_x"])
        out = ctx.outlines_render()
        off = 0
        n_checked = 0
        for g in range(len(o["ids"])):
            cmds = o["cmds"][o["cmd_off"][g]:o["cmd_off"][g + 1]]
            rings = oracle.build_rings([(int(c["kind"]), float(c["x1"]), float(c["y1"]), float(c["x2"]), float(c["y2"]),
                                         float(c["x"]), float(c["y"])) for c in cmds])
            segs = []
            for r in rings:
                p = r * o["scale"][g]
                p[:, 0] += o["shift_x"][g]
                p[:, 1] += 0.0
                segs.append(np.concatenate([p[:-1], p[1:]], axis=1))
            r = rects[g]
            if not r["has_raster"]:
                assert not segs
                continue
            segs = np.concatenate(segs)
            n = int(r["w"]) * int(r["h"])
            want = oracle.sdf_render(segs, int(r["x0"]), int(r["y0"]), int(r["w"]), int(r["h"]))
            assert np.array_equal(out[off:off + n].reshape(want.shape), want), int(o["ids"][g])
            off += n
            n_checked += 1
        assert off == out_bytes and n_checked >= 4
    ctx.close()
