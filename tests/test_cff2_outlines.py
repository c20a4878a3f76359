"""`CFF2` outlines (variable OpenType fonts) through the host reader, at the position the reference draws them.

The reference gets them from ttf-parser (`face.outline_glyph`, src/render/renderer.rs:110) and never sets variation
coordinates, so a variable font is drawn at the default position of its design space.  None of the reference's
fixtures holds a CFF2 font: PARITY WITH THE CRATE IS UNPINNED for this file, and the crate-specific rules (no `fvar`
-> every delta is added; no VariationStore -> no outline; `endchar` / `return` are errors; 513 operands) are restated
from its documented behaviour, see csrc/host/cff.hpp.  What is checked: the product's C++ reader and the oracle's C
reader (written separately) against each other and against fontTools' charstring interpreter, on fonts built here
with fontTools — hand-written programs around `blend` / `vsindex`, and a variable font merged by fontTools.varLib
from two masters made of Fira Sans outlines — and, on the GPU, the PBF files of such a font against the oracle's.
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
from fontTools.ttLib import TTFont  # noqa: E402

from test_cff_outlines import _oracle_callbacks, _product_callbacks  # noqa: E402

M, L, Q, C, Z = 0, 1, 2, 3, 4
AXES = [("wght", 100, 400, 900, "Weight"), ("wdth", 50, 100, 200, "Width")]
# factors at the default position: 0, 0, 0 (a peak off 0 on some axis) and 1 (no axis named: every peak is 0)
REGIONS = [{"wght": (0, 1, 1)}, {"wdth": (0, 1, 1)}, {"wght": (0, 1, 1), "wdth": (0, 1, 1)}, {}]


def _build2(names, programs, regions=REGIONS, axes=AXES, local_subrs=(), global_subrs=(), extra_vardata=(), fvar=True, vstore=True, recalc=True):
    """a CFF2 font from charstring programs; extra_vardata: region index lists of further ItemVariationData subtables"""
    from fontTools.varLib.builder import buildVarData
    fb = FontBuilder(1000, isTTF=False)
    fb.setupGlyphOrder(names)
    fb.setupCharacterMap({0x41 + i: n for i, n in enumerate(names[1:])})
    fb.setupNameTable({"familyName": "Synth CFF2", "styleName": "Regular"})
    fb.setupFvar(axes, [])
    fb.setupCFF2({n: T2CharString(program=list(p)) for n, p in zip(names, programs)}, regions=regions if vstore else None)
    fb.setupHorizontalMetrics({g: (600, 0) for g in names})
    fb.setupHorizontalHeader(ascent=935, descent=-265)
    fb.setupOS2()
    fb.setupPost()
    cff = fb.font["CFF2"].cff
    top = cff.topDictIndex[0]
    for idx in extra_vardata:
        top.VarStore.otVarStore.VarData.append(buildVarData(list(idx), None, optimize=False))
        top.VarStore.otVarStore.VarDataCount = len(top.VarStore.otVarStore.VarData)
    if local_subrs:
        priv = top.FDArray[0].Private
        priv.Subrs = SubrsIndex()
        for prog in local_subrs:
            priv.Subrs.append(T2CharString(program=list(prog)))
    for prog in global_subrs:
        cff.GlobalSubrs.append(T2CharString(program=list(prog)))
    if not fvar:
        del fb.font["fvar"]
    fb.font.recalcBBoxes = recalc   # (off for programs fontTools itself cannot run)
    buf = io.BytesIO()
    fb.save(buf)
    return buf.getvalue()


def _fonttools_callbacks(font_bytes, location=None):
    """callbacks of fontTools' interpreter, without the closePath at the end of the glyph: a CFF2 charstring has no
    `endchar`, the crate emits close() only in front of a further move_to (RingBuilder::into_rings saves the last ring,
    /root/reference/src/render/ring_builder.rs:26-29)"""
    f = TTFont(io.BytesIO(font_bytes))
    if location is None:
        # an explicit default position: without a location fontTools drops every delta, also those of a region that
        # names no axis (factor 1 everywhere, for fontTools' blender and for the crate)
        location = {a.axisTag: a.defaultValue for a in f["fvar"].axes}
    gs, cmap = f.getGlyphSet(location=location), f.getBestCmap()
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
        if seq and seq[-1][0] == Z:
            seq.pop()
        out[cp] = [(t[0],) + tuple(float(np.float32(v)) for v in t[1:]) for t in seq]
    return out


def _blend(values, deltas):
    """operands of one blend: the values, then for each value one delta per region, then the count"""
    assert len(values) == len(deltas)
    return list(values) + [d for ds in deltas for d in ds] + [len(values)]


# blend in front of every kind of operator; four regions -> four deltas per value
_B_MOVE = _blend([100, 200], [(10, 1, 2, 3), (20, 4, 5, 6)]) + ["blend", "rmoveto"]
_B_LINES = (_blend([50], [(5, 6, 7, 8)]) + ["blend", "hlineto"] + [60, "vlineto"]
            + _blend([-50, 10], [(1, 1, 1, -9), (2, 2, 2, 4)]) + ["blend", "rlineto"])
_B_CURVES = (_blend([1, 2, 3, 4, 5, 6], [(1, 0, 0, i) for i in range(6)]) + ["blend", "rrcurveto"]
             + [10, 20, 30] + _blend([40], [(0, 0, 0, 2.5)]) + ["blend", "hhcurveto"]
             + _blend([10, 20, 30, 40, 50, 60, 70], [(9, 9, 9, -i) for i in range(7)]) + ["blend", "hflex"])
_B_SECOND = [-300] + _blend([40], [(1, 2, 3, 4)]) + ["blend", "rmoveto", 20, 20, -20, 20, "rlineto"]
_PROG_BLEND = _B_MOVE + _B_LINES + _B_CURVES + _B_SECOND
# hints: stems with blended edges, a hintmask (2 stems -> 1 mask byte), no width operand anywhere
_PROG_HINTS = (_blend([10, 20], [(1, 1, 1, 1), (2, 2, 2, 2)]) + ["blend", "hstemhm", 5, 10, "hintmask", b"\xc0", 100, 100, "rmoveto",
               50, "hlineto", 50, "vlineto", "cntrmask", b"\x40", -50, "hlineto"])
# subroutines end with their data (no `return`); a blend's operands may come from the caller
_PROG_SUBRS = [100, 200, "rmoveto", -107, "callsubr", -106, "callsubr", 5, 5, -107, "callgsubr", 7, 8, 1, 2, 3, 4, -106, "callgsubr"]
_LOCAL = [[10, 20, "rlineto"], _blend([30], [(1, 2, 3, 4)]) + ["blend", 0, "rlineto", -107, "callgsubr"]]
_GLOBAL = [[-5, 40, "rlineto"], [1, "blend", "rlineto"]]
# the second ItemVariationData (regions 3 and 0: factors 1 and 0) selected with vsindex
_PROG_VSINDEX = [1, "vsindex"] + _blend([100, 200], [(7, 100), (9, 100)]) + ["blend", "rmoveto", 30, 40, "rlineto", -10, 30, "rlineto"]
# 400 operands on the stack (the CFF version 1 limit is 48)
_PROG_DEEP = [0, 0, "rmoveto"] + [((i * 7) % 23) - 11 for i in range(400)] + ["rlineto"]
_NAMES = [".notdef", "blend", "hints", "subrs", "vsindex", "deep", "empty"]
_PROGS = [[0, "hmoveto"], _PROG_BLEND, _PROG_HINTS, _PROG_SUBRS, _PROG_VSINDEX, _PROG_DEEP, []]


@pytest.fixture(scope="module")
def ops_cff2():
    return _build2(_NAMES, _PROGS, local_subrs=_LOCAL, global_subrs=_GLOBAL, extra_vardata=[(3, 0)])


def _variable_fira(n_glyphs=300):
    """two masters (Fira Sans outlines as charstrings; the second one sheared, widened and moved) merged by
    fontTools.varLib into a one-axis CFF2 font: blend operators on most operands, VariationStore, fvar, HVAR"""
    from fontTools import varLib
    from fontTools.designspaceLib import AxisDescriptor, DesignSpaceDocument, SourceDescriptor
    from fontTools.pens.t2CharStringPen import T2CharStringPen
    from fontTools.pens.transformPen import TransformPen
    src = TTFont(FIRA)
    gs = src.getGlyphSet()
    order = src.getGlyphOrder()[:n_glyphs]
    cmap = {cp: g for cp, g in src.getBestCmap().items() if g in order}

    def master(xform, style):
        fb = FontBuilder(src["head"].unitsPerEm, isTTF=False)
        fb.setupGlyphOrder(order)
        fb.setupCharacterMap(cmap)
        cs = {}
        for g in order:
            pen = T2CharStringPen(gs[g].width, gs)
            gs[g].draw(TransformPen(pen, xform))
            cs[g] = pen.getCharString()
        fb.setupCFF("SynthVar-" + style, {"FullName": "Synth Var " + style}, cs, {})
        fb.setupHorizontalMetrics({g: (gs[g].width, 0) for g in order})
        fb.setupHorizontalHeader(ascent=935, descent=-265)
        fb.setupNameTable({"familyName": "Synth Var", "styleName": style})
        fb.setupOS2()
        fb.setupPost()
        buf = io.BytesIO()
        fb.save(buf)
        return TTFont(io.BytesIO(buf.getvalue()))

    ds = DesignSpaceDocument()
    axis = AxisDescriptor()
    axis.tag, axis.name, axis.minimum, axis.default, axis.maximum = "wght", "Weight", 400, 400, 900
    ds.addAxis(axis)
    for font, w in ((master((1, 0, 0, 1, 0, 0), "Regular"), 400), (master((1.25, 0, 0.1, 1.05, 7, -3), "Bold"), 900)):
        s = SourceDescriptor()
        s.font, s.location, s.familyName, s.styleName = font, {"Weight": w}, "Synth Var", "x"
        ds.addSource(s)
    vf, _, _ = varLib.build(ds, optimize=True)
    buf = io.BytesIO()
    vf.save(buf)
    return buf.getvalue()


@pytest.fixture(scope="module")
def fira_cff2():
    return _variable_fira()


def test_blend_operators_at_the_default_position(oracle, vg, ops_cff2):
    want = _fonttools_callbacks(ops_cff2)
    got, _ = _product_callbacks(vg, ops_cff2)
    orc = _oracle_callbacks(oracle, ops_cff2)
    for cp, name in zip(range(0x41, 0x47), _NAMES[1:]):
        assert got[cp] == want[cp], name
        assert orc[cp] == want[cp], name
    z = (0.0,) * 4
    # region 4 has no axis: its factor is 1 and its deltas count — (100 + 3, 200 + 6), then 50 + 8 to the right
    assert got[0x41][:2] == [(M,) + z + (103.0, 206.0), (L,) + z + (161.0, 206.0)]
    assert sum(1 for t in got[0x41] if t[0] == C) == 4 and sum(1 for t in got[0x41] if t[0] == Z) == 1
    # vsindex 1 = regions (3, 0): the first delta of each value counts, the second does not
    assert got[0x44][0] == (M,) + z + (107.0, 209.0)
    assert len(got[0x45]) == 201   # "deep": 400 operands = 200 lines
    assert got[0x46] == []         # no operator at all: no callbacks -> PbfGlyph::empty (renderer.rs:118-120)


def test_variable_fira_at_the_default_position(oracle, vg, fira_cff2):
    f = TTFont(io.BytesIO(fira_cff2))
    assert "CFF2" in f and "fvar" in f and "glyf" not in f and "CFF " not in f
    top = f["CFF2"].cff.topDictIndex[0]
    n_blend = 0
    for name in f.getGlyphOrder():
        cs = top.CharStrings[name]
        cs.decompile()
        n_blend += sum(1 for t in cs.program if t == "blend")
    assert n_blend > 2000   # (the masters differ everywhere: nearly every operator takes blended operands)
    want = _fonttools_callbacks(fira_cff2)
    got, _ = _product_callbacks(vg, fira_cff2)
    orc = _oracle_callbacks(oracle, fira_cff2)
    assert set(got) == set(want) == set(orc) and len(got) > 200
    n_curves = 0
    for cp in want:
        assert got[cp] == want[cp] == orc[cp], hex(cp)
        n_curves += sum(1 for t in got[cp] if t[0] == C)
    assert n_curves > 1500


def test_a_face_without_fvar_has_no_coordinates_so_every_delta_counts(oracle, vg, fira_cff2):
    """ttf-parser evaluates a region over the face's coordinates; a face without `fvar` has none, the product over no
    axes is 1 for every region and `blend` adds every delta.  With one axis that is the outline at the axis' maximum."""
    want = _fonttools_callbacks(fira_cff2, location={"wght": 900})
    f = TTFont(io.BytesIO(fira_cff2))
    del f["fvar"]
    for tag in ("HVAR", "STAT"):
        if tag in f:
            del f[tag]
    buf = io.BytesIO()
    f.save(buf)
    bare = buf.getvalue()
    got, _ = _product_callbacks(vg, bare)
    orc = _oracle_callbacks(oracle, bare)
    differs = 0
    default = _fonttools_callbacks(fira_cff2)
    for cp in want:
        assert got[cp] == want[cp] == orc[cp], hex(cp)
        differs += want[cp] != default[cp]
    assert differs > 200

    # two axes, hand-written: all four regions count
    bare2 = _build2(_NAMES, _PROGS, local_subrs=_LOCAL, global_subrs=_GLOBAL, extra_vardata=[(3, 0)], fvar=False)
    got, _ = _product_callbacks(vg, bare2)
    z = (0.0,) * 4
    assert got[0x41][:2] == [(M,) + z + (100.0 + 16, 200.0 + 35), (L,) + z + (116.0 + 50 + 26, 235.0)]
    assert got[0x44][0] == (M,) + z + (207.0, 309.0)
    assert _oracle_callbacks(oracle, bare2) == got


def test_rules_of_the_crate_for_cff2_charstrings(oracle, vg):
    """operators that CFF2 dropped end the glyph where they stand (callbacks so far stay: renderer.rs:110 ignores the
    result), `vsindex` comes once and before any `blend`, a blend needs its operands, the stack holds 513 numbers"""
    z = (0.0,) * 4
    start = [(M,) + z + (10.0, 20.0), (L,) + z + (40.0, 20.0)]
    names = [".notdef", "endchar", "return", "vs_twice", "vs_late", "vs_none", "short", "full", "over", "mask_end"]
    b1 = _blend([5], [(0, 0, 0, 1)])
    progs = [[0, "hmoveto"],
             [10, 20, "rmoveto", 30, "hlineto", "endchar", 5, 5, "rlineto"],
             [10, 20, "rmoveto", 30, "hlineto", "return", 5, 5, "rlineto"],
             [0, "vsindex", 0, "vsindex", 10, 20, "rmoveto", 30, "hlineto"],
             [10, 20, "rmoveto"] + b1 + ["blend", "hlineto", 1, "vsindex", 5, 5, "rlineto"],
             [7, "vsindex", 10, 20, "rmoveto", 30, "hlineto"],
             [10, 20, "rmoveto", 30, "hlineto", 1, 2, 3, 2, "blend", 5, "rlineto"],              # 2 values need 10 operands
             [10, 20, "rmoveto"] + [1] * 512 + ["rlineto"],
             [10, 20, "rmoveto"] + [1] * 514 + ["rlineto"],
             [10, 20, "rmoveto", 30, "hlineto"] + list(range(1, 41)) + ["hintmask", b"\xff"]]     # 20 stems: two mask bytes are missing
    font = _build2(names, progs, extra_vardata=[(3, 0)], recalc=False)
    got, _ = _product_callbacks(vg, font)
    assert got[0x41] == start and got[0x42] == start
    assert got[0x43] == []                                            # the second vsindex fails before any callback
    assert got[0x44] == [start[0], (L,) + z + (16.0, 20.0)]           # blend ran (5 + 1), then vsindex is refused
    assert got[0x45] == []                                            # ItemVariationData 7 does not exist
    assert got[0x46] == start
    assert len(got[0x47]) == 257 and got[0x48] == [start[0]]          # 512 operands pass, the 514th does not
    assert got[0x49] == start                                         # (no endchar to miss: the glyph simply ends)
    assert _oracle_callbacks(oracle, font) == got


def test_without_a_variation_store_there_is_no_outline(oracle, vg):
    """the crate loads the scalars of ItemVariationData 0 before the first operator; a CFF2 table without a VariationStore
    has no such subtable, the glyph fails before any callback — static CFF2 fonts render as empty glyphs in the reference
    (restated from the crate's source as documented; no fixture: parity unpinned)"""
    names = [".notdef", "box"]
    progs = [[0, "hmoveto"], [10, 20, "rmoveto", 30, "hlineto", 30, "vlineto", -30, "hlineto"]]
    with_store = _build2(names, progs)
    got, _ = _product_callbacks(vg, with_store)
    assert len(got[0x41]) == 4
    static = _build2(names, progs, vstore=False, fvar=False)
    assert "VarStore" not in TTFont(io.BytesIO(static))["CFF2"].cff.topDictIndex[0].rawDict
    got, _ = _product_callbacks(vg, static)
    assert got[0x41] == []
    assert _oracle_callbacks(oracle, static) == got


def test_cff2_font_to_pbf_product_equals_oracle(oracle, vg, fira_cff2):
    """whole path on a variable font with the dummy raster (CPU): every PBF file of the product equals the oracle's"""
    mgr = vg.FontManager(True)
    fid = mgr.add_font_data("Fira CFF2", fira_cff2)
    w = vg.DummyWriter()
    mgr.render_glyphs(w, vg.Renderer.new_dummy())
    font = oracle.Font(fira_cff2)
    n_glyphs = 0
    for blk in range(256):
        want, n, _ = oracle.render_block([font], fid, blk * 256, oracle.DUMMY)
        assert w.files[f"{fid}/{blk * 256}-{blk * 256 + 255}.pbf"] == want, blk
        n_glyphs += n
    assert n_glyphs > 200


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
at = font.index(b"CFF2")
off, ln = int.from_bytes(font[at + 8:at + 12], "big"), int.from_bytes(font[at + 12:at + 16], "big")
r = vg.Renderer.new_dummy()
ok = bad = same = 0
for i in range(150):
    b = bytearray(font)
    if i:  # (0: control) damage inside the CFF2 table: header / Top DICT / INDEX offsets at its start, charstrings and the store further in
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
def test_damaged_cff2_tables_never_crash(tmp_path, fira_cff2, seed):
    """random byte damage inside the CFF2 table (Top DICT, INDEX offsets, VariationStore, charstring programs), in a
    child process whose exit status is checked: an error or missing glyphs, never a crash (as ttf-parser)"""
    import subprocess
    import sys
    from conftest import ROOT
    path = tmp_path / "fira_cff2.otf"
    path.write_bytes(fira_cff2)
    p = subprocess.run([sys.executable, "-c", _CHILD, str(ROOT), str(path), str(seed)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, f"child died with {p.returncode}\n{p.stdout[-2000:]}\n{p.stderr[-4000:]}"
    assert "loaded" in p.stdout


@pytest.mark.gpu
def test_cff2_font_to_pbf_on_the_gpu_equals_oracle(oracle, vg, fira_cff2):
    """variable font -> PBF bytes through the HIP renderer (device front-end, cubics flattened on the GPU) = the oracle's files"""
    mgr = vg.FontManager(True)
    fid = mgr.add_font_data("Fira CFF2", fira_cff2)
    w = vg.DummyWriter()
    mgr.render_glyphs(w, vg.Renderer.new_precise(0))
    font = oracle.Font(fira_cff2)
    n_glyphs = 0
    for blk in range(256):
        want, n, _ = oracle.render_block([font], fid, blk * 256, oracle.PRECISE)
        assert w.files[f"{fid}/{blk * 256}-{blk * 256 + 255}.pbf"] == want, blk
        n_glyphs += n
    assert n_glyphs > 200
