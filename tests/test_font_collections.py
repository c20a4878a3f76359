"""Font collections and the sfnt magic.  The reference hands a file's bytes to `Face::parse(data, 0)` (ttf-parser; call site
/root/reference/src/font/file_entry.rs): face 0 — the file itself, or the FIRST face of a font collection ('ttcf') — and
refuses any other magic than 0x00010000 / 'true' / 'OTTO'.  No fixture of the reference holds a collection: parity with the
crate unpinned, product and oracle are kept to its documented rule; the collection is built here with fontTools."""
import io

import pytest

from conftest import FIRA, NOTO

fontTools = pytest.importorskip("fontTools")


def _collection(paths):
    from fontTools.ttLib import TTCollection, TTFont
    ttc = TTCollection()
    ttc.fonts = [TTFont(p) for p in paths]
    buf = io.BytesIO()
    ttc.save(buf)
    return buf.getvalue()


def _render(vg, name, data):
    mgr = vg.FontManager(True)
    fid = mgr.add_font_data(name, data)
    w = vg.DummyWriter()
    mgr.render_glyphs(w, vg.Renderer.new_dummy())
    return fid, w.files, mgr


def test_a_collection_renders_as_its_first_face(oracle, vg):
    from pathlib import Path
    ttc = _collection([FIRA, NOTO])
    assert ttc[:4] == b"ttcf"
    fid_c, files_c, mgr = _render(vg, "Some Font", ttc)
    fid_f, files_f, _ = _render(vg, "Some Font", Path(FIRA).read_bytes())
    assert fid_c == fid_f and files_c == files_f and len(files_c) == 256
    o = mgr.record_outlines(fid_c)
    assert len(o["ids"]) == 1686                      # Fira's glyphs, not Noto's
    # the oracle's reader takes the same face
    fo = oracle.Font(ttc)
    want, n, _ = oracle.render_block([fo], fid_c, 0, oracle.DUMMY)
    assert files_c[f"{fid_c}/0-255.pbf"] == want and n > 150
    # the other way round: Noto first
    _, files_n, mgr_n = _render(vg, "Some Font", _collection([NOTO, FIRA]))
    assert files_n != files_f and len(mgr_n.record_outlines("some_font")["ids"]) == 3006


def test_unknown_magic_is_not_a_font(oracle, vg):
    from pathlib import Path
    data = bytearray(Path(FIRA).read_bytes())
    data[0:4] = b"wOFF"          # a valid table directory behind a magic ttf-parser does not take
    with pytest.raises(RuntimeError):
        vg.FontManager(False).add_font_data("Woff", bytes(data))
    with pytest.raises(Exception):
        oracle.Font(bytes(data))
    # a collection whose first face is a collection again, or that lists no face
    ttc = bytearray(_collection([FIRA]))
    at = int.from_bytes(ttc[12:16], "big")
    bad = bytearray(ttc)
    bad[at:at + 4] = b"ttcf"
    with pytest.raises(RuntimeError):
        vg.FontManager(False).add_font_data("Nested", bytes(bad))
    empty = bytearray(ttc)
    empty[8:12] = (0).to_bytes(4, "big")
    with pytest.raises(RuntimeError):
        vg.FontManager(False).add_font_data("Empty", bytes(empty))


def test_collection_offset_array_is_read_whole(oracle, vg):
    """ttf-parser's RawFace::parse reads all numFonts offsets (out of bounds: no face) and requires the face behind the
    array (`face_offset.checked_sub(s.offset())`); round-3 advice: this reader only looked at offsets[0]."""
    ttc = bytearray(_collection([FIRA, NOTO]))
    assert int.from_bytes(ttc[8:12], "big") == 2
    # numFonts claims more offsets than the file holds
    trunc = bytearray(ttc)
    trunc[8:12] = (len(ttc) // 4 + 10).to_bytes(4, "big")
    for bad in (trunc,):
        with pytest.raises(RuntimeError):
            vg.FontManager(False).add_font_data("Truncated", bytes(bad))
        with pytest.raises(Exception):
            oracle.Font(bytes(bad))
    # a valid face placed INSIDE the offset array's extent: numFonts = 6 -> the array ends at 36; the face sits at 16
    face = Path_read(FIRA)
    inside = bytearray(b"ttcf" + (0x00010000).to_bytes(4, "big") + (6).to_bytes(4, "big") + (16).to_bytes(4, "big")) + bytearray(face)
    # (tables of the embedded face are addressed from the start of the file: shift them)
    n_tables = int.from_bytes(face[4:6], "big")
    for i in range(n_tables):
        rec = 16 + 12 + 16 * i
        off = int.from_bytes(inside[rec + 8:rec + 12], "big") + 16
        inside[rec + 8:rec + 12] = off.to_bytes(4, "big")
    with pytest.raises(RuntimeError):
        vg.FontManager(False).add_font_data("Inside", bytes(inside))
    with pytest.raises(Exception):
        oracle.Font(bytes(inside))
    # the same bytes with numFonts = 1 (array ends at 16 = the face's offset) load: the shifted directory is sound
    ok = bytearray(inside)
    ok[8:12] = (1).to_bytes(4, "big")
    fid, files, _ = _render(vg, "Some Font", bytes(ok))
    assert len(files) == 256
    assert oracle.Font(bytes(ok)) is not None


def Path_read(p):
    from pathlib import Path
    return Path(p).read_bytes()
