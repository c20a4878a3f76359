"""Robustness of the two TrueType readers (the product's C++ `vg::Face`, the oracle's C reader)
against damaged input: truncated files and random byte flips of a real font.  ttf-parser, which
the reference uses (`file_entry.rs:48`), answers such input with an error or with missing glyphs,
never with a crash; here every mutated font is pushed through parse -> cmap -> outline -> flatten ->
bbox (+ PBF encode with the dummy raster) in a CHILD process whose exit status is checked."""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent

CHILD = r"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(sys.argv[1]); which = sys.argv[2]; seed = int(sys.argv[3])
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import load_product, FIRA
data = Path(FIRA).read_bytes()
rng = np.random.default_rng(seed)
def mutants():
    yield data  # control
    for cut in (0, 3, 11, 12, 100, 300, 1000, 5000, 20000, len(data) // 2, len(data) - 1):
        yield data[:cut]
    # damage concentrated in the table directory / head / maxp / cmap / loca region and spread over the file
    for _ in range(60):
        b = bytearray(data)
        hi = (400, 4000, 60000, len(b))[int(rng.integers(0, 4))]
        for pos in rng.integers(0, hi, int(rng.integers(1, 24))):
            b[int(pos)] = int(rng.integers(0, 256))
        yield bytes(b)
ok = bad = 0
if which == "product":
    vg = load_product()
    r = vg.Renderer.new_dummy()
    for i, m in enumerate(mutants()):
        mgr = vg.FontManager(False)
        try:
            fid = mgr.add_font_data(f"Mutant {i}", m)
        except RuntimeError:
            bad += 1
            continue
        rec = mgr.record_outlines(fid)
        assert len(rec["cmd_off"]) == len(rec["ids"]) + 1
        try:   # the same glyphs for the device's decoder: nothing it lists lies outside its byte store
            parts = mgr.record_glyf_parts(fid)
        except RuntimeError:   # (the damage hit the table directory: a font without glyf outlines has no parts)
            parts = None
        if parts is not None:
            assert list(parts["ids"]) == list(rec["ids"]) and len(parts["bytes"]) % 4 == 0
            assert (parts["parts"]["byte_off"].astype("int64") + parts["parts"]["byte_len"] <= len(parts["bytes"])).all()
            assert int(parts["cmd_off"][-1]) == int(parts["parts"]["cmd_cap"].sum())
        w = vg.DummyWriter()
        try:
            mgr.render_glyphs(w, r)
        except RuntimeError:
            pass
        ok += 1
else:
    from oracle import oracle as O
    import tempfile, os
    for i, m in enumerate(mutants()):
        with tempfile.NamedTemporaryFile(suffix=".ttf", delete=False) as fh:
            fh.write(m)
        try:
            try:
                f = O.Font(fh.name)
            except Exception:
                bad += 1
                continue
            for cp in f.codepoints()[:400]:
                f.prepare_glyph(int(cp))
            ok += 1
        finally:
            os.unlink(fh.name)
assert ok >= 1, "the undamaged control font must load"
print(f"{which}: {ok} loaded, {bad} rejected")
"""


@pytest.mark.parametrize("which", ["product", "oracle"])
@pytest.mark.parametrize("seed", [1, 2])
def test_damaged_fonts_never_crash(which, seed):
    p = subprocess.run([sys.executable, "-c", CHILD, str(ROOT), which, str(seed)], capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, f"child died with {p.returncode}\n{p.stdout[-2000:]}\n{p.stderr[-4000:]}"
    assert "loaded" in p.stdout


def _retag(data: bytes, old: bytes, new: bytes) -> bytes:
    """rename a table in the sfnt directory (the table's bytes stay where they are)"""
    n = int.from_bytes(data[4:6], "big")
    b = bytearray(data)
    for i in range(n):
        rec = 12 + 16 * i
        if bytes(b[rec:rec + 4]) == old:
            b[rec:rec + 4] = new
            return bytes(b)
    raise AssertionError(f"table {old!r} not found")


def test_cff_and_cmapless_fonts_are_refused(vg):
    """A font whose outlines live in a table this reader cannot walk (here: a `CFF ` table that does not parse;
    a version-1 table under the `CFF2` tag in tests/test_cff_outlines.py) must not silently render as empty glyphs, and a font without a cmap
    table fails as in the reference ("Font has no cmap table", src/font/metadata.rs:104-107)."""
    from conftest import FIRA
    data = Path(FIRA).read_bytes()
    cff = _retag(_retag(data, b"glyf", b"CFF "), b"loca", b"xxxx")
    with pytest.raises(RuntimeError, match="CFF"):
        vg.FontManager(False).add_font_data("Fake CFF", cff)
    with pytest.raises(RuntimeError, match="no cmap"):
        vg.FontManager(False).add_font_data("No cmap", _retag(data, b"cmap", b"xmap"))
    # control: the untouched font loads
    vg.FontManager(False).add_font_data("Fira", data)
