"""Pins the CPU oracle to every known-answer test the reference holds for the render path
(SURVEY.md §4 / §8c).  Expected values below are DATA transcribed from the reference's own
test assertions; each test cites where.  No GPU."""
import numpy as np
import pytest

from conftest import FIRA, NOTO, noto_files


def digit_art(bm):
    """utils/decode_bitmap.rs:15-28"""
    return [" ".join("%02d" % min(int(v) * 100 // 256, 99) for v in row) for row in bm]


def ascii_art(bm):
    """utils/decode_bitmap.rs:60-78"""
    def sym(v):
        if v <= 60:
            return "  "
        if v <= 120:
            return "░░"
        if v <= 180:
            return "▒▒"
        if v <= 240:
            return "▓▓"
        return "█"
    return ["".join(sym(int(v)) for v in row) for row in bm]


# ---- renderer_precise.rs:96-135 test_render_sdf_simple_square ---------------------------------
SQUARE_ART = [
    "30 38 42 43 43 43 43 42 38 30",
    "38 48 54 55 55 55 55 54 48 38",
    "42 54 65 68 68 68 68 65 54 42",
    "43 55 68 80 80 80 80 68 55 43",
    "43 55 68 80 93 93 80 68 55 43",
    "43 55 68 80 93 93 80 68 55 43",
    "43 55 68 80 80 80 80 68 55 43",
    "42 54 65 68 68 68 68 65 54 42",
    "38 48 54 55 55 55 55 54 48 38",
    "30 38 42 43 43 43 43 42 38 30",
]


@pytest.mark.parametrize("mode", ["PRECISE", "BRUTE"])
def test_square_digit_art(oracle, mode):
    segs = np.array([[1, 2, 5, 2], [5, 2, 5, 6], [5, 6, 1, 6], [1, 6, 1, 2]], dtype=np.float64)
    bm = oracle.sdf_render(segs, -2, -1, 10, 10, getattr(oracle, mode))
    assert bm.shape == (10, 10)
    assert digit_art(bm) == SQUARE_ART


# ---- renderer.rs:176-287 test_render_glyph_32/65/230/96 ---------------------------------------
ART_65 = [
    "            ░░░░░░░░░░░░░░░░            ",
    "          ░░░░▒▒▒▒▒▒▒▒▒▒░░░░░░          ",
    "        ░░░░▒▒▒▒▒▒▒▒▒▒▒▒▒▒░░░░          ",
    "        ░░░░▒▒▒▒▓▓▓▓▓▓▓▓▒▒▒▒░░░░        ",
    "        ░░░░▒▒▒▒▓▓▓▓▓▓▓▓▒▒▒▒░░░░        ",
    "      ░░░░▒▒▒▒▓▓▓▓▓▓▓▓▓▓▒▒▒▒░░░░        ",
    "      ░░░░▒▒▒▒▓▓▓▓▓▓▓▓▓▓▓▓▒▒▒▒░░░░      ",
    "      ░░░░▒▒▒▒▓▓▓▓▓▓▓▓▓▓▓▓▒▒▒▒░░░░      ",
    "      ░░▒▒▒▒▓▓▓▓▓▓▒▒▓▓▓▓▓▓▒▒▒▒░░░░      ",
    "    ░░░░▒▒▒▒▓▓▓▓▓▓▒▒▒▒▓▓▓▓▓▓▒▒▒▒░░░░    ",
    "    ░░░░▒▒▒▒▓▓▓▓▓▓▒▒▒▒▓▓▓▓▓▓▒▒▒▒░░░░    ",
    "    ░░░░▒▒▓▓▓▓▓▓▒▒▒▒▒▒▓▓▓▓▓▓▒▒▒▒░░░░    ",
    "  ░░░░▒▒▒▒▓▓▓▓▓▓▒▒▒▒▒▒▒▒▓▓▓▓▓▓▒▒░░░░    ",
    "  ░░░░▒▒▒▒▓▓▓▓▓▓▒▒▒▒▒▒▒▒▓▓▓▓▓▓▒▒▒▒░░░░  ",
    "  ░░░░▒▒▒▒▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▒▒▒▒░░░░  ",
    "░░░░▒▒▒▒▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▒▒▒▒░░░░  ",
    "░░░░▒▒▒▒▓▓▓▓▓▓▒▒▒▒▒▒▒▒▒▒▒▒▓▓▓▓▓▓▒▒▒▒░░░░",
    "░░░░▒▒▒▒▓▓▓▓▓▓▒▒▒▒▒▒▒▒▒▒▒▒▓▓▓▓▓▓▒▒▒▒░░░░",
    "░░▒▒▒▒▓▓▓▓▓▓▒▒▒▒░░░░░░▒▒▒▒▓▓▓▓▓▓▒▒▒▒░░░░",
    "░░▒▒▒▒▓▓▓▓▓▓▒▒▒▒░░░░░░░░▒▒▒▒▓▓▓▓▓▓▒▒▒▒░░",
    "░░▒▒▒▒▒▒▒▒▒▒▒▒▒▒░░░░░░░░▒▒▒▒▒▒▒▒▒▒▒▒▒▒░░",
    "░░▒▒▒▒▒▒▒▒▒▒▒▒░░░░  ░░░░░░▒▒▒▒▒▒▒▒▒▒░░░░",
    "░░░░░░░░░░░░░░░░░░    ░░░░░░░░░░░░░░░░░░",
]

ART_230 = [
    "      ░░░░░░░░░░░░░░░░░░░░░░░░░░░░░░░░░░░░░░      ",
    "    ░░░░░░▒▒▒▒▒▒▒▒▒▒▒▒▒▒░░░░▒▒▒▒▒▒▒▒▒▒▒▒░░░░░░    ",
    "  ░░░░▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒░░░░  ",
    "  ░░░░▒▒▒▒▓▓▓▓▓▓▓▓▓▓▓▓▓▓▒▒▒▒▓▓▓▓▓▓▓▓▓▓▓▓▒▒▒▒░░░░░░",
    "  ░░░░▒▒▒▒▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▒▒▒▒░░░░",
    "  ░░░░▒▒▒▒▓▓▓▓▒▒▒▒▒▒▓▓▓▓▓▓▓▓▓▓▒▒▒▒▒▒▓▓▓▓▓▓▓▓▒▒▒▒░░",
    "  ░░░░▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒▓▓▓▓▓▓▓▓▒▒▒▒▒▒▒▒▓▓▓▓▓▓▒▒▒▒░░",
    "  ░░░░░░▒▒▒▒▒▒▒▒▒▒▒▒▒▒▓▓▓▓▓▓▒▒▒▒▒▒▒▒▒▒▓▓▓▓▓▓▒▒▒▒░░",
    "  ░░░░▒▒▒▒▒▒▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▒▒▒▒░░",
    "░░░░▒▒▒▒▒▒▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▒▒▒▒░░",
    "░░░░▒▒▒▒▓▓▓▓▓▓▓▓▒▒▒▒▒▒▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▒▒▒▒░░",
    "░░░░▒▒▒▒▓▓▓▓▓▓▒▒▒▒▒▒▒▒▓▓▓▓▓▓▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒░░",
    "░░░░▒▒▒▒▓▓▓▓▓▓▒▒▒▒▒▒▒▒▓▓▓▓▓▓▓▓▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒░░░░",
    "░░░░▒▒▒▒▓▓▓▓▓▓▒▒▒▒▒▒▒▒▓▓▓▓▓▓▓▓▒▒▒▒▒▒▒▒▒▒▓▓▒▒▒▒░░░░",
    "░░░░▒▒▒▒▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▒▒▒▒░░",
    "░░░░▒▒▒▒▒▒▓▓▓▓▓▓▓▓▓▓▓▓▓▓▒▒▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▓▒▒▒▒▒▒░░",
    "  ░░░░▒▒▒▒▒▒▒▒▓▓▓▓▒▒▒▒▒▒▒▒▒▒▒▒▒▒▓▓▓▓▒▒▒▒▒▒▒▒▒▒░░░░",
    "    ░░░░▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒░░▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒▒░░░░░░  ",
    "      ░░░░░░░░░░░░░░░░░░░░░░░░░░░░░░░░░░░░░░░░    ",
    "        ░░░░░░░░░░░░░░░░  ░░░░░░░░░░░░░░░░        ",
]

ART_96 = [
    "    ░░░░░░░░░░            ",
    "  ░░░░░░░░░░░░░░░░        ",
    "  ░░░░▒▒▒▒▒▒▒▒░░░░░░░░    ",
    "░░░░▒▒▒▒▒▒▒▒▒▒▒▒▒▒░░░░░░  ",
    "░░░░▒▒▒▒▓▓▓▓▓▓▒▒▒▒▒▒░░░░░░",
    "░░░░▒▒▓▓▓▓▓▓▓▓▓▓▒▒▒▒▒▒▒▒░░",
    "░░░░▒▒▒▒▒▒▓▓▓▓▓▓▓▓▓▓▒▒▒▒░░",
    "░░░░░░▒▒▒▒▒▒▒▒▒▒▓▓▒▒▒▒▒▒░░",
    "  ░░░░░░░░▒▒▒▒▒▒▒▒▒▒▒▒░░░░",
    "      ░░░░░░░░▒▒▒▒▒▒░░░░░░",
    "          ░░░░░░░░░░░░░░  ",
]

GLYPH_KATS = {  # cp: (width, height, left, top, advance), art
    65: ((14, 17, 0, -7, 13), ART_65),
    230: ((19, 14, 0, -11, 19), ART_230),
    96: ((7, 5, 0, -4, 7), ART_96),
}


def test_render_glyph_32(fira_oracle):
    info, bm = fira_oracle.render_glyph(32)
    assert info.metrics() == (0, 0, 0, 0, 6)
    assert bm is None


@pytest.mark.parametrize("cp", sorted(GLYPH_KATS))
@pytest.mark.parametrize("mode", ["PRECISE", "BRUTE"])
def test_render_glyph_art(oracle, fira_oracle, cp, mode):
    metrics, art = GLYPH_KATS[cp]
    info, bm = fira_oracle.render_glyph(cp, getattr(oracle, mode))
    assert info.metrics() == metrics
    assert bm.size == (info.width + 6) * (info.height + 6)  # renderer.rs:165
    assert ascii_art(bm) == art


# ---- recurse.rs:341-367 / merge.rs:158-184: PBF byte sizes under the dummy renderer ------------
FIRA_PBF_SIZES = {
    0: 80022, 256: 130750, 512: 92634, 768: 63760, 1024: 118037, 1280: 26296, 3584: 592,
    7424: 7260, 7680: 87078, 7936: 124520, 8192: 20301, 8448: 17395, 8704: 6511, 8960: 4375,
    9472: 853, 11264: 3579, 42752: 5761, 43776: 487, 64256: 1032, 65024: 50,
}


def test_fira_pbf_sizes_dummy(oracle, fira_oracle):
    for b in range(256):
        pbf, _, _ = oracle.render_block([fira_oracle], "fira_sans_regular", b * 256, oracle.DUMMY)
        if b * 256 in FIRA_PBF_SIZES:
            assert len(pbf) == FIRA_PBF_SIZES[b * 256], b * 256
        else:
            assert 32 <= len(pbf) <= 34, b * 256  # recurse.rs:204-206 filter


def test_fira_pbf_sizes_precise_sample(oracle, fira_oracle):
    # sizes do not depend on bitmap content; check three blocks with the real renderer
    for start in (3584, 9472, 65024):
        pbf, _, _ = oracle.render_block([fira_oracle], "fira_sans_regular", start, oracle.PRECISE)
        assert len(pbf) == FIRA_PBF_SIZES[start]


# ---- wrapper.rs:197-221 test_get_blocks (glyph counts per block) ------------------------------
FIRA_BLOCK_COUNTS = {
    0: 192, 256: 256, 512: 219, 768: 177, 1024: 240, 1280: 48, 3584: 1, 7424: 20, 7680: 157,
    7936: 233, 8192: 67, 8448: 28, 8704: 16, 8960: 5, 9472: 2, 11264: 7, 42752: 14, 43776: 1,
    64256: 2, 65024: 1,
}


def test_fira_block_counts(fira_oracle):
    cps = fira_oracle.codepoints()
    assert len(cps) == 1686  # metadata.rs:142
    counts = {}
    for cp in cps[cps <= 0xFFFF]:
        counts[int(cp) // 256 * 256] = counts.get(int(cp) // 256 * 256, 0) + 1
    assert counts == FIRA_BLOCK_COUNTS


def test_codepoint_counts(fira_oracle, noto_oracle):
    assert len(fira_oracle.codepoints()) == 1686  # metadata.rs:142
    assert len(noto_oracle.codepoints()) == 3094  # metadata.rs:152
    assert fira_oracle.num_glyphs == 2677  # file_entry.rs:69


def encode_codeblocks(cps):
    """index_files.rs:65-101: hex ranges of 16-code-point blocks (pins cmap coverage)"""
    blocks = sorted({int(c) >> 4 for c in cps})
    ranges, start, prev = [], blocks[0], blocks[0]
    for b in blocks[1:]:
        if b != prev + 1:
            ranges.append((start, prev))
            start = b
        prev = b
    ranges.append((start, prev))
    return ",".join("%X" % s if s == e else "%X-%X" % (s, e) for s, e in ranges)


# index_files.rs:193,205 test_build_font_families_json
FIRA_CODEBLOCKS = ("0,2-7,A-2E,30-52,E3,1D4,1D6-1D7,1D9,1DB-1DC,1E0-204,207-208,20A-20B,210-212,215,219,"
                   "21E,220-222,224,226,22C,232,23C,25A,25C,2C6-2C7,A78,A7A-A7B,AB5,FB0,FEF")
NOTO_CODEBLOCKS = ("0,2-7,A-52,90-97,10F,1AB-1AC,1C8,1D0-20C,20F-215,218,221,25C,2C6-2C7,2DE-2E5,A64-A69,"
                   "A70-A7D,A7F,A8F,A92,AB3-AB6,FB0,FE0,FE2,FEF,FFF,1078-107B,1DF0-1DF1")


def test_codeblocks(fira_oracle, noto_oracle):
    assert encode_codeblocks(fira_oracle.codepoints()) == FIRA_CODEBLOCKS
    assert encode_codeblocks(noto_oracle.codepoints()) == NOTO_CODEBLOCKS


# ---- ring_builder.rs:197-229 flattening KATs --------------------------------------------------
def test_quad_flatten_17_points(oracle):
    M, Q, C = oracle.MOVE, oracle.QUAD, oracle.CURVE
    rings = oracle.build_rings([(M, 0, 0, 0, 0, 0, 0), (Q, 10, 10, 0, 0, 20, 0)])
    assert len(rings) == 1
    pts = rings[0]
    assert len(pts) == 18  # 17 points + the closing copy of the first (ring.rs:53-63)
    assert tuple(pts[16]) == (20.0, 0.0)
    assert tuple(pts[17]) == (0.0, 0.0)
    rings = oracle.build_rings([(M, 0, 0, 0, 0, 0, 0), (C, 10, 10, 20, 10, 30, 0)])
    pts = rings[0]
    assert len(pts) == 18
    assert tuple(pts[16]) == (30.0, 0.0)


def test_ring_builder_rules(oracle):
    M, L, Q, CL = oracle.MOVE, oracle.LINE, oracle.QUAD, oracle.CLOSE
    # ring_builder.rs:139-149: a lone move_to is dropped
    assert oracle.build_rings([(M, 0, 0, 0, 0, 10, 20), (M, 0, 0, 0, 0, 30, 40)]) == []
    # ring_builder.rs:151-185: 3 points + close -> 4 points, 3 segments
    r = oracle.build_rings([(M, 0, 0, 0, 0, 0, 0), (L, 0, 0, 0, 0, 1, 2), (L, 0, 0, 0, 0, -1, 3), (CL,) + (0,) * 6])
    assert len(r) == 1 and [tuple(p) for p in r[0]] == [(0, 0), (1, 2), (-1, 3), (0, 0)]
    # ring_builder.rs:187-195: quad_to on an empty ring is ignored
    assert oracle.build_rings([(Q, 10, 10, 0, 0, 20, 20)]) == []
    # ring_builder.rs:256-271: into_rings saves the trailing ring
    r = oracle.build_rings([(M, 0, 0, 0, 0, 0, 0), (L, 0, 0, 0, 0, 1, 0), (L, 0, 0, 0, 0, 0, 2)])
    assert len(r) == 1 and len(r[0]) == 4
    # already-closed contour: no extra point, 3 points total -> dropped (<4)
    assert oracle.build_rings([(M, 0, 0, 0, 0, 0, 0), (L, 0, 0, 0, 0, 1, 0), (L, 0, 0, 0, 0, 0, 0), (CL,) + (0,) * 6]) == []


# ---- segment.rs:118-199 / rtree_segments.rs:94-198 --------------------------------------------
def test_distance_semantics(oracle):
    def d(seg, p, mode):
        # one pixel whose sample point is p: x0+0.5 = px
        x0, y0 = int(np.floor(p[0] - 0.5)), int(np.floor(p[1] - 0.5))
        assert x0 + 0.5 == p[0] and y0 + 0.5 == p[1]
        return int(oracle.sdf_render(np.array([seg], dtype=float), x0, y0, 1, 1, mode)[0, 0])
    # outside, distance 1 -> 255 - (32+64) = 159
    assert d([0, 0, 4, 0], (2.5, 1.5), oracle.BRUTE) == round(255 - (1.5 * 32 + 64))
    # no candidate within the ±8 box -> inf -> 0 (rtree_segments.rs:122-139)
    assert d([0, 0, 4, 0], (100.5, 100.5), oracle.PRECISE) == 0
    assert d([0, 0, 4, 0], (100.5, 100.5), oracle.BRUTE) == 0
    # degenerate segment (segment.rs:131-141)
    assert d([2, 3, 2, 3], (2.5, 3.5), oracle.BRUTE) == round(255 - (np.sqrt(0.5) * 32 + 64))


# ---- family merge: counts from SURVEY §8(d) config 3 (probe values; canonical order) ----------
def test_noto_all_first_provider_wins(oracle):
    fonts = [oracle.Font(p) for p in noto_files()]
    assert noto_files()[0].name == "Noto Sans - Regular.ttf"
    seen = set()
    for f in fonts:
        seen.update(int(c) for c in f.codepoints() if c <= 0xFFFF)
    assert len(seen) == 6480
    assert len({c // 256 for c in seen}) == 45
