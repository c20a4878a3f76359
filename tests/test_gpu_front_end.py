"""GPU: the device front-end (outline commands -> flattened rings -> scaled segments + rects,
csrc/outline_kernels.hip) against the oracle, and the whole pipeline through it against the
golden PBF hashes.  Commands are what ttf-parser's OutlineBuilder receives (taken from the
oracle's TTF reader here; the product's own reader is pinned to it in test_host_vs_oracle)."""
import hashlib
import json

import numpy as np
import pytest

from conftest import FIRA, GOLDEN, NOTO, NOTO_DIR
from test_golden_cpu import set_paths

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(vg):
    c = vg.SdfContext(0)
    yield c
    c.close()


def record(vg, font, cps):
    """host half of render_glyph up to the outline callbacks (renderer.rs:104-116,130)"""
    cmds, cmd_off, scale, shift, ids = [], [0], [], [], []
    sc = 24.0 / font.units_per_em
    for cp in cps:
        gid = font.glyph_index(int(cp))
        if gid is None:
            continue
        for (kind, x1, y1, x2, y2, x, y) in font.outline(gid):
            cmds.append((x1, y1, x2, y2, x, y, kind))
        adv = (font.hor_advance(gid) or 0) * sc * 0.95
        a = np.floor(adv + 0.5)  # round half away (adv >= 0)
        cmd_off.append(len(cmds))
        scale.append(sc)
        shift.append((a - adv) / 2.0)
        ids.append(int(cp))
    return (np.array(cmd_off, np.uint32), np.array(cmds, dtype=vg.OUTLINE_CMD_DTYPE), np.array(scale), np.array(shift), ids)


@pytest.mark.parametrize("path", [FIRA, NOTO, NOTO_DIR / "Noto Sans Arabic - Regular.ttf",
                                  NOTO_DIR / "Noto Sans Myanmar - Regular.ttf"], ids=lambda p: p.stem[:18])
def test_segments_and_rects_match_oracle(oracle, vg, ctx, path):
    font = oracle.Font(path)
    cps = font.codepoints()
    cmd_off, cmds, scale, shift, ids = record(vg, font, cps[cps <= 0xFFFF])
    rects, out_bytes, n_segs = ctx.outlines_prepare(cmd_off, cmds, scale, shift)
    seg_off, segs = ctx.outlines_segments()
    assert len(rects) == len(ids)
    tot_px = 0
    for g, cp in enumerate(ids):
        info, want = font.prepare_glyph(cp)
        r = rects[g]
        assert bool(r["has_raster"]) == bool(info.has_bitmap), cp
        if not info.has_bitmap:
            assert seg_off[g + 1] == seg_off[g]
            continue
        assert (int(r["x0"]), int(r["y0"]), int(r["w"]), int(r["h"]), int(r["n_segments"])) == \
            (info.x0, info.y0, info.w, info.h, info.n_segments), cp
        got = segs[seg_off[g]:seg_off[g + 1]]
        assert got.tobytes() == want.tobytes(), f"segments differ for U+{cp:04X}"
        tot_px += info.w * info.h
    assert out_bytes == tot_px and n_segs == seg_off[-1]


def test_bitmaps_through_front_end(oracle, vg, ctx, fira_oracle):
    cmd_off, cmds, scale, shift, ids = record(vg, fira_oracle, range(0x20, 0x180))
    rects, out_bytes, _ = ctx.outlines_prepare(cmd_off, cmds, scale, shift)
    out = ctx.outlines_render()
    off = 0
    for g, cp in enumerate(ids):
        info, bm = fira_oracle.render_glyph(cp, oracle.PRECISE)
        if bm is None:
            continue
        n = bm.size
        assert np.array_equal(out[off:off + n].reshape(bm.shape), bm), cp
        off += n
    assert off == out_bytes


def test_one_submission_form_equals_prepare_plus_render(vg, fira_oracle):
    """vgsdf_outlines_render_into: the raster is enqueued behind the front-end with guessed sizes; whatever the
    guesses (fresh context, batches growing and shrinking, capacity too small, page-locked or pageable destination)
    the rects and bitmaps are those of prepare + render."""
    ref_ctx = vg.SdfContext(0)
    ctx2 = vg.SdfContext(0)
    for lo, hi in [(0x20, 0x60), (0x20, 0x400), (0x41, 0x48), (0x100, 0x180), (0x20, 0x2000)]:
        cmd_off, cmds, scale, shift, ids = record(vg, fira_oracle, range(lo, hi))
        rects, out_bytes, n_seg = ref_ctx.outlines_prepare(cmd_off, cmds, scale, shift)
        want = ref_ctx.outlines_render()
        for pinned in (True, False):
            r2, out, ob, ns = ctx2.outlines_render_into(cmd_off, cmds, scale, shift, out_bytes + 1000, pinned=pinned)
            assert (ob, ns) == (out_bytes, n_seg) and r2.tobytes() == rects.tobytes()
            assert out is not None and out.tobytes() == want.tobytes(), (lo, hi, pinned)
        if out_bytes > 1:  # too small: nothing rendered, the batch stays prepared
            r2, out, ob, ns = ctx2.outlines_render_into(cmd_off, cmds, scale, shift, out_bytes - 1)
            assert out is None and ob == out_bytes and r2.tobytes() == rects.tobytes()
            assert ctx2.outlines_render().tobytes() == want.tobytes()
    ref_ctx.close()
    ctx2.close()


def test_two_submissions_in_flight_on_two_contexts(vg, fira_oracle):
    """vgsdf_outlines_submit / _wait: two contexts each hold a batch in flight (what FontManager does with its two
    lanes); results equal prepare + render, a second submit before the wait is refused."""
    ref = vg.SdfContext(0)
    a, b = vg.SdfContext(0), vg.SdfContext(0)
    batches = [record(vg, fira_oracle, range(lo, hi)) for lo, hi in [(0x20, 0x200), (0x200, 0x500), (0x30, 0x40), (0x20, 0x1000)]]
    want = []
    for cmd_off, cmds, scale, shift, _ in batches:
        rects, ob, ns = ref.outlines_prepare(cmd_off, cmds, scale, shift)
        want.append((rects.tobytes(), ref.outlines_render().tobytes(), ob, ns))
    for rnd in range(2):  # second round: the contexts have their guesses from the first
        for k in range(0, len(batches), 2):
            a.outlines_submit(*batches[k][:4], want[k][2] + 64)
            b.outlines_submit(*batches[k + 1][:4], want[k + 1][2] + 64)
            with pytest.raises(vg.VgsdfError):
                a.outlines_submit(*batches[k][:4], 1 << 20)   # one batch in flight per context
            for c, j in ((a, k), (b, k + 1)):
                rects, out, ob, ns = c.outlines_wait()
                assert rects.tobytes() == want[j][0] and (ob, ns) == want[j][2:], (rnd, j)
                assert out is not None and out.tobytes() == want[j][1], (rnd, j)
    for c in (ref, a, b):
        c.close()


def test_packed_upload_form_equals_the_record_form(vg, fira_oracle):
    """vgsdf_outlines_submit_packed (kind bytes + only the coordinates a kind carries, what FontManager uploads)
    gives the rects, segments and bitmaps of the 28-byte records: a font, and streams with cubics, closes, commands on
    empty rings and an unknown kind; offsets that do not match the kinds are refused."""
    ref, c = vg.SdfContext(0), vg.SdfContext(0)
    M, L, Q, C3, Z = 0, 1, 2, 3, 4
    cmd_off, cmds, scale, shift, _ = record(vg, fira_oracle, range(0x20, 0x600))
    extra = [(0, 0, 0, 0, 100, 100, M), (300, 500, 700, 500, 900, 100, C3), (0, 0, 0, 0, 500, -300, L), (0,) * 6 + (Z,),
             (10, 10, 0, 0, 20, 20, Q), (0, 0, 0, 0, 0, 0, L), (0, 0, 0, 0, 400, 0, L), (600, 300, 0, 0, 400, 600, Q), (0, 0, 0, 0, 0, 600, L),
             (0,) * 6 + (Z,), (0,) * 6 + (Z,)]
    cmds2 = np.concatenate([cmds, np.array(extra, dtype=vg.OUTLINE_CMD_DTYPE)])
    cmd_off2 = np.concatenate([cmd_off, [len(cmds) + 4, len(cmds) + 9, len(cmds2)]]).astype(np.uint32)
    scale2 = np.concatenate([scale, [0.024, 0.024, 0.024]])
    shift2 = np.concatenate([shift, [0.0, 0.125, -0.25]])
    rects, ob, ns = ref.outlines_prepare(cmd_off2, cmds2, scale2, shift2)
    want_out = ref.outlines_render()
    want_off, want_segs = ref.outlines_segments()
    dat_off, kinds, coords = vg.SdfContext.pack_outlines(cmd_off2, cmds2)
    assert len(coords) * 4 + len(kinds) < 0.5 * cmds2.nbytes   # the point of the form
    for _ in range(2):
        c.outlines_submit_packed(cmd_off2, dat_off, kinds, coords, scale2, shift2, ob + 64)
        r2, out, ob2, ns2 = c.outlines_wait()
        assert r2.tobytes() == rects.tobytes() and (ob2, ns2) == (ob, ns)
        assert out is not None and out.tobytes() == want_out.tobytes()
        off2, segs2 = c.outlines_segments()
        assert off2.tobytes() == want_off.tobytes() and segs2.tobytes() == want_segs.tobytes()
    bad = dat_off.copy()
    bad[5:] -= 2                                                # glyph 4 is given two coordinates too few
    c.outlines_submit_packed(cmd_off2, bad, kinds, coords[:-2], scale2, shift2, ob + 64)
    with pytest.raises(vg.VgsdfError, match="dat_off"):
        c.outlines_wait()
    kinds_bad = kinds.copy()
    kinds_bad[3] = 9                                            # none of the five callbacks
    c.outlines_submit_packed(cmd_off2, dat_off, kinds_bad, coords, scale2, shift2, ob + 64)
    with pytest.raises(vg.VgsdfError):
        c.outlines_wait()
    ref.close()
    c.close()


def test_one_submission_with_a_glyph_for_the_brute_force_class(oracle, vg):
    """a bitmap too wide for the span kernel's winding histogram goes to the brute-force kernel: the raster enqueued
    behind the plan must not run (PlanHeader::ok = 0) and the second launches must give the oracle's bytes"""
    M, L, Z = 0, 1, 4
    wide = [(0, 0, 0, 0, 0, 0, M), (0, 0, 0, 0, 60000, 0, L), (0, 0, 0, 0, 60000, 1000, L), (0, 0, 0, 0, 0, 1000, L), (0,) * 6 + (Z,)]
    small = [(0, 0, 0, 0, 100, 100, M), (0, 0, 0, 0, 900, 100, L), (0, 0, 0, 0, 500, 800, L), (0,) * 6 + (Z,)]
    cmds = np.array(small + wide + small, dtype=vg.OUTLINE_CMD_DTYPE)
    cmd_off = np.array([0, 4, 9, 13], np.uint32)
    scale = np.full(3, 24.0 / 1000.0)
    shift = np.array([0.0, 0.25, -0.125])
    c = vg.SdfContext(0)
    for _ in range(2):
        rects, out, ob, ns = c.outlines_render_into(cmd_off, cmds, scale, shift, 1 << 20)
        assert out is not None and int(rects[1]["w"]) > 1400 and len(out) == ob
        off = 0
        for g, st in enumerate((small, wide, small)):
            segs = []
            for r in oracle.build_rings([(k[6],) + tuple(k[:6]) for k in st]):
                p = r * scale[g]
                p[:, 0] += shift[g]
                p[:, 1] += 0.0
                segs.append(np.concatenate([p[:-1], p[1:]], axis=1))
            r = rects[g]
            want = oracle.sdf_render(np.concatenate(segs), int(r["x0"]), int(r["y0"]), int(r["w"]), int(r["h"]))
            n = want.size
            assert np.array_equal(out[off:off + n].reshape(want.shape), want), g
            off += n
        assert off == ob
    c.close()


def test_arbitrary_command_streams(oracle, vg, ctx):
    # streams ttf-parser never emits: curve_to, quad_to on an empty ring, line_to starting a
    # ring, missing close, degenerate rings, repeated closes; rings via the oracle's RingBuilder
    M, L, Q, C, Z = 0, 1, 2, 3, 4
    streams = [
        [(M, 0, 0, 0, 0, 100, 100), (C, 300, 500, 700, 500, 900, 100), (L, 0, 0, 0, 0, 500, -300), (Z,) + (0,) * 6],
        [(Q, 10, 10, 0, 0, 20, 20), (L, 0, 0, 0, 0, 0, 0), (L, 0, 0, 0, 0, 400, 0), (Q, 600, 300, 0, 0, 400, 600),
         (L, 0, 0, 0, 0, 0, 600)],  # no close: into_rings saves it
        [(M, 0, 0, 0, 0, 0, 0), (L, 0, 0, 0, 0, 10, 0), (Z,) + (0,) * 6, (Z,) + (0,) * 6,
         (M, 0, 0, 0, 0, 50, 50), (L, 0, 0, 0, 0, 450, 50), (L, 0, 0, 0, 0, 450, 450), (L, 0, 0, 0, 0, 50, 50), (Z,) + (0,) * 6],
        [(M, 0, 0, 0, 0, 0, 0), (Z,) + (0,) * 6, (Q, 1, 1, 0, 0, 2, 2), (Q, 5, 5, 0, 0, 9, 9)],  # nothing survives
        [],
    ]
    cmds, cmd_off = [], [0]
    for st in streams:
        for c in st:
            c = tuple(c) + (0,) * (7 - len(c))
            cmds.append((c[1], c[2], c[3], c[4], c[5], c[6], c[0]))
        cmd_off.append(len(cmds))
    scale = np.full(len(streams), 24.0 / 1000.0)
    shift = np.array([0.0, 0.125, -0.25, 0.0, 0.0])
    rects, _, _ = ctx.outlines_prepare(np.array(cmd_off, np.uint32), np.array(cmds, dtype=vg.OUTLINE_CMD_DTYPE), scale, shift)
    seg_off, segs = ctx.outlines_segments()
    for g, st in enumerate(streams):
        rings = oracle.build_rings([tuple(c) + (0,) * (7 - len(c)) for c in st])
        want = []
        for r in rings:
            p = r * scale[g]
            p[:, 0] += shift[g]
            p[:, 1] += 0.0
            want.append(np.concatenate([p[:-1], p[1:]], axis=1))
        want = np.concatenate(want) if want else np.zeros((0, 4))
        got = segs[seg_off[g]:seg_off[g + 1]]
        if len(want):
            assert rects[g]["has_raster"] == 1 and got.tobytes() == want.tobytes(), g
            allp = np.concatenate([want[:, :2], want[:, 2:]])
            assert int(rects[g]["x0"]) == int(np.floor(allp[:, 0].min())) - 3
            assert int(rects[g]["h"]) == int(np.ceil(allp[:, 1].max())) + 3 - (int(np.floor(allp[:, 1].min())) - 3)
        else:
            assert rects[g]["has_raster"] == 0 and len(got) == 0, g


def test_quadratics_at_the_flatness_threshold(oracle, vg, ctx):
    """The flattening kernels treat a quad_to as a complete tree of depth L when |s + e - 2c|^2 / 16^L is clear of
    tolerance^2 by 1e-6 (relative) and walk it sequentially otherwise: glyphs whose curves sit at, just inside and
    just outside that margin at every depth, large coordinates (beyond the 1e6 bound of the argument), a scale
    that is not monotone (boxes on transformed points), all against the oracle's sequential RingBuilder."""
    M, L, Q, Z = 0, 1, 2, 4
    rng = np.random.default_rng(11)
    streams, scales = [], []
    for lev in range(0, 9):
        h0 = 0.05 * 4.0 ** lev  # s + e - 2c = (0, -2h): D / 16^lev = 0.01 at h = h0
        for rel in (0.0, 3e-8, -3e-8, 4e-7, -4e-7, 9e-7, -9e-7, 1.2e-6, -1.2e-6, 3e-6, -3e-6, 1e-4, -1e-4, 0.3, -0.3):
            h = np.float32(h0 * (1.0 + rel) ** 0.5)
            a = np.float32(rng.uniform(50, 900))
            ox, oy = (np.float32(v) for v in rng.uniform(-300, 300, 2))
            st = [(M, 0, 0, 0, 0, ox, oy), (Q, ox + a, oy + h, 0, 0, ox + 2 * a, oy), (L, 0, 0, 0, 0, ox + a, oy - 500), (Z,) + (0,) * 6]
            streams.append(st)
            scales.append(24.0 / 1000.0)
    # coordinates beyond the bound of the complete-tree argument; a mirrored glyph (scale < 0)
    streams.append([(M, 0, 0, 0, 0, 2.0e6, 2.0e6), (Q, 2.0e6 + 400, 2.0e6 + 300, 0, 0, 2.0e6 + 800, 2.0e6), (L, 0, 0, 0, 0, 2.0e6 + 400, 2.0e6 - 400), (Z,) + (0,) * 6])
    scales.append(24.0 / 1000.0)
    streams.append([(M, 0, 0, 0, 0, 10, 10), (Q, 300, 700, 0, 0, 600, 10), (Q, 300, -200, 0, 0, 10, 10), (Z,) + (0,) * 6])
    scales.append(-24.0 / 1000.0)
    cmds, cmd_off = [], [0]
    for st in streams:
        for c in st:
            cmds.append((c[1], c[2], c[3], c[4], c[5], c[6], c[0]))
        cmd_off.append(len(cmds))
    scale = np.array(scales)
    shift = rng.uniform(-0.5, 0.5, len(streams))
    rects, _, _ = ctx.outlines_prepare(np.array(cmd_off, np.uint32), np.array(cmds, dtype=vg.OUTLINE_CMD_DTYPE), scale, shift)
    seg_off, segs = ctx.outlines_segments()
    depths = set()
    for g, st in enumerate(streams):
        rings = oracle.build_rings([tuple(c) for c in st])
        want = []
        for r in rings:
            p = r * scale[g]
            p[:, 0] += shift[g]
            p[:, 1] += 0.0
            want.append(np.concatenate([p[:-1], p[1:]], axis=1))
        want = np.concatenate(want)
        got = segs[seg_off[g]:seg_off[g + 1]]
        assert got.tobytes() == want.tobytes(), g
        allp = np.concatenate([want[:, :2], want[:, 2:]])
        assert int(rects[g]["x0"]) == int(np.floor(allp[:, 0].min())) - 3, g
        assert int(rects[g]["y0"]) == int(np.floor(allp[:, 1].min())) - 3, g
        assert int(rects[g]["w"]) == int(np.ceil(allp[:, 0].max())) + 3 - int(rects[g]["x0"]), g
        depths.add(len(want))
    assert len(depths) >= 9   # the sweep really crossed depths (3 + 2^L segments per glyph)


@pytest.mark.parametrize("name", ["fira", "noto_all"])
def test_pbf_sha_with_device_front_end(vg, name):
    disp, paths = set_paths(name)
    m = vg.FontManager(True)
    m.set_device_front_end(True)
    fid = m.add_font_with_name(disp, paths)
    w = vg.DummyWriter()
    m.render_glyphs(w, vg.Renderer.new_precise(0))
    want = json.loads((GOLDEN / "pbf_sha256.json").read_text())[name]
    bad = [s for s, sha in want.items()
           if hashlib.sha256(w.files[f"{fid}/{s}-{int(s) + 255}.pbf"]).hexdigest() != sha]
    assert not bad, bad


def test_front_end_argument_validation(vg, ctx):
    """malformed command batches are refused with an error (and leave the context usable)"""
    M, L, Z = 0, 1, 4
    good = np.array([(0, 0, 0, 0, 10, 10, M), (0, 0, 0, 0, 400, 10, L), (0, 0, 0, 0, 400, 400, L), (0, 0, 0, 0, 10, 10, L),
                     (0, 0, 0, 0, 0, 0, Z)], dtype=vg.OUTLINE_CMD_DTYPE)
    sc, sh = np.array([0.024]), np.array([0.0])
    bad_kind = good.copy()
    bad_kind["kind"][2] = 9
    with pytest.raises(Exception, match="unknown command kind"):
        ctx.outlines_prepare(np.array([0, 5], np.uint32), bad_kind, sc, sh)
    with pytest.raises(Exception, match="cmd_off"):
        ctx.outlines_prepare(np.array([1, 5], np.uint32), good, sc, sh)
    with pytest.raises(Exception, match="monotone"):
        ctx.outlines_prepare(np.array([0, 5, 3], np.uint32), good, np.array([0.024, 0.024]), np.array([0.0, 0.0]))
    rects, out_bytes, n_segs = ctx.outlines_prepare(np.array([0, 5], np.uint32), good, sc, sh)
    assert int(rects[0]["has_raster"]) == 1 and n_segs == 3 and out_bytes == int(rects[0]["w"]) * int(rects[0]["h"])
    assert ctx.outlines_render().size == out_bytes


def test_point_count_overflow_is_an_error_not_a_fault(vg, ctx):
    """a curve that never becomes flat (non-finite or absurd control points) emits the capped
    131072 points; tens of thousands of them overflow every 32-bit total.  The totals are kept in 64 bits on
    the device, nothing is written past a capacity, and the batch is refused with VGSDF_E_ARG."""
    def batch(n_glyphs, quads_per_glyph):
        cmds = []
        cmd_off = [0]
        for _ in range(n_glyphs):
            cmds.append((0, 0, 0, 0, 0.0, 0.0, 0))  # move_to
            for i in range(quads_per_glyph):
                cmds.append((1e30, -1e30, 0, 0, float(i + 1), 0.0, 2))  # quad_to with an absurd control point
            cmds.append((0, 0, 0, 0, 0, 0, 4))  # close
            cmd_off.append(len(cmds))
        n = n_glyphs
        return (np.array(cmd_off, np.uint32), np.array(cmds, dtype=vg.OUTLINE_CMD_DTYPE), np.full(n, 0.024), np.zeros(n))
    # one glyph beyond 2^28 points (3000 x 131072)
    with pytest.raises(vg.VgsdfError, match="2\\^28"):
        ctx.outlines_prepare(*batch(1, 3000))
    # every glyph below 2^28 points, the batch beyond 2^32 segments (40 x 1000 x 131072 = 5.2e9)
    with pytest.raises(vg.VgsdfError, match="2\\^32"):
        ctx.outlines_prepare(*batch(40, 1000))
    # the context is still usable
    cmd_off, cmds, scale, shift = batch(1, 1)
    cmds["x1"], cmds["y1"] = 5.0, 5.0
    rects, _, n_segs = ctx.outlines_prepare(cmd_off, cmds, scale, shift)
    assert len(rects) == 1 and n_segs >= 0


def _varint_len(v):
    n = 1
    while v >= 0x80:
        v >>= 7
        n += 1
    return n


def test_in_place_pbf_layout_at_the_c_abi(vg, fira_oracle):
    """vgsdf_outlines_packed::pbf_pre / pbf_fix: the device lays the glyphs out as the `glyphs` entries of a fontstack
    message (glyph.rs:10-41, fontstack.rs:9-25) and the raster stores every bitmap where the finished file has it.  Checked
    against a Python restatement of the wire format: positions, arena size, and the bitmap bytes against the packed form."""
    ref, c = vg.SdfContext(0), vg.SdfContext(0)
    cps = list(range(0x20, 0x180))
    cmd_off, cmds, scale, shift, _ = record(vg, fira_oracle, cps)
    n = len(scale)
    dat_off, kinds, coords = vg.SdfContext.pack_outlines(cmd_off, cmds)
    ref.outlines_submit_packed(cmd_off, dat_off, kinds, coords, scale, shift, 1 << 20)
    rects, packed, ob, ns = ref.outlines_wait()
    ids = np.array(cps[:n], dtype=np.uint32) * 37 + 5        # any ids / advances: only their varint lengths matter here
    adv = (np.arange(n, dtype=np.uint32) * 11) % 300
    fix = np.array([(1 + _varint_len(int(i))) | ((1 + _varint_len(int(a))) << 4) for i, a in zip(ids, adv)], dtype=np.uint8)
    pre = np.zeros(n, dtype=np.uint32)
    pre[0], pre[100], pre[101] = 6 + 30, 6 + 31, 77          # "block" starts
    for cap in (1 << 20, 64):                                # the second: the arena does not fit -> second launches
        c.outlines_submit_packed(cmd_off, dat_off, kinds, coords, scale, shift, cap, pbf_pre=pre, pbf_fix=fix)
        r2, arena, ob2, ns2 = c.outlines_wait()
        assert r2.tobytes() == rects.tobytes() and ns2 == ns
        if arena is None:
            arena = c.outlines_render()
        at = c.outlines_pbf_positions()
        pos, poff = 0, 0
        for g in range(n):
            r = rects[g]
            has = bool(r["has_raster"])
            px = int(r["w"]) * int(r["h"]) if has else 0
            idlen, advlen = int(fix[g]) & 15, int(fix[g]) >> 4
            msg = idlen + advlen
            if has:
                left, top = int(r["x0"]) + 3, int(r["y0"]) + int(r["h"]) - 27
                zz = lambda v: (v << 1) ^ (v >> 31)  # noqa: E731
                msg += 1 + _varint_len(px) + px + 4 + _varint_len(int(r["w"]) - 6) + _varint_len(int(r["h"]) - 6) + \
                    _varint_len(zz(left) & 0xFFFFFFFF) + _varint_len(zz(top) & 0xFFFFFFFF)
            else:
                msg += 8
            want_at = pos + int(pre[g]) + 1 + _varint_len(msg) + idlen + ((1 + _varint_len(px)) if has else 0)
            assert int(at[g]) == want_at, g
            if has:
                assert arena[want_at:want_at + px].tobytes() == packed[poff:poff + px].tobytes(), g
                poff += px
            pos += int(pre[g]) + 1 + _varint_len(msg) + msg
        assert ob2 == pos and poff == ob
    # between submit and wait the front-end's results are available while the raster is still running (vgsdf_outlines_peek)
    c.outlines_submit_packed(cmd_off, dat_off, kinds, coords, scale, shift, 1 << 20, pbf_pre=pre, pbf_fix=fix)
    r3, ob3, in_place = c.outlines_peek()
    assert r3.tobytes() == rects.tobytes() and ob3 == ob2 and in_place  # page-locked destination, capacities held
    assert c.outlines_pbf_positions().tobytes() == at.tobytes()
    r4, arena4, ob4, _ = c.outlines_wait()
    assert r4.tobytes() == rects.tobytes() and ob4 == ob2 and arena4 is not None
    c.outlines_submit_packed(cmd_off, dat_off, kinds, coords, scale, shift, 64, pbf_pre=pre, pbf_fix=fix)
    assert c.outlines_peek()[2] is False                                # the arena does not fit: bitmaps only after the wait
    c.outlines_wait()
    with pytest.raises(vg.VgsdfError, match="nothing was submitted"):
        c.outlines_peek()
    with pytest.raises(vg.VgsdfError, match="pbf_pre"):
        c.outlines_submit_packed(cmd_off, dat_off, kinds, coords, scale, shift, 1 << 20, pbf_pre=pre)
    ref.close()
    c.close()


@pytest.mark.parametrize("name", ["fira", "noto_all"])
def test_in_place_assembly_equals_encoding_afterwards(vg, name):
    """FontManager with blocks assembled in place (default) and encoded after the render give the same files, through both
    dispatchers; all of them carry the golden SHA-256s (test_pbf_sha_with_device_front_end covers the default)."""
    disp, paths = set_paths(name)
    m = vg.FontManager(True)
    fid = m.add_font_with_name(disp, paths)
    r = vg.Renderer.new_precise(0)
    got = {}
    for fe in (True, False):
        for in_place in (True, False):
            m.set_device_front_end(fe)
            m.set_in_place_pbf(in_place)
            w = vg.DummyWriter()
            m.render_glyphs(w, r)
            got[(fe, in_place)] = w.files
    first = got[(True, True)]
    assert all(v == first for v in got.values())
    want = json.loads((GOLDEN / "pbf_sha256.json").read_text())[name]
    assert all(hashlib.sha256(first[f"{fid}/{s}-{int(s) + 255}.pbf"]).hexdigest() == sha for s, sha in want.items())
