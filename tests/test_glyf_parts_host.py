"""Host half of the device's glyf decoder (CPU): what FontManager hands to vgsdf_outlines_submit_glyf.

`record_glyf_parts` lists every glyph's simple glyphs as PARTS — end points + flag / coordinate arrays copied as they stand,
the component transform ttf-parser accumulates (glyf.rs, call site /root/reference/src/render/renderer.rs:110), the command
slots the part may fill.  Here a plain sequential decoder written in Python from the OpenType `glyf` specification and
ttf-parser's Builder rules (implied on-curve midpoints, the closing curve, close()) walks those parts; its callbacks must be
the ones the host's reader records for the same glyph (csrc/host/ttf_face.cpp, pinned by the oracle and the golden SHAs),
coordinate for coordinate in f32, and fit the slots.  The device kernel is checked against the same reader on the GPU
(tests/test_gpu_glyf_decode.py).
"""
import numpy as np
import pytest

from conftest import FIRA, NOTO

M, L, Q, C, Z = 0, 1, 2, 3, 4
f32 = np.float32


def _decode_part(part, data):
    """callbacks of one part -> list of (kind, x1, y1, x, y) in f32"""
    b = bytes(data[int(part["byte_off"]):int(part["byte_off"]) + int(part["byte_len"])])
    nc = int(part["n_contours"])
    ends = [int.from_bytes(b[2 * k:2 * k + 2], "big") for k in range(nc)]
    n_points = ends[-1] + 1
    # flags (run-length), then the x and y deltas
    flags, cur = [], 2 * nc
    while len(flags) < n_points:
        fl = b[cur]
        cur += 1
        run = 1
        if fl & 8:
            run += b[cur]
            cur += 1
        flags += [fl] * run
    assert len(flags) == n_points

    def deltas(at, short, same):
        out = []
        for fl in flags:
            if fl & short:
                v = b[at]
                at += 1
                out.append(v if fl & same else -v)
            elif fl & same:
                out.append(0)
            else:
                out.append(int.from_bytes(b[at:at + 2], "big", signed=True))
                at += 2
        return out, at

    dxs, y_at = deltas(cur, 2, 0x10)
    dys, end = deltas(y_at, 4, 0x20)
    assert end <= len(b)
    xs = ((np.cumsum(dxs) + 32768) % 65536 - 32768).astype(np.int64)   # wrapping i16
    ys = ((np.cumsum(dys) + 32768) % 65536 - 32768).astype(np.int64)
    a, bb, c, d, e, f = (f32(part[k]) for k in "abcdef")
    plain = bool(part["plain"])

    def tr(x, y):
        if plain:
            return f32(x), f32(y)
        return f32(f32(a * x) + f32(c * y)) + e, f32(f32(bb * x) + f32(d * y)) + f

    def mid(p, q):
        return (p[0] + f32(0.5) * (q[0] - p[0]), p[1] + f32(0.5) * (q[1] - p[1]))

    out = []

    def move(p):
        out.append((M, f32(0), f32(0)) + tr(*p))

    def line(p):
        out.append((L, f32(0), f32(0)) + tr(*p))

    def quad(cp, p):
        out.append((Q,) + tr(*cp) + tr(*p))

    first = 0
    for k in range(nc):
        # EndpointsIter: a contour runs to its end point; an end point that does not ascend still takes one point
        length = ends[0] + 1 if k == 0 else (ends[k] - ends[k - 1] if ends[k] > ends[k - 1] else 1)
        pts = [((f32(xs[i]), f32(ys[i])), bool(flags[i] & 1)) for i in range(first, min(first + length, n_points))]
        complete = first + length <= n_points
        first += length
        start = lead = pend = None
        for p, on in pts:   # Builder::push_point
            if start is None:
                if on:
                    start = p
                    move(p)
                elif lead is not None:
                    start = mid(lead, p)
                    pend = p
                    move(start)
                else:
                    lead = p
            elif pend is not None:
                cp = pend
                if on:
                    pend = None
                    quad(cp, p)
                else:
                    pend = p
                    quad(cp, mid(cp, p))
            elif on:
                line(p)
            else:
                pend = p
        if not complete:
            break
        if lead is not None and pend is not None:   # Builder::finish
            cp, pend = pend, None
            quad(cp, mid(cp, lead))
        if start is not None and lead is not None:
            quad(lead, start)
        elif start is not None and pend is not None:
            quad(pend, start)
        elif start is not None:
            line(start)
        out.append((Z, f32(0), f32(0), f32(0), f32(0)))
    return out


@pytest.mark.parametrize("path", [FIRA, NOTO], ids=["fira", "noto_regular"])
def test_parts_decode_to_the_callbacks_the_host_reader_records(vg, path):
    mgr = vg.FontManager(False)
    fid = mgr.add_font_with_name("Font", [path])
    o = mgr.record_outlines(fid)
    g = mgr.record_glyf_parts(fid)
    assert list(o["ids"]) == list(g["ids"]) and list(o["advances"]) == list(g["advances"])
    assert np.array_equal(o["scale"], g["scale"]) and np.array_equal(o["shift_x"], g["shift_x"])
    parts, data = g["parts"], g["bytes"]
    # the parts tile the command slots in order; their bytes lie 4-aligned inside the store
    assert len(data) % 4 == 0 and (parts["byte_off"] % 4 == 0).all() and (parts["n_contours"] > 0).all()
    assert (parts["byte_off"].astype(np.int64) + parts["byte_len"] <= len(data)).all()
    assert np.array_equal(parts["cmd_at"], np.concatenate([[0], np.cumsum(parts["cmd_cap"])[:-1]]).astype(np.uint32))
    assert int(g["cmd_off"][-1]) == int(parts["cmd_cap"].sum()) and g["cmd_off"][0] == 0
    n_glyphs, pi, n_composite, n_moved = len(o["ids"]), 0, 0, 0
    for gi in range(n_glyphs):
        want = [(int(c["kind"]), f32(c["x1"]), f32(c["y1"]), f32(c["x"]), f32(c["y"])) for c in o["cmds"][o["cmd_off"][gi]:o["cmd_off"][gi + 1]]]
        got, slots0, slots1 = [], int(g["cmd_off"][gi]), int(g["cmd_off"][gi + 1])
        n_parts_here = 0
        while pi < len(parts) and int(parts["cmd_at"][pi]) < slots1:
            assert int(parts["cmd_at"][pi]) >= slots0
            cmds = _decode_part(parts[pi], data)
            assert len(cmds) <= int(parts["cmd_cap"][pi])        # the slots hold what the entry decodes to
            got += cmds
            n_moved += not parts["plain"][pi]
            n_parts_here += 1
            pi += 1
        n_composite += n_parts_here > 1
        assert len(got) == len(want), hex(int(o["ids"][gi]))
        for a, b in zip(got, want):
            assert a[0] == b[0] and all(x.tobytes() == y.tobytes() or (x == 0 and y == 0) for x, y in zip(a[1:], b[1:])), hex(int(o["ids"][gi]))
    assert pi == len(parts) and n_glyphs > 1000
    if path == FIRA:
        assert n_composite > 400 and n_moved > 400   # accented letters: components moved by their offsets


def test_fonts_without_glyf_outlines_have_no_parts(vg):
    pytest.importorskip("fontTools")
    from test_cff2_outlines import _GLOBAL, _LOCAL, _NAMES, _PROGS, _build2
    cff2 = _build2(_NAMES, _PROGS, local_subrs=_LOCAL, global_subrs=_GLOBAL, extra_vardata=[(3, 0)])
    mgr = vg.FontManager(False)
    fid = mgr.add_font_data("CFF2", cff2)
    with pytest.raises(RuntimeError, match="glyf"):
        mgr.record_glyf_parts(fid)
    assert len(mgr.record_outlines(fid)["ids"]) == len(_NAMES) - 1   # the host's reader still walks them
