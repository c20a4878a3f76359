"""Random `glyf` entries through the device's decoder (vgsdf_outlines_submit_glyf) against the sequential Python decoder of
tests/test_glyf_parts_host.py (written from the specification, checked against the host's reader on the fixture fonts).

Fonts written by tools repeat the same few encodings; here every entry chooses its own: flag runs compressed or not (a run
may be cut anywhere, also at 64-byte windows of the decoder), repeat counts whose own bit 3 is set, deltas as one byte
with either sign, as two bytes (also when one byte would do), or the "same" form, contours of 1 .. 70 points starting on
or off the curve, coordinates that wrap the i16 range, composite transforms.  Both decoders' callbacks go through the same
device front-end; the segments, rects and bitmaps must be equal bit for bit.
"""
import numpy as np
import pytest

from test_glyf_parts_host import Z, _decode_part

pytestmark = pytest.mark.gpu


def _random_entry(rng, big=False):
    """-> (bytes of the entry: end points + flags + x + y, n_contours, n_points)"""
    nc = int(rng.integers(1, 7))
    lengths = [int(rng.integers(1, 70 if big else 25)) for _ in range(nc)]
    n = sum(lengths)
    ends = np.cumsum(lengths) - 1
    style = int(rng.integers(0, 4))          # how the coordinates move: small steps, larger ones, axis-parallel runs, wild
    flags, xs, ys = [], [], []
    for _ in range(n):
        on = int(rng.random() < (0.5 if style != 2 else 0.9))
        if style == 0:
            dx, dy = int(rng.integers(-40, 41)), int(rng.integers(-40, 41))
        elif style == 1:
            dx, dy = int(rng.integers(-600, 601)), int(rng.integers(-600, 601))
        elif style == 2:
            dx, dy = (int(rng.integers(-90, 91)), 0) if rng.random() < 0.5 else (0, int(rng.integers(-90, 91)))
        else:
            dx, dy = int(rng.integers(-32768, 32768)), int(rng.integers(-32768, 32768))
        fl = on

        def enc(d, short_bit, same_bit):
            nonlocal fl
            if d == 0 and rng.random() < 0.8:
                fl |= same_bit                      # "same as before"
                return b""
            if -255 <= d <= 255 and d != 0 and rng.random() < 0.8:
                fl |= short_bit | (same_bit if d > 0 else 0)
                return bytes([abs(d)])
            return int(d).to_bytes(2, "big", signed=True)  # two bytes (also for values one byte could hold, and for 0)
        xs.append(enc(dx, 0x02, 0x10))
        ys.append(enc(dy, 0x04, 0x20))
        flags.append(fl)
    # flag stream: runs of equal flags compressed with a repeat count — or not, or only partly
    stream, i = bytearray(), 0
    while i < n:
        run = 1
        while i + run < n and flags[i + run] == flags[i] and run < 256:
            run += 1
        if run > 1 and rng.random() < 0.7:
            take = run if rng.random() < 0.6 else int(rng.integers(2, run + 1))
            stream += bytes([flags[i] | 0x08, take - 1])
            i += take
        else:
            stream.append(flags[i])
            i += 1
    body = b"".join(int(e).to_bytes(2, "big") for e in ends) + bytes(stream) + b"".join(xs) + b"".join(ys)
    return body, nc, n, style == 3


def _batch(rng, n_glyphs, big=False):
    from versatiles_glyphs_rs_amd.device import GLYF_PART_DTYPE
    parts, data, cmd_off, slots, wild = [], bytearray(), [0], 0, []
    for _ in range(n_glyphs):
        wild.append(False)
        for _ in range(int(rng.integers(1, 4))):      # parts of the glyph (a composite's components)
            body, nc, n, w = _random_entry(rng, big)
            wild[-1] = wild[-1] or w
            p = np.zeros((), dtype=GLYF_PART_DTYPE)
            p["byte_off"], p["byte_len"] = len(data), len(body)
            p["cmd_at"], p["cmd_cap"] = slots, n + 2 * nc + int(rng.integers(0, 3))
            p["n_contours"] = nc
            if rng.random() < 0.5:
                p["plain"], p["a"], p["d"] = 1, 1.0, 1.0
            else:
                t = rng.choice([0.5, 0.75, 1.0, -1.0, 1.25, 0.0], 4) if rng.random() < 0.5 else rng.uniform(-1.5, 1.5, 4)
                p["a"], p["b"], p["c"], p["d"] = (np.float32(v) for v in t)
                p["e"], p["f"] = np.float32(rng.integers(-300, 301)), np.float32(rng.integers(-300, 301))
            data += body + b"\0" * (-len(body) % 4)
            slots += int(p["cmd_cap"])
            parts.append(p)
        cmd_off.append(slots)
    return (np.array(parts, dtype=GLYF_PART_DTYPE), np.frombuffer(bytes(data), dtype=np.uint8), np.array(cmd_off, dtype=np.uint32),
            np.array(wild))


@pytest.mark.parametrize("seed,big", [(1, False), (2, False), (3, True), (4, True)])
def test_random_entries_device_decoder_equals_the_python_decoder(vg, seed, big):
    from versatiles_glyphs_rs_amd.device import OUTLINE_CMD_DTYPE
    rng = np.random.default_rng(seed)
    n_glyphs = 150
    parts, data, cmd_off, wild = _batch(rng, n_glyphs, big)
    # commands of the sequential decoder, glyph by glyph
    cmds, host_off, pi = [], [0], 0
    for g in range(n_glyphs):
        while pi < len(parts) and int(parts["cmd_at"][pi]) < int(cmd_off[g + 1]):
            got = _decode_part(parts[pi], data)
            assert len(got) <= int(parts["cmd_cap"][pi])
            cmds += got
            pi += 1
        host_off.append(len(cmds))
    arr = np.zeros(len(cmds), dtype=OUTLINE_CMD_DTYPE)
    for k, (kind, x1, y1, x, y) in enumerate(cmds):
        arr[k]["kind"], arr[k]["x1"], arr[k]["y1"], arr[k]["x"], arr[k]["y"] = kind, x1, y1, x, y
    scale = np.full(n_glyphs, 24.0 / 1000.0) * rng.choice([1.0, 0.5, 2.0], n_glyphs)
    scale[wild] = 24.0 / 200000.0   # (coordinates all over the i16 range: keep the bitmaps small)
    shift = rng.uniform(-0.5, 0.5, n_glyphs)
    ctx = vg.SdfContext(0)
    try:
        rects_h, out_bytes_h, n_seg_h = ctx.outlines_prepare(np.array(host_off, dtype=np.uint32), arr, scale, shift)
        bitmaps_h = ctx.outlines_render()
        seg_off_h, segs_h = ctx.outlines_segments()
        ctx.outlines_submit_glyf(cmd_off, parts, data, scale, shift, capacity=int(out_bytes_h) + 64)
        rects_d, bitmaps_d, out_bytes_d, n_seg_d = ctx.outlines_wait()
        seg_off_d, segs_d = ctx.outlines_segments()
    finally:
        ctx.close()
    assert len(parts) > n_glyphs and n_seg_h > 1000
    assert np.array_equal(rects_d, rects_h) and np.array_equal(seg_off_d, seg_off_h)
    assert segs_d.tobytes() == segs_h.tobytes()
    assert bitmaps_d is not None and np.array_equal(bitmaps_d, bitmaps_h)
    assert sum(1 for c in cmds if c[0] == Z) > n_glyphs
