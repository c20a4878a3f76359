#!/usr/bin/env python3
"""Differential fuzzing of the device's glyf decoder (vgsdf_outlines_submit_glyf): random `glyf` entries in random encodings
(generator and sequential Python decoder of tests/test_gpu_glyf_fuzz.py / tests/test_glyf_parts_host.py), both decoders'
callbacks through the same device front-end; segments, rects and bitmaps must be equal bit for bit.
usage: fuzz_glyf.py [seconds=120] [seed=51]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from conftest import load_product  # noqa: E402
from test_glyf_parts_host import _decode_part  # noqa: E402
from test_gpu_glyf_fuzz import _batch  # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 51
    vg = load_product()
    from versatiles_glyphs_rs_amd.device import OUTLINE_CMD_DTYPE
    rng = np.random.default_rng(seed)
    ctx = vg.SdfContext(0)
    t0, batches, glyphs, n_parts, n_points, bad = time.time(), 0, 0, 0, 0, 0
    while time.time() - t0 < seconds:
        n_glyphs = int(rng.integers(20, 200))
        parts, data, cmd_off, wild = _batch(rng, n_glyphs, big=bool(rng.integers(0, 2)))
        cmds, host_off, pi = [], [0], 0
        for g in range(n_glyphs):
            while pi < len(parts) and int(parts["cmd_at"][pi]) < int(cmd_off[g + 1]):
                cmds += _decode_part(parts[pi], data)
                pi += 1
            host_off.append(len(cmds))
        arr = np.zeros(len(cmds), dtype=OUTLINE_CMD_DTYPE)
        for k, (kind, x1, y1, x, y) in enumerate(cmds):
            arr[k]["kind"], arr[k]["x1"], arr[k]["y1"], arr[k]["x"], arr[k]["y"] = kind, x1, y1, x, y
        scale = np.full(n_glyphs, 24.0 / 1000.0) * rng.choice([1.0, 0.5, 2.0], n_glyphs)
        scale[wild] = 24.0 / 200000.0
        shift = rng.uniform(-0.5, 0.5, n_glyphs)
        rects_h, out_bytes_h, _ = ctx.outlines_prepare(np.array(host_off, dtype=np.uint32), arr, scale, shift)
        bitmaps_h = ctx.outlines_render()
        seg_off_h, segs_h = ctx.outlines_segments()
        ctx.outlines_submit_glyf(cmd_off, parts, data, scale, shift, capacity=int(out_bytes_h) + 64)
        rects_d, bitmaps_d, _, _ = ctx.outlines_wait()
        seg_off_d, segs_d = ctx.outlines_segments()
        same = (np.array_equal(rects_d, rects_h) and np.array_equal(seg_off_d, seg_off_h) and segs_d.tobytes() == segs_h.tobytes()
                and bitmaps_d is not None and np.array_equal(bitmaps_d, bitmaps_h))
        if not same:
            bad += 1
            print(f"batch {batches} (seed {seed}): DIFFERENT", flush=True)
        batches += 1
        glyphs += n_glyphs
        n_parts += len(parts)
        n_points += int((parts["cmd_cap"].astype(np.int64) - 3 * parts["n_contours"]).sum())
        if batches % 50 == 0:
            print(f"{batches} batches, {glyphs} glyphs, {n_parts} entries ...", flush=True)
    ctx.close()
    print(f"done: {batches} batches, {glyphs} glyphs, {n_parts} glyf entries, ~{n_points} points, {bad} differing batches (seed {seed})")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
