"""Synthetic stress workload (BASELINE.json configs[4]).

65 536 random high-segment-count outlines at 64x64 + 2x3 buffer: bitmap 70x70, x0=y0=-3;
each outline = 4 closed rings x 256 segments (N = 1024).  Ring k: centre uniform in
[12,52]^2, r0 uniform in [4,20], vertex j at angle 2*pi*j/256 with radius
r0*(1 + 0.35*u_j), u_j uniform in [-1,1]; coordinates clamped to [0,64]; ring orientation
alternates (k odd = reversed) to exercise winding +1/0/-1/+-2.  PRNG: SplitMix64, seed
0x5DF61F95, one stream, outlines in index order; per ring the draws are cx, cy, r0,
u_0..u_255.  Pure numpy; used by bench.py and the tests (inputs only, no reference maths).
"""
from __future__ import annotations

import numpy as np

SEED = 0x5DF61F95
RINGS, RING_SEGS = 4, 256
DRAWS_PER_OUTLINE = RINGS * (3 + RING_SEGS)
W = H = 70
X0 = Y0 = -3

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64_doubles(first_draw: int, count: int, seed: int = SEED) -> np.ndarray:
    """draws [first_draw, first_draw+count) of the stream as doubles in [0,1)"""
    with np.errstate(over="ignore"):
        idx = np.arange(first_draw + 1, first_draw + count + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def outlines(first: int, count: int):
    """-> segs f64 [count, 1024, 4] (sx,sy,ex,ey) for outlines [first, first+count)"""
    u = splitmix64_doubles(first * DRAWS_PER_OUTLINE, count * DRAWS_PER_OUTLINE).reshape(count, RINGS, 3 + RING_SEGS)
    cx = 12.0 + 40.0 * u[:, :, 0]
    cy = 12.0 + 40.0 * u[:, :, 1]
    r0 = 4.0 + 16.0 * u[:, :, 2]
    uj = 2.0 * u[:, :, 3:] - 1.0
    ang = 2.0 * np.pi * np.arange(RING_SEGS, dtype=np.float64) / RING_SEGS
    rad = r0[:, :, None] * (1.0 + 0.35 * uj)
    vx = np.clip(cx[:, :, None] + rad * np.cos(ang)[None, None, :], 0.0, 64.0)
    vy = np.clip(cy[:, :, None] + rad * np.sin(ang)[None, None, :], 0.0, 64.0)
    # alternate orientation: odd rings reversed
    vx[:, 1::2, :] = vx[:, 1::2, ::-1]
    vy[:, 1::2, :] = vy[:, 1::2, ::-1]
    nx, ny = np.roll(vx, -1, axis=2), np.roll(vy, -1, axis=2)
    segs = np.stack([vx, vy, nx, ny], axis=3).reshape(count, RINGS * RING_SEGS, 4)
    return np.ascontiguousarray(segs)


def make_batch(first: int, count: int):
    """SoA batch (device.Batch) of `count` synthetic outlines starting at index `first`."""
    from .device import Batch
    segs = outlines(first, count)
    n = RINGS * RING_SEGS
    flat = segs.reshape(count * n, 4)
    seg_off = (np.arange(count + 1, dtype=np.uint64) * n).astype(np.uint32)
    out_off = np.arange(count + 1, dtype=np.uint64) * np.uint64(W * H)
    col = lambda k: np.ascontiguousarray(flat[:, k])  # noqa: E731
    return Batch(seg_off, col(0), col(1), col(2), col(3), np.full(count, X0, np.int32), np.full(count, Y0, np.int32),
                 np.full(count, W, np.uint32), np.full(count, H, np.uint32), out_off)
