"""CPU model (numpy) of a WAVE-LEVEL group cull in the span kernel, for different wave footprints.

Round-2 verdict, item 1: "map a wave to an 8 x 8 pixel block, test each group ONCE per wave against the block's box
(lane <-> group), run the per-pixel test only over the surviving groups".  This model counts, per wave and chunk,

  union     groups that are a candidate of at least one pixel of the wave (floor of ANY wave-level cull)
  cull_p1   survivors of  Dnear_g(box) <= U_max + r_g  with U_max = max over the wave's pixels of the per-pixel bound the
            kernel computes in phase 1 (min over the chunk's anchors, the carried bound, SAT) — needs phase 1's first
            half (the D^2 of all groups for every pixel) before the cull, so only the candidate TEST could be restricted
  cull_carr survivors with U_max from the bound carried from earlier chunks only (SAT in the first chunk): what a cull
            in front of phase 1 can use
  cull_c8   survivors with U_max = max over pixels of the distance to the nearest of 8 coarse anchors (every 4th group)

for three footprints: `rows` = 64 consecutive output bytes (today), `8x8` = aligned blocks, `32x2` = two-row bands.
Lane use = pixels / (64 x waves).

    python tools/model_footprints.py noto_regular [max_glyphs]
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from conftest import FIRA, NOTO, load_product, noto_files  # noqa: E402

WORK = {"fira": ("Fira Sans Regular", [FIRA]), "noto_regular": ("Noto Sans Regular", [NOTO]),
        "noto_all": ("Noto Sans Regular", None)}
INFL = 1 + 1 / 512
GRP, CH, SAT = 8, 256, 6.2


def waves_rows(w, h):
    n = w * h
    nw = (n + 63) // 64
    idx = np.full(nw * 64, -1)
    idx[:n] = np.arange(n)
    return idx.reshape(nw, 64)


def waves_blocks(w, h, bw, bh):
    out = []
    for by in range(0, h, bh):
        for bx in range(0, w, bw):
            ys, xs = np.meshgrid(np.arange(by, by + bh), np.arange(bx, bx + bw), indexing="ij")
            ok = (ys < h) & (xs < w)
            out.append(np.where(ok, ys * w + xs, -1).ravel())
    return np.array(out)


def waves_band2(w, h):
    """quads (2 x 2 pixels) in band-major order, 16 consecutive quads per wave"""
    qw, qh = (w + 1) // 2, (h + 1) // 2
    lanes = []
    for q in range(qw * qh):
        b, c = divmod(q, qw)
        for qy in range(2):
            for qx in range(2):
                x, r = 2 * c + qx, 2 * b + qy
                lanes.append(r * w + x if (x < w and r < h) else -1)
    lanes = np.array(lanes)
    nw = (len(lanes) + 63) // 64
    idx = np.full(nw * 64, -1)
    idx[:len(lanes)] = lanes
    return idx.reshape(nw, 64)


def main():
    vg = load_product()
    name, files = WORK[sys.argv[1]]
    files = files or noto_files()
    mgr = vg.FontManager(True)
    fid = mgr.add_font_with_name(name, files)
    hb = mgr.build_batch(fid)
    b = hb.batch
    n = b.n_glyphs if len(sys.argv) < 3 else min(b.n_glyphs, int(sys.argv[2]))
    shapes = {"rows": lambda w, h: waves_rows(w, h), "8x8": lambda w, h: waves_blocks(w, h, 8, 8),
              "16x4": lambda w, h: waves_blocks(w, h, 16, 4), "32x2": waves_band2}
    acc = {s: dict(wtc=0, groups=0, union=0, p1=0, carr=0, c8=0, lanes=0, pix=0, nopair=0, first=0, first_union=0, first_groups=0)
           for s in shapes}
    for g in range(n):
        a, e = int(b.seg_off[g]), int(b.seg_off[g + 1])
        w, h, x0, y0 = int(b.w[g]), int(b.h[g]), int(b.x0[g]), int(b.y0[g])
        if e == a or w * h == 0:
            continue
        sx, sy, ex, ey = (b.seg_sx[a:e] - x0), (b.seg_sy[a:e] - y0), (b.seg_ex[a:e] - x0), (b.seg_ey[a:e] - y0)
        npix = w * h
        o = np.arange(npix)
        row = o // w
        px = (o - row * w) + 0.5
        py = (h - 1 - row) + 0.5
        nseg = e - a
        M = max(np.abs(np.concatenate([sx, sy, ex, ey])).max(), w, h)
        pad = 0.01 + 1e-5 * M
        # exact distances (for the carried bound)
        dx, dy = ex - sx, ey - sy
        l2 = dx * dx + dy * dy
        pvx = px[:, None] - sx[None, :]
        pvy = py[:, None] - sy[None, :]
        t = np.clip((pvx * dx + pvy * dy) / np.where(l2 > 0, l2, 1), 0, 1)
        d2 = (pvx - t * dx) ** 2 + (pvy - t * dy) ** 2
        ub2 = np.full(npix, np.inf)
        per_chunk = []
        for c0 in range(0, nseg, CH):
            c1 = min(nseg, c0 + CH)
            cnt = c1 - c0
            ng = (cnt + GRP - 1) // GRP
            ax = np.empty(ng); ay = np.empty(ng); r = np.empty(ng)
            for k in range(ng):
                gb = k * GRP
                ai = min(gb + GRP // 2, cnt - 1)
                ax[k], ay[k] = sx[c0 + ai], sy[c0 + ai]
                m0, m1 = c0 + gb, min(c0 + gb + GRP, c1)
                rr = np.maximum((sx[m0:m1] - ax[k]) ** 2 + (sy[m0:m1] - ay[k]) ** 2, (ex[m0:m1] - ax[k]) ** 2 + (ey[m0:m1] - ay[k]) ** 2).max()
                r[k] = (np.sqrt(rr) * INFL + pad) * INFL * 1.004
            D2 = (px[:, None] - ax[None, :]) ** 2 + (py[:, None] - ay[None, :]) ** 2
            dmin = np.minimum(ub2, D2.min(axis=1))
            U = np.minimum((np.sqrt(dmin) * INFL + pad) * INFL, SAT) * 1.004
            cand = (U[:, None] + r[None, :]) ** 2 - D2 * (1 - 2.0 ** -8) >= 0
            Ucarr = np.minimum(np.sqrt(ub2), SAT) * 1.004 + pad
            Uc8 = np.minimum(np.sqrt(np.minimum(ub2, D2[:, ::4].min(axis=1))), SAT) * 1.004 + pad
            per_chunk.append((ng, ax, ay, r, cand, U, Ucarr, Uc8, c0 == 0))
            segmask = np.repeat(cand, GRP, axis=1)[:, :cnt]
            f1 = np.where(segmask, d2[:, c0:c1], np.inf).min(axis=1)
            ub2 = np.minimum(ub2, np.minimum(dmin, f1 * (1 + 1e-6)))
        for sname, fn in shapes.items():
            wv = fn(w, h)
            A = acc[sname]
            valid = wv >= 0
            A["lanes"] += wv.size
            A["pix"] += npix
            safe = np.where(valid, wv, 0)
            wpx = np.where(valid, px[safe], np.nan)
            wpy = np.where(valid, py[safe], np.nan)
            bx0, bx1 = np.nanmin(wpx, 1), np.nanmax(wpx, 1)
            by0, by1 = np.nanmin(wpy, 1), np.nanmax(wpy, 1)
            for (ng, ax, ay, r, cand, U, Ucarr, Uc8, first) in per_chunk:
                cw = cand[safe] & valid[:, :, None]              # waves x 64 x ng
                union = cw.any(axis=1)                           # waves x ng
                ddx = np.maximum(np.maximum(bx0[:, None] - ax[None, :], ax[None, :] - bx1[:, None]), 0)
                ddy = np.maximum(np.maximum(by0[:, None] - ay[None, :], ay[None, :] - by1[:, None]), 0)
                near = np.sqrt(ddx ** 2 + ddy ** 2)
                def surv(Upix):
                    um = np.where(valid, Upix[safe], 0).max(axis=1)
                    return (near <= um[:, None] + r[None, :]).sum()
                A["wtc"] += wv.shape[0]
                A["groups"] += wv.shape[0] * ng
                A["union"] += int(union.sum())
                A["nopair"] += int((~union.any(axis=1)).sum())
                A["p1"] += int(surv(U))
                A["carr"] += int(surv(Ucarr))
                A["c8"] += int(surv(Uc8))
                if first:
                    A["first"] += wv.shape[0]
                    A["first_union"] += int(union.sum())
                    A["first_groups"] += wv.shape[0] * ng
    print(f"{sys.argv[1]}: {n} glyphs; groups per wave and chunk (mean), by footprint")
    print(f"  {'shape':6s} {'lane use':>8s} {'waves x chunks':>14s} {'all':>6s} {'union':>6s} {'cull_p1':>8s} {'cull_carr':>9s} {'cull_c8':>8s} {'no pair':>8s} {'1st-chunk share':>15s} {'union in 1st':>12s}")
    for s, A in acc.items():
        k = max(A["wtc"], 1)
        print(f"  {s:6s} {A['pix'] / A['lanes']:8.3f} {A['wtc']:14d} {A['groups'] / k:6.1f} {A['union'] / k:6.1f} {A['p1'] / k:8.1f} "
              f"{A['carr'] / k:9.1f} {A['c8'] / k:8.1f} {A['nopair'] / k:8.3f} {A['first_groups'] / max(A['groups'], 1):15.2f} {A['first_union'] / max(A['first'], 1):12.1f}")


if __name__ == "__main__":
    main()
