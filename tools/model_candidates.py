"""CPU model of the span kernel's candidate selection (development aid, numpy only).

Counts, for a font's tessellated batch (product host stage, no GPU), the (pixel, group) pairs that phase 2 of
sdf_tiles_span evaluates and the number of 64-pair rounds per wave, under alternative bound policies:

  seq      current kernel: U from the anchors of the chunk being processed + the bound carried from earlier chunks
  prepass  U from the anchors of ALL chunks of the glyph before the first chunk is processed
  ideal    U = true distance (lower limit for anchor + radius bounds)

    python tools/model_candidates.py noto_regular [max_glyphs]
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
GRP = int(sys.argv[3]) if len(sys.argv) > 3 else 8
CH = 256


def seg_dist2(px, py, sx, sy, ex, ey):
    """squared distances pixels x segments"""
    dx, dy = ex - sx, ey - sy
    l2 = dx * dx + dy * dy
    pvx = px[:, None] - sx[None, :]
    pvy = py[:, None] - sy[None, :]
    t = (pvx * dx + pvy * dy) / np.where(l2 > 0, l2, 1)
    t = np.clip(t, 0, 1)
    qx = pvx - t * dx
    qy = pvy - t * dy
    return qx * qx + qy * qy


def main():
    vg = load_product()
    name, files = WORK[sys.argv[1]]
    files = files or noto_files()
    mgr = vg.FontManager(True)
    fid = mgr.add_font_with_name(name, files)
    hb = mgr.build_batch(fid)  # keep the owner alive: the arrays are views into its memory
    b = hb.batch
    n = b.n_glyphs if len(sys.argv) < 3 else min(b.n_glyphs, int(sys.argv[2]))
    tot = {k: dict(pairs=0, rounds=0, waves=0, lanes=0) for k in ("seq", "prepass", "ideal")}
    tilechunks = 0
    global Z
    Z = dict(act=0, zero=0, skip=0, skip_wrong=0)
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
        d2 = seg_dist2(px, py, sx, sy, ex, ey)  # npix x nseg
        nseg = e - a
        nch = (nseg + CH - 1) // CH
        M = max(np.abs(np.concatenate([sx, sy, ex, ey])).max(), w, h)
        pad = 0.01 + 1e-5 * M
        # groups
        chunks = []
        for c in range(nch):
            c0, c1 = c * CH, min(nseg, (c + 1) * CH)
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
            chunks.append((c0, c1, ng, D2, r))
        glob_min = np.min(np.concatenate([ch[3] for ch in chunks], axis=1), axis=1)
        true_min = d2.min(axis=1)
        ntile = (npix + 255) // 256
        for mode in tot:
            ub2 = np.full(npix, np.inf) if mode == "seq" else (glob_min.copy() if mode == "prepass" else true_min.copy())
            for (c0, c1, ng, D2, r) in chunks:
                dmin = np.minimum(ub2, D2.min(axis=1)) if mode != "ideal" else ub2
                U = np.minimum((np.sqrt(dmin) * INFL + pad) * INFL, 6.2) * 1.004
                cand = (U[:, None] + r[None, :]) ** 2 - D2 * (1 - 2.0 ** -8) >= 0  # bf16 truncation of D2: up to 2^-7 smaller
                cnt = cand.sum(axis=1)
                # per wave rounds
                padn = ntile * 256
                cw = np.zeros(padn, dtype=np.int64)
                cw[:npix] = cnt
                per_wave = cw.reshape(-1, 64).sum(axis=1)
                if mode == "seq":
                    # wave-level statistics: active waves (with pixels), waves without any pair, waves a box test could skip
                    act = (np.arange(padn).reshape(-1, 64)[:, 0] < npix)
                    Z["act"] += int(act.sum()); Z["zero"] += int(((per_wave == 0) & act).sum())
                    # box test: chunk bbox vs wave pixel bbox with the wave's carried bound (sqrt of max ub2 over its pixels)
                    ubw = np.full(padn, 0.0); ubw[:npix] = np.sqrt(np.minimum(ub2, 38.44)); ubw = ubw.reshape(-1, 64).max(axis=1)
                    bx0, bx1 = min(sx[c0:c1].min(), ex[c0:c1].min()), max(sx[c0:c1].max(), ex[c0:c1].max())
                    by0, by1 = min(sy[c0:c1].min(), ey[c0:c1].min()), max(sy[c0:c1].max(), ey[c0:c1].max())
                    pxw = np.full(padn, np.nan); pxw[:npix] = px; pyw = np.full(padn, np.nan); pyw[:npix] = py
                    pxw = pxw.reshape(-1, 64); pyw = pyw.reshape(-1, 64)
                    with np.errstate(all="ignore"):
                        wx0, wx1, wy0, wy1 = np.nanmin(pxw, 1), np.nanmax(pxw, 1), np.nanmin(pyw, 1), np.nanmax(pyw, 1)
                    ddx = np.maximum(np.maximum(bx0 - wx1, wx0 - bx1), 0); ddy = np.maximum(np.maximum(by0 - wy1, wy0 - by1), 0)
                    skip = act & (np.sqrt(ddx ** 2 + ddy ** 2) > ubw * 1.01 + 0.05)
                    Z["skip"] += int(skip.sum()); Z["skip_wrong"] += int((skip & (per_wave > 0)).sum())
                tot[mode]["pairs"] += int(cnt.sum())
                tot[mode]["rounds"] += int(((per_wave + 63) // 64).sum())
                tot[mode]["waves"] += len(per_wave)
                tot[mode]["lanes"] += npix
                # update carried bound with the chunk's true candidate minimum
                segmask = np.repeat(cand, GRP, axis=1)[:, : c1 - c0]
                f1 = np.where(segmask, d2[:, c0:c1], np.inf).min(axis=1)
                ub2 = np.minimum(ub2, np.minimum(dmin, f1 * (1 + 1e-6)))
            if mode == "seq":
                tilechunks += ntile * len(chunks)
    print(f"{sys.argv[1]}: {n} glyphs, group size {GRP}, {tilechunks} tile-chunks")
    print("  wave tile-chunks with pixels", Z["act"], "without any pair", Z["zero"], "skippable by chunk box vs carried bound", Z["skip"], "(of which wrongly)", Z["skip_wrong"])
    for mode, t in tot.items():
        print(f"  {mode:8s} pairs {t['pairs']:>10d}  per pixel-chunk {t['pairs'] / max(t['lanes'], 1):.2f}  rounds {t['rounds']:>8d} "
              f"per wave-tile-chunk {t['rounds'] / max(t['waves'], 1):.2f}  slot use {t['pairs'] / max(t['rounds'] * 64, 1):.2f}")


if __name__ == "__main__":
    main()
