"""Kernel experiments on the GPU box (development aid; bench.py is the contract).

    VGSDF_LIB=versatiles-glyphs-rs_amd/build/dev/libvgsdf.so python tools/kexp.py noto_regular 0 58 [--rep 8]

For every listed kernel variant: parity of every glyph bitmap against the committed golden SHA-256s
(tests/golden/glyphs_<workload>.csv), ms per launch of the resident batch (HIP events on the launch
stream), and ms per replica with the batch replicated --rep times (steady state, no launch tail).
Variant 58 (dev build) prints the in-kernel phase shares instead.  The batch comes from the product's
own host stage (FontManager.build_batch); nothing here touches the oracle."""
import argparse
import csv
import hashlib
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from conftest import FIRA, NOTO, load_product, noto_files  # noqa: E402

WORK = {"fira": ("Fira Sans Regular", [FIRA]), "noto_regular": ("Noto Sans Regular", [NOTO]),
        "noto_all": ("Noto Sans Regular", None)}


def replicate(vg, b, k):
    if k == 1:
        return b
    n, s = b.n_glyphs, int(b.seg_off[-1])
    seg_off = np.concatenate([b.seg_off[:-1].astype(np.uint64) + r * s for r in range(k)] + [[k * s]]).astype(np.uint32)
    px = int(b.out_off[-1])
    out_off = np.concatenate([b.out_off[:-1] + np.uint64(r * px) for r in range(k)] + [[np.uint64(k * px)]]).astype(np.uint64)
    t = lambda a: np.ascontiguousarray(np.tile(a, k))  # noqa: E731
    return type(b)(seg_off, t(b.seg_sx), t(b.seg_sy), t(b.seg_ex), t(b.seg_ey), t(b.x0), t(b.y0), t(b.w), t(b.h), out_off)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", choices=sorted(WORK))
    ap.add_argument("variants", type=int, nargs="+")
    ap.add_argument("--rep", type=int, default=8)
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    vg = load_product()
    name, files = WORK[args.workload]
    files = files or noto_files()
    mgr = vg.FontManager(True)
    fid = mgr.add_font_with_name(name, files)
    hb = mgr.build_batch(fid)
    b = hb.batch
    golden = {}
    with open(ROOT / "tests" / "golden" / f"glyphs_{args.workload}.csv") as fh:
        for r in csv.DictReader(fh):
            if int(r["bitmap_size"]):
                golden[int(r["codepoint"])] = r["sha256"]
    ids = hb.ids
    ctx = vg.SdfContext(0)
    print(f"{args.workload}: {b.n_glyphs} glyphs, {int(b.seg_off[-1])} segments, {b.out_bytes} px; lib {vg.lib_path()}", flush=True)
    for v in args.variants:
        ctx.set_variant(v)
        db = ctx.upload(b)
        db.launch()
        out = db.download()
        if v == 58:
            db.free()
            continue
        bad = 0
        for g in range(b.n_glyphs):
            a, e = int(b.out_off[g]), int(b.out_off[g + 1])
            if e > a and hashlib.sha256(out[a:e].tobytes()).hexdigest() != golden.get(int(ids[g])):
                bad += 1
        db.time(5)
        ms1 = min(db.time(args.iters) / args.iters for _ in range(3))
        st = db.stats()
        db.free()
        line = f"variant {v:3d}: parity {'OK' if bad == 0 else f'{bad} GLYPHS DIFFER'}  {ms1 * 1e3:8.2f} us/launch ({st['n_tiles']} workgroups)"
        if args.rep > 1:
            rb = replicate(vg, b, args.rep)
            db = ctx.upload(rb)
            db.time(3)
            msr = min(db.time(10) / 10 for _ in range(3)) / args.rep
            db.free()
            line += f"  {msr * 1e3:8.2f} us/replica at x{args.rep}"
        print(line, flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
