#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X SDF glyph raster (driver contract).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (the SDF raster kernel) over one batch: every glyph
of the workload font, segments already tessellated by the product's C++ host stage and
RESIDENT IN HBM when the timed region starts (BASELINE.json: configs[1], "Noto Sans
Regular, full codepoint set, 1xMI355X").  value = glyphs/s over all ranks (each rank
renders its own replica of the batch: weak scaling, no data-path collective; RCCL is used
only for the barrier, the max-over-ranks time and the final {blocks, glyphs, pixels}
counter reduce).

Rank 0 prints ONE JSON line.  The CPU baseline leg (rank 0, N=1 only) times the oracle —
the C restatement of the reference algorithm, oracle/ — on the same already tessellated
batch on the host cores; the oracle is never on the measured GPU path.
"""
import argparse
import importlib.util
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

WORKLOADS = {
    # name: (font display name, [files relative to testdata/])
    "noto_regular": ("Noto Sans Regular", ["Noto Sans/Noto Sans - Regular.ttf"]),
    "fira": ("Fira Sans Regular", ["Fira Sans - Regular.ttf"]),
    "noto_all": ("Noto Sans Regular", None),  # all 20 files of testdata/Noto Sans, sorted
    # BASELINE.json configs[4]: 65 536 outlines x 1024 segments at 70x70, split over 8 GPUs ->
    # 8192 outlines per rank (synthetic.py; generated on the host, no font involved)
    "synthetic": ("synthetic stress outlines", []),
}
SYNTHETIC_PER_RANK = 8192


def load_product():
    name = "versatiles_glyphs_rs_amd"
    if name in sys.modules:
        return sys.modules[name]
    pkg = ROOT / "versatiles-glyphs-rs_amd"
    spec = importlib.util.spec_from_file_location(name, pkg / "__init__.py", submodule_search_locations=[str(pkg)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def workload_files(name):
    disp, files = WORKLOADS[name]
    td = ROOT / "testdata"
    if files is None:
        paths = sorted((td / "Noto Sans").glob("*.ttf"), key=lambda p: p.name)
    else:
        paths = [td / f for f in files]
    return disp, paths


def traffic_bytes(workload, variant):
    """HBM bytes per launch measured with rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, gfx950
    correction applied) in a separate profiling run: profiles/traffic.json, written by
    tools/summarize_profile.py.  None when this (workload, variant) has not been profiled."""
    try:
        d = json.loads((ROOT / "profiles" / "traffic.json").read_text())
        return d[f"{workload}:{variant}"]["bytes_per_launch"]
    except Exception:
        return None


def pmc_value(workload, variant, key):
    try:
        d = json.loads((ROOT / "profiles" / "traffic.json").read_text())
        return d[f"{workload}:{variant}"].get(key)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="noto_regular", choices=sorted(WORKLOADS))
    ap.add_argument("--variant", type=int, default=0, help="kernel variant (0 = default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--synthetic-outlines", type=int, default=0, help="outlines per rank for --workload synthetic")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    import torch
    dist = None
    # Rehearsal switches (single-GPU box): VG_DIST_BACKEND=gloo keeps the collectives on the CPU,
    # VG_SHARE_GPU=1 lets several ranks use device 0.  The driver's multi-GPU run uses neither.
    backend = os.environ.get("VG_DIST_BACKEND", "nccl")
    if os.environ.get("VG_SHARE_GPU") == "1":
        local_rank = 0
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        from datetime import timedelta
        if backend == "nccl":
            try:
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank),
                                        timeout=timedelta(seconds=300))
                probe = torch.ones(1, device="cuda")
                dist.all_reduce(probe)          # the collectives below carry 32 bytes in total:
                torch.cuda.synchronize()        # make sure RCCL works before anything is timed
            except Exception as e:              # noqa: BLE001  (a number over gloo beats no number)
                print(f"[bench] RCCL unavailable ({type(e).__name__}: {e}); counters and barriers go over gloo",
                      file=sys.stderr, flush=True)
                try:
                    dist.destroy_process_group()
                except Exception:               # noqa: BLE001
                    pass
                backend, coll_dev = "gloo", "cpu"
                dist.init_process_group(backend="gloo", timeout=timedelta(seconds=300))
        else:
            dist.init_process_group(backend=backend, timeout=timedelta(seconds=300))

    vg = load_product()
    if vg.device_count() <= local_rank:
        sys.exit(f"no HIP device {local_rank}: the product has no CPU fallback")

    # ---- host stage (product C++): fonts -> SoA batch ---------------------------------
    disp, paths = workload_files(args.workload)
    synthetic = args.workload == "synthetic"
    t0 = time.perf_counter()
    if synthetic:
        from versatiles_glyphs_rs_amd import synthetic as S
        n_out = args.synthetic_outlines or SYNTHETIC_PER_RANK
        class _HB:  # same shape as GlyphBatchHost: .batch
            batch = S.make_batch(rank * n_out, n_out)   # rank r renders outlines [r*n, (r+1)*n)
        hb, mgr, fid = _HB, None, None
        args.no_e2e = True
    else:
        mgr = vg.FontManager(True)
        fid = mgr.add_font_with_name(disp, paths)
        hb = mgr.build_batch(fid)
    host_s = time.perf_counter() - t0

    ctx = vg.SdfContext(local_rank)
    ctx.set_variant(args.variant)
    db = ctx.upload(hb.batch)       # inputs resident in HBM from here on
    st = db.stats()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 0)):
        db.launch()
    ctx.sync()

    barrier()
    t0 = time.perf_counter()
    kernel_ms_total = db.time(args.steps)   # HIP events on the launch stream, K launches, waits for the last
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0

    counters = [256, st["n_glyphs"], st["n_pixels"]]   # blocks, glyphs, pixels rendered per step by this rank
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        c = torch.tensor(counters, dtype=torch.int64, device=coll_dev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)        # the only RCCL payload: 24 bytes
        counters = [int(v) for v in c.tolist()]

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    steps = args.steps
    glyphs_total = counters[1] * steps
    value = glyphs_total / elapsed
    kernel_s = kernel_ms_total * 1e-3 / steps
    alg_gbs = st["alg_bytes"] / kernel_s * 1e-9

    out = {
        "metric": "glyphs/sec (SDF raster, Noto Sans Regular full BMP set)" if args.workload == "noto_regular"
                  else f"glyphs/sec (SDF raster, {args.workload})",
        "value": value,
        "unit": "glyphs/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": ("synthetic outlines (SplitMix64 seed 0x5DF61F95), each rank its own index range" if synthetic else
                 "reference testdata font committed in-repo (no network); every rank renders its own replica"),
        "config": {
            "workload": (f"synthetic: {st['n_glyphs']} outlines x 1024 segments, 70x70 px per rank" if synthetic else
                         f"{args.workload}: {disp}, {len(paths)} file(s), all BMP code points"),
            "glyphs_per_step_per_gpu": st["n_glyphs"],
            "segments": st["n_segments"],
            "pixels": st["n_pixels"],
            "pair_evals": st["n_pairs"],
            "tiles": st["n_tiles"],
            "kernel_variant": args.variant,
            "parallelism": f"replica x{world} (glyph batches shard with no exchange)",
            "collectives": ("none" if world == 1 else f"{backend}: barrier, max(time), sum of 3 counters"),
        },
        "mpixel_sdf_per_s": counters[2] * steps / elapsed * 1e-6,
        "roofline": {
            "bound": "hbm",
            "achieved": alg_gbs,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": alg_gbs / HBM_PEAK_GBS,
            "traffic": traffic_bytes(args.workload, args.variant),
            "kernel": ("sdf_tiles_span<0>: the only kernel of a step (bounded groups over spans of tiles)"
                       if args.variant == 0 else f"variant {args.variant}"),
            "kernel_ms_avg": kernel_s * 1e3,
            "alg_bytes_per_launch": st["alg_bytes"],
            "note": "the path is VALU bound, not HBM bound (f32 bounds/filter per pixel x candidate segment, f64 only "
                    "where the byte is undecided); see `valu`",
        },
        "valu": {
            "insts_per_wave": pmc_value(args.workload, args.variant, "valu_insts_per_wave"),
            "ginst_per_s_per_simd": pmc_value(args.workload, args.variant, "valu_ginst_per_s_per_simd"),
            "issue_frac": pmc_value(args.workload, args.variant, "valu_issue_frac"),
            "brute_pairs_per_s": st["n_pairs"] / kernel_s,
            "note": "VALU instruction issue rate per SIMD from the PMC pass recorded in profiles/traffic.json (null if "
                    "this workload/variant was not profiled) against the measured full-rate f32 issue of gfx950, 0.96 G "
                    "inst/s/SIMD (tools/ubench/valu_rate.hip); brute_pairs_per_s = pixels x segments of the batch / "
                    "kernel time, i.e. the rate a brute-force evaluation would need",
        },
        "host_stage_s": host_s,
    }

    # ---- CPU baseline: oracle raster on the same tessellated batch, host cores ---------
    if world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        cores = O.default_threads()
        got = db.download()
        sample, n_sample = hb.batch, st["n_glyphs"]
        if synthetic:  # bounded sample: the first 256 outlines of this rank's range (~10 s of CPU work)
            from versatiles_glyphs_rs_amd import synthetic as S
            n_sample = min(256, st["n_glyphs"])
            sample = S.make_batch(rank * (args.synthetic_outlines or SYNTHETIC_PER_RANK), n_sample)
            got = got[:sample.out_bytes]
        best = None
        for mode, label in ((O.PRECISE, "±8 px envelope filter"), (O.BRUTE, "all segments")):
            if synthetic and mode == O.BRUTE:
                continue
            ref, secs = O.sdf_render_batch(sample, mode, cores)
            if not (ref == got).all():
                out["parity"] = f"MISMATCH vs oracle ({label})"
            if best is None or secs < best[0]:
                best = (secs, label)
        out.setdefault("parity", "bit-exact vs oracle on the compared sample")
        out["cpu_baseline"] = {
            "value": n_sample / best[0],
            "unit": "glyphs/s",
            "cores": cores,
            "kind": "port",
            "sample": f"{n_sample} glyphs of the {args.workload} batch, once, raster only, same tessellated "
                      f"segments; faster of the oracle's candidate rules: {best[1]}",
            "seconds": best[0],
        }
        out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]

    # ---- end-to-end (fonts -> PBF bytes), reported beside the headline, never as `value` --
    if world == 1 and not args.no_e2e:
        r = vg.Renderer.new_precise(local_rank)
        out["e2e"] = {"note": "parse -> outline -> (flatten) -> H2D -> kernels -> D2H -> PBF encode, PCIe inclusive, "
                              "best of 5 warm runs; device_front_end = flattening/closing/scale/bbox on the GPU"}
        for label, fe in (("device_front_end", True), ("host_tessellation", False)):
            mgr.set_device_front_end(fe)
            w = vg.DummyWriter()
            mgr.render_glyphs(w, r)  # warm-up with a collecting writer: every block arrives
            n_files, n_bytes = len(w.files), sum(len(v) for v in w.files.values())
            best = best_py = None
            for _ in range(5):
                t0 = time.perf_counter()
                mgr.render_glyphs(None, r)  # native NULL sink: no Python callback per block
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            tm = mgr.timings()
            assert tm["pbf_bytes"] == n_bytes, (tm["pbf_bytes"], n_bytes)
            for _ in range(3):
                t0 = time.perf_counter()
                mgr.render_glyphs(vg.DummyWriter(), r)
                dt = time.perf_counter() - t0
                best_py = dt if best_py is None else min(best_py, dt)
            out["e2e"][label] = {"glyphs_per_s": tm["glyphs"] / best, "seconds": best,
                                 "seconds_with_python_writer": best_py, "pbf_files": n_files, "pbf_bytes": n_bytes,
                                 "phases_s": {k: tm[k] for k in ("tessellate_s", "pack_s", "device_s", "encode_s", "write_s")}}
        out["e2e"]["gpu_path_glyphs_per_s"] = out["e2e"]["device_front_end"]["glyphs_per_s"]
        # many small fonts in one manager (every fixture file as its own font): submissions are grouped
        many = vg.FontManager(True)
        td = ROOT / "testdata"
        for i, p in enumerate([td / "Fira Sans - Regular.ttf"] + sorted((td / "Noto Sans").glob("*.ttf"), key=lambda q: q.name)):
            many.add_font_with_name(f"Font {i:02d}", [p])
        many.render_glyphs(None, r)
        best = None
        for _ in range(5):
            t0 = time.perf_counter()
            many.render_glyphs(None, r)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        tm = many.timings()
        out["e2e"]["all_21_fixture_fonts_as_separate_fonts"] = {"glyphs_per_s": tm["glyphs"] / best, "seconds": best,
                                                                 "glyphs": tm["glyphs"], "pbf_bytes": tm["pbf_bytes"]}
        if not args.no_cpu_baseline:
            from oracle import oracle as O
            fonts = [O.Font(p) for p in paths]
            secs, ctr = O.render_all(fonts, fid, O.BRUTE, O.default_threads())
            out["e2e"]["cpu_port_glyphs_per_s"] = ctr["glyphs"] / secs
            out["e2e"]["cpu_threads"] = O.default_threads()
            out["e2e"]["gpu_over_cpu"] = out["e2e"]["gpu_path_glyphs_per_s"] / out["e2e"]["cpu_port_glyphs_per_s"]

    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
