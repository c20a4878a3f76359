#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X SDF glyph raster (driver contract).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (the SDF raster kernel) over one batch: every glyph
of the workload font, segments already tessellated by the product's C++ host stage and
RESIDENT IN HBM when the timed region starts (BASELINE.json: configs[1], "Noto Sans
Regular, full codepoint set, 1xMI355X").  value = glyphs/s over all ranks.

  default           every rank renders its own replica of the batch: weak scaling, no data-path
                    collective; RCCL only for the barrier, the max-over-ranks time and the
                    {blocks, glyphs, pixels} counter reduce.
  --sharded         ONE font's glyphs are split over the ranks (glyph-level, cost-balanced:
                    FontManager.shard_glyphs; BASELINE.json configs[3]:
                    `--workload noto_all --sharded`): strong scaling, total work fixed.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment (the driver's own command) starts the N ranks
ITSELF, as a child `python -m torch.distributed.run ... bench.py <same arguments>`, before this process has imported torch
or touched a GPU, relays the one JSON line and exits with the child's status (launch_ranks below); with fewer HIP devices
than ranks the ranks share device 0 over gloo (a rehearsal, said in the line).

Exactly K steps are timed (W untimed warm-up launches first); `steady` repeats the measurement over a region of at
least --min-ms (default 50 ms) and is reported beside the headline, never as `value`.  With --gpus N > 1 the same
line also carries config 4 (`sharded_noto_all`: ONE font's glyphs split over the ranks, strong scaling), config 5
(`synthetic_ranges`: rank r renders outlines [8192 r, 8192 (r + 1)) of the 65 536) and `in_library` (rank 0 alone
drives all N devices through the C ABI's one-process form, vg_renderer_new_multi).  Rank 0 prints ONE JSON line.  The CPU
baseline leg (rank 0, N=1 only) times the oracle — the C restatement of the reference algorithm,
oracle/ — on the same tessellated batch on the host cores; the oracle is never on the measured path.
"""
import argparse
import hashlib
import importlib.util
import json
import os
import platform
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share a queue run one after the
# other.  The end-to-end path keeps two groups in flight on two device contexts (2 streams each) and this benchmark holds more
# contexts beside them (resident batches, torch): on 4 queues the two groups landed on ONE queue and ran in sequence (Noto Sans all
# files 8.9 instead of 11.2 M glyphs/s end to end, the 21 fonts 10.9 instead of 13.2 M).  Must be set before the HIP runtime starts;
# the library does the same at load time when nothing has started it yet (vgsdf_device.cpp).  Reported in the line (`config.env`).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

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


def kernel_source_sha256():
    """Identity of the raster kernel's code: the PMC-derived numbers in profiles/traffic.json are only
    reported when they were collected on exactly these sources."""
    h = hashlib.sha256()
    for f in ("sdf_kernels.hip", "sdf_span_kernel.inc", "sdf_kernels.h"):
        h.update((ROOT / "versatiles-glyphs-rs_amd" / "csrc" / f).read_bytes())
    return h.hexdigest()


def pmc_entry(workload, variant):
    """PMC numbers of a separate rocprofv3 run (tools/profile.sh + tools/summarize_profile.py ->
    profiles/traffic.json); None unless they belong to the kernel sources of this build."""
    try:
        d = json.loads((ROOT / "profiles" / "traffic.json").read_text())
        e = d[f"{workload}:{variant}"]
        return e if e.get("kernel_source_sha256") == kernel_source_sha256() else None
    except Exception:
        return None


def cpu_quota():
    """CPUs the container may use at once (cgroup v2 cpu.max = "quota period"), or None without a quota"""
    try:
        q, per = Path("/sys/fs/cgroup/cpu.max").read_text().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:
        return None


def cpu_model():
    try:
        for ln in Path("/proc/cpuinfo").read_text().splitlines():
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return platform.processor() or "unknown"


def steps_for(ctx, db, warmup, min_ms):
    """W warm-up launches, then the number of launches a region of >= min_ms needs (side measurements only: the headline
    times exactly --steps)"""
    for _ in range(max(warmup, 0)):
        db.launch()
    ctx.sync()
    est = db.time(10) / 10  # ms per launch (untimed probe)
    return int(1.15 * min_ms / max(est, 1e-6)) + 1


def in_library(n):
    """fonts -> PBF bytes of Noto Sans all languages by ONE process on n device lanes (BASELINE.json configs[3] in the form a
    host application links: SURVEY.md §8e), beside the same on one device"""
    vg = load_product()
    d4, p4 = workload_files("noto_all")
    mm = vg.FontManager(True)
    mm.add_font_with_name(d4, p4)
    single = vg.Renderer.new_precise(0)
    share = os.environ.get("VG_SHARE_GPU") == "1" or vg.device_count() < n
    lanes = vg.Renderer.new_multi([0] * n if share else list(range(n)))

    def best_of(renderer, k=8):
        mm.render_glyphs(None, renderer)
        best, tm = None, None
        for _ in range(k):
            t0 = time.perf_counter()
            mm.render_glyphs(None, renderer)
            dt = time.perf_counter() - t0
            if best is None or dt < best:
                best, tm = dt, mm.timings()
        return best, tm
    b1, t1 = best_of(single)
    bn, tn = best_of(lanes)
    lane_form = "library default"
    # a directory of fonts (every fixture file a font of its own: the `recurse` case): plenty of (font, block) tasks, the lanes
    # take whole tasks and nothing is merged
    td = ROOT / "testdata"
    mf = vg.FontManager(True)
    for i, p in enumerate([td / "Fira Sans - Regular.ttf"] + sorted((td / "Noto Sans").glob("*.ttf"), key=lambda q: q.name)):
        mf.add_font_with_name(f"Font {i:02d}", [p])
    mm_keep, mm = mm, mf   # (best_of renders `mm`)
    f1, ft1 = best_of(single)
    fn, ftn = best_of(lanes)
    many = {"fonts": 21, "glyphs": ftn["glyphs"], "seconds": fn, "glyphs_per_s": ftn["glyphs"] / fn, "one_device_seconds": f1,
            "one_device_glyphs_per_s": ft1["glyphs"] / f1, "reduced_counters": list(mf.reduced_counters()),
            "note": "the 21 fixture files as 21 fonts: whole (font, block) tasks per lane, no merge"}
    mm = mm_keep
    print(json.dumps({
        "note": "fonts -> PBF bytes (PCIe inclusive, native NULL sink) of Noto Sans all languages by ONE process: "
                "Renderer.new_multi deals whole (font, block) tasks to N device lanes (glyph shards + a merge of partial PBFs when "
                "there are fewer than four non-empty blocks per lane), counters reduced by vgsdf_reduce_counters (RCCL when the "
                "lanes sit on distinct devices)",
        "devices": n, "lanes_share_one_device": share, "seconds": bn, "glyphs_per_s": tn["glyphs"] / bn,
        "counters_reduced_by": lanes.reduce_path(), "lane_form": lane_form,
        "one_device_seconds": b1, "one_device_glyphs_per_s": t1["glyphs"] / b1, "reduced_counters": list(mm.reduced_counters()),
        "phases_s": {k: tn[k] for k in ("tessellate_s", "pack_s", "device_s", "encode_s", "write_s")},
        "many_fonts": many}), flush=True)
    lanes.close()
    single.close()


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _json_line(text):
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    for ln in reversed(lines):
        try:
            return json.loads(ln)
        except ValueError:
            continue
    return None


def launch_ranks(n, argv):
    """`python bench.py --gpus N` as the driver types it, N > 1, no launcher: this process never imports torch and never
    touches a GPU — it starts the ranks as a child (`python -m torch.distributed.run --nproc-per-node N bench.py <argv>`),
    relays their one JSON line and returns the child's status.  The reference is ONE process (manager.rs:81-125); the line's
    `in_library` field is that form (one process, N device lanes), the ranks are the benchmark contract's form.
    Second attempt with the collectives on gloo when the first produced no line; last resort N independent one-device
    processes without a barrier (labelled: a number beats no number on the first multi-GPU lease)."""
    import subprocess
    me = str(Path(__file__).resolve())
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # devices visible: counted in a child, so that this process stays clear of the HIP runtime
    try:
        cp = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=300)
        n_dev = int(cp.stdout.strip().splitlines()[-1])
    except Exception:  # noqa: BLE001
        n_dev = n
    notes = []
    if n_dev < n and env.get("VG_SHARE_GPU") != "1":
        notes.append(f"{n_dev} HIP device(s) for {n} ranks: the ranks share device 0 (VG_SHARE_GPU=1) and the collectives go over gloo — a "
                     "rehearsal of the code path, not a scaling measurement")
        env["VG_SHARE_GPU"] = "1"
    if env.get("VG_SHARE_GPU") == "1":
        env.setdefault("VG_DIST_BACKEND", "gloo")   # RCCL refuses two ranks on one device
    attempts = [dict(env)]
    if env.get("VG_DIST_BACKEND", "nccl") != "gloo":
        attempts.append(dict(env, VG_DIST_BACKEND="gloo"))
    for k, e in enumerate(attempts):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), me] + list(argv)
        try:
            cp = subprocess.run(cmd, env=e, stdout=subprocess.PIPE, text=True, timeout=1500)   # (stderr passes through)
            line, rc = _json_line(cp.stdout), cp.returncode
        except subprocess.TimeoutExpired as ex:
            line, rc = _json_line(ex.stdout.decode() if isinstance(ex.stdout, bytes) else (ex.stdout or "")), 124
        if line is not None:
            line["launcher"] = {"form": "bench.py started its own ranks: python -m torch.distributed.run --nproc-per-node "
                                        f"{n} (no WORLD_SIZE in the environment)", "attempt": k + 1, "devices_visible": n_dev, "notes": notes}
            print(json.dumps(line), flush=True)
            return rc
        notes.append(f"attempt {k + 1} ({e.get('VG_DIST_BACKEND', 'nccl')}) ended with status {rc} and no result line")
        print(f"[bench] {notes[-1]}", file=sys.stderr, flush=True)
    # last resort: one process per device, no rendezvous at all
    procs = []
    for r in range(n):
        e = dict(env, HIP_VISIBLE_DEVICES=str(0 if env.get("VG_SHARE_GPU") == "1" else r))
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "VG_SHARE_GPU"):
            e.pop(k, None)
        a = [x for x in argv]
        i = a.index("--gpus") if "--gpus" in a else -1
        if i >= 0:
            a[i + 1] = "1"
        else:
            a = [x if not x.startswith("--gpus=") else "--gpus=1" for x in a]
        procs.append(subprocess.Popen([sys.executable, me] + a + ["--no-cpu-baseline", "--no-e2e", "--no-configs", "--no-two-in-flight"],
                                      env=e, stdout=subprocess.PIPE, text=True))
    lines = []
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            pr.kill()
            out = ""
        lines.append(_json_line(out or ""))
    good = [ln for ln in lines if ln]
    if not good:
        print("[bench] no rank produced a result", file=sys.stderr, flush=True)
        return 1
    line = good[0]
    line["value"] = sum(ln["value"] for ln in good)
    line["mpixel_sdf_per_s"] = sum(ln["mpixel_sdf_per_s"] for ln in good)
    line["ms_per_step"] = max(ln["ms_per_step"] for ln in good)
    line["n_gpus"] = len(good)
    line["launcher"] = {"form": "LAST RESORT: independent one-device processes, started together, no barrier and no collective; value = "
                                "sum of the processes' own rates", "ranks_with_a_result": len(good), "ranks_asked": n, "notes": notes}
    print(json.dumps(line), flush=True)
    return 0 if len(good) == n else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--min-ms", type=float, default=50.0, help="minimum length of the timed region")
    ap.add_argument("--workload", default="noto_regular", choices=sorted(WORKLOADS))
    ap.add_argument("--sharded", action="store_true", help="split the font's glyphs over the ranks (strong scaling)")
    ap.add_argument("--variant", type=int, default=0, help="kernel variant (0 = default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-two-in-flight", action="store_true",
                    help="skip the side measurement with two batches in flight (profiling runs: overlapping launches "
                         "would distort the per-kernel durations)")
    ap.add_argument("--no-configs", action="store_true", help="skip the side measurements of the other font workloads")
    ap.add_argument("--synthetic-outlines", type=int, default=0, help="outlines per rank for --workload synthetic")
    ap.add_argument("--in-library", type=int, default=0, metavar="N",
                    help="(run by rank 0 of an N > 1 run) ONE process renders Noto Sans all languages on N device lanes "
                         "(vg_renderer_new_multi) and prints its own JSON object")
    args = ap.parse_args()
    if args.in_library:
        return in_library(args.in_library)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and "WORLD_SIZE" not in os.environ:
            sys.exit(launch_ranks(args.gpus, sys.argv[1:]))   # (nothing has touched a GPU yet)
        args.gpus = world

    import torch
    dist = None
    # Rehearsal switches (single-GPU box): VG_DIST_BACKEND=gloo keeps the collectives on the CPU,
    # VG_SHARE_GPU=1 lets several ranks use device 0.  The driver's multi-GPU run uses neither.
    backend = os.environ.get("VG_DIST_BACKEND", "nccl")
    if os.environ.get("VG_SHARE_GPU") == "1":
        local_rank = 0
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    coll_fallback = None
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
                coll_fallback = f"RCCL failed ({type(e).__name__}: {str(e)[:200]}); barrier, max(time) and the counter sum went over gloo"
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
    if synthetic and args.sharded:
        sys.exit("--sharded splits a font; the synthetic outlines are already one index range per rank")
    t0 = time.perf_counter()
    shard_info = None
    if synthetic:
        from versatiles_glyphs_rs_amd import synthetic as S
        n_out = args.synthetic_outlines or SYNTHETIC_PER_RANK
        class _HB:  # same shape as GlyphBatchHost: .batch
            batch = S.make_batch(rank * n_out, n_out)   # rank r renders outlines [r*n, (r+1)*n)
        hb, mgr, fid = _HB, None, None
        args.no_e2e = args.no_configs = True
    else:
        mgr = vg.FontManager(True)
        fid = mgr.add_font_with_name(disp, paths)
        if args.sharded and world > 1:
            owner, est = mgr.shard_glyphs(fid, world)
            loads = [float(est[owner == r].sum()) for r in range(world)]
            shard_info = {"ranks": world, "estimated_cost_max_over_mean": max(loads) / (sum(loads) / world)}
            mgr.set_glyph_shard(rank, world)     # this rank's glyphs only, from here on
        hb = mgr.build_batch(fid)
    host_s = time.perf_counter() - t0

    ctx = vg.SdfContext(local_rank)
    ctx.set_variant(args.variant)
    db = ctx.upload(hb.batch)       # inputs resident in HBM from here on
    st = db.stats()
    # what a step leaves out: making the batch resident (one H2D copy from page-locked staging + sdf_chunk_boxes, once
    # per batch) and fetching the bitmaps (one D2H copy) — wall clock around the synchronous C-ABI calls, best of 5
    prep = None
    if world == 1:
        up, down = [], []
        for _ in range(5):
            t1 = time.perf_counter()
            d_tmp = ctx.upload(hb.batch)
            up.append(time.perf_counter() - t1)
            d_tmp.launch()
            ctx.sync()
            t1 = time.perf_counter()
            d_tmp.download()
            down.append(time.perf_counter() - t1)
            d_tmp.free()
        prep = {"upload_ms": min(up) * 1e3, "download_ms": min(down) * 1e3, "total_ms": (min(up) + min(down)) * 1e3,
                "note": "NOT part of a step: vgsdf_batch_upload (pack + one H2D copy + sdf_chunk_boxes, synchronous) and "
                        "vgsdf_batch_download (one D2H copy of the bitmaps into pageable memory) of the same batch, wall clock, best of 5; "
                        "the end-to-end figures (`e2e`) include all of it and the front-end"}

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(dbatch, k, w):
        """W untimed launches, then EXACTLY k launches between barrier + synchronize on both sides.
        -> (wall seconds, max over ranks; HIP-event ms of this rank's k launches)"""
        for _ in range(max(w, 0)):
            dbatch.launch()
        ctx.sync()
        barrier()
        t0 = time.perf_counter()
        ms = dbatch.time(k)            # HIP events on the launch stream around the k launches; waits for the last
        torch.cuda.synchronize()
        barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            tmax = torch.tensor([el], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
        return el, ms

    def all_sum(vals):
        if dist is None:
            return [int(v) for v in vals]
        c = torch.tensor(vals, dtype=torch.int64, device=coll_dev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        return [int(v) for v in c.tolist()]

    # The long side measurement runs FIRST: a region of >= --min-ms (every rank times the same number of steps).  It also
    # brings the GPU to its sustained clocks — the first launches after an idle period run ~8 % slower (measured: 85.9 us
    # per launch in a cold 20-step region against 79.0 us sustained) — so the headline's W + K launches below measure
    # the kernel the way a renderer that is busy rendering meets it.
    k_long = steps_for(ctx, db, args.warmup, args.min_ms)
    if dist is not None:
        kk = torch.tensor([k_long], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(kk, op=dist.ReduceOp.MAX)
        k_long = int(kk.item())
    steady_elapsed, steady_ms = timed(db, k_long, 0)
    # headline: W untimed warm-up steps, then EXACTLY --steps steps
    steps = args.steps
    elapsed, kernel_ms_total = timed(db, steps, args.warmup)
    counters = all_sum([256, st["n_glyphs"], st["n_pixels"]])   # blocks, glyphs, pixels per step; the only RCCL payload: 24 bytes
    if dist is not None and args.sharded:
        counters[0] = 256                               # one font's blocks, whoever assembles them

    # ---- N > 1: configs 4 and 5 of BASELINE.json, measured by every rank, reported by rank 0 in the same line ----
    multi = {}
    if world > 1 and not args.sharded and not synthetic and not args.no_configs:
        # config 4: Noto Sans all languages, ONE font's glyphs sharded over the ranks (strong scaling)
        d4, p4 = workload_files("noto_all")
        m4 = vg.FontManager(True)
        f4 = m4.add_font_with_name(d4, p4)
        owner, est = m4.shard_glyphs(f4, world)
        loads = [float(est[owner == r].sum()) for r in range(world)]
        m4.set_glyph_shard(rank, world)
        hb4 = m4.build_batch(f4)
        db4 = ctx.upload(hb4.batch)
        st4 = db4.stats()
        k4 = max(10, steps_for(ctx, db4, 5, args.min_ms))
        if dist is not None:
            kk = torch.tensor([k4], dtype=torch.int64, device=coll_dev)
            dist.all_reduce(kk, op=dist.ReduceOp.MAX)
            k4 = int(kk.item())
        el4, ms4 = timed(db4, k4, 0)
        c4 = all_sum([st4["n_glyphs"], st4["n_pixels"], st4["alg_bytes"]])
        multi["sharded_noto_all"] = {
            "config": "BASELINE.json configs[3]: Noto Sans all languages (20 files, one logical font), glyphs sharded over the ranks",
            "scaling": "strong", "steps": k4, "ms_per_step": el4 / k4 * 1e3, "glyphs_per_step_all_gpus": c4[0],
            "glyphs_per_s": c4[0] * k4 / el4, "mpixel_sdf_per_s": c4[1] * k4 / el4 * 1e-6,
            "rank0_kernel_ms": ms4 / k4, "hbm_roofline_frac_all_gpus": c4[2] * k4 / el4 * 1e-9 / (HBM_PEAK_GBS * world),
            "estimated_cost_max_over_mean": max(loads) / (sum(loads) / world),
            "note": "longest-processing-time-first on estimated w*h*N per glyph, identical on every rank; no exchange in the timed "
                    "step (resident raster of the rank's shard); the partial PBFs meet afterwards (dispatch.py: one all-to-all; "
                    "in_library: shared memory)"}
        db4.free()
        del hb4, m4
        # config 5: synthetic stress, rank r renders outlines [8192 r, 8192 (r + 1))
        from versatiles_glyphs_rs_amd import synthetic as S5
        b5 = S5.make_batch(rank * SYNTHETIC_PER_RANK, SYNTHETIC_PER_RANK)
        db5 = ctx.upload(b5)
        st5 = db5.stats()
        k5 = 10
        el5, ms5 = timed(db5, k5, 2)
        c5 = all_sum([st5["n_glyphs"], st5["n_pixels"], st5["alg_bytes"]])
        multi["synthetic_ranges"] = {
            "config": f"BASELINE.json configs[4]: {c5[0]} of the 65 536 synthetic outlines (1024 segments, 70x70 px), "
                      f"{SYNTHETIC_PER_RANK} per rank, rank r = index range [8192 r, 8192 (r + 1))",
            "scaling": "weak", "steps": k5, "ms_per_step": el5 / k5 * 1e3, "glyphs_per_s": c5[0] * k5 / el5,
            "mpixel_sdf_per_s": c5[1] * k5 / el5 * 1e-6, "rank0_kernel_ms": ms5 / k5,
            "hbm_roofline_frac_all_gpus": c5[2] * k5 / el5 * 1e-9 / (HBM_PEAK_GBS * world)}
        db5.free()
        del b5
    if world > 1 and not args.sharded and not synthetic and not args.no_configs:
        # in-library form: ONE process drives all N devices through vg_renderer_new_multi.  It runs as a CHILD of rank 0
        # (this script with --in-library N: its own HIP / RCCL state, and a time limit — a collective that hangs there must
        # not take the headline with it) while the other ranks wait on the CPU (a gloo group, so that no collective kernel
        # spins on their GPUs).
        wait_group = dist.new_group(backend="gloo") if backend != "gloo" else None
        if rank == 0:
            import subprocess
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK",
                                                                     "ROLE_RANK", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
            try:
                cp = subprocess.run([sys.executable, str(Path(__file__).resolve()), "--in-library", str(world)], env=env,
                                    capture_output=True, text=True, timeout=150)
                lines = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
                multi["in_library"] = json.loads(lines[-1]) if lines else {"error": f"exit {cp.returncode}: {cp.stderr[-400:]}"}
            except subprocess.TimeoutExpired:
                multi["in_library"] = {"error": "no result within 150 s (child process killed)"}
            except Exception as e:  # noqa: BLE001  (the line must still be printed)
                multi["in_library"] = {"error": f"{type(e).__name__}: {e}"}
        dist.barrier(group=wait_group) if wait_group is not None else dist.barrier()

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    glyphs_total = counters[1] * steps
    value = glyphs_total / elapsed
    kernel_s = kernel_ms_total * 1e-3 / steps
    alg_gbs = st["alg_bytes"] / kernel_s * 1e-9
    pmc = pmc_entry(args.workload, args.variant) if not args.sharded else None

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
        "scaling": "strong" if args.sharded else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": ("synthetic outlines (SplitMix64 seed 0x5DF61F95), each rank its own index range" if synthetic else
                 "reference testdata font committed in-repo (no network); " +
                 ("the font's glyphs are split over the ranks" if args.sharded else "every rank renders its own replica")),
        "config": {
            "workload": (f"synthetic: {st['n_glyphs']} outlines x 1024 segments, 70x70 px per rank" if synthetic else
                         f"{args.workload}: {disp}, {len(paths)} file(s), all BMP code points"),
            "glyphs_per_step_per_gpu": st["n_glyphs"],
            "glyphs_per_step_all_gpus": counters[1],
            "segments": st["n_segments"],
            "pixels": st["n_pixels"],
            "pair_evals": st["n_pairs"],
            "tiles": st["n_tiles"],
            "kernel_variant": args.variant,
            "parallelism": (f"glyph-level shard x{world}: longest-processing-time-first on estimated w*h*N per glyph, "
                            "identical on every rank (no exchange in the timed step; partial PBFs meet at the block "
                            "owners afterwards)" if args.sharded else
                            f"replica x{world} (glyph batches shard with no exchange)"),
            "collectives": ("none" if world == 1 else f"{backend}: barrier, max(time), sum of 3 counters"),
            "collectives_fallback": coll_fallback,  # not None: RCCL was asked for and did not come up (said loudly here)
            "timed_region_ms": elapsed * 1e3,
            "env": {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")},
        },
        "mpixel_sdf_per_s": counters[2] * steps / elapsed * 1e-6,
        "roofline": {
            "bound": "hbm",
            "achieved": alg_gbs,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": alg_gbs / HBM_PEAK_GBS,
            "traffic": pmc["bytes_per_launch"] if pmc else None,
            "traffic_source": "builder-run rocprofv3 PMC passes, replayed from profiles/traffic.json (same kernel sources)" if pmc else None,
            "kernel": ("sdf_tiles_span<0>: the only kernel of a step (bounded groups over spans of tiles)"
                       if args.variant == 0 else f"variant {args.variant}"),
            "kernel_ms_avg": kernel_s * 1e3,
            "prep_ms": prep,
            "alg_bytes_per_launch": st["alg_bytes"],
            "note": "the path is VALU bound, not HBM bound (f32 bounds/filter per pixel x candidate segment, f64 only "
                    "where the byte is undecided); see `valu`",
        },
        "valu": {
            "insts_per_wave": pmc.get("valu_insts_per_wave") if pmc else None,
            "busy_frac": pmc.get("valu_busy_frac") if pmc else None,
            "cycles_per_inst": pmc.get("valu_cycles_per_inst") if pmc else None,
            "pmc_kernel_ms": pmc.get("kernel_ms") if pmc else None,
            "brute_pairs_per_s": st["n_pairs"] / kernel_s,
            "model_cycles_per_inst": pmc.get("valu_model_cycles_per_inst") if pmc else None,
            "mix": pmc.get("valu_mix") if pmc else None,
            "source": "builder-run profile, replayed: rocprofv3 PMC passes of an earlier run of this command on these kernel sources "
                      "(profiles/traffic.json, keyed by the kernel sources' SHA-256) — not measured by this invocation",
            "note": "VALU issue roofline from the rocprofv3 PMC passes recorded in profiles/traffic.json, reported only "
                    "when that file was collected on the kernel sources of this build (else null): VALU instructions per "
                    "wave; busy_frac = sum over the kernel's dynamic instruction mix (SQ_INSTS_VALU_* classes) of the "
                    "issue cost measured per class (tools/ubench: f32 add/mul/fma 2.5 cycles, f32 transcendental 8.5, f64 "
                    "4.9, f64 transcendental 16.3, integer/min/max/compare/convert 4.4) / (1024 SIMDs x kernel cycles at "
                    "2.4 GHz) = the share of the launch in which the VALU issue ports are necessarily busy; "
                    "cycles_per_inst = SIMD cycles of the launch per VALU instruction, model_cycles_per_inst = the "
                    "same at a fully busy port; brute_pairs_per_s = pixels x segments of the batch / kernel time, the "
                    "rate a brute-force evaluation would need",
        },
        "host_stage_s": host_s,
        "steady": {"steps": k_long, "ms_per_step": steady_elapsed / k_long * 1e3, "glyphs_per_s": counters[1] * k_long / steady_elapsed,
                   "kernel_ms_avg": steady_ms / k_long,
                   "note": f"the same measurement over a region of >= {args.min_ms:g} ms, run BEFORE the headline's warm-up + K steps (it brings "
                           "the GPU to its sustained clocks; reported beside the headline, never as `value`)"},
    }
    out.update(multi)
    if world > 1 and not args.sharded:
        out["scaling_note"] = ("weak, replicas: every rank renders its own copy of the config-2 batch, no data-path collective; `value` = "
                               "N x glyphs of one batch x steps / max-over-ranks time.  The strong-scaling figure of BASELINE.json "
                               "configs[3] (ONE font over N devices) is `strong_scaling_config4` (one rank per GPU, resident raster of each "
                               "rank's glyph shard) and `in_library` (ONE process, N device lanes, fonts -> PBF bytes)")
        if "sharded_noto_all" in multi:
            m4 = multi["sharded_noto_all"]
            out["strong_scaling_config4"] = {k: m4[k] for k in ("glyphs_per_s", "mpixel_sdf_per_s", "ms_per_step", "steps",
                                                                "glyphs_per_step_all_gpus", "estimated_cost_max_over_mean")}
            out["strong_scaling_config4"]["scaling"] = "strong"
        if isinstance(multi.get("in_library"), dict) and "glyphs_per_s" in multi["in_library"]:
            il = multi["in_library"]
            out["in_library_config4"] = {"glyphs_per_s": il["glyphs_per_s"], "one_device_glyphs_per_s": il["one_device_glyphs_per_s"],
                                         "devices": il["devices"], "lanes_share_one_device": il["lanes_share_one_device"],
                                         "counters_reduced_by": il.get("counters_reduced_by")}
    if shard_info:
        out["config"]["shard"] = shard_info

    # ---- two batches in flight (reported beside the headline, never as `value`): the same launches alternating
    # between two contexts = two streams, so the tail of one launch overlaps the ramp of the next — what a caller
    # with more than one batch to render gets (FontManager does this with its two lanes)
    if world == 1 and not args.sharded and not args.no_two_in_flight:
        c2 = vg.SdfContext(local_rank)
        c2.set_variant(args.variant)
        d2 = c2.upload(hb.batch)
        for _ in range(8):
            db.launch(); d2.launch()
        ctx.sync(); c2.sync()
        t2 = time.perf_counter()
        for i in range(k_long):
            (db if i % 2 == 0 else d2).launch()
        ctx.sync(); c2.sync()
        e2 = time.perf_counter() - t2
        out["two_in_flight"] = {
            "ms_per_step": e2 / k_long * 1e3, "glyphs_per_s": st["n_glyphs"] * k_long / e2, "steps": k_long,
            "note": "the timed steps launched alternately on two contexts (two HIP streams, each with its own resident copy "
                    "of the batch and its own output); wall clock around launches + both synchronisations",
        }
        d2.free(); c2.close()

    # ---- CPU baseline: oracle raster on the same tessellated batch, host cores ---------
    if world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        # thread counts tried: the CPU affinity of the process (what the box shows: 256 on the one-GPU boxes) and the CPUs the
        # container's cgroup lets it use at once (16 there) — the faster one is reported, with the count it used
        quota = cpu_quota()
        counts = [O.default_threads()]
        if quota and int(round(quota)) >= 1 and int(round(quota)) not in counts and not os.environ.get("VG_CPU_THREADS"):
            counts.append(int(round(quota)))
        got = db.download()
        sample, n_sample = hb.batch, st["n_glyphs"]
        if synthetic:  # bounded sample: the first 256 outlines of this rank's range (~10 s of CPU work)
            from versatiles_glyphs_rs_amd import synthetic as S
            n_sample = min(256, st["n_glyphs"])
            sample = S.make_batch(rank * (args.synthetic_outlines or SYNTHETIC_PER_RANK), n_sample)
            got = got[:sample.out_bytes]
        best, tried = None, []
        for cores in counts:
            for mode, label in ((O.PRECISE, "±8 px envelope filter"), (O.BRUTE, "all segments")):
                if synthetic and mode == O.BRUTE:
                    continue
                for _ in range(1 if synthetic else 3):
                    ref, secs = O.sdf_render_batch(sample, mode, cores)
                    if not (ref == got).all():
                        out["parity"] = f"MISMATCH vs oracle ({label})"
                    tried.append({"threads": cores, "rule": label, "seconds": secs})
                    if best is None or secs < best[0]:
                        best = (secs, label, mode, cores)
        out.setdefault("parity", "bit-exact vs oracle on the compared sample")
        _, secs1 = O.sdf_render_batch(sample, best[2], 1)   # the reference has --single-thread (recurse.rs:51-53)
        per_count = {str(c): n_sample / min(t["seconds"] for t in tried if t["threads"] == c) for c in counts}
        out["cpu_baseline"] = {
            "value": n_sample / best[0],
            "unit": "glyphs/s",
            "cores": best[3],
            "kind": "port",
            "sample": f"{n_sample} glyphs of the {args.workload} batch, raster only, same tessellated segments, best of "
                      f"3 passes per setting; faster of the oracle's candidate rules ({best[1]}) and of the thread counts tried",
            "seconds": best[0],
            "single_thread": {"value": n_sample / secs1, "seconds": secs1},
            "cpu_model": cpu_model(),
            "cpu_quota": quota,
            "glyphs_per_s_by_threads": per_count,
            "threads_note": f"thread counts tried: CPU affinity of this process ({len(os.sched_getaffinity(0))}) and the cgroup's CPU quota "
                            f"({quota}); `cores` = the count of the faster run (VG_CPU_THREADS fixes one); machine reports "
                            f"{os.cpu_count()} logical CPUs; the GPU path's host pool runs under the same quota",
        }
        out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]

    # ---- end-to-end (fonts -> PBF bytes), reported beside the headline, never as `value` --
    def e2e_of(m, renderer, fe):
        m.set_device_front_end(fe)
        w = vg.DummyWriter()
        m.render_glyphs(w, renderer)  # warm-up with a collecting writer: every block arrives
        n_files, n_bytes = len(w.files), sum(len(v) for v in w.files.values())
        best, tm, all_dt = None, None, []
        for _ in range(15):
            t0 = time.perf_counter()
            m.render_glyphs(None, renderer)  # native NULL sink: no Python callback per block
            dt = time.perf_counter() - t0
            all_dt.append(dt)
            if best is None or dt < best:
                best, tm = dt, m.timings()   # (the phases reported are those of the best run)
        assert tm["pbf_bytes"] == n_bytes, (tm["pbf_bytes"], n_bytes)
        med = sorted(all_dt)[len(all_dt) // 2]
        return {"glyphs_per_s": tm["glyphs"] / best, "seconds": best, "glyphs_per_s_median_of_15": tm["glyphs"] / med, "pbf_files": n_files,
                "pbf_bytes": n_bytes,
                "phases_s": {k: tm[k] for k in ("tessellate_s", "pack_s", "device_s", "encode_s", "write_s")},
                "glyf_decoded_on_device": bool(tm.get("glyf_groups", 0)), "glyf_fallbacks": tm.get("glyf_fallbacks", 0)}

    def cpu_stat():
        try:
            return {k: int(v) for k, v in (l.split() for l in open("/sys/fs/cgroup/cpu.stat").read().splitlines())}
        except OSError:
            return {}

    def sustained_of(m, renderer, seconds=1.0):
        """the same run repeated for `seconds` (several periods of the container's CPU quota): what a long job gets,
        and whether the host pool ran into the quota meanwhile"""
        c0, t0, runs = cpu_stat(), time.perf_counter(), 0
        while time.perf_counter() - t0 < seconds:
            m.render_glyphs(None, renderer)
            runs += 1
        wall = time.perf_counter() - t0
        c1 = cpu_stat()
        d = {k: c1.get(k, 0) - c0.get(k, 0) for k in ("usage_usec", "nr_periods", "nr_throttled")}
        return {"glyphs_per_s": runs * m.timings()["glyphs"] / wall, "runs": runs, "seconds": wall,
                "cpus_used": d["usage_usec"] / 1e6 / wall if c1 else None,
                "quota_periods_throttled": f"{d['nr_throttled']} of {d['nr_periods']}" if c1 else None}

    def cpu_e2e(font_paths, font_id):
        """the oracle's whole path (fonts -> PBF bytes), one task per (font, block) like manager.rs:117-121"""
        from oracle import oracle as O
        fonts = [O.Font(p) for p in font_paths]
        best = None
        counts = [O.default_threads()]
        q = cpu_quota()
        if q and int(round(q)) >= 1 and int(round(q)) not in counts and not os.environ.get("VG_CPU_THREADS"):
            counts.append(int(round(q)))
        for th in counts:
            for mode in (O.PRECISE, O.BRUTE):
                secs, ctr = O.render_all(fonts, font_id, mode, th)
                if best is None or secs < best[0]:
                    best = (secs, ctr, "±8 px envelope filter" if mode == O.PRECISE else "all segments", th)
        return {"glyphs_per_s": best[1]["glyphs"] / best[0], "seconds": best[0], "threads": best[3], "threads_tried": counts,
                "candidate_rule": best[2]}

    r = None
    if world == 1 and not args.no_e2e:
        r = vg.Renderer.new_precise(local_rank)
        out["e2e"] = {"note": "parse -> outline -> (flatten) -> H2D -> kernels -> D2H -> PBF encode, PCIe inclusive, "
                              "best of 15 warm runs; device_front_end = flattening/closing/scale/bbox on the GPU; sustained = the same run "
                              "repeated for 1 s (the best-of figures last less than one period of a container's CPU quota)"}
        for label, fe in (("device_front_end", True), ("host_tessellation", False)):
            out["e2e"][label] = e2e_of(mgr, r, fe)
            t0 = time.perf_counter()
            mgr.render_glyphs(vg.DummyWriter(), r)
            out["e2e"][label]["seconds_with_python_writer"] = time.perf_counter() - t0
        mgr.set_device_front_end(True)
        out["e2e"]["gpu_path_glyphs_per_s"] = out["e2e"]["device_front_end"]["glyphs_per_s"]
        out["e2e"]["device_front_end"]["sustained"] = sustained_of(mgr, r)
        # many small fonts in one manager (every fixture file as its own font): submissions are grouped
        many = vg.FontManager(True)
        td = ROOT / "testdata"
        for i, p in enumerate([td / "Fira Sans - Regular.ttf"] + sorted((td / "Noto Sans").glob("*.ttf"), key=lambda q: q.name)):
            many.add_font_with_name(f"Font {i:02d}", [p])
        many.render_glyphs(None, r)
        best = None
        for _ in range(10):
            t0 = time.perf_counter()
            many.render_glyphs(None, r)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        tm = many.timings()
        out["e2e"]["all_21_fixture_fonts_as_separate_fonts"] = {
            "glyphs_per_s": tm["glyphs"] / best, "seconds": best, "glyphs": tm["glyphs"], "pbf_bytes": tm["pbf_bytes"],
            "phases_s_last_run": {k: tm[k] for k in ("tessellate_s", "pack_s", "device_s", "encode_s", "write_s", "total_s")},
            "sustained": sustained_of(many, r)}
        if not args.no_cpu_baseline:
            c = cpu_e2e(paths, fid)
            out["e2e"]["cpu_port"] = c
            out["e2e"]["gpu_over_cpu"] = out["e2e"]["gpu_path_glyphs_per_s"] / c["glyphs_per_s"]

    # ---- the other font workloads, same measurements, in the same line ---------------------
    if world == 1 and not args.no_configs and not args.sharded:
        out["configs"] = []
        for wl in ("fira", "noto_regular", "noto_all"):
            if wl == args.workload:
                continue
            d2, p2 = workload_files(wl)
            m2 = vg.FontManager(True)
            f2 = m2.add_font_with_name(d2, p2)
            hb2 = m2.build_batch(f2)
            db2 = ctx.upload(hb2.batch)
            k2 = max(20, steps_for(ctx, db2, 5, args.min_ms))
            ms2 = db2.time(k2) / k2
            st2 = db2.stats()
            ent = {"workload": f"{wl}: {d2}, {len(p2)} file(s), all BMP code points", "glyphs": st2["n_glyphs"],
                   "segments": st2["n_segments"], "pixels": st2["n_pixels"], "steps": k2, "kernel_ms": ms2,
                   "glyphs_per_s": st2["n_glyphs"] / (ms2 * 1e-3), "mpixel_sdf_per_s": st2["n_pixels"] / (ms2 * 1e-3) * 1e-6,
                   "roofline_frac_hbm": st2["alg_bytes"] / (ms2 * 1e-3) * 1e-9 / HBM_PEAK_GBS}
            db2.free()
            del hb2
            if not args.no_e2e:
                ent["e2e_glyphs_per_s"] = e2e_of(m2, r, True)["glyphs_per_s"]
                if not args.no_cpu_baseline and wl != "noto_all":  # (the 20-file CPU pass takes several seconds)
                    ent["e2e_cpu_port_glyphs_per_s"] = cpu_e2e(p2, f2)["glyphs_per_s"]
            out["configs"].append(ent)
        # config 5 at one rank's size: outlines [0, 8192) of the 65 536 ("where the roofline is measured")
        from versatiles_glyphs_rs_amd import synthetic as S
        bs = S.make_batch(0, SYNTHETIC_PER_RANK)
        dbs = ctx.upload(bs)
        sts = dbs.stats()
        for _ in range(3):
            dbs.launch()
        ctx.sync()
        ks = 20
        mss = dbs.time(ks) / ks
        ent = {"workload": f"synthetic: outlines [0, {SYNTHETIC_PER_RANK}) of BASELINE.json configs[4] (1024 segments, 70x70 px each)",
               "glyphs": sts["n_glyphs"], "segments": sts["n_segments"], "pixels": sts["n_pixels"], "steps": ks, "kernel_ms": mss,
               "glyphs_per_s": sts["n_glyphs"] / (mss * 1e-3), "mpixel_sdf_per_s": sts["n_pixels"] / (mss * 1e-3) * 1e-6,
               "roofline_frac_hbm": sts["alg_bytes"] / (mss * 1e-3) * 1e-9 / HBM_PEAK_GBS,
               "brute_pairs_per_s": sts["n_pairs"] / (mss * 1e-3)}
        pm = pmc_entry("synthetic", args.variant)
        if pm:
            ent["valu"] = {k: pm.get(k) for k in ("valu_insts_per_wave", "valu_busy_frac", "valu_cycles_per_inst", "kernel_ms")}
        if not args.no_cpu_baseline:  # bounded CPU leg: the first 64 outlines
            from oracle import oracle as O
            smp = S.make_batch(0, 64)
            ref, secs = O.sdf_render_batch(smp, O.PRECISE, O.default_threads())
            got = dbs.download()[:smp.out_bytes]
            ent["parity"] = "bit-exact vs oracle on outlines [0, 64)" if (ref == got).all() else "MISMATCH vs oracle"
            ent["cpu_port_glyphs_per_s"] = 64 / secs
        dbs.free()
        out["configs"].append(ent)

    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
