#!/usr/bin/env python3
"""Condenses a tools/profile_front_end.sh output directory into profiles/<tag>_front_end_summary.md."""
import csv
import glob
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1]
src = ROOT / "gpurun_out" / f"prof_{tag}_front_end"
rows = []
import os
for pat in ("trace/**/*kernel_stats.csv", "trace/**/*memory_copy_stats.csv"):
    files = sorted(glob.glob(str(src / pat), recursive=True), key=os.path.getmtime)
    if files:  # (gpurun merges every run into the same directory: take the newest)
        rows += list(csv.DictReader(open(files[-1])))
front = ("glyf_decode", "outline_context", "outline_count", "outline_rings", "outline_plan", "outline_emit_segments", "sdf_chunk_boxes")
lines = [f"# rocprofv3 --kernel-trace --memory-copy-trace --stats — end-to-end run with the device front-end "
         f"(tools/e2e_time.py noto_regular fe), {tag}", "",
         "Noto Sans Regular, 2973 rasterised glyphs per call — 3993 `glyf` parts in 122-128 k command slots with the device's glyf decoder "
         "(99 k outline commands when the host records them); 8 calls (4 warm runs x 2 thread settings).", "",
         "| kernel / copy | calls | avg us |", "|---|---|---|"]
total = 0.0
for r in sorted(rows, key=lambda r: -float(r["AverageNs"])):
    lines.append(f"| `{r['Name'][:72]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} |")
    if any(k in r["Name"] for k in front):
        total += float(r["AverageNs"]) / 1e3
lines += ["", f"sum of the per-call averages of the front-end kernels (glyf decode, context, count, rings, plan, emit_segments, chunk boxes): "
          f"**{total:.0f} us** (end of round 1: 180 us; thread-per-command flattening earlier in round 2: 125 us)", ""]
log = (src / "e2e.log").read_text().splitlines()
lines += ["Phase times of the same runs (`vg_timings`, ms, best of 4):", "", "```"] + [l for l in log if l.startswith("noto_regular")] + ["```"]
(ROOT / "profiles" / f"{tag}_front_end_summary.md").write_text("\n".join(lines) + "\n")
print("\n".join(lines))
