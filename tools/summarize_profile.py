#!/usr/bin/env python3
"""Condenses a tools/profile.sh output directory into profiles/<tag>_summary.md (+ the raw
kernel_stats.csv) and updates profiles/traffic.json (HBM bytes per launch from PMC)."""
import csv
import glob
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag, workload, variant = sys.argv[1], sys.argv[2], sys.argv[3]
src = ROOT / "gpurun_out" / f"prof_{tag}"
dst = ROOT / "profiles"
dst.mkdir(exist_ok=True)


def rows(pattern):
    out = []
    for f in glob.glob(str(src / pattern), recursive=True):
        with open(f) as fh:
            out += list(csv.DictReader(fh))
    return out


lines = [f"# rocprofv3 summary — {tag} ({workload}, kernel variant {variant})", ""]
stats = rows("trace/**/*kernel_stats.csv")
lines += ["## --kernel-trace --stats", "", "| kernel | calls | avg ns | min ns | max ns | % |", "|---|---|---|---|---|---|"]
for r in stats:
    lines.append(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {r['Percentage']} |")
for f in glob.glob(str(src / "trace/**/*kernel_stats.csv"), recursive=True):
    (dst / f"{tag}_kernel_stats.csv").write_text(Path(f).read_text())
tr = rows("trace/**/*kernel_trace.csv")
if tr:
    r = tr[-1]
    lines += ["", f"dispatch: grid {r['Grid_Size_X']} x wg {r['Workgroup_Size_X']}, VGPR {r['VGPR_Count']}, "
              f"SGPR {r['SGPR_Count']}, LDS {r['LDS_Block_Size']} B, scratch {r['Scratch_Size']}"]


def pmc(sub):
    acc = {}
    for r in rows(f"{sub}/**/*counter_collection.csv"):
        if "sdf_tiles" not in r.get("Kernel_Name", ""):
            continue
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


lines += ["", "## PMC (per launch, mean over dispatches; separate passes)", ""]
allc = {}
for sub in ("fetch", "write", "sq", "sq2"):
    allc.update(pmc(sub))
for k, v in sorted(allc.items()):
    lines.append(f"* {k} = {v:.6g}")
traffic = None
if "FETCH_SIZE" in allc and "WRITE_SIZE" in allc:
    # MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reads 1/2
    # of the bytes of a coalesced stream (128-B requests tallied at 64 B) -> double it.
    traffic = (2.0 * allc["FETCH_SIZE"] + allc["WRITE_SIZE"]) * 1024.0
    lines += ["", f"HBM traffic per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 = **{traffic:.4g} B** "
              "(gfx950 correction: FETCH_SIZE counts 64 B per 128-B request; WRITE_SIZE exact for streaming stores; "
              "our 8-B-per-lane loads are not separately calibrated)"]
    tj = dst / "traffic.json"
    d = json.loads(tj.read_text()) if tj.exists() else {}
    d[f"{workload}:{variant}"] = {"bytes_per_launch": traffic, "fetch_kib": allc["FETCH_SIZE"],
                                  "write_kib": allc["WRITE_SIZE"], "profile": f"profiles/{tag}_summary.md"}
    tj.write_text(json.dumps(d, indent=1, sort_keys=True) + "\n")
(dst / f"{tag}_summary.md").write_text("\n".join(lines) + "\n")
print("\n".join(lines))
