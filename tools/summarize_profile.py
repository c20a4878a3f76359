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
    """rows of the NEWEST file that matches (gpurun merges every run into the same directory)"""
    import os
    files = sorted(glob.glob(str(src / pattern), recursive=True), key=os.path.getmtime)
    if not files:
        return []
    with open(files[-1]) as fh:
        return list(csv.DictReader(fh))


lines = [f"# rocprofv3 summary — {tag} ({workload}, kernel variant {variant})", ""]
stats = rows("trace/**/*kernel_stats.csv")
lines += ["## --kernel-trace --stats", "", "| kernel | calls | avg ns | min ns | max ns | % |", "|---|---|---|---|---|---|"]
for r in stats:
    lines.append(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {r['Percentage']} |")
import os
for f in sorted(glob.glob(str(src / "trace/**/*kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1:]:
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
for sub in ("fetch", "write", "sq", "sq2", "sq3", "sq4"):
    allc.update(pmc(sub))
for k, v in sorted(allc.items()):
    lines.append(f"* {k} = {v:.6g}")
if "SQ_LDS_BANK_CONFLICT" in allc and allc.get("SQ_ACTIVE_INST_LDS"):
    lines += ["", f"LDS bank conflicts: SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS = **{allc['SQ_LDS_BANK_CONFLICT'] / allc['SQ_ACTIVE_INST_LDS']:.2f}** "
              "of the LDS-active cycles"]
if "GRBM_GUI_ACTIVE" in allc:
    lines += ["", f"GRBM_GUI_ACTIVE / 8 XCDs = {allc['GRBM_GUI_ACTIVE'] / 8:.0f} cycles per launch (MI355X_MICROARCH.md, DVFS give-back: / kernel wall "
              "time = effective clock; reads high on dispatches this short — the clock a VALU-bound kernel holds is measured by "
              "tools/ubench/clock_probe.hip)"]
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
    import hashlib
    hsh = hashlib.sha256()
    for f in ("sdf_kernels.hip", "sdf_span_kernel.inc", "sdf_kernels.h"):
        hsh.update((ROOT / "versatiles-glyphs-rs_amd" / "csrc" / f).read_bytes())
    ent = {"bytes_per_launch": traffic, "fetch_kib": allc["FETCH_SIZE"],
           "write_kib": allc["WRITE_SIZE"], "profile": f"profiles/{tag}_summary.md",
           "kernel_source_sha256": hsh.hexdigest()}  # bench.py reports these numbers only for the same sources
    main = [r for r in stats if "sdf_tiles" in r["Name"]]
    if main and all(k in allc for k in ("SQ_INSTS_VALU", "SQ_WAVES")):
        # VALU issue rate per SIMD against the measured full-rate issue of gfx950 (tools/ubench/valu_rate.hip:
        # v_fma/v_mul/v_sub_f32 2.5 cycles per wave64 instruction at 2.4 GHz = 0.96 G inst/s/SIMD; min/max/med3,
        # f64 and transcendental instructions issue slower, so a mixed stream cannot reach 1.0)
        avg_ns = max(float(r["AverageNs"]) for r in main)
        ent["valu_insts_per_wave"] = allc["SQ_INSTS_VALU"] / allc["SQ_WAVES"]
        ent["valu_ginst_per_s_per_simd"] = allc["SQ_INSTS_VALU"] / 1024.0 / avg_ns
        ent["valu_issue_frac"] = ent["valu_ginst_per_s_per_simd"] / 0.96
        ent["kernel_ms"] = avg_ns * 1e-6
        # Issue-cost model: every VALU instruction occupies its SIMD's issue port for a measured number of cycles
        # (tools/ubench/valu_rate*.hip, profiles/r2_ubench.txt, at 2.4 GHz): f32 add/mul/fma 2.5, f32
        # transcendental 8.5, f64 add/mul/fma 4.6-5.2 (4.9), f64 transcendental 16.3, everything else (integer,
        # min/max, compare, select, convert, shifts, DPP moves) 4.2-5.3 (4.4).  The sum over the kernel's dynamic mix
        # is the time the VALU ports are NECESSARILY busy; divided by 1024 SIMDs x kernel cycles it is the
        # fraction of the VALU issue roofline the launch reaches.  (SQ_ACTIVE_INST_VALU is no busy-time counter on
        # this part: it reads one unit per instruction whatever the instruction, checked with pure v_pk_fma_f32 and
        # v_med3_u32 streams.)
        cls = {k: allc.get("SQ_INSTS_VALU_" + k) for k in ("ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "ADD_F64", "MUL_F64",
                                                            "FMA_F64", "TRANS_F64", "INT32", "INT64", "CVT")}
        if all(v is not None for v in cls.values()):
            f32 = cls["ADD_F32"] + cls["MUL_F32"] + cls["FMA_F32"]
            f64 = cls["ADD_F64"] + cls["MUL_F64"] + cls["FMA_F64"]
            other = allc["SQ_INSTS_VALU"] - f32 - f64 - cls["TRANS_F32"] - cls["TRANS_F64"]
            cost = 2.5 * f32 + 8.5 * cls["TRANS_F32"] + 4.9 * f64 + 16.3 * cls["TRANS_F64"] + 4.4 * other
            ent["valu_mix"] = {"f32_add_mul_fma": f32, "f32_trans": cls["TRANS_F32"], "f64_add_mul_fma": f64,
                               "f64_trans": cls["TRANS_F64"], "int32": cls["INT32"], "int64": cls["INT64"], "cvt": cls["CVT"],
                               "other_incl_int_cvt": other, "total": allc["SQ_INSTS_VALU"]}
            ent["valu_issue_cycles_per_simd"] = cost / 1024.0
            ent["valu_model_cycles_per_inst"] = cost / allc["SQ_INSTS_VALU"]
            ent["valu_busy_frac"] = cost / (1024.0 * avg_ns * 2.4)
            ent["valu_cycles_per_inst"] = 1024.0 * avg_ns * 2.4 / allc["SQ_INSTS_VALU"]
            lines += ["", "VALU instruction mix per launch: " + ", ".join(f"{k} {v:.4g}" for k, v in ent["valu_mix"].items()),
                      "", f"VALU issue roofline: sum of measured issue costs {cost / 1024.0:.4g} cycles per SIMD "
                      f"({ent['valu_model_cycles_per_inst']:.2f} per instruction) / ({avg_ns:.0f} ns x 2.4 GHz) = "
                      f"**{ent['valu_busy_frac']:.2f}**; the launch spends {ent['valu_cycles_per_inst']:.2f} SIMD cycles per VALU instruction"]
        lines += ["", f"VALU issue: {allc['SQ_INSTS_VALU']:.4g} instructions / 1024 SIMDs / {avg_ns:.0f} ns = "
                  f"**{ent['valu_ginst_per_s_per_simd']:.3f} G inst/s/SIMD** = {ent['valu_issue_frac']:.2f} of the measured "
                  f"full-rate f32 issue (0.96); {ent['valu_insts_per_wave']:.0f} VALU instructions per wave"]
    d[f"{workload}:{variant}"] = ent
    tj.write_text(json.dumps(d, indent=1, sort_keys=True) + "\n")
(dst / f"{tag}_summary.md").write_text("\n".join(lines) + "\n")
print("\n".join(lines))
