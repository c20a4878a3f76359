#!/usr/bin/env python3
"""Timeline of the LAST end-to-end render in a tools/profile_front_end.sh trace: start (relative to the first activity of
that render), duration and gap of every copy and kernel, in device time.   python tools/front_end_timeline.py <tag>"""
import csv
import glob
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
src = ROOT / "gpurun_out" / f"prof_{sys.argv[1]}_front_end"
ev = []
for pat, name_key in (("trace/**/*kernel_trace.csv", "Kernel_Name"), ("trace/**/*memory_copy_trace.csv", "Direction")):
    files = sorted(glob.glob(str(src / pat), recursive=True), key=os.path.getmtime)
    if files:
        for r in csv.DictReader(open(files[-1])):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[name_key][:60]))
ev.sort()
# the last render = the events after the last gap of more than 200 us
start = 0
for i in range(1, len(ev)):
    if ev[i][0] - ev[i - 1][1] > 200_000:
        start = i
run = ev[start:]
t0 = run[0][0]
prev_end = t0
print(f"{'start us':>9} {'dur us':>8} {'gap us':>7}  activity")
for s, e, n in run:
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:7.1f}  {n}")
    prev_end = max(prev_end, e)
# device busy = union of the intervals (two groups in flight overlap on two streams)
busy, cur_s, cur_e = 0, None, None
for s_, e_, _ in run:
    if cur_e is None or s_ > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s_, e_
    else:
        cur_e = max(cur_e, e_)
busy += (cur_e - cur_s) if cur_e is not None else 0
span = prev_end - t0
print(f"span {span / 1e3:.1f} us, sum of durations {sum(e - s for s, e, _ in run) / 1e3:.1f} us, device busy (union) {busy / 1e3:.1f} us = "
      f"{100.0 * busy / span:.0f} % of the span (the span starts at the first device activity: the host's recording in front of it is not in it)")
