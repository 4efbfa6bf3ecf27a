#!/usr/bin/env python3
"""Measurement helper: host API calls and kernel starts of a window of a `rocprofv3 --kernel-trace --hip-runtime-trace
--output-format csv` run on one clock (when was a call enqueued, when did its kernels start).
usage: python3 tools/api_trace.py <dir with the csv files> [fraction_into_run=0.8] [window_us=250]"""
import csv, glob, sys
d = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.8
win = float(sys.argv[3]) if len(sys.argv) > 3 else 250.0
kt = sorted(csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])), key=lambda r: int(r["Start_Timestamp"]))
api = sorted(csv.DictReader(open(glob.glob(d + "/**/*hip_api_trace.csv", recursive=True)[0])), key=lambda r: int(r["Start_Timestamp"]))
t_begin, t_end = int(kt[0]["Start_Timestamp"]), int(kt[-1]["End_Timestamp"])
t0 = t_begin + int((t_end - t_begin) * frac)
ev = []
for r in kt:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 <= s <= t0 + win * 1e3:
        ev.append((s, "K %7.1f q%s %s" % ((e - s) / 1e3, r.get("Queue_Id"), r["Kernel_Name"][:44])))
for r in api:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 <= s <= t0 + win * 1e3:
        ev.append((s, "    api %6.1f %s" % ((e - s) / 1e3, r["Function"])))
for s, txt in sorted(ev):
    print("%8.1f %s" % ((s - t0) / 1e3, txt))
