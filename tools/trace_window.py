#!/usr/bin/env python3
"""Measurement helper: kernels of a time window of a `rocprofv3 --kernel-trace --output-format csv` run, with stream /
queue ids, start offsets and durations (to see what overlaps what).  usage: python tools/trace_window.py kernel_trace.csv
[fraction_into_run=0.7] [window_us=600]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.7
win = float(sys.argv[3]) if len(sys.argv) > 3 else 600.0
t_begin, t_end = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
t0 = t_begin + int((t_end - t_begin) * frac)
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < t0 or s > t0 + win * 1e3:
        continue
    print(f"{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:7.1f} us  q{r.get('Queue_Id', '?'):>3} s{r.get('Stream_Id', '?'):>3}  {r['Kernel_Name'][:60]}")
