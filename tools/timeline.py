#!/usr/bin/env python3
"""Measurement helper: per-kernel averages and the timeline of the last calls from a rocprofv3
results .db (rocpd sqlite).  usage: python tools/timeline.py results.db [last_n]"""
import collections
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end from kernels order by start"))
last_n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
agg = collections.OrderedDict()
for n, s, e in rows[len(rows) // 2:]:
    agg.setdefault(n[:90], []).append((e - s) / 1e3)
for n, v in agg.items():
    print(f"{n:90s} {len(v):4d} {sum(v) / len(v):8.1f} us")
prev = None
t0 = rows[-last_n][1]
for n, s, e in rows[-last_n:]:
    print(f"{(s - t0) / 1e3:8.1f} dur {(e - s) / 1e3:8.1f} gap {((s - prev) / 1e3 if prev else 0):6.1f} {n[:70]}")
    prev = e
