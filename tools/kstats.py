#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv compactly: name, calls, average us."""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:70]:70s} {r['Calls']:>5s} {float(r['AverageNs'])/1000:10.1f} us  {r['Percentage']:>6s}%")
