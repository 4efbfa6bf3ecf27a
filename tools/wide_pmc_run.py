#!/usr/bin/env python3
"""Measurement helper: blocking dense searches over wide rows (the program the PMC passes of dense_wide_scan_kernel run:
tools/prof_pmc.sh <out> dense_wide -- python3 tools/wide_pmc_run.py).  usage: N=1000000 D=4096 NQ=32 REPS=8 python3 tools/wide_pmc_run.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("N", "1000000"); os.environ.setdefault("D", "4096"); os.environ.setdefault("REPS", "8")
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "dense_pmc_run.py")).read())
