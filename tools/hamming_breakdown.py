#!/usr/bin/env python3
"""Measurement helper: one Hamming search shape under rocprofv3 --kernel-trace (per-kernel breakdown).
usage: N=10000000 W=1 NQ=32 python3 tools/hamming_breakdown.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smqtk_indexing_amd import _lib

n, w, nq, k = int(os.environ.get("N", 10_000_000)), int(os.environ.get("W", 1)), int(os.environ.get("NQ", 32)), int(os.environ.get("K", 100))
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev); g.manual_seed(2)
codes = torch.randint(-2**63, 2**63 - 1, (n, w), dtype=torch.int64, device=dev, generator=g)
idx = _lib.HammingIndex(codes.data_ptr(), n=n, words=w, device_ptr=True, keepalive=codes)
_lib.set_option("profile", int(os.environ.get("PROFILE", 1)))
for item in os.environ.get("OPTS", "").split(","):      # library options of the index, name=value,...
    if item:
        a, b = item.split("=")
        idx.set_option(a, int(b))
q = torch.randint(-2**63, 2**63 - 1, (nq, w), dtype=torch.int64, device=dev, generator=g)
od = torch.empty((nq, k), dtype=torch.int32, device=dev); oi = torch.empty((nq, k), dtype=torch.int64, device=dev)
for r in range(int(os.environ.get("REPS", 10))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    idx.search_device(q.data_ptr(), nq, k, od.data_ptr(), oi.data_ptr(), st)
    torch.cuda.synchronize(); wall = time.perf_counter() - t0
s = idx.stats()
print(f"last call wall {wall * 1e3:.3f} ms", {kk: s[kk] for kk in ("scan_ms", "total_ms", "candidates", "fallback_queries")})
