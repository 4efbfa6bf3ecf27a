#!/usr/bin/env python3
"""Measurement helper (not part of the product): run the dense search for one
(N, nq) under rocprofv3 --kernel-trace --stats to get the per-kernel breakdown.
usage: N=1250000 NQ=256 python3 tools/breakdown.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smqtk_indexing_amd import _lib

n, d, k, nq = int(os.environ.get("N", 10_000_000)), int(os.environ.get("D", 128)), int(os.environ.get("K", 100)), int(os.environ.get("NQ", 1024))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(3)
db = torch.empty((n, d), dtype=torch.float32, device=dev).normal_(generator=g)
_lib.set_option("profile", 1)
_lib.set_option("force_fallback", int(os.environ.get("FORCE_FB", 0)))
_lib.set_option("sample_stride", int(os.environ.get("SAMPLE_STRIDE", 0)))
_lib.set_option("dense_debug", int(os.environ.get("DEBUG", 0)))   # ablation bits (results are garbage then)
metric = _lib.SQ_METRIC_COSINE if os.environ.get("METRIC", "l2") == "cosine" else _lib.SQ_METRIC_L2
idx = _lib.DenseIndex(db.data_ptr(), n=n, d=d, metric=metric, device_ptr=True, keepalive=db)
st = torch.cuda.current_stream().cuda_stream
q = torch.empty((nq, d), dtype=torch.float32, device=dev).normal_(generator=g)
od = torch.empty((nq, k), dtype=torch.float64 if metric == _lib.SQ_METRIC_COSINE else torch.float32, device=dev); oi = torch.empty((nq, k), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
import time
for r in range(int(os.environ.get("REPS", 10))):
    t0 = time.perf_counter()
    idx.search_device(q.data_ptr(), nq, k, od.data_ptr(), oi.data_ptr(), st)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
print(f"last call wall {wall * 1e3:.3f} ms")
s = idx.stats()
print({kk: s[kk] for kk in ("scan_ms", "total_ms", "candidates", "fallback_queries")})
