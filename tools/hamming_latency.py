#!/usr/bin/env python3
"""Measurement helper: wall time of blocking sq_hamming_search calls (profiling off) on 10 M x 64-bit codes resident on the
device.  usage: python3 tools/hamming_latency.py [n_codes] [bits]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smqtk_indexing_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 64
w = bits // 64
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(5)
codes = torch.randint(-2**63, 2**63 - 1, (n, w), dtype=torch.int64, device=dev, generator=g)
codes = torch.unique(codes, dim=0) if w == 1 else codes          # (sorted unique for one word; wider: random order is fine here)
idx = _lib.HammingIndex(codes.data_ptr(), n=codes.shape[0], words=w, device_ptr=True, keepalive=codes)
st = torch.cuda.current_stream().cuda_stream
strides = [int(x) for x in os.environ.get("STRIDES", "0").split(",")]
nqs = [int(x) for x in os.environ.get("NQS", "1,32,1024").split(",")]
for item in os.environ.get("OPTS", "").split(","):      # library options of the index, name=value,...
    if item:
        a, b = item.split("=")
        idx.set_option(a, int(b))
for nq, stride in [(a, b) for a in nqs for b in strides]:
    _lib.set_option("sample_stride", stride)
    q = codes[torch.randint(0, codes.shape[0], (nq,), device=dev, generator=g)].contiguous() ^ 5
    od = torch.empty((nq, 100), dtype=torch.int32, device=dev)
    oi = torch.empty((nq, 100), dtype=torch.int64, device=dev)
    for _ in range(20):
        idx.search_device(q.data_ptr(), nq, 100, od.data_ptr(), oi.data_ptr(), st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 300 if nq < 1024 else 30
    for _ in range(reps):
        idx.search_device(q.data_ptr(), nq, 100, od.data_ptr(), oi.data_ptr(), st)
    torch.cuda.synchronize()
    print(f"n={codes.shape[0]} bits={bits} nq={nq} sample_stride={stride} {os.environ.get('OPTS', '')}: {(time.perf_counter() - t0) / reps * 1e3:.4f} ms per call, cands/q {idx.stats()['candidates'] / nq:.0f}", flush=True)
