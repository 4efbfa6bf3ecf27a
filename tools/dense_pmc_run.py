#!/usr/bin/env python3
"""Measurement helper: blocking dense searches with rotating query batches over a resident matrix -- the program the
PMC passes of the headline's full-pass kernel run (tools/prof_pmc.sh <out> dense8_body -- python3 tools/dense_pmc_run.py).
usage: N=10000000 D=128 NQ=32 REPS=24 [OPTS=name=v,...] [ASYNC=1] python3 tools/dense_pmc_run.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smqtk_indexing_amd import _lib

n, d, nq, k = int(os.environ.get("N", 10_000_000)), int(os.environ.get("D", 128)), int(os.environ.get("NQ", 32)), int(os.environ.get("K", 100))
reps = int(os.environ.get("REPS", 24))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(3)
db = torch.empty((n, d), dtype=torch.float32, device=dev)
for s in range(0, n, 1 << 20):
    db[s:s + (1 << 20)].normal_(generator=g)
qs = [torch.empty((nq, d), dtype=torch.float32, device=dev).normal_(generator=g) for _ in range(8)]
idx = _lib.DenseIndex(db.data_ptr(), n=n, d=d, device_ptr=True, keepalive=db)
for item in os.environ.get("OPTS", "").split(","):
    if item:
        a, b = item.split("=")
        idx.set_option(a, int(b))
st = torch.cuda.current_stream().cuda_stream
outs = [(torch.empty((nq, k), dtype=torch.float32, device=dev), torch.empty((nq, k), dtype=torch.int64, device=dev)) for _ in range(4)]
use_async = os.environ.get("ASYNC", "0") == "1"
if use_async:
    idx.set_option("dense_async_depth", 3)
for i in range(reps):
    od, oi = outs[i % 4]
    fn = idx.search_device_async if use_async else idx.search_device
    fn(qs[i % 8].data_ptr(), nq, k, od.data_ptr(), oi.data_ptr(), st)
if use_async:
    idx.sync()
torch.cuda.synchronize()
print("cands/q", idx.stats()["candidates"] / nq, "bytes", idx.stats()["bytes_scanned"], file=sys.stderr)
