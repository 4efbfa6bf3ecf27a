#!/usr/bin/env python3
"""Measurement helper: when each pipelined search call's head (prep, sample pass, threshold), full pass, re-rank and select
END, on one clock (SQ_TRACE=1: the library prints hipEvent times of every profiled call relative to the first one).
usage: SQ_TRACE=1 python3 tools/call_trace.py   (env: N rows, DEPTH calls in flight).  The five event records per call cost
~10 us of host time: the traced step is slower than the untraced one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smqtk_indexing_amd import _lib
n, d, nq, k = int(os.environ.get("N", 1250000)), 128, 32, 100
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(3)
db = torch.empty((n, d), dtype=torch.float32, device=dev).normal_(generator=g)
q = torch.empty((nq, d), dtype=torch.float32, device=dev).normal_(generator=g)
idx = _lib.DenseIndex(db.data_ptr(), n=n, d=d, device_ptr=True, keepalive=db)
st = torch.cuda.current_stream().cuda_stream
od = [torch.empty((nq, k), dtype=torch.float32, device=dev) for _ in range(6)]
oi = [torch.empty((nq, k), dtype=torch.int64, device=dev) for _ in range(6)]
_lib.set_option("dense_async_depth", int(os.environ.get("DEPTH", 3)))
for i in range(200):
    idx.search_device_async(q.data_ptr(), nq, k, od[i % 6].data_ptr(), oi[i % 6].data_ptr(), st)
idx.sync()
_lib.set_option("profile", 1)
os.environ["SQ_TRACE"] = "1"
t0 = time.perf_counter()
for i in range(40):
    idx.search_device_async(q.data_ptr(), nq, k, od[i % 6].data_ptr(), oi[i % 6].data_ptr(), st)
idx.sync()
print("step us with events", (time.perf_counter() - t0) / 40 * 1e6)
