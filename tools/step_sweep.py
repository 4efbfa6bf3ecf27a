#!/usr/bin/env python3
"""Measurement helper: ms per pipelined search step (10 M x 128, 32 queries, SQ_MEM_DEVICE_ASYNC) under option sweeps.
usage: python3 tools/step_sweep.py name=v1,v2,... [name2=...]   (options of sq_set_option; one axis at a time)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smqtk_indexing_amd import _lib

n, d, nq, k = int(os.environ.get("N", 10_000_000)), int(os.environ.get("D", 128)), int(os.environ.get("NQ", 32)), 100
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(3)
db = torch.empty((n, d), dtype=torch.float32, device=dev)
for s in range(0, n, 1 << 20):
    db[s:s + (1 << 20)].normal_(generator=g)
q = torch.empty((nq, d), dtype=torch.float32, device=dev).normal_(generator=g)
idx = _lib.DenseIndex(db.data_ptr(), n=n, d=d, device_ptr=True, keepalive=db)
st = torch.cuda.current_stream().cuda_stream
od = [torch.empty((nq, k), dtype=torch.float32, device=dev) for _ in range(4)]   # (up to dense_async_depth 4 in flight)
oi = [torch.empty((nq, k), dtype=torch.int64, device=dev) for _ in range(4)]

STEPS = int(os.environ.get("STEPS", 60))
DEPTH = int(os.environ.get("DEPTH", 0))
if DEPTH:
    _lib.set_option("dense_async_depth", DEPTH)


def run(steps=STEPS):
    for i in range(6):
        idx.search_device_async(q.data_ptr(), nq, k, od[i & 3].data_ptr(), oi[i & 3].data_ptr(), st)
    idx.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        idx.search_device_async(q.data_ptr(), nq, k, od[i & 3].data_ptr(), oi[i & 3].data_ptr(), st)
    idx.sync(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

def run_sync(steps=30):
    for i in range(3):
        idx.search_device(q.data_ptr(), nq, k, od[0].data_ptr(), oi[0].data_ptr(), st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        idx.search_device(q.data_ptr(), nq, k, od[i & 1].data_ptr(), oi[i & 1].data_ptr(), st)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

print(f"blocking calls: {run_sync():.4f} ms")
print(f"default: {run():.4f} ms  cands/q {idx.stats()['candidates'] / nq:.0f}")
if os.environ.get("REPS"):
    print("repeated:", " ".join(f"{run():.4f}" for _ in range(int(os.environ["REPS"]))))
for arg in sys.argv[1:]:
    name, vals = arg.split("=")
    for v in vals.split(","):
        _lib.set_option(name, int(v))
        print(f"{name}={v}: {run():.4f} ms  cands/q {idx.stats()['candidates'] / nq:.0f} fallbacks {idx.stats()['fallback_queries']}")
    _lib.set_option(name, 2 if name in ("dense_async_streams", "dense_async_depth") else 0)
