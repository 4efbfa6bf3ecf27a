#!/usr/bin/env python3
"""Measurement helper: where the host's time of a pipelined search call goes (SQ_HOSTPROF=1: the library accumulates the
nanoseconds of an asynchronous sq_dense_search by section -- entry, resolve of the slot being reused, stream / event
setup, enqueue, wait -- and prints them per call at sq_dense_sync), on a 100 k-row matrix (the device is never the limit
for bandwidth: what remains is the latency of the call's chain of kernels) and on the 1.25 M-row shard, with the
captured graph and with eager launches.  usage: SQ_HOSTPROF=1 python3 tools/host_sections.py   (env WAIT, ORDER)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smqtk_indexing_amd import _lib
d, nq, k = 128, 32, 100
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
for n in (100_000, 1_250_000):
  for graph in (1, 0):
    g = torch.Generator(device=dev); g.manual_seed(3)
    db = torch.empty((n, d), dtype=torch.float32, device=dev).normal_(generator=g)
    q = torch.empty((nq, d), dtype=torch.float32, device=dev).normal_(generator=g)
    idx = _lib.DenseIndex(db.data_ptr(), n=n, d=d, device_ptr=True, keepalive=db)
    idx.set_option("dense_async_depth", 3); idx.set_option("dense_graph", graph)
    idx.set_option("dense_async_wait", int(os.environ.get("WAIT", 0))); idx.set_option("dense_async_order", int(os.environ.get("ORDER", 0)))
    od = [torch.empty((nq, k), dtype=torch.float32, device=dev) for _ in range(8)]
    oi = [torch.empty((nq, k), dtype=torch.int64, device=dev) for _ in range(8)]
    for i in range(20):
        idx.search_device_async(q.data_ptr(), nq, k, od[i & 7].data_ptr(), oi[i & 7].data_ptr(), st)
    idx.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(600):
        idx.search_device_async(q.data_ptr(), nq, k, od[i & 7].data_ptr(), oi[i & 7].data_ptr(), st)
    t1 = time.perf_counter()
    print(f"n={n} graph={graph}: loop {(t1 - t0) / 600 * 1e6:.1f} us/step", flush=True)
    idx.sync(); idx.close()
