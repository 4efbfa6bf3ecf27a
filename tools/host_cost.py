#!/usr/bin/env python3
"""Measurement helper: host time of one pipelined search step.  On a matrix so small that the device is never the
limit (N rows, default 100 k: the filter path with a ~10 us scan) the step time of a depth-4 pipeline IS the host's
share: the launches and event calls inside sq_dense_search plus the ctypes call.  Printed beside the same loop on the
8-GPU shard shape (1.25 M rows), where the two can be compared."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smqtk_indexing_amd import _lib

d, nq, k = 128, int(os.environ.get("NQ", 32)), 100
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream


def measure(n, depth, wait, order, steps=400):
    g = torch.Generator(device=dev); g.manual_seed(3)
    db = torch.empty((n, d), dtype=torch.float32, device=dev).normal_(generator=g)
    q = torch.empty((nq, d), dtype=torch.float32, device=dev).normal_(generator=g)
    idx = _lib.DenseIndex(db.data_ptr(), n=n, d=d, device_ptr=True, keepalive=db)
    idx.set_option("dense_async_depth", depth)
    idx.set_option("dense_async_wait", wait)
    idx.set_option("dense_async_order", order)
    od = [torch.empty((nq, k), dtype=torch.float32, device=dev) for _ in range(8)]
    oi = [torch.empty((nq, k), dtype=torch.int64, device=dev) for _ in range(8)]
    for i in range(10):
        idx.search_device_async(q.data_ptr(), nq, k, od[i & 7].data_ptr(), oi[i & 7].data_ptr(), st)
    idx.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        idx.search_device_async(q.data_ptr(), nq, k, od[i & 7].data_ptr(), oi[i & 7].data_ptr(), st)
    t1 = time.perf_counter()
    idx.sync(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    idx.close()
    return (t1 - t0) / steps * 1e6, (t2 - t0) / steps * 1e6


for n in (int(os.environ.get("N", 100_000)), 1_250_000):
    for depth, wait, order in ((6, 0, 0), (5, 0, 0), (4, 0, 0), (3, 0, 0), (3, 1, 1), (2, 1, 1)):
        a, b = measure(n, depth, wait, order)
        print(f"n={n} depth={depth} wait={wait} order={order}: host loop {a:.1f} us/step, with drain {b:.1f} us/step")
