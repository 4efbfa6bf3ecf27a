#!/usr/bin/env python3
"""Measurement helper: the int8 first-stage filter (dense_int8) against the bfloat16 one on the same index and queries --
results must be identical (both paths re-rank in the reference's arithmetic and certify); prints candidates, tiers taken
and step times.  usage: python3 tools/int8_check.py   (env: N, NQ, K, DATA=normal|uniform|clustered|scaled, STEPS)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smqtk_indexing_amd import _lib

n, d = int(os.environ.get("N", 10_000_000)), int(os.environ.get("D", 128))
nq, k = int(os.environ.get("NQ", 32)), int(os.environ.get("K", 100))
data = os.environ.get("DATA", "normal")
STEPS = int(os.environ.get("STEPS", 100))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(3)
db = torch.empty((n, d), dtype=torch.float32, device=dev)
for s in range(0, n, 1 << 20):
    blk = db[s:s + (1 << 20)]
    if data == "uniform":
        blk.uniform_(0.0, 1.0, generator=g)
    else:
        blk.normal_(generator=g)
if data == "clustered":
    cent = torch.empty((1000, d), dtype=torch.float32, device=dev).normal_(generator=g) * 4.0
    for s in range(0, n, 1 << 20):
        blk = db[s:s + (1 << 20)]
        blk += cent[torch.randint(0, 1000, (blk.shape[0],), device=dev, generator=g)]
if data == "scaled":
    db *= 37.5
    db += 11.0
qs = []
for b in range(4):
    sel = torch.randint(0, n, (nq,), device=dev, generator=g)
    qs.append((db[sel] + 0.3 * db.std() * torch.empty((nq, d), device=dev).normal_(generator=g)).contiguous() if b & 1 else
              (torch.empty((nq, d), dtype=torch.float32, device=dev).normal_(generator=g) * db.std() + db.mean()).contiguous())
t0 = time.perf_counter()
idx = _lib.DenseIndex(db.data_ptr(), n=n, d=d, device_ptr=True, keepalive=db)
torch.cuda.synchronize()
print(f"create: {time.perf_counter() - t0:.3f} s")
st = torch.cuda.current_stream().cuda_stream
od = [torch.empty((nq, k), dtype=torch.float32, device=dev) for _ in range(4)]
oi = [torch.empty((nq, k), dtype=torch.int64, device=dev) for _ in range(4)]


def blocking(q):
    idx.search_device(q.data_ptr(), nq, k, od[0].data_ptr(), oi[0].data_ptr(), st)
    torch.cuda.synchronize()
    return od[0].clone(), oi[0].clone(), idx.stats()


def step(q, steps=STEPS):
    for i in range(6):
        idx.search_device_async(q.data_ptr(), nq, k, od[i & 3].data_ptr(), oi[i & 3].data_ptr(), st)
    idx.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        idx.search_device_async(q.data_ptr(), nq, k, od[i & 3].data_ptr(), oi[i & 3].data_ptr(), st)
    idx.sync(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


ok = True
for b, q in enumerate(qs):
    res = {}
    for mode in (0, 1):
        _lib.set_option("dense_int8", mode)
        _lib.set_option("profile", 1)
        dd, ii, stt = blocking(q)
        dd, ii, stt = blocking(q)
        _lib.set_option("profile", 0)
        res[mode] = (dd, ii)
        print(f"batch {b} int8={mode}: scan {stt['scan_ms']:.4f} ms rerank {stt['rerank_ms']:.4f} total {stt['total_ms']:.4f} "
              f"bytes {stt['bytes_scanned'] / 1e9:.3f} GB cands/q {stt['candidates'] / nq:.0f} mid {stt['mid_tier_queries']} "
              f"exact {stt['fallback_queries']}   step {step(q):.4f} ms")
    same = torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    ok &= same
    print(f"batch {b}: identical = {same}")
print("ALL IDENTICAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
