#!/usr/bin/env python3
"""Measurement helper: the int8 full pass with and without its score epilogue (option dense_debug = 1024: the stream and
the MFMAs alone; the results of such a call are garbage) -- how much of the pass is the vector unit's.
usage: python3 tools/int8_ablation.py   (env D: 128, 256, 512)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smqtk_indexing_amd import _lib
n, d, nq, k = 10_000_000, int(os.environ.get("D", 128)), 32, 100
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(3)
db = torch.empty((n, d), dtype=torch.float32, device=dev)
for s in range(0, n, 1 << 20):
    db[s:s + (1 << 20)].normal_(generator=g)
q = torch.empty((nq, d), dtype=torch.float32, device=dev).normal_(generator=g)
idx = _lib.DenseIndex(db.data_ptr(), n=n, d=d, device_ptr=True, keepalive=db)
st = torch.cuda.current_stream().cuda_stream
od = torch.empty((nq, k), dtype=torch.float32, device=dev); oi = torch.empty((nq, k), dtype=torch.int64, device=dev)
_lib.set_option("profile", 1)
for dbg in (0, 1024, 0, 1024):
    _lib.set_option("dense_debug", dbg)
    ts = []
    for i in range(12):
        idx.search_device(q.data_ptr(), nq, k, od.data_ptr(), oi.data_ptr(), st)
        ts.append(idx.stats()["scan_ms"])
    print(f"dense_debug={dbg}: scan {sum(ts[2:]) / len(ts[2:]):.4f} ms (min {min(ts[2:]):.4f})", flush=True)
_lib.set_option("dense_debug", 0)
