#!/usr/bin/env python3
"""Measurement helper: where the workgroups of the fused int8 full pass (dense8_body_kernel) spend their time -- the
library prints per-workgroup clock statistics for blocking calls under "dense_debug" = 8192 | 16384.
usage: N=10000000 NQ=32 [OPTS=name=v,...] python3 tools/body_clocks.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smqtk_indexing_amd import _lib

n, d, nq, k = int(os.environ.get("N", 10_000_000)), int(os.environ.get("D", 128)), int(os.environ.get("NQ", 32)), 100
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(3)
db = torch.empty((n, d), dtype=torch.float32, device=dev)
for s in range(0, n, 1 << 20):
    db[s:s + (1 << 20)].normal_(generator=g)
qs = [torch.empty((nq, d), dtype=torch.float32, device=dev).normal_(generator=g) for _ in range(4)]
idx = _lib.DenseIndex(db.data_ptr(), n=n, d=d, device_ptr=True, keepalive=db)
for item in os.environ.get("OPTS", "").split(","):
    if item:
        a, b = item.split("=")
        idx.set_option(a, int(b))
st = torch.cuda.current_stream().cuda_stream
od = torch.empty((nq, k), dtype=torch.float32, device=dev); oi = torch.empty((nq, k), dtype=torch.int64, device=dev)
for i in range(6):
    idx.search_device(qs[i % 4].data_ptr(), nq, k, od.data_ptr(), oi.data_ptr(), st)
torch.cuda.synchronize()
idx.set_option("dense_debug", 8192 | 16384)
idx.set_option("profile", 1)
for i in range(3):
    idx.search_device(qs[i % 4].data_ptr(), nq, k, od.data_ptr(), oi.data_ptr(), st)
    print(f"scan_ms {idx.stats()['scan_ms']:.4f} total_ms {idx.stats()['total_ms']:.4f} cands/q {idx.stats()['candidates'] / nq:.0f}", file=sys.stderr)
