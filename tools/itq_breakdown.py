#!/usr/bin/env python3
"""Measurement helper: sq_itq_hash at one shape under rocprofv3 --kernel-trace (per-kernel breakdown).
usage: N=10000000 D=128 BITS=64 NORM=-1 python3 tools/itq_breakdown.py [option=value ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smqtk_indexing_amd import _lib

n, d, bits, norm = (int(os.environ.get(k, v)) for k, v in (("N", 10_000_000), ("D", 128), ("BITS", 64), ("NORM", -1)))
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.empty((n, d), dtype=torch.float32, device=dev).normal_(generator=g)
mean = x[:100_000].to(torch.float64).mean(dim=0).contiguous()
q, _ = np.linalg.qr(np.random.default_rng(0).standard_normal((d, d)))
rot = torch.from_numpy(np.ascontiguousarray(q[:, :bits])).to(dev)
out = torch.empty((n, (bits + 63) // 64), dtype=torch.int64, device=dev)
for name, val in [a.split("=") for a in sys.argv[1:]]:
    _lib.set_option(name, int(val))
for r in range(int(os.environ.get("REPS", 6))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _lib.itq_hash_device(x.data_ptr(), 0, n, d, mean.data_ptr(), rot.data_ptr(), bits, norm, out.data_ptr(), st)
    torch.cuda.synchronize(); wall = time.perf_counter() - t0
print(f"last call wall {wall * 1e3:.3f} ms")
