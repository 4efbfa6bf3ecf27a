#!/usr/bin/env python3
"""Measurement helper: duration of the full scan pass (hipEvents inside the library, option `profile`) for one batch
shape, e.g. to compare ablation builds (SMQTK_HIP_LIBRARY=... built with -DSQ_ABL=bits: 1 no ring refill, 2 no LDS
fragment reads, 4 no epilogue).  usage: N=10000000 NQ=1024 python3 tools/scan_abl.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smqtk_indexing_amd import _lib

n, d, nq, k = int(os.environ.get("N", 10_000_000)), 128, int(os.environ.get("NQ", 1024)), 100
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(3)
db = torch.empty((n, d), dtype=torch.float32, device=dev)
for s in range(0, n, 1 << 20):
    db[s:s + (1 << 20)].normal_(generator=g)
q = torch.empty((nq, d), dtype=torch.float32, device=dev).normal_(generator=g)
idx = _lib.DenseIndex(db.data_ptr(), n=n, d=d, device_ptr=True, keepalive=db)
_lib.set_option("profile", 1)
for name, val in [a.split("=") for a in sys.argv[1:]]:
    _lib.set_option(name, int(val))
st = torch.cuda.current_stream().cuda_stream
od = torch.empty((nq, k), dtype=torch.float32, device=dev)
oi = torch.empty((nq, k), dtype=torch.int64, device=dev)
ms = []
for i in range(int(os.environ.get("STEPS", 4))):
    idx.search_device(q.data_ptr(), nq, k, od.data_ptr(), oi.data_ptr(), st)
    torch.cuda.synchronize()
    s_ = idx.stats()
    ms.append(s_["scan_ms"])
print(f"{os.environ.get('SMQTK_HIP_LIBRARY', 'default')}: scan_ms {['%.3f' % m for m in ms]} cands/q {s_['candidates'] / nq:.0f} fallbacks {s_['fallback_queries']}")
