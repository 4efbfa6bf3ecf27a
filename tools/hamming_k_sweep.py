import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smqtk_indexing_amd import _lib
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(1)
codes = torch.randint(-2**63, 2**63 - 1, (10_000_000, 1), dtype=torch.int64, device=dev, generator=g)
u = torch.unique(codes.view(-1)).contiguous()
idx = _lib.HammingIndex(u.data_ptr(), n=int(u.numel()), words=1, device_ptr=True, keepalive=u)
st = torch.cuda.current_stream().cuda_stream
for nq in (1, 32):
    q = u[torch.randint(0, u.numel(), (nq,), device=dev, generator=g)].contiguous()
    for k in (100, 300, 1000, 3000, 10000, 30000):
        od = torch.empty((nq, k), dtype=torch.int32, device=dev); oi = torch.empty((nq, k), dtype=torch.int64, device=dev)
        idx.search_device(q.data_ptr(), nq, k, od.data_ptr(), oi.data_ptr(), st); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            idx.search_device(q.data_ptr(), nq, k, od.data_ptr(), oi.data_ptr(), st)
        torch.cuda.synchronize()
        s = idx.stats()
        print(nq, k, f"{(time.perf_counter()-t0)/3*1e3:.3f} ms", "fallbacks", s["fallback_queries"], "cands/q", s["candidates"]/nq)
