#!/usr/bin/env python3
"""Measurement helper: what each part of the collective costs a PipelinedShardedSearch step on one rank (1.25 M-row shard,
32 queries, depth 3, no wait, groups of 4): SKIP=gather|copy|merge (comma separated) drops the all-gather, the pinned
copy or the merge hand-off (results meaningless).  usage: SKIP=gather,copy python3 tools/pipe_ablate.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from smqtk_indexing_amd import _lib, distributed as D

skip = set(x for x in os.environ.get("SKIP", "").split(",") if x)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
n = int(os.environ.get("N", 1_250_000))
g = torch.Generator(device=dev); g.manual_seed(3)
db = torch.empty((n, 128), dtype=torch.float32, device=dev).normal_(generator=g)
q = torch.empty((32, 128), dtype=torch.float32, device=dev).normal_(generator=g)
idx = _lib.DenseIndex(db.data_ptr(), n=n, d=128, device_ptr=True, keepalive=db)


class _Done:
    def is_completed(self): return True
    def wait(self): return True


orig_gather = dist.all_gather_into_tensor
if "gather" in skip:
    dist.all_gather_into_tensor = lambda recv, send, group=None, async_op=False: _Done()
if os.environ.get("STREAM0") == "1":   # searches ordered behind the NULL stream instead of a stream of the pipeline's own
    idx = D.HipSearcher(idx, 0, True, 3, False, os.environ.get("READY", "1") == "1")
pipe = D.PipelinedShardedSearch(idx, 32, 100, torch.float32, merge_on=0, device=dev, use_async=True, depth=3, gather_every=4,
                                wait=False, queries_ready=os.environ.get("READY", "1") == "1")
if "copy" in skip:
    class _NoCopy:
        def __init__(self, t): self.t = t
        def copy_(self, *a, **k): return self.t
    pipe.host = [_NoCopy(h) for h in pipe.host]
if "merge" in skip:
    class _M:
        def submit(self, *a): return 0
        def result(self, t):
            import numpy as np
            return np.zeros((128, 100), np.float32), np.zeros((128, 100), np.int64)
        def close(self): pass
    pipe.merger.close(); pipe.merger = _M()
for _ in range(60):
    pipe.submit(q)
pipe.flush(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3000):
    pipe.submit(q)
pipe.flush(); torch.cuda.synchronize()
print(f"skip={sorted(skip)}: {(time.perf_counter() - t0) / 3000 * 1e3:.4f} ms per step")
pipe.close(); dist.destroy_process_group()
