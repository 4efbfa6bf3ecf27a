#!/usr/bin/env python3
"""Measurement helper: where the host time of a PipelinedShardedSearch step goes (one rank, collective forced over a
world of one): cProfile of 3000 submits on a 1.25 M-row shard.  usage: python3 tools/pipe_profile.py [rows] [gather_every]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from smqtk_indexing_amd import _lib
from smqtk_indexing_amd.distributed import PipelinedShardedSearch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
every = int(sys.argv[2]) if len(sys.argv) > 2 else 4
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(3)
db = torch.empty((n, 128), dtype=torch.float32, device=dev).normal_(generator=g)
q = torch.empty((32, 128), dtype=torch.float32, device=dev).normal_(generator=g)
idx = _lib.DenseIndex(db.data_ptr(), n=n, d=128, device_ptr=True, keepalive=db)
pipe = PipelinedShardedSearch(idx, 32, 100, torch.float32, merge_on=0, device=dev, use_async=True, depth=3, gather_every=every)
for _ in range(50):
    pipe.submit(q)
pipe.flush(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3000):
    pipe.submit(q)
pipe.flush(); torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 3000 * 1e3:.4f} ms per step (rows {n}, gather_every {every})")
pr = cProfile.Profile(); pr.enable()
for _ in range(3000):
    pipe.submit(q)
pipe.flush(); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
pipe.close(); idx.close(); dist.destroy_process_group()
