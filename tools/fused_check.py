#!/usr/bin/env python3
"""Measurement / debugging helper: the three-launch int8 call ("dense_fused" = 1) against the six-launch chain and the bf16
filter on the same index and queries -- ids and distance bits must be identical -- for blocking and pipelined calls.
usage: N=1000000 D=128 NQ=32 METRIC=l2 python3 tools/fused_check.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smqtk_indexing_amd import _lib

n, d, k = int(os.environ.get("N", 1_000_000)), int(os.environ.get("D", 128)), int(os.environ.get("K", 100))
nqs = [int(v) for v in os.environ.get("NQ", "32,7,64").split(",")]
metric = os.environ.get("METRIC", "l2")
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(11)
db = torch.empty((n, d), dtype=torch.float32, device=dev)
for s in range(0, n, 1 << 20):
    db[s:s + (1 << 20)].normal_(generator=g)
idx = _lib.DenseIndex(db.data_ptr(), n=n, d=d, device_ptr=True, keepalive=db, metric=_lib.SQ_METRIC_COSINE if metric == "cosine" else _lib.SQ_METRIC_L2)
st = torch.cuda.current_stream().cuda_stream
ddt = torch.float64 if metric == "cosine" else torch.float32


def blocking(q, nq):
    od = torch.empty((nq, k), dtype=ddt, device=dev); oi = torch.empty((nq, k), dtype=torch.int64, device=dev)
    idx.search_device(q.data_ptr(), nq, k, od.data_ptr(), oi.data_ptr(), st)
    torch.cuda.synchronize()
    return od.cpu().numpy(), oi.cpu().numpy(), dict(idx.stats())


def pipelined(qs, nq):
    outs = [(torch.empty((nq, k), dtype=ddt, device=dev), torch.empty((nq, k), dtype=torch.int64, device=dev)) for _ in qs]
    for q, (od, oi) in zip(qs, outs):
        idx.search_device_async(q.data_ptr(), nq, k, od.data_ptr(), oi.data_ptr(), st)
    idx.sync(); torch.cuda.synchronize()
    return [(od.cpu().numpy(), oi.cpu().numpy()) for od, oi in outs]


for nq in nqs:
    qs = [torch.empty((nq, d), dtype=torch.float32, device=dev).normal_(generator=g) for _ in range(6)]
    qs[0][0] = db[n // 3]
    res = {}
    for name, opts in (("fused", {"dense_fused": 1}), ("fused_untightened", {"dense_tighten": 0}), ("chain", {"dense_fused": 0}), ("bf16", {"dense_int8": 0})):
        for o, v in (("dense_fused", 1), ("dense_int8", -1), ("dense_tighten", 1)):
            idx.set_option(o, v)
        for o, v in opts.items():
            idx.set_option(o, v)
        print(f"nq={nq} {name}: blocking ...", file=sys.stderr, flush=True)
        dd, ii, stt = blocking(qs[0], nq)
        print(f"nq={nq} {name}: pipelined ...", file=sys.stderr, flush=True)
        pp = pipelined(qs, nq)
        pp2 = pipelined(qs, nq)   # (second round: the captured graph)
        res[name] = (dd, ii, pp, pp2)
        print(f"nq={nq} {name}: cands/q {stt['candidates'] / nq:.0f} fallbacks {stt['fallback_queries']} mid {stt.get('mid_tier_queries')} "
              f"bytes {stt['bytes_scanned']} launches {stt['scan_launches']}")
    ref = res["bf16"]
    for name in ("fused", "fused_untightened", "chain"):
        r = res[name]
        ok = np.array_equal(r[1], ref[1]) and np.array_equal(r[0].view(np.uint64 if metric == "cosine" else np.uint32), ref[0].view(np.uint64 if metric == "cosine" else np.uint32))
        for a, b in zip(r[2] + r[3], ref[2] + ref[3]):
            ok = ok and np.array_equal(a[1], b[1]) and np.array_equal(a[0].tobytes(), b[0].tobytes())
        print(f"nq={nq} {name} == bf16: {ok}")
    assert res["fused"][1][0, 0] == n // 3
print("done")
