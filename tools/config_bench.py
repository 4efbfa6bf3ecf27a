#!/usr/bin/env python3
"""Measurement helper: the per-GPU work of BASELINE.json configs C2 (1M x 128 L2, 1024 queries)
and C4 (one of eight shards of 100M x 512 cosine: 12.5M rows, 256 queries) on one MI355X."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smqtk_indexing_amd import _lib

dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream

def case(tag, n, d, nq, k, metric):
    g = torch.Generator(device=dev); g.manual_seed(4)
    db = torch.empty((n, d), dtype=torch.float32, device=dev).normal_(generator=g)
    t0 = time.perf_counter()
    idx = _lib.DenseIndex(db.data_ptr(), n=n, d=d, metric=metric, device_ptr=True, keepalive=db)
    torch.cuda.synchronize(); build_s = time.perf_counter() - t0
    _lib.set_option("profile", 1)
    q = torch.empty((nq, d), dtype=torch.float32, device=dev).normal_(generator=g)
    dsz = torch.float64 if metric == _lib.SQ_METRIC_COSINE else torch.float32
    od = torch.empty((nq, k), dtype=dsz, device=dev); oi = torch.empty((nq, k), dtype=torch.int64, device=dev)
    idx.search_device(q.data_ptr(), nq, k, od.data_ptr(), oi.data_ptr(), st); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); idx.search_device(q.data_ptr(), nq, k, od.data_ptr(), oi.data_ptr(), st)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    s = idx.stats(); dt = float(np.median(ts))
    print(json.dumps({"config": tag, "rows": n, "dim": d, "queries": nq, "k": k, "metric": "cosine" if metric else "l2",
                      "ms_per_call": dt * 1e3, "queries_per_s": nq / dt, "scan_ms": s["scan_ms"],
                      "candidates_per_query": s["candidates"] / nq, "fallback_queries": s["fallback_queries"],
                      "index_build_s": build_s}), flush=True)
    idx.close()

if __name__ == "__main__":
    case("C2", 1_000_000, 128, 1024, 100, _lib.SQ_METRIC_L2)
    case("C4_shard", 12_500_000, 512, 256, 100, _lib.SQ_METRIC_COSINE)
    case("C4_shard_32q", 12_500_000, 512, 32, 100, _lib.SQ_METRIC_COSINE)


def c1_plugin():
    """C1: 10k x 128 float32, k = 10, 100 queries through the plugin class (host containers + small exact path)."""
    from smqtk_indexing_amd._compat import DescriptorMemoryElement
    from smqtk_indexing_amd.impls.nn_index.hip_bruteforce import HipBruteForceNearestNeighborsIndex
    rng = np.random.default_rng(1)
    x = rng.random((10_000, 128), dtype=np.float32)
    qs = rng.random((100, 128), dtype=np.float32)
    idx = HipBruteForceNearestNeighborsIndex()
    t0 = time.perf_counter(); idx.build_index(DescriptorMemoryElement(i).set_vector(v) for i, v in enumerate(x)); build_s = time.perf_counter() - t0
    qd = [DescriptorMemoryElement(f"q{i}").set_vector(v) for i, v in enumerate(qs)]
    idx.nn(qd[0], 10)
    t0 = time.perf_counter()
    for q in qd: idx.nn(q, 10)
    single = 100 / (time.perf_counter() - t0)
    idx.nn_many(qs, 10)
    t0 = time.perf_counter(); idx.nn_many(qs, 10); many = 100 / (time.perf_counter() - t0)
    print(json.dumps({"config": "C1_plugin", "rows": 10_000, "dim": 128, "k": 10, "nn_queries_per_s": single,
                      "nn_many100_queries_per_s": many, "index_build_s": build_s}), flush=True)


if __name__ == "__main__" and os.environ.get("C1", "1") == "1":
    c1_plugin()
