#!/usr/bin/env python3
"""Measurement helper (not part of the product): ITQ hashing and Hamming top-k
at the BASELINE.json config sizes on one GPU.  Prints one JSON line per case."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smqtk_indexing_amd import _lib

dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    return float(np.median(ts))

def itq_case(n, d, bits, norm):
    g = torch.Generator(device=dev); g.manual_seed(1)
    x = torch.empty((n, d), dtype=torch.float32, device=dev).normal_(generator=g)
    mean = torch.zeros(d, dtype=torch.float64, device=dev)
    q, _ = np.linalg.qr(np.random.default_rng(0).standard_normal((d, d)))
    rot = torch.from_numpy(np.ascontiguousarray(q[:, :bits] if bits <= d else np.random.default_rng(0).standard_normal((d, bits)))).to(dev)
    w = (bits + 63) // 64
    out = torch.empty((n, w), dtype=torch.int64, device=dev)
    dt = timed(lambda: _lib.itq_hash_device(x.data_ptr(), 0, n, d, mean.data_ptr(), rot.data_ptr(), bits, norm, out.data_ptr(), st))
    print(json.dumps({"kernel": "itq_hash", "n": n, "d": d, "bits": bits, "norm": norm, "ms": dt * 1e3,
                      "GBps": (n * d * 4 + n * w * 8) / dt / 1e9, "f64_TFLOPs": 2.0 * n * d * bits / dt / 1e12}), flush=True)

def hamming_case(n, words, nqs, k=100):
    g = torch.Generator(device=dev); g.manual_seed(2)
    codes = torch.randint(-2**63, 2**63 - 1, (n, words), dtype=torch.int64, device=dev, generator=g)
    # unique + sorted like the host index would hold them (row order irrelevant for timing)
    idx = _lib.HammingIndex(codes.data_ptr(), n=n, words=words, device_ptr=True, keepalive=codes)
    _lib.set_option("profile", 1)
    for nq in nqs:
        q = torch.randint(-2**63, 2**63 - 1, (nq, words), dtype=torch.int64, device=dev, generator=g)
        od = torch.empty((nq, k), dtype=torch.int32, device=dev); oi = torch.empty((nq, k), dtype=torch.int64, device=dev)
        dt = timed(lambda: idx.search_device(q.data_ptr(), nq, k, od.data_ptr(), oi.data_ptr(), st))
        s = idx.stats()
        print(json.dumps({"kernel": "hamming_search", "n": n, "bits": words * 64, "nq": nq, "k": k, "ms_per_call": dt * 1e3,
                          "queries_per_s": nq / dt, "scan_ms": s["scan_ms"], "scan_GBps": n * words * 8 / (s["scan_ms"] * 1e-3) / 1e9 if s["scan_ms"] else None,
                          "pair_Gops": n * nq / (s["scan_ms"] * 1e-3) / 1e9 if s["scan_ms"] else None,
                          "candidates_per_query": s["candidates"] / nq, "fallback": s["fallback_queries"]}), flush=True)
    idx.close()

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "itq"):
        itq_case(10_000_000, 128, 64, -1)
        itq_case(10_000_000, 128, 64, 2)
        itq_case(2_000_000, 512, 256, -1)
    if which in ("all", "hamming"):
        hamming_case(10_000_000, 1, (1, 32, 1024))
        hamming_case(125_000_000, 4, (1, 16, 256))
