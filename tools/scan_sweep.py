#!/usr/bin/env python3
"""Measurement helper (not part of the product): times the dense scan kernel
under different ring depths / grid sizes / ablations on one GPU."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smqtk_indexing_amd import _lib

def main():
    n, d, k = int(os.environ.get("N", 10_000_000)), int(os.environ.get("D", 128)), 100
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(3)
    db = torch.empty((n, d), dtype=torch.float32, device=dev).normal_(generator=g)
    _lib.set_option("profile", 1)
    idx = _lib.DenseIndex(db.data_ptr(), n=n, d=d, device_ptr=True, keepalive=db)
    st = torch.cuda.current_stream().cuda_stream
    def run(nq, reps=6, **opts):
        for kk, v in opts.items(): _lib.set_option(kk, v)
        q = torch.empty((nq, d), dtype=torch.float32, device=dev).normal_(generator=g)
        od = torch.empty((nq, k), dtype=torch.float32, device=dev); oi = torch.empty((nq, k), dtype=torch.int64, device=dev)
        ms, tot = [], []
        for r in range(reps):
            idx.search_device(q.data_ptr(), nq, k, od.data_ptr(), oi.data_ptr(), st)
            s = idx.stats(); ms.append(s["scan_ms"]); tot.append(s["total_ms"])
        for kk in opts: _lib.set_option(kk, 0)
        return float(np.median(ms[1:])), float(np.median(tot[1:])), s["fallback_queries"]
    bpr = (-(-d // 128) * 256 + 4)
    for nq in (32,):
        for name, opts in [("default", {}), ("dma_only", {"dense_debug": 1}), ("no_emit", {"dense_debug": 4}),
                           ("w4", {"dense_waves": 4}), ("w4_s4", {"dense_waves": 4, "dense_stages": 4}),
                           ("stride8", {"sample_stride": 8}), ("stride32", {"sample_stride": 32}),
                           ("stride64", {"sample_stride": 64}), ("blocks512", {"dense_blocks": 512})]:
            sm, tm, fb = run(nq, **opts)
            c = idx.stats()["candidates"] / nq
            print(f"nq={nq:5d} {name:12s} scan_ms={sm:8.4f} total_ms={tm:8.4f} scan_GBps={n * bpr / sm / 1e6 if sm else 0:9.1f} cand/q={c:8.0f} fb={fb}", flush=True)
    for nq in (1, 64, 128, 256, 1024):
        sm, tm, fb = run(nq, reps=3)
        print(f"nq={nq:5d} default      scan_ms={sm:8.4f} total_ms={tm:8.4f} fb={fb}", flush=True)

if __name__ == "__main__":
    main()
