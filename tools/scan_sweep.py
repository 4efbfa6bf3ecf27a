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
    for qt in (4,):
        for name, dbg in (("full", 0), ("dma_only", 1), ("no_dma", 2), ("no_emit", 4)):
            sm, tm, fb = run(256, reps=3, dense_qt=qt, dense_debug=dbg)
            print(f"nq=  256 qt={qt} {name:9s} scan_ms={sm:8.4f} total_ms={tm:8.4f} fb={fb}", flush=True)

def qt_sweep():
    pass

if __name__ == "__main__":
    main()
