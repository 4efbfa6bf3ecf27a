#!/usr/bin/env python3
"""Measurement helper (not part of the product): the dense scan under its measurement options on
one GPU -- ablations of the single-tile kernel, sample strides, query tiles per wave, batch sizes.
usage: [N=10000000] [D=128] [NQS=32,128,1024] python tools/scan_sweep.py [ablate] [stride] [qt] [rerank] [batch]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from smqtk_indexing_amd import _lib  # noqa: E402


def main():
    which = set(sys.argv[1:]) or {"ablate", "stride", "qt", "batch"}
    n, d, k = int(os.environ.get("N", 10_000_000)), int(os.environ.get("D", 128)), 100
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    db = torch.empty((n, d), dtype=torch.float32, device=dev).normal_(generator=g)
    _lib.set_option("profile", 1)
    idx = _lib.DenseIndex(db.data_ptr(), n=n, d=d, device_ptr=True, keepalive=db)
    st = torch.cuda.current_stream().cuda_stream
    bytes_per_row = -(-d // 128) * 256 + 4

    def run(nq, reps=5, **opts):
        for name, v in opts.items():
            _lib.set_option(name, v)
        q = torch.empty((nq, d), dtype=torch.float32, device=dev).normal_(generator=g)
        od = torch.empty((nq, k), dtype=torch.float32, device=dev)
        oi = torch.empty((nq, k), dtype=torch.int64, device=dev)
        ms, tot = [], []
        for _ in range(reps):
            idx.search_device(q.data_ptr(), nq, k, od.data_ptr(), oi.data_ptr(), st)
            s = idx.stats()
            ms.append(s["scan_ms"])
            tot.append(s["total_ms"])
        for name in opts:
            _lib.set_option(name, 0)
        return float(np.median(ms[1:])), float(np.median(tot[1:])), s

    def show(tag, nq, sm, tm, s):
        print(f"{tag:26s} nq={nq:5d} scan_ms={sm:8.4f} total_ms={tm:8.4f} "
              f"GB/s={n * bytes_per_row / sm / 1e6 if sm else 0:8.1f} cand/q={s['candidates'] / nq:7.0f} "
              f"fallback={s['fallback_queries']}", flush=True)

    if "ablate" in which:   # dma_only / no_emit return garbage on purpose: every query then takes the exact path
        for tag, opts in (("default", {}), ("dma_only (debug 1)", {"dense_debug": 1}), ("no_emit (debug 4)", {"dense_debug": 4}),
                          ("4 waves", {"dense_waves": 4}), ("4 waves, 4 stages", {"dense_waves": 4, "dense_stages": 4}),
                          ("512 row blocks", {"dense_blocks": 512}),
                          ("rerank: no reservation atomics (debug 32)", {"dense_debug": 32}),
                          ("rerank: cache-resident rows (debug 64)", {"dense_debug": 64}),
                          ("rerank: both (debug 96)", {"dense_debug": 96})):
            nq_ab = int(os.environ.get("NQ_ABLATE", 32))
            show(tag, nq_ab, *run(nq_ab, **opts))
    if "stride" in which:
        for nq in [int(x) for x in os.environ.get("NQS", "32,128,1024").split(",")]:
            for stride in (2, 3, 4, 6, 8, 12, 16, 24):
                show(f"sample_stride={stride}", nq, *run(nq, reps=4, sample_stride=stride))
    if "qt" in which:
        for nq in (64, 128, 1024):
            for qt in (1, 2, 4):
                for qp in (0, 2):
                    show(f"dense_qt={qt} qplanes={qp or 'auto'}", nq, *run(nq, reps=4, dense_qt=qt, dense_qplanes=qp))
    if "rerank" in which:
        for nq in [int(x) for x in os.environ.get("NQS", "32,256").split(",")]:
            for seg in (1, 2, 4, 8):
                show(f"dense_rerank_segments={seg}", nq, *run(nq, reps=6, dense_rerank_segments=seg))
    if "batch" in which:
        for nq in (1, 32, 64, 128, 256, 1024):
            show("auto", nq, *run(nq, reps=4))


if __name__ == "__main__":
    main()
