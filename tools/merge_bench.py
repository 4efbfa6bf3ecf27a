#!/usr/bin/env python3
"""Measurement helper (not part of the product): the host-side shard merge (sq_merge_topk_strided)
on this machine's CPU, at the shapes a multi-GPU step produces.  No GPU needed.
usage: python tools/merge_bench.py [shards nq k]..."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smqtk_indexing_amd import _lib  # noqa: E402


def main():
    L = _lib.load()
    a = [int(x) for x in sys.argv[1:]]
    shapes = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)] or [(8, 256, 100), (8, 32, 100), (2, 64, 100), (1, 256, 100), (8, 1024, 100)]
    rng = np.random.default_rng(0)
    _lib.set_option("merge_threads", int(os.environ.get("THREADS", 0)))
    for world, nq, k in shapes:
        per = nq * k * 12
        buf = np.empty((world, per), dtype=np.uint8)
        for r in range(world):
            d = np.sort(rng.random((nq, k)).astype(np.float32), axis=1)
            i = np.sort(rng.integers(0, 10 ** 6, (nq, k)), axis=1).astype(np.int64) + r * 10 ** 6
            buf[r, :nq * k * 8] = i.view(np.uint8).reshape(-1)
            buf[r, nq * k * 8:] = d.view(np.uint8).reshape(-1)
        od = np.empty((nq, k), np.float32)
        oi = np.empty((nq, k), np.int64)
        base = buf.ctypes.data
        ts = []
        for _ in range(50):
            t = time.perf_counter()
            L.sq_merge_topk_strided(base + nq * k * 8, base, 0, world, nq, k, k, per, per, od.ctypes.data, oi.ctypes.data)
            ts.append(time.perf_counter() - t)
        print(f"shards={world} nq={nq} k={k}: median {np.median(ts) * 1e3:.3f} ms  min {min(ts) * 1e3:.3f} ms  "
              f"{np.median(ts) * 1e9 / (nq * k):.1f} ns per output", flush=True)


if __name__ == "__main__":
    main()
