"""Cosine middle tier (sq_dense_mid.hpp) on descriptors that share an offset: per-call time and the tier each query ends
on, with the tier on and off.  One JSON line per case.

    python tools/cosine_mid_bench.py [n] [nq]
"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import cpu_ref as O  # noqa: E402  (the checker of one query per case, never the thing timed)
from smqtk_indexing_amd import _lib  # noqa: E402


def case(n, d, nq, k, offset, seed=0):
    rng = np.random.default_rng(seed)
    off = (offset * rng.standard_normal(d)).astype(np.float32)
    db = rng.standard_normal((n, d), dtype=np.float32) + off
    qs = rng.standard_normal((nq, d), dtype=np.float32) + off
    idx = _lib.DenseIndex(db, metric=_lib.SQ_METRIC_COSINE)
    out = {"n": n, "d": d, "queries": nq, "k": k, "offset_sigma": offset}
    for tier in (1, 0):                        # warm both configurations before either is timed
        idx.set_option("dense_mid_tier", tier)
        for _ in range(4):
            idx.search(qs, k)
    for tier in (1, 0):
        idx.set_option("dense_mid_tier", tier)
        dist, ids = idx.search(qs, k)
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            dist, ids = idx.search(qs, k)
        ms = (time.perf_counter() - t0) / reps * 1e3
        st = idx.stats()
        out[f"tier_{'on' if tier else 'off'}"] = {
            "ms_per_call": round(ms, 3), "mid_tier_queries": st["mid_tier_queries"], "exact_path_queries": st["fallback_queries"],
            "first_filter_candidates_per_query": round(st["candidates"] / nq, 1)}
        if tier:
            keep = (dist.copy(), ids.copy())
    out["same_answers"] = bool(np.array_equal(keep[0].view(np.uint64), dist.view(np.uint64)) and np.array_equal(keep[1], ids))
    if n <= 2_000_000:
        rd, ri = O.dense_topk(db, qs[0], k, "cosine")
        out["oracle_query0"] = bool(np.allclose(keep[0][0], rd, rtol=1e-12, atol=1e-15) and np.array_equal(keep[1][0], ri))
    idx.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    nq = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    cases = ((128, 50.0), (128, 10.0), (128, 3.0), (256, 50.0), (512, 50.0), (64, 200.0))
    if len(sys.argv) > 3:                      # one case (for a profiler run)
        cases = cases[int(sys.argv[3]):int(sys.argv[3]) + 1]
    for d, offset in cases:
        case(n if d <= 256 else n // 2, d, nq, 100, offset, seed=d)
