#!/usr/bin/env python3
"""Randomised parity sweep (GPU): random shapes / k / batch / metric / data distributions through the
C ABI against the oracle.  usage: python tools/fuzz_parity.py [seed] [cases]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cpu_ref as O
from smqtk_indexing_amd import _lib

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(seed)
fails = 0

def data(n, d, kind):
    if kind == 0: return rng.standard_normal((n, d)).astype(np.float32)
    if kind == 1: return rng.random((n, d)).astype(np.float32)                         # all positive
    if kind == 2: return (50.0 + rng.standard_normal((n, d))).astype(np.float32)        # far from the origin
    if kind == 3:                                                                       # clusters + duplicates
        c = rng.standard_normal((max(2, n // 500), d)).astype(np.float32) * 3
        x = c[rng.integers(0, len(c), n)] + 0.05 * rng.standard_normal((n, d)).astype(np.float32)
        x[rng.integers(0, n, n // 50)] = x[rng.integers(0, n, n // 50)]
        return x
    x = rng.standard_normal((n, d)).astype(np.float32) * rng.lognormal(0, 1.5, (n, 1)).astype(np.float32)  # wild norms
    return x

t0 = time.time()
for c in range(cases):
    n = int(rng.choice([300, 5000, 70_000, 150_000, 400_000]))
    d = int(rng.choice([1, 7, 32, 64, 100, 128, 129, 200, 256, 384, 512, 700]))
    k = int(rng.choice([1, 5, 50, 100, 600]))
    nq = int(rng.choice([1, 3, 32, 33, 70, 130]))
    metric = str(rng.choice(["euclidean", "cosine"]))
    kind = int(rng.integers(0, 5))
    if n * d > 60_000_000: n = 60_000_000 // d
    db = data(n, d, kind)
    qs = data(nq, d, kind)
    qs[0] = db[rng.integers(0, n)]
    if kind == 1 and d > 1: db[rng.integers(0, n)] = 0.0                               # a zero row
    m = _lib.SQ_METRIC_L2 if metric == "euclidean" else _lib.SQ_METRIC_COSINE
    try:
        idx = _lib.DenseIndex(db, metric=m)
        dd, ii = idx.search(qs, k)
        st = idx.stats()
        for qi in rng.choice(nq, size=min(nq, 4), replace=False):
            rd, ri = O.dense_topk(db, qs[qi], k, metric)
            kk = len(rd)
            if metric == "euclidean":
                assert np.array_equal(dd[qi, :kk].view(np.uint32), rd.view(np.uint32)), "dist bits"
                assert np.array_equal(ii[qi, :kk], ri), "ids"
            else:
                assert np.allclose(dd[qi, :kk], rd, rtol=1e-12, atol=1e-15, equal_nan=True), "cos dist"
                mism = ii[qi, :kk] != ri
                if mism.any():
                    full = O.dense_distances(db, qs[qi], "cosine")
                    a, b = full[ii[qi, :kk][mism]], full[ri[mism]]
                    assert np.all((np.abs(a - b) < 1e-14) | (np.isnan(a) & np.isnan(b))), "cos ids"
        idx.close()
        print(f"ok   case {c}: n={n} d={d} k={k} nq={nq} {metric} kind={kind} fallback={st['fallback_queries']} cand/q={st['candidates'] / nq:.0f}", flush=True)
    except Exception as e:  # noqa: BLE001
        fails += 1
        print(f"FAIL case {c}: n={n} d={d} k={k} nq={nq} {metric} kind={kind}: {type(e).__name__} {e}", flush=True)
print(f"{cases - fails}/{cases} passed in {time.time() - t0:.0f} s")
sys.exit(1 if fails else 0)
