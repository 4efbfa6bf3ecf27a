#!/usr/bin/env python3
"""Randomised parity run of the int8 first-stage filter: random shapes (n, d <= 512, nq <= 64, k), metrics and data
families (ties, sparse rows, offsets, outliers), a third of the cases with the last rows arriving through
sq_dense_append; every case compares the int8 filter with the bf16 filter bit for bit and one query with the oracle.
usage: python3 tools/int8_fuzz.py [cases] [first seed]   (a GPU run of ~1 s per case)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cpu_ref as O
from smqtk_indexing_amd import _lib

FAMILIES = ["normal", "uniform", "offset", "clustered", "nonneg", "sparse", "integers", "duplicates", "outliers", "tiny", "huge"]


def run_case(seed):
    """One random case: (passed, int8 engaged, description)."""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(66_000, 260_000))
    d = int(rng.choice([2, 3, 17, 64, 100, 128, 129, 200, 256, 300, 384, 512]))
    nq = int(rng.integers(1, 65)) if seed % 3 == 0 else int(rng.integers(1, 33))
    k = int(rng.choice([1, 2, 10, 100, 300]))
    metric = "cosine" if rng.random() < 0.35 else "euclidean"
    fam = FAMILIES[int(rng.integers(0, len(FAMILIES)))]
    db = rng.standard_normal((n, d)).astype(np.float32)
    qs = rng.standard_normal((nq, d)).astype(np.float32)
    if fam == "uniform":
        db, qs = rng.random((n, d), dtype=np.float32), rng.random((nq, d), dtype=np.float32)
    elif fam == "offset":
        off = (50.0 * rng.standard_normal(d)).astype(np.float32)
        db, qs = db + off, qs + off
    elif fam == "clustered":
        cent = (3.0 * rng.standard_normal((64, d))).astype(np.float32)
        db = (0.3 * db + cent[rng.integers(0, 64, n)]).astype(np.float32)
        qs = (0.3 * qs + cent[rng.integers(0, 64, nq)]).astype(np.float32)
    elif fam == "nonneg":
        db, qs = np.maximum(db, 0), np.maximum(qs, 0)
    elif fam == "sparse":
        db = db * (rng.random((n, d)) < 0.1)
        qs = qs * (rng.random((nq, d)) < 0.3)
    elif fam == "integers":
        db = rng.integers(-3, 4, (n, d)).astype(np.float32)
        qs = rng.integers(-3, 4, (nq, d)).astype(np.float32)
    elif fam == "duplicates":
        db[rng.integers(0, n, n // 3)] = db[rng.integers(0, n, n // 3)]
        qs[: nq // 2] = db[rng.integers(0, n, nq // 2)]
    elif fam == "outliers":
        db[rng.integers(0, n, 50)] *= np.float32(200.0)
        db[rng.integers(0, n, 20), rng.integers(0, d, 20)] = np.float32(1e4)
    elif fam == "tiny":
        db, qs = db * np.float32(1e-12), qs * np.float32(1e-12)
    elif fam == "huge":
        db, qs = db * np.float32(1e12), qs * np.float32(1e12)
    db, qs = np.ascontiguousarray(db, dtype=np.float32), np.ascontiguousarray(qs, dtype=np.float32)
    m = _lib.SQ_METRIC_L2 if metric == "euclidean" else _lib.SQ_METRIC_COSINE
    appended = 0
    with np.errstate(all="ignore"):
        if seed % 3 == 1:
            # the last rows arrive by one or two appends (the int8 copy follows: sq_dense.hip dense8_append)
            appended = int(rng.integers(1, n // 3))
            idx = _lib.DenseIndex(np.ascontiguousarray(db[: n - appended]), metric=m)
            cut = n - appended + int(rng.integers(0, appended + 1))
            if cut > n - appended:
                idx.append(np.ascontiguousarray(db[n - appended:cut]))
            if cut < n:
                idx.append(np.ascontiguousarray(db[cut:]))
        else:
            idx = _lib.DenseIndex(db, metric=m)
        idx.set_option("dense_int8", 1)
        d8, i8 = idx.search(qs, k)
        st8 = idx.stats()
        idx.set_option("dense_int8", 0)
        d16, i16 = idx.search(qs, k)
        st16 = idx.stats()
        int8_ran = st8["bytes_scanned"] != st16["bytes_scanned"]
        view = np.uint32 if metric == "euclidean" else np.uint64
        # (distances as bits: the NaN distances of zero vectors must be the same NaNs)
        same = np.array_equal(d8.view(view), d16.view(view)) and (metric == "cosine" or np.array_equal(i8, i16))
        if metric == "cosine" and not np.array_equal(i8, i16):
            # ids may differ only among exact ties of the float64 distance
            same = same and np.array_equal(np.sort(d8, axis=1).view(view), np.sort(d16, axis=1).view(view))
        qi = int(rng.integers(0, nq))
        rd, ri = O.dense_topk(db, qs[qi], k, metric)
        kk = len(rd)
        if metric == "euclidean":
            ok_or = np.array_equal(i8[qi, :kk], ri) and np.array_equal(d8[qi, :kk].view(np.uint32), rd.view(np.uint32))
        else:
            ok_or = np.allclose(d8[qi, :kk], rd, rtol=1e-12, atol=1e-15, equal_nan=True)
        idx.close()
    desc = (f"seed {seed}: n={n}{f' (+{appended} appended)' if appended else ''} d={d} nq={nq} k={k} {metric} {fam}: int8 {'ran' if int8_ran else 'declined'} "
            f"cands/q {st8['candidates'] / nq:.0f} later tiers {st8['mid_tier_queries']}+{st8['fallback_queries']} "
            f"same={same} oracle={ok_or}")
    return bool(same and ok_or), bool(int8_ran), desc


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = used8 = 0
    for seed in range(seed0, seed0 + cases):
        ok, ran, desc = run_case(seed)
        used8 += ran
        bad += 0 if ok else 1
        print(desc + ("" if ok else "  <-- MISMATCH"), flush=True)
    print(f"{cases} cases, int8 engaged in {used8}, mismatches {bad}")
    sys.exit(1 if bad else 0)
