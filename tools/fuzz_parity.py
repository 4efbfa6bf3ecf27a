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
    k = int(rng.choice([1, 5, 50, 100, 600, 3000, 20000]))
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
        if n > 64 and rng.integers(0, 3) == 0:            # built in pieces: create + sq_dense_append
            cuts = np.sort(rng.choice(np.arange(1, n), size=int(rng.integers(1, 4)), replace=False))
            idx = _lib.DenseIndex(db[:cuts[0]], metric=m)
            for a_, b_ in zip(cuts, list(cuts[1:]) + [n]):
                idx.append(db[a_:b_])
        else:
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
# ---- Hamming: random widths / sizes / tie-heavy code sets
for c in range(cases // 2):
    bits = int(rng.choice([1, 7, 33, 64, 70, 128, 192, 256, 320]))
    n = int(rng.choice([50, 3000, 90_000, 300_000]))
    k = int(rng.choice([1, 10, 100, 700, 5000]))
    nq = int(rng.choice([1, 5, 40]))
    w = (bits + 63) // 64
    low = bool(rng.integers(0, 2))                                   # low-entropy codes: huge tie groups
    raw = rng.integers(0, 2 ** 63, size=(n, w), dtype=np.int64).astype(np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, w)).astype(np.uint64)
    if low:
        raw &= np.uint64(0x0F0F000000000F0F)
    pad = w * 64 - bits
    if pad:
        raw[:, 0] &= np.uint64((1 << (64 - pad)) - 1) if pad < 64 else np.uint64(0)
    codes = np.unique(raw, axis=0)
    qs = codes[rng.integers(0, len(codes), nq)].copy()
    qs[0] ^= np.uint64(1)
    try:
        idx = _lib.HammingIndex(codes)
        dd, ii = idx.search(qs, k)
        st = idx.stats()
        for qi in range(min(nq, 3)):
            rd, ri = O.hamming_topk(codes, qs[qi], k)
            kk = len(rd)
            assert np.array_equal(dd[qi, :kk], rd), "hamming dist"
            assert np.array_equal(ii[qi, :kk], ri), "hamming ids"
            assert (ii[qi, kk:] == -1).all(), "hamming padding"
        if len(codes) > 20 and rng.integers(0, 2):                  # in-place removal + re-insertion (sq_hamming_remove/append)
            drop = np.sort(rng.choice(len(codes), size=int(rng.integers(1, min(len(codes) - 1, 500))), replace=False))
            idx.remove(drop)
            rest = np.delete(codes, drop, axis=0)
            dd, ii = idx.search(qs, k)
            rd, ri = O.hamming_topk(rest, qs[0], k)
            assert np.array_equal(dd[0, :len(rd)], rd) and np.array_equal(ii[0, :len(rd)], ri), "hamming after remove"
            back = codes[drop]
            be = [tuple(int(v) for v in r) for r in rest]
            import bisect
            pos = np.array([bisect.bisect_left(be, tuple(int(v) for v in r)) for r in back], dtype=np.int64)
            idx.append(back, pos)
            dd, ii = idx.search(qs, k)
            rd, ri = O.hamming_topk(codes, qs[0], k)
            assert np.array_equal(dd[0, :len(rd)], rd) and np.array_equal(ii[0, :len(rd)], ri), "hamming after append"
        idx.close()
        print(f"ok   hamming {c}: n={len(codes)} bits={bits} k={k} nq={nq} low={low} fallback={st['fallback_queries']}", flush=True)
    except Exception as e:  # noqa: BLE001
        fails += 1
        print(f"FAIL hamming {c}: n={len(codes)} bits={bits} k={k} nq={nq} low={low}: {type(e).__name__} {e}", flush=True)

# ---- ITQ: random dims / bits / dtypes / normalisation / mean dtype
for c in range(cases // 2):
    d = int(rng.choice([3, 20, 64, 100, 128, 192, 256, 300]))
    bits = int(rng.choice([1, 8, 33, 64, 100, 128, 200]))
    n = int(rng.choice([1, 31, 33, 1000, 40_000, 300_000]))
    xdt = np.float32 if rng.integers(0, 2) else np.float64
    mdt = np.float32 if rng.integers(0, 2) else np.float64
    norm = [None, None, 2, 2, 1, 0, float('inf'), float('-inf')][int(rng.integers(0, 8))]
    code = {None: _lib.SQ_NORM_NONE, 2: _lib.SQ_NORM_L2, 1: _lib.SQ_NORM_L1, 0: _lib.SQ_NORM_L0,
            float('inf'): _lib.SQ_NORM_INF, float('-inf'): _lib.SQ_NORM_NEG_INF}[norm]
    x = (rng.standard_normal((n, d)) * rng.lognormal(0, 1, (n, 1)) + rng.standard_normal(d)).astype(xdt)
    if n > 3: x[2] = 0
    mean = x[: max(1, n // 2)].mean(axis=0).astype(mdt)
    rot = rng.standard_normal((d, bits))
    try:
        got = _lib.itq_hash(x, mean, rot, code)
        z = O.itq_z(x, mean, rot, norm)
        ref = O.pack_bits_msb(z >= 0)
        bad = (got != ref).any(axis=1)
        if bad.any():
            scale = np.abs(z[bad]).max(axis=1)
            assert (np.abs(z[bad]).min(axis=1) <= 1e-9 * np.maximum(scale, 1e-30)).all(), "itq bits"
        print(f"ok   itq {c}: n={n} d={d} bits={bits} x={xdt.__name__} mean={mdt.__name__} norm={norm} borderline_rows={int(bad.sum())}", flush=True)
    except Exception as e:  # noqa: BLE001
        fails += 1
        print(f"FAIL itq {c}: n={n} d={d} bits={bits} x={xdt.__name__} mean={mdt.__name__} norm={norm}: {type(e).__name__} {e}", flush=True)

# ---- re-rank stage (sq_rows_rerank): random candidate lists, both dtypes and metrics
for c in range(cases // 2):
    dt = np.float32 if rng.integers(0, 2) else np.float64
    n = int(rng.choice([1, 40, 3000, 50_000]))
    d = int(rng.choice([1, 5, 64, 96, 128, 200, 512]))
    k = int(rng.choice([1, 10, 100, 400]))
    nq = int(rng.choice([1, 4, 17]))
    name = str(rng.choice(["euclidean", "cosine"]))
    rows = data(n, d, int(rng.integers(0, 5))).astype(dt)
    qs = data(nq, d, 0).astype(dt)
    cands = [rng.integers(0, n, int(rng.choice([0, 1, 7, 300, 5000]))) for _ in range(nq)]   # duplicates allowed
    off = np.zeros(nq + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(x) for x in cands])
    flat = np.concatenate(cands).astype(np.int64) if off[-1] else np.zeros(0, np.int64)
    try:
        m = _lib.RowMatrix(rows)
        dist, pos = m.rerank(qs, _lib.SQ_METRIC_L2 if name == "euclidean" else _lib.SQ_METRIC_COSINE, flat, off, k)
        for qi, cd in enumerate(cands):
            full = O.dense_distances(rows[cd], qs[qi], name) if len(cd) else np.zeros(0)
            order = np.argsort(full, kind="stable")[:k]
            kk = len(order)
            assert (pos[qi, kk:] == -1).all(), "rows padding"
            if name == "euclidean":
                assert np.array_equal(pos[qi, :kk], order), "rows order"
                assert np.array_equal(dist[qi, :kk], full[order]), "rows dist"
            else:
                assert np.allclose(dist[qi, :kk], full[order], rtol=1e-12, atol=1e-15, equal_nan=True), "rows cos dist"
                mism = pos[qi, :kk] != order
                if mism.any():
                    a, b = full[pos[qi, :kk][mism]], full[order[mism]]
                    assert np.all((np.abs(a - b) < 1e-14) | (np.isnan(a) & np.isnan(b))), "rows cos order"
        m.close()
        print(f"ok   rows {c}: n={n} d={d} k={k} nq={nq} {name} {dt.__name__}", flush=True)
    except Exception as e:  # noqa: BLE001
        fails += 1
        print(f"FAIL rows {c}: n={n} d={d} k={k} nq={nq} {name} {dt.__name__}: {type(e).__name__} {e}", flush=True)

# ---- the LSH plugin: device mirror (CSR expansion + device re-rank) against the host path
from smqtk_indexing_amd._compat import DescriptorMemoryElement, MemoryDescriptorSet, MemoryKeyValueStore
from smqtk_indexing_amd.impls.hash_index.hip_linear import HipLinearHashIndex
from smqtk_indexing_amd.impls.lsh_functor.hip_itq import HipItqFunctor
from smqtk_indexing_amd.impls.nn_index.hip_lsh import HipLSHNearestNeighborIndex
for c in range(max(2, cases // 8)):
    dt = np.float32 if rng.integers(0, 2) else np.float64
    n = int(rng.choice([200, 2000, 6000]))
    d = int(rng.choice([8, 64, 100]))
    bits = int(rng.choice([4, 10, 24]))
    name = str(rng.choice(["euclidean", "cosine"]))
    use_hi = bool(rng.integers(0, 2))
    x = data(n, d, int(rng.choice([0, 3]))).astype(dt)
    try:
        f = HipItqFunctor(bit_length=min(bits, d), itq_iterations=4, random_seed=int(rng.integers(0, 100)))
        f.fit([DescriptorMemoryElement(i).set_vector(v) for i, v in enumerate(x[: max(50, n // 5)])])
        idxs = []
        for device in (True, False):
            ix = HipLSHNearestNeighborIndex(f, MemoryDescriptorSet(), MemoryKeyValueStore(),
                                            HipLinearHashIndex() if use_hi else None, distance_method=name,
                                            device_rerank=device)
            ix.build_index([DescriptorMemoryElement(i).set_vector(v) for i, v in enumerate(x)])
            idxs.append(ix)
        dev, host = idxs
        qv = data(5, d, 0).astype(dt)
        qv[0] = x[rng.integers(0, n)]
        qs = [DescriptorMemoryElement(f"q{i}").set_vector(v) for i, v in enumerate(qv)]
        for nn in (1, 13, 120):
            batch = dev.nn_many(qs, nn)
            for q, b in zip(qs, batch):
                for (ra, da), (rb, db_) in ((dev.nn(q, nn), host.nn(q, nn)), (b, host.nn(q, nn))):
                    assert len(ra) == len(rb), "lsh count"
                    assert np.allclose(da, db_, rtol=1e-12, atol=0, equal_nan=True), "lsh dist"
                    da = np.asarray(da)
                    # the same ids wherever the two answers are not split by a distance tie (duplicate rows at the cut:
                    # the host path takes bucket members in python-set order, the mirror in row order); every id must
                    # carry its own distance
                    ua, ub = [e.uuid() for e in ra], [e.uuid() for e in rb]
                    for uu, dd_ in ((ua, da), (ub, np.asarray(db_))):
                        if len(uu):
                            true = O.dense_distances(x[np.asarray(uu)], np.asarray(q.vector()), name)
                            assert np.allclose(true, dd_, rtol=1e-12, atol=1e-15, equal_nan=True), "lsh id/distance"
                    for i_ in range(len(ua)):
                        if ua[i_] != ub[i_]:
                            assert np.array_equal(x[ua[i_]], x[ub[i_]]) or abs(float(da[i_]) - float(np.asarray(db_)[i_])) == 0.0, "lsh order"
        print(f"ok   lsh {c}: n={n} d={d} bits={min(bits, d)} {name} {dt.__name__} hash_index={use_hi}", flush=True)
    except Exception as e:  # noqa: BLE001
        fails += 1
        print(f"FAIL lsh {c}: n={n} d={d} bits={min(bits, d)} {name} {dt.__name__} hash_index={use_hi}: {type(e).__name__} {e}", flush=True)

total = cases + 3 * (cases // 2) + max(2, cases // 8)
print(f"{total - fails}/{total} passed in {time.time() - t0:.0f} s")
sys.exit(1 if fails else 0)
