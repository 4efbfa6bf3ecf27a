#!/usr/bin/env python3
"""Measurement helper: wall time of the three device calls behind one LSH query (hash, Hamming search, re-rank)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from smqtk_indexing_amd import _lib
rng = np.random.default_rng(0)
n, d, bits = 1_000_000, 128, 64
x = rng.standard_normal((n, d)).astype(np.float32)
q, _ = np.linalg.qr(rng.standard_normal((d, d)))
rot = np.ascontiguousarray(q[:, :bits]); mean = x[:10000].mean(0).astype(np.float64)
codes = _lib.itq_hash(x, mean, rot, _lib.SQ_NORM_NONE)
uc = np.unique(codes, axis=0)
hidx = _lib.HammingIndex(uc)
rows = _lib.RowMatrix(x)
q1 = rng.standard_normal((1, d)).astype(np.float32)
def t(fn, reps=200):
    for _ in range(10): fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e6
print("itq_hash 1 row: %.1f us" % t(lambda: _lib.itq_hash(q1, mean, rot, _lib.SQ_NORM_NONE)))
model = _lib.ItqModel(mean, rot)
assert np.array_equal(model.hash(x[:5000]), codes[:5000])
print("resident model, 1 row: %.1f us" % t(lambda: model.hash(q1)))
qc = _lib.itq_hash(q1, mean, rot, _lib.SQ_NORM_NONE)
print("hamming search 1 q, k=100: %.1f us" % t(lambda: hidx.search(qc, 100)))
cand = rng.integers(0, n, 300).astype(np.int64); off = np.array([0, 300], np.int64)
print("rows rerank 1 q x 300 cand: %.1f us" % t(lambda: rows.rerank(q1, _lib.SQ_METRIC_L2, cand, off, 100)))
print("rows rerank cosine: %.1f us" % t(lambda: rows.rerank(q1, _lib.SQ_METRIC_COSINE, cand, off, 100)))
