"""
Seeded input generators shared by ``make_golden.py`` (which feeds them to the
real reference) and the parity tests (which feed the same arrays to the oracle
and to the HIP path).  Fixtures store a sha256 of each generated input so a
change in numpy's generators is detected instead of silently mis-comparing.
"""
from typing import Dict, Tuple

import numpy as np

# tag -> (n, d, bits, seed)
ITQ_CASES: Dict[str, Tuple[int, int, int, int]] = {
    "a": (600, 128, 64, 11),
    "b": (300, 96, 256, 12),
    "c": (257, 20, 7, 13),
}

# tag -> (n, bits, seed, mode)
HAMMING_CASES: Dict[str, Tuple[int, int, int, str]] = {
    "u64": (20000, 64, 21, "uniform"),
    "u256": (6000, 256, 22, "uniform"),
    "lowent64": (8000, 64, 23, "lowent"),
    "small70": (300, 70, 24, "uniform"),
}
HAMMING_KS = {
    "u64": (1, 10, 100),
    "u256": (1, 100),
    "lowent64": (10, 100),
    "small70": (7, 300),
}
HAMMING_NQ = 12

# tag -> (n, d, nq, seed, distribution, dtype)
DENSE_CASES = {
    "uni128": (6000, 128, 6, 31, "uniform", "float32"),
    "nrm128": (6000, 128, 6, 32, "normal", "float32"),
    "nrm200": (1500, 200, 3, 33, "normal", "float32"),
    "tiny5": (40, 5, 3, 34, "normal", "float32"),
}
DENSE_KMAX = 100

# SURVEY 8(c) G5 at its specified size: 20 k x 128, float32 AND float64, uniform and normal, 32 queries, the
# reference's order for k = 100 (k = 1 and 10 are its prefixes).  tag -> (n, d, nq, seed, distribution, dtype,
# cosine queries): separate fixture g5b_dense_nn_20k.npz.
DENSE_BIG_CASES = {
    "uni20k": (20000, 128, 32, 35, "uniform", "float32", 8),
    "nrm20k": (20000, 128, 32, 36, "normal", "float32", 8),
    "nrm20k_f64": (20000, 128, 32, 37, "normal", "float64", 8),
    "uni20k_f64": (20000, 128, 32, 38, "uniform", "float64", 8),
}

# tag -> (n, d, bits, seed, metric, n values)
LSH_CASES = {
    "rand_euclid": (1000, 64, 16, 41, "euclidean", (1, 10, 1000)),
    "rand_cosine": (500, 32, 8, 42, "cosine", (5,)),
}
LSH_NQ = 5


def itq_inputs(n: int, d: int, bits: int, seed: int):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((n, d)) * 1.5 + 0.25).astype(np.float32)
    x[0] = 0.0                                     # zero row: norm replaced by 1
    mean = rng.standard_normal(d) * 0.1 + 0.25
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    rot = np.ascontiguousarray(q[:, :bits]) if bits <= d else rng.standard_normal((d, bits))
    return x, mean.astype(np.float64), rot.astype(np.float64)


def hamming_inputs(n: int, bits: int, seed: int, mode: str):
    """Unique packed codes uint64[n',W] sorted ascending by integer value
    (row id = rank), and HAMMING_NQ packed queries."""
    rng = np.random.default_rng(seed)
    w = (bits + 63) // 64
    if mode == "uniform":
        codes = rng.integers(0, 2 ** 64, size=(n, w), dtype=np.uint64)
    else:
        centers = rng.integers(0, 2 ** 64, size=(5, w), dtype=np.uint64)
        pick = rng.integers(0, 5, size=n)
        codes = centers[pick]
        for _ in range(3):
            bit = rng.integers(0, 64, size=n).astype(np.uint64)
            word = rng.integers(0, w, size=n)
            codes[np.arange(n), word] ^= (np.uint64(1) << bit)
    queries = rng.integers(0, 2 ** 64, size=(HAMMING_NQ, w), dtype=np.uint64)
    pad = w * 64 - bits
    if pad:
        mask = np.uint64((1 << (64 - pad)) - 1)
        codes[:, 0] &= mask
        queries[:, 0] &= mask
    if mode != "uniform":
        queries[:4] = codes[:4]
        queries[4:8] = codes[4:8] ^ np.uint64(1)
    codes = np.unique(codes, axis=0)               # lexicographic == integer order
    return codes, queries


def dense_inputs(n: int, d: int, nq: int, seed: int, dist: str, dtype: str):
    rng = np.random.default_rng(seed)
    if dist == "uniform":
        db = rng.random((n, d))
        qs = rng.random((nq, d))
    else:
        db = rng.standard_normal((n, d))
        qs = rng.standard_normal((nq, d))
    db = db.astype(dtype)
    qs = qs.astype(dtype)
    if n > 20:
        db[7] = db[3]                              # exact duplicate rows: a tie
        qs[0] = db[11]                             # a self query: distance 0
    return db, qs


def lsh_scenario_random_euclidean():
    """The data of TestLshIndexAlgorithms._random_euclidean (tests/impls/nn_index/test_lsh.py:754-813):
    numpy's legacy generator seeded with RANDOM_SEED = 0, 1000 x 256 uniform rows drawn one by one, then the
    random query drawn AFTER the functor was trained (the training draws from the same global generator, so the
    fixture stores that query).  Returns (db, near-duplicate query of row 0)."""
    st = np.random.RandomState(0)          # == np.random.seed(0) followed by np.random.rand(dim) per row
    db = np.vstack([st.rand(256) for _ in range(1000)])
    v = db[0].copy()
    v_min = max(v.min(), 0.1)
    v[0] += v_min
    v[255] -= v_min
    return db, v


def lsh_scenario_known_ordered(order_seed: int = 0):
    """TestLshIndexAlgorithms._known_ordered_euclidean (test_lsh.py:934-966): rows (j, 2j), j < 1000, build order
    shuffled by python's `random` (seeded here; the test leaves it unseeded).  Returns (uuids in build order, rows)."""
    import random
    uu = list(range(1000))
    random.Random(order_seed).shuffle(uu)
    rows = np.array([[j, 2 * j] for j in uu], dtype=float)
    return uu, rows


def lsh_inputs(n: int, d: int, seed: int):
    rng = np.random.default_rng(seed)
    db = rng.random((n, d))
    qs = rng.random((LSH_NQ, d))
    qs[0] = db[min(5, n - 1)]
    return db, qs
