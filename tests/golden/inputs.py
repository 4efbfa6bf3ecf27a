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


def lsh_inputs(n: int, d: int, seed: int):
    rng = np.random.default_rng(seed)
    db = rng.random((n, d))
    qs = rng.random((LSH_NQ, d))
    qs[0] = db[min(5, n - 1)]
    return db, qs
