"""
ORACLE -- CPU restatement of the SMQTK-Indexing kNN hot path.  TEST
INFRASTRUCTURE ONLY: imported by ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py``; never by the product package
(``smqtk_indexing_amd``), which must fail loudly without its HIP library.

Parity pinning: every function here is checked against outputs of the real
reference (imported from ``/root/reference`` in the build container by
``tests/golden/make_golden.py``) through the committed fixtures in
``tests/golden/*.npz`` and against the reference's own known-answer tests
(SURVEY.md section 8c).  See ``tests/test_oracle_golden.py``.

Each function cites the reference file:line it follows (paths relative to the
reference root).  Nothing is copied: the reference is a handful of numpy
one-liners and python loops whose *arithmetic* (dtype, operation order,
tie behaviour) is restated here in vectorised form.

Canonical tie rule: the reference's order inside a
distance tie is implementation-defined (python ``set`` iteration order in
``linear.py:235-238``; bucket/set order + stable sort in ``lsh.py:491-514``).
The build fixes **(distance ascending, then row id ascending)**, which is what
a stable sort over rows in id order gives; the reference is compared with
tie-group-set semantics.
"""
import heapq
from math import pi
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np
from scipy.spatial.distance import cdist


# --------------------------------------------------------------------------
# bits  (smqtk_indexing/utils/bits.py)
# --------------------------------------------------------------------------
def bit_vector_to_int_large(v: np.ndarray) -> int:
    """bool[b] -> python int, element 0 is the MOST significant bit.
    Follows smqtk_indexing/utils/bits.py:4-20."""
    c = 0
    for b in v:
        c = (c << 1) + int(b)
    return c


def int_to_bit_vector_large(integer: int, bits: int = 0) -> np.ndarray:
    """python int -> bool[bits or minimal], big endian.
    Follows smqtk_indexing/utils/bits.py:23-56 (ValueError when ``bits`` is
    too small; a zero integer still occupies one position)."""
    size = max(int(integer).bit_length(), 1)
    if bits and bits < size:
        raise ValueError(
            "%d bits too small to represent integer value %d." % (bits, integer))
    n = bits or size
    v = np.zeros(n, np.bool_)
    i = 1
    while integer and i <= n:
        v[-i] = integer & 1
        integer >>= 1
        i += 1
    return v


def words_for_bits(bits: int) -> int:
    return (int(bits) + 63) // 64


def pack_bits_msb(bitvecs: np.ndarray) -> np.ndarray:
    """bool[n,b] -> uint64[n, ceil(b/64)], word 0 most significant; the
    integer value of the row equals ``bit_vector_to_int_large`` (bits.py:17-20),
    i.e. the code is right-aligned: padding zeros sit in the TOP of word 0."""
    a = np.atleast_2d(np.asarray(bitvecs)).astype(bool)
    n, b = a.shape
    w = words_for_bits(b)
    padded = np.zeros((n, w * 64), dtype=np.uint8)
    padded[:, w * 64 - b:] = a
    by = np.packbits(padded, axis=1, bitorder="big")          # [n, w*8] big endian
    return by.reshape(n, w, 8).view(">u8").reshape(n, w).astype(np.uint64)


def unpack_bits_msb(words: np.ndarray, bits: int) -> np.ndarray:
    """Inverse of :func:`pack_bits_msb`: uint64[n,w] -> bool[n,bits]."""
    wv = np.atleast_2d(np.asarray(words, dtype=np.uint64))
    n, w = wv.shape
    by = wv.astype(">u8").view(np.uint8).reshape(n, w * 8)
    allbits = np.unpackbits(by, axis=1, bitorder="big")
    return allbits[:, w * 64 - bits:].astype(bool)


def packed_to_int(row: np.ndarray) -> int:
    c = 0
    for w in np.asarray(row, dtype=np.uint64).tolist():
        c = (c << 64) | int(w)
    return c


def int_to_packed(value: int, words: int) -> np.ndarray:
    out = np.zeros(words, dtype=np.uint64)
    for i in range(words - 1, -1, -1):
        out[i] = value & 0xFFFFFFFFFFFFFFFF
        value >>= 64
    if value:
        raise ValueError("integer does not fit in %d words" % words)
    return out


# --------------------------------------------------------------------------
# metrics  (smqtk_indexing/utils/metrics.py)
# --------------------------------------------------------------------------
def hamming_distance(i: int, j: int) -> int:
    """popcount(i xor j) on python ints.  smqtk_indexing/utils/metrics.py:140-155."""
    return bin(i ^ j).count("1")


def euclidean_distance(i: np.ndarray, j: np.ndarray) -> np.ndarray:
    """sqrt(sum((i-j)^2)) along the last axis, dtype preserving.
    smqtk_indexing/utils/metrics.py:73-86.  NB numpy reduces the contiguous
    axis with its pairwise summation (see :func:`np_pairwise_sum_f32`)."""
    sum_axis = 1
    if i.ndim == 1 and j.ndim == 1:
        sum_axis = 0
    return np.sqrt(np.square(i - j).sum(sum_axis))


def cosine_similarity(i: np.ndarray, j: np.ndarray) -> np.ndarray:
    """1 - scipy cdist(...,'cosine') in float64.  metrics.py:89-117."""
    assert i.ndim == 1
    i = i.reshape(1, -1)
    if j.ndim == 1:
        j = j.reshape(1, -1)
    s = 1 - cdist(i, j, metric="cosine")[0]
    return s[0] if s.size == 1 else s


def cosine_distance(i: np.ndarray, j: np.ndarray, pos_vectors: bool = True) -> np.ndarray:
    """(1+pos)*arccos(clip(sim,-1,1))/pi.  metrics.py:120-137."""
    sim = np.maximum(np.minimum(cosine_similarity(i, j), 1), -1)
    return (1 + bool(pos_vectors)) * np.arccos(sim) / pi


def np_pairwise_sum_f32(a: np.ndarray) -> np.float32:
    """Scalar restatement of numpy's float add-reduce over a contiguous 1-D
    float32 array (numpy ``loops_utils.h.src`` ``@TYPE@_pairwise_sum``,
    numpy 2.2.6 as pinned in this image): n<8 plain loop; n<=128 eight
    interleaved accumulators combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))
    then the n%8 tail added serially; otherwise split at n/2 rounded down to
    a multiple of 8 and recurse.  The HIP exact-distance kernel reproduces
    this order so float32 distances (and hence tie structure) are bit
    identical to ``metrics.py:86``.  Checked against ``np.sum`` in
    tests/test_oracle_golden.py."""
    a = np.asarray(a, dtype=np.float32)
    n = a.shape[0]
    f = np.float32
    if n > 8192:
        # numpy's reduction iterates the array through its buffer, NPY_BUFSIZE = 8192 elements at a time: each buffer is
        # summed pairwise and the buffers' sums are added in order (np.sum over 8300 values is NOT a pairwise split at 4144)
        total = np_pairwise_sum_f32(a[:8192])
        for s in range(8192, n, 8192):
            total = f(total + np_pairwise_sum_f32(a[s:s + 8192]))
        return total
    if n < 8:
        res = f(0.0)
        for v in a:
            res = f(res + v)
        return res
    if n <= 128:
        r = [f(a[t]) for t in range(8)]
        i = 8
        while i < n - (n % 8):
            for t in range(8):
                r[t] = f(r[t] + a[i + t])
            i += 8
        res = f(f(f(r[0] + r[1]) + f(r[2] + r[3])) + f(f(r[4] + r[5]) + f(r[6] + r[7])))
        while i < n:
            res = f(res + a[i])
            i += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return f(np_pairwise_sum_f32(a[:n2]) + np_pairwise_sum_f32(a[n2:]))


# --------------------------------------------------------------------------
# ITQ functor  (smqtk_indexing/impls/lsh_functor/itq.py)
# --------------------------------------------------------------------------
def itq_norm_vector(v: np.ndarray, normalize) -> np.ndarray:
    """itq.py:172-191: v / ||v||_ord along the last axis, zero norms -> 1;
    identity when ``normalize`` is None.  dtype follows numpy (f32 in -> f32)."""
    if normalize is not None:
        n = np.linalg.norm(v, normalize, v.ndim - 1, keepdims=True)
        n[n == 0.] = 1.
        return v / n
    return v


def itq_get_hash(x: np.ndarray, mean_vec: np.ndarray, rotation: np.ndarray,
                 normalize=None) -> np.ndarray:
    """itq.py:389-408: bits = ((norm(x) - mean) . R) >= 0 (float64 because
    mean/rotation are float64; exact zero maps to True).  Works on [d] or [n,d]."""
    z = np.dot(itq_norm_vector(x, normalize) - mean_vec, rotation)
    b = np.zeros(z.shape, dtype=bool)
    b[z >= 0] = True
    return b


def itq_z(x: np.ndarray, mean_vec: np.ndarray, rotation: np.ndarray, normalize=None) -> np.ndarray:
    """The pre-sign projection of :func:`itq_get_hash` (for margin-aware parity)."""
    return np.dot(itq_norm_vector(x, normalize) - mean_vec, rotation)


def itq_find_rotation(v: np.ndarray, n_iter: int, random_seed: Optional[int]
                      ) -> Tuple[np.ndarray, np.ndarray]:
    """itq.py:239-289 (random orthogonal init from SVD of randn, then
    n_iter x {sign, SVD of ux^T v, r = ua . ub^T})."""
    bit = v.shape[1]
    if random_seed is not None:
        np.random.seed(random_seed)
    r = np.random.randn(bit, bit)
    u11, _, _ = np.linalg.svd(r)
    r = u11[:, :bit]
    for _ in range(n_iter):
        z = np.dot(v, r)
        ux = np.ones(z.shape) * (-1)
        ux[z >= 0] = 1
        c = np.dot(ux.transpose(), v)
        ub, _, ua = np.linalg.svd(c)
        r = np.dot(ua, ub.transpose())
    z = np.dot(v, r)
    b = np.zeros(z.shape, dtype=bool)
    b[z >= 0] = True
    return b, r


def itq_fit(x: np.ndarray, bit_length: int, n_iter: int, normalize,
            random_seed: Optional[int]) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """itq.py:291-387 on an [n,d] matrix -> (mean_vec, rotation[d,b], codes)."""
    x = np.array(x)
    if x.shape[1] < bit_length:
        raise ValueError("Input descriptors have fewer features than requested bit encoding.")
    x = itq_norm_vector(x, normalize)
    mean_vec = np.mean(x, axis=0)
    x = x - mean_vec
    c = np.cov(x.transpose())
    c = np.atleast_2d(c)
    l, pc = np.linalg.eig(c)
    order = sorted(zip(l, pc.transpose()), key=lambda p: p[0], reverse=True)
    pc_top = np.array([p[1] for p in order[:bit_length]]).transpose()
    v = np.dot(x, pc_top)
    codes, rot = itq_find_rotation(v, n_iter, random_seed)
    return mean_vec, np.dot(pc_top, rot), codes


# --------------------------------------------------------------------------
# LinearHashIndex  (smqtk_indexing/impls/hash_index/linear.py)
# --------------------------------------------------------------------------
def linear_hash_nn_reference(index: Iterable[int], h: np.ndarray, n: int
                             ) -> Tuple[np.ndarray, Tuple[float, ...]]:
    """linear.py:206-244 verbatim in behaviour: ``heapq.nsmallest`` over the
    python-int set keyed by popcount(xor); ties keep iteration order; returns
    bool rows and distance/bits."""
    h_int = bit_vector_to_int_large(h)
    bits = len(h)
    near = heapq.nsmallest(n, index, lambda e: hamming_distance(h_int, e))
    dists = [hamming_distance(c, h_int) for c in near]
    rows = np.vstack([int_to_bit_vector_large(c, bits) for c in near])
    return rows, tuple(d / float(bits) for d in dists)


def popcount_u64(a: np.ndarray) -> np.ndarray:
    return np.bitwise_count(np.asarray(a, dtype=np.uint64))


def hamming_topk(codes: np.ndarray, q: np.ndarray, k: int
                 ) -> Tuple[np.ndarray, np.ndarray]:
    """Vectorised canonical restatement of linear.py:235-240 over packed codes:
    codes uint64[N,W] (unique rows, any order = row ids), q uint64[W].
    Returns (dist int32[k'], idx int64[k']) sorted by (dist, idx), k'=min(k,N)."""
    codes = np.atleast_2d(np.asarray(codes, dtype=np.uint64))
    q = np.asarray(q, dtype=np.uint64).reshape(1, -1)
    d = popcount_u64(codes ^ q).sum(axis=1).astype(np.int64)
    k = min(int(k), codes.shape[0])
    key = (d << 40) | np.arange(codes.shape[0], dtype=np.int64)
    if k < codes.shape[0]:
        part = np.argpartition(key, k - 1)[:k]
    else:
        part = np.arange(codes.shape[0])
    part = part[np.argsort(key[part], kind="stable")]
    return d[part].astype(np.int32), part.astype(np.int64)


# --------------------------------------------------------------------------
# Dense brute force  (metrics.py + stable sort + slice, as lsh.py:505-519 /
# faiss.py:751-831 express it)
# --------------------------------------------------------------------------
def dense_distances(db: np.ndarray, q: np.ndarray, metric: str, chunk: int = 1 << 18) -> np.ndarray:
    """All N reference distances from ``q`` to the rows of ``db`` using the
    reference formulas (metrics.py:73-86 for 'euclidean' in the array dtype;
    metrics.py:120-137 for 'cosine' in float64)."""
    out = []
    for s in range(0, db.shape[0], chunk):
        blk = db[s:s + chunk]
        if metric == "euclidean":
            out.append(euclidean_distance(blk, q))
        elif metric == "cosine":
            out.append(np.atleast_1d(cosine_distance(q, blk)))
        else:
            raise ValueError(metric)
    return np.concatenate(out)


def dense_topk(db: np.ndarray, q: np.ndarray, k: int, metric: str = "euclidean"
               ) -> Tuple[np.ndarray, np.ndarray]:
    """Exact brute force = distances + stable ascending sort + slice
    (lsh.py:511-518).  Returns (dist[k'], idx int64[k']), ties by row id."""
    d = dense_distances(db, q, metric)
    k = min(int(k), d.shape[0])
    order = np.argsort(d, kind="stable")[:k]
    return d[order], order.astype(np.int64)


# --------------------------------------------------------------------------
# LSH orchestration  (smqtk_indexing/impls/nn_index/lsh.py:452-519)
# --------------------------------------------------------------------------
def lsh_nn(q_vec: np.ndarray, n: int, mean_vec: np.ndarray, rotation: np.ndarray,
           normalize, uniq_codes: np.ndarray, code_rows: Sequence[Sequence[int]],
           db: np.ndarray, metric: str) -> Tuple[np.ndarray, np.ndarray]:
    """Restates lsh.py:452-519 over array-form state: ``uniq_codes`` uint64[U,W]
    (the hash index content), ``code_rows[u]`` = DB row ids in bucket u (the
    hash2uuids kvstore), ``db`` the descriptor matrix.  Steps: hash query ->
    n nearest unique codes (canonical order) -> expand buckets -> exact
    distance per candidate row -> stable sort -> top n.  Returns (row ids, dists)."""
    bits = rotation.shape[1]
    qb = itq_get_hash(q_vec, mean_vec, rotation, normalize)
    qp = pack_bits_msb(qb[None, :])[0]
    _, near = hamming_topk(uniq_codes, qp, n)
    cand: List[int] = []
    for u in near.tolist():
        cand.extend(code_rows[u])
    cand_a = np.asarray(cand, dtype=np.int64)
    vecs = db[cand_a]
    if metric == "euclidean":
        d = np.array([euclidean_distance(q_vec, v) for v in vecs])
    elif metric == "cosine":
        d = np.array([cosine_distance(q_vec, v) for v in vecs])
    else:
        raise ValueError(metric)
    order = np.argsort(d, kind="stable")[:n]
    del bits
    return cand_a[order], d[order]


# --------------------------------------------------------------------------
# parity helpers shared by the tests
# --------------------------------------------------------------------------
def assert_topk_equivalent(ref_dist: np.ndarray, ref_idx: np.ndarray,
                           got_dist: np.ndarray, got_idx: np.ndarray,
                           all_dist_of=None, rtol: float = 0.0) -> None:
    """Tie-group-set comparison (test_linear.py:150-155 semantics): distance
    sequences equal (within rtol); indices equal as sets inside every group of
    equal reference distance that lies wholly inside the top-k; for the group
    straddling rank k any members of that group are accepted (checked through
    ``all_dist_of(idx) -> reference distance`` when given)."""
    ref_dist = np.asarray(ref_dist)
    got_dist = np.asarray(got_dist)
    assert ref_dist.shape == got_dist.shape, (ref_dist.shape, got_dist.shape)
    if rtol:
        np.testing.assert_allclose(got_dist, ref_dist, rtol=rtol, atol=0)
    else:
        np.testing.assert_array_equal(got_dist, ref_dist)
    k = len(ref_dist)
    s = 0
    while s < k:
        e = s
        while e + 1 < k and ref_dist[e + 1] == ref_dist[s]:
            e += 1
        last_group = (e == k - 1)
        a = set(np.asarray(ref_idx[s:e + 1]).tolist())
        b = set(np.asarray(got_idx[s:e + 1]).tolist())
        if not last_group:
            assert a == b, (s, e, a ^ b)
        else:
            assert len(b) == e - s + 1, "duplicate index in result"
            if all_dist_of is not None:
                for i in b - a:
                    assert all_dist_of(i) == ref_dist[s], (i, all_dist_of(i), ref_dist[s])
        s = e + 1
