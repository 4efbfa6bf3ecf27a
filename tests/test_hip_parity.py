"""
GPU parity tests: the HIP path, called through the C ABI (ctypes), against the
oracle on the same seeded inputs and against the golden vectors produced by the
real reference.  Integer / index results must be identical; float32 L2
distances bit-identical; float64 cosine distances within 1e-12 relative
(north_star allows 1e-5).
"""
import numpy as np
import pytest

from oracle import cpu_ref as O
from smqtk_indexing_amd import _lib
from tests.golden import inputs as GI

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _reset_options():
    yield
    for name in ("candidate_cap", "sample_stride", "force_fallback", "profile", "dense_stages", "dense_blocks", "dense_qt",
                 "itq_exact", "hamming_no_permute", "dense_no_center", "dense_qplanes"):
        _lib.set_option(name, 0)
    _lib.set_option("dense_async_streams", 2)
    _lib.set_option("dense_mid_tier", 1)
    _lib.set_option("dense_fused_prep", 1)
    _lib.set_option("dense_int8", -1)
    _lib.set_option("dense_graph", 1)
    _lib.set_option("dense_int8_batch", 64)


# ------------------------------------------------------------------- Hamming
def _hamming_check(codes, queries, k):
    idx = _lib.HammingIndex(codes)
    d, i = idx.search(queries, k)
    for qi, q in enumerate(queries):
        rd, ri = O.hamming_topk(codes, q, k)
        kk = len(rd)
        np.testing.assert_array_equal(d[qi, :kk], rd)
        np.testing.assert_array_equal(i[qi, :kk], ri)
        assert (i[qi, kk:] == -1).all()
    return idx


@pytest.mark.parametrize("tag", list(GI.HAMMING_CASES))
def test_hamming_golden_cases(golden, tag):
    g = golden("g4_linear_hash_nn.npz")
    n, bits, seed, mode = GI.HAMMING_CASES[tag]
    codes, queries = GI.hamming_inputs(n, bits, seed, mode)
    lut = {O.packed_to_int(r): i for i, r in enumerate(codes)}
    for k in GI.HAMMING_KS[tag]:
        idx = _hamming_check(codes, queries, k)
        d, i = idx.search(queries, k)
        # against the reference's own output (tie-group semantics, test_linear.py:150-155)
        rcodes, rdist = g[f"{tag}_k{k}_codes"], g[f"{tag}_k{k}_dist"]
        for qi, q in enumerate(queries):
            ref_d = np.rint(rdist[qi] * bits).astype(np.int32)
            ref_i = np.array([lut[O.packed_to_int(r)] for r in rcodes[qi]])
            full = O.popcount_u64(codes ^ q[None, :]).sum(axis=1)
            kk = len(ref_d)
            O.assert_topk_equivalent(ref_d, ref_i, d[qi, :kk], i[qi, :kk], all_dist_of=lambda r: full[r])


@pytest.mark.parametrize("bits,n,nq,k", [(64, 200_000, 40, 100), (128, 150_000, 7, 10), (256, 120_000, 9, 100),
                                         (192, 90_000, 5, 33), (64, 70_001, 3, 1)])
def test_hamming_scan_path(bits, n, nq, k):
    """n above the candidate cap: sample histogram -> threshold -> emit -> select."""
    rng = np.random.default_rng(bits + n)
    w = bits // 64
    codes = np.unique(rng.integers(0, 2 ** 64, size=(n, w), dtype=np.uint64), axis=0)
    queries = rng.integers(0, 2 ** 64, size=(nq, w), dtype=np.uint64)
    queries[0] = codes[17]
    idx = _hamming_check(codes, queries, k)
    st = idx.stats()
    assert st["fallback_queries"] == 0
    assert st["candidates"] >= nq * k


def test_hamming_sorted_codes_large_index_no_fallback():
    """The host index keeps its codes sorted; near codes then sit in a few narrow row ranges.  The
    device copy is stored in a low-discrepancy permutation so that the block sample and the
    per-workgroup survivor lists still see an even spread (no query may fall to the exact path),
    while row ids and tie order stay those of the sorted array."""
    rng = np.random.default_rng(123)
    codes = np.unique(rng.integers(0, 2 ** 64, size=(3_000_000, 1), dtype=np.uint64), axis=0)
    queries = np.concatenate([codes[rng.integers(0, codes.shape[0], 20)],
                              rng.integers(0, 2 ** 64, size=(12, 1), dtype=np.uint64)])
    idx = _hamming_check(codes, queries, 100)
    st = idx.stats()
    assert st["fallback_queries"] == 0, st
    _lib.set_option("hamming_no_permute", 1)
    try:
        _hamming_check(codes[:500_000], queries[:4], 10)      # caller-order layout still answers identically
    finally:
        _lib.set_option("hamming_no_permute", 0)


def test_hamming_low_entropy_overflow_and_fallback():
    """Clustered codes: huge tie groups overflow the candidate list, the exact
    full-keys path must still give the canonical answer."""
    codes, queries = GI.hamming_inputs(150_000, 64, 77, "lowent")
    _lib.set_option("candidate_cap", 2048)
    idx = _hamming_check(codes, queries, 50)
    assert idx.stats()["fallback_queries"] > 0
    _lib.set_option("candidate_cap", 0)
    _lib.set_option("force_fallback", 1)
    idx = _hamming_check(codes[:100_000], queries[:3], 20)
    assert idx.stats()["fallback_queries"] == 3


def test_hamming_k_larger_than_n_and_id_base():
    codes = np.unique(np.random.default_rng(5).integers(0, 2 ** 64, size=(50, 1), dtype=np.uint64), axis=0)
    idx = _lib.HammingIndex(codes, id_base=1000)
    d, i = idx.search(codes[:2], 64)
    rd, ri = O.hamming_topk(codes, codes[0], 64)
    np.testing.assert_array_equal(i[0, :len(ri)], ri + 1000)
    assert (i[0, len(ri):] == -1).all() and (d[0, len(ri):] == np.iinfo(np.int32).max).all()


# --------------------------------------------------------------------- dense
def _dense_check(db, qs, k, metric="euclidean", exact_dist=True, options=None):
    m = _lib.SQ_METRIC_L2 if metric == "euclidean" else _lib.SQ_METRIC_COSINE
    idx = _lib.DenseIndex(db, metric=m, options=options)      # (options: this index's own, sq_dense_create_opts)
    d, i = idx.search(qs, k)
    for qi, q in enumerate(qs):
        rd, ri = O.dense_topk(db, q, k, metric)
        kk = len(rd)
        if metric == "euclidean":
            assert d.dtype == np.float32
            # float32 distances bit-identical to numpy's evaluation of metrics.py:86
            np.testing.assert_array_equal(d[qi, :kk].view(np.uint32), rd.view(np.uint32))
            np.testing.assert_array_equal(i[qi, :kk], ri)
        else:
            np.testing.assert_allclose(d[qi, :kk], rd, rtol=1e-12, atol=1e-15)
            full = O.dense_distances(db, q, "cosine")
            # identical ranks wherever the reference distances are distinguishable
            mism = i[qi, :kk] != ri
            if mism.any():
                assert np.abs(full[i[qi, :kk][mism]] - full[ri[mism]]).max() < 1e-14
    return idx


def _int8_bytes(n, d=128):
    row = 128 if d <= 128 else (256 if d <= 256 else 512)
    return (-(-n // 64) * 64) * (row + 4)


def _bf16_bytes(n, d):
    return (-(-n // 32) * 32) * (-(-d // 128) * 256 + 4)


def _int8_family(rng, family, n, d, nq):
    db = rng.standard_normal((n, d)).astype(np.float32)
    qs = rng.standard_normal((nq, d)).astype(np.float32)
    if family == "uniform":
        db = rng.random((n, d), dtype=np.float32)
        qs = rng.random((nq, d), dtype=np.float32)
    elif family == "clustered":
        cent = (4.0 * rng.standard_normal((200, d))).astype(np.float32)
        db = (db + cent[rng.integers(0, 200, n)]).astype(np.float32)
        qs = (qs + cent[rng.integers(0, 200, nq)]).astype(np.float32)
    elif family == "scaled":
        db = (db * np.float32(37.5) + np.float32(11.0)).astype(np.float32)
        qs = (qs * np.float32(37.5) + np.float32(11.0)).astype(np.float32)
    elif family == "near_rows":
        # queries next to rows of the index (the usual case of a descriptor looked up in its own index), duplicates included
        qs = (db[rng.integers(0, n, nq)] + np.float32(0.05) * qs).astype(np.float32)
        qs[0] = db[123]
        db[4567] = db[123]
    return db, qs


@pytest.mark.parametrize("n,d,nq,k,family", [(200_000, 128, 32, 100, "normal"), (150_001, 100, 7, 10, "uniform"),
                                              (100_000, 65, 1, 1, "normal"), (180_000, 128, 20, 1000, "clustered"),
                                              (70_000, 17, 32, 50, "scaled"), (130_000, 128, 31, 100, "near_rows"),
                                              (66_000, 64, 5, 3, "clustered"),
                                              # rows of 256 and 512 bytes (32-row ring units; four waves per workgroup at 512)
                                              (120_000, 256, 32, 100, "normal"), (90_001, 200, 9, 10, "uniform"),
                                              (100_000, 512, 32, 100, "normal"), (80_000, 300, 17, 25, "clustered"),
                                              (70_003, 384, 3, 1, "near_rows"), (66_000, 129, 32, 7, "scaled")])
def test_dense_int8_filter_equals_bf16_filter_and_oracle(n, d, nq, k, family):
    """The int8 first-stage filter (sq_dense_i8.hpp: L2, d <= 512, one query tile, n >= 65536) is a filter only: survivors
    are re-ranked in the reference's float32 arithmetic and every query is certified against the measured error bound, so
    neighbours and distance bits equal the bf16 filter's and the oracle's -- on benign data without a query leaving the
    first tier.  sq_stats_t.bytes_scanned tells which copy the full pass streamed."""
    rng = np.random.default_rng(n + d + k)
    db, qs = _int8_family(rng, family, n, d, nq)
    idx = _lib.DenseIndex(db)
    idx.set_option("dense_int8", 1)
    d8, i8 = idx.search(qs, k)
    st = idx.stats()
    assert st["bytes_scanned"] == _int8_bytes(n, d), st
    assert st["fallback_queries"] == 0 and st["mid_tier_queries"] == 0, st
    idx.set_option("dense_int8", 0)
    d16, i16 = idx.search(qs, k)
    assert idx.stats()["bytes_scanned"] == _bf16_bytes(n, d)
    np.testing.assert_array_equal(i8, i16)
    np.testing.assert_array_equal(d8.view(np.uint32), d16.view(np.uint32))
    for qi in range(0, nq, max(1, nq // 6)):
        rd, ri = O.dense_topk(db, qs[qi], k)
        np.testing.assert_array_equal(i8[qi], ri)
        np.testing.assert_array_equal(d8[qi].view(np.uint32), rd.view(np.uint32))
    idx.close()


@pytest.mark.parametrize("n,d,nq,k,family", [(150_000, 128, 32, 100, "normal"), (100_000, 512, 32, 50, "clustered"),
                                              (90_001, 100, 7, 10, "uniform"), (80_000, 300, 20, 25, "normal")])
def test_dense_int8_filter_cosine(n, d, nq, k, family):
    """Cosine with the int8 first stage: the copy holds the unit-length rows (row term 0), the planes -q / |q|; survivors
    are re-ranked by the float64 cosine kernel, so the answers are the bf16 filter's bit for bit and the oracle's within
    its 1e-12.  A zero row and a zero query keep the reference's behaviour (NaN distance, ranked last / all NaN)."""
    rng = np.random.default_rng(n + d)
    db, qs = _int8_family(rng, family, n, d, nq)
    db[777] = 0.0
    qs[nq - 1] = 0.0
    with np.errstate(invalid="ignore", divide="ignore"):
        idx = _lib.DenseIndex(db, metric=_lib.SQ_METRIC_COSINE)
        idx.set_option("dense_int8", 1)
        d8, i8 = idx.search(qs, k)
        st = idx.stats()
        assert st["bytes_scanned"] >= _int8_bytes(n, d) and (st["bytes_scanned"] - _int8_bytes(n, d)) % (n * d * 4) == 0, st
        assert st["fallback_queries"] <= 1, st                       # the zero query
        idx.set_option("dense_int8", 0)
        d16, i16 = idx.search(qs, k)
        np.testing.assert_array_equal(i8[:-1], i16[:-1])
        np.testing.assert_array_equal(d8[:-1].view(np.uint64), d16[:-1].view(np.uint64))
        assert np.isnan(d8[-1]).all()
        for qi in range(0, nq - 1, max(1, nq // 5)):
            rd, ri = O.dense_topk(db, qs[qi], k, "cosine")
            np.testing.assert_allclose(d8[qi], rd, rtol=1e-12, atol=1e-15)
            mism = i8[qi] != ri
            if mism.any():
                full = O.dense_distances(db, qs[qi], "cosine")
                assert np.abs(full[i8[qi][mism]] - full[ri[mism]]).max() < 1e-14
            assert 777 not in i8[qi]
    idx.close()


@pytest.mark.parametrize("n,d,nq,k,metric", [(150_000, 128, 33, 50, "euclidean"), (120_000, 100, 64, 10, "euclidean"),
                                              (200_000, 128, 100, 100, "euclidean"), (100_000, 64, 128, 25, "euclidean"),
                                              (130_000, 128, 200, 100, "euclidean"), (90_000, 17, 256, 5, "euclidean"),
                                              (110_000, 128, 70, 20, "cosine"), (80_000, 96, 129, 10, "cosine")])
def test_dense_int8_batches_of_two_and_four_query_tiles(n, d, nq, k, metric):
    """33 .. 256 queries over 128-byte rows: dense8_scan_mt_kernel scores 2 (up to 64 queries) or 4 query tiles per wave
    against each int8 row unit; more than 128 queries make groups that re-read the rows.  Same answers as the bf16
    multi-tile kernels, bit for bit, and as the oracle; beyond dense_int8_batch the bf16 kernels take the call."""
    rng = np.random.default_rng(n + nq)
    db, qs = _int8_family(rng, "normal", n, d, nq)
    m = _lib.SQ_METRIC_L2 if metric == "euclidean" else _lib.SQ_METRIC_COSINE
    idx = _lib.DenseIndex(db, metric=m)
    idx.set_option("dense_int8_batch", 256)      # (default 64: the four-tile kernel only ties with the bf16 ones)
    tiles = -(-nq // 32)
    qt = 2 if tiles == 2 else 4
    groups = -(-tiles // qt)
    d8, i8 = idx.search(qs, k)
    st = idx.stats()
    assert st["bytes_scanned"] == _int8_bytes(n, d) * groups, st
    assert st["fallback_queries"] == 0 and st["mid_tier_queries"] == 0, st
    idx.set_option("dense_int8_batch", 32)
    d16, i16 = idx.search(qs, k)
    assert idx.stats()["bytes_scanned"] != st["bytes_scanned"]
    view = np.uint32 if metric == "euclidean" else np.uint64
    np.testing.assert_array_equal(i8, i16)
    np.testing.assert_array_equal(d8.view(view), d16.view(view))
    for qi in range(0, nq, max(1, nq // 7)):
        rd, ri = O.dense_topk(db, qs[qi], k, metric)
        if metric == "euclidean":
            np.testing.assert_array_equal(i8[qi], ri)
            np.testing.assert_array_equal(d8[qi].view(np.uint32), rd.view(np.uint32))
        else:
            np.testing.assert_allclose(d8[qi], rd, rtol=1e-12, atol=1e-15)
    idx.close()


@pytest.mark.parametrize("seed", [3, 12, 25, 31, 44, 58, 101, 202])
def test_dense_int8_random_cases(seed):
    """tools/int8_fuzz.py: random shape (n, d <= 512, nq <= 32, k), metric and data family (ties, sparse rows, offsets,
    outliers, tiny / huge scales): the int8 filter's answers are the bf16 filter's bit for bit and the oracle's."""
    from tools.int8_fuzz import run_case
    ok, _, desc = run_case(seed)
    assert ok, desc


def test_dense_int8_outlier_rows_and_odd_queries():
    """Rows far outside the clamp (their residual is beyond R: they carry N_row = -inf and are re-ranked for every query),
    rows with non-finite elements (N_row = +inf), and queries the filter cannot serve (all zero: no scale; one huge
    element; a NaN) -- the answers are the oracle's in every case; the odd queries take the later tiers."""
    rng = np.random.default_rng(2024)
    n, d, k = 120_000, 128, 20
    db = rng.standard_normal((n, d)).astype(np.float32)
    out = rng.choice(n, size=40, replace=False)
    db[out[:20]] *= np.float32(50.0)                              # whole rows 50x the rest
    db[out[20:30], 5] = np.float32(400.0)                         # one wild element
    db[out[30:35], 7] = np.nan
    db[out[35:], 9] = np.inf
    qs = rng.standard_normal((12, d)).astype(np.float32)
    qs[0] = db[out[0]] * np.float32(1.001)                        # nearest neighbour is an outlier row
    qs[1] = db[out[21]]
    qs[1, 5] = 399.0
    qs[2] = 0.0
    qs[3, 17] = 1.0e6
    qs[4, 3] = np.nan
    with np.errstate(invalid="ignore", over="ignore"):
        idx = _lib.DenseIndex(db)
        idx.set_option("dense_int8", 1)
        idx.search(qs[5:], k)
        st = idx.stats()
        assert st["bytes_scanned"] == _int8_bytes(n) and st["mid_tier_queries"] + st["fallback_queries"] == 0, st
        dd, ii = idx.search(qs, k)
        st = idx.stats()
        assert 2 <= st["mid_tier_queries"] + st["fallback_queries"] <= 5, st   # the zero and the NaN query at least
        for qi in range(len(qs)):
            if qi == 4:
                assert not np.isfinite(dd[qi]).any()
                continue
            rd, ri = O.dense_topk(db, qs[qi], k)
            np.testing.assert_array_equal(ii[qi], ri)
            np.testing.assert_array_equal(dd[qi].view(np.uint32), rd.view(np.uint32))
        assert ii[0, 0] == out[0] and ii[1, 0] == out[21]
    idx.close()


def test_dense_int8_heavy_tails_and_tight_clusters():
    """Data the measured bound does not suit.  Heavy tails (Student t, 2 degrees of freedom: a few huge elements set the
    rms and the bulk of the rows falls into a few int8 steps): the build declines and the matrix keeps the bf16 filter.
    Two tight clusters (every row of the query's cluster inside the int8 slack): the filter is built, its lists overflow
    call after call, the queries are served by the later tiers, and after three such calls the handle goes back to the
    bf16 filter by itself.  Results are the oracle's throughout."""
    rng = np.random.default_rng(99)
    n, d, k = 160_000, 128, 10
    db = rng.standard_t(2.0, size=(n, d)).astype(np.float32)
    qs = rng.standard_t(2.0, size=(6, d)).astype(np.float32)
    idx = _dense_check(db, qs, k)
    assert idx.stats()["bytes_scanned"] == _bf16_bytes(n, d), idx.stats()
    idx.close()
    u = rng.standard_normal(d).astype(np.float32)
    u *= np.float32(10.0) / np.linalg.norm(u)
    sign = np.where(np.arange(n) % 2 == 0, 1.0, -1.0).astype(np.float32)[:, None]
    db = (sign * u + np.float32(0.02) * rng.standard_normal((n, d))).astype(np.float32)
    qs = (u + np.float32(0.02) * rng.standard_normal((8, d))).astype(np.float32)
    idx = _lib.DenseIndex(db)
    seen = []
    for rep in range(5):
        dd, ii = idx.search(qs, k)
        seen.append(idx.stats()["bytes_scanned"])
        for qi in (0, 5):
            rd, ri = O.dense_topk(db, qs[qi], k)
            np.testing.assert_array_equal(ii[qi], ri)
            np.testing.assert_array_equal(dd[qi].view(np.uint32), rd.view(np.uint32))
    # (the later tiers add the float32 rows they read: the same in every call here)
    later = seen[0] - _int8_bytes(n)
    assert later > 0 and later % (n * d * 4) == 0 and seen[-1] == _bf16_bytes(n, d) + later, seen
    idx.close()


@pytest.mark.parametrize("metric", ["euclidean", "cosine"])
def test_dense_int8_copy_follows_appends(metric):
    """sq_dense_append keeps the int8 copy: new rows are quantised with the build's step, rows it does not suit (here: a
    block of rows 100x the rest) become always-candidates, an index that has doubled since the clamp was chosen chooses
    again, and an index born too small gets its copy once it has grown.  After every append the answers are the oracle's
    over all rows and the pass streams the int8 copy of the current size."""
    rng = np.random.default_rng(5)
    d, k = 96, 10
    m = _lib.SQ_METRIC_L2 if metric == "euclidean" else _lib.SQ_METRIC_COSINE
    total = 290_000
    db = rng.standard_normal((total, d)).astype(np.float32)
    db[104_000:104_040] *= np.float32(100.0)                  # appended outliers
    qs = rng.standard_normal((9, d)).astype(np.float32)
    qs[0] = db[104_010] * np.float32(1.0001)

    def check(idx, n):
        dd, ii = idx.search(qs, k)
        st = idx.stats()
        for qi in range(len(qs)):
            rd, ri = O.dense_topk(db[:n], qs[qi], k, metric)
            if metric == "euclidean":
                np.testing.assert_array_equal(ii[qi], ri)
                np.testing.assert_array_equal(dd[qi].view(np.uint32), rd.view(np.uint32))
            else:
                np.testing.assert_allclose(dd[qi], rd, rtol=1e-12, atol=1e-15)
        return st

    idx = _lib.DenseIndex(np.ascontiguousarray(db[:100_000]), metric=m)
    assert check(idx, 100_000)["bytes_scanned"] == _int8_bytes(100_000, d)
    n = 100_000
    for add in (3_001, 1_039, 30_000, 80_000, 75_960):       # 104 040: the outliers are in; 214 040 > 2 x 100 000: a new clamp
        idx.append(np.ascontiguousarray(db[n:n + add]))
        n += add
        st = check(idx, n)
        assert st["bytes_scanned"] == _int8_bytes(n, d), (n, st)
        assert st["fallback_queries"] == 0, (n, st)
    idx.close()
    # born below the int8 filter's minimum size
    idx = _lib.DenseIndex(np.ascontiguousarray(db[:70_000 - 10_000]), metric=m)
    n = 60_000
    idx.append(np.ascontiguousarray(db[n:n + 10_000]))
    n += 10_000
    assert check(idx, n)["bytes_scanned"] == _int8_bytes(n, d)
    idx.close()


@pytest.mark.parametrize("d,family", [(128, "long_query"), (64, "two_clusters"), (256, "long_query"), (512, "mixed"), (128, "mixed"),
                                      (100, "two_clusters"), (300, "mixed"), (36, "long_query")])   # (rows with a partial last unit)
def test_dense_middle_tier_certifies_what_the_bf16_filter_cannot(d, family):
    """Queries the bfloat16 filter cannot certify -- a query hundreds of times longer than the rows, two tight clusters
    far from their common centre (every row of the query's cluster is inside the bf16 slack) -- take the middle tier
    (sq_dense_mid.hpp: float32 rows split into two bf16 planes on the fly, 64x less slack) instead of the exact
    all-rows path; answers stay bit-identical to the oracle either way, and without the tier the same queries fall
    through to the exact path."""
    rng = np.random.default_rng(500 + d)
    n, k = 200_000, 25
    db = rng.standard_normal((n, d)).astype(np.float32)
    qs = rng.standard_normal((12, d)).astype(np.float32)
    if family in ("long_query", "mixed"):
        qs[:6] *= np.float32(300.0)                       # |q| >> |x|: beta |q|^2 swamps the spread of the scores
    if family in ("two_clusters", "mixed"):
        off = np.zeros(d, dtype=np.float32)
        off[0] = 100.0                                     # clusters at +-100 e_0, spread 0.5: |x - c| = 100 for every row
        db = (0.5 * db + np.where(np.arange(n)[:, None] % 2 == 0, off, -off)).astype(np.float32)
        qs[6:] = (0.5 * qs[6:] + off).astype(np.float32)
    idx = _dense_check(db, qs, k, options={"dense_int8": 0})   # (the tiers behind the bf16 filter are the subject: this index
    st = idx.stats()                                           #  keeps no int8 copy -- its own choice, nothing process-wide)
    assert idx.info()["int8_copy_bytes"] == 0
    assert st["mid_tier_queries"] > 0, st
    assert st["fallback_queries"] <= st["mid_tier_queries"] // 3, st           # the tier certifies most of what it takes
    mid = st["mid_tier_queries"]
    idx.set_option("dense_mid_tier", 0)
    d0, i0 = idx.search(qs, k)
    assert idx.stats()["fallback_queries"] >= mid and idx.stats()["mid_tier_queries"] == 0
    idx.set_option("dense_mid_tier", 1)
    d1, i1 = idx.search(qs, k)
    np.testing.assert_array_equal(i0, i1)
    np.testing.assert_array_equal(d0.view(np.uint32), d1.view(np.uint32))
    idx.close()


@pytest.mark.parametrize("d,offset,int8", [(64, 50.0, -1), (128, 50.0, -1), (256, 8.0, 0), (512, 50.0, -1), (128, 20.0, 0),
                                           (100, 50.0, -1), (300, 50.0, 0), (36, 20.0, -1), (129, 50.0, -1)])   # (partial last units; 129: a padded copy)
def test_dense_cosine_middle_tier_offset_data(d, offset, int8):
    """Cosine over descriptors that share a large offset (tools/int8_fuzz.py's "cosine offset" family: every similarity
    within 1e-3 of 1, all rows inside the first filters' slack -- the round-3 review found every such query on the exact
    all-rows path).  The cosine middle tier (sq_dense_mid.hpp: unit rows split into two bf16 planes on the fly, scores
    taken about the column means so the error scales with |q - c|/|q|) certifies them; answers match the oracle's
    restatement of metrics.cosine_distance (smqtk_indexing/utils/metrics.py:120-137) with and without the tier.
    Also in the mix: zero rows and a row holding inf (NaN distance: last), a zero query, a query about the origin."""
    rng = np.random.default_rng(700 + d + int(offset))
    n, k = 200_000, 25
    off = (offset * rng.standard_normal(d)).astype(np.float32)
    db = (rng.standard_normal((n, d)).astype(np.float32) + off).astype(np.float32)
    db[5] = 0.0
    db[70_001] = 0.0
    db[123_456, 3] = np.inf
    qs = (rng.standard_normal((12, d)).astype(np.float32) + off).astype(np.float32)
    qs[3] = db[777]                                        # a stored row: distance 0 (or an ulp) first
    qs[10] = rng.standard_normal(d).astype(np.float32)      # about the origin, far from every row's direction
    qs[11] = 0.0                                           # every distance NaN: rows in id order
    options = {"dense_int8": int8} if int8 >= 0 else None
    idx = _dense_check(db, qs, k, "cosine", options=options)
    st = idx.stats()
    assert st["mid_tier_queries"] >= 9, st
    assert st["fallback_queries"] <= 3, st                  # (the zero query has nothing to certify; at most two more)
    mid = st["mid_tier_queries"]
    d1, i1 = idx.search(qs, k)
    # candidate lists that overflow call after call suspend the first filters: calls then start at the middle tier (one pass
    # over the float32 rows is all that is streamed), with the same answers
    for _ in range(8):
        d3, i3 = idx.search(qs, k)
    st3 = idx.stats()
    # (no first-filter candidates; the tier's one pass, plus one exact pass for what it could not certify: the zero query)
    assert st3["candidates"] == 0 and st3["bytes_scanned"] <= 2 * n * d * 4 and st3["mid_tier_queries"] == 12, st3
    np.testing.assert_array_equal(i3, i1)
    np.testing.assert_array_equal(d3.view(np.uint64), d1.view(np.uint64))
    idx.set_option("dense_mid_tier", 0)
    d0, i0 = idx.search(qs, k)
    # (without the tier the first filters run again -- bf16 by now, which may certify the query about the origin -- and
    # everything they cannot certify is on the exact path)
    assert idx.stats()["fallback_queries"] >= mid - 2 and idx.stats()["mid_tier_queries"] == 0
    np.testing.assert_array_equal(i0, i1)
    np.testing.assert_array_equal(d0.view(np.uint64), d1.view(np.uint64))
    # rows appended later are covered too (the tier's per-row terms are rebuilt for the grown index)
    idx.set_option("dense_mid_tier", 1)
    extra = (rng.standard_normal((5_000, d)).astype(np.float32) + off).astype(np.float32)
    extra[17] = qs[0] * np.float32(3.0)                     # the same direction as query 0: distance ~0, must come first
    idx.append(extra)
    d2, i2 = idx.search(qs[:4], k)
    assert idx.stats()["mid_tier_queries"] > 0
    both = np.concatenate([db, extra])
    for qi in range(4):
        rd, ri = O.dense_topk(both, qs[qi], k, "cosine")
        np.testing.assert_allclose(d2[qi], rd, rtol=1e-12, atol=1e-15)
        mism = i2[qi] != ri
        if mism.any():
            full = O.dense_distances(both, qs[qi], "cosine")
            assert np.abs(full[i2[qi][mism]] - full[ri[mism]]).max() < 1e-14
    assert i2[0, 0] == n + 17 or d2[0, 0] == d2[0, 1]
    idx.close()


def test_dense_suspended_first_filter_is_rearmed():
    """Three calls whose candidate lists all overflow (zero queries under cosine: every row scores the same) suspend the first
    filter; ordinary queries then start at the middle tier, the sixteenth such call probes the first filter again, finds
    it working and re-arms it.  Answers match the oracle in every state."""
    rng = np.random.default_rng(909)
    n, d, k = 100_000, 64, 10
    db = rng.standard_normal((n, d)).astype(np.float32)
    good = rng.standard_normal((8, d)).astype(np.float32)
    zero = np.zeros((8, d), dtype=np.float32)
    idx = _lib.DenseIndex(db, metric=_lib.SQ_METRIC_COSINE, options={"dense_int8": 0})
    want = [O.dense_topk(db, q, k, "cosine") for q in good]

    def check():
        dist, ids = idx.search(good, k)
        for qi in range(len(good)):
            np.testing.assert_allclose(dist[qi], want[qi][0], rtol=1e-12, atol=1e-15)
            np.testing.assert_array_equal(ids[qi], want[qi][1])
        return idx.stats()

    st = check()
    assert st["candidates"] > 0 and st["mid_tier_queries"] == 0, st          # the first filter answers
    for _ in range(3):
        idx.search(zero, k)
        assert idx.stats()["candidates"] == 8 * n
    for i in range(15):
        st = check()
        assert st["candidates"] == 0 and st["mid_tier_queries"] == 8 and st["fallback_queries"] == 0, (i, st)
    st = check()                                                              # the probe
    assert st["candidates"] > 0 and st["mid_tier_queries"] == 0, st
    st = check()                                                              # re-armed
    assert st["candidates"] > 0 and st["mid_tier_queries"] == 0, st
    idx.close()


@pytest.mark.parametrize("metric", ["euclidean", "cosine"])
def test_dense_exact_path_two_level_select(metric):
    """The exact path (forced here; also what rows wider than the scan covers take) selects in two
    levels -- sampled exact threshold, device-wide compaction, select over the survivors -- and goes
    on to the select over all keys when ties overflow the compacted list or k is too large for it."""
    rng = np.random.default_rng(21)
    _lib.set_option("force_fallback", 1)
    db = rng.standard_normal((200_000, 64)).astype(np.float32)
    db[1000:1400] = db[7]                                   # a 400-row tie group inside the top-k
    qs = rng.standard_normal((3, 64)).astype(np.float32)
    qs[0] = db[7]
    for k in (1, 100, 3000, 16_000 if metric == "euclidean" else 7000):   # 16 000: no room for a sample stride, full select
        idx = _dense_check(db, qs, k, metric)
        assert idx.stats()["fallback_queries"] == 3
    # every row the same: the compacted list overflows, the full select answers in id order
    same = np.repeat(rng.standard_normal((1, 32)).astype(np.float32), 150_000, axis=0)
    _dense_check(same, qs[:2, :32].copy(), 10, metric)
    if metric == "cosine":                                  # zero rows (NaN distance) among the sampled rows
        z = rng.standard_normal((100_000, 16)).astype(np.float32)
        z[::3] = 0.0
        idx = _lib.DenseIndex(z, metric=_lib.SQ_METRIC_COSINE)
        d, i = idx.search(z[1:2], 50)
        rd, ri = O.dense_topk(z, z[1], 50, "cosine")
        np.testing.assert_allclose(d[0], rd, rtol=1e-12, atol=1e-15)
    _lib.set_option("force_fallback", 0)
    wide = rng.standard_normal((70_000, 700)).astype(np.float32)   # wider than the ring kernels: dense_wide_scan_kernel (round 4)
    idx = _dense_check(wide, wide[:2] + 0.01, 20, metric)
    assert idx.stats()["fallback_queries"] == 0
    _lib.set_option("force_fallback", 1)                            # ... and its exact path (the group kernel at d = 700)
    try:
        idx = _dense_check(wide, wide[:2] + 0.01, 20, metric)
        assert idx.stats()["fallback_queries"] == 2
    finally:
        _lib.set_option("force_fallback", 0)


@pytest.mark.parametrize("metric", ["euclidean", "cosine"])
def test_dense_append_equals_a_fresh_index(metric):
    """sq_dense_append: rows added behind a resident matrix (partial last tile, several appends, one
    that forces the buffers to grow) answer exactly like an index built from the concatenation."""
    rng = np.random.default_rng(41)
    m = _lib.SQ_METRIC_L2 if metric == "euclidean" else _lib.SQ_METRIC_COSINE
    d = 96
    parts = [rng.standard_normal((n, d)).astype(np.float32) + (0.5 if i else 0.0) for i, n in enumerate((70_003, 1_000, 37, 120_000))]
    parts[1][5] = parts[0][9]                                        # a tie between an old and an appended row
    qs = rng.standard_normal((5, d)).astype(np.float32)
    qs[0] = parts[0][9]
    grown = _lib.DenseIndex(parts[0], metric=m)
    for upto in range(1, len(parts)):
        grown.append(parts[upto])
        whole = np.vstack(parts[: upto + 1])
        assert grown.n == len(whole)
        fresh = _lib.DenseIndex(whole, metric=m)
        for k in (1, 40):
            gd, gi = grown.search(qs, k)
            fd, fi = fresh.search(qs, k)
            np.testing.assert_array_equal(gi, fi)
            np.testing.assert_array_equal(gd.view(np.uint64 if metric == "cosine" else np.uint32),
                                          fd.view(np.uint64 if metric == "cosine" else np.uint32))
        rd, ri = O.dense_topk(whole, qs[0], 40, metric)
        np.testing.assert_array_equal(gi[0], ri)
        fresh.close()
    grown.close()
    # an index over a borrowed device matrix cannot grow in place
    import torch
    t = torch.from_numpy(parts[0][:4096]).cuda()
    b = _lib.DenseIndex(t.data_ptr(), n=4096, d=d, metric=m, device_ptr=True, keepalive=t)
    with pytest.raises(_lib.HipError):
        b.append(parts[1])
    b.close()


@pytest.mark.parametrize("tag", list(GI.DENSE_CASES))
@pytest.mark.parametrize("metric", ["euclidean", "cosine"])
def test_dense_golden_cases(golden, tag, metric):
    """Small matrices (every row is a candidate): exact-distance kernel + select."""
    g = golden("g5_dense_nn.npz")
    n, d, nq, seed, dist, dt = GI.DENSE_CASES[tag]
    db, qs = GI.dense_inputs(n, d, nq, seed, dist, dt)
    k = min(GI.DENSE_KMAX, n)
    idx = _dense_check(db, qs, k, metric)
    dd, ii = idx.search(qs, k)
    ridx, rdist = g[f"{tag}_{metric}_idx"], g[f"{tag}_{metric}_dist"]
    for qi in range(ridx.shape[0]):
        if metric == "euclidean":
            np.testing.assert_array_equal(dd[qi], rdist[qi][:k])   # the real reference's numbers
            np.testing.assert_array_equal(ii[qi], ridx[qi][:k])
        else:
            np.testing.assert_allclose(dd[qi], rdist[qi][:k], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("tag", ["uni128", "nrm128", "nrm200"])
@pytest.mark.parametrize("metric", ["euclidean", "cosine"])
def test_dense_scan_path_small(tag, metric):
    """Force the MFMA scan path on the golden inputs (tiny candidate cap)."""
    n, d, nq, seed, dist, dt = GI.DENSE_CASES[tag]
    db, qs = GI.dense_inputs(n, d, nq, seed, dist, dt)
    _lib.set_option("candidate_cap", 1024)
    idx = _dense_check(db, qs, 20, metric)
    st = idx.stats()
    assert st["scan_launches"] >= 2


@pytest.mark.parametrize("n,d,nq,k", [(300_000, 128, 33, 100), (100_000, 512, 5, 10), (120_000, 64, 64, 1),
                                      (90_001, 100, 3, 50), (200_000, 256, 2, 100)])
@pytest.mark.parametrize("stages", [0, 2])
def test_dense_scan_path_l2(n, d, nq, k, stages):
    rng = np.random.default_rng(n + d)
    db = rng.standard_normal((n, d)).astype(np.float32)
    qs = rng.standard_normal((nq, d)).astype(np.float32)
    qs[0] = db[12345]
    db[777] = db[555]
    _lib.set_option("dense_stages", stages)
    idx = _dense_check(db, qs, k, "euclidean")
    st = idx.stats()
    assert st["fallback_queries"] == 0, st
    assert st["candidates"] >= nq * k


@pytest.mark.parametrize("qplanes", [0, 2])
@pytest.mark.parametrize("metric,nq,qt", [("euclidean", 130, 0), ("euclidean", 97, 2), ("euclidean", 65, 1),
                                          ("cosine", 70, 0), ("cosine", 33, 4)])
def test_dense_scan_query_tiles_per_wave(metric, nq, qt, qplanes):
    """Batches of several 32-query tiles: 2 or 4 query tiles per scan wave (padded
    groups, survivors of several tiles in one segment, group-wide re-rank)."""
    rng = np.random.default_rng(1000 + nq)
    db = rng.standard_normal((150_000, 128)).astype(np.float32)
    qs = rng.standard_normal((nq, 128)).astype(np.float32)
    qs[nq - 1] = db[4242]
    _lib.set_option("dense_qt", qt)
    _lib.set_option("dense_qplanes", qplanes)   # 0: q_hi only in the multi-tile scan; 2: q_hi + q_lo
    try:
        idx = _dense_check(db, qs, 20, metric)
    finally:
        _lib.set_option("dense_qt", 0)
        _lib.set_option("dense_qplanes", 0)
    st = idx.stats()
    assert st["fallback_queries"] == 0, st
    assert st["candidates"] >= nq * 20


@pytest.mark.parametrize("metric,d,nq", [("euclidean", 256, 70), ("cosine", 512, 40), ("euclidean", 384, 33),
                                          ("euclidean", 200, 64)])
def test_dense_wide_rows_multi_tile_batches(metric, d, nq):
    """d_pad > 128 with more than one query tile: all k-units' query fragments in AGPRs, two query
    tiles per wave (the LDS-copy kernel serves single-tile batches only)."""
    rng = np.random.default_rng(d + nq)
    db = rng.standard_normal((120_000, d)).astype(np.float32)
    qs = rng.standard_normal((nq, d)).astype(np.float32)
    qs[1] = db[777]
    idx = _dense_check(db, qs, 15, metric)
    st = idx.stats()
    assert st["fallback_queries"] == 0, st
    assert st["candidates"] >= nq * 15


def test_dense_offset_data_is_filtered_around_its_mean():
    """Rows far from the origin (|x| >> spread): the L2 filter scores x - c against q - c (c = column
    means), so its error bound does not swallow the distance differences; no query may need the exact
    path and the candidate lists stay short.  With the centre switched off the same data overflows."""
    rng = np.random.default_rng(77)
    db = (20.0 + 0.5 * rng.standard_normal((200_000, 128))).astype(np.float32)
    qs = (20.0 + 0.5 * rng.standard_normal((5, 128))).astype(np.float32)
    qs[0] = db[31337]
    idx = _dense_check(db, qs, 50, "euclidean")
    st = idx.stats()
    assert st["fallback_queries"] == 0, st
    assert st["candidates"] < 5 * 20_000, st
    _lib.set_option("dense_no_center", 1)
    try:
        idx2 = _dense_check(db, qs, 50, "euclidean")            # still exact: the bf16 filter certifies none of them
        assert idx2.stats()["mid_tier_queries"] == 5            # (the middle tier or, behind it, the exact path answers)
    finally:
        _lib.set_option("dense_no_center", 0)


@pytest.mark.parametrize("metric", ["euclidean", "cosine"])
def test_dense_clustered_rows_in_cluster_order(metric):
    """A mixture of tight clusters stored cluster after cluster (the order an ingest pipeline
    produces), queries inside clusters, k larger than a cluster: the sampled threshold and the
    per-wave survivor segments must cope without the exact path."""
    rng = np.random.default_rng(404)
    centers = rng.standard_normal((400, 128)).astype(np.float32) * 4.0
    sizes = rng.integers(50, 1500, size=400)
    db = np.concatenate([c + 0.3 * rng.standard_normal((m, 128)).astype(np.float32) for c, m in zip(centers, sizes)])
    qs = np.stack([db[10], db[len(db) // 2] + 0.01, centers[7], centers[399] + 0.2,
                   rng.standard_normal(128).astype(np.float32)]).astype(np.float32)
    idx = _dense_check(db, qs, 300, metric)
    st = idx.stats()
    assert st["fallback_queries"] == 0, st


def test_dense_large_k_scan_path():
    rng = np.random.default_rng(9)
    db = rng.standard_normal((250_000, 64)).astype(np.float32)
    qs = rng.standard_normal((3, 64)).astype(np.float32)
    idx = _dense_check(db, qs, 3000, "euclidean")
    assert idx.stats()["fallback_queries"] == 0


def test_dense_scan_path_cosine():
    rng = np.random.default_rng(99)
    db = rng.random((150_000, 128)).astype(np.float32)
    qs = rng.random((6, 128)).astype(np.float32)
    idx = _dense_check(db, qs, 50, "cosine")
    assert idx.stats()["fallback_queries"] == 0


def test_dense_uncertifiable_data_takes_exact_path():
    """Many near-duplicate rows: the candidate list overflows / certification
    fails and the exact full-keys path answers (canonical tie order)."""
    rng = np.random.default_rng(3)
    base = rng.standard_normal((1, 64)).astype(np.float32)
    db = np.repeat(base, 100_000, axis=0)
    db[::7] += 1e-3
    qs = rng.standard_normal((2, 64)).astype(np.float32)
    _lib.set_option("candidate_cap", 4096)
    idx = _dense_check(db, qs, 10, "euclidean")
    assert idx.stats()["fallback_queries"] == 2
    _lib.set_option("candidate_cap", 0)
    _lib.set_option("force_fallback", 1)
    db = rng.standard_normal((80_000, 96)).astype(np.float32)
    idx = _dense_check(db, qs[:, :96].copy() if qs.shape[1] >= 96 else rng.standard_normal((2, 96)).astype(np.float32),
                       10, "euclidean")
    assert idx.stats()["fallback_queries"] == 2


@pytest.mark.parametrize("streams", [1, 2])
@pytest.mark.parametrize("metric", ["euclidean", "cosine"])
def test_dense_async_calls_equal_blocking_calls(metric, streams):
    """SQ_MEM_DEVICE_ASYNC: calls are enqueued back to back and finished one call later (status words read, uncertified
    queries redone); every call's answer must be the blocking call's, including calls whose queries take the exact
    path, calls of a different batch size in between, and the call still in flight at sq_dense_sync / destroy."""
    import torch
    _lib.set_option("dense_async_streams", streams)
    rng = np.random.default_rng(41)
    n, d, k = 150_000, 96, 20
    dbh = rng.standard_normal((n, d)).astype(np.float32)
    dbh[5000:5400] = dbh[17]                       # a tie group larger than k: candidate ties
    dev = torch.device("cuda", 0)
    db = torch.from_numpy(dbh).to(dev)
    m = _lib.SQ_METRIC_COSINE if metric == "cosine" else _lib.SQ_METRIC_L2
    ddt = torch.float64 if metric == "cosine" else torch.float32
    idx = _lib.DenseIndex(db.data_ptr(), n=n, d=d, metric=m, device_ptr=True, id_base=7, keepalive=db)
    stream = torch.cuda.current_stream().cuda_stream
    sizes = [5, 33, 5, 70, 1, 5, 5]
    qs = [rng.standard_normal((b, d)).astype(np.float32) for b in sizes]
    qs[2][0] = dbh[17]
    want = [idx.search(q, k) for q in qs]
    qd = [torch.from_numpy(q).to(dev) for q in qs]
    od = [torch.empty((b, k), dtype=ddt, device=dev) for b in sizes]
    oi = [torch.empty((b, k), dtype=torch.int64, device=dev) for b in sizes]
    for j, q in enumerate(qd):
        if j == 4:
            _lib.set_option("force_fallback", 1)   # calls FINISHED while this is on (3 and 4) are redone on the exact path
        idx.search_device_async(q.data_ptr(), sizes[j], k, od[j].data_ptr(), oi[j].data_ptr(), stream)
        if j == 5:
            assert idx.stats()["fallback_queries"] == 1   # call 4 (one query) was finished by this call
            _lib.set_option("force_fallback", 0)
        if j >= 1:                                 # the previous call is final now
            np.testing.assert_array_equal(oi[j - 1].cpu().numpy(), want[j - 1][1])
            np.testing.assert_array_equal(od[j - 1].cpu().numpy(), want[j - 1][0])
    idx.sync()
    np.testing.assert_array_equal(oi[-1].cpu().numpy(), want[-1][1])
    np.testing.assert_array_equal(od[-1].cpu().numpy(), want[-1][0])
    # a blocking call after asynchronous ones, and destroy with a call in flight
    d2, i2 = idx.search(qs[1], k)
    np.testing.assert_array_equal(i2, want[1][1])
    idx.search_device_async(qd[0].data_ptr(), sizes[0], k, od[0].data_ptr(), oi[0].data_ptr(), stream)
    idx.close()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(oi[0].cpu().numpy(), want[0][1])


@pytest.mark.parametrize("depth,wait", [(3, 1), (4, 1), (2, 0), (3, 0)])
def test_dense_async_depth(depth, wait):
    """Option dense_async_depth: `depth` asynchronous calls in flight; call i is final when call i + depth - 1 (or
    sq_dense_sync) returns -- one call later with dense_async_wait = 0 (the call returns right after enqueueing);
    changing the depth in mid-stream drains the pipeline first."""
    import torch
    rng = np.random.default_rng(43)
    n, d, k = 200_000, 64, 10
    dbh = rng.standard_normal((n, d)).astype(np.float32)
    dev = torch.device("cuda", 0)
    db = torch.from_numpy(dbh).to(dev)
    idx = _lib.DenseIndex(db.data_ptr(), n=n, d=d, device_ptr=True, keepalive=db)
    stream = torch.cuda.current_stream().cuda_stream
    sizes = [8, 40, 8, 1, 8, 8, 33, 8]
    qs = [rng.standard_normal((b, d)).astype(np.float32) for b in sizes]
    want = [idx.search(q, k) for q in qs]
    qd = [torch.from_numpy(q).to(dev) for q in qs]
    od = [torch.empty((b, k), dtype=torch.float32, device=dev) for b in sizes]
    oi = [torch.empty((b, k), dtype=torch.int64, device=dev) for b in sizes]
    try:
        _lib.set_option("dense_async_depth", depth)
        _lib.set_option("dense_async_wait", wait)
        for j, q in enumerate(qd):
            if j == 5:
                _lib.set_option("force_fallback", 1)     # the calls finished from here on are redone on the exact path
            idx.search_device_async(q.data_ptr(), sizes[j], k, od[j].data_ptr(), oi[j].data_ptr(), stream)
            f = j - (depth - 1) - (0 if wait else 1)
            if f >= 0:                                   # final now
                np.testing.assert_array_equal(oi[f].cpu().numpy(), want[f][1])
                np.testing.assert_array_equal(od[f].cpu().numpy(), want[f][0])
        _lib.set_option("force_fallback", 0)
        _lib.set_option("dense_async_depth", 3 if depth == 2 else 2)   # a new depth: the next call drains the pipeline first
        idx.search_device_async(qd[0].data_ptr(), sizes[0], k, od[0].data_ptr(), oi[0].data_ptr(), stream)
        for f in range(1, len(sizes)):
            np.testing.assert_array_equal(oi[f].cpu().numpy(), want[f][1])
            np.testing.assert_array_equal(od[f].cpu().numpy(), want[f][0])
        idx.sync()
        np.testing.assert_array_equal(oi[0].cpu().numpy(), want[0][1])
    finally:
        _lib.set_option("force_fallback", 0)
        _lib.set_option("dense_async_depth", 2)
        _lib.set_option("dense_async_wait", 1)
        idx.close()


@pytest.mark.parametrize("graph", [1, 0])
def test_dense_async_call_graph(graph):
    """Pipelined int8 calls of one shape run as ONE captured graph launch per call from the slot's third call on (the
    first sizes the workspace eagerly, the second captures); the caller's query / output pointers travel through a
    pinned block, so rotating buffers, a change of shape (k: re-capture) and the way back all give the blocking
    call's results.  dense_graph = 0: the same calls as eager launches."""
    import torch
    rng = np.random.default_rng(4242)
    n, d = 150_000, 128
    dbh = rng.standard_normal((n, d)).astype(np.float32)
    dev = torch.device("cuda", 0)
    db = torch.from_numpy(dbh).to(dev)
    idx = _lib.DenseIndex(db.data_ptr(), n=n, d=d, device_ptr=True, keepalive=db)
    stream = torch.cuda.current_stream().cuda_stream
    nq = 32
    qs = [rng.standard_normal((nq, d)).astype(np.float32) for _ in range(5)]
    qd = [torch.from_numpy(q).to(dev) for q in qs]
    plan = [(50, j % 5) for j in range(9)] + [(200, j % 5) for j in range(7)] + [(50, (j + 2) % 5) for j in range(8)]
    want = {}
    for k, b in set(plan):
        want[(k, b)] = idx.search(qs[b], k)
        assert idx.stats()["bytes_scanned"] == _int8_bytes(n)
    try:
        idx.set_option("dense_graph", graph)
        idx.set_option("dense_async_depth", 3)
        outs = []
        for j, (k, b) in enumerate(plan):
            od = torch.empty((nq, k), dtype=torch.float32, device=dev)
            oi = torch.empty((nq, k), dtype=torch.int64, device=dev)
            outs.append((od, oi))
            idx.search_device_async(qd[b].data_ptr(), nq, k, od.data_ptr(), oi.data_ptr(), stream)
            f = j - 2
            if f >= 0:
                np.testing.assert_array_equal(outs[f][1].cpu().numpy(), want[plan[f]][1])
                np.testing.assert_array_equal(outs[f][0].cpu().numpy(), want[plan[f]][0])
        idx.sync()
        for f in (len(plan) - 2, len(plan) - 1):
            np.testing.assert_array_equal(outs[f][1].cpu().numpy(), want[plan[f]][1])
            np.testing.assert_array_equal(outs[f][0].cpu().numpy(), want[plan[f]][0])
    finally:
        idx.close()


@pytest.mark.parametrize("metric,n,d,nq,k,qplanes", [("euclidean", 70_000, 1000, 2, 5, 0), ("euclidean", 80_000, 2048, 32, 100, 0),
                                                    ("euclidean", 66_000, 4096, 40, 10, 0), ("cosine", 70_000, 2048, 7, 10, 0),
                                                    ("euclidean", 70_003, 1500, 33, 20, 1), ("euclidean", 66_001, 8192, 3, 3, 0),
                                                    ("euclidean", 66_000, 1024, 100, 10, 0), ("cosine", 66_000, 640, 130, 5, 0)])
def test_dense_rows_wider_than_512_dimensions(metric, n, d, nq, k, qplanes):
    """Rows beyond the ring kernels' 512 padded dimensions (the reference's own examples index 2048- and 4096-d descriptors,
    docs/examples/caffe_build_index.rst:35): dense_wide_scan_kernel filters them from the bf16 copy -- fragments straight
    from global memory, one, two or four query tiles per wave by batch size -- and the exact re-rank (numpy's pairwise recursion
    at d > 128) answers; ids and float32 distance bits equal the oracle's, no query on the exact path."""
    rng = np.random.default_rng(d + nq)
    db = rng.standard_normal((n, d)).astype(np.float32)
    db += rng.standard_normal((1, d)).astype(np.float32)              # a common offset: the filter works around the column means
    qs = (db[rng.integers(0, n, nq)] + np.float32(0.5) * rng.standard_normal((nq, d))).astype(np.float32)
    qs[0] = db[4242]
    if qplanes:
        _lib.set_option("dense_qplanes", qplanes)
    try:
        idx = _dense_check(db, qs, k, metric)
    finally:
        _lib.set_option("dense_qplanes", 0)
    st = idx.stats()
    assert st["fallback_queries"] == 0, st
    d_pad = -(-d // 128) * 128
    assert st["bytes_scanned"] == (-(-n // 32) * 32) * (d_pad * 2 + (4 if metric == "euclidean" else 0)), st
    idx.close()


def test_dense_rows_beyond_the_scan_copy_take_the_exact_path():
    """d beyond every filter (padded rows above 8192 dimensions): the exact path, numpy pairwise recursion."""
    rng = np.random.default_rng(8)
    db = rng.standard_normal((66_000, 8300)).astype(np.float32)
    qs = rng.standard_normal((2, 8300)).astype(np.float32)
    idx = _dense_check(db, qs, 5, "euclidean")
    assert idx.stats()["fallback_queries"] == 2


@pytest.mark.parametrize("d", [1, 5, 8, 9, 127, 128, 129, 200, 300, 1024, 4100])
def test_dense_distances_bit_exact(d):
    rng = np.random.default_rng(d)
    rows = (rng.standard_normal((257, d)) * 3).astype(np.float32)
    q = rng.standard_normal(d).astype(np.float32)
    got = _lib.dense_distances(q, rows, _lib.SQ_METRIC_L2)
    ref = O.euclidean_distance(rows, q)
    np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))
    got = _lib.dense_distances(q, rows, _lib.SQ_METRIC_COSINE)
    ref = np.array([O.cosine_distance(q, r) for r in rows])
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-15)
    # float64 descriptors (SMQTK's default dtype): float64 arithmetic, bit identical
    rows64, q64 = rng.standard_normal((65, d)), rng.standard_normal(d)
    got = _lib.dense_distances(q64, rows64, _lib.SQ_METRIC_L2)
    assert got.dtype == np.float64
    np.testing.assert_array_equal(got.view(np.uint64), O.euclidean_distance(rows64, q64).view(np.uint64))
    got = _lib.dense_distances(q64, rows64, _lib.SQ_METRIC_COSINE)
    ref = np.array([O.cosine_distance(q64, r) for r in rows64])
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-15)


def test_dense_known_answers():
    # tests/impls/nn_index/test_faiss.py:443-515 / test_lsh.py:837-979 ported to the dense index
    dim = 5
    db = np.eye(dim, dtype=np.float32)
    idx = _lib.DenseIndex(db)
    d, i = idx.search(np.zeros((1, dim), np.float32), dim)
    np.testing.assert_array_equal(d[0], np.ones(dim, np.float32))
    np.testing.assert_array_equal(i[0], np.arange(dim))
    d, i = idx.search(db[3:4], 1)
    assert i[0, 0] == 3 and d[0, 0] == 0.0
    pts = np.array([[j, 2 * j] for j in range(1000)], dtype=np.float32)
    perm = np.random.default_rng(0).permutation(1000)
    idx = _lib.DenseIndex(pts[perm])
    d, i = idx.search(np.zeros((1, 2), np.float32), 1000)
    np.testing.assert_array_equal(perm[i[0]], np.arange(1000))
    assert (np.diff(d[0]) > 0).all()


# ----------------------------------------------------------------------- ITQ
@pytest.mark.parametrize("tag", list(GI.ITQ_CASES))
def test_itq_hash_golden(golden, tag):
    g = golden("g3_itq_hash.npz")
    n, d, bits, seed = GI.ITQ_CASES[tag]
    x32, mean, rot = GI.itq_inputs(n, d, bits, seed)
    for norm, ordv in ((None, _lib.SQ_NORM_NONE), (2, _lib.SQ_NORM_L2)):
        for dt in (np.float32, np.float64):
            key = f"{tag}_n{norm}_{np.dtype(dt).name}"
            got = _lib.itq_hash(x32.astype(dt), mean, rot, ordv)
            ref = g[key + "_packed"]
            bad = (got != ref).any(axis=1)
            # the reference's own sign is BLAS-order dependent when |z| is at rounding level
            if bad.any():
                assert g[key + "_minabsz"][bad].max() < 1e-10, (key, int(bad.sum()))
            assert bad.sum() <= 1


def test_itq_known_answers(golden):
    g = golden("g3_itq_hash.npz")
    mean = np.array([0., 0.])
    rot = np.array([[1. / np.sqrt(2)], [1. / np.sqrt(2)]])
    got = _lib.itq_hash(g["kat_x"], mean, rot)
    np.testing.assert_array_equal(O.unpack_bits_msb(got, 1)[:, 0], g["kat_bits"][:, 0])


@pytest.mark.parametrize("n,d,bits", [(50_000, 128, 64), (10_000, 512, 256), (3_000, 300, 100), (1000, 40, 33)])
def test_itq_hash_bulk(n, d, bits):
    rng = np.random.default_rng(n + bits)
    x = rng.standard_normal((n, d)).astype(np.float32)
    mean = x[:1000].mean(axis=0).astype(np.float64)
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    rot = np.ascontiguousarray(q[:, :bits])
    for norm, ordv in ((None, _lib.SQ_NORM_NONE), (2, _lib.SQ_NORM_L2)):
        got = _lib.itq_hash(x, mean, rot, ordv)
        z = O.itq_z(x, mean, rot, norm)
        ref = O.pack_bits_msb(z >= 0)
        bad = (got != ref).any(axis=1)
        if bad.any():
            assert np.abs(z[bad]).min(axis=1).max() < 1e-10
        assert bad.mean() < 1e-3


@pytest.mark.parametrize("n,d,bits", [(100_003, 128, 64), (40_000, 64, 33), (30_001, 256, 128), (20_000, 192, 100),
                                      (33, 128, 64)])
def test_itq_filter_matches_float64_kernel(n, d, bits):
    """The bf16x3 filter + float64 recompute of the undecided rows returns exactly the codes of
    the all-float64 kernel (option itq_exact), and both agree with the oracle."""
    rng = np.random.default_rng(n + d + bits)
    x = (rng.standard_normal((n, d)) * rng.uniform(0.1, 30.0, (n, 1))).astype(np.float32)
    mean = x[:2000].mean(axis=0).astype(np.float64)
    x[5] = mean.astype(np.float32)          # z ~ 0 in every bit: the whole row is undecided
    x[7] = 0.0                              # zero row (norm 0 -> 1 with normalize=2)
    x[n - 1] = x[0]
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    rot = np.ascontiguousarray(q[:, :bits])
    rot[:, 3] = 0.0                         # a degenerate hash bit: z == -mean.R == 0 -> True everywhere
    # a float32 model mean (what ItqFunctor.fit leaves for float32 descriptors): numpy subtracts it in float32
    for mean_m in (mean, mean.astype(np.float32)):
        for norm, ordv in ((None, _lib.SQ_NORM_NONE), (2, _lib.SQ_NORM_L2)):
            got = _lib.itq_hash(x, mean_m, rot, ordv)
            _lib.set_option("itq_exact", 1)
            try:
                exact = _lib.itq_hash(x, mean_m, rot, ordv)
            finally:
                _lib.set_option("itq_exact", 0)
            np.testing.assert_array_equal(got, exact)
            z = O.itq_z(x, mean_m, rot, norm)
            ref = O.pack_bits_msb(z >= 0)
            bad = (got != ref).any(axis=1)
            if bad.any():
                assert np.abs(z[bad]).min(axis=1).max() < 1e-9
            assert bad.mean() < 1e-2


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("n,d,bits", [(60_001, 512, 256), (50_000, 512, 64), (40_003, 384, 200), (70_000, 128, 256),
                                      (30_000, 320, 130), (90_001, 128, 64), (25_000, 64, 33), (33, 256, 128)])
def test_itq_wide_filter_matches_float64_kernel(n, d, bits, dt):
    """The wide filter (sq_itq_wide.hpp: rows resident as float16 fragments, R streamed through LDS; d <= 512, <= 256
    bits, float32 AND float64 rows) returns exactly the codes of the all-float64 kernel, and both agree with the oracle:
    BASELINE config 4's 512-d descriptors, SMQTK's default float64 descriptors, k-blocks that end inside a 256-k block,
    an all-undecided row, a zero row, a degenerate hash bit, both normalisations, both model dtypes."""
    if dt == np.float32 and d <= 256 and bits <= 128:
        pytest.skip("the narrow filter's shape (test_itq_filter_matches_float64_kernel)")
    rng = np.random.default_rng(n + d + bits)
    x = (rng.standard_normal((n, d)) * rng.uniform(0.1, 30.0, (n, 1))).astype(dt)
    mean = x[:2000].mean(axis=0).astype(np.float64)
    x[5] = mean.astype(dt)                  # z ~ 0 in every bit: the whole row is undecided
    x[7] = 0.0                              # zero row (norm 0 -> 1 with normalize=2)
    x[n - 1] = x[0]
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    rot = np.ascontiguousarray(q[:, :bits])
    rot[:, 3] = 0.0                         # a degenerate hash bit: z == -mean.R == 0 -> True everywhere
    for mean_m in (mean, mean.astype(np.float32)):
        for norm, ordv in ((None, _lib.SQ_NORM_NONE), (2, _lib.SQ_NORM_L2)):
            got = _lib.itq_hash(x, mean_m, rot, ordv)
            _lib.set_option("itq_exact", 1)
            try:
                exact = _lib.itq_hash(x, mean_m, rot, ordv)
            finally:
                _lib.set_option("itq_exact", 0)
            np.testing.assert_array_equal(got, exact)
            z = O.itq_z(x, mean_m, rot, norm)
            ref = O.pack_bits_msb(z >= 0)
            bad = (got != ref).any(axis=1)
            if bad.any():
                assert np.abs(z[bad]).min(axis=1).max() < 1e-9
            assert bad.mean() < 1e-2


@pytest.mark.parametrize("d", [2, 64])
def test_itq_mean_dtype_follows_numpy_promotion(d):
    """`x - mean` is float32 arithmetic when both are float32 (a model fitted on float32
    descriptors) and float64 otherwise; the rounding of the float32 difference can decide a bit."""
    x = np.zeros((40, d), dtype=np.float32)
    x[:, 0], x[:, 1] = 1e8, 1.0
    mean64 = np.zeros(d)
    mean64[0] = 3.0
    rot = np.zeros((d, 1))
    rot[0, 0], rot[1, 0] = 1.0, -99999998.5
    # float64: 99999997 - 99999998.5 < 0; float32: fl32(99999997) = 1e8 -> 1e8 - 99999998.5 > 0
    for mean, want in ((mean64, False), (mean64.astype(np.float32), True)):
        z = O.itq_z(x, mean, rot)
        assert ((z >= 0) == want).all()
        got = _lib.itq_hash(x, mean, rot)
        np.testing.assert_array_equal(got, O.pack_bits_msb(z >= 0))
    got = _lib.itq_hash(x.astype(np.float64), mean64.astype(np.float32), rot)   # float64 rows: float64 arithmetic
    np.testing.assert_array_equal(got, O.pack_bits_msb(O.itq_z(x.astype(np.float64), mean64.astype(np.float32), rot) >= 0))


def test_dense_batches_beyond_the_query_chunk():
    """More queries than one internal chunk (4096): results and statistics of the chunks are stitched."""
    rng = np.random.default_rng(4)
    db = rng.standard_normal((70_000, 64)).astype(np.float32)
    qs = rng.standard_normal((4500, 64)).astype(np.float32)
    idx = _lib.DenseIndex(db)
    d, i = idx.search(qs, 5)
    for qi in (0, 4095, 4096, 4499):
        rd, ri = O.dense_topk(db, qs[qi], 5, "euclidean")
        np.testing.assert_array_equal(i[qi], ri)
        np.testing.assert_array_equal(d[qi].view(np.uint32), rd.view(np.uint32))
    assert idx.stats()["candidates"] >= 4500 * 5


@pytest.mark.parametrize("n", [300, 90_000])
def test_cosine_zero_vectors_give_nan_ranked_last(n):
    """scipy's 0/0 for a zero vector is NaN, numpy's clips keep it and the reference's stable sort puts
    it last; the device must not turn it into distance 0."""
    rng = np.random.default_rng(n)
    db = rng.random((n, 96)).astype(np.float32)
    db[[17, 201]] = 0.0
    qs = rng.random((3, 96)).astype(np.float32)
    qs[2] = 0.0                                                     # a zero query: every distance is NaN
    k = n if n < 1000 else 50
    idx = _lib.DenseIndex(db, metric=_lib.SQ_METRIC_COSINE)
    d, i = idx.search(qs, k)
    for qi in range(3):
        rd, ri = O.dense_topk(db, qs[qi], k, "cosine")
        np.testing.assert_allclose(d[qi], rd, rtol=1e-12, atol=1e-15, equal_nan=True)
        solid = ~np.isnan(rd)
        np.testing.assert_array_equal(i[qi][solid], ri[solid])
        assert np.isnan(d[qi]).sum() == np.isnan(rd).sum()
    if n < 1000:
        assert set(i[0][-2:]) == {17, 201} and np.isnan(d[0][-2:]).all()


@pytest.mark.parametrize("norm", [None, 2])
@pytest.mark.parametrize("mean_dt", [np.float32, np.float64])
def test_itq_resident_model_equals_one_shot_hash(norm, mean_dt):
    """sq_itq_model_*: the model uploaded once gives the codes of sq_itq_hash -- single rows through the pinned
    staging, a large batch without it, float32 and float64 rows, both mean dtypes."""
    rng = np.random.default_rng(61)
    d, bits = 96, 40
    x = rng.standard_normal((200_000, d)).astype(np.float32)
    x[3] = 0.0
    mean = x[:5000].mean(axis=0).astype(mean_dt)
    rot = rng.standard_normal((d, bits))
    no = _lib.SQ_NORM_NONE if norm is None else _lib.SQ_NORM_L2
    model = _lib.ItqModel(mean, rot, no)
    for rows in (x[:1], x[3:4], x[:33], x, x[:50].astype(np.float64)):
        np.testing.assert_array_equal(model.hash(rows), _lib.itq_hash(rows, mean, rot, no))
    z = O.itq_z(x[:2000], mean, rot, norm)
    ref = O.pack_bits_msb(z >= 0)
    got = model.hash(x[:2000])
    bad = (got != ref).any(axis=1)
    assert bad.sum() <= 1
    model.close()
    with pytest.raises(_lib.HipError):
        model.hash(x[:1])


# ------------------------------------------- parity holes named by the round-1 review
@pytest.mark.parametrize("bits", [512, 1024])
def test_hamming_wide_codes(golden, bits):
    """512- and 1024-bit codes (the reference pins 1024-bit hamming_distance: tests/utils/test_metrics.py:85-104).
    Index = the 1000 `a` codes of fixture g2 plus random ones; the distance from b[j] to a[j] must be the REFERENCE's
    d[j], and the whole top-k must equal the oracle's (small index: exact keys; large index: the streaming scan)."""
    g = golden("g2_hamming.npz")
    w = bits // 64
    rng = np.random.default_rng(bits)
    if bits == 1024:
        a, b, dref = g["a_1024"], g["b_1024"], g["d_1024"]
    else:   # the first / last 8 words of the 1024-bit pairs are 512-bit pairs; their reference distance by popcount
        a = np.ascontiguousarray(g["a_1024"][:, :8])
        b = np.ascontiguousarray(g["b_1024"][:, :8])
        dref = O.popcount_u64(a ^ b).sum(axis=1).astype(np.int32)
    for extra in (0, 120_000):
        codes = np.unique(np.vstack([a, rng.integers(0, 2 ** 64, size=(extra, w), dtype=np.uint64)]), axis=0)
        row_of = {c.tobytes(): i for i, c in enumerate(codes)}
        idx = _hamming_check(codes, b[:6], 50)
        dd, ii = idx.search(b[:40], codes.shape[0] if extra == 0 else 100)
        if extra == 0:
            for j in range(40):          # every code is returned: find a[j] among b[j]'s results
                at = np.nonzero(ii[j] == row_of[a[j].tobytes()])[0]
                assert len(at) == 1 and dd[j, at[0]] == dref[j]
        # a near-duplicate of an indexed code is its own nearest neighbour at distance 1
        q = codes[[5, 17]].copy()
        q[:, w - 1] ^= np.uint64(1)
        d1, i1 = idx.search(q, 3)
        assert d1[:, 0].tolist() == [1, 1] and i1[:, 0].tolist() == [5, 17]


@pytest.mark.parametrize("tag", list(GI.DENSE_BIG_CASES))
def test_dense_golden_20k(golden, tag):
    """Fixture G5 at SURVEY 8(c)'s size (20 k x 128, 32 queries), float32 AND float64 descriptors, against the
    REFERENCE's per-row distances + stable sort.  float32 rows: sq_dense_search (ids and float32 distances bit
    identical).  float64 rows: the descriptor matrix resident as float64 (sq_rows_*: the LSH re-rank stage, the
    only float64 consumer) with every row a candidate, and sq_dense_distances."""
    g = golden("g5b_dense_nn_20k.npz")
    n, d, nq, seed, dist, dt, nq_cos = GI.DENSE_BIG_CASES[tag]
    db, qs = GI.dense_inputs(n, d, nq, seed, dist, dt)
    ridx, rdist = g[f"{tag}_euclidean_idx"], g[f"{tag}_euclidean_dist"]
    cidx, cdist = g[f"{tag}_cosine_idx"], g[f"{tag}_cosine_dist"]
    if dt == "float32":
        for cap in (0, 2048):                                     # exact keys for all rows / the MFMA scan path
            _lib.set_option("candidate_cap", cap)
            idx = _lib.DenseIndex(db)
            dd, ii = idx.search(qs, 100)
            np.testing.assert_array_equal(ii, ridx)
            np.testing.assert_array_equal(dd.view(np.uint32), rdist.view(np.uint32))
            assert (cap == 0) == (idx.stats()["scan_launches"] == 1)
            for k in (1, 10):
                np.testing.assert_array_equal(idx.search(qs, k)[1], ridx[:, :k])
            idx.close()
            ic = _lib.DenseIndex(db, metric=_lib.SQ_METRIC_COSINE)
            dd, ii = ic.search(qs[:nq_cos], 100)
            np.testing.assert_allclose(dd, cdist, rtol=1e-9, atol=1e-12)
            ic.close()
    else:
        m = _lib.RowMatrix(db)
        cand = np.tile(np.arange(n, dtype=np.int64), nq)
        off = np.arange(nq + 1, dtype=np.int64) * n
        dd, pos = m.rerank(qs, _lib.SQ_METRIC_L2, cand, off, 100)
        assert dd.dtype == np.float64
        np.testing.assert_array_equal(pos, ridx)
        np.testing.assert_array_equal(dd.view(np.uint64), rdist.view(np.uint64))    # float64, bit identical
        dd, pos = m.rerank(qs[:nq_cos], _lib.SQ_METRIC_COSINE, cand[: nq_cos * n], off[: nq_cos + 1], 100)
        np.testing.assert_allclose(dd, cdist, rtol=1e-9, atol=1e-12)
        m.close()
        full = _lib.dense_distances(qs[3], db, _lib.SQ_METRIC_L2)
        np.testing.assert_array_equal(full[ridx[3]].view(np.uint64), rdist[3].view(np.uint64))


@pytest.mark.parametrize("n", [900, 150_000])
def test_dense_non_finite_rows(n):
    """Rows holding NaN / +-inf elements.  numpy's distance to such a row is NaN or +inf; the stable sort ranks
    +inf after every number and NaN after that.  n = 150 000 goes through the bf16 scan (the scan copy of a
    non-finite row must never look like a small finite score), n = 900 through the exact keys."""
    rng = np.random.default_rng(77)
    d = 64
    db = rng.standard_normal((n, d)).astype(np.float32)
    bad = rng.choice(n, size=60, replace=False)
    for j, r in enumerate(bad):
        db[r, rng.integers(0, d)] = [np.nan, np.inf, -np.inf][j % 3]
        if j % 7 == 0:
            db[r, :] = np.nan
    db[bad[1], :2] = [np.inf, -np.inf]
    qs = rng.standard_normal((5, d)).astype(np.float32)
    with np.errstate(invalid="ignore", over="ignore"):
        idx = _lib.DenseIndex(db)
        for k in (10, 100):
            dd, ii = idx.search(qs, k)
            for qi, q in enumerate(qs):
                rd, ri = O.dense_topk(db, q, k)
                assert np.isfinite(rd).all()
                np.testing.assert_array_equal(ii[qi], ri)
                np.testing.assert_array_equal(dd[qi].view(np.uint32), rd.view(np.uint32))
        if n <= 1000:
            # k = n: the +inf and NaN rows come last, +inf before NaN, each group in row order
            dd, ii = idx.search(qs[:2], n)
            for qi in range(2):
                rd, ri = O.dense_topk(db, qs[qi], n)
                fin = np.isfinite(rd)
                np.testing.assert_array_equal(ii[qi][fin], ri[fin])
                assert set(ii[qi][~fin].tolist()) == set(ri[~fin].tolist())
                assert np.isinf(dd[qi][~fin]).sum() == np.isinf(rd[~fin]).sum()
        # a query with a non-finite element: every distance is NaN / inf; nothing may crash or hang
        qbad = qs[:1].copy()
        qbad[0, 3] = np.nan
        dd, ii = idx.search(qbad, 5)
        assert not np.isfinite(dd).any()


@pytest.mark.parametrize("bits", [64, 200, 512])
def test_hamming_append_remove_equals_a_fresh_index(bits):
    """sq_hamming_append / sq_hamming_remove: the device copy follows a set union / difference
    (linear.py:167-204) with only the new codes / the leaving ranks uploaded; after every mutation the index answers
    exactly like one created from the resulting sorted code array (ids = ranks in that array)."""
    rng = np.random.default_rng(bits)
    w = (bits + 63) // 64
    pool = rng.integers(0, 2 ** 64, size=(60_000, w), dtype=np.uint64)
    pad = w * 64 - bits
    if pad:
        pool[:, 0] &= np.uint64((1 << (64 - pad)) - 1)
    pool = np.unique(pool, axis=0)
    perm = rng.permutation(pool.shape[0])
    cur = np.unique(pool[perm[:30_000]], axis=0)
    qs = np.ascontiguousarray(pool[perm[100:110]])
    qs[5:, w - 1] ^= np.uint64(5)
    idx = _lib.HammingIndex(cur)

    def check(k=60):
        assert idx.n == cur.shape[0]
        d, i = idx.search(qs, k)
        for qi, q in enumerate(qs):
            rd, ri = O.hamming_topk(cur, q, k)
            np.testing.assert_array_equal(d[qi], rd)
            np.testing.assert_array_equal(i[qi], ri)

    check()
    steps = [("add", perm[30_000:30_700]), ("del", perm[:5000:7]), ("add", perm[30_700:52_000]), ("del", perm[40_000:41_000]),
             ("del", perm[20_000:29_000]), ("add", perm[52_000:52_003])]
    for op, sel in steps:
        other = np.unique(pool[sel], axis=0)
        merged, inv = np.unique(np.vstack([cur, other]), axis=0, return_inverse=True)
        at = np.asarray(inv).reshape(-1)[cur.shape[0]:]
        if op == "add":
            assert merged.shape[0] == cur.shape[0] + other.shape[0]           # all new
            idx.append(other, at - np.arange(other.shape[0]))
            cur = merged
        else:
            assert merged.shape[0] == cur.shape[0]                            # all present
            idx.remove(at)
            keep = np.ones(cur.shape[0], dtype=bool)
            keep[at] = False
            cur = np.ascontiguousarray(cur[keep])
        check()
    with pytest.raises(_lib.HipError):
        idx.remove(np.array([3, 3]))                                          # not strictly ascending: nothing changes
    with pytest.raises(_lib.HipError):
        idx.append(cur[:2], np.array([5, 1]))
    check(k=1)
    idx.close()


def test_k_beyond_the_one_workgroup_select():
    """The reference slices whatever n is asked (linear.py:235-238, lsh.py:513-518): no cap on k.  k above the
    one-workgroup select's capacity (16384 keys; 7168 for float64 / cosine keys) is answered by a full device sort
    (sq_select.hpp, sort_select_large) -- same canonical order, checked against the oracle."""
    rng = np.random.default_rng(61)
    db = rng.standard_normal((90_000, 32)).astype(np.float32)
    db[500:520] = db[3]                                       # ties
    qs = rng.standard_normal((3, 32)).astype(np.float32)
    _dense_check(db, qs, 20_000, "euclidean")                 # scan-sized matrix, exact path + sort
    _dense_check(db, qs[:2], 90_000, "euclidean")             # k = n
    _dense_check(db[:30_000], qs, 17_000, "euclidean")        # small matrix (every row a candidate) + sort
    _dense_check(db, qs[:2], 8_000, "cosine")
    _dense_check(db[:20_000], qs[:2], 20_000, "cosine")
    codes = np.unique(rng.integers(0, 2 ** 64, size=(80_000, 2), dtype=np.uint64), axis=0)
    hq = rng.integers(0, 2 ** 64, size=(3, 2), dtype=np.uint64)
    _hamming_check(codes, hq, 20_000)
    _hamming_check(codes[:30_000], hq, 30_000)
    for dt, kbig in ((np.float32, 18_000), (np.float64, 9_000)):
        rows = rng.standard_normal((40_000, 24)).astype(dt)
        m = _lib.RowMatrix(rows)
        cand = rng.permutation(40_000)[:25_000].astype(np.int64)
        off = np.array([0, 25_000], dtype=np.int64)
        q = rng.standard_normal((1, 24)).astype(dt)
        for metric, name in ((_lib.SQ_METRIC_L2, "euclidean"), (_lib.SQ_METRIC_COSINE, "cosine")):
            dist, pos = m.rerank(q, metric, cand, off, kbig)
            full = O.dense_distances(rows[cand], q[0], name)
            order = np.argsort(full, kind="stable")[:kbig]
            np.testing.assert_array_equal(pos[0], order)
            if name == "euclidean":
                np.testing.assert_array_equal(dist[0], full[order])
            else:
                np.testing.assert_allclose(dist[0], full[order], rtol=1e-12, atol=1e-15)
        m.close()


@pytest.mark.parametrize("metric", ["euclidean", "cosine"])
@pytest.mark.parametrize("nq,d", [(300, 128), (513, 100), (1024, 128)])
def test_dense_large_batches(metric, nq, d):
    """Batches of several hundred queries (many groups of four query tiles per scan wave, a partial last group, a
    partial last row tile): ids and float32 distances bit-identical to the oracle, no query on the exact path."""
    rng = np.random.default_rng(nq + d)
    n = 200_000 + 17
    db = rng.standard_normal((n, d)).astype(np.float32)
    db[5000:5050] = db[3]                              # ties
    qs = rng.standard_normal((nq, d)).astype(np.float32)
    qs[0], qs[nq - 1] = db[3], db[n - 1]
    m = _lib.SQ_METRIC_L2 if metric == "euclidean" else _lib.SQ_METRIC_COSINE
    idx = _lib.DenseIndex(db, metric=m)
    k = 20
    dd, ii = idx.search(qs, k)
    assert idx.stats()["fallback_queries"] == 0 and idx.stats()["scan_launches"] == 2
    for qi in list(range(0, nq, 37)) + [nq - 1]:
        rd, ri = O.dense_topk(db, qs[qi], k, metric)
        if metric == "euclidean":
            np.testing.assert_array_equal(ii[qi], ri)
            np.testing.assert_array_equal(dd[qi].view(np.uint32), rd.view(np.uint32))
        else:
            np.testing.assert_allclose(dd[qi], rd, rtol=1e-12, atol=1e-15)
            full = O.dense_distances(db, qs[qi], "cosine")
            mism = ii[qi] != ri
            assert not mism.any() or np.abs(full[ii[qi][mism]] - full[ri[mism]]).max() < 1e-14
    idx.close()
