"""
GPU tests of the plugin classes: ports of the reference's result-pinning tests
 run against the HIP implementations, plus comparisons
with the oracle and the reference-generated golden vectors.
"""
from math import sqrt

import numpy as np
import pytest

from oracle import cpu_ref as O
from smqtk_indexing_amd import _lib
from smqtk_indexing_amd._compat import (DataMemoryElement, DescriptorMemoryElement,
                                        MemoryDescriptorSet, MemoryKeyValueStore)
from smqtk_indexing_amd.impls.hash_index.hip_linear import HipLinearHashIndex
from smqtk_indexing_amd.impls.lsh_functor.hip_itq import HipItqFunctor
from smqtk_indexing_amd.impls.nn_index.hip_bruteforce import HipBruteForceNearestNeighborsIndex
from smqtk_indexing_amd.impls.nn_index.hip_lsh import HipLSHNearestNeighborIndex
from tests.golden import inputs as GI

pytestmark = pytest.mark.gpu


def _elems(x, base=0):
    return [DescriptorMemoryElement(base + i).set_vector(v) for i, v in enumerate(x)]


# ------------------------------------------------------------ HipLinearHashIndex
def test_linear_nn_known_answer():
    # tests/impls/hash_index/test_linear.py:141-155
    i = HipLinearHashIndex()
    i.build_index([[0, 1, 0], [1, 1, 0], [0, 1, 1], [0, 0, 1]])
    near_codes, near_dists = i.nn([0, 0, 0], 4)
    assert set(map(tuple, near_codes[:2].astype(int))) == {(0, 1, 0), (0, 0, 1)}
    assert set(map(tuple, near_codes[2:].astype(int))) == {(1, 1, 0), (0, 1, 1)}
    np.testing.assert_array_almost_equal(near_dists, (1 / 3., 1 / 3., 2 / 3., 2 / 3.))
    assert near_codes.dtype == bool and isinstance(near_dists, tuple)
    codes, dists = i.nn([0, 0, 0], 10)                        # n > count -> count results
    assert len(codes) == 4 and len(dists) == 4


def test_linear_nn_matches_reference_golden(golden):
    g = golden("g4_linear_hash_nn.npz")
    for tag in ("u64", "small70"):
        n, bits, seed, mode = GI.HAMMING_CASES[tag]
        codes, queries = GI.hamming_inputs(n, bits, seed, mode)
        idx = HipLinearHashIndex()
        idx.build_index(O.unpack_bits_msb(codes, bits))
        assert idx.count() == codes.shape[0]
        k = GI.HAMMING_KS[tag][-1]
        for qi, q in enumerate(queries[:4]):
            rows, dists = idx.nn(O.unpack_bits_msb(q[None], bits)[0], k)
            np.testing.assert_allclose(dists, g[f"{tag}_k{k}_dist"][qi][:len(dists)], rtol=0, atol=1e-15)
            assert len({tuple(r) for r in rows.astype(int)}) == len(rows)      # never the same code twice
        rows, dist = idx.nn_many(O.unpack_bits_msb(queries, bits), k)
        assert rows.shape == (len(queries), min(k, idx.count()), bits)


def test_linear_update_remove_then_search():
    rng = np.random.default_rng(2)
    v = rng.random((500, 64)) > 0.5
    idx = HipLinearHashIndex(DataMemoryElement())
    idx.build_index(v[:300])
    idx.update_index(v[300:])
    idx.remove_from_index(v[:50])
    left = np.unique(O.pack_bits_msb(v[50:]), axis=0)
    assert idx.count() == left.shape[0]
    rows, dists = idx.nn(v[7], 5)                             # removed code: not returned at distance 0
    rd, ri = O.hamming_topk(left, O.pack_bits_msb(v[7:8])[0], 5)
    np.testing.assert_allclose(dists, rd / 64.0)
    np.testing.assert_array_equal(O.pack_bits_msb(rows), left[ri])


# ---------------------------------------------------------------- HipItqFunctor
def test_itq_get_hash_known_answers():
    # tests/impls/lsh_functor/test_itq.py:304-336
    itq = HipItqFunctor(bit_length=1, random_seed=0)
    itq.mean_vec = np.array([0., 0.])
    itq.rotation = np.array([[1. / sqrt(2)], [1. / sqrt(2)]])
    for v, want in (([1, 1], True), ([-1, -1], False), ([-1, 1], True), ([-1.001, 1], False),
                    ([-1, 1.001], True), ([1, -1], True), ([1, -1.001], False), ([1.001, -1], True)):
        np.testing.assert_array_equal(itq.get_hash(np.array(v)), [want])
    assert itq(np.array([1, 1])).dtype == bool


def test_itq_fit_known_answer_and_cache():
    # tests/impls/lsh_functor/test_itq.py:255-302
    fit = _elems([[-2. + i, -2. + i] for i in range(5)])
    itq = HipItqFunctor(DataMemoryElement(), DataMemoryElement(), bit_length=1, random_seed=0)
    codes = itq.fit(fit)
    np.testing.assert_array_almost_equal(itq.mean_vec, [0, 0])
    np.testing.assert_array_almost_equal(itq.rotation, [[1 / sqrt(2)], [1 / sqrt(2)]])
    assert codes.shape == (5, 1)
    again = HipItqFunctor(itq.mean_vec_cache_elem, itq.rotation_cache_elem, bit_length=1)
    np.testing.assert_array_equal(again.get_hash(np.array([3., 3.])), [True])


@pytest.mark.parametrize("dt,normalize,d,bits", [(np.float32, None, 128, 64), (np.float64, 2, 96, 32),
                                                 (np.float32, 2, 40, 40), (np.float64, None, 128, 100),
                                                 # beyond one 128 x 128 output tile: the reference's own 256-d scenario
                                                 # (test_lsh.py:754-832) and BASELINE config 4's 512-d, up to 256 bits
                                                 (np.float64, None, 256, 32), (np.float32, 2, 512, 256),
                                                 (np.float32, None, 200, 130)])
def test_itq_fit_on_device_matches_host(dt, normalize, d, bits):
    """sq_itqfit_*: mean / covariance / projection / per-iteration B^T V on the device against
    numpy on identical inputs, then the fitted model against the host fit (same seed) by its
    quantisation error."""
    rng = np.random.default_rng(d + bits)
    basis = rng.standard_normal((d, d)) * np.linspace(3.0, 0.2, d)[None, :]
    x = (rng.standard_normal((6000, d)) @ basis.T + rng.standard_normal(d) * 2.0).astype(dt)
    ordv = _lib.SQ_NORM_NONE if normalize is None else _lib.SQ_NORM_L2
    xn = O.itq_norm_vector(x, normalize)
    fit = _lib.ItqFit(x, ordv)
    np.testing.assert_allclose(fit.mean, xn.astype(np.float64).mean(axis=0), rtol=1e-6, atol=1e-7)
    mean = xn.mean(axis=0)
    fit.set_mean(mean)
    xc = xn.astype(np.float64) - mean.astype(np.float64)
    cov = fit.cov()
    np.testing.assert_allclose(cov, np.cov(xc.T), rtol=1e-5, atol=1e-6 * np.abs(cov).max())
    evals, evecs = np.linalg.eigh(np.cov(xc.T))
    pc = evecs[:, np.argsort(evals)[::-1][:bits]]
    fit.project(pc)
    v = xc @ pc
    r, _ = np.linalg.qr(rng.standard_normal((bits, bits)))
    c = fit.iterate(r)
    ref_c = np.where(v @ r >= 0, 1.0, -1.0).T @ v
    np.testing.assert_allclose(c, ref_c, rtol=1e-5, atol=1e-5 * np.abs(ref_c).max())
    fit.close()

    elems = [DescriptorMemoryElement(i).set_vector(row) for i, row in enumerate(x)]
    dev = HipItqFunctor(bit_length=bits, itq_iterations=15, normalize=normalize, random_seed=7)
    host = HipItqFunctor(bit_length=bits, itq_iterations=15, normalize=normalize, random_seed=7, fit_on_device=False)
    cd, ch = dev.fit(elems), host.fit(elems)
    assert dev.mean_vec.dtype == host.mean_vec.dtype
    np.testing.assert_allclose(dev.mean_vec, host.mean_vec, rtol=1e-4, atol=1e-5)

    def quant_error(f):
        z = O.itq_z(x, f.mean_vec, np.real(f.rotation), normalize)
        return np.linalg.norm(np.where(z >= 0, 1.0, -1.0) - z)

    # eigenvector signs (LAPACK's choice on two covariances that differ in the last bits) make the two
    # runs land in different but equivalent optima: the objective agrees, the codes need not
    assert abs(quant_error(dev) - quant_error(host)) <= 1e-2 * quant_error(host)
    assert cd.shape == ch.shape == (6000, bits) and abs(cd.mean() - 0.5) < 0.05


def test_itq_functor_matches_reference_golden(golden):
    g = golden("g3_itq_hash.npz")
    n, d, bits, seed = GI.ITQ_CASES["a"]
    x32, mean, rot = GI.itq_inputs(n, d, bits, seed)
    for norm in (None, 2):
        f = HipItqFunctor(bit_length=bits, normalize=norm)
        f.mean_vec, f.rotation = mean, rot
        got = O.pack_bits_msb(f.get_hash(x32))
        key = f"a_n{norm}_float32"
        bad = (got != g[key + "_packed"]).any(axis=1)
        assert bad.sum() == 0 or g[key + "_minabsz"][bad].max() < 1e-10
        one = f.get_hash(x32[5])
        assert one.shape == (bits,) and one.dtype == bool
        np.testing.assert_array_equal(O.pack_bits_msb(one[None])[0], got[5])


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("norm", [1, 0, np.inf, -np.inf, 3, 0.5])
def test_itq_every_normalize_order(norm, dt):
    """ItqFunctor accepts any order numpy.linalg.norm takes for a vector (itq.py:172-191).  1, 0 and +-inf are
    evaluated on the device in numpy's arithmetic (row norms in the rows' dtype, 0 -> 1, element-wise division);
    a general p is normalised by numpy on the host and hashed on the device.  Codes = the oracle's."""
    rng = np.random.default_rng(31)
    n, d, bits = 3000, 128, 64
    x = (rng.standard_normal((n, d)) * 2).astype(dt)
    x[::9, ::3] = 0.0                      # zeros: ord 0 / -inf see them
    x[5] = 0.0                             # a zero row: norm replaced by 1
    mean = (rng.standard_normal(d) * 0.01).astype(np.float64)
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    rot = np.ascontiguousarray(q[:, :bits])
    f = HipItqFunctor(bit_length=bits, normalize=norm)
    f.mean_vec, f.rotation = mean, rot
    got = f.get_hash(x)
    z = O.itq_z(x, mean, rot, norm)
    ref = z >= 0
    bad = (got != ref).any(axis=1)
    assert bad.sum() == 0 or np.abs(z[bad]).min(axis=1).max() < 1e-9 * np.abs(z).max()
    one = f.get_hash(x[17])
    np.testing.assert_array_equal(one, got[17])
    if norm in (1, np.inf):                # the packed C ABI call with the order code
        code = _lib.SQ_NORM_L1 if norm == 1 else _lib.SQ_NORM_INF
        np.testing.assert_array_equal(_lib.itq_hash(x, mean, rot, code), O.pack_bits_msb(got))


# ------------------------------------------- HipBruteForceNearestNeighborsIndex
def test_bruteforce_known_answers():
    # tests/impls/nn_index/test_faiss.py:443-515 (the reference's exact-index KATs)
    dim = 5
    index = HipBruteForceNearestNeighborsIndex()
    index.build_index(_elems(np.eye(dim)))
    q = DescriptorMemoryElement("q").set_vector(np.zeros(dim))
    r, dists = index.nn(q, dim)
    assert len(r) == dim and all(d == 1.0 for d in dists)
    r, dists = index.nn(_elems(np.eye(dim))[3], 1)
    assert r[0].uuid() == 3 and dists[0] == 0.0
    pts = np.array([[j, 2. * j] for j in range(1000)])
    order = np.random.default_rng(0).permutation(1000)
    index.build_index([DescriptorMemoryElement(int(j)).set_vector(pts[j]) for j in order])
    q.set_vector(np.zeros(2))
    r, dists = index.nn(q, 100)
    assert [e.uuid() for e in r] == list(range(100))
    assert all(b > a for a, b in zip(dists, dists[1:]))
    x = np.random.default_rng(1).random((10_000, 256))
    index.build_index(_elems(x))
    r, dists = index.nn(_elems(x[:1] + 1e-3)[0], 10)
    assert len(r) == 10 and r[0].uuid() == 0


def test_bruteforce_float64_descriptors_get_original_dtype_distances():
    """faiss.py:776, 818-824: float32 search, then the distances of the n results from the
    original (float64) vectors against the float32 query, results ordered by them."""
    rng = np.random.default_rng(31)
    x = rng.standard_normal((5000, 48))                         # float64 descriptors (SMQTK's default dtype)
    index = HipBruteForceNearestNeighborsIndex()
    index.build_index(_elems(x))
    q = rng.standard_normal(48)
    r, dists = index.nn(DescriptorMemoryElement("q").set_vector(q), 25)
    q32 = q.astype(np.float32)
    ref_f32 = O.dense_topk(x.astype(np.float32), q32, 25, "euclidean")[1]          # the float32 search's choice
    exact = O.dense_distances(x[ref_f32], q32, "euclidean")                       # float64 rows, float32 query
    order = np.argsort(exact, kind="stable")
    assert [e.uuid() for e in r] == [int(ref_f32[i]) for i in order]
    np.testing.assert_array_equal(np.asarray(dists), exact[order])
    assert np.asarray(dists).dtype == np.float64 and all(b >= a for a, b in zip(dists, dists[1:]))


def test_bruteforce_float64_descriptors_cosine():
    """The same recompute for the cosine metric: float64 descriptors (SMQTK's default dtype), float32 search, the
    distances of the n results from the ORIGINAL float64 vectors against the float32 query (faiss.py:776, 818-824
    with metrics.cosine_distance, utils/metrics.py:89-137), results ordered by them."""
    rng = np.random.default_rng(32)
    x = rng.standard_normal((6000, 40)) + 0.3
    index = HipBruteForceNearestNeighborsIndex(distance_method="cosine")
    index.build_index(_elems(x))
    q = rng.standard_normal(40) + 0.3
    r, dists = index.nn(DescriptorMemoryElement("q").set_vector(q), 30)
    q32 = q.astype(np.float32)
    ref_f32 = O.dense_topk(x.astype(np.float32), q32, 30, "cosine")[1]              # the float32 search's choice
    exact = O.dense_distances(x[ref_f32], q32, "cosine")                            # float64 rows, float32 query
    order = np.argsort(exact, kind="stable")
    assert [e.uuid() for e in r] == [int(ref_f32[i]) for i in order]
    np.testing.assert_allclose(np.asarray(dists), exact[order], rtol=1e-12, atol=1e-15)
    assert np.asarray(dists).dtype == np.float64 and all(b >= a for a, b in zip(dists, dists[1:]))


def test_bruteforce_matches_reference_golden(golden):
    g = golden("g5_dense_nn.npz")
    n, d, nq, seed, dist, dt = GI.DENSE_CASES["nrm128"]
    db, qs = GI.dense_inputs(n, d, nq, seed, dist, dt)
    for metric in ("euclidean", "cosine"):
        index = HipBruteForceNearestNeighborsIndex(metric)
        index.build_index(_elems(db))
        for qi in range(2):
            r, dists = index.nn(DescriptorMemoryElement("q").set_vector(qs[qi]), 100)
            if metric == "euclidean":
                assert [e.uuid() for e in r] == g["nrm128_euclidean_idx"][qi].tolist()
                np.testing.assert_array_equal(np.float32(dists), g["nrm128_euclidean_dist"][qi])
            else:
                np.testing.assert_allclose(dists, g["nrm128_cosine_dist"][qi], rtol=1e-9, atol=1e-12)
        index.remove_from_index([int(g[f"nrm128_{metric}_idx"][0][0])])
        r, _ = index.nn(DescriptorMemoryElement("q").set_vector(qs[0]), 1)
        assert r[0].uuid() == int(g[f"nrm128_{metric}_idx"][0][1])


def test_bruteforce_removal_keeps_the_resident_index():
    """remove_from_index with a resident device index: tombstones (no re-upload, no re-index), searches answer like
    an index built from the live rows -- ids, float32 distances and tie order -- until a quarter of it is dead."""
    rng = np.random.default_rng(77)
    x = rng.standard_normal((4000, 48)).astype(np.float32)
    x[1500] = x[20]                                             # a duplicate pair: tie order = row order
    for metric in ("euclidean", "cosine"):
        index = HipBruteForceNearestNeighborsIndex(metric)
        index.build_index(_elems(x))
        q = DescriptorMemoryElement("q").set_vector(x[20])
        index.nn(q, 3)
        dev = index._dev
        gone = sorted(rng.choice(4000, 300, replace=False).tolist() + [20])
        gone = sorted(set(gone) - {1500})
        index.remove_from_index(gone)
        assert index._dev is dev and len(index._dead) == len(gone) and index.count() == 4000 - len(gone)
        live = np.setdiff1d(np.arange(4000), gone)
        for qv in (x[20], x[7], rng.standard_normal(48).astype(np.float32)):
            r, dists = index.nn(DescriptorMemoryElement("q").set_vector(qv), 40)
            rd, ri = O.dense_topk(x[live], qv, 40, metric)
            assert [e.uuid() for e in r] == live[ri].tolist()
            if metric == "euclidean":
                np.testing.assert_array_equal(np.float32(dists), rd)
            else:
                np.testing.assert_allclose(dists, rd, rtol=1e-12, atol=1e-15)
        with pytest.raises(KeyError):
            index.remove_from_index([gone[0]])                  # already gone
        # everything that is left, in one query (k + tombstones = the whole resident matrix)
        r, _ = index.nn(q, index.count())
        assert sorted(e.uuid() for e in r) == live.tolist()
        # new rows behind the tombstones, then removals past a quarter: rebuilt from the live rows
        index.update_index(_elems(x[:10] + 1.0, base=9000))
        assert index._dev is dev and index.count() == 4010 - len(gone)
        index.remove_from_index(live[:900].tolist())
        assert not index._dead and index.count() == 4010 - len(gone) - 900
        r, _ = index.nn(q, 5)
        keep = np.concatenate([live[900:], 9000 + np.arange(10)])
        allx = np.vstack([x, np.zeros((5000, 48), np.float32), x[:10] + 1.0])
        assert [e.uuid() for e in r] == keep[O.dense_topk(allx[keep], x[20], 5, metric)[1]].tolist()


# ------------------------------------------------ HipLSHNearestNeighborIndex
def _lsh(bits, metric, hash_index=True, x=None, seed=0, iters=50):
    f = HipItqFunctor(bit_length=bits, random_seed=seed, itq_iterations=iters)
    if x is not None:
        f.fit(_elems(x))
    return HipLSHNearestNeighborIndex(f, MemoryDescriptorSet(), MemoryKeyValueStore(),
                                      HipLinearHashIndex() if hash_index else None, distance_method=metric)


@pytest.mark.parametrize("hash_index", [True, False])
def test_lsh_random_euclidean(hash_index):
    # tests/impls/nn_index/test_lsh.py:754-832
    np.random.seed(0)
    x = np.random.rand(1000, 256)
    index = _lsh(32, "euclidean", hash_index, x)
    index.build_index(_elems(x))
    assert index.count() == 1000
    q = DescriptorMemoryElement("q")
    for i in (0, 500, 999):
        q.set_vector(x[i])
        r, dists = index.nn(q, 1)
        assert r[0].uuid() == i and dists[0] == 0.0
        q.set_vector(x[i] + 1e-4)
        r, dists = index.nn(q, 1)
        assert r[0].uuid() == i
    q.set_vector(np.random.rand(256))
    for n in (10, 1000):
        r, dists = index.nn(q, n)
        assert len(dists) <= n
        assert all(b > a for a, b in zip(dists, dists[1:]))      # strictly increasing


def test_lsh_known_unit_and_ordered():
    # tests/impls/nn_index/test_lsh.py:837-979
    dim = 5
    index = _lsh(dim, "euclidean", True, np.eye(dim))
    index.build_index(_elems(np.eye(dim)))
    q = DescriptorMemoryElement("q").set_vector(np.zeros(dim))
    r, dists = index.nn(q, dim)
    assert len(r) == dim and all(d == 1.0 for d in dists)
    r, dists = index.nn(_elems(np.eye(dim))[2], 1)
    assert r[0].uuid() == 2 and dists[0] == 0.0
    pts = np.array([[j, 2. * j] for j in range(1000)])
    order = np.random.default_rng(0).permutation(1000)
    d_set = [DescriptorMemoryElement(int(j)).set_vector(pts[j]) for j in order]
    index = _lsh(1, "euclidean", True, pts)
    index.build_index(d_set)
    q.set_vector(np.zeros(2))
    r, dists = index.nn(q, 5)
    assert [e.uuid() for e in r] == [0, 1, 2, 3, 4]
    r, dists = index.nn(q, 1000)                                  # n covers every code: exact brute force
    assert [e.uuid() for e in r] == list(range(1000))


def test_lsh_matches_reference_golden(golden):
    g = golden("g6_lsh_nn.npz")
    for tag, (n, d, bits, seed, metric, ns) in GI.LSH_CASES.items():
        db, qs = GI.lsh_inputs(n, d, seed)
        f = HipItqFunctor(bit_length=bits)
        f.mean_vec, f.rotation = g[f"{tag}_mean"], g[f"{tag}_rot"]      # the model the reference fitted
        index = HipLSHNearestNeighborIndex(f, MemoryDescriptorSet(), MemoryKeyValueStore(), HipLinearHashIndex(),
                                           distance_method=metric)
        index.build_index(_elems(db))
        assert index.count() == int(g[f"{tag}_count"])
        assert len(list(index.hash2uuids_kvstore.keys())) == int(g[f"{tag}_ncodes"])
        nn_all = max(ns)
        if nn_all >= n:
            for qi, q in enumerate(qs):
                r, dists = index.nn(DescriptorMemoryElement("q").set_vector(q), nn_all)
                ru, rd = g[f"{tag}_n{nn_all}_uuids"][qi], g[f"{tag}_n{nn_all}_dist"][qi]
                np.testing.assert_allclose(dists, rd, rtol=1e-12)
                dd = np.asarray(dists)
                if (dd[1:] != dd[:-1]).all():
                    assert [e.uuid() for e in r] == ru.tolist()
        for qi, q in enumerate(qs):                                # first result always agrees on distance
            r, dists = index.nn(DescriptorMemoryElement("q").set_vector(q), min(ns))
            assert len(r) >= 1 and dists[0] >= 0


def test_row_matrix_rerank_matches_oracle():
    """sq_rows_rerank: gathered candidates, reference-arithmetic distances, stable top-k."""
    rng = np.random.default_rng(5)
    for dt in (np.float32, np.float64):
        rows = rng.standard_normal((5000, 96)).astype(dt)
        rows[100] = rows[7]                                        # a distance tie inside a candidate list
        qs = rng.standard_normal((4, 96)).astype(dt)
        m = _lib.RowMatrix(rows)
        cands = [np.array([7, 3, 100, 4999, 12, 7]), rng.permutation(5000)[:700], np.zeros(0, dtype=np.int64),
                 np.arange(64)[::-1].copy()]
        off = np.zeros(5, dtype=np.int64)
        off[1:] = np.cumsum([len(c) for c in cands])
        flat = np.concatenate(cands).astype(np.int64)
        for metric, name in ((_lib.SQ_METRIC_L2, "euclidean"), (_lib.SQ_METRIC_COSINE, "cosine")):
            dist, pos = m.rerank(qs, metric, flat, off, 50)
            for qi, c in enumerate(cands):
                full = O.dense_distances(rows[c], qs[qi], name) if len(c) else np.zeros(0)
                order = np.argsort(full, kind="stable")[:50]
                kk = len(order)
                np.testing.assert_array_equal(pos[qi, :kk], order)
                assert (pos[qi, kk:] == -1).all()
                if name == "euclidean":
                    np.testing.assert_array_equal(dist[qi, :kk], full[order])
                else:
                    np.testing.assert_allclose(dist[qi, :kk], full[order], rtol=1e-12, atol=1e-15)
        m.close()


@pytest.mark.parametrize("metric", ["euclidean", "cosine"])
@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("hash_index", [True, False])
def test_lsh_device_rerank_equals_host_path(metric, dt, hash_index):
    """The device mirror (CSR bucket expansion + sq_rows_rerank) returns what the host path
    (dictionary lookups + sq_dense_distances + stable sort) returns, single and batched,
    also after update_index / remove_from_index."""
    rng = np.random.default_rng(11)
    x = rng.standard_normal((3000, 64)).astype(dt)
    f = HipItqFunctor(bit_length=12, itq_iterations=5, random_seed=3)
    f.fit([DescriptorMemoryElement(i).set_vector(v) for i, v in enumerate(x[:500])])

    def make(device):
        idx = HipLSHNearestNeighborIndex(f, MemoryDescriptorSet(), MemoryKeyValueStore(),
                                         HipLinearHashIndex() if hash_index else None, distance_method=metric,
                                         device_rerank=device)
        idx.build_index(_elems(x[:2500]))
        return idx

    dev, host = make(True), make(False)
    assert dev._mirror is not None and host._mirror is None
    qs = [DescriptorMemoryElement(f"q{i}").set_vector(v) for i, v in enumerate(rng.standard_normal((6, 64)).astype(dt))]

    def same(a, b):
        (ra, da), (rb, db) = a, b
        np.testing.assert_allclose(da, db, rtol=1e-12, atol=0)
        da = np.asarray(da)
        if len(da) > 1 and (da[1:] != da[:-1]).all():
            assert [e.uuid() for e in ra] == [e.uuid() for e in rb]
        assert len(ra) == len(rb)

    for n in (1, 7, 40):
        batch = dev.nn_many(qs, n)
        for q, b in zip(qs, batch):
            same(dev.nn(q, n), host.nn(q, n))
            same(b, host.nn(q, n))
    mirror = dev._mirror
    for idx in (dev, host):
        idx.update_index(_elems(x[2500:], base=2500))
        idx.remove_from_index(list(range(0, 300)))
    # a tenth of the rows removed: the resident descriptors stay (tombstones), only the bucket map is re-derived
    assert dev._mirror is mirror and int(mirror.dead.sum()) == 300 and mirror.rows.n == 3000
    for q in qs:
        same(dev.nn(q, 25), host.nn(q, 25))
        assert all(e.uuid() >= 300 for e in dev.nn(q, 25)[0])
    for b, q in zip(dev.nn_many(qs, 25), qs):
        same(b, host.nn(q, 25))
    extra = rng.standard_normal((50, 64)).astype(dt)
    for idx in (dev, host):                       # an append behind tombstones, then removals past a quarter
        idx.update_index(_elems(extra, base=5000))
    assert dev._mirror is mirror and mirror.rows.n == 3050
    for q in qs:
        same(dev.nn(q, 25), host.nn(q, 25))
    for idx in (dev, host):
        idx.remove_from_index(list(range(300, 1000)))
    assert dev._mirror is None                     # too many tombstones: rebuilt from the descriptor set
    for q in qs:
        same(dev.nn(q, 25), host.nn(q, 25))
    assert dev._mirror is not None and len(dev._mirror.uuids) == 2050 and not dev._mirror.dead.any()


def test_lsh_config_roundtrip_on_gpu():
    idx = _lsh(4, "euclidean")
    j = HipLSHNearestNeighborIndex.from_config(idx.get_config())
    assert isinstance(j.hash_index, HipLinearHashIndex) and j.lsh_functor.bit_length == 4


def test_sharded_index_rccl_world1():
    """The collective path on the real backend (RCCL) with one rank: shard with an
    id offset, all-gather of the per-shard top-k, host merge."""
    import os
    import torch
    import torch.distributed as dist
    from smqtk_indexing_amd.distributed import dense_shard, hamming_shard

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = "29517"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        rng = np.random.default_rng(21)
        db = rng.standard_normal((90_000, 128)).astype(np.float32)
        qs = rng.standard_normal((5, 128)).astype(np.float32)
        shard = dense_shard(torch.from_numpy(db).to(dev), row0=1000)
        d, i = shard.search(torch.from_numpy(qs).to(dev), 20)
        for j in range(5):
            rd, ri = O.dense_topk(db, qs[j], 20)
            np.testing.assert_array_equal(i[j], ri + 1000)
            np.testing.assert_array_equal(d[j].view(np.uint32), rd.view(np.uint32))
        codes = np.unique(rng.integers(0, 2 ** 63, size=(80_000, 1), dtype=np.int64).astype(np.uint64), axis=0)
        hs = hamming_shard(torch.from_numpy(codes.view(np.int64)).to(dev), row0=7)
        qc = codes[:3]
        d, i = hs.search(torch.from_numpy(qc.view(np.int64)).to(dev), 10)
        for j in range(3):
            rd, ri = O.hamming_topk(codes, qc[j], 10)
            np.testing.assert_array_equal(i[j], ri + 7)
            np.testing.assert_array_equal(d[j], rd)
    finally:
        dist.destroy_process_group()


def test_mutable_sharded_index_rccl_world1():
    """SURVEY 8e "Mutations" on the real backend with one rank: the HIP dense / Hamming indexes behind
    MutableShardedIndex -- append (new ids), tombstones (answered with k + dead), compaction, KeyError."""
    import os
    import torch
    import torch.distributed as dist
    from smqtk_indexing_amd.distributed import MutableShardedIndex, dense_local_builder, hamming_local_builder

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = "29519"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        rng = np.random.default_rng(31)
        db = rng.standard_normal((70_000, 64)).astype(np.float32)
        db[60_000] = db[11]
        qs = rng.standard_normal((4, 64)).astype(np.float32)
        qs[0] = db[11]
        idx = MutableShardedIndex(torch.from_numpy(db).to(dev), 0, len(db), dense_local_builder(), compact_at=0.01)
        live = {i: db[i] for i in range(len(db))}

        def check(k=30):
            ids = np.array(sorted(live))
            mat = np.stack([live[i] for i in ids])
            d, i = idx.search(torch.from_numpy(qs).to(dev), k)
            for j in range(len(qs)):
                rd, ri = O.dense_topk(mat, qs[j], k)
                np.testing.assert_array_equal(i[j], ids[ri])
                np.testing.assert_array_equal(d[j].view(np.uint32), rd.view(np.uint32))
        check()
        new = rng.standard_normal((500, 64)).astype(np.float32)
        new[7] = db[11]
        nid = idx.append(torch.from_numpy(new))
        live.update({int(a): v for a, v in zip(nid, new)})
        check()
        gone = [11, 60_000, int(nid[7])] + list(range(200, 260))
        idx.remove(gone)                              # 63 tombstones: below the 1 % compaction mark
        assert int(idx.dead.sum()) == 63
        for g_ in gone:
            del live[g_]
        check()
        with pytest.raises(KeyError):
            idx.remove([11])
        idx.remove(list(range(1000, 1800)))           # passes 1 %: the shard compacts and rebuilds
        assert int(idx.dead.sum()) == 0 and idx.count() == len(live) - 800
        for g_ in range(1000, 1800):
            del live[g_]
        check()
        # packed codes
        codes = np.unique(rng.integers(0, 2 ** 63, size=(90_000, 1), dtype=np.int64).astype(np.uint64), axis=0)
        rng.shuffle(codes)                            # arbitrary row order: ids are positions
        hidx = MutableShardedIndex(torch.from_numpy(codes.view(np.int64)).to(dev), 0, len(codes), hamming_local_builder(),
                                   dist_dtype=torch.int32)
        hidx.remove([5, 6, 7])
        keep = np.ones(len(codes), bool)
        keep[[5, 6, 7]] = False
        d, i = hidx.search(torch.from_numpy(codes[4:8].view(np.int64)).to(dev), 10)
        ids = np.nonzero(keep)[0]
        for j in range(4):
            dist_all = np.array([bin(int(c) ^ int(codes[4 + j, 0])).count("1") for c in codes[keep, 0]])
            order = np.lexsort((ids, dist_all))[:10]
            np.testing.assert_array_equal(i[j], ids[order])
            np.testing.assert_array_equal(d[j], dist_all[order])
    finally:
        dist.destroy_process_group()


def test_bruteforce_update_appends_on_the_device():
    """update_index with only new uuids takes sq_dense_append (no rebuild); with a replaced uuid it rebuilds.
    Either way the answers are those of a freshly built index."""
    rng = np.random.default_rng(8)
    x = rng.standard_normal((66_000, 32)).astype(np.float32)
    idx = HipBruteForceNearestNeighborsIndex()
    idx.build_index(_elems(x[:65_700]))
    q = DescriptorMemoryElement("q").set_vector(x[65_800] + 0.001)
    idx.nn(q, 3)                                                  # the device copy exists from here on
    dev = idx._dev
    idx.update_index(_elems(x[65_700:], base=65_700))
    assert idx._dev is dev and dev.n == 66_000                    # appended in place
    r, dist = idx.nn(q, 5)
    fresh = HipBruteForceNearestNeighborsIndex()
    fresh.build_index(_elems(x))
    r2, dist2 = fresh.nn(q, 5)
    assert [e.uuid() for e in r] == [e.uuid() for e in r2] and r[0].uuid() == 65_800
    assert dist == dist2
    idx.update_index([DescriptorMemoryElement(7).set_vector(x[65_800] + 0.0001)])   # replaces uuid 7: rebuild
    assert idx._dev is not dev and idx.count() == 66_000
    assert idx.nn(q, 1)[0][0].uuid() == 7


@pytest.mark.parametrize("use_async", [False, True])
def test_pipelined_sharded_search_rccl_world1(use_async):
    """PipelinedShardedSearch: the all-gather and the host merge of a batch run under the next batch's
    search; results come back two (asynchronous searches: three) submits later, in order, and equal the direct search."""
    import os
    import torch
    import torch.distributed as dist
    from smqtk_indexing_amd.distributed import PipelinedShardedSearch

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = "29522" if use_async else "29521"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        rng = np.random.default_rng(17)
        db = torch.from_numpy(rng.standard_normal((120_000, 64)).astype(np.float32)).to(dev)
        index = _lib.DenseIndex(db.data_ptr(), n=db.shape[0], d=64, device_ptr=True, id_base=500, keepalive=db)
        nq, k = 7, 15
        batches = [torch.from_numpy(rng.standard_normal((nq, 64)).astype(np.float32)).to(dev) for _ in range(6)]
        pipe = PipelinedShardedSearch(index, nq, k, torch.float32, device=dev, use_async=use_async)
        assert pipe.lag == (1 if use_async else 0)
        got = []
        for i, q in enumerate(batches):
            r = pipe.submit(q)
            assert (r is None) == (i < 2 + pipe.lag)
            if r is not None:
                got.append(r)
        got += pipe.flush()
        assert len(got) == len(batches) and pipe.flush() == []
        for q, (d, i) in zip(batches, got):
            rd, ri = index.search(q.cpu().numpy(), k)
            np.testing.assert_array_equal(i, ri)
            np.testing.assert_array_equal(d.view(np.uint32), rd.view(np.uint32))
        r = pipe.submit(batches[0])                                   # usable again after a flush
        assert r is None and len(pipe.flush()) == 1
        pipe.close()
        index.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("hash_index", [True, False])
def test_lsh_update_appends_to_the_device_mirror(hash_index):
    """update_index with new uuids only keeps the device mirror and uploads just the new descriptors
    (sq_rows_append); the answers are those of an index built from everything.  A replaced uuid drops the mirror."""
    rng = np.random.default_rng(23)
    x = rng.standard_normal((4000, 48)).astype(np.float32)
    f = HipItqFunctor(bit_length=10, itq_iterations=4, random_seed=1)
    f.fit([DescriptorMemoryElement(i).set_vector(v) for i, v in enumerate(x[:600])])

    def make(rows):
        idx = HipLSHNearestNeighborIndex(f, MemoryDescriptorSet(), MemoryKeyValueStore(),
                                         HipLinearHashIndex() if hash_index else None, distance_method="euclidean")
        idx.build_index(_elems(rows))
        return idx

    grown, whole = make(x[:3000]), make(x)
    qs = [DescriptorMemoryElement(f"q{i}").set_vector(v) for i, v in enumerate(rng.standard_normal((5, 48)).astype(np.float32))]
    grown.nn(qs[0], 3)
    mirror = grown._mirror
    assert mirror is not None
    grown.update_index(_elems(x[3000:3600], base=3000))
    grown.update_index(_elems(x[3600:], base=3600))
    assert grown._mirror is mirror and mirror.rows.n == 4000 and len(mirror.uuids) == 4000
    for q in qs:
        for n in (1, 20):
            (ra, da), (rb, db) = grown.nn(q, n), whole.nn(q, n)
            assert da == db and [e.uuid() for e in ra] == [e.uuid() for e in rb]
    grown.update_index([DescriptorMemoryElement(5).set_vector(x[5] + 1.0)])     # replaces uuid 5
    assert grown._mirror is None


# ----------------------------------------- reference scenarios and reference-written caches (fixtures g6b, g8)
@pytest.mark.parametrize("hi_tag", ["none", "linear"])
def test_lsh_reference_scenarios(golden, hi_tag):
    """The three TestLshIndexAlgorithms scenarios (tests/impls/nn_index/test_lsh.py:754-979) with the models the
    REFERENCE fitted and the results it returned: the HIP index must return admissible results with the reference's
    distances (which Hamming-tied codes enter at rank n is set-order dependent upstream), the identical answer when
    n covers every code, and pass the scenarios' own assertions."""
    from tests.test_oracle_golden import _check_lsh_admissible, _lsh_state, lsh_scenarios
    g = golden("g6b_lsh_scenarios.npz")
    for name, uuids, rows, (mean, rot), queries in lsh_scenarios(g, hi_tag):
        f = HipItqFunctor(bit_length=rot.shape[1])
        f.mean_vec, f.rotation = mean, rot
        index = HipLSHNearestNeighborIndex(f, MemoryDescriptorSet(), MemoryKeyValueStore(),
                                           HipLinearHashIndex() if hi_tag == "linear" else None,
                                           distance_method="euclidean")
        index.build_index([DescriptorMemoryElement(u).set_vector(r) for u, r in zip(uuids, rows)])
        uniq, buckets = _lsh_state(rows, mean, rot)
        row_of = {u: r for r, u in enumerate(uuids)}
        for qname, qv, nn in queries:
            r, dists = index.nn(DescriptorMemoryElement("q").set_vector(qv), nn)
            r_uu, r_dist = g[f"{name}_{hi_tag}_{qname}_uuids"], g[f"{name}_{hi_tag}_{qname}_dist"]
            got_rows = np.array([row_of[e.uuid()] for e in r])
            _check_lsh_admissible(qv, nn, mean, rot, uniq, buckets, rows, "euclidean", got_rows, np.asarray(dists))
            if nn >= uniq.shape[0]:
                np.testing.assert_allclose(dists, r_dist, rtol=1e-12)
                dd = np.asarray(dists)
                if (dd[1:] != dd[:-1]).all():
                    assert [e.uuid() for e in r] == r_uu.tolist()
            if qname in ("self255_n1", "near0_n1", "e3_n1", "origin_n5"):
                assert [e.uuid() for e in r] == r_uu.tolist() and list(dists) == r_dist.tolist()
            if qname == "zero_n5":
                assert list(dists) == [1.0] * 5
        if name == "rand":
            assert index.count() == int(g[f"rand_{hi_tag}_count"])


def test_reference_written_caches_on_device(golden):
    """Fixture g8: model / cache bytes written by the reference's ItqFunctor.save_model (itq.py:222-237) and
    LinearHashIndex.save_cache (linear.py:133-142) are loaded by the HIP classes, which then answer as the reference."""
    g = golden("g8_reference_caches.npz")
    x, _ = GI.lsh_inputs(300, 24, 8)
    for tag in ("float64", "float32"):
        f = HipItqFunctor(mean_vec_cache=DataMemoryElement(g[f"itq_{tag}_mean_bytes"].tobytes()),
                          rotation_cache=DataMemoryElement(g[f"itq_{tag}_rot_bytes"].tobytes()), bit_length=12)
        probe = x[:40].astype(tag)
        np.testing.assert_array_equal(f.get_hash(probe), g[f"itq_{tag}_probe_codes"])
        np.testing.assert_array_equal(np.vstack([f.get_hash(r) for r in probe]), g[f"itq_{tag}_probe_codes"])
    for tag in ("b20", "b62"):
        bits = int(tag[1:])
        idx = HipLinearHashIndex(cache_element=DataMemoryElement(g[f"lin_{tag}_cache_bytes"].tobytes()))
        rows, dists = idx.nn(g[f"lin_{tag}_q"], 7)
        np.testing.assert_allclose(dists, g[f"lin_{tag}_nn_dist"], rtol=0, atol=1e-15)
        codes = g[f"lin_{tag}_codes"]
        full = O.popcount_u64(codes ^ O.pack_bits_msb(g[f"lin_{tag}_q"][None])[0][None, :]).sum(axis=1)
        lut = {O.packed_to_int(r): i for i, r in enumerate(codes)}
        O.assert_topk_equivalent(np.rint(g[f"lin_{tag}_nn_dist"] * bits).astype(np.int32),
                                 np.array([lut[O.packed_to_int(r)] for r in g[f"lin_{tag}_nn_codes"]]),
                                 np.rint(np.asarray(dists) * bits).astype(np.int32),
                                 np.array([lut[O.packed_to_int(r)] for r in O.pack_bits_msb(rows)]),
                                 all_dist_of=lambda r: full[r])


def test_linear_update_remove_keep_the_device_copy():
    """HipLinearHashIndex.update_index / remove_from_index with a live device index: the device copy is mutated in
    place (sq_hamming_append / sq_hamming_remove), never re-uploaded, and answers like a freshly built index."""
    rng = np.random.default_rng(44)
    bits = 96
    hv = rng.random((9000, bits)) > 0.5
    a = HipLinearHashIndex()
    a.build_index(hv[:5000])
    q = hv[77]
    a.nn(q, 5)                                   # creates the device copy
    dev = a._dev
    assert dev is not None
    a.update_index(hv[5000:8000])
    a.update_index(hv[4990:5010])                # mostly known codes: only the new ones are appended
    assert a._dev is dev and dev.n == a.count()
    with pytest.raises(KeyError):
        a.remove_from_index([hv[8500]])          # unknown code: nothing changes
    assert a._dev is dev and dev.n == a.count()
    a.remove_from_index(hv[100:900])
    assert a._dev is dev and dev.n == a.count()
    b = HipLinearHashIndex()
    b.build_index(np.vstack([hv[:100], hv[900:8000]]))
    assert a.count() == b.count()
    for qv in (hv[77], hv[6000], hv[8999], ~hv[3]):
        for n in (1, 40, 700):
            ra, da = a.nn(qv, n)
            rb, db = b.nn(qv, n)
            assert da == db
            np.testing.assert_array_equal(ra, rb)


def test_plugins_answer_large_n():
    """nn(d, n) with n far above the kernels' one-workgroup select: the plugin classes answer like the reference
    would (every descriptor / code, ascending), they do not raise."""
    rng = np.random.default_rng(71)
    x = rng.standard_normal((20_000, 16)).astype(np.float32)
    bf = HipBruteForceNearestNeighborsIndex()
    bf.build_index(_elems(x))
    r, d = bf.nn(DescriptorMemoryElement("q").set_vector(x[5]), 19_000)
    assert len(r) == 19_000 and r[0].uuid() == 5 and (np.diff(d) >= 0).all()
    rd, ri = O.dense_topk(x, x[5], 19_000)
    assert [e.uuid() for e in r[:50]] == ri[:50].tolist() and [e.uuid() for e in r[-50:]] == ri[-50:].tolist()
    hv = rng.random((40_000, 40)) > 0.5
    hi = HipLinearHashIndex()
    hi.build_index(hv)
    codes, dists = hi.nn(hv[3], 25_000)
    assert len(codes) == min(25_000, hi.count()) and dists[0] == 0.0 and (np.diff(dists) >= 0).all()
    f = HipItqFunctor(bit_length=12, itq_iterations=3, random_seed=0)
    f.fit(_elems(x[:2000]))
    for metric in ("euclidean", "cosine"):
        lsh = HipLSHNearestNeighborIndex(f, MemoryDescriptorSet(), MemoryKeyValueStore(), HipLinearHashIndex(),
                                         distance_method=metric)
        lsh.build_index(_elems(x))
        r, d = lsh.nn(DescriptorMemoryElement("q").set_vector(x[9]), 20_000)      # n covers every code: everything
        assert len(r) == 20_000 and r[0].uuid() == 9 and (np.diff(d) >= 0).all()


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("metric", ["euclidean", "cosine"])
def test_lsh_query_composite_matches_oracle(dt, metric):
    """sq_lsh_query: hash -> nearest codes -> bucket expansion -> exact re-rank in one device call, against the
    oracle's restatement of lsh.py:452-519 (the same canonical code order, so rows AND distances must agree)."""
    rng = np.random.default_rng(83)
    n, d, bits = 6000, 40, 9                                    # 9 bits: ~500 buckets of ~12 rows
    db = rng.standard_normal((n, d)).astype(dt)
    db[100] = db[7]
    mean = db[:500].mean(axis=0).astype(np.float64)
    q_, _ = np.linalg.qr(rng.standard_normal((d, d)))
    rot = np.ascontiguousarray(q_[:, :bits])
    packed = _lib.itq_hash(db, mean, rot)
    uniq, inv = np.unique(packed, axis=0, return_inverse=True)
    inv = np.asarray(inv).reshape(-1)
    order = np.argsort(inv, kind="stable").astype(np.int64)
    off = np.searchsorted(inv[order], np.arange(uniq.shape[0] + 1)).astype(np.int64)
    buckets = [order[off[u]:off[u + 1]].tolist() for u in range(uniq.shape[0])]
    hidx = _lib.HammingIndex(uniq)
    rows = _lib.RowMatrix(db)
    rows.set_buckets(off, order)
    model = _lib.ItqModel(mean, rot)
    m = _lib.SQ_METRIC_L2 if metric == "euclidean" else _lib.SQ_METRIC_COSINE
    qs = rng.standard_normal((5, d)).astype(dt)
    qs[0] = db[7]
    for ncodes in (1, 7, 60, 10_000):
        k = min(ncodes, n) if ncodes < 100 else 300
        dist, got = rows.lsh_query(hidx, model, qs, ncodes, m, k)
        for qi, q in enumerate(qs):
            ids, rd = O.lsh_nn(q, ncodes, mean, rot, None, uniq, buckets, db, metric)   # the n nearest codes, top n rows
            ids, rd = ids[:k], rd[:k]
            kk = len(ids)
            np.testing.assert_array_equal(got[qi, :kk], ids)
            assert (got[qi, kk:] == -1).all()
            if metric == "euclidean":
                np.testing.assert_array_equal(dist[qi, :kk], rd.astype(dist.dtype))
            else:
                np.testing.assert_allclose(dist[qi, :kk], rd, rtol=1e-12, atol=1e-15)
    rows.close()
    hidx.close()
    model.close()
