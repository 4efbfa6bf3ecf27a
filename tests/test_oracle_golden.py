"""
Pins the oracle (oracle/cpu_ref.py) against the golden vectors produced by the
REAL reference (tests/golden/make_golden.py) and against the reference's own
known-answer tests.  CPU only.
"""
import hashlib
from math import sqrt

import numpy as np
import pytest

from oracle import cpu_ref as O
from tests.golden import inputs as GI


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ------------------------------------------------------------------ G1 bits
def test_bits_roundtrip_golden(golden):
    g = golden("g1_bits.npz")
    for b in (1, 3, 64, 65, 256):
        v = g[f"bits_{b}"]
        ints = [int(s) for s in g[f"ints_{b}"]]
        assert [O.bit_vector_to_int_large(r) for r in v] == ints
        assert [O.packed_to_int(r) for r in O.pack_bits_msb(v)] == ints
        back = np.vstack([O.int_to_bit_vector_large(i, b) for i in ints])
        np.testing.assert_array_equal(back, v)
        np.testing.assert_array_equal(O.unpack_bits_msb(O.pack_bits_msb(v), b), v)


def test_bits_known_answers(golden):
    # tests/utils/test_bits.py:10-54
    g = golden("g1_bits.npz")
    np.testing.assert_array_equal(O.int_to_bit_vector_large(0), g["kat_zero"])
    np.testing.assert_array_equal(O.int_to_bit_vector_large(1), g["kat_one"])
    np.testing.assert_array_equal(O.int_to_bit_vector_large(2 ** 256 - 1), g["kat_2p256m1"])
    np.testing.assert_array_equal(O.int_to_bit_vector_large(2 ** 512), g["kat_2p512"])
    assert g["kat_2p256m1"].all() and g["kat_2p256m1"].size == 256
    with pytest.raises(ValueError):
        O.int_to_bit_vector_large(2 ** 10, 5)


# --------------------------------------------------------------- G2 hamming
def test_hamming_pairs_golden(golden):
    g = golden("g2_hamming.npz")
    for b in (64, 256, 1024):
        a, c, d = g[f"a_{b}"], g[f"b_{b}"], g[f"d_{b}"]
        got = O.popcount_u64(a ^ c).sum(axis=1)
        np.testing.assert_array_equal(got, d)
        assert O.hamming_distance(O.packed_to_int(a[0]), O.packed_to_int(c[0])) == d[0]


# -------------------------------------------------------------- G3 ITQ hash
def test_itq_hash_golden(golden):
    g = golden("g3_itq_hash.npz")
    for tag, (n, d, bits, seed) in GI.ITQ_CASES.items():
        x32, mean, rot = GI.itq_inputs(n, d, bits, seed)
        assert sha(x32) == str(g[f"{tag}_sha_x"])
        np.testing.assert_array_equal(mean, g[f"{tag}_mean"])
        np.testing.assert_array_equal(rot, g[f"{tag}_rot"])
        for norm in (None, 2):
            for dt in (np.float32, np.float64):
                key = f"{tag}_n{norm}_{np.dtype(dt).name}"
                x = x32.astype(dt)
                # reference per-row call == oracle per-row call, bit for bit
                rows = np.vstack([O.itq_get_hash(r, mean, rot, norm) for r in x])
                np.testing.assert_array_equal(O.pack_bits_msb(rows), g[key + "_packed"])
                # batched GEMM form may only differ where |z| is at rounding level
                diff = O.pack_bits_msb(O.itq_get_hash(x, mean, rot, norm)) != g[key + "_packed"]
                if diff.any():
                    assert g[key + "_minabsz"][diff.any(axis=1)].max() < 1e-12


def test_itq_known_answers(golden):
    # tests/impls/lsh_functor/test_itq.py:304-336
    g = golden("g3_itq_hash.npz")
    mean = np.array([0., 0.])
    rot = np.array([[1. / sqrt(2)], [1. / sqrt(2)]])
    got = np.vstack([O.itq_get_hash(r, mean, rot) for r in g["kat_x"]])
    np.testing.assert_array_equal(got, g["kat_bits"])
    np.testing.assert_array_equal(got[:, 0], [True, False, True, False, True, True, False, True])


def test_itq_norm_vector():
    # tests/impls/lsh_functor/test_itq.py:74-97
    v = np.random.default_rng(0).random(16)
    assert O.itq_norm_vector(v, None) is v
    np.testing.assert_allclose(np.linalg.norm(O.itq_norm_vector(v, 2)), 1.0)


# ------------------------------------------------------ G4 LinearHashIndex.nn
def test_linear_hash_nn_golden(golden):
    g = golden("g4_linear_hash_nn.npz")
    for tag, (n, bits, seed, mode) in GI.HAMMING_CASES.items():
        codes, queries = GI.hamming_inputs(n, bits, seed, mode)
        assert sha(codes) == str(g[f"{tag}_sha_codes"])
        assert sha(queries) == str(g[f"{tag}_sha_queries"])
        lut = {O.packed_to_int(r): i for i, r in enumerate(codes)}
        for k in GI.HAMMING_KS[tag]:
            rcodes, rdist = g[f"{tag}_k{k}_codes"], g[f"{tag}_k{k}_dist"]
            for qi, q in enumerate(queries):
                d, idx = O.hamming_topk(codes, q, k)
                ref_d = np.rint(rdist[qi] * bits).astype(np.int32)
                np.testing.assert_allclose(d / float(bits), rdist[qi], rtol=0, atol=1e-15)
                ref_idx = np.array([lut[O.packed_to_int(r)] for r in rcodes[qi]])
                full = O.popcount_u64(codes ^ q[None, :]).sum(axis=1)
                O.assert_topk_equivalent(ref_d, ref_idx, d, idx, all_dist_of=lambda i: full[i])
                assert len(set(idx.tolist())) == len(idx)


def test_linear_hash_nn_known_answer():
    # tests/impls/hash_index/test_linear.py:141-155
    codes = O.pack_bits_msb(np.array([[0, 1, 0], [1, 1, 0], [0, 1, 1], [0, 0, 1]]))
    codes = np.unique(codes, axis=0)
    d, idx = O.hamming_topk(codes, O.pack_bits_msb(np.array([[0, 0, 0]]))[0], 4)
    near = [tuple(r) for r in O.unpack_bits_msb(codes[idx], 3).astype(int)]
    assert set(near[:2]) == {(0, 1, 0), (0, 0, 1)}
    assert set(near[2:]) == {(1, 1, 0), (0, 1, 1)}
    np.testing.assert_array_almost_equal(d / 3.0, (1 / 3., 1 / 3., 2 / 3., 2 / 3.))
    # reference-faithful python-int form agrees
    rows, dist = O.linear_hash_nn_reference({2, 6, 3, 1}, np.array([0, 0, 0], bool), 4)
    np.testing.assert_array_almost_equal(dist, (1 / 3., 1 / 3., 2 / 3., 2 / 3.))


# ------------------------------------------------------------- G5 dense kNN
def test_dense_nn_golden(golden):
    g = golden("g5_dense_nn.npz")
    for tag, (n, d, nq, seed, dist, dt) in GI.DENSE_CASES.items():
        db, qs = GI.dense_inputs(n, d, nq, seed, dist, dt)
        assert sha(db) == str(g[f"{tag}_sha_db"])
        assert sha(qs) == str(g[f"{tag}_sha_q"])
        for metric in ("euclidean", "cosine"):
            ridx, rdist = g[f"{tag}_{metric}_idx"], g[f"{tag}_{metric}_dist"]
            for qi in range(ridx.shape[0]):
                k = min(GI.DENSE_KMAX, n)
                gd, gi = O.dense_topk(db, qs[qi], k, metric)
                if metric == "euclidean":
                    # float32 in, float32 out, bit identical (metrics.py:73-86)
                    assert gd.dtype == np.float32
                    np.testing.assert_array_equal(gd, rdist[qi][:k])
                    np.testing.assert_array_equal(gi, ridx[qi][:k])
                else:
                    # batched cdist vs per-row cdist: float64 rounding only
                    np.testing.assert_allclose(gd, rdist[qi][:k], rtol=1e-9, atol=1e-9)
                    full = O.dense_distances(db, qs[qi], "cosine")
                    assert (np.abs(full[gi] - full[ridx[qi][:k]]) < 1e-9).all()


def test_pairwise_sum_restatement():
    rng = np.random.default_rng(5)
    # (beyond 8192 elements numpy sums its iterator's 8192-element buffers one after the other)
    for n in (1, 7, 8, 9, 100, 128, 129, 200, 256, 300, 512, 1000, 4096, 8192, 8193, 8300, 16384, 20000):
        a = np.square(rng.standard_normal(n).astype(np.float32) * 7)
        assert O.np_pairwise_sum_f32(a).tobytes() == np.sum(a).tobytes(), n


def test_dist_func_known_answers():
    # tests/impls/nn_index/test_lsh.py:102-143
    assert O.euclidean_distance(np.array([0., 0.]), np.array([0., 1.])) == 1.0
    assert O.cosine_distance(np.array([1., 0.]), np.array([0., 1.])) == pytest.approx(1.0)
    assert O.cosine_distance(np.array([1., 0.]), np.array([1., 1.])) == pytest.approx(0.5)


# ------------------------------------------------------- G6 LSH end to end
def _lsh_state(db, mean, rot):
    bits = O.itq_get_hash(db, mean, rot)
    # the reference hashes one descriptor at a time (lsh.py:317)
    bits = np.vstack([O.itq_get_hash(r, mean, rot) for r in db])
    packed = O.pack_bits_msb(bits)
    uniq, inv = np.unique(packed, axis=0, return_inverse=True)
    inv = np.asarray(inv).reshape(-1)
    rows = [[] for _ in range(uniq.shape[0])]
    for r, u in enumerate(inv.tolist()):
        rows[u].append(r)
    return uniq, rows


def test_lsh_nn_golden(golden):
    g = golden("g6_lsh_nn.npz")
    for tag, (n, d, bits, seed, metric, ns) in GI.LSH_CASES.items():
        db, qs = GI.lsh_inputs(n, d, seed)
        assert sha(db) == str(g[f"{tag}_sha_db"])
        mean, rot = g[f"{tag}_mean"], g[f"{tag}_rot"]
        uniq, rows = _lsh_state(db, mean, rot)
        assert uniq.shape[0] == int(g[f"{tag}_ncodes"])
        assert sum(map(len, rows)) == int(g[f"{tag}_count"]) == n
        for nn in ns:
            ru, rd = g[f"{tag}_n{nn}_uuids"], g[f"{tag}_n{nn}_dist"]
            for qi, q in enumerate(qs):
                ids, dist = O.lsh_nn(q, nn, mean, rot, None, uniq, rows, db, metric)
                valid = ru[qi] >= 0
                r_ids, r_dist = ru[qi][valid], rd[qi][valid]
                _check_lsh_admissible(q, nn, mean, rot, uniq, rows, db, metric, r_ids, r_dist)
                if nn >= n:
                    # n covers every code: exact brute force, order fully determined
                    np.testing.assert_allclose(dist, r_dist, rtol=1e-12)
                    if (dist[1:] != dist[:-1]).all():
                        np.testing.assert_array_equal(ids, r_ids)


def _check_lsh_admissible(q, nn, mean, rot, uniq, rows, db, metric, r_ids, r_dist):
    """Which codes enter at the rank-n Hamming tie group is set-order dependent
    in the reference (linear.py:235-238), so the reference result R is checked
    for admissibility against the oracle's building blocks: every member of R
    lies in a bucket within the n-th smallest code distance t*; every row of a
    bucket strictly inside t* that is closer than R's farthest member is in R;
    R's distances are the oracle's exact distances, ascending."""
    qp = O.pack_bits_msb(O.itq_get_hash(q, mean, rot)[None, :])[0]
    hd = O.popcount_u64(uniq ^ qp[None, :]).sum(axis=1)
    t_star = np.sort(hd)[min(nn, len(hd)) - 1]
    fn = O.euclidean_distance if metric == "euclidean" else O.cosine_distance
    exact = {i: float(fn(q, db[i])) for u in np.nonzero(hd <= t_star)[0] for i in rows[u]}
    inner = {i for u in np.nonzero(hd < t_star)[0] for i in rows[u]}
    assert set(r_ids.tolist()) <= set(exact), "reference returned a row outside the admissible pool"
    np.testing.assert_allclose([exact[i] for i in r_ids.tolist()], r_dist, rtol=1e-12)
    assert (np.diff(r_dist) >= 0).all()
    if len(r_ids) == nn:
        far = r_dist[-1]
        missing = [i for i in inner if exact[i] < far and i not in set(r_ids.tolist())]
        assert not missing, missing


# -------------------------------------------------------------- G7 ITQ fit
def test_itq_fit_golden(golden):
    g = golden("g7_itq_fit.npz")
    x = np.array([[-2. + i, -2. + i] for i in range(5)])
    mean, rot, codes = O.itq_fit(x, 1, 50, None, 0)
    # tests/impls/lsh_functor/test_itq.py:255-270
    np.testing.assert_array_almost_equal(mean, [0, 0])
    np.testing.assert_array_almost_equal(rot, [[1 / sqrt(2)], [1 / sqrt(2)]])
    np.testing.assert_array_almost_equal(mean, g["kat_mean"])
    np.testing.assert_array_almost_equal(rot, g["kat_rot"])
    np.testing.assert_array_equal(codes, g["kat_codes"])
    x, _ = GI.lsh_inputs(400, 32, 7)
    mean, rot, codes = O.itq_fit(x, 16, 10, 2, 3)
    np.testing.assert_allclose(mean, g["r_mean"], rtol=1e-12)
    np.testing.assert_allclose(rot, g["r_rot"], rtol=1e-9, atol=1e-12)
    np.testing.assert_array_equal(codes, g["r_codes"])


def test_scipy_cosine_order():
    """Pins the summation order of the scipy build in this image that the HIP
    cosine kernel mirrors: two interleaved accumulators (even / odd elements),
    summed, then the odd tail (float32-valued inputs: products are exact)."""
    import math
    from scipy.spatial.distance import cdist

    def dot2(u, v):
        a0 = a1 = 0.0
        m = len(u) - (len(u) & 1)
        for i in range(0, m, 2):
            a0 += u[i] * v[i]
            a1 += u[i + 1] * v[i + 1]
        s = a0 + a1
        if len(u) & 1:
            s += u[m] * v[m]
        return s

    rng = np.random.default_rng(1)
    for t in range(120):
        d = int(rng.integers(1, 200))
        u = rng.standard_normal(d).astype(np.float32).astype(np.float64)
        v = u.copy() if t % 7 == 0 else rng.standard_normal(d).astype(np.float32).astype(np.float64)
        ul, vl = u.tolist(), v.tolist()
        c = dot2(ul, vl) / (math.sqrt(dot2(ul, ul)) * math.sqrt(dot2(vl, vl)))
        if abs(c) > 1:
            c = math.copysign(1, c)
        assert cdist(u[None], v[None], "cosine")[0, 0] == 1.0 - c


# --------------------------- G5b dense kNN at SURVEY 8(c)'s size (20 k x 128, float32 AND float64, 32 queries)
@pytest.mark.parametrize("tag", list(GI.DENSE_BIG_CASES))
def test_dense_nn_golden_20k(golden, tag):
    g = golden("g5b_dense_nn_20k.npz")
    n, d, nq, seed, dist, dt, nq_cos = GI.DENSE_BIG_CASES[tag]
    db, qs = GI.dense_inputs(n, d, nq, seed, dist, dt)
    assert sha(db) == str(g[f"{tag}_sha_db"]) and sha(qs) == str(g[f"{tag}_sha_q"])
    ridx, rdist = g[f"{tag}_euclidean_idx"], g[f"{tag}_euclidean_dist"]
    assert ridx.shape == (32, 100) and rdist.dtype == np.dtype(dt)          # metrics.py:73-86 keeps the dtype
    for qi in range(nq):
        gd, gi = O.dense_topk(db, qs[qi], 100, "euclidean")
        assert gd.dtype == np.dtype(dt)
        np.testing.assert_array_equal(gd, rdist[qi])                         # bit identical, float32 and float64
        np.testing.assert_array_equal(gi, ridx[qi])
        for k in (1, 10):                                                    # k = 1, 10 are prefixes of the same sort
            np.testing.assert_array_equal(O.dense_topk(db, qs[qi], k, "euclidean")[1], ridx[qi][:k])
    ridx, rdist = g[f"{tag}_cosine_idx"], g[f"{tag}_cosine_dist"]
    for qi in range(nq_cos):
        gd, gi = O.dense_topk(db, qs[qi], 100, "cosine")
        np.testing.assert_allclose(gd, rdist[qi], rtol=1e-9, atol=1e-9)      # batched vs per-row cdist
        full = O.dense_distances(db, qs[qi], "cosine")
        assert (np.abs(full[gi] - full[ridx[qi]]) < 1e-9).all()


# ----------------- G6b the three TestLshIndexAlgorithms scenarios (test_lsh.py:754-979), reference outputs
def lsh_scenarios(g, hi_tag):
    """(name, uuids in row order, rows, model, [(query name, vector, n)]) of the three scenarios."""
    db, near0 = GI.lsh_scenario_random_euclidean()
    pre = f"rand_{hi_tag}_"
    qrand = g[pre + "qrand"]
    yield ("rand", list(range(1000)), db, (g[pre + "mean"], g[pre + "rot"]),
           [("self255_n1", db[255], 1), ("near0_n1", near0, 1), ("rand_n10", qrand, 10), ("rand_n1000", qrand, 1000)])
    pre = f"unit_{hi_tag}_"
    yield ("unit", list(range(5)), np.eye(5), (g[pre + "mean"], g[pre + "rot"]),
           [("zero_n5", np.zeros(5), 5), ("e3_n1", np.eye(5)[3], 1), ("e3_n5", np.eye(5)[3], 5)])
    uu, rows = GI.lsh_scenario_known_ordered()
    pre = f"ord_{hi_tag}_"
    yield ("ord", uu, rows, (g[pre + "mean"], g[pre + "rot"]),
           [("origin_n5", np.zeros(2), 5), ("origin_n1000", np.zeros(2), 1000)])


@pytest.mark.parametrize("hi_tag", ["none", "linear"])
def test_lsh_reference_scenarios_golden(golden, hi_tag):
    g = golden("g6b_lsh_scenarios.npz")
    for name, uuids, rows, (mean, rot), queries in lsh_scenarios(g, hi_tag):
        uniq, buckets = _lsh_state(rows, mean, rot)
        ua = np.asarray(uuids)
        row_of = {u: r for r, u in enumerate(uuids)}
        for qname, qv, nn in queries:
            r_uu, r_dist = g[f"{name}_{hi_tag}_{qname}_uuids"], g[f"{name}_{hi_tag}_{qname}_dist"]
            ids, dist = O.lsh_nn(qv, nn, mean, rot, None, uniq, buckets, rows, "euclidean")
            r_rows = np.array([row_of[int(u)] for u in r_uu])
            _check_lsh_admissible(qv, nn, mean, rot, uniq, buckets, rows, "euclidean", r_rows, r_dist)
            _check_lsh_admissible(qv, nn, mean, rot, uniq, buckets, rows, "euclidean", ids, dist)
            if nn >= uniq.shape[0]:
                # n covers every code: exact brute force over all rows, order fully determined up to distance ties
                np.testing.assert_allclose(dist, r_dist, rtol=1e-12)
                if (dist[1:] != dist[:-1]).all():
                    np.testing.assert_array_equal(ua[ids], r_uu)
        if name == "rand":
            assert uniq.shape[0] == int(g[f"rand_{hi_tag}_ncodes"]) and len(rows) == int(g[f"rand_{hi_tag}_count"])
    # the scenario assertions themselves (test_lsh.py:783-813, 856-876, 953-966)
    assert g[f"rand_{hi_tag}_self255_n1_uuids"][0] == 255 and g[f"rand_{hi_tag}_near0_n1_uuids"][0] == 0
    assert (np.diff(g[f"rand_{hi_tag}_rand_n1000_dist"]) > 0).all()
    assert (g[f"unit_{hi_tag}_zero_n5_dist"] == 1.0).all() and g[f"unit_{hi_tag}_e3_n5_dist"][0] == 0.0
    assert g[f"ord_{hi_tag}_origin_n1000_uuids"].tolist() == list(range(1000))


# --------------- G8 cache bytes written by the reference (itq.py:222-237 save_model, linear.py:133-142 save_cache)
def test_reference_written_caches_load(golden):
    """Host side only (no kernel call): HipItqFunctor / HipLinearHashIndex read the bytes the reference wrote."""
    from smqtk_indexing_amd._compat import DataMemoryElement
    from smqtk_indexing_amd.impls.hash_index.hip_linear import HipLinearHashIndex
    from smqtk_indexing_amd.impls.lsh_functor.hip_itq import HipItqFunctor
    g = golden("g8_reference_caches.npz")
    for tag in ("float64", "float32"):
        f = HipItqFunctor(mean_vec_cache=DataMemoryElement(g[f"itq_{tag}_mean_bytes"].tobytes()),
                          rotation_cache=DataMemoryElement(g[f"itq_{tag}_rot_bytes"].tobytes()), bit_length=12)
        assert f.has_model() and f.mean_vec.dtype == np.dtype(tag)
        np.testing.assert_array_equal(f.mean_vec, g[f"itq_{tag}_mean"])
        np.testing.assert_array_equal(np.real(f.rotation), g[f"itq_{tag}_rot"])
        x, _ = GI.lsh_inputs(300, 24, 8)
        probe = x[:40].astype(tag)
        got = np.vstack([O.itq_get_hash(r, f.mean_vec, np.real(f.rotation)) for r in probe])
        np.testing.assert_array_equal(got, g[f"itq_{tag}_probe_codes"])
        # and the bytes this build writes are the bytes the reference wrote (numpy.save of the same arrays)
        m2, r2 = DataMemoryElement(), DataMemoryElement()
        f.mean_vec_cache_elem, f.rotation_cache_elem = m2, r2
        f.save_model()
        assert m2.get_bytes() == g[f"itq_{tag}_mean_bytes"].tobytes()
    for tag in ("b20", "b62"):
        assert str(g[f"lin_{tag}_cache_dtype"]) == "int64"
        idx = HipLinearHashIndex(cache_element=DataMemoryElement(g[f"lin_{tag}_cache_bytes"].tobytes()))
        np.testing.assert_array_equal(idx.codes_packed(), g[f"lin_{tag}_codes"])
        assert idx.count() == g[f"lin_{tag}_codes"].shape[0]
        bits = int(tag[1:])
        rows, dist = O.linear_hash_nn_reference(set(O.packed_to_int(c) for c in idx.codes_packed()), g[f"lin_{tag}_q"], 7)
        np.testing.assert_allclose(dist, g[f"lin_{tag}_nn_dist"])
        assert bits == len(g[f"lin_{tag}_q"])
    # codes at and above 2**63 next to smaller ones: numpy.save(tuple(ints)) stores FLOAT64 (lossy upstream, SURVEY 8f
    # rank 2); that cache is refused rather than loaded as corrupted codes
    assert str(g["lin_b64_cache_dtype"]) == "float64"
    with pytest.raises(ValueError):
        HipLinearHashIndex(cache_element=DataMemoryElement(g["lin_b64_cache_bytes"].tobytes()))
