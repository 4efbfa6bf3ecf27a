"""The C-ABI library loads without a GPU and exports every symbol that
include/smqtk_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from smqtk_indexing_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "smqtk_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sq_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    names = _declared()
    assert set(names) == set(_lib.EXPORTS), (names, _lib.EXPORTS)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), n


def test_library_answers_without_gpu_calls():
    lib = _lib.load()
    assert lib.sq_version() >= 100
    assert _lib.device_count() >= 0
    with pytest.raises(_lib.HipError):
        _lib.set_option("no_such_option", 1)
    assert b"unknown option" in lib.sq_last_error()
    # the options the pipelined / multi-rank path sets (include/smqtk_hip.h, sq_dense_search): known, and back to defaults
    for name, default in (("dense_async_streams", 2), ("dense_async_depth", 2), ("dense_async_wait", 1),
                          ("dense_async_order", 1), ("profile", 0),
                          # the int8 first-stage filter and its captured call graph (round 3)
                          ("dense_int8", -1), ("dense_int8_batch", 64), ("dense_graph", 1), ("dense_mid_tier", 1),
                          # the three-launch calls (round 4): dense int8 head / body / select, Hamming sample / stream / pick
                          ("dense_fused", 1), ("dense_tighten", 1), ("hamming_fused", 1), ("hamming_tighten", 1)):
        _lib.set_option(name, default)


def test_host_merge_orders_by_distance_then_id():
    d = np.array([[[1., 3., np.inf]], [[1., 2., 2.]]], dtype=np.float32)      # [shards=2][nq=1][k_in=3]
    i = np.array([[[7, 9, -1]], [[3, 4, 5]]], dtype=np.int64)      # shard lists arrive sorted by (dist, id)
    od, oi = _lib.merge_topk(d, i, 5)
    np.testing.assert_array_equal(oi[0], [3, 7, 4, 5, 9])
    np.testing.assert_array_equal(od[0], [1., 1., 2., 2., 3.])
    od, oi = _lib.merge_topk(d.astype(np.float64), i, 6)
    assert oi[0, 5] == -1 and np.isinf(od[0, 5])
    od, oi = _lib.merge_topk(np.array([[[1, 4]], [[1, 9]]], dtype=np.int32), np.array([[[2, 8]], [[1, 0]]]), 3)
    np.testing.assert_array_equal(oi[0], [1, 2, 8])


def test_host_merge_reads_the_gather_buffer_in_place():
    """sq_merge_topk_strided over the one-buffer all-gather layout equals sq_merge_topk."""
    ns, nq, k = 3, 4, 5
    rng = np.random.default_rng(0)
    for dt in (np.float32, np.float64, np.int32):
        d = np.sort((rng.random((ns, nq, k)) * 50).astype(dt), axis=2)
        i = rng.integers(0, 1000, (ns, nq, k)).astype(np.int64)
        i[1, 2, 3:] = -1                                            # padding inside a shard list
        ref = _lib.merge_topk(d, i, 7)
        es = np.dtype(dt).itemsize
        buf = np.empty(ns * nq * k * (8 + es), np.uint8)
        for s in range(ns):
            o = s * nq * k * (8 + es)
            buf[o:o + nq * k * 8] = i[s].view(np.uint8).reshape(-1)
            buf[o + nq * k * 8:o + nq * k * (8 + es)] = d[s].view(np.uint8).reshape(-1)
        got = _lib.merge_topk_gathered(buf, ns, nq, k, 7, dt)
        np.testing.assert_array_equal(ref[0], got[0])
        np.testing.assert_array_equal(ref[1], got[1])


@pytest.mark.parametrize("shards,dtype,levels", [(1, np.float32, 0), (2, np.float32, 0), (8, np.float32, 0), (8, np.int32, 9),
                                                  (8, np.float64, 4), (20, np.float32, 7), (70, np.int32, 5), (5, np.float32, 1)])
def test_host_merge_equals_a_lexsort_of_the_concatenation(shards, dtype, levels):
    """Random shard lists -- ragged (padded with id -1), tie-heavy (``levels`` distinct distances) with ids
    interleaved across shards, signed zeros, NaN distances, more shards than one merge tree takes --
    against numpy's lexsort on (distance, id).  k_out below, at and above what the shards hold."""
    rng = np.random.default_rng(shards * 31 + levels)
    nq, k = 9, 23
    d = rng.random((shards, nq, k))
    if levels:
        d = np.floor(d * levels) / levels
    d = (d * 40 - (3 if dtype != np.int32 else 0)).astype(dtype)
    if dtype != np.int32:
        d[d == 0] = -0.0 if levels else 0.0
        d[0, 0, :2] = [0.0, -0.0]
        d[-1, 1, -1] = np.nan                                        # a zero-vector cosine: ranks last
    ids = (rng.permutation(shards * nq * k).reshape(shards, nq, k) * 3).astype(np.int64)
    valid = rng.integers(0, k + 1, (shards, nq))
    valid[0] = k
    valid[-1, 2] = 0                                                 # an empty shard list
    for s in range(shards):
        for q in range(nq):
            v = valid[s, q]
            key_d = np.where(np.isnan(d[s, q, :v].astype(np.float64)), np.inf, d[s, q, :v].astype(np.float64))
            o = np.lexsort((ids[s, q, :v], key_d + 0.0))
            d[s, q, :v], ids[s, q, :v] = d[s, q, :v][o], ids[s, q, :v][o]
            ids[s, q, v:] = -1
            d[s, q, v:] = np.iinfo(np.int32).max if dtype == np.int32 else np.inf
    for k_out in (1, 17, 23, 60, shards * k + 3):
        od, oi = _lib.merge_topk(d, ids, k_out)
        for q in range(nq):
            keep = ids[:, q, :].reshape(-1) >= 0
            ad, ai = d[:, q, :].reshape(-1)[keep], ids[:, q, :].reshape(-1)[keep]
            sort_d = np.where(np.isnan(ad.astype(np.float64)), np.inf, ad.astype(np.float64)) + 0.0
            if dtype != np.int32:                                    # NaN after +inf
                sort_d = np.where(np.isnan(ad.astype(np.float64)), np.finfo(np.float64).max, sort_d)
                sort_d = np.where(np.isposinf(ad.astype(np.float64)), np.finfo(np.float64).max / 2, sort_d)
            o = np.lexsort((ai, sort_d))[:k_out]
            m = len(o)
            np.testing.assert_array_equal(oi[q, :m], ai[o])
            np.testing.assert_array_equal(od[q, :m], ad[o])
            assert (oi[q, m:] == -1).all()


def test_pipelined_merger_matches_the_direct_merge():
    """The worker-thread merger used by the multi-GPU bench: several buffers in flight, results by ticket,
    errors surfaced at result()."""
    from smqtk_indexing_amd.distributed import PipelinedMerger
    rng = np.random.default_rng(12)
    shards, nq, k = 4, 6, 9
    bufs = []
    for _ in range(5):
        d = np.sort(rng.random((shards, nq, k)).astype(np.float32), axis=2)
        i = rng.permutation(shards * nq * k).reshape(shards, nq, k).astype(np.int64)
        buf = np.empty((shards, nq * k * 12), np.uint8)
        for s_ in range(shards):
            buf[s_, :nq * k * 8] = i[s_].view(np.uint8).reshape(-1)
            buf[s_, nq * k * 8:] = d[s_].view(np.uint8).reshape(-1)
        bufs.append((buf.reshape(-1), _lib.merge_topk(d, i, k)))
    pm = PipelinedMerger()
    tickets = [pm.submit(b, shards, nq, k, k, np.float32) for b, _ in bufs]
    for t, (_, ref) in reversed(list(zip(tickets, bufs))):          # collected out of order
        od, oi = pm.result(t)
        np.testing.assert_array_equal(od, ref[0])
        np.testing.assert_array_equal(oi, ref[1])
    bad = pm.submit(np.zeros(7, np.uint8), shards, nq, k, k, np.float32)   # wrong buffer size
    with pytest.raises(ValueError):
        pm.result(bad)
    pm.close()


def test_bad_arguments_are_reported_not_crashed():
    with pytest.raises(_lib.HipError):
        _lib.merge_topk(np.zeros((1, 1, 1), np.float32), np.zeros((1, 1, 1), np.int64), 0)
    if _lib.device_count() == 0:
        with pytest.raises(_lib.HipError):
            _lib.DenseIndex(np.zeros((4, 8), np.float32))


def test_argument_validation_through_the_raw_abi():
    """Status codes and sq_last_error for calls that must fail before touching a device."""
    lib = _lib.load()
    h = ctypes.c_int64(0)
    null = None
    assert lib.sq_dense_create(null, 10, 8, 0, 0, 0, ctypes.byref(h)) != 0
    assert b"sq_dense_create" in lib.sq_last_error()
    x = np.zeros((4, 8), np.float32)
    assert lib.sq_dense_create(x.ctypes.data, 4, 8, 99, 0, 0, ctypes.byref(h)) != 0          # unknown metric
    assert b"metric" in lib.sq_last_error()
    assert lib.sq_hamming_create(x.ctypes.data, 0, 1, 0, 0, ctypes.byref(h)) != 0             # n <= 0
    assert lib.sq_hamming_search(12345, x.ctypes.data, 1, 1, x.ctypes.data, x.ctypes.data, 0, null) != 0
    assert b"unknown handle" in lib.sq_last_error()
    assert lib.sq_dense_search(12345, x.ctypes.data, 1, 1, x.ctypes.data, x.ctypes.data, 0, null) != 0
    assert lib.sq_itq_model_hash(12345, x.ctypes.data, 0, 4, x.ctypes.data, 0, null) != 0
    assert b"sq_itq_model_hash: unknown handle" in lib.sq_last_error()
    assert lib.sq_itq_model_create(x.ctypes.data, 7, x.ctypes.data, 8, 4, -1, ctypes.byref(h)) != 0   # unknown mean dtype
    assert lib.sq_rows_append(12345, x.ctypes.data, 4, 0) != 0 and lib.sq_itq_model_destroy(12345) != 0
    assert lib.sq_dense_append(12345, x.ctypes.data, 4, 0) != 0
    assert b"sq_dense_append: unknown handle" in lib.sq_last_error()
    assert lib.sq_rows_create(x.ctypes.data, 7, 4, 8, 0, ctypes.byref(h)) != 0                # unknown dtype
    assert b"dtype" in lib.sq_last_error()
    assert lib.sq_rows_rerank(12345, x.ctypes.data, 1, 0, x.ctypes.data, x.ctypes.data, 1, x.ctypes.data,
                              x.ctypes.data, null) != 0
    assert lib.sq_rows_destroy(12345) != 0 and lib.sq_dense_destroy(12345) != 0 and lib.sq_hamming_destroy(12345) != 0
    m = np.zeros(8)
    assert lib.sq_itq_hash(x.ctypes.data, 0, 4, 8, m.ctypes.data, 1, m.ctypes.data, 0, -1, x.ctypes.data, 0, null) != 0  # bits <= 0
    assert lib.sq_itq_hash(x.ctypes.data, 0, 4, 8, m.ctypes.data, 1, m.ctypes.data, 4, 7, x.ctypes.data, 0, null) != 0   # no such normalize code
    assert b"normalize" in lib.sq_last_error()
    assert lib.sq_hamming_append(12345, x.ctypes.data, 1, x.ctypes.data) != 0 and lib.sq_hamming_remove(12345, x.ctypes.data, 1) != 0
    assert b"sq_hamming_remove: unknown handle" in lib.sq_last_error()
    assert lib.sq_dense_sync(12345) != 0
    st = _lib.SqStats()
    assert lib.sq_get_stats(12345, ctypes.byref(st)) != 0
