"""
GPU tests of the call pipelines around the kernels (round 3): the Hamming LDS-DMA ring kernel and the asynchronous
Hamming search, per-handle options (two indexes searched from two threads), the lifetime of query tensors handed to
the sharded pipeline, and the multi-rank RCCL path (as many ranks as the box has GPUs, at most 4).  Everything is
checked against the oracle; the HIP path is called through the C ABI.
"""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from oracle import cpu_ref as O
from smqtk_indexing_amd import _lib

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dev():
    return torch.device("cuda", 0)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _codes(rng, n, w):
    return np.unique(rng.integers(0, 2 ** 64, size=(n, w), dtype=np.uint64), axis=0)


# ----------------------------------------------------------------------------------------- Hamming ring kernel
@pytest.mark.parametrize("w,n,nq,k", [(1, 300_001, 5, 100), (2, 200_000, 33, 10), (4, 262_144, 3, 100), (4, 150_017, 70, 7),
                                      (8, 100_003, 4, 50), (16, 70_001, 2, 20)])
def test_hamming_ring_kernel_matches_oracle(w, n, nq, k):
    """hamming_ring_kernel (LDS-DMA ring, whole codes per lane, physical rows mapped by the compaction): every code
    width it covers, arrays that end inside a unit (guarded tail), batches beyond one launch's 64 queries, the
    arithmetic row permutation and -- after an append -- the explicit rank table."""
    rng = np.random.default_rng(1000 * w + nq)
    codes = _codes(rng, n, w)
    queries = rng.integers(0, 2 ** 64, size=(nq, w), dtype=np.uint64)
    queries[0] = codes[len(codes) // 3]
    queries[-1] = codes[-1]                          # lives in the tail unit
    idx = _lib.HammingIndex(codes)
    idx.set_option("hamming_ring", 1)
    d, i = idx.search(queries, k)
    assert idx.stats()["fallback_queries"] == 0
    for qi, q in enumerate(queries):
        rd, ri = O.hamming_topk(codes, q, k)
        np.testing.assert_array_equal(d[qi], rd)
        np.testing.assert_array_equal(i[qi], ri)
    # the register kernel (or, for widths it does not cover, the atomic scan) answers identically
    idx.set_option("hamming_ring", 0)
    d0, i0 = idx.search(queries, k)
    np.testing.assert_array_equal(d0, d)
    np.testing.assert_array_equal(i0, i)
    # explicit ranks: append a few codes, search again through the ring
    new = _codes(rng, 300, w)
    new = new[~(new[:, None, :] == codes[None, :1, :]).all(axis=2).any(axis=1)]
    merged = np.unique(np.concatenate([codes, new]), axis=0)
    if merged.shape[0] == codes.shape[0] + new.shape[0]:
        keys_old = [tuple(r) for r in codes.tolist()]
        import bisect
        pos = np.array([bisect.bisect_left(keys_old, tuple(r)) for r in new.tolist()], dtype=np.int64)
        idx.append(new, pos)
        idx.set_option("hamming_ring", 1)
        d2, i2 = idx.search(queries[:3], k)
        for qi in range(min(3, nq)):
            rd, ri = O.hamming_topk(merged, queries[qi], k)
            np.testing.assert_array_equal(d2[qi], rd)
            np.testing.assert_array_equal(i2[qi], ri)
    idx.close()


# ------------------------------------------------------------------- Hamming small batches in three launches
@pytest.mark.parametrize("w,n,nq,k,ring", [(1, 300_001, 1, 100, 0), (1, 1_000_003, 32, 100, 0), (2, 200_000, 17, 10, 0),
                                           (4, 150_017, 8, 1000, 0), (1, 400_000, 5, 2048, 1), (2, 262_144, 3, 1, 1),
                                           (1, 500_009, 100, 50, 0), (2, 300_000, 333, 10, 0), (4, 200_000, 70, 100, 0)])
def test_hamming_fused_small_batch_matches_oracle(w, n, nq, k, ring):
    """Head / stream / pick (sq_hamming_fused.hpp) against the oracle and against the general chain on the same index:
    distances and rows identical, for the register stream and the ring (physical rows mapped by the pick kernel), for
    a query that is a stored code, k = 1 and k at the fused path's limit, and repeated calls (the head kernel leaves
    its histogram clean for the next one)."""
    rng = np.random.default_rng(31 * w + nq + k)
    codes = _codes(rng, n, w)
    queries = rng.integers(0, 2 ** 64, size=(nq, w), dtype=np.uint64)
    queries[0] = codes[len(codes) // 3]
    idx = _lib.HammingIndex(codes, id_base=7)
    idx.set_option("hamming_ring", ring)
    idx.set_option("hamming_fused", 1)
    for rep in range(3):
        d, i = idx.search(queries, k)
        assert idx.stats()["fallback_queries"] == 0
        if rep == 0:
            for qi, q in enumerate(queries):
                rd, ri = O.hamming_topk(codes, q, k)
                np.testing.assert_array_equal(d[qi], rd)
                np.testing.assert_array_equal(i[qi], ri + 7)
            d0, i0 = d, i
        else:
            np.testing.assert_array_equal(d, d0)
            np.testing.assert_array_equal(i, i0)
    cands_fused = idx.stats()["candidates"]
    # the safe rank-k threshold instead of the tightened one: more candidates, the same answer
    idx.set_option("hamming_tighten", 0)
    d1, i1 = idx.search(queries, k)
    np.testing.assert_array_equal(d1, d0)
    np.testing.assert_array_equal(i1, i0)
    assert idx.stats()["candidates"] >= cands_fused
    cands_safe = idx.stats()["candidates"]
    # a bet that is lost (threshold = the smallest distance of the sample: fewer than k codes pass for most queries): the
    # pick kernel sees it and the call is redone by the general chain
    idx.set_option("hamming_tighten", 2)
    d2, i2 = idx.search(queries, k)
    np.testing.assert_array_equal(d2, d0)
    np.testing.assert_array_equal(i2, i0)
    if k > 1:
        assert idx.stats()["fallback_queries"] == nq
    idx.set_option("hamming_tighten", 1)
    idx.set_option("hamming_fused", 0)
    d3, i3 = idx.search(queries, k)
    np.testing.assert_array_equal(d3, d0)
    np.testing.assert_array_equal(i3, i0)
    assert idx.stats()["candidates"] == cands_safe      # the general chain: the safe threshold's mini-lists
    idx.close()


def test_hamming_fused_tie_group_beyond_the_sort_buffer():
    """41 664 codes at distance exactly 3 from the query (every 64-bit word with three bits set) among 150 k random ones:
    the pick kernel's sort buffer (4096 keys) cannot hold the tie group at the k-th distance, flags the query, and
    the general compaction + select answers from the same mini-lists -- the lowest rows of the tie group, no exact path."""
    rng = np.random.default_rng(8)
    bits3 = []
    for a in range(64):
        for b in range(a + 1, 64):
            for c in range(b + 1, 64):
                bits3.append((1 << a) | (1 << b) | (1 << c))
    codes = np.unique(np.concatenate([np.array(bits3, dtype=np.uint64),
                                      rng.integers(0, 2 ** 64, size=150_000, dtype=np.uint64)]))[:, None]
    queries = np.zeros((2, 1), dtype=np.uint64)
    queries[1, 0] = np.uint64(0xffff)
    idx = _lib.HammingIndex(codes)
    d, i = idx.search(queries, 100)
    assert idx.stats()["fallback_queries"] == 0
    for qi, q in enumerate(queries):
        rd, ri = O.hamming_topk(codes, q, 100)
        np.testing.assert_array_equal(d[qi], rd)
        np.testing.assert_array_equal(i[qi], ri)
    assert (d[0] == 3).all()
    idx.close()


@pytest.mark.parametrize("depth,wait,ring", [(2, 1, 0), (3, 1, 1), (3, 0, 1), (4, 0, 0)])
def test_hamming_async_calls_equal_blocking_calls(depth, wait, ring):
    """sq_hamming_search with SQ_MEM_DEVICE_ASYNC: `depth` calls in flight on the slots' own streams; call i is final when
    call i + depth - 1 returns (one later with hamming_async_wait = 0), including calls whose queries overflow their
    candidate lists (exact path at resolve time), calls of other batch sizes and the calls in flight at sync / destroy."""
    rng = np.random.default_rng(97 + depth)
    dev = _dev()
    codes_h = _codes(rng, 400_000, 2)
    codes = torch.from_numpy(codes_h.view(np.int64)).to(dev)
    idx = _lib.HammingIndex(codes.data_ptr(), n=codes_h.shape[0], words=2, device_ptr=True, id_base=5, keepalive=codes)
    idx.set_option("hamming_ring", ring)
    k = 30
    sizes = [4, 40, 4, 1, 4, 70, 4, 4]
    qs = [rng.integers(0, 2 ** 64, size=(b, 2), dtype=np.uint64) for b in sizes]
    qs[2][0] = codes_h[123]
    want = [idx.search(q, k) for q in qs]
    qd = [torch.from_numpy(q.view(np.int64)).to(dev) for q in qs]
    od = [torch.empty((b, k), dtype=torch.int32, device=dev) for b in sizes]
    oi = [torch.empty((b, k), dtype=torch.int64, device=dev) for b in sizes]
    idx.set_option("hamming_async_depth", depth)
    idx.set_option("hamming_async_wait", wait)
    for j, q in enumerate(qd):
        if j == 4:
            idx.set_option("force_fallback", 1)        # calls enqueued from here on take the exact path when finished
        idx.search_device_async(q.data_ptr(), sizes[j], k, od[j].data_ptr(), oi[j].data_ptr(), _stream())
        f = j - (depth - 1) - (0 if wait else 1)
        if f >= 0:
            np.testing.assert_array_equal(oi[f].cpu().numpy(), want[f][1])
            np.testing.assert_array_equal(od[f].cpu().numpy(), want[f][0])
    idx.set_option("force_fallback", 0)
    idx.sync()
    for f in range(len(sizes)):
        np.testing.assert_array_equal(oi[f].cpu().numpy(), want[f][1])
        np.testing.assert_array_equal(od[f].cpu().numpy(), want[f][0])
    # a blocking call after asynchronous ones; destroy with calls in flight
    d2, i2 = idx.search(qs[1], k)
    np.testing.assert_array_equal(i2, want[1][1])
    for j in range(2):
        idx.search_device_async(qd[j].data_ptr(), sizes[j], k, od[j].data_ptr(), oi[j].data_ptr(), _stream())
    idx.close()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(oi[1].cpu().numpy(), want[1][1])


# ------------------------------------------------------------------------------------------ per-handle options
def test_options_are_per_handle_two_threads():
    """sq_handle_set_option: two dense indexes searched from two threads with different pipeline depths, candidate
    caps and forced exact paths do not disturb each other (the reference's contract: implementations are thread
    safe, interfaces/nearest_neighbor_index.py:22-23), and the process-wide options stay untouched."""
    rng = np.random.default_rng(7)
    dev = _dev()
    n, d, k = 150_000, 64, 10
    dbs = [rng.standard_normal((n, d)).astype(np.float32) for _ in range(2)]
    dts = [torch.from_numpy(x).to(dev) for x in dbs]
    idx = [_lib.DenseIndex(t.data_ptr(), n=n, d=d, device_ptr=True, keepalive=t) for t in dts]
    idx[0].set_option("dense_async_depth", 2)
    idx[1].set_option("dense_async_depth", 4)
    idx[1].set_option("dense_async_wait", 0)
    idx[1].set_option("force_fallback", 1)               # index 1: every query through the exact path
    idx[0].set_option("candidate_cap", 1 << 17)
    qs = [[rng.standard_normal((6, d)).astype(np.float32) for _ in range(12)] for _ in range(2)]
    want = [[[O.dense_topk(dbs[t], q, k) for q in b] for b in qs[t]] for t in range(2)]
    errors = []

    def run(t):
        try:
            torch.cuda.set_device(0)
            depth = 2 if t == 0 else 4
            lag = depth - 1 if t == 0 else depth
            st = torch.cuda.Stream(device=dev)
            qd = [torch.from_numpy(b).to(dev) for b in qs[t]]
            od = [torch.empty((6, k), dtype=torch.float32, device=dev) for _ in qd]
            oi = [torch.empty((6, k), dtype=torch.int64, device=dev) for _ in qd]
            torch.cuda.synchronize()
            for j, q in enumerate(qd):
                idx[t].search_device_async(q.data_ptr(), 6, k, od[j].data_ptr(), oi[j].data_ptr(), st.cuda_stream)
                f = j - lag
                if f >= 0:
                    got_i, got_d = oi[f].cpu().numpy(), od[f].cpu().numpy()
                    for r, (rd, ri) in enumerate(want[t][f]):
                        np.testing.assert_array_equal(got_i[r], ri)
                        np.testing.assert_array_equal(got_d[r].view(np.uint32), rd.view(np.uint32))
                    fb = idx[t].stats()["fallback_queries"]
                    assert fb == (0 if t == 0 else 6), (t, fb)
            idx[t].sync()
            for f in range(len(qd)):
                got_i = oi[f].cpu().numpy()
                for r, (rd, ri) in enumerate(want[t][f]):
                    np.testing.assert_array_equal(got_i[r], ri)
        except Exception as ex:  # noqa: BLE001 -- reported by the main thread
            errors.append((t, repr(ex)))

    threads = [threading.Thread(target=run, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    # nothing process-wide changed: a third index runs with the defaults (no exact path)
    third = _lib.DenseIndex(dts[0].data_ptr(), n=n, d=d, device_ptr=True, keepalive=dts[0])
    third.search(qs[0][0], k)
    assert third.stats()["fallback_queries"] == 0
    idx[1].reset_options()
    idx[1].search(qs[1][0], k)
    assert idx[1].stats()["fallback_queries"] == 0
    for h in idx + [third]:
        h.close()


def test_any_k_select_scratch_is_per_call():
    """k beyond the one-workgroup select on a small index (every row a candidate: the sorted select runs inside the
    enqueued call): two asynchronous calls in flight, and a second handle on another thread, each sort in their own
    scratch (it was one buffer per device)."""
    rng = np.random.default_rng(11)
    dev = _dev()
    n, d, k = 40_000, 32, 20_000
    dbh = rng.standard_normal((n, d)).astype(np.float32)
    db = torch.from_numpy(dbh).to(dev)
    idx = _lib.DenseIndex(db.data_ptr(), n=n, d=d, device_ptr=True, keepalive=db)
    qs = [rng.standard_normal((2, d)).astype(np.float32) for _ in range(4)]
    qd = [torch.from_numpy(q).to(dev) for q in qs]
    od = [torch.empty((2, k), dtype=torch.float32, device=dev) for _ in qs]
    oi = [torch.empty((2, k), dtype=torch.int64, device=dev) for _ in qs]
    for j in range(4):
        idx.search_device_async(qd[j].data_ptr(), 2, k, od[j].data_ptr(), oi[j].data_ptr(), _stream())
    idx.sync()
    for j in range(4):
        for r in range(2):
            rd, ri = O.dense_topk(dbh, qs[j][r], k)
            np.testing.assert_array_equal(oi[j][r].cpu().numpy(), ri)
            np.testing.assert_array_equal(od[j][r].cpu().numpy().view(np.uint32), rd.view(np.uint32))
    idx.close()


# ------------------------------------------------------------------------------ query lifetime in the pipeline
def test_pipeline_keeps_temporary_queries_alive():
    """HipSearcher.search_into(temporary) -- what PipelinedShardedSearch.submit calls: the asynchronous searches read
    their queries on the library's internal streams until the call is final, `lag` submits later, which torch's
    caching allocator does not know; the searcher holds the reference.  The temporaries are dropped at once and their
    blocks overwritten between submits (tests/rccl_worker.py drives the whole pipeline the same way)."""
    from smqtk_indexing_amd.distributed import HipSearcher
    rng = np.random.default_rng(5)
    dev = _dev()
    n, d, k, nq = 300_000, 128, 10, 16
    dbh = rng.standard_normal((n, d)).astype(np.float32)
    db = torch.from_numpy(dbh).to(dev)
    index = _lib.DenseIndex(db.data_ptr(), n=n, d=d, device_ptr=True, keepalive=db)
    batches = [rng.standard_normal((nq, d)).astype(np.float32) for _ in range(9)]
    s = HipSearcher(index, _stream(), use_async=True, depth=3, wait=False)
    assert s.lag == 3
    outs_d = [torch.empty((nq, k), dtype=torch.float32, device=dev) for _ in batches]
    outs_i = [torch.empty((nq, k), dtype=torch.int64, device=dev) for _ in batches]
    for j, b in enumerate(batches):
        s.search_into(torch.from_numpy(b).to(dev), k, outs_d[j], outs_i[j])      # the only reference is the searcher's
        junk = torch.empty((nq, d), dtype=torch.float32, device=dev).fill_(float(j))   # would reuse a freed block
        del junk
    s.finish()
    for j, b in enumerate(batches):
        got_i = outs_i[j].cpu().numpy()
        for r in (0, nq - 1):
            rd, ri = O.dense_topk(dbh, b[r], k)
            np.testing.assert_array_equal(got_i[r], ri)
    # the pipeline's depth / wait / order sit on the handle: a second live pipeline on it is refused, close() takes
    # them off again (a direct asynchronous caller then finds the header's contract: final when the NEXT call returns)
    with pytest.raises(RuntimeError):
        HipSearcher(index, _stream(), use_async=True, depth=2)
    s.close()
    q0 = torch.from_numpy(batches[0]).to(dev)
    index.search_device_async(q0.data_ptr(), nq, k, outs_d[1].data_ptr(), outs_i[1].data_ptr(), _stream())
    index.search_device_async(q0.data_ptr(), nq, k, outs_d[2].data_ptr(), outs_i[2].data_ptr(), _stream())
    np.testing.assert_array_equal(outs_i[1].cpu().numpy(), outs_i[0].cpu().numpy())    # depth 2, wait 1: call 1 is final
    index.sync()
    s2 = HipSearcher(index, _stream(), use_async=True, depth=2)
    s2.close()
    index.close()


# ------------------------------------------------------------------------------------------- multi-rank RCCL
def test_sharded_search_rccl_multirank():
    """One process per GPU over RCCL (min(device_count, 4) ranks; one rank on a one-GPU box, still through
    torch.distributed.run): dense L2 / cosine and Hamming shards, the packed all-gather and the pipelined search
    (depth 3, grouped gathers, wait=False) against the oracle over the whole database (tests/rccl_worker.py)."""
    ranks = max(1, min(torch.cuda.device_count(), 4))
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["OMP_NUM_THREADS"] = "2"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "rccl_worker.py")]
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    text = p.stdout.decode("utf-8", "replace")
    assert p.returncode == 0 and f"ok on {ranks} rank(s)" in text, text[-4000:]
