"""
world_size-2 gloo test of the N>1 path on CPU: row sharding, global ids, the
all-gather of per-shard top-k lists and the host merge (sq_merge_topk).  The
per-shard search is supplied by the oracle here (no GPU in this container);
on the GPU box the same ShardedIndex wraps the HIP index (tests/test_hip_plugins.py).
"""
import os
import socket

import numpy as np
import pytest

from oracle import cpu_ref as O


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank: int, world: int, port: int, kind: str, out_dir: str) -> None:
    import torch
    import torch.distributed as dist
    from smqtk_indexing_amd.distributed import ShardedIndex, shard_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(123)
    k = 25
    if kind == "dense":
        db = rng.standard_normal((5003, 32)).astype(np.float32)
        db[4000] = db[10]                      # a tie across shards
        qs = rng.standard_normal((4, 32)).astype(np.float32)
        r0, r1 = shard_range(db.shape[0], world, rank)

        def local(queries, kk):
            d = np.full((len(queries), kk), np.inf, np.float32)
            i = np.full((len(queries), kk), -1, np.int64)
            for j, q in enumerate(queries):
                dd, ii = O.dense_topk(db[r0:r1], q, kk)
                d[j, :len(dd)], i[j, :len(ii)] = dd, ii + r0
            return torch.from_numpy(d), torch.from_numpy(i)
        full = [O.dense_topk(db, q, k) for q in qs]
    else:
        codes = np.unique(rng.integers(0, 2 ** 64, size=(4001, 2), dtype=np.uint64), axis=0)
        qs = rng.integers(0, 2 ** 64, size=(3, 2), dtype=np.uint64)
        r0, r1 = shard_range(codes.shape[0], world, rank)

        def local(queries, kk):
            d = np.full((len(queries), kk), np.iinfo(np.int32).max, np.int32)
            i = np.full((len(queries), kk), -1, np.int64)
            for j, q in enumerate(queries):
                dd, ii = O.hamming_topk(codes[r0:r1], q, kk)
                d[j, :len(dd)], i[j, :len(ii)] = dd, ii + r0
            return torch.from_numpy(d), torch.from_numpy(i)
        full = [O.hamming_topk(codes, q, k) for q in qs]
    res = ShardedIndex(local).search(qs, k, merge_on=0)
    if rank == 0:
        d, i = res
        for j, (rd, ri) in enumerate(full):
            np.testing.assert_array_equal(d[j], rd)
            np.testing.assert_array_equal(i[j], ri)
        open(os.path.join(out_dir, f"ok_{kind}"), "w").write("ok")
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


def _mutation_worker(rank: int, world: int, port: int, out_dir: str) -> None:
    """SURVEY 8e "Mutations": appends to the least-full shard, tombstones + compaction, KeyError before any
    change; after every step the sharded answer equals the oracle over the live (id -> row) set."""
    import torch
    import torch.distributed as dist
    from smqtk_indexing_amd.distributed import MutableShardedIndex, shard_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(7)
    d, k = 16, 12
    db = rng.standard_normal((301, d)).astype(np.float32)
    db[250] = db[3]                                   # a cross-shard tie
    qs = rng.standard_normal((5, d)).astype(np.float32)
    qs[0] = db[3]

    def build_local(rows):                            # oracle-backed shard searcher (no GPU here)
        mat = rows.numpy()

        def search(queries, kk):
            qn = np.asarray(queries, dtype=np.float32)
            dd = np.full((len(qn), kk), np.inf, np.float32)
            ii = np.full((len(qn), kk), -1, np.int64)
            for j, q in enumerate(qn):
                a, b = O.dense_topk(mat, q, kk)
                dd[j, :len(a)], ii[j, :len(b)] = a, b
            return torch.from_numpy(dd), torch.from_numpy(ii)
        return search

    r0, r1 = shard_range(len(db), world, rank)
    idx = MutableShardedIndex(torch.from_numpy(db[r0:r1].copy()), r0, len(db), build_local, compact_at=0.1)
    live = {i: db[i] for i in range(len(db))}

    def check():
        ids = np.array(sorted(live))
        mat = np.stack([live[i] for i in ids])
        got = idx.search(torch.from_numpy(qs), k)
        assert idx.count() == len(live)
        for j, q in enumerate(qs):
            rd, ri = O.dense_topk(mat, q, k)          # ids ascending -> canonical (distance, id) order
            np.testing.assert_array_equal(got[1][j, :len(ri)], ids[ri])
            np.testing.assert_array_equal(got[0][j, :len(rd)], rd)

    check()
    new = rng.standard_normal((40, d)).astype(np.float32)
    new[5] = db[3]                                    # ties with two existing rows, larger id
    nid = idx.append(torch.from_numpy(new))
    assert list(nid) == list(range(301, 341))
    live.update({int(i): v for i, v in zip(nid, new)})
    assert idx.live == [151, 150 + 40]                # the second shard held fewer rows: it takes the batch
    check()
    with pytest.raises(KeyError):
        idx.remove([3, 99999])                        # unknown id: nothing changes anywhere
    check()
    gone = [3, 250, 17, 300, 305] + list(range(100, 140))
    idx.remove(gone)                                  # 45 rows: the first shard passes 10 % dead and compacts
    for g in gone:
        del live[g]
    check()
    with pytest.raises(KeyError):
        idx.remove([17])                              # already removed
    nid2 = idx.append(torch.from_numpy(new[:3] + 1))
    live.update({int(i): v for i, v in zip(nid2, new[:3] + 1)})
    check()
    if rank == 0:
        open(os.path.join(out_dir, "ok_mut"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_mutations_gloo_world2(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_mutation_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok_mut").exists()


@pytest.mark.parametrize("kind", ["dense", "hamming"])
def test_sharded_search_gloo_world2(tmp_path, kind):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, kind, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / f"ok_{kind}").exists()


class _OracleSearcher:
    """Shard searcher for PipelinedShardedSearch backed by the oracle.  ``lag`` >= 1 imitates
    sq_dense_search(SQ_MEM_DEVICE_ASYNC) with ``dense_async_depth`` = lag + 1: a call's answer only appears in its
    output tensors when ``lag`` further calls (or finish()) have returned -- so the test fails if the pipeline gathers
    a send buffer too early or reuses one too soon."""

    def __init__(self, db, r0, lag):
        self.db, self.r0, self.lag = db, r0, lag
        self.pending = []                      # calls whose answers are not "final" yet, oldest first

    def _run(self, job):
        import torch
        q, k, out_d, out_i = job
        for j, qv in enumerate(q.numpy()):
            dd, ii = O.dense_topk(self.db, qv, k)
            out_d[j].fill_(float("inf"))
            out_i[j].fill_(-1)
            out_d[j, :len(dd)] = torch.from_numpy(dd)
            out_i[j, :len(ii)] = torch.from_numpy(ii + self.r0)

    def search_into(self, queries, k, out_d, out_i):
        job = (queries.clone(), k, out_d, out_i)
        if not self.lag:
            self._run(job)
            return
        out_d.fill_(-1.0)              # garbage until the call is "finished": `lag` calls later
        out_i.fill_(-7)
        self.pending.append(job)
        while len(self.pending) > self.lag:
            self._run(self.pending.pop(0))

    def finish(self):
        while self.pending:
            self._run(self.pending.pop(0))


def _pipeline_worker(rank: int, world: int, port: int, lag: int, every: int, out_dir: str) -> None:
    import torch
    import torch.distributed as dist
    from smqtk_indexing_amd.distributed import PipelinedShardedSearch, ShardedIndex, shard_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(99)
    nq, k, d = 3, 7, 24                                   # nq * k odd: the packed blocks need their 8-byte padding
    db = rng.standard_normal((2001, d)).astype(np.float32)
    db[1500] = db[20]                                      # a tie across the two shards
    batches = [rng.standard_normal((nq, d)).astype(np.float32) for _ in range(8)]
    batches[1][0] = db[20]
    r0, r1 = shard_range(db.shape[0], world, rank)
    pipe = PipelinedShardedSearch(_OracleSearcher(db[r0:r1], r0, lag), nq, k, torch.float32, merge_on=0, device="cpu",
                                  gather_every=every)
    got = []
    for i, q in enumerate(batches):
        r = pipe.submit(torch.from_numpy(q))
        if rank == 0 and every == 1:
            assert (r is None) == (i < max(2, lag + 1) + lag), (i, lag)   # buffers in rotation + the lag
        elif rank != 0:
            assert r is None
        if r is not None:
            got += r if isinstance(r, list) else [r]       # gather_every > 1: a whole group at a time
    got += pipe.flush()                                    # (8 batches in groups of 3: the last group is short)
    assert pipe.flush() == []
    if every > 1:                                          # the pipeline is reusable after a short last group
        assert pipe.submit(torch.from_numpy(batches[0])) is None
        again = pipe.flush()
        if rank == 0:
            np.testing.assert_array_equal(again[0][1], got[0][1])
    # the packed single-buffer branch of allgather_merge, on CPU tensors
    s = _OracleSearcher(db[r0:r1], r0, 0)
    od, oi = torch.empty((nq, k), dtype=torch.float32), torch.empty((nq, k), dtype=torch.int64)

    def local(queries, kk):
        s.search_into(queries, kk, od, oi)
        return od, oi
    packed = ShardedIndex(local, packed=True).search(torch.from_numpy(batches[1]), k, merge_on=0)
    if rank == 0:
        assert len(got) == len(batches)
        for q, (dd, ii) in zip(batches, got):
            for j in range(nq):
                rd, ri = O.dense_topk(db, q[j], k)        # the oracle over the WHOLE database
                np.testing.assert_array_equal(ii[j], ri)
                np.testing.assert_array_equal(dd[j].view(np.uint32), rd.view(np.uint32))
        np.testing.assert_array_equal(packed[1], got[1][1])
        np.testing.assert_array_equal(packed[0], got[1][0])
        open(os.path.join(out_dir, "ok_pipe"), "w").write("ok")
    else:
        assert got == [] and packed is None
    pipe.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("lag,every", [(0, 1), (1, 1), (2, 1), (1, 3), (2, 2), (3, 2)])
def test_pipelined_sharded_search_and_packed_gather_gloo_world2(tmp_path, lag, every):
    """Two ranks through PipelinedShardedSearch (blocking and asynchronous-style searches) and through the packed
    single-buffer branch of allgather_merge, checked against the oracle over the whole database."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_pipeline_worker, args=(2, port, lag, every, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok_pipe").exists()


def test_merge_gathered_padded_blocks_and_nan_order():
    """Host only: the strided merge over blocks padded to 8 bytes (odd nq*k, float32), and NaN distances of either
    sign rank after every number (the single-GPU contract: test_cosine_zero_vectors_give_nan_ranked_last)."""
    from smqtk_indexing_amd import _lib
    from smqtk_indexing_amd.distributed import packed_block_bytes
    nq, k = 1, 3
    per = packed_block_bytes(nq, k, 4)
    assert per == 40 and per % 8 == 0
    neg_nan = np.array([0xFFC00000], dtype=np.uint32).view(np.float32)[0]     # what x86 0/0 gives
    pos_nan = np.float32("nan")
    shards = [(np.array([0.1, 0.2, neg_nan], np.float32), np.array([5, 6, 7], np.int64)),
              (np.array([0.15, 0.3, 0.4], np.float32), np.array([10, 11, 12], np.int64)),
              (np.array([0.35, pos_nan, pos_nan], np.float32), np.array([20, 21, 22], np.int64))]
    buf = np.zeros(3 * per, dtype=np.uint8)
    for s, (dd, ii) in enumerate(shards):
        buf[s * per: s * per + 24] = ii.view(np.uint8)
        buf[s * per + 24: s * per + 36] = dd.view(np.uint8)
    od, oi = _lib.merge_topk_gathered(buf, 3, nq, k, 9, np.float32)
    assert oi[0, :6].tolist() == [5, 10, 6, 11, 20, 12]
    np.testing.assert_array_equal(od[0, :6], np.array([0.1, 0.15, 0.2, 0.3, 0.35, 0.4], np.float32))
    assert np.isnan(od[0, 6:]).all() and sorted(oi[0, 6:].tolist()) == [7, 21, 22]
    d3 = np.stack([s[0] for s in shards])[:, None, :].astype(np.float64)
    d3[0, 0, 2] = np.array([0xFFF8000000000000], dtype=np.uint64).view(np.float64)[0]
    i3 = np.stack([s[1] for s in shards])[:, None, :]
    od, oi = _lib.merge_topk(d3, i3, 4)
    assert oi[0].tolist() == [5, 10, 6, 11]                # -nan no longer displaces 0.3


def test_shard_range_covers_rows():
    from smqtk_indexing_amd.distributed import shard_range
    for n in (1, 7, 8, 10_000_000, 100_000_001):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def test_hip_searcher_owns_the_handle_options_until_closed():
    """HipSearcher writes a pipeline's depth / wait / order onto the index handle (sq_handle_set_option): a second live
    asynchronous pipeline on the same handle is refused, close() puts the library's defaults (2 / 1 / 1) back and frees the
    handle for the next pipeline; a blocking searcher touches nothing.  (Host logic only: a recording stand-in for the index.)"""
    from smqtk_indexing_amd.distributed import HipSearcher

    class FakeIndex:
        async_option_prefix = "dense_async"

        def __init__(self):
            self.options, self.synced = [], 0

        def set_option(self, name, value):
            self.options.append((name, value))

        def search_device_async(self, *a):
            pass

        def search_device(self, *a):
            pass

        def sync(self):
            self.synced += 1

    idx = FakeIndex()
    s = HipSearcher(idx, 0, use_async=True, depth=3, wait=False, queries_ready=True)
    assert s.lag == 3
    assert idx.options == [("dense_async_depth", 3), ("dense_async_wait", 0), ("dense_async_order", 0)]
    with pytest.raises(RuntimeError):
        HipSearcher(idx, 0, use_async=True)
    blocking = HipSearcher(idx, 0, use_async=False)            # no options of its own: allowed beside the pipeline
    assert blocking.lag == 0 and len(idx.options) == 3
    s.close()
    assert idx.synced == 1
    assert idx.options[3:] == [("dense_async_depth", 2), ("dense_async_wait", 1), ("dense_async_order", 1)]
    s.close()                                                   # idempotent
    assert len(idx.options) == 6
    s2 = HipSearcher(idx, 0, use_async=True, depth=9)           # the handle is free again; depth clamps to 4
    assert s2.lag == 3 and idx.options[6] == ("dense_async_depth", 4)
    s2.close()
    other = FakeIndex()                                         # pipelines on two handles do not meet
    a, b = HipSearcher(idx, 0, use_async=True), HipSearcher(other, 0, use_async=True)
    a.close(), b.close()
