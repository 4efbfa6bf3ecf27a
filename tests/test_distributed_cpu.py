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


@pytest.mark.parametrize("kind", ["dense", "hamming"])
def test_sharded_search_gloo_world2(tmp_path, kind):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, kind, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / f"ok_{kind}").exists()


def test_shard_range_covers_rows():
    from smqtk_indexing_amd.distributed import shard_range
    for n in (1, 7, 8, 10_000_000, 100_000_001):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
