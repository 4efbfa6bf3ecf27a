"""
One rank of the multi-rank RCCL test (tests/test_hip_pipelines.py::test_sharded_search_rccl_multirank): started
`world` times by torch.distributed.run, one process per GPU.  Every rank builds the same seeded database on the host,
keeps its contiguous row shard on its GPU (HIP index, global ids through id_base) and answers through

  * ShardedIndex.search                   -- blocking shard search, ONE packed all-gather, host merge;
  * PipelinedShardedSearch(depth 3, gather_every 4, wait=False)
                                          -- pipelined shard searches (SQ_MEM_DEVICE_ASYNC), grouped all-gathers and
                                             the merge thread, fed with TEMPORARY query tensors;

for dense L2, dense cosine and Hamming shards; rank 0 compares every merged answer with the oracle over the WHOLE
database (bit-exact ids, integer / float32 distances; cosine 1e-12).  Exit code 0 = all good on every rank.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main() -> int:
    import torch
    import torch.distributed as dist
    from oracle import cpu_ref as O
    from smqtk_indexing_amd import _lib
    from smqtk_indexing_amd.distributed import (PipelinedShardedSearch, dense_shard, hamming_shard, shard_range)

    world = int(os.environ["WORLD_SIZE"])
    rank = int(os.environ["RANK"])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group(backend="nccl", device_id=dev)
    rng = np.random.default_rng(2024)
    k = 20

    def check(tag, got, want, exact=True):
        if rank != 0:
            assert got is None, tag
            return
        d, i = got
        for j, (rd, ri) in enumerate(want):
            np.testing.assert_array_equal(i[j], ri, err_msg=f"{tag}: ids of query {j}")
            if exact:
                np.testing.assert_array_equal(d[j], rd, err_msg=f"{tag}: distances of query {j}")
            else:
                np.testing.assert_allclose(d[j], rd, rtol=1e-12, atol=1e-15, err_msg=f"{tag}: distances of query {j}")

    # ------------------------------------------------------------------ dense shards (the filter path: n_local > cap)
    n, d, nq = 160_000 * world, 64, 8
    db = rng.standard_normal((n, d)).astype(np.float32)
    db[n - 5] = db[11]                                   # a tie across the first and the last shard
    batches = [rng.standard_normal((nq, d)).astype(np.float32) for _ in range(11)]
    batches[3][0] = db[11]
    r0, r1 = shard_range(n, world, rank)
    for metric, mname in ((_lib.SQ_METRIC_L2, "euclidean"), (_lib.SQ_METRIC_COSINE, "cosine")):
        shard = dense_shard(torch.from_numpy(db[r0:r1]).to(dev), row0=r0, metric=metric)
        ddt = torch.float64 if metric == _lib.SQ_METRIC_COSINE else torch.float32
        want = [[O.dense_topk(db, q, k, metric=mname) for q in b] for b in (batches if rank == 0 else [])]
        got = shard.search(torch.from_numpy(batches[3]).to(dev), k, merge_on=0)
        check(f"{mname} packed all-gather", got, want[3] if rank == 0 else None, exact=metric == _lib.SQ_METRIC_L2)
        pipe = PipelinedShardedSearch(shard.index, nq, k, ddt, merge_on=0, device=dev, use_async=True, depth=3,
                                      gather_every=4, wait=False)
        outs = []
        for b in batches:
            r = pipe.submit(torch.from_numpy(b).to(dev))      # a temporary: the searcher keeps it alive while in flight
            torch.empty(1 << 20, device=dev).fill_(1.0)       # churn the caching allocator between submits
            if r is not None:
                outs += r
        outs += pipe.flush()
        pipe.close()
        if rank == 0:
            assert len(outs) == len(batches), (len(outs), len(batches))
            for bi, res in enumerate(outs):
                check(f"{mname} pipelined batch {bi}", res, want[bi], exact=metric == _lib.SQ_METRIC_L2)
        else:
            assert not outs
        shard.index.close()
        dist.barrier()

    # ---------------------------------------------------------------- Hamming shards (256-bit codes, scan path)
    nc, w = 90_000 * world, 4
    codes = np.unique(rng.integers(0, 2 ** 64, size=(nc, w), dtype=np.uint64), axis=0)
    nc = codes.shape[0]
    hq = [rng.integers(0, 2 ** 64, size=(5, w), dtype=np.uint64) for _ in range(7)]
    hq[2][1] = codes[nc // 2]
    c0, c1 = shard_range(nc, world, rank)
    hs = hamming_shard(torch.from_numpy(codes[c0:c1].view(np.int64)).to(dev), row0=c0)
    hwant = [[O.hamming_topk(codes, q, k) for q in b] for b in (hq if rank == 0 else [])]
    got = hs.search(torch.from_numpy(hq[2].view(np.int64)).to(dev), k, merge_on=0)
    check("hamming packed all-gather", got, hwant[2] if rank == 0 else None)
    pipe = PipelinedShardedSearch(hs.index, 5, k, torch.int32, merge_on=0, device=dev, use_async=True, depth=3,
                                  gather_every=2, wait=True)
    outs = []
    for b in hq:
        r = pipe.submit(torch.from_numpy(b.view(np.int64)).to(dev))
        if r is not None:
            outs += r
    outs += pipe.flush()
    pipe.close()
    if rank == 0:
        assert len(outs) == len(hq)
        for bi, res in enumerate(outs):
            check(f"hamming pipelined batch {bi}", res, hwant[bi])
    hs.index.close()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(f"rccl_worker: ok on {world} rank(s)", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
