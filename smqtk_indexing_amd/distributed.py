"""
Row-sharded multi-GPU search: one process per GPU (``torch.distributed``,
backend ``nccl`` = RCCL over xGMI), contiguous row shards, one all-gather of the
per-shard top-k ``(distance, id)`` lists per query batch and a host-side k-way
merge (BASELINE.json north_star).  The database never
moves; the exchange is ``12 * nq * k`` bytes per rank for L2 (float32 distance
+ int64 id), latency bound.

The reference is single-process (no collective call site exists in it), so
there is no reference interface to mirror here; the per-shard search is the C
ABI (``sq_dense_search`` / ``sq_hamming_search`` with ``id_base`` = first row
of the shard) and the merge is ``sq_merge_topk``.
"""
from typing import Callable, Optional, Tuple

import numpy as np


def shard_range(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Rows ``[r0, r1)`` of rank ``rank``: contiguous blocks of ``ceil(n/world)``."""
    per = (int(n_total) + world - 1) // world
    r0 = min(rank * per, n_total)
    return r0, min(r0 + per, n_total)


def allgather_merge(local_dist, local_idx, k: int, group=None, merge_on: Optional[int] = None):
    """All-gather every rank's ``[nq, k_in]`` top-k lists and merge them.

    ``local_dist`` / ``local_idx`` are torch tensors on the rank's device (CUDA
    tensors go over RCCL, CPU tensors over gloo).  Ids must already be global.
    Returns ``(dist [nq,k], idx [nq,k])`` numpy arrays on every rank, or only on
    rank ``merge_on`` (others get ``None``) when given.
    """
    import torch
    import torch.distributed as dist
    from . import _lib

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    nq, k_in = int(local_dist.shape[0]), int(local_dist.shape[1])
    if local_dist.is_cuda:
        # one collective: [ids int64 nq*k_in][dist nq*k_in] per rank in a byte buffer; the merge reads the
        # host copy of the receive buffer in place (sq_merge_topk_strided)
        esz = local_dist.element_size()
        send = torch.empty(nq * k_in * (8 + esz), dtype=torch.uint8, device=local_dist.device)
        send[: nq * k_in * 8].view(torch.int64).view(nq, k_in).copy_(local_idx)
        send[nq * k_in * 8:].view(local_dist.dtype).view(nq, k_in).copy_(local_dist)
        recv = torch.empty((world, send.numel()), dtype=torch.uint8, device=local_dist.device)
        dist.all_gather_into_tensor(recv, send, group=group)
        if merge_on is not None and rank != merge_on:
            return None
        host = recv.cpu().numpy().reshape(-1)
        np_dt = {torch.float32: np.float32, torch.float64: np.float64, torch.int32: np.int32}[local_dist.dtype]
        return _lib.merge_topk_gathered(host, world, nq, k_in, int(k), np_dt)
    gd = torch.empty((world,) + tuple(local_dist.shape), dtype=local_dist.dtype, device=local_dist.device)
    gi = torch.empty((world,) + tuple(local_idx.shape), dtype=local_idx.dtype, device=local_idx.device)
    dist.all_gather(list(gd.unbind(0)), local_dist.contiguous(), group=group)
    dist.all_gather(list(gi.unbind(0)), local_idx.contiguous(), group=group)
    if merge_on is not None and rank != merge_on:
        return None
    return _lib.merge_topk(gd.numpy(), gi.numpy(), int(k))


class ShardedIndex:
    """A rank's shard of a row-sharded index plus the collective search.

    ``local_search(queries, k) -> (dist, idx)`` answers over the local shard
    with GLOBAL ids and returns torch tensors; on a GPU box it is built by
    :func:`dense_shard` / :func:`hamming_shard` from the HIP index.
    """

    def __init__(self, local_search: Callable, group=None):
        self.local_search = local_search
        self.group = group

    def search(self, queries, k: int, merge_on: Optional[int] = None):
        d, i = self.local_search(queries, k)
        return allgather_merge(d, i, k, self.group, merge_on)


def dense_shard(db_shard, row0: int, metric: int = 0, group=None) -> ShardedIndex:
    """Shard from a CUDA float32 tensor ``[n_local, d]`` (borrowed, d % 64 == 0)."""
    import torch
    from . import _lib

    index = _lib.DenseIndex(db_shard.data_ptr(), n=db_shard.shape[0], d=db_shard.shape[1], metric=metric,
                            device_ptr=True, id_base=row0, keepalive=db_shard)
    ddt = torch.float64 if metric == _lib.SQ_METRIC_COSINE else torch.float32

    def local_search(queries, k):
        q = queries.contiguous()
        od = torch.empty((q.shape[0], k), dtype=ddt, device=q.device)
        oi = torch.empty((q.shape[0], k), dtype=torch.int64, device=q.device)
        index.search_device(q.data_ptr(), q.shape[0], k, od.data_ptr(), oi.data_ptr(),
                            torch.cuda.current_stream().cuda_stream)
        return od, oi

    s = ShardedIndex(local_search, group)
    s.index = index  # type: ignore[attr-defined]
    return s


def hamming_shard(codes_shard, row0: int, group=None) -> ShardedIndex:
    """Shard from a CUDA int64/uint64-viewed tensor ``[n_local, words]`` of packed codes."""
    import torch
    from . import _lib

    index = _lib.HammingIndex(codes_shard.data_ptr(), n=codes_shard.shape[0], words=codes_shard.shape[1],
                              device_ptr=True, id_base=row0, keepalive=codes_shard)

    def local_search(queries, k):
        q = queries.contiguous()
        od = torch.empty((q.shape[0], k), dtype=torch.int32, device=q.device)
        oi = torch.empty((q.shape[0], k), dtype=torch.int64, device=q.device)
        index.search_device(q.data_ptr(), q.shape[0], k, od.data_ptr(), oi.data_ptr(),
                            torch.cuda.current_stream().cuda_stream)
        return od, oi

    s = ShardedIndex(local_search, group)
    s.index = index  # type: ignore[attr-defined]
    return s
