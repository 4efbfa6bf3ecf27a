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
import time
from typing import Callable, Optional, Tuple

import numpy as np


def shard_range(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Rows ``[r0, r1)`` of rank ``rank``: contiguous blocks of ``ceil(n/world)``."""
    per = (int(n_total) + world - 1) // world
    r0 = min(rank * per, n_total)
    return r0, min(r0 + per, n_total)


def packed_block_bytes(nq: int, k_in: int, dist_itemsize: int) -> int:
    """Bytes of one shard's block ``[ids int64 nq*k_in][dist nq*k_in]`` in a packed all-gather buffer, rounded
    up to 8 so that every shard's id block stays 8-byte aligned (float32 / int32 distances with odd nq*k_in)."""
    return (int(nq) * int(k_in) * (8 + int(dist_itemsize)) + 7) // 8 * 8


def allgather_merge(local_dist, local_idx, k: int, group=None, merge_on: Optional[int] = None,
                    packed: Optional[bool] = None):
    """All-gather every rank's ``[nq, k_in]`` top-k lists and merge them.

    ``local_dist`` / ``local_idx`` are torch tensors on the rank's device (CUDA
    tensors go over RCCL, CPU tensors over gloo).  Ids must already be global.
    ``packed`` (default: on for CUDA tensors): ONE collective of a byte buffer per rank,
    ``[ids int64 nq*k_in][dist nq*k_in]``, merged in place from the receive buffer
    (``sq_merge_topk_strided``); otherwise two plain all-gathers.
    Returns ``(dist [nq,k], idx [nq,k])`` numpy arrays on every rank, or only on
    rank ``merge_on`` (others get ``None``) when given.
    """
    import torch
    import torch.distributed as dist
    from . import _lib

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    nq, k_in = int(local_dist.shape[0]), int(local_dist.shape[1])
    if packed is None:
        packed = bool(local_dist.is_cuda)
    if packed:
        esz = local_dist.element_size()
        per = packed_block_bytes(nq, k_in, esz)
        send = torch.zeros(per, dtype=torch.uint8, device=local_dist.device)
        send[: nq * k_in * 8].view(torch.int64).view(nq, k_in).copy_(local_idx)
        send[nq * k_in * 8: nq * k_in * (8 + esz)].view(local_dist.dtype).view(nq, k_in).copy_(local_dist)
        recv = torch.empty(world * per, dtype=torch.uint8, device=local_dist.device)   # 1-D: gloo wants it flat
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
    return _lib.merge_topk(gd.cpu().numpy(), gi.cpu().numpy(), int(k))


class ShardedIndex:
    """A rank's shard of a row-sharded index plus the collective search.

    ``local_search(queries, k) -> (dist, idx)`` answers over the local shard
    with GLOBAL ids and returns torch tensors; on a GPU box it is built by
    :func:`dense_shard` / :func:`hamming_shard` from the HIP index.
    """

    def __init__(self, local_search: Callable, group=None, packed: Optional[bool] = None):
        self.local_search = local_search
        self.group = group
        self.packed = packed

    def search(self, queries, k: int, merge_on: Optional[int] = None):
        d, i = self.local_search(queries, k)
        return allgather_merge(d, i, k, self.group, merge_on, self.packed)


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


class MutableShardedIndex:
    """Row shards that accept mutations ("Mutations"; not on the timed path).

    * ``append(rows)`` -- every rank calls it with the same rows; they go to the shard with the fewest
      live rows (lowest rank on ties) and get the next global ids, so every shard stays in ascending id
      order and the canonical (distance, id) tie order survives mutation.  Returns the new ids.
    * ``remove(ids)`` -- tombstones: the owning shard marks the rows dead; ``KeyError`` on every rank,
      before anything changes, when an id is not live anywhere (the reference's
      ``remove_from_index`` contract, nearest_neighbor_index.py:84-94 / lsh.py:385-450).  A shard whose
      dead rows exceed ``compact_at`` of its rows (or ``max_dead``: a shard answers ``k + dead`` locally and the
      kernels cap k) drops them and rebuilds its local index.
    * ``search(queries, k)`` -- each shard answers ``k + dead`` locally, drops dead rows, maps local rows to
      global ids, then the usual all-gather + host merge.

    ``build_local(rows) -> search(queries, k) -> (dist [nq,k], local_row [nq,k])`` builds the per-shard
    searcher (padding: row -1); :func:`dense_local_builder` / :func:`hamming_local_builder` wrap the HIP
    indexes, the CPU tests pass an oracle-backed one.  ``rows`` is a torch tensor (CUDA on a GPU box).
    The only collectives of a mutation are all-gathers / all-reduces of a few integers.
    """

    def __init__(self, rows, row0: int, n_total: int, build_local: Callable, group=None, compact_at: float = 0.25,
                 dist_dtype=None, max_dead: int = 1024):
        import torch
        import torch.distributed as dist

        self.group, self.build_local, self.compact_at = group, build_local, float(compact_at)
        self.max_dead = int(max_dead)   # a shard answers k + dead locally: keep that under the kernels' k limit
        self.pad_dtype = dist_dtype if dist_dtype is not None else torch.float32   # distances of an empty shard
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.rows = rows
        self.ids = torch.arange(int(row0), int(row0) + int(rows.shape[0]), dtype=torch.int64)
        self.dead = torch.zeros(int(rows.shape[0]), dtype=torch.bool)
        self.next_id = int(n_total)
        self.live = self._gather_int(int(rows.shape[0]))       # live rows of every shard, identical on all ranks
        self._local = build_local(rows) if rows.shape[0] else None

    # ------------------------------------------------------------ helpers
    def _gather_int(self, v: int):
        import torch
        import torch.distributed as dist
        dev = self.rows.device if self.rows.is_cuda else torch.device("cpu")
        mine = torch.tensor([int(v)], dtype=torch.int64, device=dev)
        outs = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(self.world)]
        dist.all_gather(outs, mine, group=self.group)
        return [int(x.item()) for x in outs]

    def _rebuild(self) -> None:
        self._local = self.build_local(self.rows) if self.rows.shape[0] else None

    def count(self) -> int:
        return int(sum(self.live))

    # ---------------------------------------------------------- mutations
    def append(self, new_rows):
        import torch
        m = int(new_rows.shape[0])
        ids = torch.arange(self.next_id, self.next_id + m, dtype=torch.int64)
        self.next_id += m
        target = min(range(self.world), key=lambda r: (self.live[r], r))
        if m and self.rank == target:
            add = new_rows.to(self.rows.device, self.rows.dtype)
            self.rows = torch.cat([self.rows, add], dim=0) if self.rows.shape[0] else add.contiguous()
            self.ids = torch.cat([self.ids, ids])
            self.dead = torch.cat([self.dead, torch.zeros(m, dtype=torch.bool)])
            self._rebuild()
        self.live[target] += m
        return ids.numpy()

    def remove(self, ids) -> None:
        import torch
        want = torch.as_tensor(np.asarray(list(ids), dtype=np.int64))
        if want.numel() != torch.unique(want).numel():
            raise KeyError("duplicate ids")
        # local rows are in ascending id order: binary search
        pos = torch.searchsorted(self.ids, want)
        pos_c = pos.clamp(max=max(int(self.ids.numel()) - 1, 0))
        hit = (pos < self.ids.numel()) & (self.ids[pos_c] == want) & ~self.dead[pos_c] if self.ids.numel() else \
            torch.zeros_like(want, dtype=torch.bool)
        found = self._gather_int(int(hit.sum()))
        if sum(found) != int(want.numel()):
            raise KeyError("some ids are not in the index")        # nothing modified, on every rank
        self.dead[pos_c[hit]] = True
        for r in range(self.world):
            self.live[r] -= found[r]
        n_dead = int(self.dead.sum())
        if n_dead and (n_dead > self.compact_at * int(self.dead.numel()) or n_dead > self.max_dead):
            keep = ~self.dead
            self.rows = self.rows[keep.to(self.rows.device)].contiguous()
            self.ids, self.dead = self.ids[keep], torch.zeros(int(keep.sum()), dtype=torch.bool)
            self._rebuild()

    # -------------------------------------------------------------- search
    def search(self, queries, k: int, merge_on: Optional[int] = None):
        import torch
        nq = int(queries.shape[0])
        n_local, n_dead = int(self.dead.numel()), int(self.dead.sum())
        kk = min(n_local, int(k) + n_dead)
        if self._local is not None and kk > 0:
            d, r = self._local(queries, kk)
            d, r = d.cpu(), r.cpu()
        else:
            d, r = None, None
        dt = d.dtype if d is not None else self.pad_dtype
        pad = torch.iinfo(dt).max if not dt.is_floating_point else float("inf")
        od = torch.full((nq, int(k)), pad, dtype=dt)
        oi = torch.full((nq, int(k)), -1, dtype=torch.int64)
        if d is not None:
            ok = (r >= 0) & ~self.dead[r.clamp(min=0)]
            # stable left-compaction of the live entries of every row
            order = torch.argsort((~ok).to(torch.int8), dim=1, stable=True)[:, : int(k)]
            live_sorted = torch.gather(ok, 1, order)
            dd, rr = torch.gather(d, 1, order), torch.gather(r, 1, order)
            m = min(int(k), kk)
            od[:, :m] = torch.where(live_sorted, dd, torch.full_like(dd, pad))[:, :m]
            oi[:, :m] = torch.where(live_sorted, self.ids[rr.clamp(min=0)], torch.full_like(rr, -1))[:, :m]
        dev = self.rows.device if self.rows.is_cuda else torch.device("cpu")
        return allgather_merge(od.to(dev), oi.to(dev), int(k), self.group, merge_on)



def dense_local_builder(metric: int = 0) -> Callable:
    """``build_local`` for :class:`MutableShardedIndex` over CUDA float32 rows (HIP dense index, local row ids)."""
    import torch
    from . import _lib
    ddt = torch.float64 if metric == _lib.SQ_METRIC_COSINE else torch.float32

    def build(rows):
        index = _lib.DenseIndex(rows.data_ptr(), n=rows.shape[0], d=rows.shape[1], metric=metric, device_ptr=True,
                                keepalive=rows)

        def search(queries, k):
            q = queries.to(rows.device, torch.float32).contiguous()
            od = torch.empty((q.shape[0], k), dtype=ddt, device=q.device)
            oi = torch.empty((q.shape[0], k), dtype=torch.int64, device=q.device)
            index.search_device(q.data_ptr(), q.shape[0], k, od.data_ptr(), oi.data_ptr(),
                                torch.cuda.current_stream().cuda_stream)
            torch.cuda.current_stream().synchronize()
            return od, oi
        search.index = index  # type: ignore[attr-defined]
        return search
    return build


def hamming_local_builder() -> Callable:
    """``build_local`` for :class:`MutableShardedIndex` over CUDA int64-viewed packed codes."""
    import torch
    from . import _lib

    def build(codes):
        index = _lib.HammingIndex(codes.data_ptr(), n=codes.shape[0], words=codes.shape[1], device_ptr=True,
                                  keepalive=codes)

        def search(queries, k):
            q = queries.to(codes.device, torch.int64).contiguous()
            od = torch.empty((q.shape[0], k), dtype=torch.int32, device=q.device)
            oi = torch.empty((q.shape[0], k), dtype=torch.int64, device=q.device)
            index.search_device(q.data_ptr(), q.shape[0], k, od.data_ptr(), oi.data_ptr(),
                                torch.cuda.current_stream().cuda_stream)
            torch.cuda.current_stream().synchronize()
            return od, oi
        search.index = index  # type: ignore[attr-defined]
        return search
    return build


class PipelinedMerger:
    """Host merges of gathered top-k buffers on a worker thread, so that rank 0 merges batch i while the
    GPUs already search batch i + 1 (the merge is ``sq_merge_topk_strided``: a ctypes call, the GIL is
    released for its duration).  ``submit`` hands over a receive buffer (one all-gather's host copy,
    ``[shards][ids int64 nq*k | dist nq*k]``) and returns a ticket; ``result(ticket)`` waits for that merge.
    Depth is the caller's business: a buffer must not be refilled before its ticket has been collected.
    """

    def __init__(self):
        import queue
        import threading
        self._jobs: "queue.Queue" = queue.Queue()
        self._done = {}
        self._cv = threading.Condition()
        self._next = 0
        self.merge_seconds = []      # duration of every merge (the worker's clock): reported by bench.py
        self._thread = threading.Thread(target=self._run, name="sq-merge", daemon=True)
        self._thread.start()

    def _run(self) -> None:
        from . import _lib
        while True:
            job = self._jobs.get()
            if job is None:
                return
            ticket, buf, shards, nq, k_in, k_out, dt = job
            t0 = time.perf_counter()
            try:
                res = _lib.merge_topk_gathered(buf, shards, nq, k_in, k_out, dt)
            except Exception as ex:  # noqa: BLE001 -- handed to the caller of result()
                res = ex
            self.merge_seconds.append(time.perf_counter() - t0)
            with self._cv:
                self._done[ticket] = res
                self._cv.notify_all()

    def submit(self, buf: np.ndarray, shards: int, nq: int, k_in: int, k_out: int, dist_dtype) -> int:
        ticket = self._next
        self._next += 1
        self._jobs.put((ticket, buf, int(shards), int(nq), int(k_in), int(k_out), dist_dtype))
        return ticket

    def result(self, ticket: int):
        with self._cv:
            while ticket not in self._done:
                self._cv.wait()
            res = self._done.pop(ticket)
        if isinstance(res, Exception):
            raise res
        return res

    def close(self) -> None:
        self._jobs.put(None)
        self._thread.join()


class HipSearcher:
    """Adapter of a ``_lib.DenseIndex`` / ``_lib.HammingIndex`` for :class:`PipelinedShardedSearch`.

    ``search_into(q, k, out_d, out_i)`` writes the shard's top-k for the query batch into the two device tensors.
    ``lag`` says when that answer is final: 0 -- when ``search_into`` returns (the synchronous C ABI call);
    ``depth - 1`` -- when that many further ``search_into`` calls (or ``finish``) have returned: the searches with
    ``SQ_MEM_DEVICE_ASYNC``, which keep ``depth`` calls on the device (include/smqtk_hip.h; option
    ``dense_async_depth`` / ``hamming_async_depth``, 2 by default -- small shards gain from 3: DESIGN.md section 5).

    The pipeline's depth / wait / order are options of THIS index's handle (``sq_handle_set_option``): another index
    searched in the same process -- another pipeline, another thread -- keeps its own.

    Query lifetime: the C ABI reads ``queries`` until the call is final (``lag`` calls later: the kernels run on the
    library's internal streams, which torch's caching allocator knows nothing about, and an uncertified query is
    redone from the same pointer).  ``search_into`` therefore keeps a reference to the last ``lag + 1`` query
    tensors (and output tensors); callers may hand over temporaries and drop them at once."""

    _live: dict = {}    # id(index) -> weak reference to the searcher whose pipeline options sit on that handle

    def __init__(self, index, stream_handle: int = 0, use_async: bool = False, depth: int = 2, wait: bool = True,
                 queries_ready: bool = False):
        import collections
        import weakref
        self.index, self.stream = index, int(stream_handle)
        self.lag = 0
        self._prefix = None
        if use_async and hasattr(index, "search_device_async"):
            # the depth / wait / order below are written on the index's handle and read by every asynchronous call on
            # it: a second live pipeline on the same handle would overwrite them under the first one's feet (its `lag`
            # going stale), so it is refused; close() puts the library's defaults back
            other = HipSearcher._live.get(id(index))
            if other is not None and other() is not None and other().index is index and other()._prefix is not None:
                raise RuntimeError("HipSearcher: this index already has a live asynchronous pipeline (close() it first)")
            HipSearcher._live[id(index)] = weakref.ref(self)
            depth = min(max(int(depth), 2), 4)
            # wait=False: a call returns right after enqueueing (option *_async_wait = 0) and the wait for the oldest
            # call moves to the start of the next one: one more call of lag, and the host work between two calls (the
            # pipeline's gather and merge bookkeeping) overlaps the device instead of delaying the next enqueue
            # queries_ready=True: every query tensor handed to search_into is complete (nothing still writing it on a
            # stream): the per-call event that orders the search behind the caller's stream is skipped
            prefix = getattr(index, "async_option_prefix", "dense_async")
            index.set_option(prefix + "_depth", depth)
            index.set_option(prefix + "_wait", 1 if wait else 0)
            index.set_option(prefix + "_order", 0 if queries_ready else 1)
            self._prefix = prefix
            self.lag = depth - 1 if wait else depth
        self._alive = collections.deque(maxlen=self.lag + 1)   # (queries, out_d, out_i) of the calls not yet final

    def search_into(self, queries, k: int, out_d, out_i) -> None:
        q = queries if queries.is_contiguous() else queries.contiguous()
        self._alive.append((q, out_d, out_i))
        fn = self.index.search_device_async if self.lag else self.index.search_device
        fn(q.data_ptr(), int(q.shape[0]), int(k), out_d.data_ptr(), out_i.data_ptr(), self.stream)

    def finish(self) -> None:
        if self.lag:
            self.index.sync()
        self._alive.clear()

    def close(self) -> None:
        """Finish what is in flight and take the pipeline's options off the handle again (the library's defaults:
        two calls in flight, a call returns when the oldest is final, calls ordered behind the caller's stream), so a
        later direct ``search_device_async`` caller finds the contract of include/smqtk_hip.h."""
        self.finish()
        if self._prefix is not None:
            try:
                for name, value in (("_depth", 2), ("_wait", 1), ("_order", 1)):
                    self.index.set_option(self._prefix + name, value)
            except Exception:      # (the index was closed first: nothing left to restore)
                pass
            self._prefix = None
            HipSearcher._live.pop(id(self.index), None)


class PipelinedShardedSearch:
    """Batches through a sharded index with the collective and the merge off the critical path.

    ``submit(queries)`` runs the local search of batch i (``searcher.search_into``: see :class:`HipSearcher`).
    ``gather_every`` (G) consecutive batches share one send buffer -- laid out as the packed block of ONE batch of
    G * nq queries -- and one all-gather: G = 1 exchanges every batch (lowest latency); a larger G pays the fixed
    host and launch cost of a collective once per G batches, which is what limits small shards (a 1.25 M-row shard
    answers 32 queries in ~80 us, a collective with its pinned copy and hand-off costs about as much again).  Once
    the last batch of a group is final (``lag`` submits later) its all-gather starts asynchronously (RCCL's stream /
    gloo's thread) and, on the merging rank, the copy of the gathered buffer to pinned host memory on torch's current
    stream; both run under the following searches.  The group before goes to the merge thread
    (:class:`PipelinedMerger`) and the merged group whose buffers are taken over comes back.
    ``submit`` returns ``None`` while nothing left the pipeline (and always on the other ranks), else the merged
    ``(dist, idx)`` of ONE batch (G = 1) or the list of a group's G results, oldest first (G > 1).  ``flush()``
    finishes the searches and returns every result still in flight as a flat list, oldest first.
    At least two send / receive / host buffers rotate (more when ``lag`` exceeds a group); every reuse is ordered
    behind the previous user.  CPU tensors (gloo) take the same path without streams or pinned copies: the CPU tests
    drive it with an oracle-backed searcher.
    ``searcher``: an object with ``lag``, ``search_into`` and ``finish`` (or a ``_lib`` index: wrapped in a
    :class:`HipSearcher` on a compute stream of the pipeline's own; ``depth`` = asynchronous searches in flight,
    ``wait`` = False: searches return right after enqueueing; ``queries_ready`` = True: the query tensors handed to
    ``submit`` are complete, no stream ordering needed -- see :class:`HipSearcher`).
    Lifetime rule: a query tensor handed to ``submit`` is read by the device until its batch is final, ``lag`` submits
    later; :class:`HipSearcher` keeps the reference until then, so callers may submit temporaries.  A searcher of your
    own must do the same.  ``results_lag`` = submits between a batch going in and its merged result coming out.
    """

    def __init__(self, searcher, nq: int, k: int, dist_dtype, group=None, merge_on: int = 0, device=None,
                 use_async: bool = False, depth: int = 2, gather_every: int = 1, wait: bool = True,
                 queries_ready: bool = False):
        import torch
        import torch.distributed as dist
        self.nq, self.k, self.group = int(nq), int(k), group
        self.world, self.rank, self.merge_on = dist.get_world_size(group), dist.get_rank(group), merge_on
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        dev = torch.device(device)
        self.cuda = dev.type == "cuda"
        if not hasattr(searcher, "search_into"):
            # a _lib index.  Blocking searches get a stream of their own (collectives and copies stay on the current
            # stream).  Asynchronous searches already run on the library's internal streams and are only ORDERED behind
            # the stream handed over: no stream is created for them -- the device has few hardware queues (4 by default
            # on ROCm), and one more active stream made two of the library's streams share a queue: the overlap between
            # neighbouring searches was lost, 0.075 -> 0.107 ms per step on a 1.25 M-row shard (tools/pipe_ablate.py)
            if use_async and hasattr(searcher, "search_device_async"):
                handle = torch.cuda.current_stream(dev).cuda_stream
            else:
                self.compute = torch.cuda.Stream(device=dev)
                handle = self.compute.cuda_stream
            searcher = HipSearcher(searcher, handle, use_async, depth, wait, queries_ready)
        self.searcher = searcher
        self.lag = int(searcher.lag)
        self.G = G = max(1, int(gather_every))
        # a group's buffer is refilled (nb - 1) groups after its last batch; its gather starts `lag` batches after it
        self.nb = nb = max(2, 2 + (self.lag - 1) // G) if self.lag else 2
        nqt = self.nq * G
        esz = torch.empty(0, dtype=dist_dtype).element_size()
        self.np_dt = {torch.float32: np.float32, torch.float64: np.float64, torch.int32: np.int32}[dist_dtype]
        per = packed_block_bytes(nqt, self.k, esz)
        nk = nqt * self.k
        self.send = [torch.zeros(per, dtype=torch.uint8, device=dev) for _ in range(nb)]
        self.out_i = [s[: nk * 8].view(torch.int64).view(nqt, self.k) for s in self.send]
        self.out_d = [s[nk * 8: nk * (8 + esz)].view(dist_dtype).view(nqt, self.k) for s in self.send]
        self.recv = [torch.empty(self.world * per, dtype=torch.uint8, device=dev) for _ in range(nb)]   # 1-D: gloo wants it flat
        self.work = [None] * nb
        self.valid = [0] * nb      # batches in the group a buffer holds (the last group of a run may be short)
        self.i = 0                 # batches submitted
        self.gathered = 0          # groups whose all-gather has been started
        self.gather_host_seconds = []   # host time of starting a group's all-gather + pinned copy (bench.py reports it)
        self.merging = self.rank == merge_on
        # submits between batch i going in and its merged result coming out of submit(): its group must be complete and
        # final (G - 1 + lag), then gathered, copied and merged under the next `nb` groups' searches
        self.results_lag = (G - 1) + self.lag + nb * G
        if self.merging:
            if self.cuda:
                self.host = [torch.empty(self.world * per, dtype=torch.uint8, pin_memory=True) for _ in range(nb)]
                self.copied = [torch.cuda.Event() for _ in range(nb)]    # host[j] holds the gathered buffer of its group
            else:
                self.host = self.recv
            self.host_np = [h.numpy().reshape(-1) for h in self.host]
            self.copy_pending = [False] * nb
            self.ticket = [None] * nb                                 # merge reading host[j]: (ticket, batches)
            self.merger = PipelinedMerger()

    def _wait_work(self, j: int) -> None:
        w = self.work[j]
        if w is not None:
            if self.cuda:
                while not w.is_completed():   # finished long ago in the steady state -- make it formal
                    time.sleep(0)             # (yields the GIL: the merge thread may be waiting for it)
            else:
                w.wait()
            self.work[j] = None

    def _merge_ready(self, j: int) -> None:
        """Group in buffer j: its host copy is complete -> to the merge thread."""
        if self.copy_pending[j]:
            if self.cuda:
                ev = self.copied[j]
                while not ev.query():
                    time.sleep(0)
            else:
                self._wait_work(j)
            self.copy_pending[j] = False
            t = self.merger.submit(self.host_np[j], self.world, self.nq * self.G, self.k, self.k, self.np_dt)
            self.ticket[j] = (t, self.valid[j])

    def _collect(self, j: int):
        """The merged batches of the group whose merge reads host[j] (oldest first)."""
        t, batches = self.ticket[j]
        self.ticket[j] = None
        dd, ii = self.merger.result(t)
        return [(dd[g * self.nq:(g + 1) * self.nq], ii[g * self.nq:(g + 1) * self.nq]) for g in range(batches)]

    def _gather_next(self):
        """Start the all-gather of the oldest group not yet gathered (its send buffer is final) and move the
        groups behind it one stage on; returns the merged batches that left the pipeline (a list, maybe empty)."""
        import torch.distributed as dist
        j = self.gathered % self.nb
        first = self.gathered * self.G
        self.valid[j] = min(self.G, self.i - first)
        self.gathered += 1
        ready = []
        if self.merging and self.ticket[j] is not None:
            ready = self._collect(j)                        # `nb` groups back: merged under the searches since
        t0 = time.perf_counter()
        self.work[j] = w = dist.all_gather_into_tensor(self.recv[j], self.send[j], group=self.group, async_op=True)
        if self.merging:
            if self.cuda:
                w.wait()                                   # orders the current stream (not the host) behind the gather
                self.host[j].copy_(self.recv[j], non_blocking=True)
                self.copied[j].record()
            self.copy_pending[j] = True
        self.gather_host_seconds.append(time.perf_counter() - t0)
        if self.merging:
            self._merge_ready((j - 1) % self.nb)           # the group before: gathered and copied meanwhile
        return ready

    def submit(self, queries):
        grp, g = divmod(self.i, self.G)
        j = grp % self.nb
        self.i += 1
        if g == 0:
            self._wait_work(j)   # the all-gather `nb` groups back read send[j] and wrote recv[j]
        lo, hi = g * self.nq, (g + 1) * self.nq
        self.searcher.search_into(queries, self.k, self.out_d[j][lo:hi], self.out_i[j][lo:hi])
        # groups whose last batch is batch i - lag or older are final in their send buffers now
        if (self.gathered + 1) * self.G + self.lag <= self.i:
            ready = self._gather_next()
            if ready:
                return ready[0] if self.G == 1 else ready
        return None

    def flush(self):
        out = []
        self.searcher.finish()
        while self.gathered * self.G < self.i:
            out += self._gather_next()
        if self.merging:
            from . import _lib
            newest = (self.gathered - 1) % self.nb if self.gathered else 0
            order = [(newest + 1 + t) % self.nb for t in range(self.nb)]     # oldest first
            for j in order:
                if self.ticket[j] is not None:             # already with the merge thread
                    out += self._collect(j)
                if self.copy_pending[j]:
                    # nothing is left to overlap with: merge here instead of paying two thread hand-offs
                    if self.cuda:
                        self.copied[j].synchronize()
                    else:
                        self._wait_work(j)
                    self.copy_pending[j] = False
                    dd, ii = _lib.merge_topk_gathered(self.host_np[j], self.world, self.nq * self.G, self.k, self.k, self.np_dt)
                    out += [(dd[g * self.nq:(g + 1) * self.nq], ii[g * self.nq:(g + 1) * self.nq]) for g in range(self.valid[j])]
        for j in range(self.nb):
            self._wait_work(j)
        self.i = self.gathered * self.G          # (a short last group is closed: the next batch starts a new one)
        return out

    def close(self) -> None:
        self.flush()
        if hasattr(self.searcher, "close"):
            self.searcher.close()
        if self.merging:
            self.merger.close()
