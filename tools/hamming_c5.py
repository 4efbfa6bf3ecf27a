"""
bench.py --workload c5_hamming_shard: BASELINE.json config 5 -- 1 B x 256-bit codes, Hamming top-k, row-sharded over
8 MI355X = a 125 M-code shard (4 GB) per GPU (four uniform random 64-bit words per code, generated on the
device per shard from seed + rank; 256 queries; k = 100).  LinearHashIndex._nn (impls/hash_index/linear.py:206-244)
for batches of queries.

A step is ONE sq_hamming_search call on `--queries` query codes resident in HBM (pipelined, SQ_MEM_DEVICE_ASYNC), plus
-- with several ranks -- the all-gather of the per-shard top-k lists and the host merge (PipelinedShardedSearch).
Weak scaling: the shard is the config's, the code set grows with N; `value` = queries of all steps / time.
roofline: the code stream kernel (hamming_ring_kernel for few queries, hamming_stream_kernel beyond), algorithmic
bytes N * 32 per launch over its hipEvent-bracketed duration in blocking calls after the timed region.
`other_batches`: 1 and 16 queries per step (the HBM-bound operating points) measured the same way.
CPU leg: the oracle's hamming_topk on a bounded sample, scaled linearly in codes.
"""
import os
import time

import numpy as np

HBM_PEAK_GBS = 8000.0


def run(args, torch, dist, _lib, world, rank, dev, use_dist, emit):
    n_local = 125_000_000 if args.rows == 10_000_000 else args.rows // world     # (--rows: codes in total)
    w, k = 4, args.k
    nq = 256 if args.queries == 0 else args.queries
    stream = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator(device=dev)
    gen.manual_seed(5 + rank)
    codes = torch.empty((n_local, w), dtype=torch.int64, device=dev)
    for s in range(0, n_local, 1 << 24):
        e = min(n_local, s + (1 << 24))
        codes[s:e] = torch.randint(-2 ** 63, 2 ** 63 - 1, (e - s, w), dtype=torch.int64, device=dev, generator=gen)
    gq = torch.Generator(device=dev)
    gq.manual_seed(4321)                                   # the same queries on every rank
    nbatch = max(1, args.query_batches)
    max_b = max(nq, 16)
    all_q = torch.randint(-2 ** 63, 2 ** 63 - 1, (max_b * nbatch, w), dtype=torch.int64, device=dev, generator=gq)
    torch.cuda.synchronize()
    index = _lib.HammingIndex(codes.data_ptr(), n=n_local, words=w, device_ptr=True, id_base=rank * n_local, keepalive=codes)
    depth = min(max(args.async_depth if args.async_depth > 0 else 2, 2), 4)
    index.set_option("hamming_async_depth", depth)
    use_async = not args.sync_search

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(b, steps, warmup):
        """ms per step for b queries per step (rotating query batches), every result final inside the timed region."""
        qs = [all_q[j * b:(j + 1) * b].contiguous() for j in range(nbatch)]
        pipe = None
        if use_dist:
            from smqtk_indexing_amd.distributed import PipelinedShardedSearch
            pipe = PipelinedShardedSearch(index, b, k, torch.int32, merge_on=0, device=dev, use_async=use_async, depth=depth,
                                          gather_every=1, wait=True, queries_ready=True)
        od = [torch.empty((b, k), dtype=torch.int32, device=dev) for _ in range(depth)]
        oi = [torch.empty((b, k), dtype=torch.int64, device=dev) for _ in range(depth)]
        fallbacks = 0

        def step(i):
            nonlocal fallbacks
            q = qs[i % nbatch]
            if pipe is not None:
                pipe.submit(q)
            elif use_async:
                index.search_device_async(q.data_ptr(), b, k, od[i % depth].data_ptr(), oi[i % depth].data_ptr(), stream)
            else:
                index.search_device(q.data_ptr(), b, k, od[i % depth].data_ptr(), oi[i % depth].data_ptr(), stream)
            fallbacks += index.stats()["fallback_queries"]

        def drain():
            if pipe is not None:
                pipe.flush()
            elif use_async:
                index.sync()

        for i in range(warmup):
            step(i)
        drain()
        fallbacks = 0
        fence()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        drain()
        fence()
        dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        lag = pipe.results_lag if pipe is not None else (depth - 1 if use_async else 0)
        if pipe is not None:
            pipe.close()
        # the stream kernel alone: blocking calls, library hipEvents around the launch
        index.set_option("profile", 1)
        ks = []
        for i in range(3 + nbatch):
            q = qs[i % nbatch]
            index.search_device(q.data_ptr(), b, k, od[0].data_ptr(), oi[0].data_ptr(), stream)
            ks.append(index.stats()["scan_ms"])
        index.set_option("profile", 0)
        return dt / steps * 1e3, float(np.mean(ks[3:])), int(fallbacks), lag

    steps = max(1, args.steps if args.steps != 100 else 40)
    ms, kernel_ms, fb, lag = measure(nq, steps, args.warmup)
    extra = {}
    for b in (1, 16):
        if b != nq:
            m2, k2, f2, _ = measure(b, 40, 3)
            extra[f"batch_{b}"] = {"ms_per_step": m2, "queries_per_s": b * world / (m2 * 1e-3), "scan_kernel_ms": k2,
                                   "scan_GBps": n_local * w * 8 / (k2 * 1e-3) / 1e9,
                                   "scan_frac_of_hbm_peak": n_local * w * 8 / (k2 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "step_frac_of_hbm_peak": n_local * w * 8 / (m2 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "fallback_queries": f2}

    # parity spot check (one rank): integer exact against torch's top-k of (distance, row) keys over the whole shard
    parity = None
    if world == 1 and not args.no_parity_check:
        q = all_q[:2].contiguous()
        od = torch.empty((2, k), dtype=torch.int32, device=dev)
        oi = torch.empty((2, k), dtype=torch.int64, device=dev)
        index.search_device(q.data_ptr(), 2, k, od.data_ptr(), oi.data_ptr(), stream)
        torch.cuda.synchronize()
        rows = torch.arange(n_local, dtype=torch.int64, device=dev)

        def pop(x):
            x = x - ((x >> 1) & 0x5555555555555555)
            x = (x & 0x3333333333333333) + ((x >> 2) & 0x3333333333333333)
            x = (x + (x >> 4)) & 0x0F0F0F0F0F0F0F0F
            return (x * 0x0101010101010101) >> 56
        ok = True
        for j in range(2):
            dsum = torch.zeros(n_local, dtype=torch.int64, device=dev)
            for c in range(w):
                dsum += pop(codes[:, c] ^ q[j, c])
            want = torch.topk((dsum << 32) | rows, k, largest=False, sorted=True).values
            ok &= bool(torch.equal(od[j].to(torch.int64), want >> 32) and torch.equal(oi[j], want & 0xFFFFFFFF))
            del dsum
        parity = {"queries": 2, "integer_exact_vs_torch_topk": ok}

    if rank == 0:
        bytes_launch = float(n_local) * w * 8
        achieved = bytes_launch / (kernel_ms * 1e-3) / 1e9
        line = {
            "metric": f"queries/sec, Hamming top-{k} over {n_local * world} x {w * 64}-bit codes ({n_local} per GPU); code-stream HBM GB/s vs 8 TB/s peak",
            "value": nq * steps / (ms * steps * 1e-3), "unit": "queries/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64 (xor + popcount, integer exact)",
            "data": "synthetic uniform random 256-bit codes generated on device per shard (seed + rank); rotating query batches",
            "config": {"workload": f"c5_hamming_shard_{n_local}x{w * 64}b_per_gpu_k{k}", "ranks": world, "codes_total": n_local * world,
                       "codes_per_gpu": n_local, "queries_per_step": nq, "k": k,
                       "search_calls": f"pipelined (SQ_MEM_DEVICE_ASYNC, {depth} calls in flight)" if use_async else "blocking",
                       "results_lag_steps": lag, "fallback_queries": fb,
                       "collective": "one all_gather(top-k dist,idx) per step + host merge under the next steps' searches" if use_dist else "none"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "kernel": "hamming_ring_kernel" if nq <= 24 else "hamming_stream_kernel (xor + v_bcnt bound at this batch)",
                         "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": bytes_launch,
                         "frac_step": bytes_launch / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "other_batches": extra,
        }
        if parity is not None:
            line["parity_check"] = parity
        if not args.no_cpu_baseline and world == 1:
            from oracle import cpu_ref as O
            rng = np.random.default_rng(7)
            m = 4_000_000
            ch = rng.integers(0, 2 ** 64, size=(m, w), dtype=np.uint64)
            qh = rng.integers(0, 2 ** 64, size=(4, w), dtype=np.uint64)
            t1 = time.perf_counter()
            for qq in qh:
                O.hamming_topk(ch, qq, k)
            dt = time.perf_counter() - t1
            line["cpu_baseline"] = {"value": 4 / dt * m / n_local, "unit": "queries/s", "cores": 1, "kind": "port",
                                    "sample": f"oracle hamming_topk (vectorised numpy popcount + lexsort): 4 queries x {m} codes x 256 bit in "
                                              f"{dt:.2f} s on 1 thread (host has {len(os.sched_getaffinity(0))} cores); scaled linearly to {n_local} codes"}
        emit(line)
    index.close()
