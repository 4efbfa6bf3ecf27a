#!/usr/bin/env python3
"""
bench.py -- the reference's headline metric on MI355X: queries/sec of exact
brute-force L2 kNN (k=100) over a 10M x 128 float32 database, with the achieved
HBM bandwidth of the scan kernel against the ~8 TB/s peak (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

A "step" is one search call: `--queries` (default 32) queries through the whole hot path
(query prep -> sampled threshold -> MFMA scan of the shard -> exact re-rank ->
top-k select -> certification [-> all-gather + host merge when N > 1]).

Scaling definition (STRONG): the database (`--rows`, 10M in total) is row-sharded over the N
ranks and already resident in HBM, and EVERY N -- N = 1 included -- answers the same
`--queries` queries per step, so `value` at N GPUs over `value` at 1 GPU is the speed-up of
the same job.  (Round 1 grew the batch with N; that compared different jobs.)  Larger total
batches are measured after the timed region and reported as `other_batches` at every N.

Searches are pipelined (`sq_dense_search` with SQ_MEM_DEVICE_ASYNC, include/smqtk_hip.h):
the kernels of step i + 1 are enqueued while step i runs and the status words of step i are
read one call later, so the device never waits for the host; every step's results are final
and every merge is collected before the closing fence (`--sync-search`: one blocking call per
step, the round-1 behaviour).  One process per GPU (torch.distributed / RCCL); rank 0 prints
ONE JSON line.  Two searches are in flight on one GPU, three on shards (`--async-depth`: the short kernels
of a call -- head, select -- overlap the neighbours' passes); with several ranks a search call returns
right after enqueueing (`--async-wait 0`: the gather / merge bookkeeping overlaps the device), four
steps share one all-gather (`--gather-every`: the collective's fixed cost is that of a whole step on
a 1.25 M-row shard) and the library's timing events are taken on every 4th search; results come
back a group at a time, all inside the timed region.

Other workloads (reported with the same contract, not the driver's default):
    --workload lsh_c3        BASELINE config 3: 10M x 128 -> 64-bit ITQ codes -> Hamming top-n
                             -> bucket expansion -> exact re-rank (HipLSH pipeline on device),
                             queries/s AND recall@k against the exact brute-force ground truth
    --data {normal,uniform,clustered,nonneg}, --metric {l2,cosine}
                             other descriptor distributions / the cosine metric for the
                             brute-force workload (candidates per query, fallbacks)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
MFMA_BF16_PEAK_TF = 2500.0  # v_mfma_f32_32x32x16_bf16 dense peak (MI355X_MICROARCH.md)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)   # (short runs weigh the pipeline's fill and drain: with shards the
    ap.add_argument("--warmup", type=int, default=5)    #  last groups' gathers and merges are inside the timed region)
    ap.add_argument("--rows", type=int, default=10_000_000, help="database rows in total (all shards)")
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--queries", type=int, default=0, help="queries per step in total -- the same at every N (0 = the workload's: 32; c4_cosine_shard 256)")
    ap.add_argument("--queries-per-gpu", type=int, default=0,
                    help="(round-1 semantics, kept for experiments) queries per step = this x N; overrides --queries")
    ap.add_argument("--workload", choices=["bruteforce", "lsh_c3", "c4_cosine_shard", "c5_hamming_shard"], default="bruteforce")
    ap.add_argument("--query-batches", type=int, default=8,
                    help="distinct query batches the timed steps rotate through (1 = the same queries every step)")
    ap.add_argument("--no-mid-tier", action="store_true", help="uncertified queries go straight to the exact path (measurement)")
    ap.add_argument("--lib-options", default=os.environ.get("SQ_BENCH_OPTS", ""),
                    help="measurement: library options set on the index, name=value,... (reported in config.lib_options)")
    ap.add_argument("--no-int8", action="store_true", help="bf16 first-stage filter only: no int8 copy is built (measurement)")
    ap.add_argument("--preroll-ms", type=float, default=30.0,
                    help="untimed pipelined steps for this long before the warm-up steps (start-up transient of the device; 0 = none)")
    ap.add_argument("--profile-every", type=int, default=0,
                    help="library hipEvents on every N-th search of the timed region (0 = 1 on one GPU, 4 on shards)")
    ap.add_argument("--data", choices=["normal", "uniform", "clustered", "nonneg", "far_clusters"], default="normal")
    ap.add_argument("--query-scale", type=float, default=1.0,
                    help="multiply the queries by this (300: queries far longer than the rows -- with --data far_clusters the "
                         "two families the bf16 filter cannot certify; they take the middle tier, sq_dense_mid.hpp)")
    ap.add_argument("--metric", choices=["l2", "cosine"], default="l2")
    ap.add_argument("--sync-search", action="store_true", help="one blocking search call per step (no pipelining)")
    ap.add_argument("--async-streams", type=int, default=0, help="option dense_async_streams (0 = library default)")
    ap.add_argument("--async-depth", type=int, default=0,
                    help="asynchronous searches in flight (option dense_async_depth); 0 = 2 on one GPU, 3 on shards "
                         "(the short kernels of a shard's call overlap two neighbours' passes, DESIGN.md sections 4.1b and 5)")
    ap.add_argument("--async-wait", type=int, default=-1,
                    help="several GPUs: 1 = a search call returns when the oldest call in flight is final, 0 = right after "
                         "enqueueing (option dense_async_wait): the gather / merge bookkeeping between two calls then overlaps "
                         "the device, at one more call of lag.  -1 (default): 0 (round 4, three-launch calls: the one-rank proxy "
                         "of the 8-GPU step reads 0.082 against 0.090 ms per step at 20 steps and 0.065 against 0.068 at 100; "
                         "with round 3's six-launch calls short runs were better off with 1)")
    ap.add_argument("--gather-every", type=int, default=0,
                    help="several GPUs: steps per all-gather (PipelinedShardedSearch gather_every); 0 = 4: a collective "
                         "with its pinned copy and merge hand-off costs about as much host time as a 1.25 M-row shard "
                         "needs for a step, so four steps share one (all merges still inside the timed region)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-check", action="store_true")
    ap.add_argument("--no-other-paths", action="store_true",
                    help="skip the ITQ / Hamming timings reported beside the headline metric")
    ap.add_argument("--force-collective", action="store_true",
                    help="testing: run the all-gather + merge path even with one rank (launch under torch.distributed.run)")
    ap.add_argument("--extra-batches", type=str, default="1,64,128,256,1024",
                    help="other total batch sizes measured after the timed region; '' to skip")
    ap.add_argument("--lsh-n", type=int, default=0, help="lsh_c3: nearest codes asked of the hash index (0 = k)")
    ap.add_argument("--fit", action="store_true",
                    help="lsh_c3: hash with a model TRAINED on the data (HipItqFunctor.fit on the device) instead of the fixed "
                         "random rotation of SURVEY 8d; with --data clustered the recall then says something about the method")
    ap.add_argument("--launch-selftest", action="store_true",
                    help="no GPU work: the ranks rendezvous over gloo, all-reduce one integer and rank 0 prints a line "
                         "(tests/test_host_logic.py checks the --gpus N self-launch with it)")
    return ap.parse_args()


def cpu_baseline(dim: int, k: int, seed: int):
    """The oracle (numpy restatement of metrics.euclidean_distance + stable
    sort, oracle/cpu_ref.py) timed on this host on a bounded sample of the same
    workload; scaled linearly in rows to the full database and labelled so."""
    from oracle import cpu_ref as O
    rows, nq = max(20_000, 2_000_000 * 128 // max(dim, 128)), 32   # ~15 s of single-thread numpy on the GPU box's host (1 GB of rows)
    rng = np.random.default_rng(seed)
    db = rng.standard_normal((rows, dim), dtype=np.float32)
    qs = rng.standard_normal((nq, dim), dtype=np.float32)
    O.dense_topk(db[:100_000], qs[0], k)            # warm numpy
    t0 = time.perf_counter()
    for q in qs:
        O.dense_topk(db, q, k)
    dt = time.perf_counter() - t0
    return rows, nq, dt


def make_rows(torch, kind: str, n: int, d: int, dev, seed: int, chunk: int = 1 << 20):
    """Synthetic float32 descriptors generated on the device, chunk by chunk.
    normal: N(0,1) (SURVEY 8d north-star set).  uniform: U[0,1) (C1's distribution).  nonneg: ReLU-like,
    max(N(0,1), 0) (half the entries exactly zero).  clustered: 1024 Gaussian centres N(0,1), rows = centre +
    0.25 N(0,1) (tight clusters: many near-ties around every query)."""
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    out = torch.empty((n, d), dtype=torch.float32, device=dev)
    centres = None
    if kind == "clustered":
        gc = torch.Generator(device=dev)
        gc.manual_seed(99)                      # the same centres on every rank and for the queries
        centres = torch.empty((1024, d), dtype=torch.float32, device=dev).normal_(generator=gc)
    for s in range(0, n, chunk):
        e = min(s + chunk, n)
        blk = out[s:e]
        if kind == "uniform":
            blk.uniform_(generator=gen)
        else:
            blk.normal_(generator=gen)
            if kind == "nonneg":
                blk.clamp_(min=0)
            elif kind == "clustered":
                which = torch.randint(0, 1024, (e - s,), device=dev, generator=gen)
                blk.mul_(0.25).add_(centres[which])
            elif kind == "far_clusters":
                # two tight clusters at +-100 e_0 (spread 0.5): every row is 100 away from the common centre, so a whole
                # cluster sits inside the bf16 filter's slack
                blk.mul_(0.5)
                sign = (torch.randint(0, 2, (e - s,), device=dev, generator=gen).to(torch.float32) * 2.0 - 1.0) * 100.0
                blk[:, 0].add_(sign)
    return out


def bruteforce(args, torch, dist, _lib, world, rank, dev, use_dist, emit):
    n_total, d, k = args.rows, args.dim, args.k
    nq = args.queries_per_gpu * world if args.queries_per_gpu > 0 else args.queries
    metric = _lib.SQ_METRIC_COSINE if args.metric == "cosine" else _lib.SQ_METRIC_L2
    ddt = torch.float64 if args.metric == "cosine" else torch.float32
    # contiguous row shards
    per = (n_total + world - 1) // world
    r0 = min(rank * per, n_total)
    r1 = min(r0 + per, n_total)
    n_local = r1 - r0

    db = make_rows(torch, args.data, n_local, d, dev, 3 + rank)
    if args.metric == "cosine" and args.workload == "c4_cosine_shard":
        db /= db.norm(dim=1, keepdim=True).clamp_min(1e-30)     # rows L2-normalised at load
    max_b = max([nq] + [int(x) for x in args.extra_batches.split(",") if x])
    # `--query-batches` (default 8) DISTINCT query batches, identical on every rank, rotated through the timed loop:
    # step i asks batch i mod 8, so the ~3.3 k rows per query the exact re-rank gathers differ from step to step and
    # come from HBM, as a real query stream's would (one batch repeated finds its rows in the 256 MB MALL)
    nbatch = max(1, args.query_batches)
    all_q = make_rows(torch, args.data, max(max_b, nq * nbatch), d, dev, 1234)
    if args.query_scale != 1.0:
        all_q.mul_(args.query_scale)
    batches = [all_q[j * nq:(j + 1) * nq].contiguous() for j in range(nbatch)]
    queries = batches[0]
    torch.cuda.synchronize()

    # (create-time choices are the index's own: sq_dense_create_opts -- nothing process-wide is touched)
    index = _lib.DenseIndex(db.data_ptr(), n=n_local, d=d, metric=metric, device_ptr=True, id_base=r0, keepalive=db,
                            options={"dense_int8": 0} if args.no_int8 else None)
    build_info = index.info()     # sq_dense_info: build time of sq_dense_create and the bytes each copy keeps resident
    lib_opts = {}
    for item in args.lib_options.split(","):       # measurement: library options of the index, "name=value,..."
        if item:
            name_, val_ = item.split("=")
            index.set_option(name_, int(val_))
            lib_opts[name_] = int(val_)
    # hipEvent timing of the scan inside the library (`roofline.kernel_ms_in_pipeline`): every search on one GPU
    # (`--profile-every`), every 4th on shards -- the event records of a call cost ~10 us of a step, a tenth of a
    # 1.25 M-row shard's.  Options are per handle (sq_handle_set_option): nothing process-wide is touched.
    prof_every = args.profile_every if args.profile_every > 0 else (4 if use_dist else 1)
    index.set_option("profile", prof_every)
    if args.async_streams:
        index.set_option("dense_async_streams", args.async_streams)
    if args.no_mid_tier:
        index.set_option("dense_mid_tier", 0)
    stream = torch.cuda.current_stream().cuda_stream
    use_async = not args.sync_search
    # calls in flight: 3 on shards (the short kernels of a 1.25 M-row shard's call overlap two neighbours' passes: one-rank proxy
    # 0.065 against 0.081 ms per step), 2 on one GPU since round 4 (three launches per call: 0.229-0.232 against 0.235-0.238 ms
    # per step at 20 steps, equal at 100 -- profiles/r04_async_depth.txt)
    depth = args.async_depth if args.async_depth > 0 else (3 if use_dist else 2)
    depth = min(max(depth, 2), 4)
    if use_async:
        index.set_option("dense_async_depth", depth)

    scan_ms, rerank_ms, cands, fallbacks, mids = [], [], [], [], []

    def note_stats():
        st = index.stats()
        if st["scan_launches"]:
            if st["scan_ms"] > 0:          # (0: a call that was not timed, see the profile option above)
                scan_ms.append(st["scan_ms"])
                rerank_ms.append(st["rerank_ms"])
            cands.append(st["candidates"])
            fallbacks.append(st["fallback_queries"])
            mids.append(st["mid_tier_queries"])

    class Runner:
        """`nq_` queries per step through the index (and, with several ranks, the all-gather + host merge)."""

        def __init__(self, nq_, qs_):
            self.nq, self.qs = nq_, qs_          # qs_: the query batches the steps rotate through
            self.n = 0
            self.pipe = None
            if use_dist:
                from smqtk_indexing_amd.distributed import PipelinedShardedSearch
                # the all-gather of a finished batch (asynchronous, RCCL's stream), its copy to pinned memory and the
                # host merge (rank 0, a worker thread) run under the searches of the following batches
                self.pipe = PipelinedShardedSearch(index, nq_, k, ddt, merge_on=0, device=dev, use_async=use_async, depth=depth,
                                                   gather_every=args.gather_every if args.gather_every > 0 else 4,
                                                   wait=bool(args.async_wait if args.async_wait >= 0 else 0),
                                                   queries_ready=True)   # (the queries were generated and synchronised long ago)
            else:
                self.od = [torch.empty((nq_, k), dtype=ddt, device=dev) for _ in range(depth)]   # one per call in flight
                self.oi = [torch.empty((nq_, k), dtype=torch.int64, device=dev) for _ in range(depth)]
                self.i = 0

        def step(self):
            q = self.qs[self.n % len(self.qs)]
            self.n += 1
            if self.pipe is not None:
                res = self.pipe.submit(q)
            else:
                j = self.i % depth
                self.i += 1
                fn = index.search_device_async if use_async else index.search_device
                fn(q.data_ptr(), self.nq, k, self.od[j].data_ptr(), self.oi[j].data_ptr(), stream)
                res = None
            note_stats()        # the last FINISHED call (the previous step when pipelined)
            return res

        def drain(self):
            """Finish what is in flight (searches; with several ranks the gathers and merges too): timed."""
            if self.pipe is not None:
                got = self.pipe.flush()
                note_stats()
                return got[-1] if got else None
            if use_async:
                index.sync()
                note_stats()
            j = (self.i - 1) % depth
            return self.od[j], self.oi[j]

        def close(self):
            if self.pipe is not None:
                self.pipe.close()

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    run = Runner(nq, batches)
    # Pre-roll (untimed, disclosed in config.preroll_ms): ~30 ms of the same pipelined steps before the W warm-up steps.
    # After the index build and the host-side setup the first ~10 ms of back-to-back searches run measurably slower
    # (a 20-step run read 0.44 ms per step where the same process reads 0.41 on its second 20 steps,
    # tools/step_sweep.py REPS=...): with a 9 ms timed region that start-up transient would be most of the figure.
    t_pre = time.perf_counter()
    while args.preroll_ms > 0 and (time.perf_counter() - t_pre) * 1e3 < args.preroll_ms:
        for _ in range(8):
            run.step()
    run.drain()
    for _ in range(args.warmup):
        run.step()
    run.drain()
    run.n = 0
    scan_ms.clear(), rerank_ms.clear(), cands.clear(), fallbacks.clear(), mids.clear()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run.step()
    result = run.drain()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    n_stats = len(scan_ms)
    pipe_scan_ms = float(np.mean(scan_ms)) if scan_ms else 0.0
    pipe_rerank_ms = float(np.mean(rerank_ms)) if rerank_ms else 0.0
    gather_ms = merge_ms = None
    results_lag = (depth - 1 if use_async else 0)
    if run.pipe is not None:
        results_lag = run.pipe.results_lag
        if run.pipe.gather_host_seconds:
            gather_ms = 1e3 * float(np.mean(run.pipe.gather_host_seconds))
        if run.pipe.merging and run.pipe.merger.merge_seconds:
            merge_ms = 1e3 * float(np.mean(run.pipe.merger.merge_seconds))
    # The full-pass kernel's OWN duration: hipEvents (the library's, recorded on the stream the kernel is launched on)
    # around the launch in blocking calls after the timed region, same rotating query batches.  In the pipelined steps
    # above the scan of one call shares the device with the short kernels of its neighbours (and runs on 3/4 of the
    # CUs to leave them room), so its bracketed duration there is longer than its own streaming time -- it can exceed
    # ms_per_step -- and is reported separately as kernel_ms_in_pipeline.  roofline.achieved / frac use THIS one.
    index.sync() if use_async else None
    index.set_option("profile", 1)
    # (the fused int8 call re-ranks inside the full-pass kernel: its per-workgroup clocks split the kernel's duration into
    # the stream and the tail -- sq_stats_t.rerank_ms then is the tail; measurement bit of option dense_debug)
    index.set_option("dense_debug", lib_opts.get("dense_debug", 0) | 8192)
    od1 = torch.empty((nq, k), dtype=ddt, device=dev)
    oi1 = torch.empty((nq, k), dtype=torch.int64, device=dev)
    al, al_rr = [], []
    for j in range(4 + 2 * nbatch):
        qb_ = batches[j % nbatch]
        index.search_device(qb_.data_ptr(), nq, k, od1.data_ptr(), oi1.data_ptr(), stream)
        st_ = index.stats()
        al.append(st_["scan_ms"])
        al_rr.append(st_["rerank_ms"])
        lib_bytes = float(st_["bytes_scanned"])
    head_scan_ms = float(np.mean(al[4:]))
    alone_rerank_ms = float(np.mean(al_rr[4:]))
    index.set_option("profile", prof_every)
    index.set_option("dense_debug", lib_opts.get("dense_debug", 0))
    head_cands = float(np.mean(cands)) / nq if cands else None
    head_fb = int(np.sum(fallbacks)) if fallbacks else 0
    head_mid = int(np.sum(mids)) if mids else 0

    # ---- one rank through the collective path: the merged answer must be the shard's own answer
    if use_dist and world == 1 and result is not None:
        md, mi = result
        od = torch.empty((nq, k), dtype=ddt, device=dev)
        oi = torch.empty((nq, k), dtype=torch.int64, device=dev)
        last_q = batches[(args.steps - 1) % nbatch]          # the batch of the last step
        index.search_device(last_q.data_ptr(), nq, k, od.data_ptr(), oi.data_ptr(), stream)
        assert np.array_equal(mi, oi.cpu().numpy()) and np.array_equal(md, od.cpu().numpy()), \
            "merged result differs from the shard's top-k"
    run.close()

    # ---- parity spot check against the oracle (outside the timed region)
    parity = None
    if not args.no_parity_check and world == 1:
        from oracle import cpu_ref as O
        sub = min(n_local, max(20_000, 2_000_000 * 128 // max(d, 128)))   # (1 GB of rows through the oracle)
        idx_sub = _lib.DenseIndex(db.data_ptr(), n=sub, d=d, metric=metric, device_ptr=True, keepalive=db)
        qn = queries[:2].cpu().numpy()
        gd, gi = idx_sub.search(qn, k)
        dbh = db[:sub].cpu().numpy()
        ok, hits = True, 0
        for j in range(qn.shape[0]):
            if args.metric == "cosine":
                rd, ri = O.dense_topk(dbh, qn[j], k, metric="cosine")
                ok &= bool(np.array_equal(gi[j], ri) and np.allclose(gd[j], rd, rtol=1e-12, atol=0))
            else:
                rd, ri = O.dense_topk(dbh, qn[j], k)
                ok &= bool(np.array_equal(gi[j], ri) and np.array_equal(gd[j].view(np.uint32), rd.view(np.uint32)))
            hits += len(set(gi[j].tolist()) & set(ri.tolist()))
        # the pipelined (asynchronous) calls must give what the blocking call gives
        ga = [torch.empty((2, k), dtype=ddt, device=dev) for _ in range(2)]
        gb = [torch.empty((2, k), dtype=torch.int64, device=dev) for _ in range(2)]
        q2 = queries[:2].contiguous()
        for j in range(3):
            idx_sub.search_device_async(q2.data_ptr(), 2, k, ga[j & 1].data_ptr(), gb[j & 1].data_ptr(), stream)
        idx_sub.sync()
        ok_async = all(np.array_equal(gb[j].cpu().numpy(), gi) and np.array_equal(ga[j].cpu().numpy(), gd) for j in range(2))
        parity = {"rows": sub, "queries": int(qn.shape[0]),
                  "bit_identical_topk" if args.metric == "l2" else "identical_idx_dist_1e-12": ok,
                  "recall_at_k": hits / float(k * qn.shape[0]), "async_equals_sync": bool(ok_async)}
        del dbh
        idx_sub.close()

    # ---- other total batch sizes, same shards, after the timed region (every N: the N = 1 run gives the reference)
    extra = {}
    for b in [int(x) for x in args.extra_batches.split(",") if x]:
        if b == nq:
            continue
        qb = all_q[:b].contiguous()
        rb = Runner(b, [qb])
        reps = 20 if b >= 256 else 40     # (enough steps that the pipeline's fill and drain do not set the figure)
        for _ in range(3):
            rb.step()
        rb.drain()
        scan_ms.clear(), rerank_ms.clear(), cands.clear(), fallbacks.clear()
        fence()
        t1 = time.perf_counter()
        for _ in range(reps):
            rb.step()
        rb.drain()
        fence()
        dt = (time.perf_counter() - t1) / reps
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        rb.close()
        s_ms = float(np.mean(scan_ms)) if scan_ms else 0.0
        qp = 2.0 if b <= 32 else 1.0    # one 32-query tile runs two bfloat16 query planes (q_hi + q_lo), larger batches one
        extra[f"batch_{b}"] = {
            "queries_per_s": b / dt, "ms_per_step": dt * 1e3, "scan_kernel_ms": s_ms,
            "mean_candidates_per_query": float(np.mean(cands)) / b if cands else None,
            "fallback_queries": int(np.sum(fallbacks)) if fallbacks else 0,
            "scan_bf16_TFLOPs_executed": 2 * qp * n_local * (-(-d // 128) * 128) * (-(-b // 32) * 32) / (s_ms * 1e-3) / 1e12 if s_ms > 0 else None,
        }

    # ---- the other two kernels of the hot path on the same resident matrix (BASELINE config C3):
    # ITQ codes of all rows, then Hamming top-k over the unique codes.  Outside the timed region.
    other_paths = None
    if world == 1 and not args.no_other_paths and d % 4 == 0:
        def timed(fn, reps=5):
            fn()
            torch.cuda.synchronize()
            ts = []
            for _ in range(reps):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t1)
            return float(np.median(ts))

        bits = 64
        rot_np, _ = np.linalg.qr(np.random.default_rng(5).standard_normal((d, d)))
        rot = torch.from_numpy(np.ascontiguousarray(rot_np[:, :bits])).to(dev)
        mean = db[:100_000].to(torch.float64).mean(dim=0).contiguous()
        codes = torch.empty((n_local, 1), dtype=torch.int64, device=dev)
        other_paths = {}
        for norm, tag in ((_lib.SQ_NORM_NONE, "itq_hash"), (_lib.SQ_NORM_L2, "itq_hash_normalize2")):
            dt = timed(lambda: _lib.itq_hash_device(db.data_ptr(), 0, n_local, d, mean.data_ptr(), rot.data_ptr(), bits,
                                                    norm, codes.data_ptr(), stream))
            other_paths[tag] = {"rows": n_local, "dim": d, "bits": bits, "ms": dt * 1e3,
                                "GBps_of_algorithmic_bytes": (n_local * d * 4 + n_local * 8) / dt / 1e9,
                                "frac_of_hbm_peak": (n_local * d * 4 + n_local * 8) / dt / 1e9 / HBM_PEAK_GBS}
        ucodes = torch.unique(codes.view(-1), sorted=True).contiguous()   # int64 order != uint64 order: timing only
        hidx = _lib.HammingIndex(ucodes.data_ptr(), n=int(ucodes.numel()), words=1, device_ptr=True, keepalive=ucodes)
        gq = torch.Generator(device=dev)
        gq.manual_seed(77)
        for hq in (1, 32, 1024):
            qc = ucodes[torch.randint(0, ucodes.numel(), (hq,), device=dev, generator=gq)].contiguous()
            hd = torch.empty((hq, k), dtype=torch.int32, device=dev)
            hi_ = torch.empty((hq, k), dtype=torch.int64, device=dev)
            dt = timed(lambda: hidx.search_device(qc.data_ptr(), hq, k, hd.data_ptr(), hi_.data_ptr(), stream))
            other_paths[f"hamming_top{k}_batch_{hq}"] = {"codes": int(ucodes.numel()), "bits": bits, "ms_per_call": dt * 1e3,
                                                        "queries_per_s": hq / dt,
                                                        "fallback_queries": hidx.stats()["fallback_queries"]}
        hidx.close()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = nq * args.steps / elapsed
        cosine = args.metric == "cosine"
        # Bytes one full-pass launch streams: the bfloat16 scan copy (d_pad*2 bytes per row) plus the float32
        # |x|^2 per row for L2.  SURVEY 8(d) prices a pass at N*d*4 bytes (the float32 matrix);
        # the kernel gets the same exact answers from about half of that, so BOTH fractions are printed:
        # `frac` from the bytes really streamed (<= 1 by construction), `frac_survey_8d` from N*d*4 (> 1 means
        # the float32 matrix is not what is being read).
        d_pad = -(-d // 128) * 128
        n_pad = -(-n_local // 32) * 32
        streamed = float(n_pad) * (d_pad * 2 + (0 if cosine else 4))
        # the int8 first-stage filter (L2, d <= 128, one query tile: DESIGN.md 4.1b) streams d_pad + 4 bytes per row; the library
        # reports what its full pass read (sq_stats_t.bytes_scanned) and that is what the roofline prices
        n_pad64 = -(-n_local // 64) * 64
        row8 = 128 if d <= 128 else (256 if d <= 256 else 512)      # the int8 copy's row (sq_dense_i8.hpp i8_row_bytes)
        int8_filter = d <= 512 and lib_bytes == float(n_pad64) * (row8 + 4)
        if int8_filter:
            streamed = lib_bytes
        row_bytes = (row8 + 4) if int8_filter else d_pad * 2 + (0 if cosine else 4)
        f32_bytes = float(n_local) * d * 4
        replay = None
        tpath = os.path.join(ROOT, "profiles", "latest_traffic.json")
        if os.path.isfile(tpath) and n_local == 10_000_000 and d == 128 and not cosine:
            tj = json.load(open(tpath))
            replay = {"hbm_bytes_per_launch": tj.get("dense8_body_full_pass_hbm_bytes" if (int8_filter and nq <= 32 and lib_opts.get("dense_fused", 1) != 0)
                                                     else "dense8_scan_full_pass_hbm_bytes" if int8_filter else "dense_scan_full_pass_hbm_bytes"),
                      "file": "profiles/latest_traffic.json",
                      "note": "NOT measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the committed profile"}
        achieved = streamed / (head_scan_ms * 1e-3) / 1e9 if head_scan_ms > 0 else 0.0
        qp = 2.0 if nq <= 32 else 1.0
        gather_every = (args.gather_every if args.gather_every > 0 else 4) if use_dist else None
        fused_call = int8_filter and nq <= 32 and lib_opts.get("dense_fused", 1) != 0
        bi = build_info
        resident_total = (bi["f32_rows_bytes"] + bi["bf16_copy_bytes"] + bi["int8_copy_bytes"] + bi["row_stats_bytes"])
        line = {
            "metric": (f"queries/sec, exact brute-force {args.metric} kNN k={k} over {n_total}x{d} float32 (recall@{k} = 1.0 by "
                       "construction); scan HBM GB/s vs 8 TB/s peak"),
            "value": value,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "index_build_ms": bi["build_ms"],
            "index_build_note": (f"sq_dense_create on this rank's {n_local} x {d} float32 rows already in HBM (column means, row statistics, "
                                 f"bf16 scan copy, int8 scan copy: {bi['build_int8_ms']:.1f} ms of it); outside the timed region -- the "
                                 "build side of FaissNearestNeighborsIndex._build_index, faiss.py:486-559"),
            "resident_bytes": {"f32_rows": bi["f32_rows_bytes"], "f32_rows_owned_by_library": bi["f32_rows_owned"],
                               "bf16_scan_copy": bi["bf16_copy_bytes"], "int8_scan_copy_and_row_terms": bi["int8_copy_bytes"],
                               "row_statistics": bi["row_stats_bytes"], "total": resident_total,
                               "total_over_f32_rows": resident_total / float(bi["f32_rows_bytes"]) if bi["f32_rows_bytes"] else None},
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": ("f32 results (distances bit-identical to numpy float32); filter arithmetic int8 MFMA (i32 accumulation, measured "
                      "per-row error bound), exact re-rank f32" if int8_filter else
                      "f32 results (distances bit-identical to numpy float32); filter arithmetic bf16 MFMA (x_hi*(q_hi+q_lo)), exact re-rank f32")
                     if not cosine else "f64 results (cosine, 1e-12); filter arithmetic bf16 MFMA, exact re-rank f64",
            "data": f"synthetic {args.data} float32 descriptors generated on device; {nbatch} distinct query batches from the "
                    "same distribution, rotated through the timed steps",
            "config": {
                "workload": f"{args.workload}_{args.metric}_{n_total}x{d}_k{k}",
                "ranks": world,
                "db_rows_total": n_total, "db_rows_per_gpu": n_local, "dim": d, "k": k,
                "queries_per_step": nq,
                "query_batches_rotated": nbatch,
                "scaling_definition": ("strong: fixed rows in total (row-sharded over the ranks) and the same queries per step at every N, N = 1 included"
                                       if args.scaling == "strong" else
                                       "weak: fixed rows PER GPU (the BASELINE config's shard), the same queries per step at every N"),
                "search_calls": f"pipelined (SQ_MEM_DEVICE_ASYNC, {depth} calls in flight)" if use_async else "one blocking call per step",
                "preroll_ms": args.preroll_ms,
                "lib_options": lib_opts or None,
                "results_lag_steps": results_lag,
                "results_lag_note": "steps between a batch going in and its final (N > 1: merged) result coming out; every "
                                    "result of the timed steps is final and collected before the closing fence",
                "sharding": "rows" if world > 1 else "none",
                "collective": (f"one all_gather(top-k dist,idx) per {gather_every} steps + host merge, "
                               "both under the following steps' searches; every merge inside the timed region") if use_dist else "none",
                "gather_host_ms": gather_ms, "merge_ms": merge_ms,
                "mean_candidates_per_query": head_cands,
                "fallback_queries": head_fb,
                "mid_tier_queries": head_mid,
                "steps_with_stats": n_stats,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": None,
                "traffic_replayed": replay,
                "kernel": (f"dense8_body_kernel<{row8 // 32}, false> (full pass over the int8 copy, then the exact re-rank of its survivors as the kernel's tail)" if fused_call
                           else f"dense8_scan_kernel<{row8 // 32}, false> (full pass)" if int8_filter else "dense_scan_kernel (full pass)"),
                "kernel_ms": head_scan_ms,
                "kernel_ms_note": "mean hipEvent-bracketed duration (events recorded by the library on the stream the kernel is "
                                  "launched on) of the full-pass launch in BLOCKING calls right after the timed region, same "
                                  "rotating query batches: the kernel's own streaming time.  achieved = streamed bytes / this.",
                "kernel_ms_in_pipeline": pipe_scan_ms,
                "kernel_ms_in_pipeline_note": "the same bracket INSIDE the timed region: with pipelined calls the scan shares the "
                                              "device with the neighbouring calls' short kernels (and runs on 3/4 of the CUs to "
                                              "leave them room), so this can exceed ms_per_step; not used for achieved / frac",
                "pass_ms": (head_scan_ms - alone_rerank_ms) if fused_call else None,
                "frac_pass": (streamed / ((head_scan_ms - alone_rerank_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS) if (fused_call and head_scan_ms > alone_rerank_ms > 0) else None,
                "pass_note": "fused int8 call: the kernel's duration minus its tail (thresholds + exact re-rank of its survivors), the tail "
                             "measured by the kernel's own per-workgroup clocks (wall_clock64: from the last wave leaving the stream "
                             "to the last workgroup done); frac_pass = streamed bytes / pass_ms / peak -- the stream alone; frac prices the whole kernel",
                "rerank_kernel_ms": alone_rerank_ms, "rerank_kernel_ms_in_pipeline": pipe_rerank_ms,
                "rerank_note": "exact re-rank of the survivors; its rows are cold (distinct query batches: ~3.3 k x nq rows of "
                               "512 B gathered from HBM per step)",
                "bytes_definition": ("achieved/frac: bytes the kernel streams per launch = n_pad64*(row bytes + 4) (int8 scan copy: 128 / 256 / 512 bytes per row, + f32 row terms)"
                                     if int8_filter else
                                     "achieved/frac: bytes the kernel streams per launch = n_pad*(2*d_pad + 4) (bf16 scan copy + f32 norms)"),
                "streamed_bytes_per_launch": streamed,
                "bytes_per_row": row_bytes,
                "first_stage_filter": "int8" if int8_filter else "bf16",
                "algorithmic_bytes_survey_8d": f32_bytes,
                "achieved_survey_8d": f32_bytes / (head_scan_ms * 1e-3) / 1e9 if head_scan_ms > 0 else None,
                "frac_survey_8d": f32_bytes / (head_scan_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if head_scan_ms > 0 else None,
                "frac_survey_8d_note": "SURVEY 8(d) prices a pass at N*d*4 (the float32 matrix); a fraction above 1 says the kernel does not read that matrix: it streams the half-size bf16 copy and only the re-rank touches float32 rows",
                "frac_step": streamed / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "frac_step_note": "streamed bytes / ms_per_step / peak: the whole search step (sample pass, threshold, re-rank, select, launch gaps) priced as if it were the scan",
                "mfma_TFLOPs_executed": 2 * qp * n_pad * d_pad * (-(-nq // 32) * 32) / (head_scan_ms * 1e-3) / 1e12 if head_scan_ms > 0 else None,
                "mfma_peak_TFLOPs": 2 * MFMA_BF16_PEAK_TF if int8_filter else MFMA_BF16_PEAK_TF,
            },
        }
        if parity is not None:
            line["parity_check"] = parity
        if extra:
            line["other_batches"] = extra
        if other_paths:
            line["other_paths"] = other_paths
        if not args.no_cpu_baseline and world == 1:
            rows, cq, dt = cpu_baseline(d, k, 7)
            line["cpu_baseline"] = {
                "value": cq / dt * rows / n_total, "unit": "queries/s", "cores": 1, "kind": "port",
                "sample": f"oracle/cpu_ref.dense_topk (VECTORISED numpy restatement of euclidean_distance + stable sort, not the "
                          f"per-row Python loop of lsh.py:511): {cq} queries x {rows} rows x {d} f32 in {dt:.2f} s on 1 thread "
                          f"(host has {len(os.sched_getaffinity(0))} cores); scaled linearly in rows to {n_total}",
            }
            # what SMQTK executes today -- one Python distance call per candidate row (lsh.py:511)
            from oracle import cpu_ref as O
            rng = np.random.default_rng(7)
            rows_l = max(5_000, 200_000 * 128 // max(d, 128))
            dbl = rng.standard_normal((rows_l, d), dtype=np.float32)
            ql = rng.standard_normal(d, dtype=np.float32)
            t1 = time.perf_counter()
            dl = [O.euclidean_distance(ql, r) for r in dbl]
            sorted(range(rows_l), key=dl.__getitem__)[:k]
            dtl = time.perf_counter() - t1
            line["cpu_baseline_reference_loop"] = {
                "value": 1.0 / dtl * rows_l / n_total, "unit": "queries/s", "cores": 1,
                "kind": "port (per-row Python loop, the shape of lsh.py:505-519)",
                "sample": f"1 query x {rows_l} rows in {dtl:.2f} s; scaled linearly in rows to {n_total}",
            }
            # what a vectorised CPU implementation reaches on the same sample -- |x|^2 - 2 x.q through
            # BLAS on every host thread numpy's BLAS uses, argpartition for the top k.  Not order-exact; a reported
            # baseline only (neither the oracle nor the product).
            rows_v, nq_v = max(20_000, 2_000_000 * 128 // max(d, 128)), 32
            dbv = rng.standard_normal((rows_v, d), dtype=np.float32)
            qv = rng.standard_normal((nq_v, d), dtype=np.float32)
            n2 = np.einsum("ij,ij->i", dbv, dbv)
            t1 = time.perf_counter()
            sc = n2[None, :] - 2.0 * (qv @ dbv.T)
            np.argpartition(sc, k, axis=1)[:, :k]
            dtv = time.perf_counter() - t1
            line["cpu_baseline_vectorised"] = {
                "value": nq_v / dtv * rows_v / n_total, "unit": "queries/s", "cores": len(os.sched_getaffinity(0)),
                "kind": "numpy BLAS + argpartition (not order-exact)",
                "sample": f"{nq_v} queries x {rows_v} rows x {d} f32 in {dtv:.3f} s; scaled linearly in rows to {n_total}",
            }
        emit(line)
    index.close()


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, as CHILD processes of a parent
    that never touches the GPU (torch is not even imported here), exactly the way the driver would
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
    ...`).  Rank 0's one JSON line is relayed to our stdout; the exit code is the launcher's."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:     # a free rendezvous port
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL between processes needs it on these hosts
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, cwd=ROOT)
    line = None
    for raw in proc.stdout:                                  # children write everything but the line to stderr
        text = raw.decode("utf-8", "replace").strip()
        if text.startswith("{") and text.endswith("}"):
            line = text
        elif text:
            print(text, file=sys.stderr)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    if rc == 0 and line is None:
        print("bench.py: the ranks exited without a result line", file=sys.stderr)
        rc = 1
    return rc


def main() -> None:
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    # Only the one JSON line may reach stdout: libraries (RCCL prints a version banner) write to
    # fd 1 too, so fd 1 is pointed at stderr for the run and the line goes to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    from smqtk_indexing_amd import _lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.launch_selftest:
        dist.init_process_group(backend="gloo")
        t = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(t)
        if rank == 0:
            os.write(real_stdout, (json.dumps({"launch_selftest": True, "ranks": dist.get_world_size(),
                                               "sum": int(t.item())}) + "\n").encode())
        dist.destroy_process_group()
        return
    if world != args.gpus and not (args.force_collective and world == 1):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local_rank}, the node has {torch.cuda.device_count()}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_collective
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=dev)

    def emit(line):
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())

    args.scaling = "strong"
    if args.workload == "c4_cosine_shard":
        # BASELINE config 4: 100 M x 512 float32 cosine over 8 GPUs = a 12.5 M x 512 shard per GPU, 256
        # queries per step; every rank generates its shard on the device from seed + rank.  Weak scaling: the shard is
        # the config's, the database grows with N.  (--rows / --dim / --queries still override.)
        args.metric, args.scaling = "cosine", "weak"
        if args.rows == 10_000_000:
            args.rows = 12_500_000 * world
        if args.dim == 128:
            args.dim = 512
        if args.queries == 0:
            args.queries = 256
        if args.extra_batches == "1,64,128,256,1024":
            args.extra_batches = "32"
        args.no_other_paths = True
    if args.queries == 0 and args.workload != "c5_hamming_shard":
        args.queries = 32
    if args.workload == "c5_hamming_shard":
        from tools.hamming_c5 import run as c5_run
        args.scaling = "weak"
        c5_run(args, torch, dist, _lib, world, rank, dev, use_dist, emit)
    elif args.workload == "lsh_c3":
        from tools.lsh_c3 import run as lsh_run
        lsh_run(args, torch, dist, _lib, world, rank, dev, use_dist, emit, make_rows)
    else:
        bruteforce(args, torch, dist, _lib, world, rank, dev, use_dist, emit)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
