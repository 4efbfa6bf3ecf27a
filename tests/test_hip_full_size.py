"""
BASELINE.json's FULL per-GPU sizes on one MI355X.  The headline shape (10 M x 128 L2, k = 100) is compared
LITERALLY with the oracle over all rows for a handful of queries (about 2 s of numpy per query:
`test_dense_l2_10m_x_128_oracle_literal`; cosine over offset descriptors through the middle tier:
`test_dense_cosine_10m_x_128_offset_oracle_literal`); larger batches and the other shapes are checked through
size-independent properties: (a) structural -- ascending distances, a query that is a row finds itself first;
(b) the returned distances recomputed by the oracle from the returned rows (a few hundred rows,
bit-exact); (c) COMPLETENESS against a plain torch evaluation of every distance on the same
device -- integer exact for Hamming and for ITQ's packed codes, a tolerance band around the k-th
distance for the floating-point metrics (torch sums in another order); (d) the partition
property: two half-shards with id offsets, merged on the host, equal the whole index.

Data is generated on the device (torch); nothing here reads /root/reference.
"""
import numpy as np
import pytest

from oracle import cpu_ref as O
from smqtk_indexing_amd import _lib

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _dev():
    return torch.device("cuda", 0)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _popcount64(x):
    """SWAR popcount of int64 tensors taken as 64 raw bits."""
    x = x - ((x >> 1) & 0x5555555555555555)
    x = (x & 0x3333333333333333) + ((x >> 2) & 0x3333333333333333)
    x = (x + (x >> 4)) & 0x0F0F0F0F0F0F0F0F
    return (x * 0x0101010101010101) >> 56


def _search_dense(index, q, k, cosine=False):
    nq = q.shape[0]
    od = torch.empty((nq, k), dtype=torch.float64 if cosine else torch.float32, device=q.device)
    oi = torch.empty((nq, k), dtype=torch.int64, device=q.device)
    index.search_device(q.data_ptr(), nq, k, od.data_ptr(), oi.data_ptr(), _stream())
    torch.cuda.synchronize()
    return od.cpu().numpy(), oi.cpu().numpy()


def test_dense_l2_10m_x_128_properties():
    """The north-star shape: 10 M x 128 float32, k = 100."""
    dev = _dev()
    n, d, k = 10_000_000, 128, 100
    g = torch.Generator(device=dev)
    g.manual_seed(101)
    db = torch.empty((n, d), dtype=torch.float32, device=dev)
    for s in range(0, n, 1 << 21):
        db[s:s + (1 << 21)].normal_(generator=g)
    q = torch.empty((6, d), dtype=torch.float32, device=dev).normal_(generator=g)
    q[0] = db[1_234_567]
    q[1] = db[n - 1] + 0.01
    index = _lib.DenseIndex(db.data_ptr(), n=n, d=d, device_ptr=True, keepalive=db)
    dist, ids = _search_dense(index, q, k)
    assert index.stats()["fallback_queries"] == 0
    # (the pass streamed the int8 copy: 128 + 4 bytes per row)
    assert index.stats()["bytes_scanned"] == (-(-n // 64) * 64) * 132, index.stats()
    qh = q.cpu().numpy()
    assert ids[0, 0] == 1_234_567 and dist[0, 0] == 0.0
    assert ids[1, 0] == n - 1
    for j in range(q.shape[0]):
        assert (np.diff(dist[j]) >= 0).all() and len(set(ids[j].tolist())) == k
        assert ids[j].min() >= 0 and ids[j].max() < n
        rows = db[torch.from_numpy(ids[j]).to(dev)].cpu().numpy()
        # (b) the float32 distances are the reference's own arithmetic on the returned rows
        np.testing.assert_array_equal(dist[j].view(np.uint32), O.dense_distances(rows, qh[j], "euclidean").view(np.uint32))
        # (c) completeness: no row clearly nearer than the k-th is missing (torch float32, other summation order)
        kth = float(dist[j, -1])
        got = torch.from_numpy(ids[j]).to(dev)
        near_total, within = 0, 0
        for s in range(0, n, 1 << 21):
            blk = db[s:s + (1 << 21)]
            dd = torch.sqrt(((blk - q[j]) ** 2).sum(dim=1))
            near = torch.nonzero(dd < kth * (1.0 - 1e-4)).flatten() + s
            near_total += int(near.numel())
            assert bool(torch.isin(near, got).all()), "a row nearer than the k-th neighbour is missing"
            within += int((dd <= kth * (1.0 + 1e-4)).sum())
        assert near_total <= k and within >= k
    # (d) partition: two half-shards with id offsets + host merge == the whole index
    half = n // 2
    lo = _lib.DenseIndex(db.data_ptr(), n=half, d=d, device_ptr=True, keepalive=db)
    hi = _lib.DenseIndex(db[half:].data_ptr(), n=n - half, d=d, device_ptr=True, id_base=half, keepalive=db)
    d0, i0 = _search_dense(lo, q, k)
    d1, i1 = _search_dense(hi, q, k)
    md, mi = _lib.merge_topk(np.stack([d0, d1]), np.stack([i0, i1]), k)
    np.testing.assert_array_equal(mi, ids)
    np.testing.assert_array_equal(md.view(np.uint32), dist.view(np.uint32))
    for h in (index, lo, hi):
        h.close()


def test_dense_l2_10m_x_128_oracle_literal():
    """The headline shape against the oracle itself: ids and float32 distance bits of `O.dense_topk` over ALL 10 M rows
    (the reference's full-order equality, tests/impls/nn_index/test_lsh.py:958-961), through (a) the blocking call,
    (b) the pipelined SQ_MEM_DEVICE_ASYNC calls with the captured graph and rotating batches exactly as bench.py drives
    them, (c) the bf16 first stage (dense_int8 = 0).  Queries: a row, a near-duplicate of a row, four random ones."""
    dev = _dev()
    n, d, k, nq = 10_000_000, 128, 100, 32
    g = torch.Generator(device=dev)
    g.manual_seed(303)
    db = torch.empty((n, d), dtype=torch.float32, device=dev)
    for s in range(0, n, 1 << 21):
        db[s:s + (1 << 21)].normal_(generator=g)
    nb = 4
    batches = [torch.empty((nq, d), dtype=torch.float32, device=dev).normal_(generator=g) for _ in range(nb)]
    batches[0][0] = db[4_321_987]
    batches[1][7] = db[n - 3] * (1.0 + 2e-3)
    checked = [(0, 0), (1, 7), (2, 31), (3, 16), (0, 13), (1, 0)]          # (batch, query in batch)
    dbh = np.empty((n, d), dtype=np.float32)
    for s in range(0, n, 1 << 21):
        dbh[s:s + (1 << 21)] = db[s:s + (1 << 21)].cpu().numpy()
    want = {}
    for b, j in checked:
        want[(b, j)] = O.dense_topk(dbh, batches[b][j].cpu().numpy(), k)
    del dbh
    assert want[(0, 0)][1][0] == 4_321_987 and want[(0, 0)][0][0] == 0.0 and want[(1, 7)][1][0] == n - 3

    def compare(tag, b, dist, ids):
        for (bb, j), (rd, ri) in want.items():
            if bb != b:
                continue
            np.testing.assert_array_equal(ids[j], ri, err_msg=f"{tag}: ids of batch {b} query {j}")
            np.testing.assert_array_equal(dist[j].view(np.uint32), rd.view(np.uint32), err_msg=f"{tag}: distances of batch {b} query {j}")

    index = _lib.DenseIndex(db.data_ptr(), n=n, d=d, device_ptr=True, keepalive=db)
    # (a) blocking calls
    for b in range(nb):
        dist, ids = _search_dense(index, batches[b], k)
        assert index.stats()["fallback_queries"] == 0
        assert index.stats()["bytes_scanned"] == (-(-n // 64) * 64) * 132       # the int8 first stage answered
        compare("blocking", b, dist, ids)
    # (b) pipelined calls, three in flight, the call graph captured: bench.py's timed loop
    depth, steps = 3, 24
    index.set_option("dense_graph", 1)
    index.set_option("dense_async_depth", depth)
    od = [torch.empty((nq, k), dtype=torch.float32, device=dev) for _ in range(depth)]
    oi = [torch.empty((nq, k), dtype=torch.int64, device=dev) for _ in range(depth)]
    got = {}
    for i in range(steps):
        slot = i % depth
        if i >= depth:       # the results of step i - depth are final once call i - 1 has returned (include/smqtk_hip.h)
            got[(i - depth) % nb] = (od[slot].cpu().numpy().copy(), oi[slot].cpu().numpy().copy())
        index.search_device_async(batches[i % nb].data_ptr(), nq, k, od[slot].data_ptr(), oi[slot].data_ptr(), _stream())
    index.sync()
    torch.cuda.synchronize()
    for i in range(steps - depth, steps):
        got[i % nb] = (od[i % depth].cpu().numpy().copy(), oi[i % depth].cpu().numpy().copy())
    assert sorted(got) == list(range(nb))
    for b in range(nb):
        compare("pipelined+graph", b, *got[b])
    # (c) the bf16 first stage
    index.set_option("dense_int8", 0)
    for b in range(nb):
        dist, ids = _search_dense(index, batches[b], k)
        assert index.stats()["fallback_queries"] == 0
        assert index.stats()["bytes_scanned"] != (-(-n // 64) * 64) * 132
        compare("bf16 filter", b, dist, ids)
    index.close()


def test_dense_cosine_10m_x_128_offset_oracle_literal():
    """Cosine at the headline size over descriptors that share a 50-sigma offset (every list of the first filters
    overflows) against the oracle itself over ALL 10 M rows (metrics.cosine_distance, smqtk_indexing/utils/metrics.py:120-137:
    float64 distances within 1e-12, ids equal wherever the reference's distances differ): through the cosine middle tier behind
    the overflowing first filters, then -- the filters suspended after three such calls -- with calls starting at the tier,
    blocking and pipelined (SQ_MEM_DEVICE_ASYNC, three calls in flight).  Queries: a stored row scaled by 2.5 (distance 0
    up to rounding), three random ones about the same offset."""
    dev = _dev()
    n, d, k, nq = 10_000_000, 128, 100, 32
    g = torch.Generator(device=dev)
    g.manual_seed(404)
    off = torch.empty((d,), dtype=torch.float32, device=dev).normal_(generator=g) * 50.0
    db = torch.empty((n, d), dtype=torch.float32, device=dev)
    for s in range(0, n, 1 << 21):
        db[s:s + (1 << 21)].normal_(generator=g).add_(off)
    nb = 2
    batches = [torch.empty((nq, d), dtype=torch.float32, device=dev).normal_(generator=g).add_(off) for _ in range(nb)]
    batches[0][5] = db[7_654_321] * 2.5
    checked = [(0, 5), (0, 0), (1, 31), (1, 9)]
    dbh = np.empty((n, d), dtype=np.float32)
    for s in range(0, n, 1 << 21):
        dbh[s:s + (1 << 21)] = db[s:s + (1 << 21)].cpu().numpy()
    want = {}
    full = {}
    for b, j in checked:
        dist_all = O.dense_distances(dbh, batches[b][j].cpu().numpy(), "cosine")
        order = np.argsort(dist_all, kind="stable")[:k]
        want[(b, j)] = (dist_all[order], order.astype(np.int64))
        full[(b, j)] = dist_all
    del dbh
    assert want[(0, 5)][1][0] == 7_654_321 and want[(0, 5)][0][0] < 1e-7

    def compare(tag, b, dist, ids):
        for (bb, j), (rd, ri) in want.items():
            if bb != b:
                continue
            np.testing.assert_allclose(dist[j], rd, rtol=1e-12, atol=1e-15, err_msg=f"{tag}: distances of batch {b} query {j}")
            mism = ids[j] != ri
            if mism.any():   # only among distances the reference itself cannot tell apart
                assert np.abs(full[(bb, j)][ids[j][mism]] - full[(bb, j)][ri[mism]]).max() < 1e-14, f"{tag}: ids of batch {b} query {j}"

    index = _lib.DenseIndex(db.data_ptr(), n=n, d=d, metric=_lib.SQ_METRIC_COSINE, device_ptr=True, keepalive=db)
    # (a) blocking calls: the first ones run the (overflowing) first filters in front of the tier, the later ones start at it
    direct = 0
    for i in range(10):
        b = i % nb
        dist, ids = _search_dense(index, batches[b], k, cosine=True)
        st = index.stats()
        assert st["mid_tier_queries"] == nq and st["fallback_queries"] == 0, st
        direct += st["candidates"] == 0
        compare(f"blocking call {i}", b, dist, ids)
    assert direct >= 3 and st["candidates"] == 0 and st["bytes_scanned"] == n * d * 4      # one pass over the float32 rows
    # (b) pipelined calls, three in flight
    depth, steps = 3, 12
    index.set_option("dense_async_depth", depth)
    od = [torch.empty((nq, k), dtype=torch.float64, device=dev) for _ in range(depth)]
    oi = [torch.empty((nq, k), dtype=torch.int64, device=dev) for _ in range(depth)]
    got = {}
    for i in range(steps):
        slot = i % depth
        if i >= depth:
            got[(i - depth) % nb] = (od[slot].cpu().numpy().copy(), oi[slot].cpu().numpy().copy())
        index.search_device_async(batches[i % nb].data_ptr(), nq, k, od[slot].data_ptr(), oi[slot].data_ptr(), _stream())
    index.sync()
    torch.cuda.synchronize()
    for i in range(steps - depth, steps):
        got[i % nb] = (od[i % depth].cpu().numpy().copy(), oi[i % depth].cpu().numpy().copy())
    for b in range(nb):
        compare("pipelined", b, *got[b])
    index.close()


def test_dense_l2_10m_batches_of_every_kernel_agree():
    """10 M x 128, 1024 queries: the bf16 multi-tile scans (4 query tiles per wave for 1024 and 256 queries), the int8
    two-tile scan (64 queries) and the int8 one-tile scan (32 queries) are different kernels -- and different filters --
    behind the same exact answer: ids and float32 distances must agree bit for bit, and no query may need the exact path
    (also BASELINE config 2's batch, at full rows)."""
    dev = _dev()
    n, d, k = 10_000_000, 128, 100
    g = torch.Generator(device=dev)
    g.manual_seed(202)
    db = torch.empty((n, d), dtype=torch.float32, device=dev)
    for s in range(0, n, 1 << 21):
        db[s:s + (1 << 21)].normal_(generator=g)
    q = torch.empty((1024, d), dtype=torch.float32, device=dev).normal_(generator=g)
    q[5] = db[777_777]
    index = _lib.DenseIndex(db.data_ptr(), n=n, d=d, device_ptr=True, keepalive=db)

    def run(chunk):
        od = torch.empty((1024, k), dtype=torch.float32, device=dev)
        oi = torch.empty((1024, k), dtype=torch.int64, device=dev)
        fallbacks = 0
        for s in range(0, 1024, chunk):
            index.search_device(q[s:s + chunk].data_ptr(), chunk, k, od[s:s + chunk].data_ptr(), oi[s:s + chunk].data_ptr(),
                                _stream())
            fallbacks += index.stats()["fallback_queries"]
        torch.cuda.synchronize()
        return od.cpu().numpy(), oi.cpu().numpy(), fallbacks

    ref_d, ref_i, ref_fb = run(32)
    assert ref_fb == 0 and ref_i[5, 0] == 777_777 and ref_d[5, 0] == 0.0
    assert (np.diff(ref_d, axis=1) >= 0).all()
    for chunk in (64, 256, 1024):
        dd, ii, fb = run(chunk)
        assert fb == 0, chunk
        np.testing.assert_array_equal(ii, ref_i, err_msg=f"batch {chunk}")
        np.testing.assert_array_equal(dd.view(np.uint32), ref_d.view(np.uint32), err_msg=f"batch {chunk}")
    # the returned distances recomputed by the oracle from the returned rows (two queries)
    for qi in (5, 1000):
        rows = db[torch.from_numpy(ref_i[qi]).to(dev)].cpu().numpy()
        want = O.dense_distances(rows, q[qi].cpu().numpy(), "euclidean")
        np.testing.assert_array_equal(ref_d[qi].view(np.uint32), want.astype(np.float32).view(np.uint32))
    index.close()


@pytest.mark.parametrize("d", [256, 512])
def test_dense_l2_wide_rows_int8_equals_bf16(d):
    """4 M x 256 / 512 float32: the int8 first stage over 256- and 512-byte rows (32-row ring units; four waves per
    workgroup at 512) against the bf16 filter on the same index -- the same bits -- and the oracle's arithmetic on the
    returned rows."""
    dev = _dev()
    n, k, nq = 4_000_000, 100, 32
    g = torch.Generator(device=dev)
    g.manual_seed(300 + d)
    db = torch.empty((n, d), dtype=torch.float32, device=dev)
    for s in range(0, n, 1 << 20):
        db[s:s + (1 << 20)].normal_(generator=g)
    q = torch.empty((nq, d), dtype=torch.float32, device=dev).normal_(generator=g)
    q[3] = db[3_999_999]
    index = _lib.DenseIndex(db.data_ptr(), n=n, d=d, device_ptr=True, keepalive=db)
    d8, i8 = _search_dense(index, q, k)
    st = index.stats()
    assert st["bytes_scanned"] == (-(-n // 64) * 64) * (d + 4) and st["fallback_queries"] == 0 and st["mid_tier_queries"] == 0, st
    index.set_option("dense_int8", 0)
    d16, i16 = _search_dense(index, q, k)
    assert index.stats()["bytes_scanned"] == (-(-n // 32) * 32) * (2 * d + 4)
    np.testing.assert_array_equal(i8, i16)
    np.testing.assert_array_equal(d8.view(np.uint32), d16.view(np.uint32))
    assert i8[3, 0] == 3_999_999 and d8[3, 0] == 0.0
    qh = q.cpu().numpy()
    for j in (0, 3, 31):
        rows = db[torch.from_numpy(i8[j]).to(dev)].cpu().numpy()
        np.testing.assert_array_equal(d8[j].view(np.uint32), O.dense_distances(rows, qh[j], "euclidean").view(np.uint32))
    index.close()


def test_dense_cosine_12m_x_512_shard_properties():
    """One shard of BASELINE config 4 (100 M x 512 cosine over 8 GPUs): 12.5 M x 512."""
    dev = _dev()
    n, d, k = 12_500_000, 512, 100
    g = torch.Generator(device=dev)
    g.manual_seed(102)
    db = torch.empty((n, d), dtype=torch.float32, device=dev)
    for s in range(0, n, 1 << 20):
        db[s:s + (1 << 20)].normal_(generator=g)
    q = torch.empty((3, d), dtype=torch.float32, device=dev).normal_(generator=g)
    q[0] = db[7_654_321] * 3.0            # same direction: distance ~0
    index = _lib.DenseIndex(db.data_ptr(), n=n, d=d, metric=_lib.SQ_METRIC_COSINE, device_ptr=True, keepalive=db)
    dist, ids = _search_dense(index, q, k, cosine=True)
    assert index.stats()["fallback_queries"] == 0
    assert index.stats()["bytes_scanned"] == (-(-n // 64) * 64) * 516, index.stats()   # (the int8 copy of the unit rows: 512 + 4 bytes)
    qh = q.cpu().numpy()
    assert ids[0, 0] == 7_654_321 and dist[0, 0] < 1e-6
    for j in range(q.shape[0]):
        assert (np.diff(dist[j]) >= 0).all() and len(set(ids[j].tolist())) == k
        rows = db[torch.from_numpy(ids[j]).to(dev)].cpu().numpy()
        np.testing.assert_allclose(dist[j], O.dense_distances(rows, qh[j], "cosine"), rtol=1e-12, atol=1e-15)
        # completeness against a float64 torch evaluation of the reference formula 2 acos(sim) / pi
        kth = float(dist[j, -1])
        got = torch.from_numpy(ids[j]).to(dev)
        q64 = q[j].double()
        qn = torch.sqrt((q64 * q64).sum())
        near_total, within = 0, 0
        for s in range(0, n, 1 << 20):
            blk = db[s:s + (1 << 20)].double()
            sim = (blk @ q64) / (torch.sqrt((blk * blk).sum(dim=1)) * qn)
            dd = 2.0 * torch.arccos(sim.clamp(-1.0, 1.0)) / np.pi
            near = torch.nonzero(dd < kth - 1e-9).flatten() + s
            near_total += int(near.numel())
            assert bool(torch.isin(near, got).all()), "a row nearer than the k-th neighbour is missing"
            within += int((dd <= kth + 1e-9).sum())
        assert near_total <= k and within >= k
    index.close()


def test_hamming_125m_x_256bit_shard_exact():
    """One shard of BASELINE config 5 (1 B x 256-bit codes over 8 GPUs): 125 M codes; the whole
    answer is integer exact against torch (xor, popcount, top-k of (distance, row) keys)."""
    dev = _dev()
    n, w, k = 125_000_000, 4, 100
    g = torch.Generator(device=dev)
    g.manual_seed(103)
    codes = torch.empty((n, w), dtype=torch.int64, device=dev)
    for s in range(0, n, 1 << 24):
        e = min(n, s + (1 << 24))
        codes[s:e] = torch.randint(-2 ** 63, 2 ** 63 - 1, (e - s, w), dtype=torch.int64, device=dev, generator=g)
    q = torch.randint(-2 ** 63, 2 ** 63 - 1, (3, w), dtype=torch.int64, device=dev, generator=g)
    q[0] = codes[77_777_777]
    q[1] = codes[5]
    q[1, 3] ^= 1                          # one bit away from row 5
    index = _lib.HammingIndex(codes.data_ptr(), n=n, words=w, device_ptr=True, id_base=1_000, keepalive=codes)
    od = torch.empty((3, k), dtype=torch.int32, device=dev)
    oi = torch.empty((3, k), dtype=torch.int64, device=dev)
    index.search_device(q.data_ptr(), 3, k, od.data_ptr(), oi.data_ptr(), _stream())
    torch.cuda.synchronize()
    assert index.stats()["fallback_queries"] == 0
    assert int(oi[0, 0]) == 77_777_777 + 1_000 and int(od[0, 0]) == 0
    assert int(oi[1, 0]) == 5 + 1_000 and int(od[1, 0]) == 1
    rows = torch.arange(n, dtype=torch.int64, device=dev)
    for j in range(3):
        dist = torch.zeros(n, dtype=torch.int64, device=dev)
        for c in range(w):
            dist += _popcount64(codes[:, c] ^ q[j, c])
        key = (dist << 32) | rows
        want = torch.topk(key, k, largest=False, sorted=True).values
        assert torch.equal(od[j].to(torch.int64), want >> 32)
        assert torch.equal(oi[j], (want & 0xFFFFFFFF) + 1_000)
        del dist, key
    index.close()


@pytest.mark.parametrize("normalize", [None, 2])
def test_itq_10m_x_128_codes_equal_float64_torch(normalize):
    """BASELINE config 3's hashing: 10 M x 128 float32 -> 64-bit codes, every row against a float64
    torch evaluation of (norm(x) - mean) . R (rows with a |z| below 1e-9 of its scale excluded:
    there the reference's own sign depends on its BLAS summation order)."""
    dev = _dev()
    n, d, bits = 10_000_000, 128, 64
    g = torch.Generator(device=dev)
    g.manual_seed(104)
    x = torch.empty((n, d), dtype=torch.float32, device=dev)
    for s in range(0, n, 1 << 21):
        x[s:s + (1 << 21)].normal_(generator=g)
    x[12_345] = 0.0                        # a zero row: z = -mean . R
    rot_np, _ = np.linalg.qr(np.random.default_rng(9).standard_normal((d, d)))
    rot = torch.from_numpy(np.ascontiguousarray(rot_np[:, :bits])).to(dev)
    mean = (x[:200_000].double() / (torch.linalg.norm(x[:200_000].double(), dim=1, keepdim=True) if normalize else 1.0)).mean(dim=0).contiguous()
    out = torch.empty((n, 1), dtype=torch.int64, device=dev)
    _lib.itq_hash_device(x.data_ptr(), 0, n, d, mean.data_ptr(), rot.data_ptr(), bits,
                         _lib.SQ_NORM_L2 if normalize else _lib.SQ_NORM_NONE, out.data_ptr(), _stream())
    torch.cuda.synchronize()
    shifts = (63 - torch.arange(bits, device=dev)).to(torch.int64)
    bad = 0
    for s in range(0, n, 1 << 20):
        blk = x[s:s + (1 << 20)]
        if normalize:
            nrm = torch.linalg.norm(blk, dim=1, keepdim=True)        # float32 norm, like numpy on float32 rows
            nrm = torch.where(nrm == 0, torch.ones_like(nrm), nrm)
            v = (blk / nrm).double() - mean
        else:
            v = blk.double() - mean
        z = v @ rot
        code = ((z >= 0).to(torch.int64) << shifts).sum(dim=1)       # MSB first; wraps into the sign bit like the packed word
        differ = code != out[s:s + (1 << 20), 0]
        if bool(differ.any()):
            zz = z[differ]
            scale = zz.abs().max(dim=1).values.clamp_min(1e-30)
            # a differing row must owe it to a borderline bit.  normalize=2: torch's float32 norm may be an ulp
            # off numpy's pairwise one, which moves z by ~1e-7 of its scale
            assert bool((zz.abs().min(dim=1).values <= (1e-6 if normalize else 1e-9) * scale).all())
            bad += int(differ.sum())
    assert bad <= (3000 if normalize else 4)
    if normalize:
        # the 3000 above is torch's float32 norm, not the kernel: against numpy's own norm (pairwise float32 sum,
        # itq.py:185 -- the order itq_norms_kernel claims) on a 1 M-row slice the codes agree as in the other case
        m = 1 << 20
        xs = x[:m].cpu().numpy()
        nrm = np.linalg.norm(xs, ord=2, axis=1, keepdims=True)
        nrm[nrm == 0] = 1.0
        v = torch.from_numpy(xs / nrm).to(dev).double() - mean          # float32 division, float64 subtraction (itq.py:404)
        z = v @ rot
        code = ((z >= 0).to(torch.int64) << shifts).sum(dim=1)
        differ = code != out[:m, 0]
        if bool(differ.any()):
            zz = z[differ]
            scale = zz.abs().max(dim=1).values.clamp_min(1e-30)
            assert bool((zz.abs().min(dim=1).values <= 1e-9 * scale).all())
        assert int(differ.sum()) <= 4


def test_c2_literal_1m_x_128_1024_queries():
    """BASELINE config 2 at its literal shape: 1 M x 128 float32 N(0,1), 1024 queries, k = 100 -- the
    sample stride and the select sizing are functions of n, so the 10 M-row tests do not cover it.  32 of the
    queries against the oracle over the whole matrix (ids and float32 distances bit for bit), no query on the exact
    path, and the 32-query kernel agreeing with the 1024-query kernel on all of them."""
    dev = _dev()
    n, d, k, nq = 1_000_000, 128, 100, 1024
    g = torch.Generator(device=dev)
    g.manual_seed(2)
    db = torch.empty((n, d), dtype=torch.float32, device=dev).normal_(generator=g)
    q = torch.empty((nq, d), dtype=torch.float32, device=dev).normal_(generator=g)
    q[7] = db[123_456]
    index = _lib.DenseIndex(db.data_ptr(), n=n, d=d, device_ptr=True, keepalive=db)
    dist, ids = _search_dense(index, q, k)
    assert index.stats()["fallback_queries"] == 0
    assert ids[7, 0] == 123_456 and dist[7, 0] == 0.0
    dbh, qh = db.cpu().numpy(), q.cpu().numpy()
    for j in list(range(0, 1024, 64)) + list(range(5, 21)):
        rd, ri = O.dense_topk(dbh, qh[j], k)
        np.testing.assert_array_equal(ids[j], ri, err_msg=f"query {j}")
        np.testing.assert_array_equal(dist[j].view(np.uint32), rd.view(np.uint32), err_msg=f"query {j}")
    fb = 0
    for s in range(0, nq, 32):
        d32, i32 = _search_dense(index, q[s:s + 32].contiguous(), k)
        fb += index.stats()["fallback_queries"]
        np.testing.assert_array_equal(i32, ids[s:s + 32])
        np.testing.assert_array_equal(d32.view(np.uint32), dist[s:s + 32].view(np.uint32))
    assert fb == 0
    index.close()


def test_c3_hamming_topk_on_10m_real_itq_codes():
    """BASELINE config 3's Hamming stage on REAL codes: hash 10 M x 128 descriptors to 64 bits, index the unique codes
    (row id = rank in unsigned order), top-100 for 32 and for 1024 query codes; integer exact against torch's top-k of
    (distance, row) keys, no query on the exact path.  (ITQ codes of Gaussian rows are not uniform bit strings: the
    tie groups and the sampled thresholds differ from the random codes of the other tests.)"""
    dev = _dev()
    n, d, bits, k = 10_000_000, 128, 64, 100
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    x = torch.empty((n, d), dtype=torch.float32, device=dev)
    for s in range(0, n, 1 << 21):
        x[s:s + (1 << 21)].normal_(generator=g)
    rot_np, _ = np.linalg.qr(np.random.default_rng(5).standard_normal((d, d)))
    rot = torch.from_numpy(np.ascontiguousarray(rot_np[:, :bits])).to(dev)
    mean = x[:100_000].double().mean(dim=0).contiguous()
    codes = torch.empty((n, 1), dtype=torch.int64, device=dev)
    _lib.itq_hash_device(x.data_ptr(), 0, n, d, mean.data_ptr(), rot.data_ptr(), bits, _lib.SQ_NORM_NONE, codes.data_ptr(),
                         _stream())
    torch.cuda.synchronize()
    top = torch.tensor(-0x8000000000000000, dtype=torch.int64, device=dev)
    ucodes = (torch.unique(codes.view(-1) ^ top, sorted=True) ^ top).contiguous()     # ascending as UNSIGNED integers
    del x, codes
    m = int(ucodes.numel())
    assert m > 9_000_000
    hidx = _lib.HammingIndex(ucodes.data_ptr(), n=m, words=1, device_ptr=True, keepalive=ucodes)
    gq = torch.Generator(device=dev)
    gq.manual_seed(77)
    qc = ucodes[torch.randint(0, m, (1024,), device=dev, generator=gq)].contiguous()
    qc[1] ^= 5                                         # two bits away from an indexed code
    rows = torch.arange(m, dtype=torch.int64, device=dev)

    def check(sel, od, oi):
        for j in sel:
            key = (_popcount64(ucodes ^ qc[j]) << 32) | rows
            want = torch.topk(key, k, largest=False, sorted=True).values
            assert torch.equal(od[j].to(torch.int64), want >> 32), j
            assert torch.equal(oi[j], want & 0xFFFFFFFF), j

    od32 = torch.empty((32, k), dtype=torch.int32, device=dev)
    oi32 = torch.empty((32, k), dtype=torch.int64, device=dev)
    hidx.search_device(qc.data_ptr(), 32, k, od32.data_ptr(), oi32.data_ptr(), _stream())
    torch.cuda.synchronize()
    assert hidx.stats()["fallback_queries"] == 0
    check(range(32), od32, oi32)
    od = torch.empty((1024, k), dtype=torch.int32, device=dev)
    oi = torch.empty((1024, k), dtype=torch.int64, device=dev)
    hidx.search_device(qc.data_ptr(), 1024, k, od.data_ptr(), oi.data_ptr(), _stream())
    torch.cuda.synchronize()
    assert hidx.stats()["fallback_queries"] == 0
    assert torch.equal(od[:32], od32) and torch.equal(oi[:32], oi32)
    check(range(32, 1024, 16), od, oi)
    hidx.close()
