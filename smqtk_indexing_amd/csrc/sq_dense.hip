// Exact brute-force L2 / cosine top-k over a float32 descriptor matrix (gfx950).
//
// What it replaces: the per-candidate python distance calls + stable sort of
// LSHNearestNeighborIndex._nn (smqtk_indexing/impls/nn_index/lsh.py:505-519)
// and the faiss "IDMap,Flat" search + recompute of
// FaissNearestNeighborsIndex._nn (impls/nn_index/faiss.py:751-831), both of
// which evaluate metrics.euclidean_distance / cosine_distance
// (utils/metrics.py:73-86, 89-137) for every row and keep the n smallest.
//
// Structure:
//   1. dense_scan_kernel (sq_dense_scan.hpp) streams a bfloat16 copy of the
//      matrix (L2: of the rows minus their column means) once per group of 1, 2
//      or 4 32-query tiles: LDS-DMA ring per wave, bf16 MFMA (x_hi*q_hi
//      [+ x_hi*q_lo]), scores s = |x|^2 - 2 x.q (cosine: -x^.q^) compared with a
//      per-query threshold; survivors leave as per-wave lists of (first row,
//      mask, query) entries.  The threshold comes from the same kernel in SAMPLE
//      mode over every S-th tile + kth_threshold_f32_kernel.
//   1b. calls of up to 32 queries over rows of up to 512 dimensions stream an int8 copy instead (dense8_scan_kernel,
//      sq_dense_i8.hpp: one scale per matrix, residuals measured per row at build): half the bytes, same downstream.
//   2. dense_rerank_*_kernel (sq_dense_exact.hpp) recomputes the distance of every
//      survivor from the ORIGINAL float32 rows in the REFERENCE arithmetic
//      (float32 subtract, square, numpy pairwise order, correctly rounded sqrt;
//      cosine in float64, scipy's order) and forms (distance, row) keys.
//   3. select_topk_kernel sorts the keys; its post-op (DenseFinalize*) converts and
//      CERTIFIES each query against the filter's error bound.  Queries that fail
//      (or overflow their list) are redone on the exact full-keys path.
#include <algorithm>
#include <vector>
#include <cmath>

#include "sq_dense_exact.hpp"
#include "sq_dense_scan.hpp"
#include "sq_dense_mid.hpp"
#include "sq_dense_i8.hpp"
#include "sq_dense_wide.hpp"
#include "sq_dense_tighten.hpp"

namespace sq {

// Per-call workspace.  A synchronous search uses slot 0; asynchronous searches (SQ_MEM_DEVICE_ASYNC)
// alternate between the two slots so that the kernels of call i + 1 are enqueued -- and, with the slots'
// own streams, partly run -- while call i is still on the device; the status words of call i are read when
// call i + 1 has been enqueued (dense_resolve).
struct DenseCall {
    bool pending = false;
    const float* q = nullptr;
    int nq = 0, k = 0, nq_pad = 0;
    void* out_dist = nullptr;
    long long* out_idx = nullptr;
    hipStream_t st = nullptr;     // the stream the call's kernels were enqueued on
    bool small = false, all_fallback = false, prof = false, use_event = false, int8 = false;
    bool mid_direct = false;    // no first filter ran (it is suspended: DenseHandle::first_suspended): every query starts at the middle tier
    u32 cap = 0;                  // candidate-list length the call was enqueued with
    sq_stats_t stats{};
};
struct DenseSlot {
    DevBuf q_scaled, q_al, qn2, thr, wave_out, wave_cnt, cnt, keys, sample, out_keys, cos_nq, oflag;
    DevBuf q8, par8;              // int8 filter: the query tile's plane and {score unit, e_q}
    DevBuf clk8;                  // measurement ("dense_debug" & 8192): the body kernel's per-workgroup clocks
    int clk8_wgs = 0;
    DevBuf wave_score, hist8;     // ... the fused call's tightened threshold: an entry's smallest score, the per-query histograms
    DevBuf tg;                    // the wide-row path's second-level threshold (sq_dense_tighten.hpp): [hist | raw thresholds | T'' | keys of T'']
    void* tg_zeroed = nullptr;    // the allocation of tg that has been wiped once (its histogram is zero between calls)
    DevBuf sort_tmp;              // scratch of the any-k sorted select (k beyond the one-workgroup select): one per call in flight
    HostPinned status_host;
    // captured call graph of the int8 path ("dense_graph"): one hipGraphLaunch instead of six kernel launches per call
    HostPinned call_ptrs;         // DenseCallPtrs of the call in flight
    hipGraphExec_t gexec = nullptr;
    u64 gkey = 0, seen_key = 0;   // what gexec was captured for / the key of the slot's last eager call
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // call start, scan start, scan end, call end, re-rank end
    hipEvent_t ev_in = nullptr, ev_done = nullptr;
    hipStream_t own = nullptr;    // internal stream of the slot (asynchronous calls with "dense_async_streams" = 2)
    DenseCall call;
    void release() {
        for (DevBuf* b : {&q_scaled, &q_al, &qn2, &thr, &wave_out, &wave_cnt, &cnt, &keys, &sample, &out_keys, &cos_nq, &oflag, &sort_tmp, &q8, &par8, &wave_score, &hist8, &clk8, &tg})
            b->release();
        status_host.release();
        call_ptrs.release();
        if (gexec) (void)hipGraphExecDestroy(gexec), gexec = nullptr;
        gkey = seen_key = 0;
        for (auto& e : ev)
            if (e) (void)hipEventDestroy(e), e = nullptr;
        if (ev_in) (void)hipEventDestroy(ev_in), ev_in = nullptr;
        if (ev_done) (void)hipEventDestroy(ev_done), ev_done = nullptr;
        if (own) (void)hipStreamDestroy(own), own = nullptr;
    }
};

struct DenseHandle : HandleBase {
    const float* db = nullptr;  // device [n][ld], the caller's float32 rows (borrowed or owned)
    DevBuf owned;
    DevBuf scan;                // bfloat16 scan copy [n_pad][d_pad*2 bytes]
    DevBuf norms;               // float32 |x - c|^2 [n_pad] (cosine: of the rows themselves)
    DevBuf center;              // float32 c [d_pad]: column means (L2), the filter's origin
    DevBuf cos_nx;              // float64 |x|^2 [n] in the reference order (cosine re-rank)
    long long n = 0, n_pad = 0;
    int d = 0, d_pad = 0;
    long long ld = 0;
    int metric = SQ_METRIC_L2;
    long long id_base = 0;
    double xn2_max = 0.0;       // max squared row norm (error bound of the L2 filter)
    // the int8 first-stage filter (sq_dense_i8.hpp): copy, row terms, and what the build measured
    DevBuf scan8, nrow8;
    bool use8 = false;          // the int8 copy exists (and follows appends)
    bool suspended8 = false;    // ... but automatic mode (dense_int8 = -1) leaves it alone: its lists overflowed three calls in a row.  dense_int8 = 1 re-arms it, a rebuild (the index doubled) too
    int row8 = 0;               // bytes per row of the copy: 128, 256 or 512
    long long n_pad64 = 0;      // rows a pass covers (what sq_stats_t.bytes_scanned prices)
    long long n_alloc8 = 0;     // rows the copy is allocated and padded for: a multiple of 128 (the largest ring unit)
    double dx8 = 0.0, rmax8 = 0.0, xmax8 = 0.0;
    long long flagged8 = 0;
    int overflow8 = 0;          // calls in a row in which the int8 filter's lists overflowed (data it does not suit): it is dropped
    bool graph_broken = false;  // a call-graph capture failed on this handle: eager launches from then on
    float dxf8 = 0.f, inv_dxf8 = 0.f, cut8 = 0.f;   // the build's step and residual cut, for rows appended later
    long long build_us = 0, build8_us = 0;   // sq_dense_info: wall time of sq_dense_create / of its int8 part
    long long n8_built = 0;     // rows the clamp was chosen from (an index twice that size chooses again)
    DevBuf norms1;  // L2: |x|^2 (1 - alpha) for one query plane (`norms`: two planes)
    DevBuf zeros;   // cosine: the 32 zero "norms" every tile of an AGPR-configuration scan starts from (norm_step 0)
    static constexpr int kMaxDepth = 6;
    DenseSlot slot[kMaxDepth];
    int depth = 2;                       // asynchronous calls in flight (option dense_async_depth, fixed while any is)
    unsigned long long async_calls = 0;  // asynchronous calls so far (slot = calls % depth)
    // workspace shared by all calls: host-memory staging, the exact path (runs synchronously), index build
    DevBuf q_dev, out_dist_dev, out_idx_dev, big_keys, fb_sample, fb_keys, fb_out, scratch, fb_cnt, fb_sort;
    DevBuf mid_q, mid_planes, mid_small, mid_qal, mid_wave_out, mid_wave_cnt, mid_keys, mid_out, mid_sample;  // the middle tier (synchronous)
    DevBuf mid_cos_center, mid_cos_rows;   // cosine tier: column means c [d_pad], [2][mid_cos_ld] float32 1/|x| and x.c/|x| (built at first use)
    long long mid_cos_n = -1, mid_cos_ld = 0;   // rows mid_cos_rows covers (an append makes it stale)
    // The first filters' candidate lists overflowed for most queries of three calls in a row (descriptors sharing a large
    // offset under cosine, one tight cluster: every row inside the slack): calls go straight to the middle tier -- the
    // overflowing pass, its re-rank of `cap` rows per query and its select bought nothing (2 M x 128 cosine, 32 queries:
    // 5.6 of a 5.8 ms call).  The first filter is tried again after 16 such calls, then 32, 64 ... 1024 while it keeps
    // overflowing (a burst of atypical queries costs sixteen slower calls, data of that geometry a probe in a thousand);
    // the first probe that does not overflow re-arms it.
    int overflow16 = 0;
    bool first_suspended = false;
    unsigned direct_calls = 0, probe_interval = 16;
    PinnedStage stage;
    hipEvent_t ev_ref = nullptr;   // SQ_TRACE (measurement aid): the origin of the printed call timelines
    ~DenseHandle() override {
        if (ev_ref) (void)hipEventDestroy(ev_ref);
        for (DevBuf* b : {&owned, &scan, &scan8, &nrow8, &norms, &norms1, &zeros, &center, &cos_nx, &q_dev, &out_dist_dev, &out_idx_dev, &big_keys,
                          &fb_sample, &fb_keys, &fb_out, &scratch, &fb_cnt, &fb_sort, &mid_q, &mid_planes, &mid_small, &mid_qal,
                          &mid_wave_out, &mid_wave_cnt, &mid_keys, &mid_out, &mid_sample, &mid_cos_center, &mid_cos_rows})
            b->release();
        for (auto& sl : slot) sl.release();
        stage.release();
    }
};

// SQ_HOSTPROF=1 (measurement aid): host nanoseconds of an asynchronous sq_dense_search by section, printed by sq_dense_sync
struct HostProf {
    bool on = getenv("SQ_HOSTPROF") != nullptr;
    long long ns[6] = {0, 0, 0, 0, 0, 0};   // entry (lookup, lock, options, device), resolve of the slot, stream / event setup, enqueue, wait-resolve, calls
    std::chrono::steady_clock::time_point t;
    void start() { if (on) t = std::chrono::steady_clock::now(); }
    void lap(int i) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        ns[i] += std::chrono::duration_cast<std::chrono::nanoseconds>(now - t).count();
        t = now;
    }
};
static HostProf g_hostprof;

// -------------------------------------------------------------- host driver
static constexpr int kSelectLdsKeys64 = 16384;
static constexpr int kSelectLdsKeys128 = 7168;

// select + the finalisation post-op (keys -> distances / ids, certification, status) in one launch
// `expect`: candidates per query the caller expects (0 = unknown).  The kernel keeps up to lds_keys keys of a query in
// LDS and reads longer lists from global memory; sizing the LDS for the expected list instead of the largest possible
// lets two workgroups share a CU (36-40 registers per thread: LDS is what limits them) -- with one workgroup per query
// a 1024-query batch is four rounds of workgroups otherwise.
template <class K, class Post>
static int select_launch_t(const K* keys, const u32* cnt, u32 cap, long long stride, int k, int nq, K* out,
                           const Post& post, hipStream_t st, DevBuf& sort_scratch, long long expect = 0, int cnt_shift = 0) {
    static std::atomic<unsigned long long> attr_done{0};
    const int lds_max = sizeof(K) == 8 ? kSelectLdsKeys64 : kSelectLdsKeys128;
    // beyond the one-workgroup select: full sort (sq_select.hpp, "any-k sorted select"); the scratch belongs to the
    // call slot (asynchronous calls in flight, or two handles on two threads, must not share it)
    if (k > lds_max) return sort_select_large<K, Post>(keys, cnt, cap, stride, k, nq, out, sort_scratch, post, st);
    int lds_keys = lds_max;
    if (expect > 0 && nq > 256) {  // (fewer queries than CUs: one round of workgroups either way)
        const int half = (int)((80 * 1024) / sizeof(K)) - SELECT_SORT_MAX;  // two workgroups in 160 KB
        if (half >= k && 2 * expect <= half) lds_keys = half;
    }
    const size_t lds_full = (size_t)(lds_max + SELECT_SORT_MAX) * sizeof(K);
    const size_t lds = (size_t)(lds_keys + SELECT_SORT_MAX) * sizeof(K);
    SQ_TRY(ensure_dyn_lds(reinterpret_cast<const void*>(&select_topk_kernel<K, Post>), (int)lds_full, attr_done));
    hipLaunchKernelGGL((select_topk_kernel<K, Post>), dim3(nq), dim3(1024), lds, st, keys, cnt, cap, stride, k, lds_keys,
                       out, post, cnt_shift);
    return SQ_OK;
}

template <int WAVES, int NSTAGE, int KU, int QT, int QP, bool AB, bool SAMPLE, bool NT = false>
static int scan_launch_t(const DenseScanArgs& a, size_t lds, hipStream_t st) {
    static std::atomic<unsigned long long> attr_done{0};
    SQ_TRY(ensure_dyn_lds(reinterpret_cast<const void*>(&dense_scan_kernel<WAVES, NSTAGE, KU, QT, QP, AB, SAMPLE, NT>), 160 * 1024, attr_done));
    hipLaunchKernelGGL((dense_scan_kernel<WAVES, NSTAGE, KU, QT, QP, AB, SAMPLE, NT>), dim3((unsigned)(a.nrb * a.nqt)),
                       dim3(WAVES * 64), lds, st, a);
    return SQ_OK;
}

static constexpr int SCAN_LDS_TAIL = 8 * 16;  // per-wave survivor counters

// Launch geometry of the scan for a padded dimension and `qt` query tiles per wave:
// waves per workgroup and ring depth.
struct ScanGeom {
    int waves, stages;
    size_t lds;
};
static ScanGeom scan_geometry(const Options& o, int d_pad, int qt, int qp) {
    const int ku = d_pad / KT;
    const bool ab = qt > 1 || qp == 1;   // query fragments in AGPRs (multi-tile batches)
    const bool qreg = ab || ku <= 1;     // not re-read from an LDS copy
    const int qb = qreg ? 0 : TILE_ROWS * d_pad * 4;
    ScanGeom g{};
    // eight waves (two per SIMD) whenever the query fragments leave room: one tile, or two tiles of one plane
    g.waves = (ku <= 1 && (qt == 1 || (qt == 2 && qp == 1))) ? 8 : 4;
    if (ku <= 1 && o.dense_waves == 4) g.waves = 4;
    int ns = (160 * 1024 - qb - SCAN_LDS_TAIL) / (g.waves * SLOT_BYTES);
    const int ns_max = g.waves == 8 ? 2 : 4;
    if (ns > ns_max) ns = ns_max;
    if (qt == 1 && !ab && o.dense_stages >= 2 && o.dense_stages <= ns) ns = o.dense_stages;
    g.stages = ns;
    g.lds = (size_t)qb + (size_t)g.waves * ns * SLOT_BYTES + SCAN_LDS_TAIL;
    // staging the query tiles through the ring needs the ring to be at least as large
    if (qreg && (size_t)qt * TILE_ROWS * d_pad * 4 > (size_t)g.waves * ns * SLOT_BYTES) g.stages = 0;
    return g;
}

template <int KU, bool SAMPLE>
static int scan_launch_ku(const DenseScanArgs& a, const ScanGeom& g, int qt, int qp, hipStream_t st) {
    if constexpr (KU <= 1) {
        // one group of query tiles over a copy far beyond the MALL: the non-temporal build of the two kernels that serve
        // 33 .. 128 queries
        const bool nt = a.nt && a.nqt == 1;
        if (qt == 4 && qp == 1 && nt) return scan_launch_t<4, 4, KU, 4, 1, true, SAMPLE, true>(a, g.lds, st);
        if (qt == 4) return qp == 1 ? scan_launch_t<4, 4, KU, 4, 1, true, SAMPLE>(a, g.lds, st) : scan_launch_t<4, 4, KU, 4, 2, true, SAMPLE>(a, g.lds, st);
        if (qt == 2 && qp == 1 && g.waves == 8 && nt) return scan_launch_t<8, 2, KU, 2, 1, true, SAMPLE, true>(a, g.lds, st);
        if (qt == 2 && qp == 1 && g.waves == 8) return scan_launch_t<8, 2, KU, 2, 1, true, SAMPLE>(a, g.lds, st);
        if (qt == 2) return qp == 1 ? scan_launch_t<4, 4, KU, 2, 1, true, SAMPLE>(a, g.lds, st) : scan_launch_t<4, 4, KU, 2, 2, true, SAMPLE>(a, g.lds, st);
        if (g.waves == 8) return scan_launch_t<8, 2, KU, 1, 2, false, SAMPLE>(a, g.lds, st);
    } else if (qp == 1) {
        // d_pad > 128, several query tiles in the batch: all k-units' q_hi fragments in AGPRs (KU * QT * 32 <= 256)
        if (qt == 2) return scan_launch_t<4, 4, KU, 2, 1, true, SAMPLE>(a, g.lds, st);
        return fail(SQ_ERR_UNSUPPORTED, "dense scan: no kernel for d_pad=%d qt=%d", KU * KT, qt);
    }
    switch (g.stages) {
        case 4: return scan_launch_t<4, 4, KU, 1, 2, false, SAMPLE>(a, g.lds, st);
        case 3: return scan_launch_t<4, 3, KU, 1, 2, false, SAMPLE>(a, g.lds, st);
        default: return scan_launch_t<4, 2, KU, 1, 2, false, SAMPLE>(a, g.lds, st);
    }
}

template <bool SAMPLE>
static int scan_launch(const Options& o, const DenseScanArgs& a, int d_pad, int qt, int qp, hipStream_t st) {
    if (d_pad > RING_MAX_DPAD) {   // rows beyond the ring kernels: one query tile per wave, fragments straight from global memory
        if (qt != 1 && qt != 2 && qt != 4) return fail(SQ_ERR_UNSUPPORTED, "dense scan: d_pad=%d takes one, two or four query tiles per wave", d_pad);
        const dim3 grid((unsigned)(a.nrb * a.nqt)), block(WIDE_WAVES * 64);
        if (qt == 4) {
            if (qp == 2)
                hipLaunchKernelGGL((dense_wide_scan_kernel<2, 4, SAMPLE>), grid, block, 0, st, a, d_pad / KT);
            else
                hipLaunchKernelGGL((dense_wide_scan_kernel<1, 4, SAMPLE>), grid, block, 0, st, a, d_pad / KT);
        } else if (qt == 2) {
            if (qp == 2)
                hipLaunchKernelGGL((dense_wide_scan_kernel<2, 2, SAMPLE>), grid, block, 0, st, a, d_pad / KT);
            else
                hipLaunchKernelGGL((dense_wide_scan_kernel<1, 2, SAMPLE>), grid, block, 0, st, a, d_pad / KT);
        } else if (qp == 2) {
            hipLaunchKernelGGL((dense_wide_scan_kernel<2, 1, SAMPLE>), grid, block, 0, st, a, d_pad / KT);
        } else {
            hipLaunchKernelGGL((dense_wide_scan_kernel<1, 1, SAMPLE>), grid, block, 0, st, a, d_pad / KT);
        }
        return SQ_OK;
    }
    const ScanGeom g = scan_geometry(o, d_pad, qt, qp);
    if (g.stages < 2 || ((qt > 1 || qp == 1) && g.waves == 4 && g.stages != 4))
        return fail(SQ_ERR_UNSUPPORTED, "dense scan: d_pad=%d qt=%d leaves no room for the LDS ring", d_pad, qt);
    switch (d_pad / KT) {
        case 1: return scan_launch_ku<1, SAMPLE>(a, g, qt, qp, st);
        case 2: return scan_launch_ku<2, SAMPLE>(a, g, qt, qp, st);
        case 3: return scan_launch_ku<3, SAMPLE>(a, g, qt, qp, st);
        case 4: return scan_launch_ku<4, SAMPLE>(a, g, qt, qp, st);
        default: return fail(SQ_ERR_UNSUPPORTED, "dense scan: d_pad=%d", d_pad);
    }
}

// Error coefficients of the filter score (derivation at their use in dense_search_device); the index build
// needs them too: the L2 scan starts from row norms shrunk by the row's share of the bound.
static constexpr double kEpsA2 = 0.0078125 + 6.103515625e-05;   // two query planes: 2^-7 + 2^-14
static constexpr double kEpsA1 = 0.015625 + 1.220703125e-04;    // one query plane:  2^-6 + 2^-13
static double dense_eps_b(int d_pad) { return (3.0 * d_pad + 8.0) * 1.1920928955078125e-07; }

// Query tiles per wave for a batch of `nqt` 32-query tiles (sq_dense_scan.hpp): one tile keeps the
// HBM-bound configuration; larger batches reuse each streamed row tile for 2 or 4 query tiles, as
// many as the register budget allows (four at d_pad = 128, two beyond).
static int scan_query_tiles(const Options& o, int d_pad, int nqt) {
    const int ku = d_pad / KT;
    if (nqt <= 1) return 1;
    if (d_pad > RING_MAX_DPAD) {   // (sq_dense_wide.hpp: two tiles per wave for batches beyond 32 queries, four beyond 64)
        if (o.dense_qt == 1 || o.dense_qt == 2 || o.dense_qt == 4) return o.dense_qt;
        return nqt >= 3 ? 4 : 2;
    }
    int want = nqt >= 3 ? 4 : 2;
    if (o.dense_qt == 1 || o.dense_qt == 2 || o.dense_qt == 4) want = o.dense_qt;
    if (ku >= 2 && want > 2) want = 2;  // four tiles of a 256-wide row spill past 512 registers
    if (ku > 1 && want == 1) return 1;  // forced: the LDS-copy kernel
    return want;
}

// Wait for an event the way stream_wait waits for a stream (poll, then block).
static hipError_t event_wait(hipEvent_t ev) {
    const long long budget_us = g_opt.spin_wait_us;
    if (budget_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            for (int i = 0; i < 64; ++i) {
                const hipError_t e = hipEventQuery(ev);
                if (e != hipErrorNotReady) return e;
            }
            if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > budget_us)
                break;
        }
    }
    return hipEventSynchronize(ev);
}

// the int8 scan for a row width (sq_dense_i8.hpp I8Geom): launch, and the attribute every instantiation needs once
template <int KS, bool SAMPLE>
static int dense8_scan_launch_t(const Dense8ScanArgs& a, hipStream_t st) {
    using G = I8Geom<KS>;
    static std::atomic<unsigned long long> attr_done{0};
    // (the kernel has a few bytes of static LDS of its own: the attribute is the ring, not the CU's 160 KiB)
    SQ_TRY(ensure_dyn_lds(reinterpret_cast<const void*>(&dense8_scan_kernel<KS, SAMPLE>), G::WAVES * G::NSTAGE * G::SLOT_BYTES, attr_done));
    hipLaunchKernelGGL((dense8_scan_kernel<KS, SAMPLE>), dim3((unsigned)a.nrb), dim3(G::WAVES * 64), (size_t)G::WAVES * G::NSTAGE * G::SLOT_BYTES, st, a);
    return SQ_OK;
}
template <bool SAMPLE>
static int dense8_scan_launch(int row_bytes, const Dense8ScanArgs& a, hipStream_t st) {
    switch (row_bytes) {
        case 128: return dense8_scan_launch_t<4, SAMPLE>(a, st);
        case 256: return dense8_scan_launch_t<8, SAMPLE>(a, st);
        case 512: return dense8_scan_launch_t<16, SAMPLE>(a, st);
    }
    return fail(SQ_ERR_INVALID, "int8 scan: unsupported row width %d", row_bytes);
}
template <int QT, bool SAMPLE>
static int dense8_scan_mt_launch_t(const Dense8ScanArgs& a, hipStream_t st) {
    using G = I8Geom<4>;
    static std::atomic<unsigned long long> attr_done{0};
    SQ_TRY(ensure_dyn_lds(reinterpret_cast<const void*>(&dense8_scan_mt_kernel<QT, SAMPLE>), 160 * 1024, attr_done));
    hipLaunchKernelGGL((dense8_scan_mt_kernel<QT, SAMPLE>), dim3((unsigned)(a.nrb * a.nqt)), dim3(G::WAVES * 64), (size_t)G::WAVES * G::NSTAGE * G::SLOT_BYTES, st, a);
    return SQ_OK;
}
// one query tile per wave: dense8_scan_kernel for the row width; 2 or 4 tiles (128-byte rows): dense8_scan_mt_kernel
template <bool SAMPLE>
static int dense8_scan_any(int row_bytes, int qt, const Dense8ScanArgs& a, hipStream_t st) {
    if (qt == 1) return dense8_scan_launch<SAMPLE>(row_bytes, a, st);
    if (row_bytes == 128 && qt == 2) return dense8_scan_mt_launch_t<2, SAMPLE>(a, st);
    if (row_bytes == 128 && qt == 4) return dense8_scan_mt_launch_t<4, SAMPLE>(a, st);
    return fail(SQ_ERR_INVALID, "int8 scan: %d query tiles per wave over %d-byte rows", qt, row_bytes);
}
// the three-launch form of a one-tile int8 call (sq_dense_i8.hpp): head and body per row width / metric
template <int KS>
static int dense8_head_launch_t(const Dense8HeadArgs& a, hipStream_t st) {
    using G = I8Geom<KS>;
    static std::atomic<unsigned long long> attr_done{0};
    // (the kernel has static LDS of its own: the attribute is the ring, not the CU's 160 KiB)
    SQ_TRY(ensure_dyn_lds(reinterpret_cast<const void*>(&dense8_head_kernel<KS>), G::WAVES * G::NSTAGE * G::SLOT_BYTES, attr_done));
    hipLaunchKernelGGL((dense8_head_kernel<KS>), dim3((unsigned)a.s.nrb), dim3(G::WAVES * 64), (size_t)G::WAVES * G::NSTAGE * G::SLOT_BYTES, st, a);
    return SQ_OK;
}
static int dense8_head_launch(int row_bytes, const Dense8HeadArgs& a, hipStream_t st) {
    switch (row_bytes) {
        case 128: return dense8_head_launch_t<4>(a, st);
        case 256: return dense8_head_launch_t<8>(a, st);
        case 512: return dense8_head_launch_t<16>(a, st);
    }
    return fail(SQ_ERR_INVALID, "int8 head: unsupported row width %d", row_bytes);
}
template <int KS, bool COSINE>
static int dense8_body_launch_t(const Dense8ScanArgs& a, const Dense8TailArgs& t, hipStream_t st) {
    using G = I8Geom<KS>;
    static std::atomic<unsigned long long> attr_done{0};
    SQ_TRY(ensure_dyn_lds(reinterpret_cast<const void*>(&dense8_body_kernel<KS, COSINE>), G::WAVES * G::NSTAGE * G::SLOT_BYTES, attr_done));
    hipLaunchKernelGGL((dense8_body_kernel<KS, COSINE>), dim3((unsigned)a.nrb), dim3(G::WAVES * 64), (size_t)G::WAVES * G::NSTAGE * G::SLOT_BYTES, st, a, t);
    return SQ_OK;
}
static int dense8_body_launch(int row_bytes, bool cosine, const Dense8ScanArgs& a, const Dense8TailArgs& t, hipStream_t st) {
    switch (row_bytes) {
        case 128: return cosine ? dense8_body_launch_t<4, true>(a, t, st) : dense8_body_launch_t<4, false>(a, t, st);
        case 256: return cosine ? dense8_body_launch_t<8, true>(a, t, st) : dense8_body_launch_t<8, false>(a, t, st);
        case 512: return cosine ? dense8_body_launch_t<16, true>(a, t, st) : dense8_body_launch_t<16, false>(a, t, st);
    }
    return fail(SQ_ERR_INVALID, "int8 body: unsupported row width %d", row_bytes);
}
static int i8_waves(int row_bytes) { return row_bytes == 512 ? I8Geom<16>::WAVES : I8Geom<4>::WAVES; }

// Shapes the middle tier covers (sq_dense_mid.hpp)
static bool dense_mid_shape_ok(const DenseHandle* h) {
    const bool cosine = h->metric == SQ_METRIC_COSINE;
    return h->d <= 512 && (reinterpret_cast<uintptr_t>(h->db) & 15u) == 0 && (h->ld & 3) == 0 && h->ld >= (h->d + 3) / 4 * 4 && h->n >= 32 &&
           (cosine ? h->cos_nx.p != nullptr : h->norms.p != nullptr);
}

// Enqueue one search (nq <= kDenseQueryChunk queries) on `st` with the workspace of slot `s`; nothing is
// waited for.  dense_resolve() finishes the call: it waits for the kernels, reads the status words and
// sends uncertified queries down the exact path.
static int dense_enqueue(DenseHandle* h, DenseSlot& s, const float* q, int nq, int k, void* out_dist, long long* out_idx,
                         hipStream_t st, bool use_event) {
    const long long n = h->n;
    const int d = h->d, d_pad = h->d_pad;
    const int kk = (int)(k < n ? k : n);
    const bool cosine = h->metric == SQ_METRIC_COSINE;
    // (profile = N > 1: every N-th asynchronous call only -- four event records per call weigh on a small shard's step)
    const bool prof = h->opt.profile == 1 || (h->opt.profile > 1 && (!use_event || h->async_calls % (unsigned)h->opt.profile == 0));
    const size_t key_bytes = cosine ? sizeof(K128) : sizeof(u64);
    u32 cap = h->opt.candidate_cap > 0 ? (u32)h->opt.candidate_cap : 65536u;
    if (cap < (u32)(4 * kk)) cap = (u32)(4 * kk);
    const bool small = n <= (long long)cap;
    // k beyond the one-workgroup select (16384; cosine 7168): every query takes the exact path, whose select sorts
    // (the reference has no limit on n: lsh.py:513-518)
    const bool scan_ok = h->scan.p != nullptr && !small && kk <= (cosine ? kSelectLdsKeys128 : kSelectLdsKeys64);
    const int qt = scan_query_tiles(h->opt, d_pad, (nq + TILE_ROWS - 1) / TILE_ROWS);  // query tiles per wave
    // query planes: the multi-tile configuration is MFMA bound, so it drops q_lo (half the MFMAs, twice the
    // product bound: ~1.4x more rows pass the filter) unless asked otherwise
    // (rows beyond the ring kernels, sq_dense_wide.hpp: two planes unless option dense_qplanes = 1 -- half the query bytes read from L2 per row tile, twice the slack)
    const int qp = d_pad > RING_MAX_DPAD ? (h->opt.dense_qplanes == 1 ? 1 : 2)
                                        : ((qt > 1 && (h->opt.dense_qplanes != 2 || d_pad > KT)) ? 1 : 2);
    const int group_q = qt * TILE_ROWS;                                           // queries per scan workgroup
    const int nqt = (nq + group_q - 1) / group_q;                                 // groups of qt query tiles
    const int nq_pad = nqt * group_q;
    DenseCall& c = s.call;
    c = DenseCall{};
    c.q = q;
    c.nq = nq;
    c.k = k;
    c.nq_pad = nq_pad;
    c.out_dist = out_dist;
    c.out_idx = out_idx;
    c.st = st;
    c.prof = prof;
    c.small = small;
    c.cap = cap;
    c.use_event = use_event;
    if (prof) {
        for (auto& e : s.ev)
            if (!e) SQ_HIP(hipEventCreate(&e));
        if (!h->ev_ref && getenv("SQ_TRACE")) {
            SQ_HIP(hipEventCreate(&h->ev_ref));
            SQ_HIP(hipEventRecord(h->ev_ref, st));
        }
        SQ_HIP(hipEventRecord(s.ev[0], st));
    }
    SQ_TRY(s.cnt.reserve((size_t)nq_pad * 4 * 32));   // (the fused int8 call keeps its counters a cache line apart: I8_CNT_SHIFT)
    SQ_TRY(s.thr.reserve((size_t)nq_pad * 4));
    SQ_TRY(s.qn2.reserve((size_t)nq_pad * 8));
    SQ_TRY(s.q_scaled.reserve((size_t)nq_pad * d_pad * 4));
    SQ_TRY(s.out_keys.reserve((size_t)nq * k * key_bytes));
    SQ_TRY(s.status_host.reserve((size_t)(nq_pad + nq) * 4));
    if (!s.oflag.p) {   // [0] overflow flag of a call, [1] ticket counter of the fused head (self-cleaning: zero between calls)
        SQ_TRY(s.oflag.reserve(64));
        SQ_HIP(hipMemsetAsync(s.oflag.p, 0, 64, st));
    }
    u32* cnt = s.cnt.as<u32>();
    float* thr = s.thr.as<float>();
    double* qn2 = s.qn2.as<double>();
    uint4* qs = s.q_scaled.as<uint4>();
    // per-query candidate counts and status words land in pinned host memory straight from the
    // finalisation (no copy launch): [cnt (nq_pad) | status (nq)]
    u32* hs_raw = reinterpret_cast<u32*>(s.status_host.p);
    u32* hs_raw_dev = nullptr;
    SQ_TRY(s.status_host.device_ptr(reinterpret_cast<void**>(&hs_raw_dev)));
    u32* hs_dev = hs_raw_dev + nq_pad;
    // error bound of the bf16 filter score (sq_dense_exact.hpp filter_eps):
    //   products: |x q' - x_hi (q'_hi + q'_lo)| <= (2^-8 + 2^-15) |x||q'|, q' = -2q  ->  eps_a = 2^-7 + 2^-14 (times X|q|)
    //             (cosine: unit vectors and q' = -q^ without the factor 2: half of that)
    //   float32 accumulation of 2d+1 terms and the float32 norm                     ->  eps_b = (3d+8) 2^-23
    //   one query plane: |x q' - x_hi q'_hi| <= (2^-8 + 2^-8 + 2^-16) |x||q'|                 ->  eps_a = 2^-6 + 2^-13
    const double eps_a = (qp == 2 ? kEpsA2 : kEpsA1) * (cosine ? 0.5 : 1.0);
    const double eps_b = dense_eps_b(d_pad);
    const size_t l2_lds = (size_t)((d + 3) / 4 * 4) * 4;
    if (cosine) {
        SQ_TRY(s.cos_nq.reserve((size_t)nq * 8));
        hipLaunchKernelGGL(dense_cos_qnorm_kernel, dim3((nq + 63) / 64), dim3(64), 0, st, q, nq, d, s.cos_nq.as<double>());
    }
    const double* cnx = h->cos_nx.as<double>();
    const double* cnq = s.cos_nq.as<double>();

    const long long key_stride = small ? n : (long long)cap;
    if (small) {
        // every row is a candidate: exact keys for all rows, no scan
        SQ_TRY(s.keys.reserve((size_t)nq * key_stride * key_bytes));
        hipLaunchKernelGGL(fill_u32_kernel, dim3((nq_pad + 255) / 256), dim3(256), 0, st, cnt, (long long)nq_pad, (u32)n);
        const unsigned gx = (unsigned)((n + 255) / 256);
        if (prof) SQ_HIP(hipEventRecord(s.ev[1], st));
        if (cosine)
            hipLaunchKernelGGL(dense_exact_cos_kernel, dim3(gx, nq), dim3(256), 0, st, h->db, h->ld, d, q, nullptr, cnt,
                               (u32)n, n, 0ll, s.keys.as<K128>(), key_stride, cnx, cnq, nullptr, 0);
        else
            hipLaunchKernelGGL(dense_exact_l2_kernel, dim3(gx, nq), dim3(256), l2_lds, st, h->db, h->ld, d, q, nullptr,
                               cnt, (u32)n, n, 0ll, s.keys.as<u64>(), key_stride, nullptr, 0);
        if (prof) SQ_HIP(hipEventRecord(s.ev[2], st));
        c.stats.scan_launches = 1;
        c.stats.bytes_scanned = n * (long long)d * 4;
        if (cosine) {
            SQ_TRY(select_launch_t<K128>(s.keys.as<K128>(), cnt, (u32)n, key_stride, k, nq, s.out_keys.as<K128>(),
                                         DenseFinalizeCos{cnt, (u32)n, kk, h->id_base, thr, 0.0, 0, (double*)out_dist, out_idx,
                                                          hs_dev, hs_raw_dev, nullptr, 0},
                                         st, s.sort_tmp));
        } else {
            SQ_TRY(select_launch_t<u64>(s.keys.as<u64>(), cnt, (u32)n, key_stride, k, nq, s.out_keys.as<u64>(),
                                        DenseFinalizeL2{cnt, (u32)n, kk, h->id_base, thr, qn2, 0.0, 0,
                                                        (float*)out_dist, out_idx, hs_dev, hs_raw_dev, nullptr, 0},
                                        st, s.sort_tmp));
        }
    } else if (scan_ok && h->first_suspended && h->opt.dense_mid_tier != 0 && !h->opt.force_fallback && dense_mid_shape_ok(h) &&
               ++h->direct_calls % h->probe_interval != 0) {
        c.all_fallback = true;   // (nothing enqueued: dense_resolve starts every query at the middle tier)
        c.mid_direct = true;
    } else if (scan_ok && h->use8 && h->opt.dense_int8 != 0 && !(h->suspended8 && h->opt.dense_int8 < 0) && kk <= (cosine ? kSelectLdsKeys128 : kSelectLdsKeys64) &&
               (nq <= TILE_ROWS || (h->row8 == 128 && (qt == 2 || qt == 4) && nq <= h->opt.dense_int8_batch))) {
        // ---- the int8 first-stage filter (sq_dense_i8.hpp): half the bytes per row, measured error bound
        c.int8 = true;
        if (h->suspended8) {   // dense_int8 = 1 on the handle: the filter is armed again (and judged again from the next calls)
            h->suspended8 = false;
            h->overflow8 = 0;
        }
        // (128-byte rows stream in ring units of 64 rows on eight waves.  Two other geometries were built and measured --
        // commit 5048766: 128-row units on four waves 0.197 against 0.204 ms per pass alone but 0.244-0.251 against 0.223-0.226 ms
        // per pipelined step; 32-row units on sixteen waves 0.236 -- and removed again.)
        const int row8 = h->row8, unit_rows = i8_unit_rows(row8), spu = 2 * (unit_rows / 32), waves8 = i8_waves(row8);   // samples per unit
        const long long n_units = (n + unit_rows - 1) / unit_rows;
        long long stride = h->opt.sample_stride;
        if (stride <= 0) {
            // units of 64 rows, four samples per unit: the bf16 path's cost model in units of two tiles
            // (measured at 10 M x 128, k = 100: 0.293 / 0.269 / 0.262 / 0.258 / 0.257 / 0.257 / 0.263 ms per step at 6 / 8 / 10 / 12 / 16 / 20 / 24)
            // 256- and 512-byte rows: flat from 8 to 14, best at 10 (0.471 / 0.988 ms per step at 10 M x 256 / 512); the float64 cosine
            // re-rank costs ~3.6x the float32 one per candidate: sqrt of that off the stride, as in the bf16 path
            stride = (long long)((row8 == 128 ? 14.0 : 10.0) * sqrt((double)n / 1e7 * 100.0 / (double)kk / (double)qt) / (cosine ? 1.9 : 1.0) + 0.5);
            if (stride > 16) stride = 16;
            // The fused call with the tightened threshold (sq_dense_i8.hpp): the sample only has to keep the first-level
            // entries inside the wave segments -- what is re-ranked no longer depends on it -- so it is several times sparser
            const bool will_tighten = h->opt.dense_fused != 0 && h->opt.dense_tighten != 0 && qt == 1 && nqt == 1;
            if (will_tighten) {
                stride = (long long)(40.0 * sqrt((double)n / 1e7 * 100.0 / (double)kk) + 0.5);
                if (stride > 64) stride = 64;
            }
            if (stride < 1) stride = 1;
            if (stride > (long long)cap / (16ll * kk)) stride = std::max<long long>(1, (long long)cap / (16ll * kk));
        }
        while (stride > 1 && (n_units / stride) * spu < 8ll * kk) stride >>= 1;
        const long long ns_units = (n_units + stride - 1) / stride;
        const long long ns = ns_units * spu;
        SQ_TRY(s.keys.reserve((size_t)nq * key_stride * key_bytes));
        SQ_TRY(s.q8.reserve((size_t)2 * nq_pad * row8));
        SQ_TRY(s.par8.reserve((size_t)nq_pad * 8));
        const int cus = cu_count(h->device);
        int nrb = h->opt.dense_blocks > 0 ? h->opt.dense_blocks : (use_event && h->opt.dense_async_streams == 2 && nqt == 1 ? cus * 3 / 4 : cus);
        nrb = (nrb + 7) / 8 * 8;
        const long long n_waves = (long long)nrb * nqt * waves8;
        const u32 wave_cap = 2048;
        const int ldq = (d + 3) / 4 * 4;
        SQ_TRY(s.wave_out.reserve((size_t)n_waves * wave_cap * 8));
        SQ_TRY(s.wave_cnt.reserve((size_t)n_waves * 8));
        SQ_TRY(s.q_al.reserve((size_t)nq * ldq * 4));
        u32* oflag = s.oflag.as<u32>();
        const float* centerp = (!cosine && h->center.p) ? h->center.as<float>() : nullptr;
        Dense8ScanArgs a{};
        a.scan8 = h->scan8.as<signed char>();
        a.nrow = h->nrow8.as<float>();
        a.n = n;
        a.n_units = n_units;
        a.qs8 = s.q8.as<signed char>();
        a.par = s.par8.as<float2>();
        a.thr = thr;
        a.wave_out = s.wave_out.as<uint2>();
        a.wave_cnt = s.wave_cnt.as<u32>();
        a.wave_cap = wave_cap;
        a.sample_out = s.sample.as<float>();
        a.ns = ns;
        a.nqt = nqt;
        a.plane_rows = nq_pad;
        a.debug = h->opt.dense_debug;
        {
            // (cacheable head: 32-64 MB measured best here -- 0.244 ms per step at 10 M rows against 0.250 at the bf16 copy's 192 MB)
            const size_t copy_bytes = (size_t)h->n_pad64 * row8;
            const long long keep_mb = h->opt.dense_nt_keep_mb > 0 ? h->opt.dense_nt_keep_mb : 64;
            a.nt = h->opt.dense_nt >= 0 ? h->opt.dense_nt : (copy_bytes > ((size_t)512 << 20) && nqt == 1 ? 1 : 0);   // (several groups re-read the rows from L2)
            a.nt_from_row = h->opt.dense_nt == 0 ? 0x7fffffffffffffffll : h->opt.dense_nt == 1 ? 0ll : (keep_mb << 20) / row8;
        }
        int nrb_sample = nrb;
        if (use_event && h->opt.dense_async_streams == 2 && h->opt.dense_blocks <= 0 && nqt == 1) {
            int sb = h->opt.dense_sample_blocks > 0 ? h->opt.dense_sample_blocks : (h->opt.dense_sample_blocks < 0 ? nrb : cus - nrb);
            sb = (sb + 7) / 8 * 8;
            if (sb >= 8 && sb < nrb_sample) nrb_sample = sb;
        }
        if (ns_units < (long long)nrb_sample * waves8) nrb_sample = (int)(((ns_units + waves8 - 1) / waves8 + 7) / 8 * 8);
        // Three launches instead of six (sq_dense_i8.hpp, "a call in three launches"): one query tile, and a head grid whose
        // M = workgroups x waves x 2 lane minima per query number at least 4 k (the threshold is then within a few per cent of
        // the k-th smallest sample) and at most 2048 (what the last workgroup's waves hold in registers).
        bool fused = h->opt.dense_fused != 0 && qt == 1 && nqt == 1 && ns_units >= 2ll * kk;
        const bool tighten = fused && h->opt.dense_tighten != 0;
        if (fused) {
            const int lanes_per_wg = waves8 * 2;
            const int wg_min = ((4 * kk + lanes_per_wg - 1) / lanes_per_wg + 7) / 8 * 8, wg_max = 2048 / lanes_per_wg;
            if (wg_min > wg_max || wg_min > cus)
                fused = false;
            else
                nrb_sample = std::min(std::max(nrb_sample, wg_min), std::min(wg_max, (cus + 7) / 8 * 8));
        }
        // minima a head workgroup hands over per query: the k best of a call fall on a workgroup k / workgroups at a time
        int keep8 = waves8 * 2;
        if (fused) {
            const double per_wg = (double)kk / nrb_sample;
            if (per_wg <= 1.7 && keep8 > 4) keep8 = 4;
            else if (per_wg <= 4.5 && keep8 > 8) keep8 = 8;
        }
        const long long lane_m = (long long)nrb_sample * keep8;
        SQ_TRY(s.sample.reserve(fused ? (size_t)TILE_ROWS * lane_m * 4 : (size_t)nq_pad * ns * 4));
        a.sample_out = s.sample.as<float>();
        if (fused) {
            SQ_TRY(s.wave_score.reserve((size_t)n_waves * wave_cap * 4));
            SQ_TRY(s.hist8.reserve((size_t)I8_HIST_WORDS * 4));
            a.wave_score = s.wave_score.as<float>();
            a.hist = s.hist8.as<u32>();
        }
        int wpb = 2;   // survivor segments per re-rank workgroup (128 threads each)
        if (!cosine && h->opt.dense_rerank_segments > 0 && waves8 % h->opt.dense_rerank_segments == 0 && h->opt.dense_rerank_segments <= 4)
            wpb = h->opt.dense_rerank_segments;
        const size_t rr_lds = (qt == 1 && ldq <= 156) ? (size_t)32 * (ldq + 4) * 4 : 0;
        // The chain of six launches, eagerly or as a captured graph.  A launch costs ~2.8 us of host time
        // (tools/micro/launch_cost.hip: 19.3 us for a chain of seven, 5.6-6.3 us for one hipGraphLaunch of the same chain; in
        // this call 20 against 9.5 us, tools/host_sections.py).  It does not shorten a 1.25 M-row shard's ~55 us step -- that
        // is the latency of the chain itself, three deep -- but frees the host thread for the gather / merge work of a
        // sharded search.  The slot's first call of a shape runs eagerly (it sizes the workspace and sets the kernels'
        // attributes), the second captures, later ones launch the graph; the caller's pointers travel through a pinned
        // block (DenseCallPtrs).
        const DenseCallPtrs* ind = nullptr;
        auto chain = [&](hipStream_t cs) -> int {
            if (fused) {
                // head: query prep + sample pass (lane minima) + thresholds by the last workgroup
                Dense8HeadArgs ha{};
                ha.s = a;
                ha.s.unit_step = stride;
                ha.s.n_sel = ns_units;
                ha.s.nrb = nrb_sample;
                ha.q = q;
                ha.nq = nq;
                ha.d = d;
                ha.center = centerp;
                ha.dx = h->dx8;
                ha.r_max = h->rmax8;
                ha.x_max = h->xmax8;
                ha.cosine = cosine ? 1 : 0;
                ha.qn2 = qn2;
                ha.cnt = cnt;
                ha.oflag = oflag;
                ha.q_al = s.q_al.as<float>();
                ha.ldq = ldq;
                ha.ind = ind;
                ha.lane_min = s.sample.as<float>();
                ha.keep = keep8;
                ha.kk = kk;
                if (const int rc = dense8_head_launch(row8, ha, cs)) return rc;
                // body: the full pass; its workgroups re-rank their own survivors when the stream has drained
                Dense8ScanArgs b = a;
                b.unit_step = 1;
                b.n_sel = n_units;
                b.nrb = nrb;
                Dense8TailArgs ta{h->db, h->ld, d, s.q_al.as<float>(), ldq, nq, s.keys.p, cnt, cap, oflag, cnx, cnq, h->opt.dense_debug, qn2, kk, tighten ? 1 : 0, nullptr};
                s.clk8_wgs = 0;
                if ((h->opt.dense_debug & 8192) && !use_event) {   // (blocking calls only: printed by dense_resolve)
                    if (const int rc = s.clk8.reserve((size_t)nrb * 64)) return rc;
                    SQ_HIP(hipMemsetAsync(s.clk8.p, 0, (size_t)nrb * 64, cs));
                    ta.clk = s.clk8.as<long long>();
                    s.clk8_wgs = nrb;
                }
                if (prof) SQ_HIP(hipEventRecord(s.ev[1], cs));
                if (const int rc = dense8_body_launch(row8, cosine, b, ta, cs)) return rc;
                if (prof) {
                    SQ_HIP(hipEventRecord(s.ev[2], cs));
                    SQ_HIP(hipEventRecord(s.ev[4], cs));
                }
                if (cosine) {
                    DenseFinalizeCos fin{cnt, cap, kk, h->id_base, thr, 0.0, 1, (double*)out_dist, out_idx, hs_dev, hs_raw_dev, oflag, 0};
                    fin.lin = s.par8.as<float2>();
                    fin.ind = ind;
                    if (tighten) fin.thr2k = a.hist + TILE_ROWS * I8_HIST_BINS;
                    fin.cnt_shift = I8_CNT_SHIFT;
                    return select_launch_t<K128>(s.keys.as<K128>(), cnt, cap, key_stride, k, nq, s.out_keys.as<K128>(), fin, cs, s.sort_tmp, 8 * stride * kk, I8_CNT_SHIFT);
                }
                DenseFinalizeL2 fin{cnt, cap, kk, h->id_base, thr, qn2, 0.0, 1, (float*)out_dist, out_idx, hs_dev, hs_raw_dev, oflag, 0};
                fin.lin = s.par8.as<float2>();
                fin.ind = ind;
                if (tighten) fin.thr2k = a.hist + TILE_ROWS * I8_HIST_BINS;
                fin.cnt_shift = I8_CNT_SHIFT;
                return select_launch_t<u64>(s.keys.as<u64>(), cnt, cap, key_stride, k, nq, s.out_keys.as<u64>(), fin, cs, s.sort_tmp, 8 * stride * kk, I8_CNT_SHIFT);
            }
            {
                auto prep = row8 == 128 ? dense8_prep_queries_kernel<2> : (row8 == 256 ? dense8_prep_queries_kernel<4> : dense8_prep_queries_kernel<8>);
                hipLaunchKernelGGL(prep, dim3(nq_pad / 4), dim3(64), 0, cs, q, nq, d, centerp, h->dx8, h->rmax8, h->xmax8,
                                   s.q8.as<signed char>(), s.par8.as<float2>(), qn2, thr, cnt, oflag, s.q_al.as<float>(), ldq, ind, cosine ? 1 : 0, nq_pad);
            }
            Dense8ScanArgs b = a;
            b.unit_step = stride;   // sample pass (on the CUs the pipelined full pass leaves free)
            b.n_sel = ns_units;
            b.nrb = nrb_sample;
            if (const int rc = dense8_scan_any<true>(row8, qt, b, cs)) return rc;
            hipLaunchKernelGGL((kth_threshold_f32_kernel<Dense8ThrPost>), dim3(nq), dim3(1024), 0, cs, a.sample_out, ns, kk, thr,
                               Dense8ThrPost{s.par8.as<float2>(), qn2});
            b.unit_step = 1;        // full pass
            b.n_sel = n_units;
            b.nrb = nrb;
            if (prof) SQ_HIP(hipEventRecord(s.ev[1], cs));
            if (const int rc = dense8_scan_any<false>(row8, qt, b, cs)) return rc;
            if (prof) SQ_HIP(hipEventRecord(s.ev[2], cs));
            if (cosine) {
                hipLaunchKernelGGL(dense_rerank_cos_kernel, dim3((unsigned)((n_waves + wpb - 1) / wpb)), dim3(128 * wpb), rr_lds, cs, h->db, h->ld, d,
                                   s.q_al.as<float>(), ldq, a.wave_out, a.wave_cnt, wave_cap, n_waves, wpb, nq, group_q, s.keys.as<K128>(), cnt,
                                   cap, oflag, cnx, cnq, h->opt.dense_debug);
                if (prof) SQ_HIP(hipEventRecord(s.ev[4], cs));
                DenseFinalizeCos fin{cnt, cap, kk, h->id_base, thr, 0.0, 1, (double*)out_dist, out_idx, hs_dev, hs_raw_dev, oflag, 0};
                fin.lin = s.par8.as<float2>();
                fin.ind = ind;
                return select_launch_t<K128>(s.keys.as<K128>(), cnt, cap, key_stride, k, nq, s.out_keys.as<K128>(), fin, cs, s.sort_tmp, 8 * stride * kk);
            }
            hipLaunchKernelGGL(dense_rerank_l2_kernel, dim3((unsigned)((n_waves + wpb - 1) / wpb)), dim3(128 * wpb), rr_lds, cs, h->db, h->ld, d,
                               s.q_al.as<float>(), ldq, a.wave_out, a.wave_cnt, wave_cap, n_waves, wpb, nq, group_q, s.keys.as<u64>(), cnt,
                               cap, oflag, h->opt.dense_debug);
            if (prof) SQ_HIP(hipEventRecord(s.ev[4], cs));
            DenseFinalizeL2 fin{cnt, cap, kk, h->id_base, thr, qn2, 0.0, 1, (float*)out_dist, out_idx, hs_dev, hs_raw_dev, oflag, 0};
            fin.lin = s.par8.as<float2>();
            fin.ind = ind;
            return select_launch_t<u64>(s.keys.as<u64>(), cnt, cap, key_stride, k, nq, s.out_keys.as<u64>(), fin, cs, s.sort_tmp, 8 * stride * kk);
        };
        c.stats.scan_launches = 2;
        c.stats.bytes_scanned = h->n_pad64 * ((long long)row8 + 4) * nqt;
        bool launched = false;
        if (use_event && !prof && h->opt.dense_graph != 0 && !h->graph_broken && st != nullptr && st == s.own) {   // (never a capture on the caller's stream)
            // everything the captured launches were given by value: shapes, the slot's and the handle's buffers
            u64 key = 0xcbf29ce484222325ull;
            auto mix = [&key](u64 v) { key = (key ^ v) * 0x100000001b3ull; };
            for (u64 v : {(u64)(fused ? 1 : 0), (u64)(tighten ? 1 : 0), (u64)(uintptr_t)s.wave_score.p, (u64)(uintptr_t)s.hist8.p, (u64)lane_m, (u64)nq, (u64)k, (u64)row8, (u64)qt, (u64)nqt, (u64)wpb, (u64)stride, (u64)nrb, (u64)nrb_sample, (u64)ns, (u64)a.nt, (u64)a.nt_from_row, (u64)n, (u64)cap,
                          (u64)h->opt.dense_debug, (u64)h->id_base, (u64)(uintptr_t)st, (u64)(uintptr_t)h->db, (u64)(uintptr_t)centerp,
                          (u64)(uintptr_t)h->scan8.p, (u64)(uintptr_t)h->nrow8.p, (u64)(uintptr_t)s.q8.p, (u64)(uintptr_t)s.par8.p,
                          (u64)(uintptr_t)s.sample.p, (u64)(uintptr_t)s.keys.p, (u64)(uintptr_t)s.wave_out.p, (u64)(uintptr_t)s.wave_cnt.p,
                          (u64)(uintptr_t)s.q_al.p, (u64)(uintptr_t)cnt, (u64)(uintptr_t)thr, (u64)(uintptr_t)qn2, (u64)(uintptr_t)oflag,
                          (u64)(uintptr_t)s.out_keys.p, (u64)(uintptr_t)hs_raw_dev, (u64)(uintptr_t)cnx, (u64)(uintptr_t)cnq, (u64)(cosine ? 1 : 0)})
                mix(v);
            if (key == 0) key = 1;
            if (s.gexec && s.gkey != key) {
                (void)hipGraphExecDestroy(s.gexec);
                s.gexec = nullptr;
            }
            if (!s.gexec && s.seen_key == key) {   // second call of this shape on this slot: capture
                SQ_TRY(s.call_ptrs.reserve(sizeof(DenseCallPtrs)));
                void* ind_dev = nullptr;
                SQ_TRY(s.call_ptrs.device_ptr(&ind_dev));
                ind = static_cast<const DenseCallPtrs*>(ind_dev);
                // (a capture that fails is no reason to fail the search: the handle goes back to eager launches for good)
                hipGraph_t g = nullptr;
                bool ok = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess;
                if (ok) {
                    const int rc = chain(st);
                    const hipError_t ec = hipStreamEndCapture(st, &g);
                    ok = rc == SQ_OK && ec == hipSuccess && g != nullptr;
                }
                if (ok) ok = hipGraphInstantiate(&s.gexec, g, nullptr, nullptr, 0) == hipSuccess;
                if (g) (void)hipGraphDestroy(g);
                if (!ok) {
                    (void)hipGetLastError();
                    s.gexec = nullptr;
                    h->graph_broken = true;
                } else {
                    s.gkey = key;
                    if (getenv("SQ_INT8_REPORT")) fprintf(stderr, "[smqtk_hip] call graph captured (%d queries, k = %d)\n", nq, k);
                }
            }
            s.seen_key = key;
            if (s.gexec) {
                *static_cast<DenseCallPtrs*>(s.call_ptrs.p) = DenseCallPtrs{q, out_dist, out_idx};
                SQ_HIP(hipGraphLaunch(s.gexec, st));
                launched = true;
            }
        }
        if (!launched) {
            ind = nullptr;
            SQ_TRY(chain(st));
        }
    } else if (scan_ok) {
        const long long n_tiles = (n + TILE_ROWS - 1) / TILE_ROWS;
        // Sample every stride-th tile.  The sample pass costs ~ n * groups / stride, the re-rank + select
        // ~ nq * stride * k (candidates per query ~ stride * k * slack); with groups ~ nq / (32 qt) the
        // balance is stride ~ sqrt(n / (qt k)), independent of the batch: 20 at 10 M rows, k = 100, one
        // query tile per wave (measured optimum 16-24; 8-12 with four tiles; 4 on a 1.25 M-row shard).
        long long stride = h->opt.sample_stride;
        if (stride <= 0) {
            // (the float64 cosine re-rank costs ~3.6x the float32 L2 one per candidate: sqrt of that off the stride)
            stride = (long long)(20.0 * sqrt((double)n / 1e7 * 100.0 / (double)kk / (double)qt) / (cosine ? 1.9 : 1.0) + 0.5);
            if (stride > 24) stride = 24;
            if (stride < 2) stride = 2;
            if (stride > (long long)cap / (8ll * kk)) stride = (long long)cap / (8ll * kk);  // room in the key lists
        }
        if (stride > 64) stride = 64;
        if (stride < 1) stride = 1;
        while (stride > 1 && (n_tiles / stride) * 2 < 8ll * kk) stride >>= 1;
        const long long ns_tiles = (n_tiles + stride - 1) / stride;
        // one sample (a 16-row group minimum) per lane half per tile; the four-tile sample pass writes runs of four
        // tiles ([query][half][tiles rounded up to 4], +inf in the padding: dense_scan_kernel CHUNK)
        const bool sample_runs = qt == 4 && qp == 1 && d_pad == KT;
        const long long ns = (sample_runs ? (ns_tiles + 3) / 4 * 4 : ns_tiles) * 2;
        SQ_TRY(s.sample.reserve((size_t)nq_pad * ns * 4));
        SQ_TRY(s.keys.reserve((size_t)nq * key_stride * key_bytes));
        const int cus = cu_count(h->device);
        // One scan workgroup per CU fills the chip -- and leaves nothing for the other call slot's short kernels
        // (re-rank, select of the previous call; query prep, sample pass, threshold of the next), whose workgroups
        // need LDS of their own.  When the two slots run on their own streams the scan takes three quarters of
        // the CUs: alone it is as fast (HBM bound: 0.433 ms on 160 workgroups against 0.439 on 256 at 10 M x 128),
        // and the pipelined step drops from 0.50 to 0.46 ms (tools/step_sweep.py dense_blocks=...).
        // (one query tile per wave only: the multi-tile configurations are MFMA bound and want every CU)
        int nrb = h->opt.dense_blocks > 0 ? h->opt.dense_blocks : (use_event && h->opt.dense_async_streams == 2 && nqt == 1 ? cus * 3 / 4 : cus);
        // (the wide kernel, 122 registers and 32 KB of LDS: two eight-wave workgroups per CU -- twice the row bytes in flight)
        if (d_pad > RING_MAX_DPAD && h->opt.dense_blocks <= 0) nrb = qt == 1 ? 2 * cus : cus;   // (two tiles per wave: one workgroup per CU)
        nrb = (nrb + 7) / 8 * 8;
        const int wv = d_pad > RING_MAX_DPAD ? WIDE_WAVES : scan_geometry(h->opt, d_pad, qt, qp).waves;
        // survivors leave the scan as per-wave segments; the re-rank kernel turns them into per-query key lists
        const long long n_waves = (long long)nrb * nqt * wv;
        const u32 wave_cap = 2048;
        const int ldq = (d + 3) / 4 * 4;
        SQ_TRY(s.wave_out.reserve((size_t)n_waves * wave_cap * 8));
        SQ_TRY(s.wave_cnt.reserve((size_t)n_waves * 8));
        SQ_TRY(s.q_al.reserve((size_t)nq * ldq * 4));
        u32* oflag = s.oflag.as<u32>();
        // rows beyond the ring kernels: a second-level threshold between the pass and the re-rank (sq_dense_tighten.hpp).
        // (The ring kernels do not store their entries' scores: the few instructions that would in dense_scan_kernel's
        // emission path changed the answers of its hand-scheduled multi-tile builds -- one query of a hundred lost the
        // survivors of its last row tiles -- and were taken out again; tools/tighten_debug.py.  In the compiler-scheduled
        // one-tile builds alone they were correct but not worth it: 10 M x 128 without the int8 copy, 3.3 k -> 2.1 k rows
        // re-ranked per query, 0.424 -> 0.451 ms per step: two more launches against 512-byte gathers.)
        const bool wide_tighten = h->opt.dense_tighten != 0 && d_pad > RING_MAX_DPAD && group_q <= TG_MAX_GROUP && nq_pad <= TG_CAP_Q;
        u32 *tg_hist = nullptr, *tg_thr2k = nullptr;
        float *tg_traw = nullptr, *tg_thr2 = nullptr;
        if (wide_tighten) {
            SQ_TRY(s.wave_score.reserve((size_t)n_waves * wave_cap * 4));
            SQ_TRY(s.tg.reserve((size_t)TG_CAP_Q * (TG_BINS + 3) * 4));   // (fixed layout: the histogram region never meets old thresholds)
            if (s.tg_zeroed != s.tg.p) {   // a new allocation: wiped once, dense_tighten_thr_kernel leaves the histogram clean
                SQ_HIP(hipMemsetAsync(s.tg.p, 0, s.tg.cap, st));
                s.tg_zeroed = s.tg.p;
            }
            tg_hist = s.tg.as<u32>();
            tg_traw = reinterpret_cast<float*>(tg_hist + (size_t)TG_CAP_Q * TG_BINS);
            tg_thr2 = tg_traw + TG_CAP_Q;
            tg_thr2k = reinterpret_cast<u32*>(tg_thr2 + TG_CAP_Q);
        }
        // L2, one query tile per wave (the HBM-bound configuration a pipelined step runs): no prep launch -- the scan
        // kernels build the planes of their query tile themselves and the threshold kernel's prologue does the rest
        // (DenseScanArgs::raw_q, DenseThrPost).  The head of a call is then sample pass -> threshold: on a 1.25 M-row
        // shard the three-launch head (prep 6-27 us beside the neighbours' kernels, sample, threshold) no longer
        // fitted under the previous call's 59 us scan, and the scans did not run back to back.
        const bool fused_prep = !cosine && qt == 1 && qp == 2 && h->opt.dense_fused_prep != 0 && d_pad <= RING_MAX_DPAD;
        const float* centerp = h->center.p ? h->center.as<float>() : nullptr;
        if (!fused_prep)
            hipLaunchKernelGGL(dense_prep_queries_kernel, dim3(nq_pad), dim3(256), 0, st, q, nq, d, d_pad, h->metric, qs,
                               qn2, thr, cnt, oflag, s.q_al.as<float>(), ldq, centerp);
        DenseScanArgs a{};
        if (fused_prep) {
            a.raw_q = q;
            a.raw_nq = nq;
            a.raw_d = d;
            a.center = centerp;
        }
        a.scan = h->scan.as<uint4>();
        a.norms = cosine ? nullptr : (qp == 1 ? h->norms1.as<float>() : h->norms.as<float>());  // n' of this plane count
        a.norm_step = 1;
        // non-temporal stream (32 queries, pipelined step): 10 M rows 0.456 -> 0.401 ms, 5 M 0.251 -> 0.227, 2.5 M 0.147 ->
        // 0.139.  One-tile kernel: everything behind a cacheable head of 192 MB ("dense_nt_keep_mb"); multi-tile kernels
        // with one query group: their non-temporal build when the copy is far beyond the MALL.
        {
            const size_t copy_bytes = (size_t)h->n_pad * d_pad * 2;
            const long long keep_mb = h->opt.dense_nt_keep_mb > 0 ? h->opt.dense_nt_keep_mb : 192;
            a.nt = h->opt.dense_nt >= 0 ? h->opt.dense_nt : (copy_bytes > ((size_t)512 << 20) ? 1 : 0);
            a.nt_from_row = h->opt.dense_nt == 0 ? 0x7fffffffffffffffll : h->opt.dense_nt == 1 ? 0ll : (keep_mb << 20) / ((long long)d_pad * 2);
        }
        if (cosine && (qt > 1 || qp == 1)) {  // the AGPR configurations always stream a norm piece
            a.norms = h->zeros.as<float>();
            a.norm_step = 0;
        }
        a.n = n;
        a.n_tiles = n_tiles;
        a.qs = qs;
        a.thr = thr;
        a.wave_out = s.wave_out.as<uint2>();
        a.wave_cnt = s.wave_cnt.as<u32>();
        a.wave_cap = wave_cap;
        a.sample_out = s.sample.as<float>();
        a.ns = ns;
        a.nqt = nqt;
        a.debug = h->opt.dense_debug;
        // sample pass
        a.tile_step = stride;
        a.n_sel = ns_tiles;
        a.nrb = nrb;
        // Pipelined one-group calls: the sample pass is a scan kernel too (one workgroup wants most of a CU's LDS), so on
        // as many workgroups as the full pass it cannot run BESIDE the neighbouring call's full pass (192 of the 256
        // CUs) -- its tail queues behind that pass and the two scans of a step serialise (a 1.25 M-row shard: 16 + 63 us
        // of an 83 us step).  On the spare quarter of the CUs it runs wholly under the neighbour's full pass.
        if (use_event && h->opt.dense_async_streams == 2 && nqt == 1 && h->opt.dense_blocks <= 0) {
            int sb = h->opt.dense_sample_blocks > 0 ? h->opt.dense_sample_blocks : (h->opt.dense_sample_blocks < 0 ? nrb : cus - nrb);
            sb = (sb + 7) / 8 * 8;
            if (sb >= 8 && sb < a.nrb) a.nrb = sb;
        }
        {
            const long long work = sample_runs ? (ns_tiles + 3) / 4 : ns_tiles;  // runs / tiles a wave takes at a time
            if (work < (long long)nrb * wv) a.nrb = (int)(((work + wv - 1) / wv + 7) / 8 * 8);
        }
        SQ_TRY(scan_launch<true>(h->opt, a, d_pad, qt, qp, st));
        DenseThrPost tp{qn2, cosine ? 1 : 0, filter_bound(cosine ? 1 : 0, eps_a, eps_b, h->xn2_max)};
        {
            tp.traw_out = tg_traw;
            if (fused_prep) {
                tp.raw_q = q;
                tp.nq = nq;
                tp.nq_pad = nq_pad;
                tp.d = d;
                tp.ldq = ldq;
                tp.center = centerp;
                tp.qn2_out = qn2;
                tp.thr_out = thr;
                tp.cnt = cnt;
                tp.oflag = oflag;
                tp.q_al = s.q_al.as<float>();
            }
            hipLaunchKernelGGL((kth_threshold_f32_kernel<DenseThrPost>), dim3(nq), dim3(1024), 0, st, a.sample_out, ns, kk, thr, tp);
        }
        // full pass
        a.tile_step = 1;
        a.n_sel = n_tiles;
        a.nrb = nrb;
        if (prof) SQ_HIP(hipEventRecord(s.ev[1], st));
        a.wave_score = wide_tighten ? s.wave_score.as<float>() : nullptr;
        SQ_TRY(scan_launch<false>(h->opt, a, d_pad, qt, qp, st));
        if (prof) SQ_HIP(hipEventRecord(s.ev[2], st));
        if (wide_tighten) {
            DenseThrPost tp2 = tp;   // (the same slack rule, nothing else)
            tp2.traw_out = nullptr;
            tp2.raw_q = nullptr;
            hipLaunchKernelGGL(dense_tighten_hist_kernel, dim3(nqt > 4 ? 16 : 64, nqt), dim3(256), 0, st, a.wave_out, a.wave_score, a.wave_cnt,
                               wave_cap, n_waves, (const float*)thr, (const float*)tg_traw, group_q, tg_hist);
            hipLaunchKernelGGL((dense_tighten_thr_kernel<DenseThrPost>), dim3((nq_pad + 63) / 64), dim3(64), 0, st, tg_hist, (const float*)thr,
                               (const float*)tg_traw, nq, nq_pad, kk, tp2, tg_thr2, tg_thr2k);
        }
        c.stats.scan_launches = 2;
        c.stats.bytes_scanned = h->n_pad * ((long long)d_pad * 2 + (cosine ? 0 : 4));
        // exact re-rank of the survivors (wave segments -> per-query keys), select, certify
        // A re-rank workgroup takes `wpb` survivor segments of one scan workgroup (its waves share the query
        // group), 128 threads per segment.  Many small workgroups win: the kernel is a chain of dependent
        // memory round trips per workgroup, not the per-query atomics (2 vs 8 segments: 37 vs 50 us at 10 M rows).
        // (L2, four-wave scan workgroups, i.e. multi-tile batches: 4 measured 2 % faster than 2; the cosine kernel's
        // row stage is sized for 256 threads)
        int wpb = (wv == 4 && !cosine) ? 4 : 2;
        if (h->opt.dense_rerank_segments > 0 && wv % h->opt.dense_rerank_segments == 0) wpb = h->opt.dense_rerank_segments;
        const unsigned rr_threads = (unsigned)std::min(512, 128 * wpb);
        const size_t rr_lds = (qt == 1 && ldq <= 156) ? (size_t)32 * (ldq + 4) * 4 : 0;  // the query tile in LDS (rerank_block)
        const unsigned gxr = (unsigned)((n_waves + wpb - 1) / wpb);
        if (cosine) {
            if (wide_tighten)
                hipLaunchKernelGGL((dense_rerank_filtered_kernel<K128, true>), dim3(gxr), dim3(rr_threads), rr_lds, st, h->db, h->ld, d,
                                   s.q_al.as<float>(), ldq, a.wave_out, a.wave_cnt, wave_cap, n_waves, wpb, nq, group_q, s.keys.as<K128>(), cnt,
                                   cap, oflag, cnx, cnq, h->opt.dense_debug, (const float*)a.wave_score, (const float*)tg_thr2);
            else
                hipLaunchKernelGGL(dense_rerank_cos_kernel, dim3(gxr), dim3(rr_threads), rr_lds, st, h->db, h->ld, d, s.q_al.as<float>(),
                                   ldq, a.wave_out, a.wave_cnt, wave_cap, n_waves, wpb, nq, group_q, s.keys.as<K128>(), cnt,
                                   cap, oflag, cnx, cnq, h->opt.dense_debug);
            if (prof) SQ_HIP(hipEventRecord(s.ev[4], st));
            DenseFinalizeCos fin{cnt, cap, kk, h->id_base, thr, eps_a + eps_b, 1, (double*)out_dist, out_idx, hs_dev, hs_raw_dev, oflag, 0};
            if (wide_tighten) fin.thr2k = tg_thr2k;
            SQ_TRY(select_launch_t<K128>(s.keys.as<K128>(), cnt, cap, key_stride, k, nq, s.out_keys.as<K128>(), fin, st, s.sort_tmp));
        } else {
            if (wide_tighten)
                hipLaunchKernelGGL((dense_rerank_filtered_kernel<u64, false>), dim3(gxr), dim3(rr_threads), rr_lds, st, h->db, h->ld, d,
                                   s.q_al.as<float>(), ldq, a.wave_out, a.wave_cnt, wave_cap, n_waves, wpb, nq, group_q, s.keys.as<u64>(), cnt,
                                   cap, oflag, (const double*)nullptr, (const double*)nullptr, h->opt.dense_debug, (const float*)a.wave_score,
                                   (const float*)tg_thr2);
            else
                hipLaunchKernelGGL(dense_rerank_l2_kernel, dim3(gxr), dim3(rr_threads), rr_lds, st, h->db, h->ld, d, s.q_al.as<float>(),
                                   ldq, a.wave_out, a.wave_cnt, wave_cap, n_waves, wpb, nq, group_q, s.keys.as<u64>(), cnt,
                                   cap, oflag, h->opt.dense_debug);
            if (prof) SQ_HIP(hipEventRecord(s.ev[4], st));
            DenseFinalizeL2 fin{cnt, cap, kk, h->id_base, thr, qn2, 0.5 * eps_a + eps_b, 1, (float*)out_dist, out_idx, hs_dev, hs_raw_dev, oflag, 0};
            if (wide_tighten) fin.thr2k = tg_thr2k;
            SQ_TRY(select_launch_t<u64>(s.keys.as<u64>(), cnt, cap, key_stride, k, nq, s.out_keys.as<u64>(), fin, st, s.sort_tmp,
                                        4 * stride * kk));   // ~2.7 stride k candidates per query on N(0,1) data
        }
    } else {
        c.all_fallback = true;  // rows wider than the MFMA scan covers: exact path for every query
    }
    if (prof) SQ_HIP(hipEventRecord(s.ev[3], st));
    if (use_event) {
        if (!s.ev_done) SQ_HIP(hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming));
        SQ_HIP(hipEventRecord(s.ev_done, st));
    }
    c.pending = true;
    return SQ_OK;
}

// Finish the call enqueued on slot `s`: wait for its kernels, collect the statistics, and redo every query the
// filter could not certify on the exact path (synchronously, on the call's stream).  h->stats = this call's.
static int dense_resolve(DenseHandle* h, DenseSlot& s) {
    DenseCall& c = s.call;
    if (!c.pending) return SQ_OK;
    c.pending = false;
    const long long n = h->n;
    const int d = h->d;
    const int nq = c.nq, k = c.k;
    const int kk = (int)(k < n ? k : n);
    const bool cosine = h->metric == SQ_METRIC_COSINE;
    const size_t key_bytes = cosine ? sizeof(K128) : sizeof(u64);
    const u32 cap = c.cap;
    const bool force_fb = h->opt.force_fallback != 0;
    const bool small = c.small, all_fallback = c.all_fallback;
    hipStream_t st = c.st;
    const float* q = c.q;
    void* out_dist = c.out_dist;
    long long* out_idx = c.out_idx;
    u32* cnt = s.cnt.as<u32>();
    float* thr = s.thr.as<float>();
    double* qn2 = s.qn2.as<double>();
    const double* cnx = h->cos_nx.as<double>();
    const double* cnq = s.cos_nq.as<double>();
    u32* hs_raw = reinterpret_cast<u32*>(s.status_host.p);
    u32* hs = hs_raw + c.nq_pad;
    u32* hs_raw_dev = nullptr;
    SQ_TRY(s.status_host.device_ptr(reinterpret_cast<void**>(&hs_raw_dev)));
    u32* hs_dev = hs_raw_dev + c.nq_pad;
    const size_t l2_lds = (size_t)((d + 3) / 4 * 4) * 4;
    h->stats = c.stats;
    if (!all_fallback) {
        SQ_HIP(c.use_event ? event_wait(s.ev_done) : stream_wait(st));  // counts and status words are in hs_raw / hs now
        SQ_HIP(hipGetLastError());
        double clk_tail_ms = -1.0;
        if (s.clk8_wgs > 0 && (h->opt.dense_debug & 8192)) {   // measurement: where the body kernel's workgroups spent their time (100 MHz clock; + 16384: printed)
            std::vector<long long> ck((size_t)s.clk8_wgs * 8);
            SQ_HIP(hipMemcpy(ck.data(), s.clk8.p, ck.size() * 8, hipMemcpyDeviceToHost));
            long long t0 = ck[0], tend = 0;
            for (int w = 0; w < s.clk8_wgs; ++w) t0 = std::min(t0, ck[(size_t)w * 8]), tend = std::max(tend, ck[(size_t)w * 8 + 4]);
            double sum[5] = {0, 0, 0, 0, 0}, mx[5] = {0, 0, 0, 0, 0}, mn[5] = {1e30, 1e30, 1e30, 1e30, 1e30};
            for (int w = 0; w < s.clk8_wgs; ++w) {
                const long long* e = &ck[(size_t)w * 8];
                const double v[5] = {(e[0] - t0) * 0.01, (e[1] - e[0]) * 0.01, (e[2] - e[1]) * 0.01, (e[3] - e[2]) * 0.01, (e[4] - e[3]) * 0.01};
                for (int j = 0; j < 5; ++j) sum[j] += v[j], mx[j] = std::max(mx[j], v[j]), mn[j] = std::min(mn[j], v[j]);
            }
            const char* names[5] = {"start offset", "stream (first wave out)", "wave skew (all waves out)", "thresholds", "re-rank"};
            if (h->opt.dense_debug & 16384) {
                fprintf(stderr, "[body clocks] %d workgroups, kernel span %.1f us\n", s.clk8_wgs, (tend - t0) * 0.01);
                for (int j = 0; j < 5; ++j) fprintf(stderr, "   %-28s min %8.1f  mean %8.1f  max %8.1f us\n", names[j], mn[j], sum[j] / s.clk8_wgs, mx[j]);
            }
            // the kernel's own split for sq_stats_t (set below when the call was timed): the part after the last wave of any
            // workgroup has left the stream is the tail (thresholds + re-rank)
            long long stream_end = 0;
            for (int w = 0; w < s.clk8_wgs; ++w) stream_end = std::max(stream_end, ck[(size_t)w * 8 + 2]);
            clk_tail_ms = (double)(tend - stream_end) * 1e-5;
            s.clk8_wgs = 0;
        }
        if (c.prof) {
            float t1 = 0, t2 = 0;
            SQ_HIP(hipEventElapsedTime(&t1, s.ev[1], s.ev[2]));
            SQ_HIP(hipEventElapsedTime(&t2, s.ev[0], s.ev[3]));
            h->stats.scan_ms = t1;
            h->stats.total_ms = t2;
            if (h->ev_ref) {   // when the call's head / scan / re-rank / select ended, on the clock of the first traced call
                float e[5] = {0, 0, 0, 0, 0};
                const int order[5] = {0, 1, 2, 4, 3};
                for (int j = 0; j < 5; ++j) (void)hipEventElapsedTime(&e[j], h->ev_ref, s.ev[order[j]]);
                fprintf(stderr, "[smqtk_hip] trace slot %d: start %.1f | head-> %.1f | scan-> %.1f | rerank-> %.1f | select-> %.1f us\n",
                        (int)(&s - h->slot), e[0] * 1e3f, e[1] * 1e3f, e[2] * 1e3f, e[3] * 1e3f, e[4] * 1e3f);
            }
            if (c.stats.scan_launches == 2) {  // (the filter path: a re-rank kernel followed the scan)
                float t3 = 0;
                SQ_HIP(hipEventElapsedTime(&t3, s.ev[2], s.ev[4]));
                h->stats.rerank_ms = t3;
                // the fused int8 call re-ranks inside the full-pass kernel: with its clocks recorded (dense_debug & 8192) the
                // tail's share of scan_ms is reported as rerank_ms
                if (clk_tail_ms >= 0.0) h->stats.rerank_ms = clk_tail_ms;
            }
        }
        for (int qi = 0; qi < nq; ++qi) h->stats.candidates += hs_raw[qi];
        if (c.int8) {
            // data the measured bound does not suit (every row of a tight cluster inside the slack): the lists overflow call
            // after call and each query pays a second tier -- after three such calls in a row the handle goes back to bf16
            int over = 0;
            for (int qi = 0; qi < nq; ++qi) over += (hs[qi] & 1u) ? 1 : 0;
            long long cands = 0;
            for (int qi = 0; qi < nq; ++qi) cands += hs_raw[qi];
            // (... or pass so many rows that the re-rank outweighs the bytes saved: the bf16 filter passes ~50 k of 10 M)
            const bool heavy = 2 * over > nq || cands > (long long)nq * std::max<long long>(24ll * 1024, n / 128);
            h->overflow8 = heavy ? h->overflow8 + 1 : 0;
            if (h->overflow8 >= 3 && h->opt.dense_int8 < 0) h->suspended8 = true;
        }
        if (!small && (!c.int8 || h->opt.dense_int8 > 0)) {   // (an int8 stage in automatic mode suspends itself first, above)
            int over = 0;
            for (int qi = 0; qi < nq; ++qi) over += (hs[qi] & 1u) ? 1 : 0;
            const bool heavy = 2 * over > nq;
            if (h->first_suspended) {          // this call was a probe
                if (heavy) {
                    h->probe_interval = std::min(1024u, 2u * h->probe_interval);
                    h->direct_calls = 0;   // (the next probe a whole interval from here)
                } else {
                    h->first_suspended = false;
                    h->overflow16 = 0;
                    h->probe_interval = 16u;
                }
            } else {
                h->overflow16 = heavy ? h->overflow16 + 1 : 0;
                if (h->overflow16 >= 3) {
                    h->first_suspended = true;
                    h->direct_calls = 0;
                }
            }
        }
    }
    // Exact full-keys path, a group of up to 8 queries per pass over the matrix (dense_exact_group_kernel):
    // exact keys for all n rows, then a two-level select -- the k-th smallest of every fb_stride-th exact
    // distance bounds the k-th of all, the keys at or below it (~2 k fb_stride of them) are compacted by the
    // whole device and the one-workgroup select runs on those.  (That select over all n keys directly: 13 ms
    // per query at 10 M rows against 1.2 ms for the keys.)  Ties that overflow the compacted list, or too few
    // finite distances, go on to the select over all keys.
    long long fb_stride = std::min<long long>(64, std::min<long long>((long long)cap / (4ll * kk), n / (8ll * kk)));
    if (n < 65536 || fb_stride < 2 || (h->opt.dense_debug & 128)) fb_stride = 0;  // debug 128: measurement, full select
    const long long fb_ns = fb_stride ? (n + fb_stride - 1) / fb_stride : 0;
    std::vector<int> todo;
    for (int qi = 0; qi < nq; ++qi)
        if (all_fallback || (!small && (hs[qi] != 0 || force_fb))) todo.push_back(qi);
    if (todo.empty()) return SQ_OK;
    // ---- middle tier (sq_dense_mid.hpp): the uncertified queries of a filtered L2 call, 32 per pass over the float32
    // rows, scored with 64 times less slack; whatever it certifies is final, the rest goes on to the exact path
    if (c.mid_direct)
        for (int qi = 0; qi < nq; ++qi) hs[qi] = 1u;   // (no first-filter result to take a bound from: DenseMid*ThrPost)
    if ((!all_fallback || c.mid_direct) && !small && !force_fb && h->opt.dense_mid_tier != 0 && dense_mid_shape_ok(h)) {
        const int d_pad = h->d_pad;
        const int ldq = (d + 3) / 4 * 4;
        const double eps_b_mid = (4.0 * d_pad + 8.0) * 1.1920928955078125e-07;
        const FilterBound fb_mid = filter_bound(0, kEpsAMid, eps_b_mid, h->xn2_max);
        const double alpha1 = filter_bound(0, kEpsA2, dense_eps_b(d_pad), 0.0).alpha;
        float c1 = (float)((1.0 - fb_mid.alpha) / (1.0 - alpha1) * (1.0 - 4.8e-7));
        const int waves = (size_t)TILE_ROWS * d_pad * 4 + (size_t)d_pad * 4 + (size_t)8 * MID_NSTAGE * MID_SLOT_BYTES <= 160 * 1024 - 64 ? 8 : 4;
        const size_t mid_lds = (size_t)TILE_ROWS * d_pad * 4 + (size_t)d_pad * 4 + (size_t)waves * MID_NSTAGE * MID_SLOT_BYTES;
        const int cus = cu_count(h->device);
        const long long n_tiles = (n + 31) / 32;
        int nrb = cus;
        if ((long long)nrb * waves > n_tiles) nrb = (int)((n_tiles + waves - 1) / waves);
        const long long n_waves = (long long)nrb * waves;
        const u32 wave_cap = 2048;
        SQ_TRY(h->mid_q.reserve((size_t)MID_MAX_Q * d * 4));
        SQ_TRY(h->mid_planes.reserve((size_t)MID_MAX_Q * d_pad * 4));
        // [qn2 f64 x32][thr f32 x32][cnt u32 x32][oflag u32 x16][qmap i32 x32]
        // cosine adds [cnq f64 x32][qw f32 x32][lin float2 x32]
        SQ_TRY(h->mid_small.reserve(32 * 8 + 32 * 4 + 32 * 4 + 64 + 32 * 4 + 32 * 8 + 32 * 4 + 32 * 8));
        SQ_TRY(h->mid_qal.reserve((size_t)MID_MAX_Q * ldq * 4));
        SQ_TRY(h->mid_wave_out.reserve((size_t)n_waves * wave_cap * 8));
        SQ_TRY(h->mid_wave_cnt.reserve((size_t)n_waves * 8));
        SQ_TRY(h->mid_keys.reserve((size_t)MID_MAX_Q * cap * key_bytes));
        SQ_TRY(h->mid_out.reserve((size_t)MID_MAX_Q * k * key_bytes));
        double* m_qn2 = h->mid_small.as<double>();
        float* m_thr = reinterpret_cast<float*>(m_qn2 + 32);
        u32* m_cnt = reinterpret_cast<u32*>(m_thr + 32);
        u32* m_oflag = m_cnt + 32;
        int* m_map = reinterpret_cast<int*>(m_oflag + 16);
        double* m_cnq = reinterpret_cast<double*>(m_map + 32);
        float* m_qw = reinterpret_cast<float*>(m_cnq + 32);
        float2* m_lin = reinterpret_cast<float2*>(m_qw + 32);
        static std::atomic<unsigned long long> attr8{0}, attr4{0}, attr8c{0}, attr4c{0};
        if (waves == 8 && !cosine) SQ_TRY(ensure_dyn_lds(reinterpret_cast<const void*>(&dense_mid_scan_kernel<8, false>), 160 * 1024, attr8));
        if (waves == 4 && !cosine) SQ_TRY(ensure_dyn_lds(reinterpret_cast<const void*>(&dense_mid_scan_kernel<4, false>), 160 * 1024, attr4));
        if (waves == 8 && cosine) SQ_TRY(ensure_dyn_lds(reinterpret_cast<const void*>(&dense_mid_scan_kernel<8, true>), 160 * 1024, attr8c));
        if (waves == 4 && cosine) SQ_TRY(ensure_dyn_lds(reinterpret_cast<const void*>(&dense_mid_scan_kernel<4, true>), 160 * 1024, attr4c));
        if (cosine && h->mid_cos_n != n) {
            // the cosine tier's origin and per-row terms (sq_dense_mid.hpp), once per index and again after an append
            if (!h->mid_cos_center.p) {
                SQ_TRY(h->mid_cos_center.reserve((size_t)d_pad * 4));
                DevBuf colsum;
                SQ_TRY(colsum.reserve((size_t)d * 8));
                SQ_HIP(hipMemsetAsync(colsum.p, 0, (size_t)d * 8, st));
                const long long rpb = 512;
                hipLaunchKernelGGL(dense_colsum_kernel, dim3((unsigned)((n + rpb - 1) / rpb)), dim3(256), 0, st, h->db, n, h->ld, d, rpb,
                                   colsum.as<double>());
                hipLaunchKernelGGL(dense_center_kernel, dim3((d_pad + 255) / 256), dim3(256), 0, st, colsum.as<double>(), n, d, d_pad,
                                   h->mid_cos_center.as<float>());
                const hipError_t e = stream_wait(st);
                colsum.release();
                SQ_HIP(e);
            }
            h->mid_cos_ld = (n + 63) / 64 * 64;
            SQ_TRY(h->mid_cos_rows.reserve((size_t)h->mid_cos_ld * 2 * 4));
            hipLaunchKernelGGL(dense_mid_cos_rows_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, st, h->db, n, h->ld, d,
                               (const float*)h->mid_cos_center.as<float>(), (const double*)h->cos_nx.as<double>(),
                               h->mid_cos_rows.as<float>(), h->mid_cos_ld);
            SQ_HIP(hipGetLastError());
            h->mid_cos_n = n;
        }
        // sample of true scores: every mid_stride-th row (about 64 k candidates per query pass the bound it gives)
        long long mid_stride = std::min<long long>(64, std::min<long long>((long long)cap / (8ll * kk), n / (8ll * kk)));
        if (mid_stride < 1) mid_stride = 1;
        const long long mid_ns = (n + mid_stride - 1) / mid_stride;
        const size_t mid_sample_lds = (size_t)d * 33 * 4;
        SQ_TRY(h->mid_sample.reserve((size_t)MID_MAX_Q * mid_ns * 4));
        {
            static std::atomic<unsigned long long> attr_s{0};
            SQ_TRY(ensure_dyn_lds(reinterpret_cast<const void*>(&dense_mid_sample_kernel), 512 * 33 * 4, attr_s));
        }
        std::vector<int> left;
        for (size_t t0 = 0; t0 < todo.size(); t0 += MID_MAX_Q) {
            MidSelection sel{};
            sel.count = (int)std::min<size_t>(MID_MAX_Q, todo.size() - t0);
            for (int j = 0; j < MID_MAX_Q; ++j) sel.idx[j] = todo[t0 + (size_t)(j < sel.count ? j : 0)];
            hipLaunchKernelGGL(dense_mid_gather_kernel, dim3(MID_MAX_Q), dim3(128), 0, st, q, d, sel, h->mid_q.as<float>(), m_map);
            if (cosine) {
                hipLaunchKernelGGL(dense_mid_cos_queries_kernel, dim3(MID_MAX_Q), dim3(256), 0, st, h->mid_q.as<float>(), sel.count, d, d_pad,
                                   (const float*)h->mid_cos_center.as<float>(), eps_b_mid, h->mid_planes.as<uint4>(), m_qn2, m_thr, m_cnt,
                                   m_oflag, h->mid_qal.as<float>(), ldq, m_qw, m_lin);
                hipLaunchKernelGGL(dense_cos_qnorm_kernel, dim3(1), dim3(64), 0, st, (const float*)h->mid_q.as<float>(), sel.count, d, m_cnq);
                hipLaunchKernelGGL(dense_mid_cos_sample_kernel, dim3((unsigned)((mid_ns + 7) / 8)), dim3(256), mid_sample_lds, st, h->db, h->ld,
                                   d, n, mid_stride, mid_ns, (const float*)h->mid_q.as<float>(), (const double*)m_qn2,
                                   (const double*)h->cos_nx.as<double>(), h->mid_sample.as<float>());
                hipLaunchKernelGGL((kth_threshold_f32_kernel<DenseMidCosThrPost>), dim3(sel.count), dim3(1024), 0, st,
                                   h->mid_sample.as<float>(), mid_ns, kk, m_thr,
                                   DenseMidCosThrPost{m_map, (const double*)out_dist, (const u32*)hs_dev, k, kk, (const float2*)m_lin});
            } else {
                hipLaunchKernelGGL(dense_prep_queries_kernel, dim3(MID_MAX_Q), dim3(256), 0, st, h->mid_q.as<float>(), sel.count, d, d_pad,
                                   h->metric, h->mid_planes.as<uint4>(), m_qn2, m_thr, m_cnt, m_oflag, h->mid_qal.as<float>(), ldq,
                                   h->center.p ? h->center.as<float>() : nullptr);
                hipLaunchKernelGGL(dense_mid_sample_kernel, dim3((unsigned)((mid_ns + 7) / 8)), dim3(256), mid_sample_lds, st, h->db, h->ld, d, n,
                                   mid_stride, mid_ns, (const float*)h->mid_q.as<float>(), (const double*)m_qn2, h->mid_sample.as<float>());
                hipLaunchKernelGGL((kth_threshold_f32_kernel<DenseMidThrPost>), dim3(sel.count), dim3(1024), 0, st, h->mid_sample.as<float>(),
                                   mid_ns, kk, m_thr,
                                   DenseMidThrPost{m_map, (const float*)out_dist, (const u32*)hs_dev, k, kk, m_qn2, fb_mid.beta});
            }
            DenseMidArgs a{};
            a.x = h->db;
            a.n = n;
            a.ld = h->ld;
            a.d = d;
            a.center = h->center.p ? h->center.as<float>() : nullptr;
            a.norms = h->norms.as<float>();
            a.norm_scale = c1;
            a.qs = h->mid_planes.as<uint4>();
            a.d_pad = d_pad;
            a.thr = m_thr;
            a.wave_out = h->mid_wave_out.as<uint2>();
            a.wave_cnt = h->mid_wave_cnt.as<u32>();
            a.wave_cap = wave_cap;
            a.n_tiles = n_tiles;
            a.nrb = nrb;
            a.rowstat = cosine ? h->mid_cos_rows.as<float>() : nullptr;
            a.rowstat_ld = h->mid_cos_ld;
            a.qw = m_qw;
            if (cosine) {
                if (waves == 8)
                    hipLaunchKernelGGL((dense_mid_scan_kernel<8, true>), dim3(nrb), dim3(512), mid_lds, st, a);
                else
                    hipLaunchKernelGGL((dense_mid_scan_kernel<4, true>), dim3(nrb), dim3(256), mid_lds, st, a);
            } else if (waves == 8) {
                hipLaunchKernelGGL((dense_mid_scan_kernel<8, false>), dim3(nrb), dim3(512), mid_lds, st, a);
            } else {
                hipLaunchKernelGGL((dense_mid_scan_kernel<4, false>), dim3(nrb), dim3(256), mid_lds, st, a);
            }
            const int wpb = 2;
            const size_t rr_lds = ldq <= 156 ? (size_t)32 * (ldq + 4) * 4 : 0;
            if (cosine) {
                hipLaunchKernelGGL(dense_rerank_cos_kernel, dim3((unsigned)((n_waves + wpb - 1) / wpb)), dim3(128 * wpb), rr_lds, st, h->db, h->ld,
                                   d, h->mid_qal.as<float>(), ldq, a.wave_out, a.wave_cnt, wave_cap, n_waves, wpb, sel.count, TILE_ROWS,
                                   h->mid_keys.as<K128>(), m_cnt, cap, m_oflag, cnx, (const double*)m_cnq, 0);
                DenseFinalizeCos fin{m_cnt, cap, kk, h->id_base, m_thr, 0.0, 1, (double*)out_dist, out_idx, hs_dev, nullptr, m_oflag, 0};
                fin.qmap = m_map;
                fin.lin = m_lin;
                SQ_TRY(select_launch_t<K128>(h->mid_keys.as<K128>(), m_cnt, cap, (long long)cap, k, sel.count, h->mid_out.as<K128>(), fin, st,
                                             h->fb_sort));
            } else {
                hipLaunchKernelGGL(dense_rerank_l2_kernel, dim3((unsigned)((n_waves + wpb - 1) / wpb)), dim3(128 * wpb), rr_lds, st, h->db, h->ld, d,
                                   h->mid_qal.as<float>(), ldq, a.wave_out, a.wave_cnt, wave_cap, n_waves, wpb, sel.count, TILE_ROWS,
                                   h->mid_keys.as<u64>(), m_cnt, cap, m_oflag, 0);
                DenseFinalizeL2 fin{m_cnt, cap, kk, h->id_base, m_thr, m_qn2, fb_mid.beta, 1, (float*)out_dist, out_idx, hs_dev, nullptr, m_oflag, 0};
                fin.qmap = m_map;
                SQ_TRY(select_launch_t<u64>(h->mid_keys.as<u64>(), m_cnt, cap, (long long)cap, k, sel.count, h->mid_out.as<u64>(), fin, st,
                                            h->fb_sort));
            }
            h->stats.scan_launches++;
            h->stats.bytes_scanned += n * (long long)d * 4;
            h->stats.mid_tier_queries += sel.count;
            SQ_HIP(stream_wait(st));  // the status words of these queries are in hs now
            SQ_HIP(hipGetLastError());
            for (int j = 0; j < sel.count; ++j)
                if (hs[sel.idx[j]] != 0) left.push_back(sel.idx[j]);
        }
        todo.swap(left);
        if (todo.empty()) return SQ_OK;
    }
    // group size: the key arrays of a group stay under 4 GB; rows beyond the group kernel's depth go one by one
    int gmax = (int)std::min<long long>(EXACT_GROUP, std::max<long long>(1, (4ll << 30) / (n * (long long)key_bytes)));
    if (d > (128 << EXACT_GROUP_DEPTH) || (h->opt.dense_debug & 256)) gmax = 1;    // debug 256: measurement, one query per pass
    const size_t grp_lds = (size_t)EXACT_GROUP * ((d + 3) / 4 * 4) * 4;
    const bool grp_ok = grp_lds <= 160 * 1024 - 256 && d <= (128 << EXACT_GROUP_DEPTH);
    if (grp_ok) {
        SQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_exact_group_kernel<false, u64>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)grp_lds));
        SQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_exact_group_kernel<true, K128>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)grp_lds));
    }
    // the exact path keeps its own per-query counters: the slot's belong to a call that may still be in flight
    // when another asynchronous call's queries are redone
    SQ_TRY(h->fb_cnt.reserve((size_t)(nq + 64) * 4));
    u32* full_cnt = h->fb_cnt.as<u32>();
    for (size_t t0 = 0; t0 < todo.size(); t0 += (size_t)gmax) {
        const int gn = (int)std::min<size_t>((size_t)gmax, todo.size() - t0);
        ExactGroup grp{};
        for (int g = 0; g < EXACT_GROUP; ++g) grp.idx[g] = todo[t0 + (size_t)(g < gn ? g : 0)];
        grp.count = gn;
        h->stats.fallback_queries += gn;
        SQ_TRY(h->big_keys.reserve((size_t)gn * n * key_bytes));
        float* fb_sample = nullptr;
        float* fb_thr = nullptr;
        u32* fb_cnt = nullptr;
        if (fb_stride) {
            SQ_TRY(h->fb_sample.reserve((size_t)gn * fb_ns * 4 + 256));
            SQ_TRY(h->fb_keys.reserve((size_t)gn * cap * key_bytes));
            fb_thr = h->fb_sample.as<float>();               // [8] thresholds | [8] counts | samples
            fb_cnt = reinterpret_cast<u32*>(fb_thr + 8);
            fb_sample = fb_thr + 64;
        }
        unsigned gx = (unsigned)((n + 255) / 256);
        if (gx > 8192) gx = 8192;
        if (grp_ok) {
            if (cosine)
                hipLaunchKernelGGL((dense_exact_group_kernel<true, K128>), dim3(gx), dim3(256), grp_lds, st, h->db, h->ld, d, q,
                                   grp, n, h->big_keys.as<K128>(), fb_sample, fb_ns, (int)fb_stride, cnx, cnq);
            else
                hipLaunchKernelGGL((dense_exact_group_kernel<false, u64>), dim3(gx), dim3(256), grp_lds, st, h->db, h->ld, d, q,
                                   grp, n, h->big_keys.as<u64>(), fb_sample, fb_ns, (int)fb_stride, nullptr, nullptr);
        } else {  // one query per pass (gmax == 1 here)
            const int qi = grp.idx[0];
            if (cosine)
                hipLaunchKernelGGL(dense_exact_cos_kernel, dim3(gx, 1), dim3(256), 0, st, h->db, h->ld, d,
                                   q + (long long)qi * d, nullptr, full_cnt + qi, (u32)n, n, 0ll, h->big_keys.as<K128>(), n, cnx,
                                   cnq + qi, fb_sample, (int)fb_stride);
            else
                hipLaunchKernelGGL(dense_exact_l2_kernel, dim3(gx, 1), dim3(256), l2_lds, st, h->db, h->ld, d,
                                   q + (long long)qi * d, nullptr, full_cnt + qi, (u32)n, n, 0ll, h->big_keys.as<u64>(), n,
                                   fb_sample, (int)fb_stride);
        }
        h->stats.scan_launches++;
        h->stats.bytes_scanned += n * (long long)d * 4;
        bool done[EXACT_GROUP] = {false, false, false, false, false, false, false, false};
        if (fb_stride) {
            hipLaunchKernelGGL((kth_threshold_f32_kernel<KthIdentity>), dim3(gn), dim3(1024), 0, st, fb_sample, fb_ns, kk, fb_thr,
                               KthIdentity{});
            hipLaunchKernelGGL(fill_u32_kernel, dim3(1), dim3(64), 0, st, fb_cnt, (long long)gn, 0u);
            const unsigned gc = (unsigned)std::min<long long>((n + 1023) / 1024, 512);
            if (cosine) {
                hipLaunchKernelGGL((dense_compact_keys_kernel<K128>), dim3(gc, gn), dim3(256), 0, st, h->big_keys.as<K128>(), n,
                                   fb_thr, h->fb_keys.as<K128>(), cap, fb_cnt);
                SQ_TRY(h->fb_out.reserve((size_t)gn * k * key_bytes));
                SQ_TRY(select_launch_t<K128>(h->fb_keys.as<K128>(), fb_cnt, cap, (long long)cap, k, gn, h->fb_out.as<K128>(),
                                             DenseFinalizeCos{fb_cnt, cap, kk, h->id_base, thr, 0.0, 2, (double*)out_dist, out_idx,
                                                              hs_dev, nullptr, nullptr, 0, grp},
                                             st, h->fb_sort));
            } else {
                hipLaunchKernelGGL((dense_compact_keys_kernel<u64>), dim3(gc, gn), dim3(256), 0, st, h->big_keys.as<u64>(), n,
                                   fb_thr, h->fb_keys.as<u64>(), cap, fb_cnt);
                SQ_TRY(h->fb_out.reserve((size_t)gn * k * key_bytes));
                SQ_TRY(select_launch_t<u64>(h->fb_keys.as<u64>(), fb_cnt, cap, (long long)cap, k, gn, h->fb_out.as<u64>(),
                                            DenseFinalizeL2{fb_cnt, cap, kk, h->id_base, thr, qn2, 0.0, 2,
                                                            (float*)out_dist, out_idx, hs_dev, nullptr, nullptr, 0, grp},
                                            st, h->fb_sort));
            }
            SQ_HIP(stream_wait(st));  // the status words of the group are in hs now
            SQ_HIP(hipGetLastError());
            for (int g = 0; g < gn; ++g) done[g] = hs[grp.idx[g]] == 0;
        }
        for (int g = 0; g < gn; ++g) {
            if (done[g]) continue;
            const int qi = grp.idx[g];
            hipLaunchKernelGGL(fill_u32_kernel, dim3(1), dim3(64), 0, st, full_cnt + qi, 1ll, (u32)n);
            SQ_TRY(h->fb_out.reserve((size_t)k * key_bytes));
            if (cosine) {
                SQ_TRY(select_launch_t<K128>(h->big_keys.as<K128>() + (long long)g * n, full_cnt + qi, (u32)n, n, k, 1,
                                             h->fb_out.as<K128>(),
                                             DenseFinalizeCos{full_cnt, (u32)n, kk, h->id_base, thr, 0.0, 0, (double*)out_dist, out_idx,
                                                              hs_dev, nullptr, nullptr, qi},
                                             st, h->fb_sort));
            } else {
                SQ_TRY(select_launch_t<u64>(h->big_keys.as<u64>() + (long long)g * n, full_cnt + qi, (u32)n, n, k, 1,
                                            h->fb_out.as<u64>(),
                                            DenseFinalizeL2{full_cnt, (u32)n, kk, h->id_base, thr, qn2, 0.0, 0,
                                                            (float*)out_dist, out_idx, hs_dev, nullptr, nullptr, qi},
                                            st, h->fb_sort));
            }
        }
    }
    SQ_HIP(hipStreamSynchronize(st));
    SQ_HIP(hipGetLastError());
    return SQ_OK;
}

// Finish every asynchronous call still in flight, oldest first.
static int dense_sync_all(DenseHandle* h) {
    for (int j = 0; j < h->depth; ++j)  // slot of call (async_calls - depth + j): oldest first
        SQ_TRY(dense_resolve(h, h->slot[(h->async_calls + (unsigned)j) % (unsigned)h->depth]));
    return SQ_OK;
}

static int dense_search_device(DenseHandle* h, const float* q, int nq, int k, void* out_dist, long long* out_idx,
                               hipStream_t st) {
    SQ_TRY(dense_enqueue(h, h->slot[0], q, nq, k, out_dist, out_idx, st, false));
    return dense_resolve(h, h->slot[0]);
}

// Large batches run in chunks so that the per-query workspace (candidate key lists of `cap` keys, the
// sample scores) stays bounded: 4096 queries need ~2.4 GB at the default list size.
static constexpr int kDenseQueryChunk = 4096;

static int dense_search_chunked(DenseHandle* h, const float* q, int nq, int k, void* out_dist, long long* out_idx,
                                hipStream_t st) {
    if (nq <= kDenseQueryChunk) return dense_search_device(h, q, nq, k, out_dist, out_idx, st);
    const size_t dsz = h->metric == SQ_METRIC_COSINE ? 8 : 4;
    sq_stats_t total{};
    for (int q0 = 0; q0 < nq; q0 += kDenseQueryChunk) {
        const int m = nq - q0 < kDenseQueryChunk ? nq - q0 : kDenseQueryChunk;
        SQ_TRY(dense_search_device(h, q + (long long)q0 * h->d, m, k, static_cast<char*>(out_dist) + (size_t)q0 * k * dsz,
                                   out_idx + (long long)q0 * k, st));
        total.scan_ms += h->stats.scan_ms;
        total.rerank_ms += h->stats.rerank_ms;
        total.total_ms += h->stats.total_ms;
        total.scan_launches += h->stats.scan_launches;
        total.candidates += h->stats.candidates;
        total.fallback_queries += h->stats.fallback_queries;
        total.bytes_scanned += h->stats.bytes_scanned;
    }
    h->stats = total;
    return SQ_OK;
}

// Row statistics and the bfloat16 scan copy of rows [row_base, n) -- row_base a multiple of 32 -- plus the padding
// rows of the last tile: the whole matrix at create (row_base = 0), the new rows on append.  The buffers are
// already large enough (dense_grow); the largest squared norm accumulates across calls.
static int dense_build_rows(DenseHandle* h, long long row_base) {
    const long long n = h->n, n_pad = h->n_pad;
    const int d = h->d, d_pad = h->d_pad;
    const bool cosine = h->metric == SQ_METRIC_COSINE;
    SQ_TRY(h->scratch.reserve(256));
    float prev = (float)h->xn2_max;
    SQ_HIP(hipMemset(h->scratch.p, 0, 256));
    SQ_HIP(hipMemcpy(h->scratch.p, &prev, 4, hipMemcpyHostToDevice));  // non-negative floats order like their bits
    DevBuf inv;  // cosine: 1/|x| of the rows being built, indexed by row - row_base
    float* invp = nullptr;
    if (cosine) {
        SQ_TRY(inv.reserve((size_t)(n_pad - row_base) * 4));
        invp = inv.as<float>() - row_base;
        hipLaunchKernelGGL(dense_cos_norm_kernel, dim3((unsigned)((n - row_base + 255) / 256)), dim3(256), 0, 0, h->db, n,
                           h->ld, d, h->cos_nx.as<double>(), row_base);
    }
    float* centerp = h->center.p ? h->center.as<float>() : nullptr;
    float* norms1p = nullptr;
    double shrink2 = 1.0, shrink1 = 1.0;
    if (!cosine) {
        // the row's share alpha |x|^2 of the filter's error bound comes off the stored norm (FilterBound)
        norms1p = h->norms1.as<float>();
        shrink2 = 1.0 - filter_bound(0, kEpsA2, dense_eps_b(d_pad), 0.0).alpha;
        shrink1 = 1.0 - filter_bound(0, kEpsA1, dense_eps_b(d_pad), 0.0).alpha;
    }
    hipLaunchKernelGGL(dense_rowstats_kernel, dim3((unsigned)((n_pad - row_base) / 32)), dim3(256), 0, 0, h->db, n, h->ld,
                       d, n_pad, h->scratch.as<u32>(), h->norms.as<float>(), invp, centerp, norms1p, shrink2, shrink1,
                       row_base);
    if (d_pad <= MAX_DPAD) {
        const long long chunks = (n_pad - row_base) * (long long)(d_pad / 8);
        hipLaunchKernelGGL(dense_build_scan_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, 0, h->db, n,
                           h->ld, d, d_pad, n_pad, invp, centerp, h->scan.as<uint4>(), row_base);
    }
    u32 bits = 0;
    hipError_t e = hipMemcpy(&bits, h->scratch.p, 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    inv.release();
    if (e != hipSuccess) return fail(SQ_ERR_HIP, "dense index build failed: %s", hipGetErrorString(e));
    float f;
    memcpy(&f, &bits, 4);
    h->xn2_max = (double)f;
    return SQ_OK;
}

// The int8 first-stage copy (sq_dense_i8.hpp), built at create for L2 matrices of up to 128 dimensions: the clamp and the
// residual bound R chosen from the measured residuals of twelve candidate clamps (the pair with the least R that leaves
// no more than 256 rows, or 20 per million, beyond it: those become always-candidates), the copy, its float64 residuals.  Data
// no clamp suits (heavy tails: too many rows beyond every bound) keeps the bf16 filter alone, and so does a failure to
// allocate: neither is an error.
static int dense8_build(DenseHandle* h) {
    h->use8 = false;
    h->suspended8 = false;
    h->overflow8 = 0;
    if (h->d > I8_MAX_ROW_BYTES || h->n < 65536 || h->opt.dense_int8 == 0) return SQ_OK;
    const bool cosine = h->metric == SQ_METRIC_COSINE;
    const double* nx64 = cosine ? h->cos_nx.as<double>() : nullptr;   // cosine: the copy holds the unit-length rows
    const long long n = h->n;
    h->n8_built = n;   // (an attempt counts whether or not the data is accepted: dense8_append tries again when the index has doubled)
    const int d = h->d;
    const int row8 = i8_row_bytes(d);
    const long long n_pad64 = (n + 63) / 64 * 64, n_alloc = (n + 127) / 128 * 128;
    const float* centerp = (!cosine && h->center.p) ? h->center.as<float>() : nullptr;
    DevBuf tmp;   // [sum f64 x2 | max bits u32 x2 | flagged u32]
    DevBuf r2row;
    auto quit = [&](int rc) {
        tmp.release();
        r2row.release();
        if (!h->use8) {
            h->scan8.release();
            h->nrow8.release();
        }
        return rc;
    };
    if (tmp.reserve(64) != SQ_OK || r2row.reserve((size_t)n_alloc * 4) != SQ_OK || h->scan8.reserve((size_t)n_alloc * row8) != SQ_OK ||
        h->nrow8.reserve((size_t)(n_alloc + 64) * 4) != SQ_OK) {   // (+ 64: a 32-row unit's DMA fetches 64 row terms)
        (void)hipGetLastError();
        return quit(SQ_OK);
    }
    const int stat_blocks = cu_count(h->device) * 8;
    double rms = 0.0, cap_e = (double)__builtin_inff();
    for (int pass = 0; pass < 3; ++pass) {
        double er[2] = {0.0, 0.0};
        SQ_HIP(hipMemset(tmp.p, 0, 64));
        switch (row8) {
            case 128: hipLaunchKernelGGL(dense8_energy_kernel<2>, dim3(stat_blocks), dim3(256), 0, 0, h->db, n, h->ld, d, centerp, nx64, cap_e, tmp.as<double>()); break;
            case 256: hipLaunchKernelGGL(dense8_energy_kernel<4>, dim3(stat_blocks), dim3(256), 0, 0, h->db, n, h->ld, d, centerp, nx64, cap_e, tmp.as<double>()); break;
            default: hipLaunchKernelGGL(dense8_energy_kernel<8>, dim3(stat_blocks), dim3(256), 0, 0, h->db, n, h->ld, d, centerp, nx64, cap_e, tmp.as<double>()); break;
        }
        SQ_HIP(hipMemcpy(er, tmp.p, 16, hipMemcpyDeviceToHost));
        if (!(er[1] >= 0.5 * (double)n)) return quit(SQ_OK);   // (half the rows non-finite or beyond 16 x the mean: not this filter's data)
        rms = sqrt(er[0] / (er[1] * d));
        cap_e = 16.0 * er[0] / er[1];
    }
    if (!(rms > 0.0) || !(rms < 1e30)) return quit(SQ_OK);
    SQ_HIP(hipMemset(tmp.p, 0, 64));
    // the clamp and R from the measured residuals of twelve candidate clamps (dense8_clip_stats_kernel)
    // (1.75 rms: where a uniform distribution ends; 9-11 rms: one-sided data -- max(N(0,1), 0) reaches 8.4 rms of its centred
    // self; 14-18 rms: sparse rows -- with 10 % of the elements set their rms is a third of an element's own scale)
    static const double kClip[I8_NCLIP] = {1.75, 2.5, 3.25, 4.0, 4.75, 5.5, 6.5, 7.5, 9.0, 11.0, 14.0, 18.0};
    static const double kCut[I8_NCUT] = {0.6, 0.8, 1.0, 1.10, 1.15, 1.20, 1.30, 1.50, 2.0, 3.0};   // R in units of Dx sqrt(d / 12)
    Dense8ClipArgs ca{};
    const double round_unit = sqrt((double)d / 12.0);
    for (int c = 0; c < I8_NCLIP; ++c) {
        ca.dx[c] = (float)(kClip[c] * rms / 127.0);
        ca.inv_dx[c] = 1.0f / ca.dx[c];
        for (int m = 0; m < I8_NCUT; ++m) {
            const double r = kCut[m] * (double)ca.dx[c] * round_unit;
            ca.cut[c][m] = (float)(r * r);
        }
    }
    DevBuf clipbuf;   // counts u32 [NCLIP][NCUT]
    const size_t clip_bytes = I8_NCLIP * I8_NCUT * 4;
    if (clipbuf.reserve(clip_bytes) != SQ_OK) return quit(SQ_OK);
    SQ_HIP(hipMemset(clipbuf.p, 0, clip_bytes));
    // large matrices choose from every 8th row (the kernel evaluates twelve clamps per element: 16 ms of a 40 ms build at
    // 10 M x 128 when it reads every row; the choice needs the shape of the residuals' tail, not every row)
    const int clip_step = n >= 2000000 ? 8 : 1;
    switch (row8) {
        case 128: hipLaunchKernelGGL(dense8_clip_stats_kernel<2>, dim3(stat_blocks), dim3(256), 0, 0, h->db, n, h->ld, d, centerp, nx64, ca, clipbuf.as<u32>(), clip_step); break;
        case 256: hipLaunchKernelGGL(dense8_clip_stats_kernel<4>, dim3(stat_blocks), dim3(256), 0, 0, h->db, n, h->ld, d, centerp, nx64, ca, clipbuf.as<u32>(), clip_step); break;
        default: hipLaunchKernelGGL(dense8_clip_stats_kernel<8>, dim3(stat_blocks), dim3(256), 0, 0, h->db, n, h->ld, d, centerp, nx64, ca, clipbuf.as<u32>(), clip_step); break;
    }
    u32 counts[I8_NCLIP][I8_NCUT];
    SQ_HIP(hipMemcpy(counts, clipbuf.p, clip_bytes, hipMemcpyDeviceToHost));
    clipbuf.release();
    for (auto& row : counts)
        for (u32& v : row) v = (u32)std::min<unsigned long long>(0xffffffffull, (unsigned long long)v * (unsigned)clip_step);
    // rows beyond R are candidates of every query: a few hundred at most (a query has a few thousand candidates anyway)
    const double budget = std::max(256.0, 2e-5 * (double)n);
    int best_c = -1, best_m = -1;
    double best_r = 0.0;
    for (int c = 0; c < I8_NCLIP; ++c)
        for (int m = 0; m < I8_NCUT; ++m)
            if ((double)counts[c][m] <= budget) {
                const double r = kCut[m] * (double)ca.dx[c] * round_unit;
                if (best_c < 0 || r < best_r) best_c = c, best_m = m, best_r = r;
                break;
            }
    if (best_c < 0) return quit(SQ_OK);   // every clamp leaves too many rows beyond every bound (heavy tails): the bf16 filter's relative bound suits such data
    const double dx = (double)ca.dx[best_c];
    {
        const dim3 grid((unsigned)((n_alloc + 3) / 4)), blk(256);
        signed char* o8 = h->scan8.as<signed char>();
        float* nr8 = h->nrow8.as<float>();
        float* r2p = r2row.as<float>();
        switch (row8) {
            case 128: hipLaunchKernelGGL(dense8_build_kernel<2>, grid, blk, 0, 0, h->db, n, h->ld, d, n_alloc, centerp, nx64, ca.inv_dx[best_c], ca.dx[best_c], o8, nr8, r2p, 0ll); break;
            case 256: hipLaunchKernelGGL(dense8_build_kernel<4>, grid, blk, 0, 0, h->db, n, h->ld, d, n_alloc, centerp, nx64, ca.inv_dx[best_c], ca.dx[best_c], o8, nr8, r2p, 0ll); break;
            default: hipLaunchKernelGGL(dense8_build_kernel<8>, grid, blk, 0, 0, h->db, n, h->ld, d, n_alloc, centerp, nx64, ca.inv_dx[best_c], ca.dx[best_c], o8, nr8, r2p, 0ll); break;
        }
        SQ_HIP(hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(nr8 + n_alloc), 0x7f800000, 64));   // (+inf behind the last unit)
    }
    double* sum_r2 = tmp.as<double>() + 1;
    u32* maxb = reinterpret_cast<u32*>(tmp.as<double>() + 2);
    u32* flagged = maxb + 2;
    // rows beyond the chosen bound become always-candidates first; R and X are then the largest residual and the largest
    // |x'| of the rows that REMAIN under the bound (a row 100x the rest would otherwise set X for every query)
    double cut = best_r * best_r * (1.0 + 1e-4);   // (the choice was made in float32)
    hipLaunchKernelGGL(dense8_flag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, r2row.as<float>(), h->nrow8.as<float>(), n,
                       (float)cut, flagged, 0ll);
    hipLaunchKernelGGL(dense8_resid_stats_kernel, dim3(1024), dim3(256), 0, 0, r2row.as<float>(), h->nrow8.as<float>(), n, sum_r2, maxb);
    struct {
        double energy, sum_r2;
        u32 max_r2, max_n, flagged, pad;
    } host{};
    SQ_HIP(hipMemcpy(&host, tmp.p, sizeof(host), hipMemcpyDeviceToHost));
    SQ_HIP(hipDeviceSynchronize());
    float max_r2, max_n;
    memcpy(&max_r2, &host.max_r2, 4);
    memcpy(&max_n, &host.max_n, 4);
    const u32 nflag = host.flagged;
    if ((double)max_r2 < cut) cut = (double)max_r2 * (1.0 + 1e-6);   // nothing near the bound: R is simply the largest
    if ((double)nflag > 4.0 * budget) return quit(SQ_OK);
    h->dx8 = dx;
    h->rmax8 = sqrt(cut) * (1.0 + 1e-6);
    h->xmax8 = cosine ? 1.0 + 1e-6 : sqrt((double)max_n) * (1.0 + 1e-6);   // (cosine: unit rows; their row term is 0)
    h->flagged8 = nflag;
    if (getenv("SQ_INT8_REPORT"))   // (measurement aid: what the build chose)
        fprintf(stderr, "[smqtk_hip] int8 filter: clamp %.2f rms, step %.5g, R %.5g (%.2f x the rounding residual), X %.5g, %u always-candidate rows of %lld\n",
                kClip[best_c], dx, h->rmax8, kCut[best_m], h->xmax8, nflag, n);
    h->n_pad64 = n_pad64;
    h->n_alloc8 = n_alloc;
    h->row8 = row8;
    h->dxf8 = ca.dx[best_c];
    h->inv_dxf8 = ca.inv_dx[best_c];
    {   // appended rows are flagged against R^2 itself (rounded down): R may be smaller than the bound the build flagged with
        float c8 = (float)cut;
        if ((double)c8 > cut) c8 = __builtin_nextafterf(c8, 0.f);
        h->cut8 = c8;
    }
    h->n8_built = n;
    h->overflow8 = 0;
    h->use8 = true;
    return quit(SQ_OK);
}

// Room for `n_new` rows in every per-row buffer, contents kept (grown by half again at least, so a stream of
// small appends copies the matrix O(log) times).
static int grow_keep(DevBuf& b, size_t used, size_t need) {
    if (need <= b.cap) return SQ_OK;
    DevBuf nb;
    SQ_TRY(nb.reserve(std::max(need, used + used / 2)));
    if (used && b.p) {
        if (hipMemcpy(nb.p, b.p, used, hipMemcpyDeviceToDevice) != hipSuccess) {
            nb.release();
            return fail(SQ_ERR_HIP, "device copy failed while growing an index buffer");
        }
    }
    b.release();
    b = nb;
    return SQ_OK;
}
// The int8 copy after an append (h->n is the new row count; the rows are in place).  The step and the residual cut of
// the build stay: new rows are quantised with them, rows they do not suit become always-candidates like the build's own.
// An index that has doubled since the clamp was chosen -- or that collected too many always-candidates -- chooses again
// from all its rows; one that had no copy (too small, or declined) tries when it has doubled since its last attempt.
static int dense8_append(DenseHandle* h, long long n_old) {
    const long long n = h->n;
    if (h->opt.dense_int8 == 0 || h->d > I8_MAX_ROW_BYTES) return SQ_OK;
    if (!h->use8) {
        if (n >= 65536 && n >= 2 * h->n8_built) {
            h->n8_built = n;   // (the attempt counts whether or not the build accepts the data)
            return dense8_build(h);
        }
        return SQ_OK;
    }
    if (n >= 2 * h->n8_built) return dense8_build(h);
    const bool cosine = h->metric == SQ_METRIC_COSINE;
    const double* nx64 = cosine ? h->cos_nx.as<double>() : nullptr;
    const float* centerp = (!cosine && h->center.p) ? h->center.as<float>() : nullptr;
    const int row8 = h->row8, d = h->d;
    const long long n_pad64 = (n + 63) / 64 * 64, n_alloc = (n + 127) / 128 * 128;
    auto drop = [&]() {
        h->use8 = false;
        h->scan8.release();
        h->nrow8.release();
        return SQ_OK;
    };
    if (grow_keep(h->scan8, (size_t)h->n_alloc8 * row8, (size_t)n_alloc * row8) != SQ_OK ||
        grow_keep(h->nrow8, (size_t)(h->n_alloc8 + 64) * 4, (size_t)(n_alloc + 64) * 4) != SQ_OK) {
        (void)hipGetLastError();
        return drop();
    }
    DevBuf tmp, r2row;
    if (tmp.reserve(64) != SQ_OK || r2row.reserve((size_t)n_alloc * 4) != SQ_OK) {
        tmp.release();
        r2row.release();
        (void)hipGetLastError();
        return drop();
    }
    auto done = [&](int rc) {
        tmp.release();
        r2row.release();
        return rc;
    };
    const long long row_base = n_old / 128 * 128;   // the unit the old rows ended in is redone with the new ones
    {
        const dim3 grid((unsigned)((n_alloc - row_base + 3) / 4)), blk(256);
        signed char* o8 = h->scan8.as<signed char>();
        float* nr8 = h->nrow8.as<float>();
        float* r2p = r2row.as<float>();
        switch (row8) {
            case 128: hipLaunchKernelGGL(dense8_build_kernel<2>, grid, blk, 0, 0, h->db, n, h->ld, d, n_alloc, centerp, nx64, h->inv_dxf8, h->dxf8, o8, nr8, r2p, row_base); break;
            case 256: hipLaunchKernelGGL(dense8_build_kernel<4>, grid, blk, 0, 0, h->db, n, h->ld, d, n_alloc, centerp, nx64, h->inv_dxf8, h->dxf8, o8, nr8, r2p, row_base); break;
            default: hipLaunchKernelGGL(dense8_build_kernel<8>, grid, blk, 0, 0, h->db, n, h->ld, d, n_alloc, centerp, nx64, h->inv_dxf8, h->dxf8, o8, nr8, r2p, row_base); break;
        }
        SQ_HIP(hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(nr8 + n_alloc), 0x7f800000, 64));
    }
    SQ_HIP(hipMemset(tmp.p, 0, 64));
    double* sum_r2 = tmp.as<double>() + 1;
    u32* maxb = reinterpret_cast<u32*>(tmp.as<double>() + 2);
    u32* flagged = maxb + 2;
    // the largest |x'|^2 of the new rows BEFORE their always-candidates are marked (a flagged row needs no X)
    hipLaunchKernelGGL(dense8_flag_kernel, dim3((unsigned)((n - row_base + 255) / 256)), dim3(256), 0, 0, r2row.as<float>(), h->nrow8.as<float>(), n,
                       h->cut8, flagged, row_base);
    hipLaunchKernelGGL(dense8_resid_stats_kernel, dim3(256), dim3(256), 0, 0, r2row.as<float>() + row_base, h->nrow8.as<float>() + row_base,
                       n - row_base, sum_r2, maxb);
    struct {
        double energy, sum_r2;
        u32 max_r2, max_n, flagged, pad;
    } host{};
    SQ_HIP(hipMemcpy(&host, tmp.p, sizeof(host), hipMemcpyDeviceToHost));
    SQ_HIP(hipDeviceSynchronize());
    float max_n;
    memcpy(&max_n, &host.max_n, 4);
    if (!cosine && max_n > 0.f) h->xmax8 = std::max(h->xmax8, sqrt((double)max_n) * (1.0 + 1e-6));
    // (the redone unit's old always-candidates are counted again: an over-estimate on the safe side)
    h->flagged8 += host.flagged;
    h->n_pad64 = n_pad64;
    h->n_alloc8 = n_alloc;
    const double budget = std::max(256.0, 2e-5 * (double)n);
    if ((double)h->flagged8 > 4.0 * budget) return done(dense8_build(h));   // the new rows do not look like the old ones: choose again
    return done(SQ_OK);
}

static int dense_grow(DenseHandle* h, long long n_new) {
    const long long n_pad_new = (n_new + TILE_ROWS - 1) / TILE_ROWS * TILE_ROWS;
    const bool cosine = h->metric == SQ_METRIC_COSINE;
    if (h->owned.p || h->n == 0) {
        SQ_TRY(grow_keep(h->owned, (size_t)h->n * h->ld * 4, (size_t)n_new * h->ld * 4));
        h->db = h->owned.as<float>();
    }
    SQ_TRY(grow_keep(h->norms, (size_t)h->n_pad * 4, (size_t)n_pad_new * 4));
    if (!cosine) SQ_TRY(grow_keep(h->norms1, (size_t)h->n_pad * 4, (size_t)n_pad_new * 4));
    if (cosine) SQ_TRY(grow_keep(h->cos_nx, (size_t)h->n * 8, (size_t)n_new * 8));
    if (h->d_pad <= MAX_DPAD) SQ_TRY(grow_keep(h->scan, (size_t)h->n_pad * h->d_pad * 2, (size_t)n_pad_new * h->d_pad * 2));
    return SQ_OK;
}

}  // namespace sq

using namespace sq;

static int dense_create_impl(const float* db, int64_t n, int d, int metric, int mem, int64_t id_base,
                             const std::vector<std::pair<int Options::*, int>>& overrides, sq_handle_t* out);

extern "C" int sq_dense_create(const float* db, int64_t n, int d, int metric, int mem, int64_t id_base,
                               sq_handle_t* out) {
    return dense_create_impl(db, n, d, metric, mem, id_base, {}, out);
}

extern "C" int sq_dense_create_opts(const float* db, int64_t n, int d, int metric, int mem, int64_t id_base,
                                    const char* const* opt_names, const int64_t* opt_values, int n_opts, sq_handle_t* out) {
    if (n_opts < 0 || (n_opts > 0 && (!opt_names || !opt_values))) return fail(SQ_ERR_INVALID, "sq_dense_create_opts: bad option arrays");
    std::vector<std::pair<int Options::*, int>> ov;
    for (int i = 0; i < n_opts; ++i) {
        int Options::*f = option_member(opt_names[i]);
        if (!f) return fail(SQ_ERR_INVALID, "sq_dense_create_opts: unknown option '%s'", opt_names[i] ? opt_names[i] : "(null)");
        ov.emplace_back(f, (int)opt_values[i]);
    }
    return dense_create_impl(db, n, d, metric, mem, id_base, ov, out);
}

// What an index keeps resident and what its build cost (bench.py's `resident_bytes` / `index_build_ms`).
extern "C" int sq_dense_info(sq_handle_t hid, int64_t* out, int n_out) {
    auto* h = static_cast<DenseHandle*>(lookup_handle(hid, H_DENSE));
    if (!h) return fail(SQ_ERR_INVALID, "sq_dense_info: unknown handle");
    if (!out || n_out < SQ_DENSE_INFO_FIELDS) return fail(SQ_ERR_INVALID, "sq_dense_info: room for %d values needed", SQ_DENSE_INFO_FIELDS);
    std::lock_guard<std::mutex> l(h->mu);
    out[0] = h->n;
    out[1] = h->d;
    out[2] = (int64_t)h->n * h->ld * 4;                                  // the float32 rows (the re-rank reads them) ...
    out[3] = h->owned.p ? 1 : 0;                                          // ... owned by the library (1) or borrowed from the caller (0)
    out[4] = (int64_t)h->scan.cap;                                        // bfloat16 scan copy
    out[5] = (int64_t)(h->scan8.cap + h->nrow8.cap);                      // int8 scan copy + its row terms
    out[6] = (int64_t)(h->norms.cap + h->norms1.cap + h->cos_nx.cap + h->center.cap + h->zeros.cap);   // row statistics
    out[7] = h->use8 && !h->suspended8 ? 1 : 0;                                             // the int8 first stage is in use
    out[8] = h->build_us;                                                 // sq_dense_create: wall time of the whole build
    out[9] = h->build8_us;                                                // ... of which the int8 copy (statistics, clamp choice, copy)
    return SQ_OK;
}

static int dense_create_impl(const float* db, int64_t n, int d, int metric, int mem, int64_t id_base,
                             const std::vector<std::pair<int Options::*, int>>& overrides, sq_handle_t* out) {
    const auto t_create = std::chrono::steady_clock::now();
    if (!db || !out || n <= 0 || d <= 0) return fail(SQ_ERR_INVALID, "sq_dense_create: bad argument");
    if (metric != SQ_METRIC_L2 && metric != SQ_METRIC_COSINE)
        return fail(SQ_ERR_INVALID, "sq_dense_create: unknown metric %d", metric);
    if (n >= (1ll << 32)) return fail(SQ_ERR_UNSUPPORTED, "sq_dense_create: more than 2^32-1 rows per shard");
    if (mem == SQ_MEM_DEVICE && ((reinterpret_cast<uintptr_t>(db) & 15u) != 0 || (d & 3) != 0))
        return fail(SQ_ERR_UNSUPPORTED,
                    "sq_dense_create: a borrowed device matrix must be 16-byte aligned with d %% 4 == 0 (d=%d)", d);
    const int d_pad = (d + KT - 1) / KT * KT;
    auto* h = new DenseHandle();
    h->kind = H_DENSE;
    h->n = n;
    h->n_pad = (n + TILE_ROWS - 1) / TILE_ROWS * TILE_ROWS;
    h->d = d;
    h->d_pad = d_pad;
    h->metric = metric;
    h->id_base = id_base;
    h->overrides = overrides;   // (create-time choices -- "dense_int8" = 0: no int8 copy -- are the handle's own)
    h->refresh_options();
    if (hipGetDevice(&h->device) != hipSuccess) {
        delete h;
        return fail(SQ_ERR_HIP, "sq_dense_create: no HIP device");
    }
    auto bail = [&](int rc) {
        delete h;
        return rc;
    };
    if (mem == SQ_MEM_DEVICE) {
        h->db = db;
        h->ld = d;
    } else {
        // owned float32 copy, rows padded to a multiple of 4 floats so the exact kernel can use 16-byte loads
        const int ldo = (d + 3) / 4 * 4;
        const size_t bytes = (size_t)n * ldo * 4;
        int rc = h->owned.reserve(bytes);
        if (rc != SQ_OK) return bail(rc);
        hipError_t e;
        if (ldo == d) {
            e = hipMemcpy(h->owned.p, db, bytes, hipMemcpyHostToDevice);
        } else {
            e = hipMemset(h->owned.p, 0, bytes);
            if (e == hipSuccess)
                e = hipMemcpy2D(h->owned.p, (size_t)ldo * 4, db, (size_t)d * 4, (size_t)d * 4, (size_t)n,
                                hipMemcpyHostToDevice);
        }
        if (e != hipSuccess) return bail(fail(SQ_ERR_HIP, "sq_dense_create: H2D copy failed: %s", hipGetErrorString(e)));
        h->db = h->owned.as<float>();
        h->ld = ldo;
    }
    // the filter's origin (L2): the column means (float64 sums over row blocks); rows appended later keep it
    if (metric == SQ_METRIC_L2 && d_pad <= MAX_DPAD && !h->opt.dense_no_center) {
        int rc = h->center.reserve((size_t)d_pad * 4);
        if (rc != SQ_OK) return bail(rc);
        DevBuf colsum;
        rc = colsum.reserve((size_t)d * 8);
        if (rc != SQ_OK) return bail(rc);
        if (hipMemset(colsum.p, 0, (size_t)d * 8) != hipSuccess) {
            colsum.release();
            return bail(fail(SQ_ERR_HIP, "memset failed"));
        }
        const long long rpb = 512;
        hipLaunchKernelGGL(dense_colsum_kernel, dim3((unsigned)((n + rpb - 1) / rpb)), dim3(256), 0, 0, h->db,
                           (long long)n, h->ld, d, rpb, colsum.as<double>());
        hipLaunchKernelGGL(dense_center_kernel, dim3((d_pad + 255) / 256), dim3(256), 0, 0, colsum.as<double>(),
                           (long long)n, d, d_pad, h->center.as<float>());
        if (hipDeviceSynchronize() != hipSuccess) {
            colsum.release();
            return bail(fail(SQ_ERR_HIP, "sq_dense_create: column means failed"));
        }
        colsum.release();
    }
    // row statistics, then the bfloat16 scan copy (skipped for rows wider than the scan kernel covers)
    {
        int rc = h->norms.reserve((size_t)h->n_pad * 4);
        if (rc == SQ_OK && metric == SQ_METRIC_L2) rc = h->norms1.reserve((size_t)h->n_pad * 4);
        if (rc == SQ_OK && metric == SQ_METRIC_COSINE) rc = h->cos_nx.reserve((size_t)n * 8);
        if (rc == SQ_OK && metric == SQ_METRIC_COSINE) {
            rc = h->zeros.reserve(256);
            if (rc == SQ_OK && hipMemset(h->zeros.p, 0, 256) != hipSuccess) rc = fail(SQ_ERR_HIP, "sq_dense_create: memset failed");
        }
        if (rc == SQ_OK && d_pad <= MAX_DPAD) rc = h->scan.reserve((size_t)h->n_pad * d_pad * 2);
        if (rc == SQ_OK) rc = dense_build_rows(h, 0);
        if (rc == SQ_OK && hipDeviceSynchronize() != hipSuccess) rc = fail(SQ_ERR_HIP, "sq_dense_create: build failed");
        const auto t8 = std::chrono::steady_clock::now();
        if (rc == SQ_OK) rc = dense8_build(h);
        if (rc == SQ_OK && hipDeviceSynchronize() != hipSuccess) rc = fail(SQ_ERR_HIP, "sq_dense_create: int8 build failed");
        if (rc != SQ_OK) return bail(rc);
        h->build8_us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t8).count();
    }
    h->build_us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_create).count();
    *out = register_handle(h);
    return SQ_OK;
}

extern "C" int sq_dense_append(sq_handle_t hid, const float* rows, int64_t n_add, int mem) {
    auto* h = static_cast<DenseHandle*>(lookup_handle(hid, H_DENSE));
    if (!h) return fail(SQ_ERR_INVALID, "sq_dense_append: unknown handle");
    if (!rows || n_add <= 0) return fail(SQ_ERR_INVALID, "sq_dense_append: bad argument");
    std::lock_guard<std::mutex> l(h->mu);
    h->refresh_options();
    if (!h->owned.p) return fail(SQ_ERR_UNSUPPORTED, "sq_dense_append: the index borrows the caller's device matrix");
    SQ_HIP(hipSetDevice(h->device));
    SQ_TRY(dense_sync_all(h));
    const long long n_old = h->n, n_new = n_old + n_add;
    if (n_new >= (1ll << 32)) return fail(SQ_ERR_UNSUPPORTED, "sq_dense_append: more than 2^32-1 rows per shard");
    SQ_HIP(hipSetDevice(h->device));
    SQ_TRY(dense_grow(h, n_new));
    float* dst = h->owned.as<float>() + n_old * h->ld;
    const hipMemcpyKind kind = mem == SQ_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    hipError_t e;
    if (h->ld == h->d) {
        e = hipMemcpy(dst, rows, (size_t)n_add * h->d * 4, kind);
    } else {
        e = hipMemset(dst, 0, (size_t)n_add * h->ld * 4);
        if (e == hipSuccess) e = hipMemcpy2D(dst, (size_t)h->ld * 4, rows, (size_t)h->d * 4, (size_t)h->d * 4, (size_t)n_add, kind);
    }
    if (e != hipSuccess) return fail(SQ_ERR_HIP, "sq_dense_append: copy failed: %s", hipGetErrorString(e));
    h->n = n_new;
    h->n_pad = (n_new + TILE_ROWS - 1) / TILE_ROWS * TILE_ROWS;
    // the tile the old rows ended in is rebuilt together with the new ones (its padding rows become real rows)
    SQ_TRY(dense_build_rows(h, n_old / TILE_ROWS * TILE_ROWS));
    return dense8_append(h, n_old);
}

extern "C" int sq_dense_search(sq_handle_t hid, const float* queries, int nq, int k, void* out_dist, int64_t* out_idx,
                               int mem, void* stream) {
    g_hostprof.start();
    auto* h = static_cast<DenseHandle*>(lookup_handle(hid, H_DENSE));
    if (!h) return fail(SQ_ERR_INVALID, "sq_dense_search: unknown handle");
    if (!queries || !out_dist || !out_idx || nq <= 0 || k <= 0) return fail(SQ_ERR_INVALID, "sq_dense_search: bad argument");
    std::lock_guard<std::mutex> lock(h->mu);
    h->refresh_options();
    SQ_HIP(hipSetDevice(h->device));
    g_hostprof.lap(0);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t dsz = h->metric == SQ_METRIC_COSINE ? 8 : 4;
    if (mem == SQ_MEM_DEVICE_ASYNC && nq <= kDenseQueryChunk) {
        // Call i is enqueued before call i - 1 is waited for: the device always has the next call's kernels
        // queued behind the running ones, and the host reads a call's status words (and redoes what the filter
        // could not certify) one call later.  With "dense_async_streams" = 2 the two slots run on streams of
        // their own, ordered behind the caller's stream by an event, so the small kernels at the end of call
        // i - 1 (re-rank, select) and at the start of call i (query prep, sample pass, threshold) overlap.
        {
            int want = h->opt.dense_async_depth;
            want = want < 2 ? 2 : want > DenseHandle::kMaxDepth ? DenseHandle::kMaxDepth : want;
            if (want != h->depth) {  // a new depth starts from an empty pipeline
                SQ_TRY(dense_sync_all(h));
                h->depth = want;
                h->async_calls = 0;
            }
        }
        DenseSlot& s = h->slot[h->async_calls % (unsigned)h->depth];
        SQ_TRY(dense_resolve(h, s));  // (the call `depth` back; normally resolved during an earlier call)
        g_hostprof.lap(1);
        hipStream_t run = st;
        if (h->opt.dense_async_streams == 2) {
            // One query tile per wave (HBM bound): the two slots alternate between two streams, so neighbouring
            // calls overlap.  Larger batches (MFMA bound, and kept cheap in HBM traffic by all workgroups of an XCD
            // walking the same rows) must not run two scans at once -- two scans at different rows evict each
            // other's rows from L2 (256 queries: 1.17 -> 1.68 ms per call): they share slot 0's stream.
            // (a batch that is ONE group of query tiles -- up to 64 queries at two tiles per wave, 128 at four -- still
            // reads the matrix once: it overlaps like a one-tile batch)
            const int tiles = (nq + TILE_ROWS - 1) / TILE_ROWS;
            const bool one_group = tiles <= scan_query_tiles(h->opt, h->d_pad, tiles);
            DenseSlot& owner = one_group ? s : h->slot[0];
            if (!owner.own) SQ_HIP(hipStreamCreateWithFlags(&owner.own, hipStreamNonBlocking));
            if (h->opt.dense_async_order) {
                if (!s.ev_in) SQ_HIP(hipEventCreateWithFlags(&s.ev_in, hipEventDisableTiming));
                SQ_HIP(hipEventRecord(s.ev_in, st));    // the caller's earlier work on `stream` (the queries) comes first
                SQ_HIP(hipStreamWaitEvent(owner.own, s.ev_in, 0));
            }  // else: the caller vouches for its queries ("dense_async_order" = 0: an event on a busy or foreign stream
               // costs ~10 us of start latency per call, as much as a small shard's whole sample pass)
            run = owner.own;
        }
        g_hostprof.lap(2);
        SQ_TRY(dense_enqueue(h, s, queries, nq, k, out_dist, reinterpret_cast<long long*>(out_idx), run, true));
        g_hostprof.lap(3);
        g_hostprof.ns[5]++;
        h->async_calls++;
        // the oldest call still in flight (depth - 1 calls back): its results are final on return -- unless the caller
        // asked not to wait here ("dense_async_wait" = 0): that call is then finished at the start of the next call (its
        // slot is the next one to be reused), and whatever the host does between the two calls overlaps the device
        if (!h->opt.dense_async_wait) return SQ_OK;
        const int rc_wait = dense_resolve(h, h->slot[h->async_calls % (unsigned)h->depth]);
        g_hostprof.lap(4);
        return rc_wait;
    }
    SQ_TRY(dense_sync_all(h));
    if (mem != SQ_MEM_HOST)
        return dense_search_chunked(h, queries, nq, k, out_dist, reinterpret_cast<long long*>(out_idx), st);
    const size_t qb = (size_t)nq * h->d * 4;
    SQ_TRY(h->q_dev.reserve(qb));
    SQ_TRY(h->out_dist_dev.reserve((size_t)nq * k * dsz));
    SQ_TRY(h->out_idx_dev.reserve((size_t)nq * k * 8));
    SQ_TRY(h->stage.begin(qb + (size_t)nq * k * (dsz + 8)));
    SQ_HIP(h->stage.in(h->q_dev.p, queries, qb, st));
    SQ_TRY(dense_search_chunked(h, h->q_dev.as<float>(), nq, k, h->out_dist_dev.p, h->out_idx_dev.as<long long>(), st));
    SQ_HIP(h->stage.out(out_dist, h->out_dist_dev.p, (size_t)nq * k * dsz, st));
    SQ_HIP(h->stage.out(out_idx, h->out_idx_dev.p, (size_t)nq * k * 8, st));
    SQ_HIP(stream_wait(st));
    h->stage.finish();
    return SQ_OK;
}

extern "C" int sq_dense_sync(sq_handle_t hid) {
    auto* h = static_cast<DenseHandle*>(lookup_handle(hid, H_DENSE));
    if (!h) return fail(SQ_ERR_INVALID, "sq_dense_sync: unknown handle");
    std::lock_guard<std::mutex> lock(h->mu);
    h->refresh_options();
    SQ_HIP(hipSetDevice(h->device));
    if (g_hostprof.on && g_hostprof.ns[5] > 0) {
        const double c = (double)g_hostprof.ns[5] * 1e3;
        fprintf(stderr, "[smqtk_hip] host us per async call over %lld calls: entry %.2f | resolve of the slot %.2f | stream setup %.2f | enqueue %.2f | wait %.2f\n",
                g_hostprof.ns[5], g_hostprof.ns[0] / c, g_hostprof.ns[1] / c, g_hostprof.ns[2] / c, g_hostprof.ns[3] / c, g_hostprof.ns[4] / c);
        for (auto& v : g_hostprof.ns) v = 0;
    }
    return dense_sync_all(h);
}

extern "C" int sq_dense_destroy(sq_handle_t hid) {
    auto* hb = remove_handle(hid, H_DENSE);
    if (!hb) return fail(SQ_ERR_INVALID, "sq_dense_destroy: unknown handle");
    auto* h = static_cast<DenseHandle*>(hb);
    (void)hipSetDevice(h->device);
    {
        std::lock_guard<std::mutex> lock(h->mu);
        h->refresh_options();
        (void)dense_sync_all(h);  // nothing of this handle is left on the device when its buffers go
    }
    delete h;
    return SQ_OK;
}

template <class T>
static void distances_launch(const void* rows, long long n, int d, const void* q, int metric, void* out,
                             hipStream_t st) {
    const unsigned gx = (unsigned)((n + 31) / 32);
    hipLaunchKernelGGL((dense_distances_kernel<T>), dim3(gx), dim3(256), 0, st, (const T*)rows, n, d, (const T*)q,
                       metric, (T*)out, (double*)out);
}

extern "C" int sq_dense_distances(const void* query, const void* rows, int dtype, int64_t n, int d, int metric,
                                  void* out, int mem, void* stream) {
    if (!query || !rows || !out || n <= 0 || d <= 0) return fail(SQ_ERR_INVALID, "sq_dense_distances: bad argument");
    if (metric != SQ_METRIC_L2 && metric != SQ_METRIC_COSINE) return fail(SQ_ERR_INVALID, "sq_dense_distances: unknown metric");
    if (dtype != SQ_DTYPE_F32 && dtype != SQ_DTYPE_F64) return fail(SQ_ERR_INVALID, "sq_dense_distances: unknown dtype");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t esz = dtype == SQ_DTYPE_F32 ? 4 : 8;
    const size_t osz = (metric == SQ_METRIC_COSINE || dtype == SQ_DTYPE_F64) ? 8 : 4;
    if (mem == SQ_MEM_DEVICE) {
        if (dtype == SQ_DTYPE_F32)
            distances_launch<float>(rows, n, d, query, metric, out, st);
        else
            distances_launch<double>(rows, n, d, query, metric, out, st);
        SQ_HIP(hipGetLastError());
        return SQ_OK;
    }
    DevBuf dq, dr, dout;
    int rc = SQ_OK;
    auto done = [&](int code) {
        dq.release();
        dr.release();
        dout.release();
        return code;
    };
    if ((rc = dq.reserve((size_t)d * esz)) != SQ_OK) return done(rc);
    if ((rc = dr.reserve((size_t)n * d * esz)) != SQ_OK) return done(rc);
    if ((rc = dout.reserve((size_t)n * osz)) != SQ_OK) return done(rc);
    if (hipMemcpyAsync(dq.p, query, (size_t)d * esz, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(dr.p, rows, (size_t)n * d * esz, hipMemcpyHostToDevice, st) != hipSuccess)
        return done(fail(SQ_ERR_HIP, "sq_dense_distances: H2D copy failed"));
    if (dtype == SQ_DTYPE_F32)
        distances_launch<float>(dr.p, n, d, dq.p, metric, dout.p, st);
    else
        distances_launch<double>(dr.p, n, d, dq.p, metric, dout.p, st);
    if (hipMemcpyAsync(out, dout.p, (size_t)n * osz, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return done(fail(SQ_ERR_HIP, "sq_dense_distances: kernel or D2H copy failed: %s",
                         hipGetErrorString(hipGetLastError())));
    return done(SQ_OK);
}
