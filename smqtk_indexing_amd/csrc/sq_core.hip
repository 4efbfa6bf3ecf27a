// Handle registry, options, error text and the host-side shard merge.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <utility>
#include <vector>

#include "sq_common.hpp"

namespace sq {

thread_local char g_err[512] = "";
Options g_opt;

static std::mutex g_reg_mu;
static std::unordered_map<sq_handle_t, HandleBase*> g_reg;
static sq_handle_t g_next = 1;

sq_handle_t register_handle(HandleBase* h) {
    std::lock_guard<std::mutex> l(g_reg_mu);
    sq_handle_t id = g_next++;
    g_reg[id] = h;
    return id;
}
HandleBase* lookup_handle(sq_handle_t id, int kind) {
    std::lock_guard<std::mutex> l(g_reg_mu);
    auto it = g_reg.find(id);
    if (it == g_reg.end() || it->second->kind != kind) return nullptr;
    return it->second;
}
HandleBase* remove_handle(sq_handle_t id, int kind) {
    std::lock_guard<std::mutex> l(g_reg_mu);
    auto it = g_reg.find(id);
    if (it == g_reg.end() || it->second->kind != kind) return nullptr;
    HandleBase* h = it->second;
    g_reg.erase(it);
    return h;
}

}  // namespace sq

using namespace sq;

extern "C" const char* sq_last_error(void) { return g_err; }
extern "C" int sq_version(void) { return 100; }

extern "C" int sq_device_count(int* out_n) {
    if (!out_n) return fail(SQ_ERR_INVALID, "sq_device_count: null argument");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *out_n = 0;
        return fail(SQ_ERR_HIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    }
    *out_n = n;
    return SQ_OK;
}

extern "C" int sq_device_name(int device, char* out_name, int name_len, int64_t* out_total_mem, int* out_cu_count) {
    hipDeviceProp_t p;
    SQ_HIP(hipGetDeviceProperties(&p, device));
    if (out_name && name_len > 0) {
        snprintf(out_name, (size_t)name_len, "%s (%s)", p.name, p.gcnArchName);
    }
    if (out_total_mem) *out_total_mem = (int64_t)p.totalGlobalMem;
    if (out_cu_count) *out_cu_count = p.multiProcessorCount;
    return SQ_OK;
}

namespace sq {
int Options::*option_member(const char* name) {
    static const struct {
        const char* name;
        int Options::*field;
    } kFields[] = {
        {"profile", &Options::profile},
        {"sample_stride", &Options::sample_stride},
        {"candidate_cap", &Options::candidate_cap},
        {"force_fallback", &Options::force_fallback},
        {"dense_stages", &Options::dense_stages},
        {"dense_blocks", &Options::dense_blocks},
        {"dense_sample_blocks", &Options::dense_sample_blocks},
        {"dense_debug", &Options::dense_debug},
        {"dense_waves", &Options::dense_waves},
        {"dense_qt", &Options::dense_qt},
        {"itq_exact", &Options::itq_exact},
        {"hamming_no_permute", &Options::hamming_no_permute},
        {"dense_no_center", &Options::dense_no_center},
        {"merge_threads", &Options::merge_threads},
        {"spin_wait_us", &Options::spin_wait_us},
        {"dense_rerank_segments", &Options::dense_rerank_segments},
        {"dense_qplanes", &Options::dense_qplanes},
        {"dense_async_streams", &Options::dense_async_streams},
        {"dense_async_depth", &Options::dense_async_depth},
        {"dense_async_wait", &Options::dense_async_wait},
        {"dense_async_order", &Options::dense_async_order},
        {"dense_nt", &Options::dense_nt},
        {"dense_nt_keep_mb", &Options::dense_nt_keep_mb},
        {"dense_mid_tier", &Options::dense_mid_tier},
        {"dense_int8", &Options::dense_int8},
        {"dense_graph", &Options::dense_graph},
        {"dense_fused", &Options::dense_fused},
        {"dense_tighten", &Options::dense_tighten},
        {"dense_int8_batch", &Options::dense_int8_batch},
        {"dense_fused_prep", &Options::dense_fused_prep},
        {"hamming_async_depth", &Options::hamming_async_depth},
        {"hamming_async_wait", &Options::hamming_async_wait},
        {"hamming_async_order", &Options::hamming_async_order},
        {"hamming_ring", &Options::hamming_ring},
        {"hamming_fused", &Options::hamming_fused},
        {"hamming_tighten", &Options::hamming_tighten},
    };
    if (!name) return nullptr;
    for (const auto& f : kFields)
        if (strcmp(f.name, name) == 0) return f.field;
    return nullptr;
}
}  // namespace sq

extern "C" int sq_set_option(const char* name, int64_t value) {
    if (!name) return fail(SQ_ERR_INVALID, "sq_set_option: null name");
    int Options::*f = option_member(name);
    if (!f) return fail(SQ_ERR_INVALID, "sq_set_option: unknown option '%s'", name);
    g_opt.*f = (int)value;
    return SQ_OK;
}

static HandleBase* any_handle(sq_handle_t hid) {
    for (int kind : {H_DENSE, H_HAMMING, H_ROWS, H_FIT, H_ITQ})
        if (HandleBase* h = lookup_handle(hid, kind)) return h;
    return nullptr;
}

extern "C" int sq_handle_set_option(sq_handle_t hid, const char* name, int64_t value) {
    if (!name) return fail(SQ_ERR_INVALID, "sq_handle_set_option: null name");
    int Options::*f = option_member(name);
    if (!f) return fail(SQ_ERR_INVALID, "sq_handle_set_option: unknown option '%s'", name);
    HandleBase* h = any_handle(hid);
    if (!h) return fail(SQ_ERR_INVALID, "sq_handle_set_option: unknown handle");
    // an override nobody would read is refused rather than silently kept: the row-matrix and fit handles read no option,
    // an ITQ model reads "itq_exact" / "dense_debug" only, and the waits and the host merge are process-wide by nature
    if (h->kind == H_ROWS || h->kind == H_FIT)
        return fail(SQ_ERR_UNSUPPORTED, "sq_handle_set_option: handles of this kind read no option");
    if (h->kind == H_ITQ && f != &Options::itq_exact && f != &Options::dense_debug)
        return fail(SQ_ERR_UNSUPPORTED, "sq_handle_set_option: an ITQ model reads 'itq_exact' and 'dense_debug' only, not '%s'", name);
    if (f == &Options::spin_wait_us || f == &Options::merge_threads)
        return fail(SQ_ERR_UNSUPPORTED, "sq_handle_set_option: '%s' is process-wide (sq_set_option)", name);
    std::lock_guard<std::mutex> l(h->mu);
    for (auto& o : h->overrides)
        if (o.first == f) {
            o.second = (int)value;
            return SQ_OK;
        }
    h->overrides.emplace_back(f, (int)value);
    return SQ_OK;
}

extern "C" int sq_handle_reset_options(sq_handle_t hid) {
    HandleBase* h = any_handle(hid);
    if (!h) return fail(SQ_ERR_INVALID, "sq_handle_reset_options: unknown handle");
    std::lock_guard<std::mutex> l(h->mu);
    h->overrides.clear();
    return SQ_OK;
}

extern "C" int sq_get_stats(sq_handle_t hid, sq_stats_t* out) {
    if (!out) return fail(SQ_ERR_INVALID, "sq_get_stats: null argument");
    HandleBase* h = lookup_handle(hid, H_DENSE);
    if (!h) h = lookup_handle(hid, H_HAMMING);
    if (!h) return fail(SQ_ERR_INVALID, "sq_get_stats: unknown handle");
    std::lock_guard<std::mutex> l(h->mu);
    *out = h->stats;
    return SQ_OK;
}

// Host-side k-way merge of per-shard sorted lists (on the timed path of a multi-GPU step).
// Each shard list is already sorted by (distance, id) with its padding (id -1) at the end, so a
// query's result is a k-way merge.  Branches on the data are what a scalar merge pays for (a
// mispredict per comparison), so the inner loop has none: distances become order-preserving
// unsigned keys, the smallest head is found with a compare/select tree over <= 8 shards at a time,
// and ties are left in (distance, shard, position) order.  Afterwards every run of equal distance
// -- extended past k_out to the whole tie group of the last distance -- is put in id order (a
// no-op test when shards hold increasing id ranges) and the list is cut to k_out.
// dshard / ishard: distance between consecutive shards' [nq][k_in] blocks, in BYTES
// (every NaN, whatever its sign or payload, gets ONE key just below an exhausted list's: NaN distances rank after
// all numbers, as on a single GPU -- x86 0/0 and numpy produce the sign-bit-set quiet NaN)
static constexpr uint64_t kMergeNaN = ~0ull - 1;
static inline uint64_t merge_key(float v) {
    if (v != v) return kMergeNaN;
    if (v == 0.0f) v = 0.0f;  // -0 and +0 are one distance
    uint32_t b;
    std::memcpy(&b, &v, 4);
    return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}
static inline uint64_t merge_key(double v) {
    if (v != v) return kMergeNaN;
    if (v == 0.0) v = 0.0;
    uint64_t b;
    std::memcpy(&b, &v, 8);
    return b ^ ((b >> 63) ? ~0ull : 0x8000000000000000ull);
}
static inline uint64_t merge_key(int32_t v) { return (uint32_t)v ^ 0x80000000u; }

template <class D>
static void merge_range(const D* dist, const int64_t* idx, int nshards, int nq, int k_in, int k_out, D* out_dist,
                        int64_t* out_idx, D pad_value, int q0, int q1, size_t dshard, size_t ishard) {
    constexpr uint64_t kDone = ~0ull;  // key of an exhausted list: above every distance (NaN included)
    constexpr int kMaxWay = 8;
    (void)nq;
    const int total_cap = nshards * k_in;
    std::vector<D> md((size_t)total_cap), md2;
    std::vector<int64_t> mi((size_t)total_cap), mi2;
    std::vector<std::pair<int64_t, D>> run;
    std::vector<int> seq_store((size_t)total_cap);
    int* const seq = seq_store.data();  // (position * 8 + shard) of every chosen entry
    bool any_tie = false;                // per query: some merge_group saw two equal keys in a row
    // merge `ways` (<= 8) sorted lists into (od, oi): at least `want` entries, then the rest of the last tie group
    auto merge_group = [&](const D* const* ld, const int64_t* const* li, const int* len, int ways, int want, D* od,
                           int64_t* oi) -> int {
        // Heads live in eight scalars (constant indexing only, so they stay in registers) and the
        // key FOLLOWING each head is kept ready in nxt[]: the loop-carried chain is then the select
        // tree plus one L1 load, and the key conversion of the next element is off that chain.
        uint64_t key[kMaxWay], nxt[kMaxWay];
        int head[kMaxWay];
        auto key_at = [&](int s, int pos) -> uint64_t { return (s < ways && pos < len[s]) ? merge_key(ld[s][pos]) : kDone; };
        for (int s = 0; s < kMaxWay; ++s) {
            head[s] = 0;
            key[s] = key_at(s, 0);
            nxt[s] = key_at(s, 1);
        }
        uint64_t k0 = key[0], k1 = key[1], k2 = key[2], k3 = key[3], k4 = key[4], k5 = key[5], k6 = key[6], k7 = key[7];
        int n = 0;
        uint64_t last = 0;
        bool tie = false;
        for (;;) {
            // compare/select tree carrying (key, shard): the lower shard wins equal keys
            const bool c01 = k1 < k0, c23 = k3 < k2, c45 = k5 < k4, c67 = k7 < k6;
            const uint64_t m01 = c01 ? k1 : k0, m23 = c23 ? k3 : k2, m45 = c45 ? k5 : k4, m67 = c67 ? k7 : k6;
            const int i01 = c01 ? 1 : 0, i23 = c23 ? 3 : 2, i45 = c45 ? 5 : 4, i67 = c67 ? 7 : 6;
            const bool c03 = m23 < m01, c47 = m67 < m45;
            const uint64_t m03 = c03 ? m23 : m01, m47 = c47 ? m67 : m45;
            const int i03 = c03 ? i23 : i01, i47 = c47 ? i67 : i45;
            const bool c07 = m47 < m03;
            const uint64_t kb = c07 ? m47 : m03;
            const int best = c07 ? i47 : i03;
            if (kb == kDone || (n >= want && kb != last)) break;
            tie |= (kb == last) & (n > 0);
            const uint64_t nk = nxt[best];
            const int hp = head[best];
            seq[n] = hp * kMaxWay + best;
            ++n;
            last = kb;
            head[best] = hp + 1;
            nxt[best] = hp + 2 < len[best] ? merge_key(ld[best][hp + 2]) : kDone;
            k0 = best == 0 ? nk : k0;
            k1 = best == 1 ? nk : k1;
            k2 = best == 2 ? nk : k2;
            k3 = best == 3 ? nk : k3;
            k4 = best == 4 ? nk : k4;
            k5 = best == 5 ? nk : k5;
            k6 = best == 6 ? nk : k6;
            k7 = best == 7 ? nk : k7;
        }
        for (int t = 0; t < n; ++t) {  // gather the chosen entries (independent loads)
            const int sh = seq[t] % kMaxWay, at = seq[t] / kMaxWay;
            od[t] = ld[sh][at];
            oi[t] = li[sh][at];
        }
        any_tie |= tie;
        return n;
    };
    // runs of equal distance into id order (nothing to do when the merges above met no equal keys)
    auto order_ties = [&](D* od, int64_t* oi, int n) {
        if (!any_tie) return;
        int i = 0;
        while (i < n) {
            int j = i + 1;
            bool sorted = true;
            while (j < n && merge_key(od[j]) == merge_key(od[i])) {
                sorted &= oi[j - 1] < oi[j];
                ++j;
            }
            if (!sorted) {
                run.clear();
                for (int t = i; t < j; ++t) run.emplace_back(oi[t], od[t]);
                std::sort(run.begin(), run.end(), [](const std::pair<int64_t, D>& x, const std::pair<int64_t, D>& y) {
                    return x.first < y.first;
                });
                for (int t = i; t < j; ++t) {
                    oi[t] = run[(size_t)(t - i)].first;
                    od[t] = run[(size_t)(t - i)].second;
                }
            }
            i = j;
        }
    };
    std::vector<const D*> ld((size_t)nshards);
    std::vector<const int64_t*> li((size_t)nshards);
    std::vector<int> len((size_t)nshards);
    for (int q = q0; q < q1; ++q) {
        for (int s = 0; s < nshards; ++s) {
            const D* ds = reinterpret_cast<const D*>(reinterpret_cast<const char*>(dist) + (size_t)s * dshard) + (size_t)q * k_in;
            const int64_t* is = reinterpret_cast<const int64_t*>(reinterpret_cast<const char*>(idx) + (size_t)s * ishard) + (size_t)q * k_in;
            ld[(size_t)s] = ds;
            li[(size_t)s] = is;
            // valid entries: a shard with fewer than k_in rows pads the tail with id -1
            int n = k_in;
            if (is[k_in - 1] < 0) {
                int lo = 0, hi = k_in - 1;  // first padded position
                while (lo < hi) {
                    const int mid = (lo + hi) / 2;
                    if (is[mid] < 0) hi = mid; else lo = mid + 1;
                }
                n = lo;
            }
            len[(size_t)s] = n;
        }
        any_tie = false;
        int n;
        if (nshards <= kMaxWay) {
            n = merge_group(ld.data(), li.data(), len.data(), nshards, k_out, md.data(), mi.data());
        } else {
            // more than 8 shards: groups of 8 into one buffer, then merge the (id-ordered) group results
            md2.resize((size_t)total_cap);
            mi2.resize((size_t)total_cap);
            std::vector<const D*> gd;
            std::vector<const int64_t*> gi;
            std::vector<int> gl;
            int used = 0;
            for (int s = 0; s < nshards; s += kMaxWay) {
                const int ways = std::min(kMaxWay, nshards - s);
                const int m = merge_group(ld.data() + s, li.data() + s, len.data() + s, ways, k_out, md2.data() + used,
                                          mi2.data() + used);
                order_ties(md2.data() + used, mi2.data() + used, m);
                gd.push_back(md2.data() + used);
                gi.push_back(mi2.data() + used);
                gl.push_back(m);
                used += m;
            }
            while (gd.size() > (size_t)kMaxWay) {  // > 64 shards: fold 8 group results at a time
                std::vector<D> td((size_t)total_cap);
                std::vector<int64_t> ti((size_t)total_cap);
                std::vector<const D*> nd;
                std::vector<const int64_t*> ni;
                std::vector<int> nl;
                int u2 = 0;
                for (size_t g = 0; g < gd.size(); g += kMaxWay) {
                    const int ways = (int)std::min<size_t>(kMaxWay, gd.size() - g);
                    const int m = merge_group(gd.data() + g, gi.data() + g, gl.data() + g, ways, k_out, td.data() + u2,
                                              ti.data() + u2);
                    order_ties(td.data() + u2, ti.data() + u2, m);
                    nd.push_back(md2.data() + u2);
                    ni.push_back(mi2.data() + u2);
                    nl.push_back(m);
                    u2 += m;
                }
                std::copy(td.begin(), td.begin() + u2, md2.begin());
                std::copy(ti.begin(), ti.begin() + u2, mi2.begin());
                gd.swap(nd);
                gi.swap(ni);
                gl.swap(nl);
            }
            n = merge_group(gd.data(), gi.data(), gl.data(), (int)gd.size(), k_out, md.data(), mi.data());
        }
        order_ties(md.data(), mi.data(), n);
        D* qd = out_dist + (size_t)q * k_out;
        int64_t* qi = out_idx + (size_t)q * k_out;
        const int keep = std::min(n, k_out);
        std::copy(md.begin(), md.begin() + keep, qd);
        std::copy(mi.begin(), mi.begin() + keep, qi);
        for (int j = keep; j < k_out; ++j) {
            qd[j] = pad_value;
            qi[j] = -1;
        }
    }
}

template <class D>
static void merge_impl(const D* dist, const int64_t* idx, int nshards, int nq, int k_in, int k_out, D* out_dist,
                       int64_t* out_idx, D pad_value, size_t dshard, size_t ishard) {
    // A few host threads when the merge is long: starting threads costs ~0.1 ms on the GPU boxes' hosts, the merge
    // ~5 ns per output on one core (8 shards x 256 queries x k=100: 0.13 ms alone, 0.11-0.19 ms with 2-8 threads;
    // x 1024 queries: 0.52 ms alone, 0.23 ms with four)
    unsigned hw = std::thread::hardware_concurrency();
    int nt = (int)std::min<unsigned>(hw ? hw : 1u, 4u);
    if ((long long)nq * k_out < 60000) nt = 1;
    if (g_opt.merge_threads > 0) nt = g_opt.merge_threads;
    nt = std::min(nt, nq);
    if (nt <= 1) {
        merge_range<D>(dist, idx, nshards, nq, k_in, k_out, out_dist, out_idx, pad_value, 0, nq, dshard, ishard);
        return;
    }
    std::vector<std::thread> pool;
    pool.reserve((size_t)nt);
    for (int t = 0; t < nt; ++t) {
        const int q0 = (int)((long long)nq * t / nt), q1 = (int)((long long)nq * (t + 1) / nt);
        pool.emplace_back(merge_range<D>, dist, idx, nshards, nq, k_in, k_out, out_dist, out_idx, pad_value, q0, q1, dshard,
                          ishard);
    }
    for (auto& th : pool) th.join();
}

static int merge_dispatch(const void* dist, const int64_t* idx, int dist_dtype, int nshards, int nq, int k_in, int k_out,
                          void* out_dist, int64_t* out_idx, size_t dshard, size_t ishard, const char* who) {
    if (!dist || !idx || !out_dist || !out_idx || nshards <= 0 || nq <= 0 || k_in <= 0 || k_out <= 0)
        return fail(SQ_ERR_INVALID, "%s: bad argument", who);
    switch (dist_dtype) {
        case 0:
            merge_impl<float>((const float*)dist, idx, nshards, nq, k_in, k_out, (float*)out_dist, out_idx,
                              std::numeric_limits<float>::infinity(), dshard, ishard);
            return SQ_OK;
        case 1:
            merge_impl<double>((const double*)dist, idx, nshards, nq, k_in, k_out, (double*)out_dist, out_idx,
                               std::numeric_limits<double>::infinity(), dshard, ishard);
            return SQ_OK;
        case 2:
            merge_impl<int32_t>((const int32_t*)dist, idx, nshards, nq, k_in, k_out, (int32_t*)out_dist, out_idx,
                                std::numeric_limits<int32_t>::max(), dshard, ishard);
            return SQ_OK;
        default:
            return fail(SQ_ERR_INVALID, "%s: unknown dist_dtype %d", who, dist_dtype);
    }
}

extern "C" int sq_merge_topk(const void* dist, const int64_t* idx, int dist_dtype, int nshards, int nq, int k_in,
                             int k_out, void* out_dist, int64_t* out_idx) {
    const size_t esz = dist_dtype == 1 ? 8 : 4;
    return merge_dispatch(dist, idx, dist_dtype, nshards, nq, k_in, k_out, out_dist, out_idx, (size_t)nq * k_in * esz,
                          (size_t)nq * k_in * 8, "sq_merge_topk");
}

extern "C" int sq_merge_topk_strided(const void* dist, const int64_t* idx, int dist_dtype, int nshards, int nq, int k_in,
                                     int k_out, int64_t dist_shard_stride, int64_t idx_shard_stride, void* out_dist,
                                     int64_t* out_idx) {
    if (dist_shard_stride <= 0 || idx_shard_stride <= 0) return fail(SQ_ERR_INVALID, "sq_merge_topk_strided: bad stride");
    return merge_dispatch(dist, idx, dist_dtype, nshards, nq, k_in, k_out, out_dist, out_idx, (size_t)dist_shard_stride,
                          (size_t)idx_shard_stride, "sq_merge_topk_strided");
}
