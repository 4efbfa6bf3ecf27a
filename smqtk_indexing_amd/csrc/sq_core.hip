// Handle registry, options, error text and the host-side shard merge.
#include <algorithm>
#include <cmath>
#include <limits>
#include <thread>
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

extern "C" int sq_set_option(const char* name, int64_t value) {
    if (!name) return fail(SQ_ERR_INVALID, "sq_set_option: null name");
    const std::string n(name);
    if (n == "profile") g_opt.profile = (int)value;
    else if (n == "sample_stride") g_opt.sample_stride = (int)value;
    else if (n == "candidate_cap") g_opt.candidate_cap = (int)value;
    else if (n == "force_fallback") g_opt.force_fallback = (int)value;
    else if (n == "dense_stages") g_opt.dense_stages = (int)value;
    else if (n == "dense_blocks") g_opt.dense_blocks = (int)value;
    else if (n == "dense_debug") g_opt.dense_debug = (int)value;
    else if (n == "dense_waves") g_opt.dense_waves = (int)value;
    else if (n == "dense_qt") g_opt.dense_qt = (int)value;
    else if (n == "itq_exact") g_opt.itq_exact = (int)value;
    else if (n == "hamming_no_permute") g_opt.hamming_no_permute = (int)value;
    else if (n == "dense_no_center") g_opt.dense_no_center = (int)value;
    else if (n == "dense_qplanes") g_opt.dense_qplanes = (int)value;
    else return fail(SQ_ERR_INVALID, "sq_set_option: unknown option '%s'", name);
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

// Host-side k-way merge of per-shard sorted lists: concatenate the shard rows
// of a query, order by (distance, id), keep k_out.  Shard lists are short
// (k_in <= 16384) so a partial sort per query is ample.
// Each shard list is already sorted by (distance, id) with its padding (id -1) at
// the end, so a query's result is a k-way merge: k_out steps, each picking the
// smallest head among the shards (nshards <= 8 on one node: a linear scan).
// dshard / ishard: distance between consecutive shards' [nq][k_in] blocks, in BYTES
template <class D>
static void merge_range(const D* dist, const int64_t* idx, int nshards, int nq, int k_in, int k_out, D* out_dist,
                        int64_t* out_idx, D pad_value, int q0, int q1, size_t dshard, size_t ishard) {
    constexpr int64_t kDone = std::numeric_limits<int64_t>::max();  // id of an exhausted list (sorts last)
    std::vector<int> head((size_t)nshards);
    std::vector<D> hd((size_t)nshards);
    std::vector<int64_t> hi((size_t)nshards);
    auto load = [&](int s, int q) {  // cache the head of shard s
        const int hpos = head[(size_t)s];
        if (hpos < k_in) {
            const size_t at = (size_t)q * k_in + hpos;
            const D* ds = reinterpret_cast<const D*>(reinterpret_cast<const char*>(dist) + (size_t)s * dshard);
            const int64_t* is = reinterpret_cast<const int64_t*>(reinterpret_cast<const char*>(idx) + (size_t)s * ishard);
            if (is[at] >= 0) {
                hd[(size_t)s] = ds[at];
                hi[(size_t)s] = is[at];
                return;
            }
        }
        hd[(size_t)s] = pad_value;  // exhausted, or the rest of the list is padding
        hi[(size_t)s] = kDone;
    };
    for (int q = q0; q < q1; ++q) {
        for (int s = 0; s < nshards; ++s) {
            head[(size_t)s] = 0;
            load(s, q);
        }
        for (int j = 0; j < k_out; ++j) {
            int best = 0;
            for (int s = 1; s < nshards; ++s)
                if (hd[(size_t)s] < hd[(size_t)best] || (hd[(size_t)s] == hd[(size_t)best] && hi[(size_t)s] < hi[(size_t)best]))
                    best = s;
            if (hi[(size_t)best] != kDone) {
                out_dist[(size_t)q * k_out + j] = hd[(size_t)best];
                out_idx[(size_t)q * k_out + j] = hi[(size_t)best];
                ++head[(size_t)best];
                load(best, q);
            } else {
                out_dist[(size_t)q * k_out + j] = pad_value;
                out_idx[(size_t)q * k_out + j] = -1;
            }
        }
    }
}

template <class D>
static void merge_impl(const D* dist, const int64_t* idx, int nshards, int nq, int k_in, int k_out, D* out_dist,
                       int64_t* out_idx, D pad_value, size_t dshard, size_t ishard) {
    // a few host threads when the batch is large (the merge is on the timed path of a multi-GPU step)
    unsigned hw = std::thread::hardware_concurrency();
    int nt = (int)std::min<unsigned>(hw ? hw : 1u, 16u);
    if ((long long)nq * k_out * nshards < 200000) nt = 1;
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
