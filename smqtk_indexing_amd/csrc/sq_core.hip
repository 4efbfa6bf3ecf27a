// Handle registry, options, error text and the host-side shard merge.
#include <algorithm>
#include <cmath>
#include <limits>
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
template <class D>
static void merge_impl(const D* dist, const int64_t* idx, int nshards, int nq, int k_in, int k_out, D* out_dist,
                       int64_t* out_idx, D pad_value) {
    std::vector<std::pair<D, int64_t>> buf;
    buf.reserve((size_t)nshards * k_in);
    for (int q = 0; q < nq; ++q) {
        buf.clear();
        for (int s = 0; s < nshards; ++s) {
            const size_t base = ((size_t)s * nq + q) * k_in;
            for (int j = 0; j < k_in; ++j)
                if (idx[base + j] >= 0) buf.emplace_back(dist[base + j], idx[base + j]);
        }
        const size_t take = std::min<size_t>(buf.size(), (size_t)k_out);
        std::partial_sort(buf.begin(), buf.begin() + take, buf.end());
        for (int j = 0; j < k_out; ++j) {
            if ((size_t)j < take) {
                out_dist[(size_t)q * k_out + j] = buf[j].first;
                out_idx[(size_t)q * k_out + j] = buf[j].second;
            } else {
                out_dist[(size_t)q * k_out + j] = pad_value;
                out_idx[(size_t)q * k_out + j] = -1;
            }
        }
    }
}

extern "C" int sq_merge_topk(const void* dist, const int64_t* idx, int dist_dtype, int nshards, int nq, int k_in,
                             int k_out, void* out_dist, int64_t* out_idx) {
    if (!dist || !idx || !out_dist || !out_idx || nshards <= 0 || nq <= 0 || k_in <= 0 || k_out <= 0)
        return fail(SQ_ERR_INVALID, "sq_merge_topk: bad argument");
    switch (dist_dtype) {
        case 0:
            merge_impl<float>((const float*)dist, idx, nshards, nq, k_in, k_out, (float*)out_dist, out_idx,
                              std::numeric_limits<float>::infinity());
            return SQ_OK;
        case 1:
            merge_impl<double>((const double*)dist, idx, nshards, nq, k_in, k_out, (double*)out_dist, out_idx,
                               std::numeric_limits<double>::infinity());
            return SQ_OK;
        case 2:
            merge_impl<int32_t>((const int32_t*)dist, idx, nshards, nq, k_in, k_out, (int32_t*)out_dist, out_idx,
                                std::numeric_limits<int32_t>::max());
            return SQ_OK;
        default:
            return fail(SQ_ERR_INVALID, "sq_merge_topk: unknown dist_dtype %d", dist_dtype);
    }
}
