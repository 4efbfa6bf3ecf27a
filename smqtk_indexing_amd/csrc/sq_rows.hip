// Device-resident descriptor rows for the LSH re-rank stage (gfx950).
//
// Replaces the tail of LSHNearestNeighborIndex._nn
// (smqtk_indexing/impls/nn_index/lsh.py:499-519): after the nearest hash codes
// are expanded to candidate descriptors, the reference fetches every candidate
// vector, calls the distance function per row (lsh.py:511) and stable-sorts
// the candidates by distance (lsh.py:513).  Here the descriptor matrix stays on
// the device; a call gathers each query's candidate rows, computes their
// distances in the reference's arithmetic (the code of sq_dense_distances) and
// returns the k smallest in (distance, position in the candidate list) order,
// which is exactly the order a stable sort over the candidate list produces.
#include <algorithm>
#include <vector>

#include "sq_dense_exact.hpp"

namespace sq {

struct RowsHandle : HandleBase {
    const void* rows = nullptr;  // device [n][d] of float32 / float64
    DevBuf owned;
    int dtype = SQ_DTYPE_F32;
    long long n = 0;
    int d = 0;
    DevBuf q_dev, cand_dev, off_dev, cnt_dev, keys, out_keys, out_dist, out_pos, sort_tmp;
    // bucket map of the LSH index (sq_rows_set_buckets): code id -> rows, CSR; and the scratch of sq_lsh_query
    const long long* csr_off = nullptr;   // device [n_codes + 1]
    const long long* csr_rows = nullptr;  // device [n]
    long long n_codes = 0;
    DevBuf csr_off_owned, csr_rows_owned, codes_dev, ham_dist, ham_idx, pre_dev, out_rows;
    HostPinned totals_host;               // [total candidates, largest list]
    PinnedStage stage;
    ~RowsHandle() override {
        stage.release();
        totals_host.release();
        for (DevBuf* b : {&owned, &q_dev, &cand_dev, &off_dev, &cnt_dev, &keys, &out_keys, &out_dist, &out_pos, &sort_tmp,
                          &csr_off_owned, &csr_rows_owned, &codes_dev, &ham_dist, &ham_idx, &pre_dev, &out_rows})
            b->release();
    }
};

// Candidate p of query q (row cand[off[q] + p]) -> key (ordered distance, p).  Eight lanes per
// candidate; float32 rows with L2 give float32 distances in 64-bit keys, everything else float64
// distances in 128-bit keys.  grid = (ceil(maxc / 32), nq).
template <class T, class K>
__global__ __launch_bounds__(256) void rows_rerank_keys_kernel(const T* __restrict__ rows, long long n, int d,
                                                                const T* __restrict__ queries, int metric,
                                                                const long long* __restrict__ cand,
                                                                const long long* __restrict__ off,
                                                                long long maxc, K* __restrict__ keys,
                                                                u32* __restrict__ cnt) {
    const int qi = blockIdx.y;
    const long long c0 = off[qi], c = off[qi + 1] - c0;
    if (blockIdx.x == 0 && threadIdx.x == 0) cnt[qi] = (u32)c;
    const int j8 = threadIdx.x & 7;
    const long long p = (long long)blockIdx.x * 32 + (threadIdx.x >> 3);
    if ((long long)blockIdx.x * 32 >= c) return;
    const long long pc = p < c ? p : c - 1;  // keep the eight-lane groups converged
    long long row = cand[c0 + pc];
    row = row < 0 ? 0 : (row >= n ? n - 1 : row);
    const T* x = rows + row * d;
    const T* q = queries + (long long)qi * d;
    if (metric == SQ_METRIC_L2) {
        auto term = [x, q](int i) {
            const T t = sub_rn(x[i], q[i]);
            return mul_rn_t(t, t);
        };
        const T s = np_pairwise_sum<T>(term, d, j8);
        if (j8 == 0 && p < c) {
            if constexpr (sizeof(K) == 8)
                keys[(long long)qi * maxc + p] = ((u64)ordered_f32(sqrt_rn_f32((float)s)) << 32) | (u64)(u32)p;
            else
                keys[(long long)qi * maxc + p] = K128{ordered_f64(sqrt((double)s)), (u64)p};
        }
    } else {
        // scipy's order (cosine_row_t): three sums, each an even and an odd chain -- lane 0 of the group runs the
        // even chains, lane 1 the odd ones (both lanes of a candidate are live or idle together)
        const int par = j8 & 1;
        double dot = 0.0, nx = 0.0, nq = 0.0;
        if (j8 < 2) {
            const int m = d - (d & 1);
            for (int i = par; i < m; i += 2) {
                const double xv = (double)x[i], qv = (double)q[i];
                dot = __dadd_rn(dot, __dmul_rn(qv, xv));
                nx = __dadd_rn(nx, __dmul_rn(xv, xv));
                nq = __dadd_rn(nq, __dmul_rn(qv, qv));
            }
        }
        const double dot_o = __shfl_xor(dot, 1), nx_o = __shfl_xor(nx, 1), nq_o = __shfl_xor(nq, 1);
        if (j8 == 0 && p < c) {
            dot = __dadd_rn(dot, dot_o);   // even + odd, as the one-lane form adds them
            nx = __dadd_rn(nx, nx_o);
            nq = __dadd_rn(nq, nq_o);
            if (d & 1) {
                const double xv = (double)x[d - 1], qv = (double)q[d - 1];
                dot = __dadd_rn(dot, __dmul_rn(qv, xv));
                nx = __dadd_rn(nx, __dmul_rn(xv, xv));
                nq = __dadd_rn(nq, __dmul_rn(qv, qv));
            }
            const double dist = cosine_dist_f64(dot, nx, nq);
            if constexpr (sizeof(K) == 16) keys[(long long)qi * maxc + p] = K128{ordered_f64(dist), (u64)p};
        }
    }
}

template <class K>
__global__ void rows_finalize_kernel(const K* __restrict__ sorted, const u32* __restrict__ cnt, int k,
                                     void* __restrict__ out_dist, long long* __restrict__ out_pos) {
    const int q = blockIdx.x;
    const u32 c = cnt[q];
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        const bool ok = (u32)i < c;
        const K key = sorted[(long long)q * k + i];
        if constexpr (sizeof(K) == 8) {
            reinterpret_cast<float*>(out_dist)[(long long)q * k + i] = ok ? unordered_f32((u32)(key >> 32)) : __builtin_inff();
            out_pos[(long long)q * k + i] = ok ? (long long)(key & 0xffffffffull) : -1ll;
        } else {
            reinterpret_cast<double*>(out_dist)[(long long)q * k + i] = ok ? unordered_f64(key.hi) : (double)__builtin_inff();
            out_pos[(long long)q * k + i] = ok ? (long long)key.lo : -1ll;
        }
    }
}

// ------------------------------------------------------------------ bucket expansion on the device
// LSHNearestNeighborIndex._nn, lsh.py:489-501: every near hash code is looked up in the hash -> uuids store and the
// buckets are concatenated in the order of the codes.  Here the store is a CSR map (code id -> rows) resident on
// the device and the nearest code ids arrive straight from the Hamming search (device memory): a query's candidate
// list is built without the ids ever visiting the host.
// (1) per query: bucket sizes of its m code ids (-1 = none), exclusive prefix (pre[q][j]) and total (tot[q]).
static __global__ __launch_bounds__(256) void lsh_bucket_sizes_kernel(const long long* __restrict__ ids, int m,
                                                                       const long long* __restrict__ csr_off, long long n_codes,
                                                                       long long* __restrict__ pre, long long* __restrict__ tot) {
    const int q = blockIdx.x;
    __shared__ long long s_wave[4];
    __shared__ long long s_run;
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int j0 = 0; j0 < m; j0 += 256) {
        const int j = j0 + threadIdx.x;
        long long len = 0;
        if (j < m) {
            const long long id = ids[(long long)q * m + j];
            if (id >= 0 && id < n_codes) len = csr_off[id + 1] - csr_off[id];
        }
        long long inc = len;  // inclusive scan inside the wave
        for (int o = 1; o < 64; o <<= 1) {
            const long long t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        if (lane == 63) s_wave[wv] = inc;
        __syncthreads();
        long long base = s_run;
        for (int w = 0; w < wv; ++w) base += s_wave[w];
        if (j < m) pre[(long long)q * m + j] = base + inc - len;
        __syncthreads();
        if (threadIdx.x == 0) s_run += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) tot[q] = s_run;
}
// (2) offsets of the queries' lists and the two numbers the host needs to size the re-rank: [total, largest list]
static __global__ void lsh_offsets_kernel(const long long* __restrict__ tot, int nq, long long* __restrict__ off,
                                          long long* __restrict__ totals_host) {
    if (blockIdx.x || threadIdx.x) return;
    long long run = 0, mx = 0;
    for (int q = 0; q < nq; ++q) {
        off[q] = run;
        run += tot[q];
        mx = tot[q] > mx ? tot[q] : mx;
    }
    off[nq] = run;
    totals_host[0] = run;
    totals_host[1] = mx;
}
// (3) the candidate lists: bucket j of query q starts at off[q] + pre[q][j]; rows of a bucket in row order
static __global__ __launch_bounds__(256) void lsh_fill_candidates_kernel(const long long* __restrict__ ids, int m,
                                                                          const long long* __restrict__ csr_off,
                                                                          const long long* __restrict__ csr_rows, long long n_codes,
                                                                          const long long* __restrict__ pre,
                                                                          const long long* __restrict__ off,
                                                                          long long* __restrict__ cand) {
    const int q = blockIdx.y;
    const int j = blockIdx.x * 32 + (threadIdx.x >> 3);  // eight lanes per bucket
    if (j >= m) return;
    const long long id = ids[(long long)q * m + j];
    if (id < 0 || id >= n_codes) return;
    const long long b0 = csr_off[id], len = csr_off[id + 1] - b0;
    long long* dst = cand + off[q] + pre[(long long)q * m + j];
    for (long long t = threadIdx.x & 7; t < len; t += 8) dst[t] = csr_rows[b0 + t];
}
// winners: positions in the candidate list -> row numbers
template <class K>
__global__ void lsh_finalize_kernel(const K* __restrict__ sorted, const u32* __restrict__ cnt, int k_sel, int k_out,
                                    const long long* __restrict__ cand, const long long* __restrict__ off,
                                    void* __restrict__ out_dist, long long* __restrict__ out_rows) {
    const int q = blockIdx.x;
    const u32 c = cnt[q];
    for (int i = threadIdx.x; i < k_out; i += blockDim.x) {
        const bool ok = i < k_sel && (u32)i < c;
        long long row = -1;
        if constexpr (sizeof(K) == 8) {
            const K key = ok ? sorted[(long long)q * k_sel + i] : 0ull;
            reinterpret_cast<float*>(out_dist)[(long long)q * k_out + i] = ok ? unordered_f32((u32)(key >> 32)) : __builtin_inff();
            if (ok) row = cand[off[q] + (long long)(key & 0xffffffffull)];
        } else {
            const K key = ok ? sorted[(long long)q * k_sel + i] : K128{0ull, 0ull};
            reinterpret_cast<double*>(out_dist)[(long long)q * k_out + i] = ok ? unordered_f64(key.hi) : (double)__builtin_inff();
            if (ok) row = cand[off[q] + (long long)key.lo];
        }
        out_rows[(long long)q * k_out + i] = row;
    }
}

template <class K>
static int rows_select(const K* keys, const u32* cnt, u32 cap, long long stride, int k, int nq, K* out, hipStream_t st,
                       DevBuf& sort_scratch) {
    static bool attr_set = false;
    const int lds_keys = sizeof(K) == 8 ? 16384 : 7168;
    if (k > lds_keys)  // lsh.py:513-518 slices whatever n is asked: the any-k sorted select (sq_select.hpp)
        return sort_select_large<K, SelectNoPost>(keys, cnt, cap, stride, k, nq, out, sort_scratch, SelectNoPost(), st);
    const size_t lds = (size_t)(lds_keys + SELECT_SORT_MAX) * sizeof(K);
    if (!attr_set) {
        SQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&select_topk_kernel<K>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((select_topk_kernel<K>), dim3(nq), dim3(1024), lds, st, keys, cnt, cap, stride, k, lds_keys, out);
    return SQ_OK;
}

template <class T, class K>
static int rows_rerank_t(RowsHandle* h, int nq, int metric, long long maxc, int k, hipStream_t st) {
    SQ_TRY(h->keys.reserve((size_t)nq * maxc * sizeof(K)));
    SQ_TRY(h->out_keys.reserve((size_t)nq * k * sizeof(K)));
    const unsigned gx = (unsigned)((maxc + 31) / 32);
    hipLaunchKernelGGL((rows_rerank_keys_kernel<T, K>), dim3(gx, nq), dim3(256), 0, st,
                       reinterpret_cast<const T*>(h->rows), h->n, h->d, h->q_dev.as<T>(), metric,
                       h->cand_dev.as<long long>(), h->off_dev.as<long long>(), maxc, h->keys.as<K>(), h->cnt_dev.as<u32>());
    SQ_TRY(rows_select<K>(h->keys.as<K>(), h->cnt_dev.as<u32>(), (u32)maxc, maxc, k, nq, h->out_keys.as<K>(), st, h->sort_tmp));
    hipLaunchKernelGGL((rows_finalize_kernel<K>), dim3(nq), dim3(256), 0, st, h->out_keys.as<K>(), h->cnt_dev.as<u32>(), k,
                       h->out_dist.p, h->out_pos.as<long long>());
    SQ_HIP(hipGetLastError());
    return SQ_OK;
}

}  // namespace sq

using namespace sq;

extern "C" int sq_rows_create(const void* rows, int dtype, int64_t n, int d, int mem, sq_handle_t* out) {
    if (!rows || !out || n <= 0 || d <= 0) return fail(SQ_ERR_INVALID, "sq_rows_create: bad argument");
    if (dtype != SQ_DTYPE_F32 && dtype != SQ_DTYPE_F64) return fail(SQ_ERR_INVALID, "sq_rows_create: unknown dtype %d", dtype);
    auto* h = new RowsHandle();
    h->kind = H_ROWS;
    h->dtype = dtype;
    h->n = n;
    h->d = d;
    if (hipGetDevice(&h->device) != hipSuccess) {
        delete h;
        return fail(SQ_ERR_HIP, "sq_rows_create: no HIP device");
    }
    if (mem == SQ_MEM_DEVICE) {
        h->rows = rows;
    } else {
        const size_t bytes = (size_t)n * d * (dtype == SQ_DTYPE_F32 ? 4 : 8);
        int rc = h->owned.reserve(bytes);
        if (rc != SQ_OK) {
            delete h;
            return rc;
        }
        if (hipMemcpy(h->owned.p, rows, bytes, hipMemcpyHostToDevice) != hipSuccess) {
            delete h;
            return fail(SQ_ERR_HIP, "sq_rows_create: H2D copy failed");
        }
        h->rows = h->owned.p;
    }
    *out = register_handle(h);
    return SQ_OK;
}

extern "C" int sq_rows_append(sq_handle_t hid, const void* rows, int64_t n_add, int mem) {
    auto* h = static_cast<RowsHandle*>(lookup_handle(hid, H_ROWS));
    if (!h) return fail(SQ_ERR_INVALID, "sq_rows_append: unknown handle");
    if (!rows || n_add <= 0) return fail(SQ_ERR_INVALID, "sq_rows_append: bad argument");
    std::lock_guard<std::mutex> lock(h->mu);
    if (!h->owned.p) return fail(SQ_ERR_UNSUPPORTED, "sq_rows_append: the matrix is borrowed from the caller");
    SQ_HIP(hipSetDevice(h->device));
    const size_t esz = h->dtype == SQ_DTYPE_F32 ? 4 : 8;
    const size_t used = (size_t)h->n * h->d * esz, need = (size_t)(h->n + n_add) * h->d * esz;
    if (need > h->owned.cap) {  // grow by half again at least, contents kept
        DevBuf nb;
        SQ_TRY(nb.reserve(std::max(need, used + used / 2)));
        if (hipMemcpy(nb.p, h->owned.p, used, hipMemcpyDeviceToDevice) != hipSuccess) {
            nb.release();
            return fail(SQ_ERR_HIP, "sq_rows_append: device copy failed");
        }
        h->owned.release();
        h->owned = nb;
        h->rows = h->owned.p;
    }
    if (hipMemcpy(static_cast<char*>(h->owned.p) + used, rows, need - used,
                  mem == SQ_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice) != hipSuccess)
        return fail(SQ_ERR_HIP, "sq_rows_append: copy failed");
    h->n += n_add;
    return SQ_OK;
}

extern "C" int sq_rows_rerank(sq_handle_t hid, const void* queries, int nq, int metric, const int64_t* cand_rows,
                              const int64_t* cand_offsets, int k, void* out_dist, int64_t* out_pos, void* stream) {
    auto* h = static_cast<RowsHandle*>(lookup_handle(hid, H_ROWS));
    if (!h) return fail(SQ_ERR_INVALID, "sq_rows_rerank: unknown handle");
    if (!queries || !cand_rows || !cand_offsets || !out_dist || !out_pos || nq <= 0 || k <= 0)
        return fail(SQ_ERR_INVALID, "sq_rows_rerank: bad argument");
    if (metric != SQ_METRIC_L2 && metric != SQ_METRIC_COSINE) return fail(SQ_ERR_INVALID, "sq_rows_rerank: unknown metric");
    const bool f32 = h->dtype == SQ_DTYPE_F32;
    const bool k64 = f32 && metric == SQ_METRIC_L2;  // float32 distances, 64-bit keys
    long long maxc = 0;
    const long long total = cand_offsets[nq];
    for (int q = 0; q < nq; ++q) {
        const long long c = cand_offsets[q + 1] - cand_offsets[q];
        if (c < 0) return fail(SQ_ERR_INVALID, "sq_rows_rerank: offsets must not decrease");
        if (c > maxc) maxc = c;
    }
    if (maxc >= (1ll << 32)) return fail(SQ_ERR_UNSUPPORTED, "sq_rows_rerank: more than 2^32-1 candidates for one query");
    std::lock_guard<std::mutex> lock(h->mu);
    SQ_HIP(hipSetDevice(h->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t esz = f32 ? 4 : 8, dsz = k64 ? 4 : 8;
    if (maxc == 0) {  // nothing to rank: padding only
        for (long long i = 0; i < (long long)nq * k; ++i) {
            if (k64) reinterpret_cast<float*>(out_dist)[i] = __builtin_inff();
            else reinterpret_cast<double*>(out_dist)[i] = (double)__builtin_inff();
            out_pos[i] = -1;
        }
        return SQ_OK;
    }
    SQ_TRY(h->q_dev.reserve((size_t)nq * h->d * esz));
    SQ_TRY(h->cand_dev.reserve((size_t)(total > 0 ? total : 1) * 8));
    SQ_TRY(h->off_dev.reserve((size_t)(nq + 1) * 8));
    SQ_TRY(h->cnt_dev.reserve((size_t)nq * 4));
    SQ_TRY(h->out_dist.reserve((size_t)nq * k * dsz));
    SQ_TRY(h->out_pos.reserve((size_t)nq * k * 8));
    SQ_TRY(h->stage.begin((size_t)nq * h->d * esz + (size_t)total * 8 + (size_t)(nq + 1) * 8 + (size_t)nq * k * (dsz + 8)));
    SQ_HIP(h->stage.in(h->q_dev.p, queries, (size_t)nq * h->d * esz, st));
    SQ_HIP(h->stage.in(h->cand_dev.p, cand_rows, (size_t)total * 8, st));
    SQ_HIP(h->stage.in(h->off_dev.p, cand_offsets, (size_t)(nq + 1) * 8, st));
    int rc;
    if (k64)
        rc = rows_rerank_t<float, u64>(h, nq, metric, maxc, k, st);
    else if (f32)
        rc = rows_rerank_t<float, K128>(h, nq, metric, maxc, k, st);
    else
        rc = rows_rerank_t<double, K128>(h, nq, metric, maxc, k, st);
    if (rc != SQ_OK) return rc;
    SQ_HIP(h->stage.out(out_dist, h->out_dist.p, (size_t)nq * k * dsz, st));
    SQ_HIP(h->stage.out(out_pos, h->out_pos.p, (size_t)nq * k * 8, st));
    SQ_HIP(stream_wait(st));
    h->stage.finish();
    return SQ_OK;
}

extern "C" int sq_rows_set_buckets(sq_handle_t hid, const int64_t* csr_off, int64_t n_codes, const int64_t* csr_rows, int mem) {
    auto* h = static_cast<RowsHandle*>(lookup_handle(hid, H_ROWS));
    if (!h) return fail(SQ_ERR_INVALID, "sq_rows_set_buckets: unknown handle");
    if (!csr_off || !csr_rows || n_codes <= 0) return fail(SQ_ERR_INVALID, "sq_rows_set_buckets: bad argument");
    std::lock_guard<std::mutex> lock(h->mu);
    SQ_HIP(hipSetDevice(h->device));
    if (mem == SQ_MEM_DEVICE) {
        h->csr_off = reinterpret_cast<const long long*>(csr_off);
        h->csr_rows = reinterpret_cast<const long long*>(csr_rows);
    } else {
        // the buckets may leave rows out (descriptors removed from the store keep their slot in the matrix)
        const long long listed = csr_off[n_codes];
        if (csr_off[0] != 0 || listed <= 0 || listed > h->n)
            return fail(SQ_ERR_INVALID, "sq_rows_set_buckets: offsets must start at 0 and end at the number of listed rows (<= the row count)");
        SQ_TRY(h->csr_off_owned.reserve((size_t)(n_codes + 1) * 8));
        SQ_TRY(h->csr_rows_owned.reserve((size_t)listed * 8));
        SQ_HIP(hipMemcpy(h->csr_off_owned.p, csr_off, (size_t)(n_codes + 1) * 8, hipMemcpyHostToDevice));
        SQ_HIP(hipMemcpy(h->csr_rows_owned.p, csr_rows, (size_t)listed * 8, hipMemcpyHostToDevice));
        h->csr_off = h->csr_off_owned.as<long long>();
        h->csr_rows = h->csr_rows_owned.as<long long>();
    }
    h->n_codes = n_codes;
    return SQ_OK;
}

namespace sq {
template <class T, class K>
static int lsh_rerank_t(RowsHandle* h, int nq, int metric, long long maxc, int k_sel, int k_out, void* out_dist_dev,
                        long long* out_rows_dev, hipStream_t st) {
    SQ_TRY(h->keys.reserve((size_t)nq * maxc * sizeof(K)));
    SQ_TRY(h->out_keys.reserve((size_t)nq * k_sel * sizeof(K)));
    const unsigned gx = (unsigned)((maxc + 31) / 32);
    hipLaunchKernelGGL((rows_rerank_keys_kernel<T, K>), dim3(gx, nq), dim3(256), 0, st,
                       reinterpret_cast<const T*>(h->rows), h->n, h->d, h->q_dev.as<T>(), metric,
                       h->cand_dev.as<long long>(), h->off_dev.as<long long>(), maxc, h->keys.as<K>(), h->cnt_dev.as<u32>());
    SQ_TRY(rows_select<K>(h->keys.as<K>(), h->cnt_dev.as<u32>(), (u32)maxc, maxc, k_sel, nq, h->out_keys.as<K>(), st, h->sort_tmp));
    hipLaunchKernelGGL((lsh_finalize_kernel<K>), dim3(nq), dim3(256), 0, st, h->out_keys.as<K>(), h->cnt_dev.as<u32>(), k_sel,
                       k_out, h->cand_dev.as<long long>(), h->off_dev.as<long long>(), out_dist_dev, out_rows_dev);
    SQ_HIP(hipGetLastError());
    return SQ_OK;
}
}  // namespace sq

extern "C" int sq_hamming_info(sq_handle_t h, int64_t* out_n, int* out_words);

extern "C" int sq_lsh_query(sq_handle_t rows_h, sq_handle_t hamming_h, sq_handle_t itq_h, const void* queries, int nq,
                            int n_codes_wanted, int metric, int k_out, void* out_dist, int64_t* out_rows, int mem,
                            void* stream) {
    auto* h = static_cast<RowsHandle*>(lookup_handle(rows_h, H_ROWS));
    if (!h) return fail(SQ_ERR_INVALID, "sq_lsh_query: unknown row matrix handle");
    if (!queries || !out_dist || !out_rows || nq <= 0 || n_codes_wanted <= 0 || k_out <= 0)
        return fail(SQ_ERR_INVALID, "sq_lsh_query: bad argument");
    if (metric != SQ_METRIC_L2 && metric != SQ_METRIC_COSINE) return fail(SQ_ERR_INVALID, "sq_lsh_query: unknown metric");
    std::lock_guard<std::mutex> lock(h->mu);
    if (!h->csr_off) return fail(SQ_ERR_INVALID, "sq_lsh_query: no bucket map (sq_rows_set_buckets)");
    int64_t ham_n = 0;
    int words = 0;
    SQ_TRY(sq_hamming_info(hamming_h, &ham_n, &words));
    if (ham_n != h->n_codes) return fail(SQ_ERR_INVALID, "sq_lsh_query: the hash index holds %lld codes, the bucket map %lld",
                                         (long long)ham_n, h->n_codes);
    SQ_HIP(hipSetDevice(h->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool f32 = h->dtype == SQ_DTYPE_F32;
    const bool k64 = f32 && metric == SQ_METRIC_L2;
    const size_t esz = f32 ? 4 : 8, dsz = k64 ? 4 : 8;
    const int m = (int)std::min<long long>(n_codes_wanted, ham_n);  // nearest codes per query
    const void* q_dev = queries;
    if (mem != SQ_MEM_DEVICE) {
        SQ_TRY(h->q_dev.reserve((size_t)nq * h->d * esz));
        SQ_TRY(h->stage.begin((size_t)nq * h->d * esz + (size_t)nq * k_out * (dsz + 8)));
        SQ_HIP(h->stage.in(h->q_dev.p, queries, (size_t)nq * h->d * esz, st));
        q_dev = h->q_dev.p;
    } else {
        // the re-rank kernels read the queries from q_dev: borrow the caller's buffer for this call
        SQ_TRY(h->q_dev.reserve((size_t)nq * h->d * esz));
        SQ_HIP(hipMemcpyAsync(h->q_dev.p, queries, (size_t)nq * h->d * esz, hipMemcpyDeviceToDevice, st));
        q_dev = h->q_dev.p;
    }
    // hash (lsh.py:473), nearest codes (lsh.py:480-487): device to device
    SQ_TRY(h->codes_dev.reserve((size_t)nq * words * 8));
    SQ_TRY(h->ham_dist.reserve((size_t)nq * m * 4));
    SQ_TRY(h->ham_idx.reserve((size_t)nq * m * 8));
    SQ_TRY(sq_itq_model_hash(itq_h, q_dev, h->dtype, nq, h->codes_dev.as<uint64_t>(), SQ_MEM_DEVICE, st));
    SQ_TRY(sq_hamming_search(hamming_h, h->codes_dev.as<uint64_t>(), nq, m, h->ham_dist.as<int32_t>(), h->ham_idx.as<int64_t>(),
                             SQ_MEM_DEVICE, st));
    // bucket expansion (lsh.py:489-501) on the device; only [total, largest list] comes back
    SQ_TRY(h->pre_dev.reserve((size_t)nq * m * 8 + (size_t)nq * 8));
    SQ_TRY(h->off_dev.reserve((size_t)(nq + 1) * 8));
    SQ_TRY(h->cnt_dev.reserve((size_t)nq * 4));
    SQ_TRY(h->totals_host.reserve(16));
    long long* pre = h->pre_dev.as<long long>();
    long long* tot = pre + (size_t)nq * m;
    long long* th = reinterpret_cast<long long*>(h->totals_host.p);
    long long* th_dev = nullptr;
    SQ_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&th_dev), th, 0));
    hipLaunchKernelGGL(lsh_bucket_sizes_kernel, dim3(nq), dim3(256), 0, st, h->ham_idx.as<long long>(), m, h->csr_off, h->n_codes,
                       pre, tot);
    hipLaunchKernelGGL(lsh_offsets_kernel, dim3(1), dim3(1), 0, st, tot, nq, h->off_dev.as<long long>(), th_dev);
    SQ_HIP(stream_wait(st));
    const long long total = th[0], maxc = th[1];
    if (maxc >= (1ll << 32)) return fail(SQ_ERR_UNSUPPORTED, "sq_lsh_query: more than 2^32-1 candidates for one query");
    void* od = out_dist;
    long long* orow = reinterpret_cast<long long*>(out_rows);
    if (mem != SQ_MEM_DEVICE) {
        SQ_TRY(h->out_dist.reserve((size_t)nq * k_out * dsz));
        SQ_TRY(h->out_rows.reserve((size_t)nq * k_out * 8));
        od = h->out_dist.p;
        orow = h->out_rows.as<long long>();
    }
    SQ_TRY(h->cand_dev.reserve((size_t)(total > 0 ? total : 1) * 8));
    const long long mc = maxc > 0 ? maxc : 1;
    hipLaunchKernelGGL(lsh_fill_candidates_kernel, dim3((unsigned)((m + 31) / 32), nq), dim3(256), 0, st, h->ham_idx.as<long long>(),
                       m, h->csr_off, h->csr_rows, h->n_codes, pre, h->off_dev.as<long long>(), h->cand_dev.as<long long>());
    const int k_sel = (int)std::min<long long>(k_out, mc);
    int rc;
    if (k64)
        rc = lsh_rerank_t<float, u64>(h, nq, metric, mc, k_sel, k_out, od, orow, st);
    else if (f32)
        rc = lsh_rerank_t<float, K128>(h, nq, metric, mc, k_sel, k_out, od, orow, st);
    else
        rc = lsh_rerank_t<double, K128>(h, nq, metric, mc, k_sel, k_out, od, orow, st);
    if (rc != SQ_OK) return rc;
    if (mem != SQ_MEM_DEVICE) {
        SQ_HIP(h->stage.out(out_dist, od, (size_t)nq * k_out * dsz, st));
        SQ_HIP(h->stage.out(out_rows, orow, (size_t)nq * k_out * 8, st));
        SQ_HIP(stream_wait(st));
        h->stage.finish();
    }
    return SQ_OK;
}

extern "C" int sq_rows_destroy(sq_handle_t hid) {
    auto* h = remove_handle(hid, H_ROWS);
    if (!h) return fail(SQ_ERR_INVALID, "sq_rows_destroy: unknown handle");
    (void)hipSetDevice(h->device);
    delete h;
    return SQ_OK;
}
