// A second-level threshold for the bf16 chain's survivors (gfx950).        (included by sq_dense.hip; used by the wide-row path)
//
// The sampled threshold T' = post(t_s) -- t_s the k-th smallest filter score of a SAMPLE of the rows, post() adding the
// filter's slack -- lets ~stride x k rows per query through where far fewer lie below the k-th filter score of ALL rows plus
// the same slack, and every one of them is a row gather for the exact re-rank: 16 KB each at 4096 dimensions, 40 % of a
// step.  The fused int8 call tightens inside its full pass (sq_dense_i8.hpp, "the tightened threshold"); here the same
// argument runs as two small kernels between the pass and the re-rank.  The pass stores each entry's smallest filter
// score m beside the entry; dense_tighten_hist_kernel counts the entries of every query in 64 bins of width w = slack / 32
// below T' (bin j: T' - (j + 1) w < m <= T' - j w, the last bin open below); dense_tighten_thr_kernel takes the largest j
// whose bins j .. 63 hold k entries: k different rows score at most T_8 = T' - (j - 1) w (one bin of margin for the float32
// binning), so T_8 bounds the k-th filter score
// exactly as t_s did, and T'' = min(T', post(T_8)) is a valid threshold.  The re-rank takes only entries with m <= T''
// (rerank_block's filtered mode) and the select certifies against T'' (DenseFinalize*::thr2k).
#pragma once
#include "sq_dense_exact.hpp"

namespace sq {

static constexpr int TG_BINS = 64;
static constexpr int TG_MAX_GROUP = 128;   // queries of a scan workgroup's group (32 x query tiles per wave)
static constexpr int TG_CAP_Q = 4096 + 128;  // padded queries of a call (larger batches run in chunks of 4096)

// hist: [nq_pad][TG_BINS], zero on entry (dense_tighten_thr_kernel wipes it).  Grid (slices, groups): block (x, g) walks
// every gridDim.x-th wave segment of query group g (a segment's queries are its group's: wave_cnt[2 w + 1] = the group's
// first query tile) with an LDS histogram of the group's queries.
static __global__ __launch_bounds__(256) void dense_tighten_hist_kernel(const uint2* __restrict__ wave_out, const float* __restrict__ wave_score,
                                                                        const u32* __restrict__ wave_cnt, u32 wave_cap, long long n_waves,
                                                                        const float* __restrict__ thr, const float* __restrict__ traw,
                                                                        int group_q, u32* __restrict__ hist) {
    __shared__ u32 lh[TG_MAX_GROUP * TG_BINS];
    const u32 g0 = (u32)blockIdx.y * (u32)group_q;   // the group's first query
    for (int i = threadIdx.x; i < group_q * TG_BINS; i += 256) lh[i] = 0u;
    __syncthreads();
    for (long long w = blockIdx.x; w < n_waves; w += gridDim.x) {
        if (wave_cnt[2 * w + 1] * 32u != g0) continue;   // (uniform)
        u32 c = wave_cnt[2 * w];
        if (c > wave_cap) c = wave_cap;
        const uint2* seg = wave_out + w * wave_cap;
        const float* sc = wave_score + w * wave_cap;
        for (u32 e = threadIdx.x; e < c; e += 256) {
            const u32 ql = seg[e].y & 0xffffu;
            const float t1 = thr[g0 + ql], wbin = (t1 - traw[g0 + ql]) * 0.03125f;
            int bin = 0;
            if (wbin > 0.f && wbin < __builtin_inff()) {
                const float f = (t1 - sc[e]) / wbin;   // (m <= T': never negative; NaN -> bin 0)
                bin = f >= (float)(TG_BINS - 1) ? TG_BINS - 1 : (f > 0.f ? (int)f : 0);
            }
            atomicAdd(&lh[ql * TG_BINS + bin], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < group_q * TG_BINS; i += 256)
        if (lh[i]) atomicAdd(&hist[(size_t)g0 * TG_BINS + i], lh[i]);
}

// one thread per query; thr2: the tightened thresholds, thr2k: their ordered keys (DenseFinalize*::thr2k)
template <class Post>
static __global__ void dense_tighten_thr_kernel(u32* __restrict__ hist, const float* __restrict__ thr, const float* __restrict__ traw,
                                                int nq, int nq_pad, int kk, Post post, float* __restrict__ thr2, u32* __restrict__ thr2k) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq_pad) return;
    u32* h = hist + (long long)q * TG_BINS;
    const float t1 = thr[q];
    float t2 = t1;
    if (q < nq) {
        const float wbin = (t1 - traw[q]) * 0.03125f;
        u32 cum = 0;
        int j = 0;
        for (int b = TG_BINS - 1; b >= 1; --b) {
            cum += h[b];
            if (cum >= (u32)kk) {
                j = b;
                break;
            }
        }
        if (j > 1 && wbin > 0.f && wbin < __builtin_inff()) {
            // k entries score at most t8.  The histogram kernel bins with float32 arithmetic: an entry a rounding error above
            // the edge of bin j may sit in bin j, so the bound is taken one bin looser (T' - (j - 1) w), rounded up
            const double t8d = (double)t1 - (double)(j - 1) * (double)wbin;   // (exact in float64)
            float t8 = (float)t8d;
            if ((double)t8 < t8d) t8 = __uint_as_float(__float_as_uint(t8) + (t8 >= 0.f ? 1 : -1));
            const float cand = post(q, t8);
            if (cand < t2) t2 = cand;
        }
    }
    thr2[q] = t2;
    thr2k[q] = ordered_f32(t2);
    for (int b = 0; b < TG_BINS; ++b) h[b] = 0u;
}

// the re-rank kernels with rerank_block's second-level filter (entries whose score is above their query's T'' are skipped)
template <class K, bool COSINE>
static __global__ __launch_bounds__(512) void dense_rerank_filtered_kernel(
    const float* __restrict__ db, long long ld, int d, const float* __restrict__ q_al, int ldq,
    const uint2* __restrict__ wave_out, const u32* __restrict__ wave_cnt, u32 wave_cap, long long n_waves,
    int waves_per_block, int nq, int group_q, K* __restrict__ keys, u32* __restrict__ cnt, u32 cap,
    u32* __restrict__ overflow, const double* __restrict__ nx64, const double* __restrict__ nq64, int debug,
    const float* __restrict__ wave_score, const float* __restrict__ thr2) {
    extern __shared__ __attribute__((aligned(16))) float s_qrows_dyn[];
    __shared__ u32 s_hist[RERANK_MAX_GROUP], s_base[RERANK_MAX_GROUP], s_fill[RERANK_MAX_GROUP];
    __shared__ uint2 s_list[2048];
    __shared__ float s_thr2[RERANK_MAX_GROUP];
    __shared__ u32 s_npass;
    const long long w0 = (long long)blockIdx.x * waves_per_block;
    if (w0 < n_waves) {
        const u32 q0 = wave_cnt[2 * w0 + 1] * 32u;
        if ((int)threadIdx.x < group_q) s_thr2[threadIdx.x] = thr2[q0 + threadIdx.x];
    }
    __syncthreads();
    RerankLds L{s_qrows_dyn, s_hist, s_base, s_fill, nullptr};
    L.scores = wave_score;
    L.thr2 = s_thr2;
    L.list = s_list;
    L.npass = &s_npass;
    L.list_cap = 2048;
    rerank_block<K, COSINE>(db, ld, d, q_al, ldq, nq, group_q, wave_out, wave_cnt, wave_cap, n_waves, waves_per_block, keys, cnt, cap,
                            overflow, nx64, nq64, debug, w0, L);
}

}  // namespace sq
