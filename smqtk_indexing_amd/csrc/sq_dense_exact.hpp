// Exact (reference-arithmetic) distance kernels, threshold slack and result
// finalisation/certification of the dense path.  See sq_dense.hip for the
// overall structure and DESIGN.md section 4.2.
#pragma once
#include "sq_pairwise.hpp"
#include "sq_select.hpp"

namespace sq {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------- reference arithmetic
// sum_{i<d} (x[i]-q[i])^2 exactly as numpy evaluates np.square(i - j).sum()
// (metrics.py:86): float32 subtract, float32 square, pairwise add-reduce.
__device__ __forceinline__ float np_sqdist_f32(const float* __restrict__ x, const float* __restrict__ q, int d, int j8) {
    auto term = [x, q](int i) {
        const float t = __fsub_rn(x[i], q[i]);
        return __fmul_rn(t, t);
    };
    return np_pairwise_sum<float>(term, d, j8);
}

__device__ __forceinline__ float sqrt_rn_f32(float v) {
    // correctly rounded: double sqrt is correctly rounded and 53 >= 2*24+2
    return (float)sqrt((double)v);
}

// cosine_distance(q, x) of metrics.py:120-137 in float64, following the order
// of scipy's C kernel behind cdist(..., 'cosine') (scipy/spatial/src/
// distance_impl.h, scipy 1.15.3 as pinned here): sequential dot products,
// c = u.v / (|u| |v|) clipped to [-1,1], cdist value 1 - c; the reference then
// forms sim = 1 - cdist, clips again and returns (1+1) * arccos(sim) / pi.
__device__ __forceinline__ double cosine_dist_f64(double dot, double nx2, double nq2) {
    double c = __ddiv_rn(dot, __dmul_rn(sqrt(nq2), sqrt(nx2)));
    if (fabs(c) > 1.0) c = copysign(1.0, c);
    double dm = 1.0 - c;
    double sim = 1.0 - dm;
    // a zero vector gives 0/0: scipy's and numpy's clips pass the NaN through and the reference returns
    // NaN (sorted last by its stable sort).  fmin / fmax would swallow it (distance 0, ranked first).
    if (sim != sim) return __longlong_as_double(0x7ff8000000000000ll);  // canonical NaN: above +inf as a key
    sim = sim > 1.0 ? 1.0 : (sim < -1.0 ? -1.0 : sim);
    return 2.0 * acos(sim) / 3.141592653589793;
}

// One lane per row.  The dot products follow the order of the scipy 1.15.3
// build pinned in this image (two interleaved accumulators over even / odd
// elements, summed, then the odd tail; established against cdist itself, see
// tests/test_oracle_golden.py::test_scipy_cosine_order).  Inputs are float32
// values, so every product is exact in float64 and FMA contraction is moot.
__device__ __forceinline__ double cosine_row_f64(const float* __restrict__ x, const float* __restrict__ q, int d) {
    double dot0 = 0.0, dot1 = 0.0, nx0 = 0.0, nx1 = 0.0, nq0 = 0.0, nq1 = 0.0;
    const int m = d - (d & 1);
    for (int i = 0; i < m; i += 2) {
        const double x0 = (double)x[i], x1 = (double)x[i + 1], q0 = (double)q[i], q1 = (double)q[i + 1];
        dot0 = __dadd_rn(dot0, __dmul_rn(q0, x0));
        dot1 = __dadd_rn(dot1, __dmul_rn(q1, x1));
        nx0 = __dadd_rn(nx0, __dmul_rn(x0, x0));
        nx1 = __dadd_rn(nx1, __dmul_rn(x1, x1));
        nq0 = __dadd_rn(nq0, __dmul_rn(q0, q0));
        nq1 = __dadd_rn(nq1, __dmul_rn(q1, q1));
    }
    double dot = __dadd_rn(dot0, dot1), nx = __dadd_rn(nx0, nx1), nq = __dadd_rn(nq0, nq1);
    if (d & 1) {
        const double xv = (double)x[m], qq = (double)q[m];
        dot = __dadd_rn(dot, __dmul_rn(qq, xv));
        nx = __dadd_rn(nx, __dmul_rn(xv, xv));
        nq = __dadd_rn(nq, __dmul_rn(qq, qq));
    }
    return cosine_dist_f64(dot, nx, nq);
}

// The same arithmetic with the two per-vector sums taken out of the per-candidate work: |x|^2 of
// every row is computed once at index build (dense_cos_norm_kernel) and |q|^2 once per query, both
// in the order above, so only the dot product is left per candidate; its even / odd accumulators
// are independent chains and may live in two lanes (`sub` = 0 / 1).  Bit-identical to
// cosine_row_f64.
__device__ __forceinline__ double cosine_sumsq_f64(const float* __restrict__ x, int d) {
    double n0 = 0.0, n1 = 0.0;
    const int m = d - (d & 1);
    for (int i = 0; i < m; i += 2) {
        const double x0 = (double)x[i], x1 = (double)x[i + 1];
        n0 = __dadd_rn(n0, __dmul_rn(x0, x0));
        n1 = __dadd_rn(n1, __dmul_rn(x1, x1));
    }
    double n = __dadd_rn(n0, n1);
    if (d & 1) n = __dadd_rn(n, __dmul_rn((double)x[m], (double)x[m]));
    return n;
}
// one lane: both chains
__device__ __forceinline__ double cosine_dot_f64(const float* __restrict__ x, const float* __restrict__ q, int d) {
    double d0 = 0.0, d1 = 0.0;
    const int m = d - (d & 1);
    for (int i = 0; i < m; i += 2) {
        d0 = __dadd_rn(d0, __dmul_rn((double)q[i], (double)x[i]));
        d1 = __dadd_rn(d1, __dmul_rn((double)q[i + 1], (double)x[i + 1]));
    }
    double dot = __dadd_rn(d0, d1);
    if (d & 1) dot = __dadd_rn(dot, __dmul_rn((double)q[m], (double)x[m]));
    return dot;
}
// two lanes: lane `sub` owns the chain over elements of its parity; 16-byte loads when aligned
__device__ __forceinline__ double cosine_dot_pair_f64(const float* __restrict__ x, const float* __restrict__ q, int d, int sub,
                                                      bool aligned) {
    double acc = 0.0;
    const int m = d - (d & 1);
    int i = 0;
    if (aligned) {
        // 16 row vectors (64 elements) requested at a time: the row is a random read from HBM and the
        // f64 chain behind it is short, so the loads in flight set the pace
        const int m4 = m & ~3;
        for (; i + 64 <= m4; i += 64) {
            f32x4 xr[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) xr[v] = *reinterpret_cast<const f32x4*>(x + i + 4 * v);
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const f32x4 qv = *reinterpret_cast<const f32x4*>(q + i + 4 * v);
                acc = __dadd_rn(acc, __dmul_rn((double)qv[sub], (double)xr[v][sub]));
                acc = __dadd_rn(acc, __dmul_rn((double)qv[2 + sub], (double)xr[v][2 + sub]));
            }
        }
        for (; i < m4; i += 4) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + i);
            const f32x4 qv = *reinterpret_cast<const f32x4*>(q + i);
            acc = __dadd_rn(acc, __dmul_rn((double)qv[sub], (double)xv[sub]));
            acc = __dadd_rn(acc, __dmul_rn((double)qv[2 + sub], (double)xv[2 + sub]));
        }
    }
    for (; i < m; i += 2) acc = __dadd_rn(acc, __dmul_rn((double)q[i + sub], (double)x[i + sub]));
    const double other = __shfl_xor(acc, 1);
    double dot = sub == 0 ? __dadd_rn(acc, other) : __dadd_rn(other, acc);  // dot0 + dot1 in both lanes
    if (d & 1) dot = __dadd_rn(dot, __dmul_rn((double)q[m], (double)x[m]));
    return dot;
}

// |x|^2 per row in the reference order (index build, cosine only).
static __global__ __launch_bounds__(256) void dense_cos_norm_kernel(const float* __restrict__ db, long long n, long long ld,
                                                                     int d, double* __restrict__ nx64, long long row_base) {
    const long long row = row_base + (long long)blockIdx.x * 256 + threadIdx.x;
    if (row < n) nx64[row] = cosine_sumsq_f64(db + row * ld, d);
}
// |q|^2 per query in the reference order.
static __global__ void dense_cos_qnorm_kernel(const float* __restrict__ q, int nq, int d, double* __restrict__ nq64) {
    const int qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi < nq) nq64[qi] = cosine_sumsq_f64(q + (long long)qi * d, d);
}

// ------------------------------------------------------ exact distance keys
// One lane per row: numpy's eight interleaved accumulators are eight registers,
// fed by two 16-byte row loads and two 16-byte LDS (query) reads per 8 elements.
// Requires 16-byte aligned rows (row stride and base a multiple of 16 bytes).
struct SqLeafLane {
    const float* x;
    const float* q;  // LDS copy of the query
    __device__ __forceinline__ float term(int i) const {
        const float t = __fsub_rn(x[i], q[i]);
        return __fmul_rn(t, t);
    }
    __device__ __forceinline__ float leaf(int off, int n) const {
        if (n < 8) {
            float r = 0.f;
            for (int i = 0; i < n; ++i) r = __fadd_rn(r, term(off + i));
            return r;
        }
        float r[8];
        {
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(x + off), x1 = *reinterpret_cast<const f32x4*>(x + off + 4);
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(q + off), q1 = *reinterpret_cast<const f32x4*>(q + off + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float t0 = __fsub_rn(x0[j], q0[j]), t1 = __fsub_rn(x1[j], q1[j]);
                r[j] = __fmul_rn(t0, t0);
                r[4 + j] = __fmul_rn(t1, t1);
            }
        }
        const int nfull = n - (n % 8);
        for (int i = 8; i < nfull; i += 8) {
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(x + off + i), x1 = *reinterpret_cast<const f32x4*>(x + off + i + 4);
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(q + off + i), q1 = *reinterpret_cast<const f32x4*>(q + off + i + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float t0 = __fsub_rn(x0[j], q0[j]), t1 = __fsub_rn(x1[j], q1[j]);
                r[j] = __fadd_rn(r[j], __fmul_rn(t0, t0));
                r[4 + j] = __fadd_rn(r[4 + j], __fmul_rn(t1, t1));
            }
        }
        float res = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])),
                              __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
        for (int i = nfull; i < n; ++i) res = __fadd_rn(res, term(off + i));
        return res;
    }
    // numpy pairwise recursion (split at n/2 rounded down to a multiple of 8)
    __device__ __forceinline__ float sum(int d) const {
        return pw_tree<float>([this](int off, int n) { return leaf(off, n); }, d);
    }
};

// Candidate j of query q (row = cand[q][j], or row_offset + j when cand == nullptr)
// -> key (ordered float32 euclidean distance, row).  Dynamic LDS: round_up(d,4)*4 bytes.
static __global__ __launch_bounds__(256) void dense_exact_l2_kernel(const float* __restrict__ db, long long ld, int d,
                                                              const float* __restrict__ q_orig,
                                                              const u32* __restrict__ cand, const u32* __restrict__ cnt,
                                                              u32 cap, long long implicit_n, long long row_offset,
                                                              u64* __restrict__ keys, long long key_stride,
                                                              float* __restrict__ sample, int sample_stride) {
    extern __shared__ __attribute__((aligned(16))) float s_q[];
    const int q = blockIdx.y;
    const long long M = cand ? (long long)(cnt[q] < cap ? cnt[q] : cap) : implicit_n;
    if ((long long)blockIdx.x * 256 >= M) return;
    for (int i = threadIdx.x; i < d; i += 256) s_q[i] = q_orig[(long long)q * d + i];
    __syncthreads();
    for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < M; j += (long long)gridDim.x * 256) {
        const long long row = cand ? (long long)cand[(long long)q * cap + j] : row_offset + j;
        const SqLeafLane w{db + row * ld, s_q};
        const float dist = sqrt_rn_f32(w.sum(d));
        keys[(long long)q * key_stride + j] = ((u64)ordered_f32(dist) << 32) | (u64)(u32)row;
        // every sample_stride-th exact distance feeds the threshold of the two-level select (one query per launch)
        if (sample && j % sample_stride == 0) sample[j / sample_stride] = dist == dist ? dist : __builtin_inff();
    }
}

static __global__ __launch_bounds__(256) void dense_exact_cos_kernel(const float* __restrict__ db, long long ld, int d,
                                                               const float* __restrict__ q_orig,
                                                               const u32* __restrict__ cand, const u32* __restrict__ cnt,
                                                               u32 cap, long long implicit_n, long long row_offset,
                                                               K128* __restrict__ keys, long long key_stride,
                                                               const double* __restrict__ nx64,
                                                               const double* __restrict__ nq64,
                                                               float* __restrict__ sample, int sample_stride) {
    const int q = blockIdx.y;
    const long long M = cand ? (long long)(cnt[q] < cap ? cnt[q] : cap) : implicit_n;
    const float* qv = q_orig + (long long)q * d;
    const double qq = nq64[q];
    for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < M; j += (long long)gridDim.x * 256) {
        const long long row = cand ? (long long)cand[(long long)q * cap + j] : row_offset + j;
        const double dist = cosine_dist_f64(cosine_dot_f64(db + row * ld, qv, d), nx64[row], qq);
        keys[(long long)q * key_stride + j] = K128{ordered_f64(dist), (u64)(u32)row};
        if (sample && j % sample_stride == 0) sample[j / sample_stride] = dist == dist ? __double2float_ru(dist) : __builtin_inff();
    }
}

// The exact path for a GROUP of queries: one lane per row, the row's elements held in registers a
// leaf (<= 128 elements) at a time and reused for all G queries (in LDS, broadcast reads), so the
// matrix is read from HBM once per group instead of once per query.  Arithmetic and order are those
// of the one-query kernels above (numpy pairwise float32 for L2; scipy's two float64 chains for
// cosine).  HBM bound up to ~12 queries per pass at 3 VALU operations per element and query.
// keys: [G][n]; sample: [G][ns] (every sample_stride-th distance, see dense_compact_keys_kernel) or null.
// Dynamic LDS: G * round_up(d, 4) floats.  Rows 16-byte aligned.
static constexpr int EXACT_GROUP = 8;
static constexpr int EXACT_GROUP_DEPTH = 6;  // rows up to 128 * 2^6 = 8192 elements
struct ExactGroup {
    int idx[EXACT_GROUP];  // query indices of the group (unused slots repeat the first)
    int count;
};

template <bool COSINE, class K>
static __global__ __launch_bounds__(256) void dense_exact_group_kernel(const float* __restrict__ db, long long ld, int d,
                                                                       const float* __restrict__ q_all, ExactGroup grp,
                                                                       long long n, K* __restrict__ keys,
                                                                       float* __restrict__ sample, long long ns,
                                                                       int sample_stride, const double* __restrict__ nx64,
                                                                       const double* __restrict__ nq64) {
    constexpr int G = EXACT_GROUP;
    extern __shared__ __attribute__((aligned(16))) float s_qg[];
    const int dq = (d + 3) / 4 * 4;
    for (int i = threadIdx.x; i < G * dq; i += 256) {
        const int g = i / dq, c = i - g * dq;
        s_qg[i] = c < d ? q_all[(long long)grp.idx[g] * d + c] : 0.f;
    }
    __syncthreads();
    for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < n; j += (long long)gridDim.x * 256) {
        const float* x = db + j * ld;
        if constexpr (!COSINE) {
            float acc[G];
            auto leaf = [&](int off, int m, float (&v)[G]) {
                if (m < 8) {
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        float r = 0.f;
                        for (int i = 0; i < m; ++i) {
                            const float t = __fsub_rn(x[off + i], s_qg[g * dq + off + i]);
                            r = __fadd_rn(r, __fmul_rn(t, t));
                        }
                        v[g] = r;
                    }
                    return;
                }
                const int nfull = m - (m % 8);
                const int nv = nfull / 4;  // 2..32 vectors of this leaf, all requested before the first use
                f32x4 xr[32];
#pragma unroll
                for (int u = 0; u < 32; ++u) xr[u] = *reinterpret_cast<const f32x4*>(x + off + 4 * (u < nv ? u : 0));
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const float* qg = s_qg + g * dq + off;
                    float r[8];
                    {
                        const f32x4 q0 = *reinterpret_cast<const f32x4*>(qg), q1 = *reinterpret_cast<const f32x4*>(qg + 4);
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float t0 = __fsub_rn(xr[0][c], q0[c]), t1 = __fsub_rn(xr[1][c], q1[c]);
                            r[c] = __fmul_rn(t0, t0);
                            r[4 + c] = __fmul_rn(t1, t1);
                        }
                    }
#pragma unroll
                    for (int u = 2; u < 32; u += 2) {
                        if (u < nv) {
                            const f32x4 q0 = *reinterpret_cast<const f32x4*>(qg + 4 * u);
                            const f32x4 q1 = *reinterpret_cast<const f32x4*>(qg + 4 * u + 4);
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const float t0 = __fsub_rn(xr[u][c], q0[c]), t1 = __fsub_rn(xr[u + 1][c], q1[c]);
                                r[c] = __fadd_rn(r[c], __fmul_rn(t0, t0));
                                r[4 + c] = __fadd_rn(r[4 + c], __fmul_rn(t1, t1));
                            }
                        }
                    }
                    float res = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])),
                                          __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
                    for (int i = nfull; i < m; ++i) {
                        const float t = __fsub_rn(x[off + i], qg[i]);
                        res = __fadd_rn(res, __fmul_rn(t, t));
                    }
                    v[g] = res;
                }
            };
            pw_tree_multi<float, G, EXACT_GROUP_DEPTH>(leaf, d, acc);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (g < grp.count) {
                    const float dist = sqrt_rn_f32(acc[g]);
                    keys[(long long)g * n + j] = ((u64)ordered_f32(dist) << 32) | (u64)(u32)j;
                    if (sample && j % sample_stride == 0) sample[(long long)g * ns + j / sample_stride] = dist == dist ? dist : __builtin_inff();
                }
            }
        } else {
            double d0[G], d1[G];
#pragma unroll
            for (int g = 0; g < G; ++g) d0[g] = d1[g] = 0.0;
            const int m = d - (d & 1);
            const int m4 = m & ~3;
            int i = 0;
            for (; i + 64 <= m4; i += 64) {  // 16 row vectors requested at a time
                f32x4 xr[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) xr[u] = *reinterpret_cast<const f32x4*>(x + i + 4 * u);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const float* qg = s_qg + g * dq + i;
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const f32x4 qv = *reinterpret_cast<const f32x4*>(qg + 4 * u);
                        d0[g] = __dadd_rn(d0[g], __dmul_rn((double)qv[0], (double)xr[u][0]));
                        d1[g] = __dadd_rn(d1[g], __dmul_rn((double)qv[1], (double)xr[u][1]));
                        d0[g] = __dadd_rn(d0[g], __dmul_rn((double)qv[2], (double)xr[u][2]));
                        d1[g] = __dadd_rn(d1[g], __dmul_rn((double)qv[3], (double)xr[u][3]));
                    }
                }
            }
            for (; i < m; i += 2) {
                const float x0 = x[i], x1 = x[i + 1];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    d0[g] = __dadd_rn(d0[g], __dmul_rn((double)s_qg[g * dq + i], (double)x0));
                    d1[g] = __dadd_rn(d1[g], __dmul_rn((double)s_qg[g * dq + i + 1], (double)x1));
                }
            }
            const double nx = nx64[j];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (g < grp.count) {
                    double dot = __dadd_rn(d0[g], d1[g]);
                    if (d & 1) dot = __dadd_rn(dot, __dmul_rn((double)s_qg[g * dq + m], (double)x[m]));
                    const double dist = cosine_dist_f64(dot, nx, nq64[grp.idx[g]]);
                    keys[(long long)g * n + j] = K128{ordered_f64(dist), (u64)(u32)j};
                    if (sample && j % sample_stride == 0)
                        sample[(long long)g * ns + j / sample_stride] = dist == dist ? __double2float_ru(dist) : __builtin_inff();
                }
            }
        }
    }
}

// Two-level select of the exact path: keys whose distance is <= *thr (an exact distance of a sampled
// row with at least k sampled rows at or below it) are appended to `out`; *cnt counts them all, also
// beyond `cap` (the caller then selects over the full key array instead).
__device__ __forceinline__ bool key_within(u64 key, float t) { return (u32)(key >> 32) <= ordered_f32(t); }
__device__ __forceinline__ bool key_within(const K128& key, float t) { return key.hi <= ordered_f64((double)t); }

template <class K>
static __global__ __launch_bounds__(256) void dense_compact_keys_kernel(const K* __restrict__ keys, long long n,
                                                                        const float* __restrict__ thr, K* __restrict__ out,
                                                                        u32 cap, u32* __restrict__ cnt) {
    keys += (long long)blockIdx.y * n;  // one query of the group per grid row
    out += (long long)blockIdx.y * cap;
    cnt += blockIdx.y;
    const float t = thr[blockIdx.y];
    // Survivors are rare (~2 k sample_stride of n).  A workgroup stages its own in LDS (wave-aggregated LDS
    // atomics) and reserves their place in `out` with ONE global atomic at the end: returning atomics on one
    // address serialise in L2 at ~7 ns each, and one per surviving wave made them the whole kernel (0.6 ms
    // for eight 10 M-key arrays).  A workgroup with more survivors than the stage holds (tie groups) sends
    // the excess straight to `out`.
    constexpr int U = 4, STAGE = 2048;
    __shared__ K s_keys[STAGE];
    __shared__ u32 s_n, s_base;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const long long step = (long long)gridDim.x * 256 * U;
    const long long n_up = (n + 63) / 64 * 64;  // whole waves take part in the ballots
    const int lane = threadIdx.x & 63;
    for (long long j0 = (long long)blockIdx.x * 256 * U + threadIdx.x; j0 < n_up; j0 += step) {
        K key[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {  // four keys per lane, all requested before the first is looked at
            const long long j = j0 + (long long)u * 256;
            key[u] = j < n ? keys[j] : K{};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long j = j0 + (long long)u * 256;
            const bool pass = j < n && key_within(key[u], t);
            const u64 m = __ballot(pass);
            if (m == 0) continue;
            u32 base = 0;
            if (lane == 0) base = atomicAdd(&s_n, (u32)__popcll(m));
            base = __shfl(base, 0);
            if (pass) {
                const u32 pos = base + (u32)__popcll(m & ((1ull << lane) - 1ull));
                if (pos < (u32)STAGE) {
                    s_keys[pos] = key[u];
                } else {
                    const u32 gpos = atomicAdd(cnt, 1u);
                    if (gpos < cap) out[gpos] = key[u];
                }
            }
        }
    }
    __syncthreads();
    const u32 staged = s_n < (u32)STAGE ? s_n : (u32)STAGE;
    if (staged == 0) return;
    if (threadIdx.x == 0) s_base = atomicAdd(cnt, staged);
    __syncthreads();
    for (u32 i = threadIdx.x; i < staged; i += 256) {
        const u32 gpos = s_base + i;
        if (gpos < cap) out[gpos] = s_keys[i];
    }
}

// Two lanes per row (lane c of the pair owns numpy's accumulators 4c..4c+3):
// 16-byte row loads, 16-byte LDS query reads, halves the serial chain of the
// one-lane form and doubles the rows in flight.  Both lanes return the sum.
struct SqLeafPair {
    const float* x;  // 16-byte aligned row
    const float* q;  // LDS copy of the query (16-byte aligned)
    int c;           // 0 / 1: which half of the eight accumulators
    __device__ __forceinline__ float term(int i) const {
        const float t = __fsub_rn(x[i], q[i]);
        return __fmul_rn(t, t);
    }
    __device__ __forceinline__ float leaf(int off, int n) const {
        if (n < 8) {
            float r = 0.f;
            for (int i = 0; i < n; ++i) r = __fadd_rn(r, term(off + i));
            return r;
        }
        // A leaf has at most 128 elements (numpy's block size): this lane's 16 row vectors are all requested
        // before the first is used -- the rows are random 512-byte reads from HBM, and a load/use loop
        // pays that latency once per unroll group (5 round trips per row at d = 128) instead of once.
        const int nfull = n - (n % 8);
        const int nv = nfull / 8;  // 1..16, uniform
        f32x4 xr[16];
#pragma unroll
        for (int v = 0; v < 16; ++v) xr[v] = *reinterpret_cast<const f32x4*>(x + off + 8 * (v < nv ? v : 0) + 4 * c);
        float r[4];
        {
            const f32x4 qv = *reinterpret_cast<const f32x4*>(q + off + 4 * c);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float t = __fsub_rn(xr[0][j], qv[j]);
                r[j] = __fmul_rn(t, t);
            }
        }
#pragma unroll
        for (int v = 1; v < 16; ++v) {
            if (v < nv) {
                const f32x4 qv = *reinterpret_cast<const f32x4*>(q + off + 8 * v + 4 * c);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float t = __fsub_rn(xr[v][j], qv[j]);
                    r[j] = __fadd_rn(r[j], __fmul_rn(t, t));
                }
            }
        }
        const float part = __fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3]));
        float res = __fadd_rn(part, __shfl_xor(part, 1));  // ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7))
        for (int i = nfull; i < n; ++i) res = __fadd_rn(res, term(off + i));
        return res;
    }
    __device__ __forceinline__ float sum(int d) const {
        return pw_tree<float>([this](int off, int n) { return leaf(off, n); }, d);
    }
};

// Per block (= `waves_per_block` survivor segments of ONE scan workgroup, hence one
// group of `group_q` = 32, 64 or 128 queries).  A segment entry is (first row,
// mask << 16 | query within the group): bit i of the mask is row first +
// (i&3) + 8(i>>2) (the MFMA accumulator layout of one lane).  (1) count the
// survivors per query in LDS, (2) reserve a range in every touched query's key
// list with ONE global atomic per (block, query), (3) exact distance per
// survivor (L2: two lanes per entry; cosine: one lane per entry; a lane walks
// the bits of its entry), key stored at its reserved slot.  A one-tile group
// stages its 32 query vectors in LDS (dynamic LDS: 32 * (ldq + 4) floats); larger
// groups read the query rows through the cache.
static constexpr int RERANK_MAX_GROUP = 128;
static constexpr int RERANK_STAGE_STRIDE = 68;  // floats per staged row piece: 64 + 4 (272 bytes)

// The LDS a re-rank workgroup works in.  The stand-alone kernels hand in arrays of their own; the fused tail of the int8
// full pass (sq_dense_i8.hpp: dense8_body_kernel) hands in pieces of the ring its waves have finished with.
struct RerankLds {
    float* qrows;   // 32 * (ldq + 4) floats when the one-tile group's queries are staged (ldq <= 156), else unused
    u32* hist;      // [RERANK_MAX_GROUP] x 3
    u32* base;
    u32* fill;
    float* stage;   // cosine, wide rows: one row stage of 32 * RERANK_STAGE_STRIDE floats per wave of the workgroup (or null)
    // Second-level filter (the int8 full pass's tail, sq_dense_i8.hpp "the tightened threshold"): an entry takes part only
    // if its score -- the smallest filter score of its rows, stored beside it -- is at or below its query's thr2; the
    // entries that pass are first compacted into `list` (all of them idle lanes otherwise: a few dozen of a few thousand).
    const float* scores = nullptr;   // global, parallel to wave_out (or null: every entry takes part)
    const float* thr2 = nullptr;     // LDS [group_q]
    uint2* list = nullptr;           // LDS [list_cap]
    u32* npass = nullptr;            // LDS counter
    u32 list_cap = 0;
    int cnt_shift = 0;               // the per-query candidate counters are 1 << cnt_shift words apart
    const u32* seg_cnt = nullptr;    // LDS [waves_per_block]: the segments' entry counts (the fused tail: no global round trip; first query tile 0)
};

template <class K, bool COSINE>
__device__ __forceinline__ void rerank_block(const float* __restrict__ db, long long ld, int d,
                                             const float* __restrict__ q_al, int ldq, int nq, int group_q,
                                             const uint2* __restrict__ wave_out, const u32* __restrict__ wave_cnt,
                                             u32 wave_cap, long long n_waves, int waves_per_block,
                                             K* __restrict__ keys, u32* __restrict__ cnt, u32 cap,
                                             u32* __restrict__ overflow, const double* __restrict__ nx64,
                                             const double* __restrict__ nq64, int debug, long long w0, const RerankLds& L) {
    float* s_qrows = L.qrows;
    u32 *s_hist = L.hist, *s_base = L.base, *s_fill = L.fill;
    const int ldl = ldq + 4;  // LDS row stride: +16 bytes so that different query rows hit different banks
    // the 32 query vectors of a one-tile group sit in LDS while that costs little occupancy (d <= 156);
    // wider rows (66 KB at d = 512: two workgroups per CU) and larger groups read them through the cache
    const bool q_in_lds = group_q == 32 && ldq <= 156;
    if (w0 >= n_waves) return;
    const u32 q0 = L.seg_cnt ? 0u : wave_cnt[2 * w0 + 1] * 32u;  // first query of the group (the same for all segments of the block)
    if (threadIdx.x < RERANK_MAX_GROUP) {
        s_hist[threadIdx.x] = 0;
        s_fill[threadIdx.x] = 0;
    }
    __syncthreads();
    // The block's segments as ONE list of entries (entry g of the block = entry g - off[wi] of segment wi): a loop over the
    // segments would be a chain of dependent gather round trips per segment (the fused tail has eight of them).
    constexpr int MAXSEG = 8;
    u32 off[MAXSEG + 1];
    off[0] = 0;
#pragma unroll
    for (int wi = 0; wi < MAXSEG; ++wi) {
        u32 c = 0;
        if (wi < waves_per_block && w0 + wi < n_waves) {
            c = L.seg_cnt ? L.seg_cnt[wi] : wave_cnt[2 * (w0 + wi)];
            if (c > wave_cap) {
                if (threadIdx.x == 0) atomicOr(overflow, 1u);
                c = wave_cap;
            }
        }
        off[wi + 1] = off[wi] + c;
    }
    const u32 total = off[MAXSEG];
    auto entry_at = [&](u32 g) __attribute__((always_inline)) -> const uint2* {
        int wi = 0;
#pragma unroll
        for (int j = 1; j < MAXSEG; ++j) wi += g >= off[j] ? 1 : 0;
        return wave_out + (w0 + wi) * wave_cap + (g - off[wi]);
    };
    if (total == 0) return;  // uniform: every thread read the same counts
    const bool filtered = L.scores != nullptr;
    auto passes = [&](u32 g, const uint2& ent) __attribute__((always_inline)) -> bool {
        const uint2* ep = entry_at(g);
        return L.scores[ep - wave_out] <= L.thr2[ent.y & 0xffffu];   // (an always-candidate row's -inf passes; NaN never)
    };
    if (filtered && threadIdx.x == 0) *L.npass = 0u;
    if (filtered) __syncthreads();
    if (filtered) {
        // (four entries and their scores per thread and round trip: the tail is a chain of dependent memory latencies)
        constexpr int PB = 4;
        for (u32 g0 = threadIdx.x; g0 < total; g0 += blockDim.x * PB) {
            uint2 ent[PB];
            float sc[PB];
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                const u32 g = g0 + (u32)j * blockDim.x;
                const uint2* ep = entry_at(g < total ? g : 0u);
                ent[j] = *ep;
                sc[j] = L.scores[ep - wave_out];
            }
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                const u32 g = g0 + (u32)j * blockDim.x;
                if (g < total && sc[j] <= L.thr2[ent[j].y & 0xffffu]) {   // (an always-candidate row's -inf passes; NaN never)
                    const u32 slot = atomicAdd(L.npass, 1u);
                    if (slot < L.list_cap) L.list[slot] = ent[j];
                    atomicAdd(&s_hist[ent[j].y & 0xffffu], (u32)__popc(ent[j].y >> 16));
                }
            }
        }
    } else {
        for (u32 g = threadIdx.x; g < total; g += blockDim.x) {
            const u32 ey = entry_at(g)->y;
            atomicAdd(&s_hist[ey & 0xffffu], (u32)__popc(ey >> 16));
        }
    }
    __syncthreads();
    // compact: entries 0 .. n_ent of the list; more passed than the list holds: all entries again, the others masked out
    const bool compact = filtered && *L.npass <= L.list_cap;
    const u32 n_ent = compact ? *L.npass : total;
    if (n_ent == 0) return;
    auto load_entry = [&](u32 e) __attribute__((always_inline)) -> uint2 {
        if (compact) return L.list[e];
        uint2 ent = *entry_at(e);
        if (filtered && !passes(e, ent)) ent.y &= 0xffffu;
        return ent;
    };
    if (threadIdx.x < group_q) {
        const u32 hcount = s_hist[threadIdx.x];
        // (debug 32 / 64: measurement ablations -- no reservation / rows from a cache-resident range; results are garbage)
        s_base[threadIdx.x] = (hcount && !(debug & 32)) ? atomicAdd(&cnt[(long long)(q0 + threadIdx.x) << L.cnt_shift], hcount) : 0u;
    }
    if (q_in_lds) {
        const int vpr = ldq / 4;  // 16-byte vectors per query row (ldq % 4 == 0, rows 16-byte aligned)
        for (int i = threadIdx.x; i < 32 * vpr; i += blockDim.x) {
            const int r = i / vpr, cc = i - r * vpr;
            const long long qg = (long long)q0 + r;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (qg < nq) v = *reinterpret_cast<const f32x4*>(q_al + qg * ldq + 4 * cc);
            *reinterpret_cast<f32x4*>(s_qrows + r * ldl + 4 * cc) = v;
        }
    }
    __syncthreads();
    constexpr int LPR = 2;  // lanes per entry (L2: four of numpy's eight accumulators each; cosine: one parity each)
    const bool rows_aligned = (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(db) & 15u) == 0;
    const int sub = threadIdx.x % LPR;
    if constexpr (COSINE) {
        // Wide rows (d % 64 == 0: the 512-wide shards of BASELINE config 4): a pair walking its row 16 bytes at
        // a time leaves each 128-byte line of 32 different rows to be touched by eight separate wave loads, and
        // with a CU's waves all doing that the lines fall out of the vector cache in between (1.97 ms for 860 k
        // candidates = 0.9 TB/s of L2 refetches).  Here a wave fetches the current rows of its 32 pairs
        // TOGETHER, 64 elements at a time: 16 lanes per row and instruction, whole lines, into a per-wave LDS
        // stage (row stride 272 bytes: two-way bank conflicts at most), and each pair then reads its row's piece
        // from the stage.  The arithmetic and its order are untouched (scipy's two chains, one per lane).
        if (L.stage && rows_aligned && (d & 63) == 0 && !(debug & 512)) {
            const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, pair = lane >> 1;
            float* stage = L.stage + wv * (32 * RERANK_STAGE_STRIDE);
            {
                const u32 per_round = blockDim.x / LPR;
                for (u32 e0 = 0; e0 < n_ent; e0 += per_round) {   // uniform trip count: the waves work in lockstep below
                    const u32 e = e0 + threadIdx.x / LPR;
                    const bool live = e < n_ent;
                    const uint2 ent = live ? load_entry(e) : make_uint2(0u, 0u);
                    const u32 ql = ent.y & 0xffffu;
                    const u32 qg = q0 + ql;
                    const float* qrow = q_in_lds ? s_qrows + ql * ldl : q_al + (long long)qg * ldq;
                    u32 mask = live ? ent.y >> 16 : 0u;
                    while (__any(mask != 0)) {
                        const bool has = mask != 0;
                        const int i = has ? __ffs((int)mask) - 1 : 0;
                        mask &= mask - 1;                       // 0 stays 0
                        u32 row = has ? ent.x + (u32)((i & 3) + 8 * (i >> 2)) : 0u;
                        if (debug & 64) row &= 1023u;
                        double acc = 0.0;
                        for (int k0 = 0; k0 < d; k0 += 64) {
                            f32x4 v[8];
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                const int slot = 4 * j + (lane >> 4);
                                const u32 r = (u32)__shfl((int)row, 2 * slot);
                                v[j] = *reinterpret_cast<const f32x4*>(db + (long long)r * ld + k0 + 4 * (lane & 15));
                            }
#pragma unroll
                            for (int j = 0; j < 8; ++j)
                                *reinterpret_cast<f32x4*>(stage + (4 * j + (lane >> 4)) * RERANK_STAGE_STRIDE + 4 * (lane & 15)) = v[j];
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                            const float* sx = stage + pair * RERANK_STAGE_STRIDE;
#pragma unroll
                            for (int u = 0; u < 16; ++u) {
                                const f32x4 xv = *reinterpret_cast<const f32x4*>(sx + 4 * u);
                                const f32x4 qv = *reinterpret_cast<const f32x4*>(qrow + k0 + 4 * u);
                                acc = __dadd_rn(acc, __dmul_rn((double)qv[sub], (double)xv[sub]));
                                acc = __dadd_rn(acc, __dmul_rn((double)qv[2 + sub], (double)xv[2 + sub]));
                            }
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();   // the stage is rewritten by the next piece
                        }
                        const double other = __shfl_xor(acc, 1);
                        const double dot = sub == 0 ? __dadd_rn(acc, other) : __dadd_rn(other, acc);  // dot0 + dot1
                        if (has && sub == 0) {
                            const double dist = cosine_dist_f64(dot, nx64[row], nq64[qg]);
                            const u32 pos = s_base[ql] + atomicAdd(&s_fill[ql], 1u);
                            if (pos < cap) keys[(long long)qg * cap + pos] = K128{ordered_f64(dist), (u64)row};
                        }
                    }
                }
            }
            return;
        }
    }
    {
        for (u32 e = threadIdx.x / LPR; e < n_ent; e += blockDim.x / LPR) {
            const uint2 ent = load_entry(e);
            const u32 ql = ent.y & 0xffffu;
            const u32 qg = q0 + ql;
            const float* qrow = q_in_lds ? s_qrows + ql * ldl : q_al + (long long)qg * ldq;
            u32 mask = ent.y >> 16;  // both lanes of a pair hold the same entry: they stay converged
            while (mask) {
                const int i = __ffs((int)mask) - 1;
                mask &= mask - 1;
                u32 row = ent.x + (u32)((i & 3) + 8 * (i >> 2));
                if (debug & 64) row &= 1023u;
                if constexpr (COSINE) {
                    const double dot = cosine_dot_pair_f64(db + (long long)row * ld, qrow, d, sub, rows_aligned);
                    if (sub == 0) {
                        const double dist = cosine_dist_f64(dot, nx64[row], nq64[qg]);
                        const u32 pos = s_base[ql] + atomicAdd(&s_fill[ql], 1u);
                        if (pos < cap) keys[(long long)qg * cap + pos] = K128{ordered_f64(dist), (u64)row};
                    }
                } else {
                    const SqLeafPair pr{db + (long long)row * ld, qrow, sub};
                    const float dist = sqrt_rn_f32(pr.sum(d));
                    if (sub == 0) {
                        const u32 pos = s_base[ql] + atomicAdd(&s_fill[ql], 1u);
                        if (pos < cap) keys[(long long)qg * cap + pos] = ((u64)ordered_f32(dist) << 32) | (u64)row;
                    }
                }
            }
        }
    }
}

// Survivors of the scan arrive as per-wave segments of (row, query) pairs
// (dense_scan_kernel writes them with plain stores); the returning atomics on
// the per-query counters live here, outside the streaming kernel.  A wave that
// overflowed its segment raises `overflow` (every query of the call then takes
// the exact path; only degenerate thresholds get there).
// q_al: queries copied to [nq_pad][ldq] floats, ldq % 4 == 0, 16-byte aligned.
static __global__ __launch_bounds__(512) void dense_rerank_l2_kernel(
    const float* __restrict__ db, long long ld, int d, const float* __restrict__ q_al, int ldq,
    const uint2* __restrict__ wave_out, const u32* __restrict__ wave_cnt, u32 wave_cap, long long n_waves,
    int waves_per_block, int nq, int group_q, u64* __restrict__ keys, u32* __restrict__ cnt, u32 cap,
    u32* __restrict__ overflow, int debug) {
    extern __shared__ __attribute__((aligned(16))) float s_qrows_dyn[];
    __shared__ u32 s_hist[RERANK_MAX_GROUP], s_base[RERANK_MAX_GROUP], s_fill[RERANK_MAX_GROUP];
    rerank_block<u64, false>(db, ld, d, q_al, ldq, nq, group_q, wave_out, wave_cnt, wave_cap, n_waves, waves_per_block,
                             keys, cnt, cap, overflow, nullptr, nullptr, debug, (long long)blockIdx.x * waves_per_block,
                             RerankLds{s_qrows_dyn, s_hist, s_base, s_fill, nullptr});
}

static __global__ __launch_bounds__(512) void dense_rerank_cos_kernel(
    const float* __restrict__ db, long long ld, int d, const float* __restrict__ q_al, int ldq,
    const uint2* __restrict__ wave_out, const u32* __restrict__ wave_cnt, u32 wave_cap, long long n_waves,
    int waves_per_block, int nq, int group_q, K128* __restrict__ keys, u32* __restrict__ cnt, u32 cap,
    u32* __restrict__ overflow, const double* __restrict__ nx64, const double* __restrict__ nq64, int debug) {
    extern __shared__ __attribute__((aligned(16))) float s_qrows_dyn[];
    __shared__ u32 s_hist[RERANK_MAX_GROUP], s_base[RERANK_MAX_GROUP], s_fill[RERANK_MAX_GROUP];
    __shared__ __attribute__((aligned(16))) float s_stage[4][32 * RERANK_STAGE_STRIDE];
    rerank_block<K128, true>(db, ld, d, q_al, ldq, nq, group_q, wave_out, wave_cnt, wave_cap, n_waves, waves_per_block,
                             keys, cnt, cap, overflow, nx64, nq64, debug, (long long)blockIdx.x * waves_per_block,
                             RerankLds{s_qrows_dyn, s_hist, s_base, s_fill, blockDim.x <= 256 ? &s_stage[0][0] : nullptr});
}

// Plain distance vectors for sq_dense_distances (one query, n gathered rows),
// in the rows' own dtype like metrics.euclidean_distance (float32 in -> float32
// out, float64 in -> float64 out); cosine is always float64 (scipy cdist).
__device__ __forceinline__ float sub_rn(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ double sub_rn(double a, double b) { return __dsub_rn(a, b); }
__device__ __forceinline__ float mul_rn_t(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ double mul_rn_t(double a, double b) { return __dmul_rn(a, b); }

template <class T>
__device__ __forceinline__ double cosine_row_t(const T* __restrict__ x, const T* __restrict__ q, int d) {
    double dot0 = 0.0, dot1 = 0.0, nx0 = 0.0, nx1 = 0.0, nq0 = 0.0, nq1 = 0.0;
    const int m = d - (d & 1);
    for (int i = 0; i < m; i += 2) {
        const double x0 = (double)x[i], x1 = (double)x[i + 1], q0 = (double)q[i], q1 = (double)q[i + 1];
        dot0 = __dadd_rn(dot0, __dmul_rn(q0, x0));
        dot1 = __dadd_rn(dot1, __dmul_rn(q1, x1));
        nx0 = __dadd_rn(nx0, __dmul_rn(x0, x0));
        nx1 = __dadd_rn(nx1, __dmul_rn(x1, x1));
        nq0 = __dadd_rn(nq0, __dmul_rn(q0, q0));
        nq1 = __dadd_rn(nq1, __dmul_rn(q1, q1));
    }
    double dot = __dadd_rn(dot0, dot1), nx = __dadd_rn(nx0, nx1), nq = __dadd_rn(nq0, nq1);
    if (d & 1) {
        const double xv = (double)x[m], qq = (double)q[m];
        dot = __dadd_rn(dot, __dmul_rn(qq, xv));
        nx = __dadd_rn(nx, __dmul_rn(xv, xv));
        nq = __dadd_rn(nq, __dmul_rn(qq, qq));
    }
    return cosine_dist_f64(dot, nx, nq);
}

template <class T>
__global__ __launch_bounds__(256) void dense_distances_kernel(const T* __restrict__ rows, long long n, int d,
                                                               const T* __restrict__ q, int metric,
                                                               T* __restrict__ out_t, double* __restrict__ out64) {
    const int j8 = threadIdx.x & 7;
    const long long j = (long long)blockIdx.x * 32 + (threadIdx.x >> 3);
    const long long jc = j < n ? j : n - 1;
    const T* x = rows + jc * d;
    if (metric == SQ_METRIC_L2) {
        auto term = [x, q](int i) {
            const T t = sub_rn(x[i], q[i]);
            return mul_rn_t(t, t);
        };
        const T s = np_pairwise_sum<T>(term, d, j8);
        if (j8 == 0 && j < n) {
            if constexpr (sizeof(T) == 4)
                out_t[j] = sqrt_rn_f32(s);
            else
                out_t[j] = sqrt(s);
        }
    } else {
        if (j8 == 0 && j < n) out64[j] = cosine_row_t<T>(x, q, d);
    }
}

// Error model of the filter score (bf16 MFMA), per row:
//   |s~ - s| <= eps_a |x||q| + eps_b (|x|^2 + 2|x||q|),   s = |x|^2 - 2 x.q
//   L2: with |x||q| <= (|x|^2 + |q|^2)/2 this is <= alpha |x|^2 + beta |q|^2,
//       alpha = eps_a/2 + 2 eps_b, beta = eps_a/2 + eps_b.  The row's share is folded into the stored norm
//       (the scan starts from n' = RD(|x|^2 (1 - alpha))), so the kernel's score s~' obeys
//           s - (2 alpha + eps_b)|x|^2 - beta|q|^2  <=  s~'  <=  s + beta|q|^2
//       and a heavy-tailed norm distribution costs nothing: a far row carries its own slack.
//   cosine: unit vectors, eps = eps_a + eps_b.
struct FilterBound {
    double alpha, beta, eps_b;  // L2
    double eps_cos;             // cosine
    double xn2_max;             // largest squared row norm
};
__device__ __host__ __forceinline__ FilterBound filter_bound(int cosine, double eps_a, double eps_b, double xn2_max) {
    return FilterBound{0.5 * eps_a + 2.0 * eps_b, 0.5 * eps_a + eps_b, eps_b, eps_a + eps_b, xn2_max};
}

// The sampled threshold T is the kernel score of an actual row with at least k sampled rows at or below
// it.  T -> T' such that every row of the true top-k has kernel score <= T' (applied by
// kth_threshold_f32_kernel as it stores the threshold):
//   L2: rows with s~' <= T have |x| <= rho, gamma rho^2 - 2|q| rho - (beta|q|^2 + T) = 0, gamma = 1 - 2alpha - eps_b
//       (from the lower bound above and x.q <= |x||q|), so the true k-th score is
//       <= U = T + (2alpha + eps_b) min(rho^2, X^2) + beta|q|^2, and a row with s <= U has s~' <= U + beta|q|^2.
//       (+ 4e-6 |T + |q|^2|: rounding of the float32 numpy-order distance the certification compares with.)
//   cosine: T' = T + 2 eps.
// With `raw_q` set the threshold kernel's prologue does what is left of dense_prep_queries_kernel for query q (the bf16
// planes are built by the scan kernel itself, DenseScanArgs::raw_q): |q - c|^2 in float64, the candidate counter, the
// overflow flag, the aligned float32 copy the re-rank reads, and the state of the padding queries of the last tile.
struct DenseThrPost {
    const double* qn2;
    int cosine;
    FilterBound fb;
    // fused prep (L2 only; nullptr: dense_prep_queries_kernel ran before the sample pass)
    const float* raw_q = nullptr;
    int nq = 0, nq_pad = 0, d = 0, ldq = 0;
    const float* center = nullptr;
    double* qn2_out = nullptr;
    float* thr_out = nullptr;
    u32* cnt = nullptr;
    u32* oflag = nullptr;
    float* q_al = nullptr;
    float* traw_out = nullptr;   // optional: the threshold before the slack (sq_dense_tighten.hpp sizes its histogram bins with it)
    __device__ __forceinline__ void prologue(int q, double* red) const {
        if (!raw_q) return;
        const int T = blockDim.x;
        for (int i = threadIdx.x; i < ldq; i += T) q_al[(long long)q * ldq + i] = i < d ? raw_q[(long long)q * d + i] : 0.f;
        double acc = 0.0;
        for (int i = threadIdx.x; i < d; i += T) {
            const float v = raw_q[(long long)q * d + i];
            const float c = center ? __fsub_rn(v, center[i]) : v;
            acc += (double)c * (double)c;
        }
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double tot = 0.0;
            for (int w = 0; w < (T >> 6); ++w) tot += red[w];
            qn2_out[q] = tot;
            cnt[q] = 0u;
            if (q == 0) *oflag = 0u;
            for (int p = nq + q; p < nq_pad; p += gridDim.x) {  // padding queries of the last tile: nothing passes
                qn2_out[p] = 0.0;
                thr_out[p] = -__builtin_inff();
                cnt[p] = 0u;
            }
        }
        __syncthreads();  // qn2[q] is read by the thread that publishes the threshold
    }
    __device__ __forceinline__ float operator()(int q, float t) const {
        if (traw_out) traw_out[q] = t;
        if (!(t < __builtin_inff())) return t;
        double slack;
        if (cosine) {
            slack = 2.0 * fb.eps_cos + 1e-8;
        } else {
            const double Q = qn2[q];
            const double gamma = 1.0 - 2.0 * fb.alpha - fb.eps_b;
            const double c = fb.beta * Q + (double)t;
            const double disc = Q + gamma * c;
            double rho = (sqrt(Q) + sqrt(disc > 0.0 ? disc : 0.0)) / gamma;
            double rho2 = rho * rho * (1.0 + 1e-9);
            if (rho2 > fb.xn2_max) rho2 = fb.xn2_max;
            slack = (2.0 * fb.alpha + fb.eps_b) * rho2 + 2.0 * fb.beta * Q + 4e-6 * fabs((double)t + Q);
        }
        // round up so the float threshold is never below T + slack
        float r = (float)((double)t + slack);
        if ((double)r < (double)t + slack) r = __uint_as_float(__float_as_uint(r) + (r >= 0.f ? 1 : -1));
        return r;
    }
};

// --------------------------------------------------------------- finalize
// status bits: 1 candidate overflow, 2 certification failed, 4 fewer than kk candidates
// certify: 0 = nothing to check (all rows were candidates), 1 = counts and the filter bound,
// 2 = counts only (two-level select of the exact path: the threshold is itself an exact distance)
// Post-ops of select_topk_kernel (one workgroup per query, all threads call): keys -> (distance, id),
// certification, status word.  `status` and `cnt_out` may be host-mapped memory: the host then only
// has to synchronise the stream.  q0: index of the launch's first query in the per-query arrays
// (the exact path runs one query per launch).
// What changes from call to call in a captured call graph (sq_dense.hip, "dense_graph"): the caller's pointers, read by the
// first and the last kernel of the chain from a pinned block the host fills before each launch of the graph.
struct DenseCallPtrs {
    const float* q;
    void* out_dist;
    long long* out_idx;
};

struct DenseFinalizeL2 {
    const u32* cnt;
    u32 cap;
    int kk;
    long long id_base;
    const float* thr;
    const double* qn2;
    double beta;  // FilterBound::beta: a non-candidate (s~' > T') has s > T' - beta |q|^2
    int certify;
    float* out_dist;
    long long* out_idx;
    u32* status;
    u32* cnt_out;
    const u32* overflow;
    int q0;
    ExactGroup sel = ExactGroup{{0, 0, 0, 0, 0, 0, 0, 0}, 0};  // count > 0: query ql of the launch is sel.idx[ql], its count cnt[ql]
    const int* qmap = nullptr;  // the middle tier: query ql of the launch is qmap[ql]; cnt, thr and qn2 are the launch's own arrays
    const float2* lin = nullptr;  // the int8 filter: a non-candidate (s~ > T') has s > T' - lin[q].y (its slack is linear in |q|, not beta |q|^2)
    const u32* thr2k = nullptr;   // the int8 full pass's tightened thresholds: ordered keys, the largest any workgroup applied (0: none did); 1 << cnt_shift words apart
    int cnt_shift = 0;            // cnt (and thr2k) entries are 1 << cnt_shift words apart (the fused int8 call: a cache line each)
    const DenseCallPtrs* ind = nullptr;  // captured call graph: the outputs of THIS launch
    __device__ __forceinline__ void operator()(int ql, const u64* sorted, int k) const {
        const int q = qmap ? qmap[ql] : (sel.count ? sel.idx[ql] : q0 + ql);
        const int qa = qmap ? ql : q;  // index into thr / qn2
        float* out_dist = ind ? static_cast<float*>(ind->out_dist) : this->out_dist;
        long long* out_idx = ind ? ind->out_idx : this->out_idx;
        for (int j = threadIdx.x; j < k; j += blockDim.x) {
            const u64 key = sorted[j];
            const bool pad = key == ~0ull;
            out_dist[(long long)q * k + j] = pad ? __builtin_inff() : unordered_f32((u32)(key >> 32));
            out_idx[(long long)q * k + j] = pad ? -1ll : id_base + (long long)(key & 0xffffffffull);
        }
        if (threadIdx.x == 0) {
            u32 st = 0;
            const u32 c = cnt[(long long)((sel.count || qmap) ? ql : q) << cnt_shift];
            if (certify) {
                if (c > cap || (overflow && *overflow)) st |= 1u;
                if (c < (u32)kk) st |= 4u;
                if (st == 0 && certify == 1) {
                    const double dk = (double)unordered_f32((u32)(sorted[kk - 1] >> 32));
                    const u32 t2k = thr2k ? thr2k[(long long)qa << cnt_shift] : 0u;
                    const double t = t2k ? (double)unordered_f32(t2k) : (double)thr[qa];
                    const double lo2 = t + qn2[qa] * (1.0 - beta) - (lin ? (double)lin[qa].y : 0.0);  // smallest squared distance a non-candidate can have
                    const double bound = lo2 > 0.0 ? sqrt(lo2) * (1.0 - 1e-6) : 0.0;
                    if (!(t == (double)__builtin_inff()) && !(dk < bound)) st |= 2u;
                }
            }
            status[q] = st;
            if (cnt_out) cnt_out[q] = c;
        }
    }
};

struct DenseFinalizeCos {
    const u32* cnt;
    u32 cap;
    int kk;
    long long id_base;
    const float* thr;
    double eps;
    int certify;
    double* out_dist;
    long long* out_idx;
    u32* status;
    u32* cnt_out;
    const u32* overflow;
    int q0;
    ExactGroup sel = ExactGroup{{0, 0, 0, 0, 0, 0, 0, 0}, 0};
    const float2* lin = nullptr;          // the int8 filter: the query's own slack lin[q].y instead of `eps`
    const u32* thr2k = nullptr;           // as DenseFinalizeL2::thr2k
    int cnt_shift = 0;
    const DenseCallPtrs* ind = nullptr;   // captured call graph: the outputs of THIS launch
    const int* qmap = nullptr;            // the middle tier: query ql of the launch is qmap[ql]; cnt, thr and lin are the launch's own arrays
    __device__ __forceinline__ void operator()(int ql, const K128* sorted, int k) const {
        const int q = qmap ? qmap[ql] : (sel.count ? sel.idx[ql] : q0 + ql);
        const int qa = qmap ? ql : q;  // index into thr / lin
        double* out_dist = ind ? static_cast<double*>(ind->out_dist) : this->out_dist;
        long long* out_idx = ind ? ind->out_idx : this->out_idx;
        for (int j = threadIdx.x; j < k; j += blockDim.x) {
            const K128 key = sorted[j];
            const bool pad = key.hi == ~0ull && key.lo == ~0ull;
            out_dist[(long long)q * k + j] = pad ? (double)__builtin_inff() : unordered_f64(key.hi);
            out_idx[(long long)q * k + j] = pad ? -1ll : id_base + (long long)(key.lo & 0xffffffffull);
        }
        if (threadIdx.x == 0) {
            u32 st = 0;
            const u32 c = cnt[(long long)((sel.count || qmap) ? ql : q) << cnt_shift];
            if (certify) {
                if (c > cap || (overflow && *overflow)) st |= 1u;
                if (c < (u32)kk) st |= 4u;
                if (st == 0 && certify == 1) {
                    const double dk = unordered_f64(sorted[kk - 1].hi);
                    const u32 t2k = thr2k ? thr2k[(long long)qa << cnt_shift] : 0u;
                    const double t = t2k ? (double)unordered_f32(t2k) : (double)thr[qa];  // threshold on -sim~
                    // non-candidates: -sim~ > t  =>  sim < -t + eps  =>  dist > 2 acos(min(1,-t+eps))/pi
                    double smax = -t + (lin ? (double)lin[qa].y : eps);
                    smax = smax > 1.0 ? 1.0 : (smax < -1.0 ? -1.0 : smax);
                    const double bound = 2.0 * acos(smax) / 3.141592653589793 - 1e-9;
                    if (!(t == (double)__builtin_inff()) && !(dk < bound)) st |= 2u;
                }
            }
            status[q] = st;
            if (cnt_out) cnt_out[q] = c;
        }
    }
};

}  // namespace sq
