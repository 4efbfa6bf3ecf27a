// The int8 first-stage filter of the dense search (L2 and cosine, d <= 512, one query tile per call): half the bytes of the
// bfloat16 scan copy, exact integer accumulation, and an error bound that is MEASURED per row instead of assumed.
//
// The bf16 filter (sq_dense_scan.hpp) streams 2 d_pad + 4 bytes per row and is HBM bound: a pass over 10 M x 128 is 2.6 GB
// and 0.39 ms at 0.82 of the peak -- the kernel is where the hardware lets it be, so the only way on is fewer bytes.  Here a
// row is d_pad signed bytes: x' = x - c (the filter's origin, as before) quantised with ONE scale for the whole matrix,
//     x8_k = clamp(rint(x'_k / Dx), -127, 127),      Dx = clamp / 127, the clamp chosen from the data (below),
// and the part a byte cannot hold is not bounded by a worst case but measured at build time, in float64, per row:
//     r_row = | x' - Dx x8 |_2        (quantisation noise ~ Dx sqrt(d / 12), plus whatever the clamp cut off).
// With the query scaled per query, Q8_k = rint(-2 q''_k / Dq), q'' = q - c, rq = |-2 q'' - Dq Q8|_2 measured the same
// way, the kernel's score
//     s~ = N_row + (Dx Dq) * sum_k x8_k Q8_k             (v_mfma_i32_32x32x32_i8: the sum is an exact integer;
//                                                          Q8 is two int8 planes, the second in units of Dq / 256:
//                                                          256 * sum + sum' is one shift-add of the two accumulators)
// differs from the true score s = |x'|^2 - 2 x'.q'' by at most
//     e(row, q) = 2 r_row |q''| + |Dx x8| rq + rounding <= 2 R |q''| + (X + R) rq + rounding =: e_q
// (Cauchy-Schwarz on the two measured residuals; R = the largest r_row of the rows that take part, X = the largest
// |x'|).  Rows whose r_row is far above the rest (an element far beyond the clamp) would widen every query's slack:
// they get N_row = -inf instead, pass every threshold and are simply re-ranked exactly (the clamp and R are chosen so
// that a few hundred rows at most end up there; data no clamp suits keeps the bf16 filter).  Everything downstream is the bf16 filter's: the sampled k-th score T_s bounds
// the true k-th score by T_s + e_q, the threshold is T' = T_s + 2 e_q, survivors leave as (first row, mask, query)
// entries of per-wave segments, are re-ranked in the reference's float32 arithmetic from the ORIGINAL rows
// (dense_rerank_l2_kernel), selected, and certified: a non-candidate has s > T' - e_q.  Queries that fail take the
// middle tier and the exact path as before, so results never depend on the filter.  Cosine: x' = x / |x|, N_row = 0, the
// planes hold -q / |q|, e_q = R + (1 + R) rq + rounding, float64 re-rank (dense_rerank_cos_kernel).
//
// Layout: the copy is plain row-major int8 [n_pad][128 | 256 | 512]; for 128-byte rows a ring unit is 64 rows (8 KiB: two
// 32-row MFMA tiles) DMA'd as 8 pieces of 8 rows, the 16-byte chunks of a row XOR-swizzled by (row >> 1) & 7 on the SOURCE
// side so that the ds_read_b128 of a fragment (lane = row, 16 consecutive k) is conflict-free (wider rows: I8Geom, i8_swz).  One 16-byte read feeds one MFMA
// (K = 32) per query plane: four reads and eight MFMAs per tile, against eight and sixteen in the bf16 kernel.
#pragma once
#include "sq_dense_scan.hpp"

namespace sq {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

// Geometry for KS = d_pad8 / 32 k-steps per row (d_pad8 = 128, 256 or 512 bytes per row): a ring unit is 64 rows of 128
// bytes or 32 rows of 256 / 512 bytes (8, 8, 16 KiB) + 256 B for its row terms; eight waves per workgroup, four for 512-byte rows
// (two 16.25 KiB slots per wave: 130 KiB of LDS either way, and the 128 query-plane registers of a 512-byte row
// want a SIMD to themselves).
template <int KS>
struct I8Geom {
    static constexpr int ROW_BYTES = KS * 32;
    static constexpr int UNIT_ROWS = KS == 4 ? 64 : 32;
    static constexpr int TILES = UNIT_ROWS / 32;
    static constexpr int UNIT_BYTES = UNIT_ROWS * ROW_BYTES;
    static constexpr int SLOT_BYTES = UNIT_BYTES + 256;
    static constexpr int WAVES = KS <= 8 ? 8 : 4;
    static constexpr int NSTAGE = 2;
    static constexpr int PIECES = UNIT_BYTES / 1024;   // DMA instructions per unit (+ 1 for the row terms)
    static constexpr int SAMPLES_PER_UNIT = 2 * TILES;
};
static constexpr int I8_MAX_ROW_BYTES = 512;
__host__ __device__ constexpr int i8_row_bytes(int d) { return d <= 128 ? 128 : (d <= 256 ? 256 : 512); }
__host__ __device__ constexpr int i8_unit_rows(int row_bytes) { return row_bytes == 128 ? 64 : 32; }

// ---------------------------------------------------------------- build
// The value the copy quantises: L2 the row minus the filter's origin; cosine the row scaled to unit length, rounded to
// float32 -- the residuals are measured against this very value; what it differs from the exact x / |x| by (2^-24 per
// element, relative) is part of the rounding term of e_q.
__device__ __forceinline__ float i8_element(float v, const float* __restrict__ center, int k, bool cosine, double inv_norm) {
    if (cosine) return (float)((double)v * inv_norm);
    return center ? __fsub_rn(v, center[k]) : v;
}

// sum over rows of |x - c|^2 (float64) and their number, rows with |x - c|^2 > cap left out: the element rms the clamp
// candidates are multiples of.  The host runs it three times, cap = inf, then 16 x the mean of the pass before: a few
// rows thousands of times the size of the rest would otherwise own the rms (they end up beyond R either way).
template <int EPL>   // elements per lane: row_bytes / 64
static __global__ __launch_bounds__(256) void dense8_energy_kernel(const float* __restrict__ db, long long n, long long ld, int d,
                                                                    const float* __restrict__ center, const double* __restrict__ nx64,
                                                                    double cap, double* __restrict__ sum) {   // [0]: energy, [1]: rows
    __shared__ double red[4][2];
    const int lane = threadIdx.x & 63;
    const long long wave0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long long)gridDim.x * 4;
    double acc = 0.0, rows = 0.0;
    for (long long row = wave0; row < n; row += nw) {
        double e = 0.0;
        bool bad = false;
        const double inv_norm = nx64 ? 1.0 / sqrt(nx64[row]) : 1.0;
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const int k = EPL * lane + j;
            if (k < d) {
                const float v = db[row * ld + k];
                const float xc = i8_element(v, center, k, nx64 != nullptr, inv_norm);
                if (!(fabsf(xc) < 3.0e38f)) bad = true;
                e += (double)xc * (double)xc;
            }
        }
        if (__ballot(bad) != 0ull) continue;
        for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o);
        if (e <= cap) {
            acc += e;
            rows += 1.0;
        }
    }
    if (lane == 0) {
        red[threadIdx.x >> 6][0] = acc;
        red[threadIdx.x >> 6][1] = rows;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(sum, red[0][0] + red[1][0] + red[2][0] + red[3][0]);
        atomicAdd(sum + 1, red[0][1] + red[1][1] + red[2][1] + red[3][1]);
    }
}

// The clamp is CHOSEN from the data: for each of I8_NCLIP candidate clamps (multiples of the element rms) the row
// residuals r_row^2 = |x' - Dx x8|^2 are evaluated (float32 is enough for a choice; the copy's own residuals are measured
// in float64 by the build kernel) and the rows beyond each of I8_NCUT candidate bounds counted.  The bounds are multiples
// of what rounding alone leaves, Dx sqrt(d / 12) -- not of the measured mean, which a few wild rows would own.  A narrow
// clamp has the finer step but cuts more elements off; the host takes the pair with the least bound R that leaves no
// more than a few hundred rows beyond it.
static constexpr int I8_NCLIP = 12, I8_NCUT = 10;
struct Dense8ClipArgs {
    float inv_dx[I8_NCLIP], dx[I8_NCLIP];
    float cut[I8_NCLIP][I8_NCUT];   // r_row^2 thresholds
};

template <int EPL>
static __global__ __launch_bounds__(256) void dense8_clip_stats_kernel(const float* __restrict__ db, long long n, long long ld, int d,
                                                                        const float* __restrict__ center, const double* __restrict__ nx64,
                                                                        Dense8ClipArgs ca, u32* __restrict__ counts, int row_step) {
    const int lane = threadIdx.x & 63;
    // (row_step > 1: every row_step-th row -- the counts only CHOOSE the clamp; the copy's own residuals are measured row by
    // row by the build kernel and rows beyond the chosen bound are flagged from those)
    const long long wave0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * row_step, nw = (long long)gridDim.x * 4 * row_step;
    u32 cnt[I8_NCLIP][I8_NCUT];
#pragma unroll
    for (int c = 0; c < I8_NCLIP; ++c)
#pragma unroll
        for (int m = 0; m < I8_NCUT; ++m) cnt[c][m] = 0u;
    for (long long row = wave0; row < n; row += nw) {
        float r2[I8_NCLIP];
#pragma unroll
        for (int c = 0; c < I8_NCLIP; ++c) r2[c] = 0.f;
        bool bad = false;
        const double inv_norm = nx64 ? 1.0 / sqrt(nx64[row]) : 1.0;
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const int k = EPL * lane + j;
            if (k < d) {
                const float v = db[row * ld + k];
                const float xc = i8_element(v, center, k, nx64 != nullptr, inv_norm);
                if (!(fabsf(xc) < 3.0e38f)) bad = true;
#pragma unroll
                for (int c = 0; c < I8_NCLIP; ++c) {
                    const float t = fminf(fmaxf(rintf(xc * ca.inv_dx[c]), -127.f), 127.f);
                    const float res = __fmaf_rn(-t, ca.dx[c], xc);
                    r2[c] = __fmaf_rn(res, res, r2[c]);
                }
            }
        }
        if (__ballot(bad) != 0ull) continue;
#pragma unroll
        for (int c = 0; c < I8_NCLIP; ++c) {
            for (int o = 32; o > 0; o >>= 1) r2[c] += __shfl_xor(r2[c], o);
#pragma unroll
            for (int m = 0; m < I8_NCUT; ++m) cnt[c][m] += r2[c] > ca.cut[c][m] ? 1u : 0u;
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < I8_NCLIP; ++c)
#pragma unroll
            for (int m = 0; m < I8_NCUT; ++m)
                if (cnt[c][m]) atomicAdd(&counts[c * I8_NCUT + m], cnt[c][m]);
    }
}

// One wave per row: the int8 row, N_row = RD(|x'|^2) and the measured residual r_row^2 (rounded up) of rows
// [row_base, n_pad).  Padding rows: zeros and N_row = +inf.  Rows with a non-finite element: N_row = +inf (their true
// distance is inf / NaN: they rank last, as in the bf16 filter) and no residual.
template <int EPL>
static __global__ __launch_bounds__(256) void dense8_build_kernel(const float* __restrict__ db, long long n, long long ld, int d,
                                                                   long long n_pad, const float* __restrict__ center,
                                                                   const double* __restrict__ nx64, float inv_dx, float dx,
                                                                   signed char* __restrict__ out8, float* __restrict__ nrow,
                                                                   float* __restrict__ r2row, long long row_base) {
    const int lane = threadIdx.x & 63;
    const long long row = row_base + (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_pad) return;
    unsigned char q[EPL];
#pragma unroll
    for (int j = 0; j < EPL; ++j) q[j] = 0;
    double e2 = 0.0, r2 = 0.0;
    bool finite = true;
    if (row < n) {
        const double inv_norm = nx64 ? 1.0 / sqrt(nx64[row]) : 1.0;
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const int k = EPL * lane + j;
            if (k < d) {
                const float v = db[row * ld + k];
                const float xc = i8_element(v, center, k, nx64 != nullptr, inv_norm);
                if (!(xc == xc) || !(fabsf(xc) < 3.0e38f)) finite = false;
                float t = rintf(xc * inv_dx);
                t = fminf(fmaxf(t, -127.f), 127.f);
                if (!(t == t)) t = 0.f;
                q[j] = (unsigned char)(signed char)(int)t;
                const double res = (double)xc - (double)t * (double)dx;
                e2 += (double)xc * (double)xc;
                r2 += res * res;
            }
        }
    }
    finite = __ballot(!finite) == 0ull;
    for (int o = 32; o > 0; o >>= 1) {
        e2 += __shfl_xor(e2, o);
        r2 += __shfl_xor(r2, o);
    }
    // EPL bytes per lane, packed: the row is EPL * 64 bytes
    unsigned char* dst = reinterpret_cast<unsigned char*>(out8) + row * (EPL * 64) + EPL * lane;
    if constexpr (EPL == 2) {
        *reinterpret_cast<unsigned short*>(dst) = (unsigned short)(q[0] | (q[1] << 8));
    } else if constexpr (EPL == 4) {
        *reinterpret_cast<u32*>(dst) = (u32)q[0] | ((u32)q[1] << 8) | ((u32)q[2] << 16) | ((u32)q[3] << 24);
    } else {
        uint2 w;
        w.x = (u32)q[0] | ((u32)q[1] << 8) | ((u32)q[2] << 16) | ((u32)q[3] << 24);
        w.y = (u32)q[4] | ((u32)q[5] << 8) | ((u32)q[6] << 16) | ((u32)q[7] << 24);
        *reinterpret_cast<uint2*>(dst) = w;
    }
    if (lane == 0) {
        float nr = __builtin_inff(), rr = 0.f;
        if (row < n && finite) {
            nr = nx64 ? 0.f : (float)e2;   // (cosine: the score is -x^.q^ alone)
            if ((double)nr > e2) nr = __uint_as_float(__float_as_uint(nr) - 1u);   // round down (e2 >= 0)
            rr = (float)r2;
            if ((double)rr < r2) rr = __uint_as_float(__float_as_uint(rr) + 1u);   // round up
        }
        nrow[row] = nr;
        r2row[row] = rr;
    }
}

// sum and maximum of the measured residuals r_row^2 and the largest N_row over the rows the bound covers (finite, not
// flagged as always-candidates: run after dense8_flag_kernel), for R and X
static __global__ __launch_bounds__(256) void dense8_resid_stats_kernel(const float* __restrict__ r2row, const float* __restrict__ nrow,
                                                                         long long n, double* __restrict__ sum_r2,
                                                                         u32* __restrict__ max_bits) {   // [0]: max r2, [1]: max N
    __shared__ double red[4];
    __shared__ float rmax[4], nmax[4];
    double acc = 0.0;
    float m = 0.f, mn = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float nv = nrow[i];
        if (nv < __builtin_inff() && nv > -__builtin_inff()) {   // (rows that take part in the bound: not padding, not always-candidates)
            const float v = r2row[i];
            acc += (double)v;
            m = fmaxf(m, v);
            mn = fmaxf(mn, nv);
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        acc += __shfl_xor(acc, o);
        m = fmaxf(m, __shfl_xor(m, o));
        mn = fmaxf(mn, __shfl_xor(mn, o));
    }
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = acc;
        rmax[threadIdx.x >> 6] = m;
        nmax[threadIdx.x >> 6] = mn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(sum_r2, red[0] + red[1] + red[2] + red[3]);
        atomicMax(max_bits, __float_as_uint(fmaxf(fmaxf(rmax[0], rmax[1]), fmaxf(rmax[2], rmax[3]))));   // non-negative floats order like their bits
        atomicMax(max_bits + 1, __float_as_uint(fmaxf(fmaxf(nmax[0], nmax[1]), fmaxf(nmax[2], nmax[3]))));
    }
}

// rows whose residual is above the cut take part as "always a candidate": N_row = -inf
static __global__ __launch_bounds__(256) void dense8_flag_kernel(const float* __restrict__ r2row, float* __restrict__ nrow, long long n,
                                                                  float r2_cut, u32* __restrict__ flagged, long long row_base) {
    const long long i = row_base + (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (nrow[i] < __builtin_inff() && r2row[i] > r2_cut) {
        nrow[i] = -__builtin_inff();
        atomicAdd(flagged, 1u);
    }
}

// ---------------------------------------------------------------- per call
// Query prep of the int8 filter: the int8 planes Q8 = rint(-2 (q - c) / Dq) with the query's own scale and
// Q8' = rint(256 (-2 (q - c) / Dq - Q8)) (the matrix cores idle under the stream: a second plane is free and takes the
// query's own quantisation out of the bound: rq shrinks ~250-fold), the measured
// residual rq, |q - c|^2, the score unit Dx Dq and the query's error bound e_q (all float64, rounded up where they
// widen the bound), plus what dense_prep_queries_kernel does besides (counters, overflow flag, the aligned copy).
//   per query p < nq_pad (= plane_rows):  qs8[plane][p][row bytes] int8 (plane 1 in units of Dq / 256), par[p] = {unit (Dx Dq), e_q}
// Sixteen lanes per query -- a wave prepares FOUR queries at once (row = lane >> 4): lane l of a row owns plane bytes
// 64 c + l + 16 e (c: 64-byte chunk of the row, e = 0 .. 3).  The stand-alone kernel below runs it with one wave per four
// queries, the fused head of a call (dense8_head_kernel) in every workgroup.  The float64 sums are taken per 64-element
// chunk as the butterfly (t, t ^ 32), (.., ^ 16), ... (^ 1) and then in chunk order: every caller gets the same bits.
//   p0 / p1: this query's rows of the two planes (global or LDS); {unit, e_q} and |q''|^2 come back in every lane of the row.
template <int EPL>   // row bytes / 64
__device__ __forceinline__ void dense8_prep_query_row(const float* __restrict__ qrow, bool real, int d, const float* __restrict__ center,
                                                      double dx, double r_max, double x_max, int cosine,
                                                      signed char* __restrict__ p0, signed char* __restrict__ p1, float2& par_out,
                                                      double& qn2_out) {
    const int l16 = threadIdx.x & 15;
    // L2: q'' = q - c and the planes hold -2 q''; cosine: q'' = q / |q| (float64 norm, rounded to float32) and the planes hold -q''
    const float sc = cosine ? -1.f : -2.f;
    float v[EPL][4];
#pragma unroll
    for (int c = 0; c < EPL; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int t = 64 * c + l16 + 16 * e;
            float raw = 0.f, cen = 0.f;
            if (real && t < d) {
                raw = qrow[t];
                if (center) cen = center[t];
            }
            v[c][e] = (real && t < d) ? (center ? __fsub_rn(raw, cen) : raw) : 0.f;
        }
    // butterfly sum of a chunk's 64 values: pairs (t, t ^ 32) and (.., ^ 16) are this lane's own four, the rest across the row
    auto chunk_sum = [](double x0, double x1, double x2, double x3) __attribute__((always_inline)) {
        double sacc = (x0 + x2) + (x1 + x3);
        for (int o = 8; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o);
        return sacc;
    };
    if (cosine) {
        double nn = 0.0;
#pragma unroll
        for (int c = 0; c < EPL; ++c)
            nn += chunk_sum((double)v[c][0] * (double)v[c][0], (double)v[c][1] * (double)v[c][1], (double)v[c][2] * (double)v[c][2],
                            (double)v[c][3] * (double)v[c][3]);
        const double rn = sqrt(nn);
#pragma unroll
        for (int c = 0; c < EPL; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[c][e] = (float)((double)v[c][e] / rn);   // (a zero query: NaN -- no scale, the exact path answers it as the reference does)
    }
    // |q''|^2 and the largest plane element
    double Q = 0.0;
    float mx = 0.f;
#pragma unroll
    for (int c = 0; c < EPL; ++c) {
        Q += chunk_sum((double)v[c][0] * (double)v[c][0], (double)v[c][1] * (double)v[c][1], (double)v[c][2] * (double)v[c][2],
                       (double)v[c][3] * (double)v[c][3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float m = fabsf(sc * v[c][e]);
            if (!(m == m)) m = __builtin_inff();
            mx = fmaxf(mx, m);
        }
    }
    for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const bool ok = real && mx < 3.0e38f && mx > 0.f;   // zero / non-finite / padding queries: an all-zero plane, nothing certified by it
    const double dq = ok ? (double)mx / 127.0 : 1.0;
    const double inv_dq = 1.0 / dq;
    double r2s = 0.0;
#pragma unroll
    for (int c = 0; c < EPL; ++c) {
        double r2[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const double w = (double)(sc * v[c][e]);
            const double ws = w * inv_dq;   // the plane value in steps of Dq (whatever its last bit, the residual below is measured against what is stored)
            float qt8 = 0.f, ql8 = 0.f;
            if (ok) {
                qt8 = fminf(fmaxf(rintf((float)ws), -127.f), 127.f);
                // second plane: what the first leaves, in steps of Dq / 256 (the kernel joins the two integer sums by a shift)
                ql8 = fminf(fmaxf(rintf((float)((ws - (double)qt8) * 256.0)), -127.f), 127.f);
            }
            const int t = 64 * c + l16 + 16 * e;
            p0[t] = (signed char)(int)qt8;
            p1[t] = (signed char)(int)ql8;
            const double res = w - ((double)qt8 + (double)ql8 * 0.00390625) * dq;
            r2[e] = res * res;
        }
        r2s += chunk_sum(r2[0], r2[1], r2[2], r2[3]);
    }
    const double rq = sqrt(r2s) * (1.0 + 1e-9);
    const double unit = dx * dq;
    // e_q: R |w| (the rows' measured residual; w = -2 q'' or -q'') + (X + R) rq (the query's) + the float32 evaluation of
    // N + unit * sum (|sum| unit <= (X + R)(2 |q''| + rq): conversion, unit and fma roundings) + N's own rounding
    const double xr = x_max + r_max, qn = sqrt(Q), wn = (cosine ? 1.0 : 2.0) * qn;   // wn: the length of what the planes hold
    double e = r_max * wn + xr * rq + 4.0 * 5.9604644775390625e-08 * (xr * (wn + rq) + x_max * x_max);
    e *= 1.0 + 1e-6;
    float ef = (float)e;
    if ((double)ef < e) ef = __uint_as_float(__float_as_uint(ef) + 1u);
    if (!ok && real) ef = __builtin_inff();   // (a zero or non-finite query: Dense8ThrPost gives it a NaN threshold, it takes the next tier)
    par_out = make_float2((float)unit, real ? ef : 0.f);
    qn2_out = real ? Q : 0.0;
}

// (one 64-thread workgroup per four queries of the padded tile)
template <int EPL>
static __global__ __launch_bounds__(64) void dense8_prep_queries_kernel(const float* __restrict__ q, int nq, int d,
                                                                         const float* __restrict__ center, double dx, double r_max,
                                                                         double x_max, signed char* __restrict__ qs8,
                                                                         float2* __restrict__ par, double* __restrict__ qn2,
                                                                         float* __restrict__ thr, u32* __restrict__ cnt,
                                                                         u32* __restrict__ oflag, float* __restrict__ q_al, int ldq,
                                                                         const DenseCallPtrs* __restrict__ ind, int cosine, int plane_rows) {
    constexpr int row_bytes = EPL * 64;
    if (ind) q = ind->q;   // (captured call graph: this launch's queries)
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 4), t = threadIdx.x & 15;
    if (t == 0) {
        // padding queries of the tile: a NaN threshold -- no comparison passes, not even an always-candidate row's -inf
        thr[qi] = qi < nq ? -__builtin_inff() : __builtin_nanf("");
        cnt[qi] = 0u;
        if (qi == 0) *oflag = 0u;
    }
    if (qi < nq)
        for (int i = t; i < ldq; i += 16) q_al[(long long)qi * ldq + i] = i < d ? q[(long long)qi * d + i] : 0.f;
    float2 pr;
    double Q;
    dense8_prep_query_row<EPL>(q + (long long)qi * d, qi < nq, d, center, dx, r_max, x_max, cosine,
                               qs8 + (long long)qi * row_bytes, qs8 + (long long)(plane_rows + qi) * row_bytes, pr, Q);
    if (t == 0) {
        par[qi] = pr;
        qn2[qi] = Q;
    }
}

// T_s (the sampled k-th score) -> T' = T_s + 2 e_q: at least k rows have a true score <= T_s + e_q, and a row
// with a true score <= T_s + e_q has a kernel score <= T_s + 2 e_q.  (+ the rounding of the float32 distance the
// certification compares with.)
struct Dense8ThrPost {
    const float2* par;
    const double* qn2;
    __device__ __forceinline__ void prologue(int, double*) const {}
    __device__ __forceinline__ float operator()(int q, float t) const {
        if (!(t < __builtin_inff())) return t;
        if (!(par[q].y < __builtin_inff())) return __builtin_nanf("");   // nothing certifiable: no candidates, the next tier takes it
        const double tt = (double)t + 2.0 * (double)par[q].y + 4e-6 * fabs((double)t + qn2[q]);
        float r = (float)tt;
        if ((double)r < tt) r = __uint_as_float(__float_as_uint(r) + (r >= 0.f ? 1 : -1));
        return r;
    }
};

struct Dense8ScanArgs {
    const signed char* scan8;   // [n_pad][row bytes]
    const float* nrow;          // [n_pad64 + 64] N_row (+inf padding, -inf always-candidate rows)
    long long n;
    long long n_units;          // ceil(n / unit rows)
    const signed char* qs8;     // [2][32][row bytes]: plane, query
    const float2* par;          // [32] {unit, e_q}
    const float* thr;           // [32]
    uint2* wave_out;
    u32* wave_cnt;
    u32 wave_cap;
    float* sample_out;          // [32][ns]
    long long ns;
    long long unit_step;        // SAMPLE: every unit_step-th unit; EMIT: 1
    long long n_sel;            // units this launch visits
    int nrb;
    int nt;                     // non-temporal stream beyond nt_from_row
    long long nt_from_row;
    // batches of 33 .. 256 queries (dense8_scan_mt_kernel): groups of QT query tiles, workgroup -> (group, row block)
    int nqt;                    // groups
    int plane_rows;             // rows of one query plane: qs8 is [2][plane_rows][128]
    int debug;                  // measurement ablations ("dense_debug"; results are garbage): 1024 = no score epilogue
    // the tightened threshold (dense8_body_kernel): per entry its smallest score, per query a histogram of those
    float* wave_score;          // parallel to wave_out
    u32* hist;                  // [32][I8_HIST_BINS] + [32] ordered keys of the largest tightened threshold applied
};

// LDS chunk position (16-byte units inside a row) of source chunk c of row r: the XOR swizzle that makes the fragment reads
// of 16 consecutive rows hit 16 different bank groups.  128-byte rows: two rows share 256 bytes, (r >> 1) & 7 over the row's
// eight chunks; wider rows: r & 15 inside each 256-byte plane.  An involution: the DMA applies it to the SOURCE chunk.
template <int KS>
__device__ __forceinline__ int i8_swz(int c, int r) {
    if constexpr (KS == 4) return c ^ ((r >> 1) & 7);
    return (c & ~15) | ((c & 15) ^ (r & 15));
}

// What a wave does with the scores of a tile: I8_EMIT compares them with the query's threshold and appends survivors to its
// segment; I8_SAMPLE writes the minimum of each lane's 16 rows to the sample matrix (the six-launch chain's sample pass);
// I8_LANEMIN keeps ONE running minimum per lane over the whole pass (the fused head, dense8_head_kernel).
// I8_EMIT_H is I8_EMIT that also keeps each entry's smallest score and counts it in the query's histogram (below).
enum { I8_EMIT = 0, I8_SAMPLE = 1, I8_LANEMIN = 2, I8_EMIT_H = 3 };

// The tightened threshold.  The sampled threshold T' = T_s + 2 e_q is the k-th score of a SAMPLE plus the slack: at a
// stride of 14 units it lets ~5 k rows per query through at 10 M x 128 where ~300 lie below the k-th score of ALL rows plus the
// same slack -- and every one of them is a 512-byte gather for the re-rank.  The full pass knows better by the time it ends:
// every entry it emits adds its smallest score m to a per-query histogram of I8_HIST_BINS bins of width w = e_q / 2 below T'
// (bin j: T' - (j + 1) w < m <= T' - j w, the last bin open below; one agent-scope atomic without return per entry, ~60 per
// wave and pass).  A workgroup whose waves have drained the stream reads the histogram as it stands: if the bins j and
// beyond hold k entries, k different rows score at most T' - j w =: T_8, so the true k-th score is at most T_8 + e_q and a
// row scoring above T'' = T_8 + 2 e_q is not among the k nearest: the tail re-ranks only entries with m <= T''.  The
// histogram a workgroup sees may lack what the others have not emitted yet -- fewer counts only make T_8 larger: every
// workgroup's T'' is valid on its own, and the select certifies against the largest one applied (kept as an ordered key beside
// the histogram).  Always-candidate rows (score -inf) are not counted (their true score is unknown) and always pass.
// With the second level in place the sample only has to keep the first-level entries inside the segments: it can be
// several times sparser (host: the fused call's stride).
static constexpr int I8_HIST_BINS = 64;
// Per-query words every workgroup of a call updates at about the same time -- the candidate counters the re-rank reserves
// its key ranges on and the tightened-threshold keys -- sit a cache line apart: 256 workgroups x 32 queries of atomics on ONE
// line serialise in one memory channel (the tail's re-rank phase read 37 us on a 1.25 M-row shard, where every workgroup
// arrives at once, against 9 us of gathers).
static constexpr int I8_CNT_SHIFT = 5;
static constexpr int I8_HIST_WORDS = TILE_ROWS * I8_HIST_BINS + (TILE_ROWS << I8_CNT_SHIFT);   // u32 words per call slot

// A wave moves its share of the workgroup's LDS histogram (TILE_ROWS * I8_HIST_BINS words / WAVES) to the global one: the
// bins are taken with an exchange, so entries the other waves add meanwhile are simply moved by the next flush.  One
// global atomic per non-empty bin: a hot bin of the global histogram is added to once per workgroup and flush, not once
// per entry (per-entry global atomics on the few bins just under T' made the pass 0.30 instead of 0.23 ms: same-address
// atomics serialise, and the ring's counted vmcnt waits queue behind them).
template <int WAVES>
__device__ __forceinline__ void dense8_hist_flush(u32* lds_hist, u32* __restrict__ ghist) {
    constexpr int PER = TILE_ROWS * I8_HIST_BINS / WAVES;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
    for (int i = 0; i < PER / 64; ++i) {
        const int b = wave * PER + i * 64 + lane;
        const u32 v = __hip_atomic_exchange(lds_hist + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (v) __hip_atomic_fetch_add(ghist + b, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The stream of one wave: its share of the launch's ring units through its private LDS ring, the MFMAs of the unit's tiles
// against the query planes in registers, the score epilogue.  `wcount`: survivors appended (I8_EMIT); `smin`: the lane's
// running minimum (I8_LANEMIN).
template <int KS, int MODE>
__device__ __forceinline__ void dense8_stream(const Dense8ScanArgs& a, unsigned char* smem, const i32x4 (&bq)[KS], const i32x4 (&bl)[KS],
                                              float unit_lo, float thr_l, u32& wcount, float& smin, u32* lds_ticket,
                                              float inv_w_l = 0.f, u32* lds_hist = nullptr) {
    using G = I8Geom<KS>;
    constexpr bool SAMPLE = MODE == I8_SAMPLE || MODE == I8_LANEMIN;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r31 = lane & 31, h = lane >> 5;
    const u32 lds_base = (u32)(uintptr_t)smem;
    const u32 ring_base = lds_base + (u32)wave * (G::NSTAGE * G::SLOT_BYTES);
    const unsigned char* ring_ptr = smem + wave * (G::NSTAGE * G::SLOT_BYTES);
    const long long wave_id = (long long)blockIdx.x * G::WAVES + wave;
    const long long nwaves = (long long)a.nrb * G::WAVES;
    uint2* wout = a.wave_out + wave_id * a.wave_cap;
    float* wsc = MODE == I8_EMIT_H ? a.wave_score + wave_id * a.wave_cap : nullptr;
    u32* hist_l = MODE == I8_EMIT_H ? lds_hist + r31 * I8_HIST_BINS : nullptr;   // (the workgroup's LDS copy: dense8_hist_flush)

    // Units are handed out by a ticket counter of the WORKGROUP (LDS, zeroed by the caller before a barrier): ticket t is unit
    // (first wave of the workgroup + t % WAVES) + (t / WAVES) * nwaves of the launch -- the same units the workgroup's waves
    // would own under a fixed split (wave w: w + it * nwaves), but a wave that runs ahead takes the next one instead of
    // waiting at the end: with the fixed split the eight waves of a workgroup left the stream 20 - 48 us apart (of 200) and
    // the tail -- and the CU -- waited for the last.
    const long long wg_first = (long long)blockIdx.x * G::WAVES;
    auto take_unit = [&]() __attribute__((always_inline)) -> long long {
        u32 t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(lds_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        t = (u32)__builtin_amdgcn_readfirstlane((int)t);
        const long long u = wg_first + (long long)(t % G::WAVES) + (long long)(t / G::WAVES) * nwaves;
        return u < a.n_sel ? u : -1ll;   // (u grows with t: the first ticket beyond the launch's units ends the wave's stream)
    };
    const long long my_units = wave_id < a.n_sel ? (a.n_sel - wave_id + nwaves - 1) / nwaves : 0;   // (the fixed split's share: sizes the flush period)
    // DMA piece j: bytes 1024 j .. 1024 j + 1023 of the unit's LDS image; lane -> 16 bytes at (row, chunk position), read
    // from the row's source chunk swz(position)
    u32 voff[G::PIECES];
#pragma unroll
    for (int j = 0; j < G::PIECES; ++j) {
        const int lin = j * 1024 + lane * 16;
        const int r = lin / G::ROW_BYTES, cpos = (lin % G::ROW_BYTES) >> 4;
        voff[j] = (u32)(r * G::ROW_BYTES + i8_swz<KS>(cpos, r) * 16);
    }
    const u32 voff_n = (u32)lane * 4u;
    static_assert(G::NSTAGE == 2, "two ring slots: the units in flight are sel_even / sel_odd");
    long long issued = 0, sel_even = -1, sel_odd = -1;   // the selected-unit index in ring slot 0 / 1
    bool more = true;
    auto issue_next = [&]() __attribute__((always_inline)) {
        if (!more) return;
        const long long sel = take_unit();
        if (sel < 0) {
            more = false;
            return;
        }
        if (issued & 1) sel_odd = sel; else sel_even = sel;
        const long long unit_idx = sel * a.unit_step;
        const long long row0 = unit_idx * G::UNIT_ROWS;
        const u32 dst = ring_base + (u32)(issued % G::NSTAGE) * G::SLOT_BYTES;
        const unsigned char* base = reinterpret_cast<const unsigned char*>(a.scan8) + row0 * G::ROW_BYTES;
#pragma unroll
        for (int j = 0; j < G::PIECES; ++j) {
            if (a.nt && row0 >= a.nt_from_row)
                glds16<true>(base, voff[j], dst + (u32)j * 1024);
            else
                glds16<false>(base, voff[j], dst + (u32)j * 1024);
        }
        glds4(a.nrow + row0, voff_n, dst + G::UNIT_BYTES);   // (64 floats: the unit's, and for 32-row units the next one's as well)
        ++issued;
    };
    for (int p = 0; p < G::NSTAGE; ++p) issue_next();
    // I8_EMIT_H: this wave's share of the workgroup's histogram goes to the global one eight times per pass (and once more
    // from the tail): a workgroup that finishes early then finds at least 7/8 of everybody's entries
    const int flush_every = (int)(my_units >> 3) > 0 ? (int)(my_units >> 3) : 1;
    int to_flush = flush_every;

    for (long long it = 0; it < issued; ++it) {   // (`issued` grows inside: issue_next)
        const long long sel = (it & 1) ? sel_odd : sel_even;
        const long long unit_idx = sel * a.unit_step;
        const long long row0 = unit_idx * G::UNIT_ROWS;
        wait_units_in_flight<G::NSTAGE, G::PIECES + 1>((int)(issued - it - 1));
        const unsigned char* sl = ring_ptr + (it % G::NSTAGE) * G::SLOT_BYTES;
        // A fragments (lane = row): all of the unit's for rows up to 256 bytes; a 512-byte row in two halves of eight
        // k-steps, the second read while the first half's MFMAs run (its slot is refilled after the second read)
        constexpr int KH = KS == 16 ? 8 : KS;      // k-steps per fragment batch
        i32x4 av[G::TILES][KH];
        f32x4 nr[G::TILES][4];
#pragma unroll
        for (int t = 0; t < G::TILES; ++t) {
            const int r = 32 * t + r31;
#pragma unroll
            for (int s = 0; s < KH; ++s)
                av[t][s] = *reinterpret_cast<const i32x4*>(sl + r * G::ROW_BYTES + i8_swz<KS>(2 * s + h, r) * 16);
            // N of the 16 rows this lane's accumulator registers hold: rows (i & 3) + 8 (i >> 2) + 4 h of tile t
#pragma unroll
            for (int c = 0; c < 4; ++c) nr[t][c] = *reinterpret_cast<const f32x4*>(sl + G::UNIT_BYTES + (32 * t + 8 * c + 4 * h) * 4);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (KS != 16) issue_next();   // the unit is in registers: its slot is free
#pragma unroll
        for (int t = 0; t < G::TILES; ++t) {
            i32x16 acc, acl;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = acl[i] = 0;
#pragma unroll
            for (int s = 0; s < KH; ++s) {
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[t][s], bq[s], acc, 0, 0, 0);
                acl = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[t][s], bl[s], acl, 0, 0, 0);
            }
            if constexpr (KS == 16) {
#pragma unroll
                for (int s = 0; s < KH; ++s)
                    av[t][s] = *reinterpret_cast<const i32x4*>(sl + r31 * G::ROW_BYTES + i8_swz<KS>(2 * (KH + s) + h, r31) * 16);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                issue_next();
#pragma unroll
                for (int s = 0; s < KH; ++s) {
                    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[t][s], bq[KH + s], acc, 0, 0, 0);
                    acl = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[t][s], bl[KH + s], acl, 0, 0, 0);
                }
            }
            if (a.debug & 1024) {   // ablation: the stream and the MFMAs alone
                if (acc[0] == 0x7fffffff && acl[5] == 0x7ffffffe) wcount += 1;
                continue;
            }
            // scores of 32 rows x 32 queries (lane = query, register i = row (i & 3) + 8 (i >> 2) + 4 h)
            float sc[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float nv = nr[t][i >> 2][i & 3];
                if constexpr (SAMPLE) nv = nv == -__builtin_inff() ? __builtin_inff() : nv;   // an always-candidate row is no sample
                // 256 acc + acl: |acc| <= 512 * 127 * 127, so the sum stays below 2^31; its float32 conversion is one of the
                // roundings e_q pays for
                sc[i] = __fmaf_rn((float)((acc[i] << 8) + acl[i]), unit_lo, nv);
            }
            float m = sc[0];
#pragma unroll
            for (int i = 1; i < 16; ++i) m = fminf(m, sc[i]);
            if constexpr (MODE == I8_LANEMIN) {
                smin = fminf(smin, m);
            } else if constexpr (MODE == I8_SAMPLE) {
                a.sample_out[(long long)r31 * a.ns + sel * G::SAMPLES_PER_UNIT + t * 2 + h] = m;
            } else {
                const u64 hit = __ballot(m <= thr_l);
                if (hit != 0) {
                    u32 mask = 0;
#pragma unroll
                    for (int i = 0; i < 16; ++i) mask |= (sc[i] <= thr_l ? 1u : 0u) << i;   // (+inf padding rows never pass)
                    const u64 bal = __ballot(mask != 0);
                    if (mask) {
                        const u32 pos = wcount + __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u));
                        if (pos < a.wave_cap) {
                            wout[pos] = make_uint2((u32)(row0 + 32 * t + 4 * h), (mask << 16) | (u32)r31);
                            if constexpr (MODE == I8_EMIT_H) wsc[pos] = m;
                        }
                        if constexpr (MODE == I8_EMIT_H) {
                            if (m > -__builtin_inff()) {
                                const int bin = min((int)((thr_l - m) * inv_w_l), I8_HIST_BINS - 1);   // (m <= thr_l: never negative)
                                __hip_atomic_fetch_add(hist_l + bin, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                        }
                    }
                    wcount += (u32)__popcll(bal);
                }
            }
        }
        if constexpr (MODE == I8_EMIT_H) {
            if (--to_flush == 0) {
                to_flush = flush_every;
                dense8_hist_flush<G::WAVES>(lds_hist, a.hist);
            }
        }
    }
}

// this lane's query (column r31 of the tile): its int8 planes as B fragments (k = 32 s + 16 h ..) from `planes`
// ([2][32][row bytes], global or LDS), complete before the ring starts
template <int KS>
__device__ __forceinline__ void dense8_load_planes(const signed char* planes, i32x4 (&bq)[KS], i32x4 (&bl)[KS]) {
    using G = I8Geom<KS>;
    const int lane = threadIdx.x & 63, r31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        bq[s] = *reinterpret_cast<const i32x4*>(planes + r31 * G::ROW_BYTES + (2 * s + h) * 16);
        bl[s] = *reinterpret_cast<const i32x4*>(planes + (TILE_ROWS + r31) * G::ROW_BYTES + (2 * s + h) * 16);
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(bq[s]), "+v"(bl[s]));
}

template <int KS, bool SAMPLE>
__global__ __launch_bounds__(I8Geom<KS>::WAVES * 64, KS == 16 ? 1 : 2) void dense8_scan_kernel(Dense8ScanArgs a) {
    using G = I8Geom<KS>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r31 = lane & 31;
    i32x4 bq[KS], bl[KS];
    dense8_load_planes<KS>(a.qs8, bq, bl);
    float unit_lo = a.par[r31].x * 0.00390625f;   // Dx Dq / 256, exact
    float thr_l = SAMPLE ? 0.f : a.thr[r31];
    asm volatile("" : "+v"(unit_lo), "+v"(thr_l));
    u32 wcount = 0;
    float smin = 0.f;
    __shared__ u32 s_unit_ticket;
    if (threadIdx.x == 0) s_unit_ticket = 0u;
    __syncthreads();
    dense8_stream<KS, SAMPLE ? I8_SAMPLE : I8_EMIT>(a, smem, bq, bl, unit_lo, thr_l, wcount, smin, &s_unit_ticket);
    if constexpr (!SAMPLE) {
        if (lane == 0) {
            const long long wave_id = (long long)blockIdx.x * G::WAVES + wave;
            a.wave_cnt[2 * wave_id] = wcount;
            a.wave_cnt[2 * wave_id + 1] = 0u;
        }
    }
}

// ---------------------------------------------------------------- a call in three launches (one query tile)
// A pipelined call was six dependent launches -- query prep, sample pass, k-th threshold, full pass, re-rank, select -- and on
// a 1.25 M-row shard the pass itself is a third of that chain.  The fused form is three:
//   head  (dense8_head_kernel): every workgroup prepares the query tile for itself (the planes never leave the CU),
//         streams its share of the sampled units keeping ONE running minimum per lane, and the workgroup that finishes
//         last turns the M = workgroups x waves x 2 lane minima of each query into the threshold;
//   body  (dense8_body_kernel): the full pass, whose workgroups re-rank their own survivor segments when their waves have
//         drained the stream (the ring's LDS becomes the re-rank's);
//   select_topk_kernel with the finalisation, as before.
// The threshold from lane minima: a lane's minimum is the score of an actual sampled row and different lanes hold
// different rows, so the k-th smallest of ANY set of lane minima has >= k sampled rows at or below it -- an upper bound of
// the k-th smallest score of the sample, hence of all rows: what the bound needs.  It is looser than the k-th smallest
// SAMPLE only where several of the k best samples share a lane: -L ln(1 - k / L) samples instead of k for L lanes (105
// for k = 100, L = 1024; the host sizes the grid for L >= 4 k), or where a workgroup holds more of the k best minima
// than the `keep` it hands over (the host picks keep from k / workgroups).

struct Dense8HeadArgs {
    Dense8ScanArgs s;           // qs8 / par / thr are OUTPUTS here (written for the body by workgroup 0 / the last one)
    const float* q;             // the caller's queries [nq][d]
    int nq, d;
    const float* center;
    double dx, r_max, x_max;
    int cosine;
    double* qn2;
    u32* cnt;
    u32* oflag;                 // [0] overflow flag of the call, [1] the head's ticket counter (self-cleaning)
    float* q_al;
    int ldq;
    const DenseCallPtrs* ind;
    float* lane_min;            // [32][gridDim.x * keep]: per query, the `keep` smallest lane minima of every workgroup
    int keep;                   // 4, 8 or 16 (<= waves x 2)
    int kk;
};

// exact k-th smallest of each of NQ value lists vals[j * stride + 0 .. M) (M <= 64 VM) by one wave: MSB-first bisection
// on the order-preserving keys, the count of a step = popcounts of wave ballots (no cross-lane arithmetic), leading bits
// common to all keys skipped.  +inf when fewer than k values are finite.  A pass over a list is VM vector compares of four
// cycles each (the reason the workgroups hand over their few smallest minima and not all of them) and a step is a chain
// compare -> scalar popcount -> add -> select: the NQ lists are walked together so that one list's chain runs under the
// others' compares.
template <int VM, int NQ>
__device__ __forceinline__ void wave_kth_smallest(const float* __restrict__ vals, long long stride, int M, int k, float (&out)[NQ]) {
    const int lane = threadIdx.x & 63;
    constexpr u32 KINF = 0xff800000u;   // ordered_f32(+inf)
    u32 key[NQ][VM];
    u32 prefix[NQ];
    u32 xall = 0u;
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        u32 kmin = KINF, kmax = 0u;
#pragma unroll
        for (int j = 0; j < VM; ++j) {
            const int i = j * 64 + lane;
            u32 kv = ordered_f32(vals[n * stride + (i < M ? i : 0)]);
            kv = (i < M && kv < KINF) ? kv : KINF;   // (padding and NaN: as +inf)
            key[n][j] = kv;
            kmin = kv < kmin ? kv : kmin;
            kmax = kv > kmax ? kv : kmax;
        }
        for (int o = 32; o > 0; o >>= 1) {
            const u32 a = (u32)__shfl_xor((int)kmin, o), b = (u32)__shfl_xor((int)kmax, o);
            kmin = a < kmin ? a : kmin;
            kmax = b > kmax ? b : kmax;
        }
        prefix[n] = __builtin_amdgcn_readfirstlane(kmin);
        xall |= __builtin_amdgcn_readfirstlane(kmin ^ kmax);
    }
    if (xall != 0u) {
        const int top = 31 - __clz((int)xall);   // highest bit in which any list's keys differ
#pragma unroll
        for (int n = 0; n < NQ; ++n) prefix[n] &= ~((2u << top) - 1u);
        for (int b = top; b >= 0; --b) {
#pragma unroll
            for (int n = 0; n < NQ; ++n) {
                const u32 cand = prefix[n] | ((1u << b) - 1u);
                u32 c = 0;
#pragma unroll
                for (int j = 0; j < VM; ++j) c += (u32)__popcll(__ballot(key[n][j] <= cand));
                if (c < (u32)k) prefix[n] |= 1u << b;
            }
        }
    }
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        // (fewer than k values at all: the walk ends on the all-ones suffix, at or above +inf)
        float r = unordered_f32(prefix[n]);
        if (prefix[n] >= KINF) {
            u32 c = 0;
#pragma unroll
            for (int j = 0; j < VM; ++j) c += (u32)__popcll(__ballot(key[n][j] < KINF));
            if (c < (u32)k) r = __builtin_inff();
        }
        out[n] = r;
    }
}

template <int KS>
__global__ __launch_bounds__(I8Geom<KS>::WAVES * 64, KS == 16 ? 1 : 2) void dense8_head_kernel(Dense8HeadArgs ha) {
    using G = I8Geom<KS>;
    constexpr int VB = G::WAVES * 2;   // lane minima per query and workgroup
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ float2 s_par[TILE_ROWS];
    __shared__ double s_qn2[TILE_ROWS];
    __shared__ float s_lm[TILE_ROWS][16];
    __shared__ u32 s_ticket, s_unit_ticket;
    if (threadIdx.x == 0) s_unit_ticket = 0u;   // (the barriers of the query prep below come before its first use)
    const Dense8ScanArgs& a = ha.s;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r31 = lane & 31, h = lane >> 5;
    const float* q = ha.ind ? ha.ind->q : ha.q;   // (captured call graph: this launch's queries)
    const bool first = blockIdx.x == 0;
    // ---- the query tile, prepared by this workgroup for itself: planes in LDS (where the ring will be)
    signed char* planes = reinterpret_cast<signed char*>(smem);
#pragma unroll
    for (int rnd = 0; rnd < TILE_ROWS / (4 * G::WAVES); ++rnd) {
        const int qi = rnd * 4 * G::WAVES + wave * 4 + (lane >> 4);
        float2 pr;
        double Q;
        dense8_prep_query_row<KS / 2>(q + (long long)qi * ha.d, qi < ha.nq, ha.d, ha.center, ha.dx, ha.r_max, ha.x_max, ha.cosine,
                                      planes + qi * G::ROW_BYTES, planes + (TILE_ROWS + qi) * G::ROW_BYTES, pr, Q);
        if ((lane & 15) == 0) {
            s_par[qi] = pr;
            s_qn2[qi] = Q;
        }
        if (first) {   // what the body, the select and the host read: written once
            if ((lane & 15) == 0) {
                const_cast<float2*>(a.par)[qi] = pr;
                ha.qn2[qi] = Q;
                ha.cnt[qi << I8_CNT_SHIFT] = 0u;
                if (qi == 0) ha.oflag[0] = 0u;
            }
            if (qi < ha.nq)
                for (int i = lane & 15; i < ha.ldq; i += 16) ha.q_al[(long long)qi * ha.ldq + i] = i < ha.d ? q[(long long)qi * ha.d + i] : 0.f;
        }
    }
    __syncthreads();
    i32x4 bq[KS], bl[KS];
    dense8_load_planes<KS>(planes, bq, bl);
    float unit_lo = s_par[r31].x * 0.00390625f;
    asm volatile("" : "+v"(unit_lo));
    if (first) {
        signed char* gp = const_cast<signed char*>(a.qs8);
        for (int i = threadIdx.x; i < 2 * TILE_ROWS * G::ROW_BYTES / 16; i += G::WAVES * 64)
            reinterpret_cast<uint4*>(gp)[i] = reinterpret_cast<const uint4*>(planes)[i];
        for (int i = threadIdx.x; i < I8_HIST_WORDS; i += G::WAVES * 64) a.hist[i] = 0u;   // the body's histogram and threshold keys
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();   // the planes are in registers (and on their way to the body): the LDS is the ring's from here
    // ---- the sample pass: one running minimum per lane
    u32 wcount = 0;
    float smin = __builtin_inff();
    if (!(a.debug & 4096)) dense8_stream<KS, I8_LANEMIN>(a, smem, bq, bl, unit_lo, 0.f, wcount, smin, &s_unit_ticket);   // (4096: measurement, no sample pass)
    // ---- the workgroup's VB minima per query, sorted (sixteen lanes per query: a bitonic network over shuffles); only the
    // `keep` smallest leave.  The k best minima of a call fall on a workgroup k / workgroups at a time (Poisson): four or
    // eight per workgroup lose next to nothing, and the last workgroup's selection is over keep / VB as many values.
    s_lm[r31][(wave * 2 + h) & 15] = smin;
    if (VB < 16 && wave == 0 && h == 0)
        for (int i = VB; i < 16; ++i) s_lm[r31][i] = __builtin_inff();
    __syncthreads();
    const int nb = (int)gridDim.x, keep = ha.keep, M = nb * keep;
#pragma unroll
    for (int rnd = 0; rnd < TILE_ROWS / (4 * G::WAVES); ++rnd) {
        const int qi = rnd * 4 * G::WAVES + wave * 4 + (lane >> 4), l16 = lane & 15;
        float v = s_lm[qi][l16];
#pragma unroll
        for (int kk2 = 2; kk2 <= 16; kk2 <<= 1)
#pragma unroll
            for (int j = kk2 >> 1; j > 0; j >>= 1) {
                const float o = __shfl_xor(v, j);
                const bool up = (l16 & kk2) == 0, lower = (l16 & j) == 0;
                v = (lower == up) ? fminf(v, o) : fmaxf(v, o);
            }
        if (l16 < keep) ha.lane_min[(long long)qi * M + (long long)blockIdx.x * keep + l16] = v;
    }
    // ---- the workgroup that arrives last turns the minima into thresholds.  An agent-scope fence is an L2 write-back /
    // invalidate on a part whose XCDs have L2s of their own: ONE thread per workgroup pays it, behind a barrier that has
    // collected the workgroup's stores (every wave waited for its own), not every wave.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const u32 t = atomicAdd(&ha.oflag[1], 1u);
        if (t == gridDim.x - 1) {
            __threadfence();
            ha.oflag[1] = 0u;   // for the slot's next call (stream order)
        }
        s_ticket = t;
    }
    __syncthreads();
    if (s_ticket != gridDim.x - 1) return;
    float* thr = const_cast<float*>(a.thr);
    const Dense8ThrPost post{s_par, s_qn2};
#pragma unroll
    for (int rnd = 0; rnd < TILE_ROWS / (4 * G::WAVES); ++rnd) {
        const int q0 = rnd * 4 * G::WAVES + wave * 4;   // this wave's four queries
        float ts[4] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
        if (q0 < ha.nq && !(a.debug & 2048)) {   // (2048: measurement, no selection)
            const float* v = ha.lane_min + (long long)q0 * M;
            if (M <= 256)
                wave_kth_smallest<4, 4>(v, M, M, ha.kk, ts);
            else if (M <= 512)
                wave_kth_smallest<8, 4>(v, M, M, ha.kk, ts);
            else if (M <= 1024)
                wave_kth_smallest<16, 4>(v, M, M, ha.kk, ts);
            else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float one[1];
                    wave_kth_smallest<32, 1>(v + (long long)j * M, M, M, ha.kk, one);
                    ts[j] = one[0];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // padding queries of the tile: a NaN threshold -- no comparison passes, not even an always-candidate row's -inf
            const float t = q0 + j < ha.nq ? post(q0 + j, ts[j]) : __builtin_nanf("");
            if (lane == 0) thr[q0 + j] = t;
        }
    }
}

// The re-rank's inputs that the scan does not know.
struct Dense8TailArgs {
    const float* db;
    long long ld;
    int d;
    const float* q_al;
    int ldq, nq;
    void* keys;
    u32* cnt;
    u32 cap;
    u32* overflow;
    const double* nx64;
    const double* nq64;
    int debug;
    const double* qn2;          // |q''|^2 (Dense8ThrPost)
    int kk;
    int tighten;                // 0: the tail re-ranks every entry (measurement)
    long long* clk;             // measurement ("dense_debug" & 8192): per workgroup, wall_clock64 at entry / first wave out of the stream / all waves out / thresholds done / tail done
};
static constexpr int I8_TAIL_QROWS_BYTES = 32 * (156 + 4) * 4;   // rerank_block stages the query tile for ldq <= 156
static constexpr int I8_TAIL_LIST = 2048;                         // passing entries a workgroup compacts (more: uncompacted walk)
static constexpr int I8_TAIL_RERANK_BYTES = I8_TAIL_QROWS_BYTES + 3 * RERANK_MAX_GROUP * 4 + 8 * 32 * RERANK_STAGE_STRIDE * 4;
static constexpr int I8_TAIL_LDS_BYTES = I8_TAIL_RERANK_BYTES + I8_TAIL_LIST * 8 + TILE_ROWS * 4 + 16;

template <int KS, bool COSINE>
__global__ __launch_bounds__(I8Geom<KS>::WAVES * 64, KS == 16 ? 1 : 2) void dense8_body_kernel(Dense8ScanArgs a, Dense8TailArgs ta) {
    using G = I8Geom<KS>;
    static_assert(I8_TAIL_LDS_BYTES <= G::WAVES * G::NSTAGE * G::SLOT_BYTES, "the re-rank works in the ring's LDS");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ u32 s_hist8[TILE_ROWS * I8_HIST_BINS];
    __shared__ u32 s_wcnt[G::WAVES];
    __shared__ u32 s_unit_ticket;
    // what the tail needs of the call's per-query state, fetched while nothing waits for it
    __shared__ float s_thr1[TILE_ROWS], s_eq[TILE_ROWS];
    __shared__ double s_qn2b[TILE_ROWS];
    __shared__ float2 s_parb[TILE_ROWS];
    if (threadIdx.x == 0) s_unit_ticket = 0u;
    if (threadIdx.x < TILE_ROWS) {
        s_thr1[threadIdx.x] = a.thr[threadIdx.x];
        s_parb[threadIdx.x] = a.par[threadIdx.x];
        s_eq[threadIdx.x] = a.par[threadIdx.x].y;
        s_qn2b[threadIdx.x] = ta.qn2[threadIdx.x];
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r31 = lane & 31;
    for (int i = threadIdx.x; i < TILE_ROWS * I8_HIST_BINS; i += G::WAVES * 64) s_hist8[i] = 0u;
    long long* clk = ta.clk ? ta.clk + (long long)blockIdx.x * 8 : nullptr;
    if (clk && threadIdx.x == 0) {
        clk[0] = (long long)wall_clock64();
        clk[1] = 0x7fffffffffffffffll;
    }
    __syncthreads();
    i32x4 bq[KS], bl[KS];
    dense8_load_planes<KS>(a.qs8, bq, bl);
    float unit_lo = a.par[r31].x * 0.00390625f;   // Dx Dq / 256, exact
    float thr_l = a.thr[r31];
    // histogram bins of e_q / 2 (no usable e_q: everything in bin 0, nothing is tightened)
    const float eq_l = a.par[r31].y;
    float inv_w_l = (eq_l > 0.f && eq_l < __builtin_inff()) ? 2.f / eq_l : 0.f;
    if (!(inv_w_l < 3.0e38f)) inv_w_l = 0.f;
    asm volatile("" : "+v"(unit_lo), "+v"(thr_l), "+v"(inv_w_l));
    u32 wcount = 0;
    float smin = 0.f;
    dense8_stream<KS, I8_EMIT_H>(a, smem, bq, bl, unit_lo, thr_l, wcount, smin, &s_unit_ticket, inv_w_l, s_hist8);
    const long long w0 = (long long)blockIdx.x * G::WAVES;
    if (lane == 0) {
        a.wave_cnt[2 * (w0 + wave)] = wcount;
        a.wave_cnt[2 * (w0 + wave) + 1] = 0u;
        s_wcnt[wave] = wcount;
        if (clk) atomicMin((unsigned long long*)(clk + 1), (unsigned long long)wall_clock64());
    }
    // ---- the tail: this workgroup's own survivor segments, re-ranked in the reference arithmetic from the original rows
    // (workgroup scope: the segments are written and read on this CU -- an agent-scope fence would write back the L2)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();    // every wave has left the ring and its stores have landed
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (clk && threadIdx.x == 0) clk[2] = (long long)wall_clock64();
    float* fl = reinterpret_cast<float*>(smem);
    u32* su = reinterpret_cast<u32*>(smem + I8_TAIL_QROWS_BYTES);
    RerankLds L{fl, su, su + RERANK_MAX_GROUP, su + 2 * RERANK_MAX_GROUP,
                COSINE ? reinterpret_cast<float*>(smem + I8_TAIL_QROWS_BYTES + 3 * RERANK_MAX_GROUP * 4) : nullptr};
    // ---- the tightened thresholds of this workgroup (I8_EMIT_H above): a wave takes four queries at a time, lane = bin
    {
        uint2* s_list = reinterpret_cast<uint2*>(smem + I8_TAIL_RERANK_BYTES);
        float* s_thr2 = reinterpret_cast<float*>(smem + I8_TAIL_RERANK_BYTES + I8_TAIL_LIST * 8);
        u32* s_npass = reinterpret_cast<u32*>(s_thr2 + TILE_ROWS);
        static_assert(I8_HIST_BINS == 64, "one lane per bin");
        bool any = false;
#pragma unroll
        for (int wi = 0; wi < G::WAVES; ++wi) any = any || s_wcnt[wi] != 0u;
        const Dense8ThrPost post{s_parb, s_qn2b};
        // (the tail is a chain of dependent round trips on a memory system the other workgroups keep saturated: the
        // histograms of a wave's queries are fetched together, and the workgroup's own unflushed counts are added from LDS)
        constexpr int NQW = TILE_ROWS / G::WAVES;   // queries per wave
        u32 hc[NQW];
        if (ta.tighten && any) {
#pragma unroll
            for (int j = 0; j < NQW; ++j) {
                const int qi = (j >> 2) * 4 * G::WAVES + wave * 4 + (j & 3);
                hc[j] = __hip_atomic_load(a.hist + qi * I8_HIST_BINS + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int j = 0; j < NQW; ++j) {
                const int qi = (j >> 2) * 4 * G::WAVES + wave * 4 + (j & 3);
                hc[j] += s_hist8[qi * I8_HIST_BINS + lane];
            }
        }
#pragma unroll
        for (int rnd = 0; rnd < TILE_ROWS / (4 * G::WAVES); ++rnd) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int qi = rnd * 4 * G::WAVES + wave * 4 + j;
                const float t1 = s_thr1[qi];
                float t2 = t1;
                if (ta.tighten && any && qi < ta.nq && t1 < __builtin_inff() && t1 > -__builtin_inff()) {
                    u32 c = hc[rnd * 4 + j];
                    // suffix sums: entries in bins lane .. 63
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const u32 up = (u32)__shfl_down((int)c, o);
                        c += lane + o < 64 ? up : 0u;
                    }
                    const u64 reach = __ballot(c >= (u32)ta.kk);
                    const float eq = s_eq[qi];
                    if (reach != 0ull && eq > 0.f && eq < __builtin_inff()) {
                        const int jb = 63 - __clzll((long long)reach);   // the highest bin with k entries at or beyond it
                        const float inv_w = 2.f / eq;                   // as the stream computed it
                        if (inv_w < 3.0e38f) {
                            // an entry of bin jb has fl(fl(T' - m) inv_w) >= jb: m <= T' - (jb / inv_w)(1 - 2^-22)
                            const double t8 = (double)t1 - (double)jb / (double)inv_w * (1.0 - 1e-6);
                            float t8f = (float)t8;
                            if ((double)t8f < t8) t8f = __uint_as_float(__float_as_uint(t8f) + (t8f >= 0.f ? 1 : -1));
                            const float cand = post(qi, t8f);
                            if (cand < t2) t2 = cand;
                        }
                    }
                    if (lane == 0) atomicMax(a.hist + TILE_ROWS * I8_HIST_BINS + (qi << I8_CNT_SHIFT), ordered_f32(t2));
                }
                if (lane == 0) s_thr2[qi] = t2;
            }
        }
        if (ta.tighten) {
            L.scores = a.wave_score;
            L.thr2 = s_thr2;
            L.list = s_list;
            L.npass = s_npass;
            L.list_cap = I8_TAIL_LIST;
        }
        L.seg_cnt = s_wcnt;
        L.cnt_shift = I8_CNT_SHIFT;
        __syncthreads();
        if (clk && threadIdx.x == 0) clk[3] = (long long)wall_clock64();
        // what the waves counted since their last flush, for the workgroups still at work (nobody here waits for it)
        if (ta.tighten) dense8_hist_flush<G::WAVES>(s_hist8, a.hist);
    }
    using K = typename std::conditional<COSINE, K128, u64>::type;
    rerank_block<K, COSINE>(ta.db, ta.ld, ta.d, ta.q_al, ta.ldq, ta.nq, TILE_ROWS, a.wave_out, a.wave_cnt, a.wave_cap,
                            (long long)a.nrb * G::WAVES, G::WAVES, static_cast<K*>(ta.keys), ta.cnt, ta.cap, ta.overflow, ta.nx64,
                            ta.nq64, ta.debug, w0, L);
    if (clk) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) clk[4] = (long long)wall_clock64();
    }
}

// ---------------------------------------------------------------- 33 .. 256 queries per call (128-byte rows)
// QT query tiles per wave (2 for up to 64 queries, 4 beyond): the unit's A fragments are read once and meet QT sets of
// query planes (32 QT registers per lane), so one pass over the int8 copy serves 32 QT queries -- half the bytes of the
// bf16 multi-tile pass.  Measured at 10 M x 128 (tools/step_sweep.py dense_int8_batch=...): 64 queries 0.468 -> 0.315 ms per
// step; 128 queries 0.531 -> 0.516 (four tiles: the conversion and comparison of 8 x 16 scores per lane and unit is as long
// as the stream); 256 and 512 queries (two and four groups) 0.90 -> 1.07 and 1.62 -> 1.94 -- the bf16 kernels, whose
// epilogue is a minimum chain on float accumulators that already hold the norms, stay ahead there.  Hence the default
// "dense_int8_batch" = 64: two tiles only.
// More than 32 QT queries: nqt groups, the workgroups of an XCD walking the groups of the same row block so that the
// rows come from L2 again (as dense_scan_kernel does).  Everything else is dense8_scan_kernel<4, .>: units of 64 rows,
// two tiles -- here read and scored one after the other to keep the registers under 256.
template <int QT, bool SAMPLE>
__global__ __launch_bounds__(512, 2) void dense8_scan_mt_kernel(Dense8ScanArgs a) {
    using G = I8Geom<4>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r31 = lane & 31, h = lane >> 5;
    const u32 lds_base = (u32)(uintptr_t)smem;
    const u32 ring_base = lds_base + (u32)wave * (G::NSTAGE * G::SLOT_BYTES);
    const unsigned char* ring_ptr = smem + wave * (G::NSTAGE * G::SLOT_BYTES);
    const long long wave_id = (long long)blockIdx.x * G::WAVES + wave;   // unique per wave of the launch: its survivor segment
    uint2* wout = a.wave_out + wave_id * a.wave_cap;
    // workgroup -> (row block, group of QT query tiles)
    const int L = blockIdx.x;
    int grp, rb;
    if (a.nqt > 1) {
        const int xcd = L & 7, j = L >> 3;
        grp = j % a.nqt;
        rb = (j / a.nqt) * 8 + xcd;
    } else {
        grp = 0;
        rb = L;
    }
    const int q_first = grp * QT * TILE_ROWS;   // first query of the group
    const long long rwave = (long long)rb * G::WAVES + wave, nwaves = (long long)a.nrb * G::WAVES;

    i32x4 bq[QT][4], bl[QT][4];
    float unit_lo[QT], thr_l[QT];
#pragma unroll
    for (int g = 0; g < QT; ++g) {
        const int qg = q_first + g * TILE_ROWS + r31;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bq[g][s] = *reinterpret_cast<const i32x4*>(a.qs8 + (long long)qg * G::ROW_BYTES + (2 * s + h) * 16);
            bl[g][s] = *reinterpret_cast<const i32x4*>(a.qs8 + (long long)(a.plane_rows + qg) * G::ROW_BYTES + (2 * s + h) * 16);
        }
        unit_lo[g] = a.par[qg].x * 0.00390625f;
        thr_l[g] = SAMPLE ? 0.f : a.thr[qg];
        asm volatile("" : "+v"(unit_lo[g]), "+v"(thr_l[g]));
#pragma unroll
        for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(bq[g][s]), "+v"(bl[g][s]));
    }

    const long long my_units = rwave < a.n_sel ? (a.n_sel - rwave + nwaves - 1) / nwaves : 0;
    u32 voff[G::PIECES];
#pragma unroll
    for (int j = 0; j < G::PIECES; ++j) {
        const int r = 8 * j + (lane >> 3);
        voff[j] = (u32)(r * G::ROW_BYTES + i8_swz<4>(lane & 7, r) * 16);
    }
    const u32 voff_n = (u32)lane * 4u;
    long long issued = 0;
    auto issue_next = [&]() __attribute__((always_inline)) {
        if (issued >= my_units) return;
        const long long unit_idx = (rwave + issued * nwaves) * a.unit_step;
        const long long row0 = unit_idx * G::UNIT_ROWS;
        const u32 dst = ring_base + (u32)(issued % G::NSTAGE) * G::SLOT_BYTES;
        const unsigned char* base = reinterpret_cast<const unsigned char*>(a.scan8) + row0 * G::ROW_BYTES;
#pragma unroll
        for (int j = 0; j < G::PIECES; ++j) {
            if (a.nt && row0 >= a.nt_from_row)
                glds16<true>(base, voff[j], dst + (u32)j * 1024);
            else
                glds16<false>(base, voff[j], dst + (u32)j * 1024);
        }
        glds4(a.nrow + row0, voff_n, dst + G::UNIT_BYTES);
        ++issued;
    };
    for (int p = 0; p < G::NSTAGE; ++p) issue_next();

    u32 wcount = 0;
    for (long long it = 0; it < my_units; ++it) {
        const long long unit_idx = (rwave + it * nwaves) * a.unit_step;
        const long long row0 = unit_idx * G::UNIT_ROWS;
        wait_units_in_flight<G::NSTAGE, G::PIECES + 1>((int)(issued - it - 1));
        const unsigned char* sl = ring_ptr + (it % G::NSTAGE) * G::SLOT_BYTES;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int r = 32 * t + r31;
            i32x4 av[4];
            f32x4 nr[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) av[s] = *reinterpret_cast<const i32x4*>(sl + r * G::ROW_BYTES + i8_swz<4>(2 * s + h, r) * 16);
#pragma unroll
            for (int c = 0; c < 4; ++c) nr[c] = *reinterpret_cast<const f32x4*>(sl + G::UNIT_BYTES + (32 * t + 8 * c + 4 * h) * 4);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (t == 1) issue_next();   // both tiles are in registers or done: the slot is free
#pragma unroll
            for (int g = 0; g < QT; ++g) {
                i32x16 acc, acl;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = acl[i] = 0;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[s], bq[g][s], acc, 0, 0, 0);
                    acl = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[s], bl[g][s], acl, 0, 0, 0);
                }
                float sc[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float nv = nr[i >> 2][i & 3];
                    if constexpr (SAMPLE) nv = nv == -__builtin_inff() ? __builtin_inff() : nv;
                    sc[i] = __fmaf_rn((float)((acc[i] << 8) + acl[i]), unit_lo[g], nv);
                }
                float m = sc[0];
#pragma unroll
                for (int i = 1; i < 16; ++i) m = fminf(m, sc[i]);
                if constexpr (SAMPLE) {
                    const long long sel = rwave + it * nwaves;
                    a.sample_out[(long long)(q_first + g * TILE_ROWS + r31) * a.ns + sel * G::SAMPLES_PER_UNIT + t * 2 + h] = m;
                } else {
                    const u64 hit = __ballot(m <= thr_l[g]);
                    if (hit != 0) {
                        u32 mask = 0;
#pragma unroll
                        for (int i = 0; i < 16; ++i) mask |= (sc[i] <= thr_l[g] ? 1u : 0u) << i;
                        const u64 bal = __ballot(mask != 0);
                        if (mask) {
                            const u32 pos = wcount + __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u));
                            if (pos < a.wave_cap) wout[pos] = make_uint2((u32)(row0 + 32 * t + 4 * h), (mask << 16) | (u32)(g * TILE_ROWS + r31));
                        }
                        wcount += (u32)__popcll(bal);
                    }
                }
            }
        }
    }
    if constexpr (!SAMPLE) {
        if (lane == 0) {
            a.wave_cnt[2 * wave_id] = wcount;
            a.wave_cnt[2 * wave_id + 1] = (u32)(grp * QT);   // first query tile of the group (the re-rank's q0 / 32)
        }
    }
}

}  // namespace sq
