// The streaming filter of the dense path: one pass over a bfloat16 "scan copy"
// of the descriptor matrix on the bf16 matrix cores (gfx950).
//
// Why a scan copy.  The filter only has to be right within a known error
// bound (every survivor is re-ranked in the reference's float32/float64
// arithmetic from the original rows), but it has to stream the
// whole matrix at the HBM rate for many queries at once.  float32 MFMA
// (v_mfma_f32_32x32x2_f32) runs at the VALU rate, 1/16 of bf16, and made the
// scan compute bound beyond ~24 queries.  The rows are therefore rounded once,
// at index build time, to bfloat16 (|x - x_hi| <= 2^-8 |x| per element) and
// stored in exactly the order the MFMA fragments are read: HALF the bytes of
// the float32 matrix, so a pass moves N*d*2 (+ N*4 of norms) bytes.  The
// queries keep two bfloat16 planes (q = q_hi + q_lo, residual <= 2^-16 |q|):
// a tile costs two MFMAs per k-step, x_hi*q_hi + x_hi*q_lo, and the score
// error is bounded by (2^-8 + 2^-15) |x||q'| (Cauchy-Schwarz over the
// element-wise bounds) plus float32 accumulation: the bound the threshold
// slack and the certification use (sq_dense.hip).
//
// Scan-copy layout: row-major, row stride d_pad*2 bytes; a row is KU = d_pad/128
// units of 256 bytes; a unit is 16 chunks of 16 bytes; chunk c = 2*s + h holds
// 8 bfloat16 = elements k = 128*unit + 16*s + 8*h + j (j = 0..7).  That is the
// A operand of MFMA k-step s (0..7) for lane half h (cdna_hip_programming.md
// section 3, bf16 operand maps).  A prepared query row is KU units of 512
// bytes: the 16 chunks of the hi plane, then the 16 chunks of the lo plane.
#pragma once
#include <type_traits>

#include "sq_common.hpp"
#include "sq_dma.hpp"

namespace sq {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

static constexpr int KT = 128;                         // elements per k-unit (256 bytes of bfloat16)
static constexpr int TILE_ROWS = 32;                   // rows per MFMA tile
static constexpr int UNIT_BYTES = TILE_ROWS * KT * 2;  // 8 KiB: 32 rows x 256 bytes
static constexpr int NORM_BYTES = 256;                 // per ring slot: |x|^2 of the tile's 32 rows (64 lanes x 4 B)
static constexpr int SLOT_BYTES = UNIT_BYTES + NORM_BYTES;
static constexpr int RING_MAX_DPAD = 512;   // widest padded row of the ring kernels (dense_scan_kernel: query tile in registers / LDS)
static constexpr int MAX_DPAD = 8192;       // widest padded row with a scan copy at all (beyond the ring kernels: sq_dense_wide.hpp)

// --------------------------------------------------------------- bf16 split
__device__ __forceinline__ u32 bf16_bits_rn(float x) {
    u32 u = __float_as_uint(x);
    // NaN stays NaN: the integer rounding below would carry a NaN with a full mantissa into the sign bit
    // (0x7FFFFFFF -> -0.0, a finite-looking value; MI355X_MICROARCH.md pitfalls).  +-inf round to themselves.
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (u >> 16) | 0x0040u;
    u += 0x7FFFu + ((u >> 16) & 1u);  // round to nearest even
    return u >> 16;
}
__device__ __forceinline__ void bf16_split(float x, u32& hi, u32& lo) {
    hi = bf16_bits_rn(x);
    const float r = __fsub_rn(x, __uint_as_float(hi << 16));  // exact
    lo = bf16_bits_rn(r);
}

// Column means of the matrix, float64 sums -> float32 centre c (L2 only).  Euclidean distance is
// translation invariant, so the FILTER scores x - c against q - c: its error bound scales with
// |x - c||q - c| instead of |x||q|, which is what keeps the candidate lists short when the data
// sits far from the origin (all-positive features, un-centred embeddings: with |x| = 20 and a
// spread of 0.5 the un-centred bound covered every row).  The exact re-rank uses the original rows.
static __global__ __launch_bounds__(256) void dense_colsum_kernel(const float* __restrict__ db, long long n, long long ld,
                                                                   int d, long long rows_per_block,
                                                                   double* __restrict__ colsum) {
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    for (int k = threadIdx.x; k < d; k += 256) {
        double acc = 0.0;
        for (long long r = r0; r < r1; ++r) {
            // non-finite elements are left out: one NaN / inf row must not turn the filter's origin (and with it
            // every filter score) into NaN -- such a row can never be a candidate anyway
            const float v = db[r * ld + k];
            if (fabsf(v) < __builtin_inff()) acc += (double)v;
        }
        atomicAdd(&colsum[k], acc);
    }
}
static __global__ void dense_center_kernel(const double* __restrict__ colsum, long long n, int d, int d_pad,
                                           float* __restrict__ center) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < d_pad) center[k] = k < d ? (float)(colsum[k] / (double)n) : 0.f;
}

// One thread per 16-byte chunk of the scan copy.  `row_scale` (cosine): 1/|x|; `center` (L2): c.
static __global__ __launch_bounds__(256) void dense_build_scan_kernel(const float* __restrict__ db, long long n,
                                                                       long long ld, int d, int d_pad,
                                                                       long long n_pad,
                                                                       const float* __restrict__ row_scale,
                                                                       const float* __restrict__ center,
                                                                       uint4* __restrict__ scan, long long row_base) {
    const int cpr = d_pad / 8;  // chunks per row
    const long long idx = row_base * cpr + (long long)blockIdx.x * 256 + threadIdx.x;  // rows from row_base on (append)
    if (idx >= n_pad * cpr) return;
    const long long row = idx / cpr;
    const int cc = (int)(idx - row * cpr);
    const int unit = cc >> 4, c = cc & 15, s = c >> 1, h = c & 1;
    const int k0 = unit * KT + 16 * s + 8 * h;
    u32 w[4];
    const float sc = (row < n && row_scale) ? row_scale[row] : 1.f;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        u32 half[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int k = k0 + j + e;
            float x = 0.f;
            if (row < n && k < d) {
                x = row_scale ? __fmul_rn(db[row * ld + k], sc) : db[row * ld + k];
                if (center) x = __fsub_rn(x, center[k]);  // L2: the filter works on x - c (see dense_colmean_kernel)
            }
            half[e] = bf16_bits_rn(x);
        }
        w[j >> 1] = half[0] | (half[1] << 16);
    }
    scan[idx] = make_uint4(w[0], w[1], w[2], w[3]);
}

// Per-row squared norms (float32, from a float64 sum), their maximum (for the
// error bound) and, for cosine, 1/|x|.  8 lanes per row.
static __global__ __launch_bounds__(256) void dense_rowstats_kernel(const float* __restrict__ db, long long n,
                                                                     long long ld, int d, long long n_pad,
                                                                     u32* __restrict__ max_bits,
                                                                     float* __restrict__ norms,
                                                                     float* __restrict__ inv_norm,
                                                                     const float* __restrict__ center,
                                                                     float* __restrict__ norms1, double shrink2,
                                                                     double shrink1, long long row_base) {
    const int lane8 = threadIdx.x & 7;
    const long long row = row_base + (long long)blockIdx.x * 32 + (threadIdx.x >> 3);  // rows from row_base on (append)
    const long long r = row < n ? row : n - 1;
    const float* x = db + r * ld;
    double acc = 0.0;
    // rows are 16-byte aligned with ld % 4 == 0 (sq_dense_create): the eight lanes of a row read 128
    // contiguous bytes per step
    const int d4 = d & ~3;
    for (int i = 4 * lane8; i < d4; i += 32) {
        f32x4 v = *reinterpret_cast<const f32x4*>(x + i);
        if (center) {  // the float32 difference the scan copy is built from
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = __fsub_rn(v[j], center[i + j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += (double)v[j] * (double)v[j];
    }
    if (lane8 == 0)
        for (int i = d4; i < d; ++i) {
            const float v = center ? __fsub_rn(x[i], center[i]) : x[i];
            acc += (double)v * (double)v;
        }
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    acc += __shfl_xor(acc, 4);
    if (lane8 == 0 && row < n_pad) {
        // L2: the scan starts a row's score from |x|^2 (1 - alpha), rounded down -- the row's share of its
        // own error bound, one array per query-plane count; cosine: plain |x|^2 (unused by the scan)
        if (row < n) {
            norms[row] = __double2float_rd(acc * shrink2);
            if (norms1) norms1[row] = __double2float_rd(acc * shrink1);
            if (inv_norm) inv_norm[row] = acc > 0.0 ? (float)(1.0 / sqrt(acc)) : 0.f;
        } else {
            norms[row] = 0.f;
            if (norms1) norms1[row] = 0.f;
        }
    }
    // one atomic per workgroup for the largest squared norm (one per row serialised 10 M atomics on one
    // address: 14 ms of a 17 ms index build)
    __shared__ float s_max[4];
    float m = row < n ? (float)(acc * (1.0 + 1e-6)) : 0.f;
    if (!(m < __builtin_inff())) m = 0.f;  // rows with non-finite elements never pass the filter: not part of the bound
    for (int o = 8; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicMax(max_bits, __float_as_uint(fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]))));
}

// Query prep: |q|^2 (f64) and the scaled query (L2: -2q, cosine: -q/|q|) split into
// bfloat16 hi + lo planes in the fragment order, [nq_pad][d_pad/4] chunks.
// Also resets the per-call state of query qi (threshold = -inf, candidate count 0,
// overflow flag) and writes the aligned float32 copy q_al[nq][ldq] the re-rank reads.
static __global__ __launch_bounds__(256) void dense_prep_queries_kernel(const float* __restrict__ q, int nq, int d,
                                                                         int d_pad, int metric,
                                                                         uint4* __restrict__ qs,
                                                                         double* __restrict__ qn2,
                                                                         float* __restrict__ thr,
                                                                         u32* __restrict__ cnt,
                                                                         u32* __restrict__ oflag,
                                                                         float* __restrict__ q_al, int ldq,
                                                                         const float* __restrict__ center) {
    const int qi = blockIdx.x;
    __shared__ double red[4];
    if (threadIdx.x == 0) {
        thr[qi] = -__builtin_inff();
        cnt[qi] = 0u;
        if (qi == 0) *oflag = 0u;
    }
    if (qi < nq)
        for (int i = threadIdx.x; i < ldq; i += 256) q_al[(long long)qi * ldq + i] = i < d ? q[(long long)qi * d + i] : 0.f;
    // L2 with a centre: everything the filter uses (|q'|^2, the split planes) is q' = q - c in float32
    auto qv = [&](int i) __attribute__((always_inline)) {
        const float v = q[(long long)qi * d + i];
        return center ? __fsub_rn(v, center[i]) : v;
    };
    double acc = 0.0;
    if (qi < nq)
        for (int i = threadIdx.x; i < d; i += 256) acc += (double)qv(i) * (double)qv(i);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    const double tot = red[0] + red[1] + red[2] + red[3];
    if (threadIdx.x == 0) qn2[qi] = qi < nq ? tot : 0.0;
    double scale = -2.0;
    if (metric == SQ_METRIC_COSINE) scale = tot > 0.0 ? -1.0 / sqrt(tot) : 0.0;
    const int cpr = d_pad / 4;
    for (int cc = threadIdx.x; cc < cpr; cc += 256) {
        const int unit = cc >> 5, p = (cc >> 4) & 1, c = cc & 15, s = c >> 1, h = c & 1;
        const int k0 = unit * KT + 16 * s + 8 * h;
        u32 w[4];
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            u32 half[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int k = k0 + j + e;
                float x = 0.f;
                if (qi < nq && k < d) x = (float)((double)qv(k) * scale);
                u32 hi, lo;
                bf16_split(x, hi, lo);
                half[e] = p ? lo : hi;
            }
            w[j >> 1] = half[0] | (half[1] << 16);
        }
        qs[(long long)qi * cpr + cc] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// ------------------------------------------------------------- the scan kernel
struct DenseScanArgs {
    const uint4* scan;      // scan copy [n_pad][d_pad/8] chunks
    const float* norms;     // |x|^2 per row [n_pad] (L2) or nullptr (cosine; the AGPR configurations get 32 zeros and norm_step 0)
    long long norm_step;    // 1: norms[row]; 0: the same 32 entries for every tile
    int nt;                 // the copy is far larger than the MALL: launches of ONE query group take the non-temporal build of the multi-tile kernels
    long long nt_from_row;  // one-tile kernel, one query group: rows from here on stream non-temporally (LLONG_MAX: none); the head
                            // of the copy keeps the default policy and is found in the 256 MB MALL again by the next call
    long long n;            // real rows (rows >= n are padding and never emitted)
    long long n_tiles;      // ceil(n / 32)
    const uint4* qs;        // [nqt*32][d_pad/4] prepared queries
    const float* thr;       // [nqt*32] score thresholds (EMIT)
    uint2* wave_out;        // [waves of the launch][wave_cap] survivor entries (first row, mask<<16 | query), one segment per wave
    u32* wave_cnt;          // [waves of the launch][2] entries each wave wrote (may exceed wave_cap: overflow), first query tile
    u32 wave_cap;
    float* sample_out;      // [nqt*32][ns] (SAMPLE)
    long long ns;
    long long tile_step;    // SAMPLE: every tile_step-th tile; EMIT: 1
    long long n_sel;        // number of tiles this launch visits
    int nqt;                // groups of QT query tiles (QT = template parameter of the launch)
    int nrb;                // row blocks (multiple of 8 when nqt > 1)
    int debug;              // measurement only: 1 = skip LDS reads + MFMA, 2 = stop the DMA after the first ring fill
    // Fused query prep (L2, one query tile per wave): the kernel builds the bf16 planes of its query tile itself from
    // the caller's float32 queries -- (q - c) * -2, split exactly as dense_prep_queries_kernel splits it -- instead of
    // reading a prepared copy: one launch less at the head of a call (what remains of the prep -- |q - c|^2, counters,
    // the aligned copy for the re-rank -- is the prologue of the threshold kernel, DenseThrPost).  nullptr: `qs`.
    const float* raw_q;     // [raw_nq][raw_d] float32
    int raw_nq, raw_d;
    const float* center;    // [d_pad] or nullptr
    float* wave_score;      // optional, parallel to wave_out: each entry's smallest filter score (sq_dense_wide.hpp -> sq_dense_tighten.hpp)
};

typedef __attribute__((address_space(3))) u32 lds_u32;
typedef __attribute__((address_space(3))) const f32x4 lds_cf32x4;

// ---- hand-placed MFMA for the multi-tile (QT > 1) configuration.
// With QT*64 registers of query fragments a wave needs more than the 256
// architectural VGPRs; hipcc then parks the fragments in AGPRs and copies each one
// back (4 v_accvgpr_read + s_nop per MFMA, ~60% of the loop's issue slots).  The
// matrix core reads its B operand from an AGPR directly, so the QT > 1 path loads
// the fragments into AGPRs itself (ds_read_b128 with an AGPR destination) and
// issues the MFMAs from inline asm with the B operand constrained to "a".
// hipcc does not pad hazards of instructions inside asm strings
// (cdna_hip_programming.md 5.7 item 2); the pads are explicit:
//   mfma_fence_in : VALU write of A / C registers -> MFMA read (2 states)
//   mfma_fence_out: MFMA result -> any non-MFMA reader, and A registers ->
//                   overwritten (8-pass MFMA: 12 states; 16 given)
__device__ __forceinline__ void mfma_bf16_agpr_b(f32x16& acc, const f32x4& a_v, const f32x4& b_a) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a_v), "a"(b_a));
}
// first MFMA of a chain: D = A*B + C with C a different register block (the tile's norms)
__device__ __forceinline__ void mfma_bf16_agpr_b_first(f32x16& acc, const f32x4& a_v, const f32x4& b_a, const f32x16& c_v) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(acc) : "v"(a_v), "a"(b_a), "v"(c_v));
}
template <int QT>
__device__ __forceinline__ void mfma_fence_in(f32x16 (&acc)[QT], f32x4 (&av)[8]) {
    // every A / C register is written before this point and the pad follows the last write
    if constexpr (QT == 4)
        asm volatile("s_nop 1"
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(av[0]), "+v"(av[1]), "+v"(av[2]),
                       "+v"(av[3]), "+v"(av[4]), "+v"(av[5]), "+v"(av[6]), "+v"(av[7]));
    else if constexpr (QT == 2)
        asm volatile("s_nop 1"
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(av[4]),
                       "+v"(av[5]), "+v"(av[6]), "+v"(av[7]));
    else
        asm volatile("s_nop 1"
                     : "+v"(acc[0]), "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(av[4]), "+v"(av[5]),
                       "+v"(av[6]), "+v"(av[7]));
    static_assert(QT == 1 || QT == 2 || QT == 4, "query tiles per wave");
}
// the same pad for the norm block (C of every chain's first MFMA) and the A fragments
__device__ __forceinline__ void mfma_fence_in_c(f32x16& c, f32x4 (&av)[8]) {
    asm volatile("s_nop 1"
                 : "+v"(c), "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(av[4]), "+v"(av[5]), "+v"(av[6]),
                   "+v"(av[7]));
}
template <int QT>
__device__ __forceinline__ void mfma_fence_out(f32x16 (&acc)[QT]) {
    if constexpr (QT == 4)
        asm volatile("s_nop 15" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
    else if constexpr (QT == 2)
        asm volatile("s_nop 15" : "+v"(acc[0]), "+v"(acc[1]));
    else
        asm volatile("s_nop 15" : "+v"(acc[0]));
}
// min of the 16 scores of a lane as eight VALU instructions that stay where they are written (asm volatile): the
// skewed epilogue below places them between hand-issued MFMAs.  NaN scores lose (IEEE min), as with fminf.
__device__ __forceinline__ float min16_pinned(const f32x16& v) {
    float m;
    // one statement: between separate asm statements hipcc pads every instruction with an s_nop (4 cycles each)
    asm volatile(
        "v_min3_f32 %0, %1, %2, %3\n\t"
        "v_min3_f32 %0, %0, %4, %5\n\t"
        "v_min3_f32 %0, %0, %6, %7\n\t"
        "v_min3_f32 %0, %0, %8, %9\n\t"
        "v_min3_f32 %0, %0, %10, %11\n\t"
        "v_min3_f32 %0, %0, %12, %13\n\t"
        "v_min3_f32 %0, %0, %14, %15\n\t"
        "v_min_f32 %0, %0, %16"
        : "=&v"(m)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]),
          "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]));
    return m;
}
// bit i of the result = (v[i] <= thr), per lane: 16 compares into three rotating scalar pairs, each shifted into the
// mask by an add-with-carry (mask = 2 mask + bit, rows 15 .. 0) two instructions after its compare (gfx950: a VALU
// read of a scalar register pair needs two wait states after the VALU that wrote it).  32 instructions where hipcc's
// compare / select / or3 sequence with its own pads comes to ~70 issue slots.
__device__ __forceinline__ u32 le_mask16(const f32x16& v, float thr) {
    u32 mask = 0;
    u64 c0, c1, c2, dump;
    asm volatile(
        "v_cmp_le_f32_e64 %1, %5, %21\n\t"
        "v_cmp_le_f32_e64 %2, %6, %21\n\t"
        "v_cmp_le_f32_e64 %3, %7, %21\n\t"
        "v_addc_co_u32_e64 %0, %4, %0, %0, %1\n\t"
        "v_cmp_le_f32_e64 %1, %8, %21\n\t"
        "v_addc_co_u32_e64 %0, %4, %0, %0, %2\n\t"
        "v_cmp_le_f32_e64 %2, %9, %21\n\t"
        "v_addc_co_u32_e64 %0, %4, %0, %0, %3\n\t"
        "v_cmp_le_f32_e64 %3, %10, %21\n\t"
        "v_addc_co_u32_e64 %0, %4, %0, %0, %1\n\t"
        "v_cmp_le_f32_e64 %1, %11, %21\n\t"
        "v_addc_co_u32_e64 %0, %4, %0, %0, %2\n\t"
        "v_cmp_le_f32_e64 %2, %12, %21\n\t"
        "v_addc_co_u32_e64 %0, %4, %0, %0, %3\n\t"
        "v_cmp_le_f32_e64 %3, %13, %21\n\t"
        "v_addc_co_u32_e64 %0, %4, %0, %0, %1\n\t"
        "v_cmp_le_f32_e64 %1, %14, %21\n\t"
        "v_addc_co_u32_e64 %0, %4, %0, %0, %2\n\t"
        "v_cmp_le_f32_e64 %2, %15, %21\n\t"
        "v_addc_co_u32_e64 %0, %4, %0, %0, %3\n\t"
        "v_cmp_le_f32_e64 %3, %16, %21\n\t"
        "v_addc_co_u32_e64 %0, %4, %0, %0, %1\n\t"
        "v_cmp_le_f32_e64 %1, %17, %21\n\t"
        "v_addc_co_u32_e64 %0, %4, %0, %0, %2\n\t"
        "v_cmp_le_f32_e64 %2, %18, %21\n\t"
        "v_addc_co_u32_e64 %0, %4, %0, %0, %3\n\t"
        "v_cmp_le_f32_e64 %3, %19, %21\n\t"
        "v_addc_co_u32_e64 %0, %4, %0, %0, %1\n\t"
        "v_cmp_le_f32_e64 %1, %20, %21\n\t"
        "v_addc_co_u32_e64 %0, %4, %0, %0, %2\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32_e64 %0, %4, %0, %0, %3\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32_e64 %0, %4, %0, %0, %1"
        : "+v"(mask), "=&s"(c0), "=&s"(c1), "=&s"(c2), "=&s"(dump)
        : "v"(v[15]), "v"(v[14]), "v"(v[13]), "v"(v[12]), "v"(v[11]), "v"(v[10]), "v"(v[9]), "v"(v[8]), "v"(v[7]),
          "v"(v[6]), "v"(v[5]), "v"(v[4]), "v"(v[3]), "v"(v[2]), "v"(v[1]), "v"(v[0]), "v"(thr));
    return mask;
}
// eight 16-byte LDS reads straight into AGPRs, complete on return
template <int N>
__device__ __forceinline__ void lds_read8_agpr(f32x4 (&dst)[N], int g0, const u32 (&ad)[8]) {
    asm volatile(
        "ds_read_b128 %0, %8\n\t"
        "ds_read_b128 %1, %9\n\t"
        "ds_read_b128 %2, %10\n\t"
        "ds_read_b128 %3, %11\n\t"
        "ds_read_b128 %4, %12\n\t"
        "ds_read_b128 %5, %13\n\t"
        "ds_read_b128 %6, %14\n\t"
        "ds_read_b128 %7, %15\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&a"(dst[g0 + 0]), "=&a"(dst[g0 + 1]), "=&a"(dst[g0 + 2]), "=&a"(dst[g0 + 3]), "=&a"(dst[g0 + 4]),
          "=&a"(dst[g0 + 5]), "=&a"(dst[g0 + 6]), "=&a"(dst[g0 + 7])
        : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]), "v"(ad[6]), "v"(ad[7])
        : "memory");
}

// WAVES: waves per workgroup (one workgroup per CU; 8 = two waves per SIMD, so
// one wave's DMA issue / epilogue runs under the other's MFMAs); NSTAGE: ring
// depth per wave; KU = d_pad/128 units per row tile; QT: 32-query tiles a wave
// scores against every row tile it streams (the A fragments are read from LDS
// once and reused QT times, so a large batch re-reads the matrix from L2 QT
// times less often; QT > 1 keeps QT*64 registers of query fragments and runs
// four waves per CU); QP: query planes used (2 = q_hi + q_lo; 1 = q_hi only in the
// multi-tile, MFMA-bound configuration: half the MFMAs for a bound of 2^-7 instead
// of 2^-8 per product, sq_dense.hip); SAMPLE: write per-lane minima of every
// visited tile instead of emitting candidates.
// AB: the query fragments of ALL k-units live in AGPRs and the MFMAs are issued from inline asm
// (every multi-tile configuration, and d_pad > 128 with more than one query tile in the batch);
// otherwise they are compiler-managed VGPRs (KU == 1) or re-read from an LDS copy per MFMA
// (KU > 1: fine while one tile per wave keeps the kernel HBM bound, LDS bound beyond that).
// NT (the AGPR configurations; compile time because their instruction stream takes no branch): non-temporal row stream
// for a launch that reads the copy once -- one group of query tiles.
template <int WAVES, int NSTAGE, int KU, int QT, int QP, bool AB, bool SAMPLE, bool NT = false>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void dense_scan_kernel(DenseScanArgs a) {
    static_assert(AB || (QT == 1 && QP == 2), "several query tiles / one query plane need the AGPR-resident fragments");
    static_assert(!AB || KU * QT * QP * 32 <= 256, "AGPR budget of the query fragments");
    constexpr bool QREG = AB || KU <= 1;  // query fragments live in registers (not re-read from LDS)
    constexpr int DPAD = KU * KT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // LDS map: [query tile (only when re-read per unit)][rings][per-wave survivor counters].
    // With QREG the query tile is staged through the not yet used ring area.
    constexpr u32 q_bytes = QREG ? 0u : (u32)TILE_ROWS * DPAD * 4;
    const u32 lds_base = (u32)(uintptr_t)smem;  // low 32 bits of a flat LDS address = LDS offset
    const u32 ring_base = lds_base + q_bytes + (u32)wave * (NSTAGE * SLOT_BYTES);
    unsigned char* ring_ptr = smem + q_bytes + wave * (NSTAGE * SLOT_BYTES);
    const long long wave_id = (long long)blockIdx.x * WAVES + wave;  // unique per wave of the launch
    uint2* wout = a.wave_out + wave_id * a.wave_cap;
    const bool add_norm = a.norms != nullptr;
    const bool nine = add_norm && KU == 1;  // DMA instructions per unit: 9 when each unit carries its norms

    // block -> (row block, group of QT query tiles); blocks that share an XCD (same id mod 8)
    // walk the query groups of the same rows so the matrix is re-read from L2.
    const int L = blockIdx.x;
    int qt, rb;  // qt: first query tile of this block's group
    if (a.nqt > 1) {
        const int xcd = L & 7, j = L >> 3;
        qt = (j % a.nqt) * QT;
        rb = (j / a.nqt) * 8 + xcd;
    } else {
        qt = 0;
        rb = L;
    }

    // stage the query tile: [32][DPAD*4 bytes] (per k-unit a hi and a lo plane of 256 bytes), 16-byte
    // chunks XOR-swizzled inside each 256-byte plane by (row & 15) -- the image the row units get from the DMA
    {
        const uint4* qsrc = a.qs + (long long)qt * TILE_ROWS * (DPAD / 4);
        constexpr int cpr = DPAD / 4;
        bool fused = false;
        if constexpr (!AB) fused = a.raw_q != nullptr;
        for (int c = threadIdx.x; c < QT * TILE_ROWS * cpr; c += WAVES * 64) {
            const int r = c / cpr, ch = c - r * cpr;
            uint4 v;
            if (fused) {
                // chunk ch of query row r: k-unit ch >> 5, plane (ch >> 4) & 1, 8 elements from 16 s + 8 h
                // (the layout and the arithmetic of dense_prep_queries_kernel, scale -2)
                const int qi = qt * TILE_ROWS + r;
                const int unit = ch >> 5, pl = (ch >> 4) & 1, cc = ch & 15;
                const int k0 = unit * KT + 16 * (cc >> 1) + 8 * (cc & 1);
                u32 w[4];
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    u32 half[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int k = k0 + j + e;
                        float x = 0.f;
                        if (qi < a.raw_nq && k < a.raw_d) {
                            const float qv = a.raw_q[(long long)qi * a.raw_d + k];
                            x = -2.0f * (a.center ? __fsub_rn(qv, a.center[k]) : qv);
                        }
                        u32 hi, lo;
                        bf16_split(x, hi, lo);
                        half[e] = pl ? lo : hi;
                    }
                    w[j >> 1] = half[0] | (half[1] << 16);
                }
                v = make_uint4(w[0], w[1], w[2], w[3]);
            } else {
                v = qsrc[c];
            }
            const int sw = (ch & ~15) | ((ch & 15) ^ (r & 15));
            *reinterpret_cast<uint4*>(smem + (u32)r * DPAD * 4 + sw * 16) = v;
        }
    }
    __syncthreads();

    // Tiles are dealt round-robin over all waves of the launch: at any moment the
    // grid reads one compact window of the matrix.
    const long long gw = (long long)rb * WAVES + wave;
    // CHUNK (the four-tile sample pass): a wave takes runs of four consecutive sampled tiles and writes its four minima
    // per (query, row half) as ONE 16-byte store -- with a tile per store the 64 lanes of a store are 64 separate
    // 4-byte writes (sample_out is query-major), and at four stores per unit the address unit, not the matrix core,
    // set the pace (10 M rows, 1024 queries: 0.48 ms against 0.2 ms of arithmetic).  Sample layout there:
    // [query][row half][n_sel rounded up to 4]; the slots behind n_sel hold +inf.
    constexpr bool CHUNK = SAMPLE && AB && WAVES == 4 && QT == 4 && QP == 1 && KU == 1;
    const long long nwaves = (long long)a.nrb * WAVES;
    long long my_tiles = gw < a.n_sel ? (a.n_sel - gw + nwaves - 1) / nwaves : 0;
    if constexpr (CHUNK) {
        const long long full = a.n_sel >> 2, rem = a.n_sel & 3;  // whole runs, tiles of the last partial run
        my_tiles = 4 * (gw < full ? (full - gw + nwaves - 1) / nwaves : 0) + ((rem && full % nwaves == gw) ? rem : 0);
    }
    const long long sel0 = CHUNK ? 4 * gw : gw;  // this wave's first tile
    // the tile after `sel` in this wave's order
    auto next_sel = [&](const long long sel) __attribute__((always_inline)) -> long long {
        if constexpr (CHUNK) return (sel & 3) == 3 ? sel + 4 * nwaves - 3 : sel + 1;
        return sel + nwaves;
    };
    const long long total_units = my_tiles * KU;

    const int r31 = lane & 31, h = lane >> 5;
    const int qglob0 = qt * TILE_ROWS + r31;  // this lane's query in tile t of the group: qglob0 + 32 t
    float thr_l[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        thr_l[t] = SAMPLE ? 0.f : a.thr[qglob0 + t * TILE_ROWS];
        // Force hipcc's wait for this load HERE.  Left to its first use inside the
        // loop the compiler emits s_waitcnt vmcnt(0) there (it cannot see the asm
        // LDS-DMAs), draining the whole ring once per tile.
        if (a.debug & 4) thr_l[t] = -__builtin_inff();  // ablation: nothing is emitted
        asm volatile("" : "+v"(thr_l[t]));
    }

    // B fragments of k-step s: chunk 2s+h of the hi plane (bq[.][2s]) and of the lo plane (bq[.][2s+1]);
    // with QREG (KU == 1) one set per query tile of the group
    // AB: bq[t][kc * 8 * QP + ...] for k-unit kc (QP == 1: entry s = hi fragment of k-step s)
    f32x4 bq[QT][AB ? KU * 8 * QP : 16];
    if constexpr (QREG) {
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            const u32 rowb = (u32)(t * TILE_ROWS + r31) * DPAD * 4;
            if constexpr (AB) {  // into AGPRs (see mfma_bf16_agpr_b)
#pragma unroll
                for (int kc = 0; kc < KU; ++kc)
#pragma unroll
                    for (int g0 = 0; g0 < 8 * QP; g0 += 8) {
                        u32 ad[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int g = QP == 2 ? g0 + j : 2 * j;  // (k-step, plane) = (g >> 1, g & 1)
                            ad[j] = lds_base + rowb + kc * 512 + (g & 1) * 256 + ((2 * (g >> 1) + h) ^ (r31 & 15)) * 16;
                        }
                        lds_read8_agpr(bq[t], kc * 8 * QP + g0, ad);
                    }
            } else {
#pragma unroll
                for (int g = 0; g < 16; ++g)
                    bq[t][g] = *reinterpret_cast<const f32x4*>(smem + rowb + (g & 1) * 256 +
                                                                ((2 * (g >> 1) + h) ^ (r31 & 15)) * 16);
            }
        }
        __syncthreads();  // every wave holds its fragments before the DMA ring overwrites the staging area
    }

    u32 wcount = 0;  // survivor entries of this wave so far (wave-uniform)
    // rows of the last, partial tile that exist (bit i <-> accumulator register i of this lane)
    u32 tail_mask = 0;
    {
        const int nvalid = (int)(a.n - (a.n_tiles - 1) * TILE_ROWS);
#pragma unroll
        for (int i = 0; i < 16; ++i) tail_mask |= ((i & 3) + 8 * (i >> 2) + 4 * h < nvalid ? 1u : 0u) << i;
    }

    // per-lane byte offsets of the 8 DMA instructions of a unit (row 4j + lane/16, swizzled chunk)
    u32 voff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = 4 * j + (lane >> 4);
        voff[j] = (u32)(r * (DPAD * 2) + (((lane & 15) ^ (r & 15)) * 16));
    }
    const u32 voff_norm = (u32)((lane & 31) * 4);

    // issue cursor: unit u of this wave = (tile gw + (u / KU) * nwaves, k-unit u % KU), slot u % NSTAGE
    long long iss_sel = sel0;
    int iss_kc = 0, iss_slot = 0;
    long long issued = 0;
    auto issue_unit = [&]() __attribute__((always_inline)) {
        const long long row0 = iss_sel * a.tile_step * TILE_ROWS;
        const u32 dst = ring_base + (u32)iss_slot * SLOT_BYTES;
        const unsigned char* base = reinterpret_cast<const unsigned char*>(a.scan) + row0 * (DPAD * 2) + iss_kc * 256;
#pragma unroll
        // non-temporal while the copy is read ONCE per pass (one group of query tiles): the stream then neither displaces
        // nor waits behind what the neighbouring kernels keep in L2 / MALL -- 10 M x 128, 32 queries: full pass 0.44 ->
        // 0.39 ms.  The first ~192 MB of the copy keep the default policy: they survive in the MALL from call to call
        // (a 1.25 M-row shard: 3 % per step).  Several query groups re-read the rows from L2: default policy there.
        // (only the compiler-scheduled one-tile kernel: a branch between the hand-issued MFMAs of the multi-tile
        // kernels upsets hipcc -- see the large-batch notes in DESIGN.md)
        for (int j = 0; j < 8; ++j) {
            if (NT || (!AB && a.nqt <= 1 && row0 >= a.nt_from_row))
                glds16<true>(base, voff[j], dst + (u32)j * 1024);
            else
                glds16<false>(base, voff[j], dst + (u32)j * 1024);
        }
        if constexpr (AB) {
            if (iss_kc == 0) glds4(a.norms + row0 * a.norm_step, voff_norm, dst + UNIT_BYTES);
        } else {
            if (add_norm && iss_kc == 0) glds4(a.norms + row0, voff_norm, dst + UNIT_BYTES);
        }
        ++issued;
        if (++iss_kc == KU) {
            iss_kc = 0;
            iss_sel = next_sel(iss_sel);
        }
        if (++iss_slot == NSTAGE) iss_slot = 0;
    };

    // Software pipeline over units (u = 0, 1, ...; slot of unit u = u % NSTAGE):
    //   registers hold the A fragments of unit u (av_cur) while its 16 MFMAs run;
    //   the fragments of unit u+1 are read from LDS (av_nxt) under those MFMAs;
    //   the slot of unit u is refilled by the DMA of unit u+NSTAGE as soon as
    //   av_cur is complete.
    const bool do_dma = AB || !(a.debug & 2), do_math = AB || !(a.debug & 1);
    auto issue_next = [&]() __attribute__((always_inline)) {
        if (issued < total_units) {
            if (do_dma || issued < NSTAGE)  // ablation: the ring is filled once, then reused
                issue_unit();
            else
                ++issued;
        }
    };
    f32x4 sb[QT];  // CHUNK: this lane's minima of the current run, per query tile
    // ---- tile complete: scores for 32 rows x QT*32 queries (lane = query, register i = row (i&3)+8(i>>2)+4h)
    // finish_q: one query tile of a row tile; PIN: the minimum is computed by instructions pinned in program order
    // the survivors of one query tile (hit: ballot of the lanes whose minimum passes): rare next to the tiles without
    auto emit_q = [&](const long long sel, auto t_c, const f32x16& sc, const u64 hit) __attribute__((always_inline)) {
        constexpr int t = decltype(t_c)::value;
        if (hit != 0) {
            const long long row0 = sel * a.tile_step * TILE_ROWS;
            const bool is_tail = row0 + TILE_ROWS > a.n;  // wave-uniform: the last, partial tile
            // survivor mask of this lane's 16 rows (bit i <-> row row0 + (i&3) + 8(i>>2) + 4h)
            u32 mask = le_mask16(sc, thr_l[t]);  // (every lane: the lanes without a survivor get 0)
            if (is_tail) mask &= tail_mask;       // drop padding rows
            if (a.debug & 16) mask = 0;           // ablation: survivors found but not recorded
            // One entry per lane that has survivors: (row0 + 4h, mask << 16 | query within the group),
            // appended to this wave's own segment with a plain store.  The position comes from a
            // ballot (no atomic, nothing the streaming loop waits for); the re-rank expands the masks.
            const u64 bal = __ballot(mask != 0);
            if (mask) {
                const u32 pos = wcount + __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u));
                if (pos < a.wave_cap) wout[pos] = make_uint2((u32)(row0 + 4 * h), (mask << 16) | (u32)(t * TILE_ROWS + r31));
            }
            wcount += (u32)__popcll(bal);
        }
    };
    // ---- tile complete: scores for 32 rows x QT*32 queries (lane = query, register i = row (i&3)+8(i>>2)+4h)
    // finish_q: one query tile of a row tile; PIN: the minimum is computed by instructions pinned in program order.
    // EMIT scan: returns the ballot of the lanes with a survivor (emit_q writes them out); SAMPLE: stores the minima.
    auto finish_q = [&](const long long sel, auto t_c, auto pin_c, const f32x16& sc, auto j_c) __attribute__((always_inline)) -> u64 {
        constexpr int t = decltype(t_c)::value;
        constexpr bool PIN = decltype(pin_c)::value;
        constexpr int J = decltype(j_c)::value;  // CHUNK: place of the tile in its run of four (compile time)
#if defined(SQ_ABL) && (SQ_ABL & 4)
        return 0;  // measurement build: no epilogue
#endif
        if constexpr (!SAMPLE) {
            float m;
            if constexpr (PIN) {
                m = min16_pinned(sc);
            } else {
                m = sc[0];
#pragma unroll
                for (int i = 1; i < 16; ++i) m = fminf(m, sc[i]);
            }
            return __ballot(m <= thr_l[t]);
        } else {
            const long long row0 = sel * a.tile_step * TILE_ROWS;
            const bool is_tail = row0 + TILE_ROWS > a.n;  // wave-uniform: the last, partial tile
            const int qglob = qglob0 + t * TILE_ROWS;
            // the minimum score of this lane's 16 rows: the score of one actual row, hence a valid
            // sample for an upper bound of the k-th smallest score (kth_threshold_f32_kernel)
            float ml = __builtin_inff();
            if (is_tail) {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if ((tail_mask >> i) & 1u) ml = fminf(ml, sc[i]);
            } else if constexpr (PIN) {
                ml = min16_pinned(sc);
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) ml = fminf(ml, sc[i]);
            }
            if constexpr (CHUNK) {
                // minima of the run so far in registers; the run's last tile (or the pass's last tile) stores them
                static_assert(J >= 0 && J < 4, "place in the run");
                if constexpr (J == 0) sb[t] = f32x4{ml, INFINITY, INFINITY, INFINITY};
                if constexpr (J > 0) sb[t][J] = ml;
                if (J == 3 || sel == a.n_sel - 1)
                    *reinterpret_cast<f32x4*>(a.sample_out + (long long)qglob * a.ns + h * (a.ns >> 1) + (sel & ~3ll)) = sb[t];
            } else {
                a.sample_out[(long long)qglob * a.ns + sel * 2 + h] = ml;
            }
            return 0;
        }
    };
    auto finish_tile = [&](const long long sel, f32x16 (&acc)[QT], auto j_c) __attribute__((always_inline)) {
        u64 hit[QT];
        hit[0] = finish_q(sel, std::integral_constant<int, 0>{}, std::false_type{}, acc[0], j_c);
        if constexpr (QT > 1) hit[1] = finish_q(sel, std::integral_constant<int, 1>{}, std::false_type{}, acc[1], j_c);
        if constexpr (QT > 2) {
            hit[2] = finish_q(sel, std::integral_constant<int, 2>{}, std::false_type{}, acc[2], j_c);
            hit[3] = finish_q(sel, std::integral_constant<int, 3>{}, std::false_type{}, acc[3], j_c);
        }
        if constexpr (!SAMPLE) {
            emit_q(sel, std::integral_constant<int, 0>{}, acc[0], hit[0]);
            if constexpr (QT > 1) emit_q(sel, std::integral_constant<int, 1>{}, acc[1], hit[1]);
            if constexpr (QT > 2) {
                emit_q(sel, std::integral_constant<int, 2>{}, acc[2], hit[2]);
                emit_q(sel, std::integral_constant<int, 3>{}, acc[3], hit[3]);
            }
        }
    };

    for (int p = 0; p < NSTAGE; ++p) issue_next();
    long long loaded = 0;  // units whose fragments have been requested from LDS
    int rd_slot = 0, rd_kc = 0;

    if constexpr (AB && WAVES == 4) {
        // ---- One wave per SIMD: nothing else hides this wave's side work, so it
        // sits in the issue shadow of the 8*QP*QT MFMAs of a unit (an MFMA holds the vector issue for 8 of its 32
        // cycles): the LDS reads of the next unit's fragments behind the first MFMAs, one of the DMA instructions
        // that refill this unit's slot behind every few MFMAs after that.  The loop is unrolled over a group of
        // G = lcm(NSTAGE, KU, 2) units, which makes everything about a unit but its data a compile-time constant:
        //  * the A fragments (and the norm block that is C of a tile's first MFMAs) alternate between two register
        //    sets -- unit u multiplies out of set u & 1 while the reads of unit u + 1 land in the other -- so no
        //    register is copied between units;
        //  * the ring slot (u % NSTAGE) is an immediate offset of the LDS reads and of the DMA destination: M0 is
        //    written once per unit (slot + 4096) and the eight 1 KiB pieces use the instruction offset, which moves
        //    the LDS and the global address alike (the per-lane global offsets carry the opposite shift);
        //  * the body is specialised on (refill?, read ahead?), and while the ring is being refilled the number
        //    of younger units in flight is the constant NSTAGE - 2.
        // The norm piece is part of every tile's first unit: a cosine scan streams 32 zeros (norm_step 0) instead.
        static_assert(NSTAGE == 2 || NSTAGE == 4, "ring depth of the AGPR configurations");
        constexpr int G0 = NSTAGE % KU == 0 ? NSTAGE : NSTAGE * KU;  // multiple of NSTAGE (even) and of KU
        constexpr int G = G0;
        static_assert(G % KU == 0 && G % NSTAGE == 0 && G % 2 == 0, "group of units");
        f32x4 fa[2][8];
        f32x16 nrm[2];
        // per-lane LDS addresses of the eight fragments of slot 0 and of the norm block
        u32 la[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            la[g] = ring_base + (u32)(r31 * 256 + ((2 * g + h) ^ (r31 & 15)) * 16);
            asm volatile("" : "+v"(la[g]));  // one register each: hipcc otherwise re-adds the wave's ring base per read
        }
        u32 ln = ring_base + (u32)(UNIT_BYTES + (4 * h) * 4);
        asm volatile("" : "+v"(ln));
        // per-lane global offsets of the eight pieces, shifted against the instruction offsets j * 1024 - 4096
        u32 voff_s[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) voff_s[j] = voff[j] + 4096u - (u32)j * 1024u;
        auto read_ab = [&](auto par_c, auto slot_c, const bool first_of_tile) __attribute__((always_inline)) {
            constexpr int P = decltype(par_c)::value, SL = decltype(slot_c)::value;
#if defined(SQ_ABL) && (SQ_ABL & 2)
            if (loaded > 1) return;  // measurement build: the fragments of the first units are reused
#endif
#pragma unroll
            for (int g = 0; g < 8; ++g) fa[P][g] = *(lds_cf32x4*)(la[g] + (u32)(SL * SLOT_BYTES));
            if (KU == 1 || first_of_tile) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f32x4 v = *(lds_cf32x4*)(ln + (u32)(SL * SLOT_BYTES + 32 * c));
                    nrm[P][4 * c + 0] = v[0];
                    nrm[P][4 * c + 1] = v[1];
                    nrm[P][4 * c + 2] = v[2];
                    nrm[P][4 * c + 3] = v[3];
                }
            }
        };
#pragma unroll
        for (int g = 0; g < 8; ++g) fa[0][g] = fa[1][g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 16; ++i) nrm[0][i] = nrm[1][i] = 0.f;
        if (total_units > 0) {
            if (KU == 1)
                wait_units_in_flight<NSTAGE, 9>((int)(issued - 1));
            else
                wait_units_in_flight<NSTAGE, 8>((int)(issued - 1));
            read_ab(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, true);
            loaded = 1;
            rd_kc = KU > 1 ? 1 : 0;
        }
        // issue cursor as running pointers (the tile of the next unit to issue)
        const long long tile_adv = nwaves * a.tile_step * TILE_ROWS;                       // rows between a wave's tiles
        const unsigned char* iss_ptr = reinterpret_cast<const unsigned char*>(a.scan) + iss_sel * a.tile_step * TILE_ROWS * (DPAD * 2);
        const float* iss_nptr = a.norms + iss_sel * a.tile_step * TILE_ROWS * a.norm_step;
        const long long iss_ptr_adv = tile_adv * (DPAD * 2), iss_nptr_adv = tile_adv * a.norm_step;
        // CHUNK: one tile on inside a run, (4 nwaves - 3) tiles on from a run's last tile
        const long long step1_rows = a.tile_step * TILE_ROWS;
        const long long run_ptr_adv = step1_rows * (DPAD * 2), run_nptr_adv = step1_rows * a.norm_step;
        const long long jump_ptr_adv = (4 * nwaves - 3) * run_ptr_adv, jump_nptr_adv = (4 * nwaves - 3) * run_nptr_adv;
        // SKEW (four query tiles, one k-unit, the emitting pass): the MFMAs of a unit run as two halves -- tiles 0,1 then tiles 2,3 (the A
        // fragments stay in registers for the whole unit, so the order is free) -- and the epilogue of a half sits
        // between the MFMAs of the next one (tiles 0,1 of this unit under its tiles 2,3; tiles 2,3 under tiles 0,1 of
        // the following unit): with one wave per SIMD nothing else would keep the matrix pipe busy during the ~50-100
        // vector instructions of the four epilogues.  An epilogue starts four MFMAs (128 cycles) after the last MFMA
        // of its accumulators, which also covers the MFMA -> VALU read hazard hipcc does not know about.
#ifdef SQ_NO_SKEW
        constexpr bool SKEW = false;  // measurement build
#else
        constexpr bool SKEW = QT == 4 && QP == 1 && KU == 1 && !SAMPLE;
#endif
        long long prev_sel = 0;
        bool prev_valid = false;
        auto unit = [&](auto par_c, auto slot_c, auto iss_c, auto rd_c, const int kc, f32x16 (&acc)[QT], const long long cur_sel) __attribute__((always_inline)) {
            constexpr int P = decltype(par_c)::value, SL = decltype(slot_c)::value;
            constexpr bool ISS = decltype(iss_c)::value, RD = decltype(rd_c)::value;
            const int units_behind = ISS ? NSTAGE - 2 : (int)(issued - loaded - 1);  // younger units that may stay in flight
            const u32 idst = ring_base + (u32)SL * SLOT_BYTES;                       // the slot this unit leaves is refilled
            const unsigned char* ibase = iss_ptr + iss_kc * 256;
            const bool inorm = iss_kc == 0;
            const bool rd_first = rd_kc == 0;
            if (kc == 0)
                mfma_fence_in_c(nrm[P], fa[P]);
            else
                mfma_fence_in<QT>(acc, fa[P]);
            constexpr int NSLOT = 8 * QP * QT;             // one slot behind every MFMA
            constexpr int DMA0 = NSLOT >= 16 ? 6 : 2, DSTEP = (NSLOT - DMA0 - 1) / 9 > 0 ? (NSLOT - DMA0 - 1) / 9 : 1;
            u64 hit0 = 0, hit1 = 0, hit2 = 0, hit3 = 0;
            auto dma_piece = [&](auto j_c) __attribute__((always_inline)) {
                constexpr int J = decltype(j_c)::value;
#if defined(SQ_ABL) && (SQ_ABL & 1)
                return;  // measurement build: no refill
#endif
                if constexpr (J == 0) glds_set_m0(idst + 4096u);
                if constexpr (J < 8) glds16_m0<J * 1024 - 4096, NT>(ibase, voff_s[J]);
                if constexpr (J == 8) {
                    if (inorm) {
                        glds_set_m0(idst + UNIT_BYTES);
                        glds4_m0<0>(iss_nptr, voff_norm);
                    }
                }
            };
#pragma unroll
            for (int m = 0; m < NSLOT; ++m) {
                // MFMA m of the unit: k-step s, plane half (two planes: lo first, then hi), query tile t
                const int s = SKEW ? (m & 15) >> 1 : m / (QP * QT);
                const int half = SKEW ? 0 : (m / QT) % QP;
                const int t = SKEW ? 2 * (m >> 4) + (m & 1) : m % QT;
                const int frag = kc * 8 * QP + (QP == 2 ? 2 * s + 1 - half : s);
                if (kc == 0 && s == 0 && half == 0)
                    mfma_bf16_agpr_b_first(acc[t], fa[P][0], bq[t][frag], nrm[P]);
                else
                    mfma_bf16_agpr_b(acc[t], fa[P][s], bq[t][frag]);
                const int slot = m;
                if constexpr (RD) {
                    if (slot == 1 || (NSLOT == 1)) {
                        if (KU == 1)
                            wait_units_in_flight<NSTAGE, 9>(units_behind);
                        else
                            wait_units_in_flight<NSTAGE, 8>(units_behind);
                        read_ab(std::integral_constant<int, P ^ 1>{}, std::integral_constant<int, (SL + 1) % NSTAGE>{}, rd_first);
                    }
                }
                if constexpr (SKEW) {
                    // minima + ballots in the shadow of the MFMAs; the (rare) survivors are written out where the
                    // MFMA stream has a seam anyway: before the half that overwrites their accumulators
                    if (slot == 3 && prev_valid) hit2 = finish_q(prev_sel, std::integral_constant<int, 2>{}, std::true_type{}, acc[2], std::integral_constant<int, (SL + 3) % 4>{});
                    if (slot == 9 && prev_valid) hit3 = finish_q(prev_sel, std::integral_constant<int, 3>{}, std::true_type{}, acc[3], std::integral_constant<int, (SL + 3) % 4>{});
                    if (slot == 15 && prev_valid) {
                        if constexpr (!SAMPLE) {
                            emit_q(prev_sel, std::integral_constant<int, 2>{}, acc[2], hit2);
                            emit_q(prev_sel, std::integral_constant<int, 3>{}, acc[3], hit3);
                        }
                    }
                    if (slot == 19) hit0 = finish_q(cur_sel, std::integral_constant<int, 0>{}, std::true_type{}, acc[0], std::integral_constant<int, SL % 4>{});
                    if (slot == 25) hit1 = finish_q(cur_sel, std::integral_constant<int, 1>{}, std::true_type{}, acc[1], std::integral_constant<int, SL % 4>{});
                    if (slot == 31) {
                        if constexpr (!SAMPLE) {
                            emit_q(cur_sel, std::integral_constant<int, 0>{}, acc[0], hit0);
                            emit_q(cur_sel, std::integral_constant<int, 1>{}, acc[1], hit1);
                        }
                    }
                }
                if constexpr (ISS) {
                    if (slot >= DMA0 && (slot - DMA0) % DSTEP == 0) {
                        const int j = (slot - DMA0) / DSTEP;
                        static_for_one<9>(j, dma_piece);
                    }
                    // fewer slots than DMA pieces (8 MFMAs per unit): the rest goes behind the last MFMA
                    if (slot == NSLOT - 1) {
#pragma unroll
                        for (int j = (NSLOT - 1 - DMA0) / DSTEP + 1; j < 9; ++j) static_for_one<9>(j, dma_piece);
                    }
                }
            }
            if constexpr (RD) {
                ++loaded;
                if (++rd_kc == KU) rd_kc = 0;
            }
            if constexpr (ISS) {
                ++issued;
                if (++iss_kc == KU) {
                    iss_kc = 0;
                    if constexpr (CHUNK) {  // the unit just issued has this unit's place in its run (NSTAGE == 4 == run)
                        iss_ptr += (SL == 3) ? jump_ptr_adv : run_ptr_adv;
                        iss_nptr += (SL == 3) ? jump_nptr_adv : run_nptr_adv;
                    } else {
                        iss_ptr += iss_ptr_adv;
                        iss_nptr += iss_nptr_adv;
                    }
                }
            }
            if constexpr (!SKEW) mfma_fence_out<QT>(acc);
        };
        // G units with compile-time (register set, slot, k-unit).  STEADY: every unit of the group refills its slot
        // (straight-line code: a three-way branch per unit makes hipcc merge the accumulators of its arms with
        // register copies); otherwise each unit picks its variant and the group ends at a tile boundary once the
        // tiles are done.
        long long sel = sel0;
        f32x16 acc[QT];
        auto group = [&](auto&& self, auto u_c, auto steady_c) __attribute__((always_inline)) -> void {
            constexpr int U = decltype(u_c)::value;
            constexpr bool STEADY = decltype(steady_c)::value;
            if constexpr (U < G) {
                constexpr int KC = U % KU;
                constexpr std::integral_constant<int, U & 1> par{};
                constexpr std::integral_constant<int, U % NSTAGE> sl{};
                if constexpr (STEADY) {
                    unit(par, sl, std::true_type{}, std::true_type{}, KC, acc, sel);
                } else {
                    if (KC == 0 && sel >= a.n_sel) return;
                    const bool rd = loaded < total_units, iss = issued < total_units;  // iss implies rd
                    if (iss)
                        unit(par, sl, std::true_type{}, std::true_type{}, KC, acc, sel);
                    else if (rd)
                        unit(par, sl, std::false_type{}, std::true_type{}, KC, acc, sel);
                    else
                        unit(par, sl, std::false_type{}, std::false_type{}, KC, acc, sel);
                    // hipcc may merge the accumulators of the three arms with register copies: not before the MFMAs are done
                    if constexpr (SKEW) mfma_fence_out<QT>(acc);
                }
                if constexpr (SKEW) {
                    prev_sel = sel;
                    prev_valid = true;
                    sel = next_sel(sel);
                } else if constexpr (KC == KU - 1) {
                    finish_tile(sel, acc, std::integral_constant<int, U % 4>{});  // (U % 4: CHUNK's place in the run)
                    sel = next_sel(sel);
                }
                self(self, std::integral_constant<int, U + 1>{}, steady_c);
            }
        };
        while (issued + G <= total_units) {
            group(group, std::integral_constant<int, 0>{}, std::true_type{});
            if constexpr (SKEW) mfma_fence_out<QT>(acc);  // (a loop-carried accumulator copy must not overtake the MFMAs)
        }
        while (sel < a.n_sel) group(group, std::integral_constant<int, 0>{}, std::false_type{});
        if constexpr (SKEW) {
            if (prev_valid) {  // the last unit's tiles 2,3
                mfma_fence_out<QT>(acc);
                u64 h2 = 0, h3 = 0;
                auto last = [&](auto j_c) __attribute__((always_inline)) {
                    h2 = finish_q(prev_sel, std::integral_constant<int, 2>{}, std::false_type{}, acc[2], j_c);
                    h3 = finish_q(prev_sel, std::integral_constant<int, 3>{}, std::false_type{}, acc[3], j_c);
                };
                if constexpr (CHUNK) {
                    static_for_one<4>((int)(prev_sel & 3), last);
                } else {
                    last(std::integral_constant<int, 0>{});
                }
                if constexpr (!SAMPLE) {
                    emit_q(prev_sel, std::integral_constant<int, 2>{}, acc[2], h2);
                    emit_q(prev_sel, std::integral_constant<int, 3>{}, acc[3], h3);
                }
            }
        }
    } else {
        f32x4 av_cur[8], av_nxt[8], nrm_nxt[4];
        auto read_frags = [&](int slot_idx, bool first_of_tile) __attribute__((always_inline)) {
            const unsigned char* sl = ring_ptr + slot_idx * SLOT_BYTES;
            const unsigned char* arow = sl + r31 * 256;
#pragma unroll
            for (int g = 0; g < 8; ++g)
                av_nxt[g] = *reinterpret_cast<const f32x4*>(arow + ((2 * g + h) ^ (r31 & 15)) * 16);
            if (first_of_tile && add_norm) {
#pragma unroll
                for (int c = 0; c < 4; ++c) nrm_nxt[c] = *reinterpret_cast<const f32x4*>(sl + UNIT_BYTES + (8 * c + 4 * h) * 4);
            }
        };
#pragma unroll
        for (int g = 0; g < 8; ++g) av_nxt[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) nrm_nxt[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (total_units > 0) {
            if (nine)
                wait_units_in_flight<NSTAGE, 9>((int)(issued - 1));
            else
                wait_units_in_flight<NSTAGE, 8>((int)(issued - 1));
            if (do_math) read_frags(0, true);
            loaded = 1;
            rd_slot = NSTAGE > 1 ? 1 : 0;
            rd_kc = KU > 1 ? 1 : 0;
        }
        for (long long sel = gw; sel < a.n_sel; sel += nwaves) {
            f32x16 acc[QT];
            if constexpr (AB) {
                // ---- Two waves per SIMD, 128 + 128 registers (the two register sets of the one-wave path do not fit):
                // the fragments of the next unit are copied into place.  As there: nothing else hides this wave's
                // side work, so it sits in the issue shadow of the 8*QP*QT MFMAs of a unit (an MFMA holds the
                // vector issue for 8 of its 32 cycles): the LDS reads of the next unit's fragments behind
                // the first MFMAs, one of the DMA instructions that refill this unit's slot behind every
                // few MFMAs after that.  The body is specialised on (refill?, read ahead?) so the block is
                // free of branches; the measurement-only ablations (debug 1 / 2) are not available here.
                auto unit = [&](auto iss_c, auto rd_c, const int kc) __attribute__((always_inline)) {
                    constexpr bool ISS = decltype(iss_c)::value, RD = decltype(rd_c)::value;
#pragma unroll
                    for (int g = 0; g < 8; ++g) av_cur[g] = av_nxt[g];
                    f32x16 nrm_c;
                    if (kc == 0) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) nrm_c[i] = add_norm ? nrm_nxt[i >> 2][i & 3] : 0.f;
                    }
                    asm volatile("" ::: "memory");
                    const int units_behind = (int)(issued - loaded - 1);  // younger units that may stay in flight
                    const long long irow0 = iss_sel * a.tile_step * TILE_ROWS;
                    const u32 idst = ring_base + (u32)iss_slot * SLOT_BYTES;
                    const unsigned char* ibase = reinterpret_cast<const unsigned char*>(a.scan) + irow0 * (DPAD * 2) + iss_kc * 256;
                    const bool inorm = add_norm && iss_kc == 0;
                    const bool rd_first = rd_kc == 0;
                    if (kc == 0)
                        mfma_fence_in_c(nrm_c, av_cur);
                    else
                        mfma_fence_in<QT>(acc, av_cur);
                    constexpr int NSLOT = 8 * QP * QT;             // one slot behind every MFMA
                    constexpr int DMA0 = NSLOT >= 16 ? 6 : 2, DSTEP = (NSLOT - DMA0 - 1) / 9 > 0 ? (NSLOT - DMA0 - 1) / 9 : 1;
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
#pragma unroll
                        for (int half = 0; half < QP; ++half) {  // two planes: lo first, then hi
#pragma unroll
                            for (int t = 0; t < QT; ++t) {
                                const int frag = kc * 8 * QP + (QP == 2 ? 2 * s + 1 - half : s);
                                if (kc == 0 && s == 0 && half == 0)
                                    mfma_bf16_agpr_b_first(acc[t], av_cur[0], bq[t][frag], nrm_c);
                                else
                                    mfma_bf16_agpr_b(acc[t], av_cur[s], bq[t][frag]);
                                const int slot = (s * QP + half) * QT + t;
                                if constexpr (RD) {
                                    if (slot == 1 || (NSLOT == 1)) {
                                        if (nine)
                                            wait_units_in_flight<NSTAGE, 9>(units_behind);
                                        else
                                            wait_units_in_flight<NSTAGE, 8>(units_behind);
                                        read_frags(rd_slot, rd_first);
                                    }
                                }
                                if constexpr (ISS) {
                                    if (slot >= DMA0 && (slot - DMA0) % DSTEP == 0) {
                                        const int j = (slot - DMA0) / DSTEP;
                                        if (j < 8) glds16<NT>(ibase, voff[j], idst + (u32)j * 1024);
                                        if (j == 8 && inorm) glds4(a.norms + irow0 * a.norm_step, voff_norm, idst + UNIT_BYTES);
                                    }
                                    // fewer slots than DMA pieces (8 MFMAs per unit): the rest goes behind the last MFMA
                                    if (slot == NSLOT - 1) {
#pragma unroll
                                        for (int j = (NSLOT - 1 - DMA0) / DSTEP + 1; j < 9; ++j) {
                                            if (j < 8) glds16<NT>(ibase, voff[j], idst + (u32)j * 1024);
                                            if (j == 8 && inorm) glds4(a.norms + irow0 * a.norm_step, voff_norm, idst + UNIT_BYTES);
                                        }
                                    }
                                }
                            }
                        }
                    }
                    if constexpr (RD) {
                        ++loaded;
                        if (++rd_slot == NSTAGE) rd_slot = 0;
                        if (++rd_kc == KU) rd_kc = 0;
                    }
                    if constexpr (ISS) {
                        ++issued;
                        if (++iss_kc == KU) {
                            iss_kc = 0;
                            iss_sel += nwaves;
                        }
                        if (++iss_slot == NSTAGE) iss_slot = 0;
                    }
                    mfma_fence_out<QT>(acc);
                };
#pragma unroll
                for (int kc = 0; kc < KU; ++kc) {
                    const bool rd = loaded < total_units, iss = issued < total_units;  // iss implies rd
                    if (iss)
                        unit(std::true_type{}, std::true_type{}, kc);
                    else if (rd)
                        unit(std::false_type{}, std::true_type{}, kc);
                    else
                        unit(std::false_type{}, std::false_type{}, kc);
                }
            } else {
#pragma unroll
                for (int kc = 0; kc < KU; ++kc) {
                    // fragments of this unit are complete once copied (hipcc waits lgkmcnt here); its slot is free
#pragma unroll
                    for (int g = 0; g < 8; ++g) av_cur[g] = av_nxt[g];
                    if (kc == 0) {
                        // the accumulator starts from |x|^2 of the tile's rows (0 for cosine): score = |x|^2 + x.q'
#pragma unroll
                        for (int i = 0; i < 16; ++i) acc[0][i] = add_norm ? nrm_nxt[i >> 2][i & 3] : 0.f;
                    }
                    asm volatile("" ::: "memory");
                    issue_next();
                    if (loaded < total_units) {
                        if (nine)  // younger units may stay in flight
                            wait_units_in_flight<NSTAGE, 9>((int)(issued - loaded - 1));
                        else
                            wait_units_in_flight<NSTAGE, 8>((int)(issued - loaded - 1));
                        if (do_math) read_frags(rd_slot, rd_kc == 0);
                        ++loaded;
                        if (++rd_slot == NSTAGE) rd_slot = 0;
                        if (++rd_kc == KU) rd_kc = 0;
                    }
                    if (do_math) {
#pragma unroll
                        for (int s = 0; s < 8; ++s) {
                            const bf16x8 ah = __builtin_bit_cast(bf16x8, av_cur[s]);
                            f32x4 bh, bl;
                            if constexpr (QREG) {
                                bh = bq[0][2 * s];
                                bl = bq[0][2 * s + 1];
                            } else {
                                const unsigned char* brow = smem + (u32)r31 * DPAD * 4 + kc * 512 + ((2 * s + h) ^ (r31 & 15)) * 16;
                                bh = *reinterpret_cast<const f32x4*>(brow);
                                bl = *reinterpret_cast<const f32x4*>(brow + 256);
                            }
                            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, __builtin_bit_cast(bf16x8, bl), acc[0], 0, 0, 0);
                            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, __builtin_bit_cast(bf16x8, bh), acc[0], 0, 0, 0);
                        }
                    }
                }
            }
            if (do_math) finish_tile(sel, acc, std::integral_constant<int, 0>{});
        }
    }
    if constexpr (!SAMPLE) {
        if (lane == 0) {
            a.wave_cnt[2 * wave_id] = wcount;        // entries written (beyond wave_cap: overflow)
            a.wave_cnt[2 * wave_id + 1] = (u32)qt;   // first query tile of this wave's group
        }
    }
}

}  // namespace sq
