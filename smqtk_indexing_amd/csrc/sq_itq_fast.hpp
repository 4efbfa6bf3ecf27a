// ITQ hash codes at the HBM rate: a certified half-precision filter (f16x3) in
// front of the float64 kernel (sq_itq.hip).
//
// ItqFunctor.get_hash (smqtk_indexing/impls/lsh_functor/itq.py:389-408) only
// keeps the SIGN of z = (v - mean) . R, v = x or x/|x|, but evaluates z in
// float64; on the f64 matrix cores that is 2*n*d*bits flop (10 M x 128 -> 64
// bits: 3.9 ms) under a 0.85 ms HBM floor.  The sign of z is known as soon as
// |z~| exceeds the error bound of a cheaper evaluation z~, so this kernel
// streams the float32 rows once (the dense scan's LDS-DMA ring), splits x and R
// into float16 pairs (x = x_hi + x_lo + dx), evaluates
// x_hi R_hi + x_hi R_lo + x_lo R_hi on v_mfma_f32_32x32x16_f16 and finishes with
// z~ = (x . R_b) s - c_b, where s = 1/|x| (normalize=2, |x| from the same pass)
// or 1 and c_b = mean . R_b is a float64 product rounded once.  Bits with
// |z~| > eps(tile, column) are final; the others are listed as (row, column
// tile, mask) entries and evaluated in float64 one bit at a time
// (itq_fix_bits_kernel, sq_itq.hip), so every code is exactly what the float64
// evaluation gives.
//
// Why float16 planes (round 1 used bfloat16).  The kernel is bound by vector
// instruction issue, not by the matrix cores (PMC, profiles/r02_itq_pmc_*.json:
// the DMA skeleton alone streams the matrix in 0.85 ms, removing every MFMA
// changes nothing).  A bfloat16 split costs 3 vector instructions per element
// (convert, widen the high part back, subtract, convert); with float16 the
// residual x - x_hi is ONE v_fma_mix_f32 reading the packed half directly:
// 2 per element (v_cvt_pkrtz_f16_f32 for two highs, two v_fma_mix_f32,
// v_cvt_pkrtz_f16_f32 for two lows).  And 11 + 11 significand bits leave
// 2^-20 |x_k| behind instead of 2^-16: the error bound is then mostly the
// float32 accumulation, 2.8x fewer bits stay undecided.  The price is range:
// see "Range" below.
//
// Error bound, per column b, all terms through Cauchy-Schwarz:
//   x:  round-toward-zero to float16 twice: |dx_k| < 2^-20 |x_k| + 2^-24
//       (the absolute part: float16 subnormals) -> 2^-20 |x||R_b| + 2^-24 |R_b|_1
//   R:  its two planes are fixed, so the prep kernel measures the residual
//       exactly: |x| * |R_b - R_hi,b - R_lo,b|_2   (`rres`, part of colnorm)
//   dropped x_lo R_lo: |x_lo| <= 2^-10 |x| (+ abs), |R_lo,b| <= 2^-11 |R_b|: 2^-21 |x||R_b|
//   float32 accumulation of 3d products: 3d 2^-24 |x||R_b|
//   the reference's float32 x/|x| and the float32 scale / subtract here: 2^-20;
//   normalize=2: s within 2^-18 (float32 sum of squares)
//   c_b: its float32 rounding and float64 summation error (cberr); without
//   normalisation c_b also rides through the float32 accumulation.
// Range.  |x_k| >= 65504 would saturate a plane silently, so a tile whose
// largest |x|^2 is not below 1e9 sends all its bits to the float64 kernel
// (as does a NaN / inf, through the NaN-safe undecided test); values below
// 2^-14 lose relative precision to the absolute term above: data of tiny
// scale (|x| << 1e-2) stays correct but leaves more bits to float64.
#pragma once
#include "sq_dma.hpp"

namespace sq {

typedef float itq_f32x4 __attribute__((ext_vector_type(4)));
typedef float itq_f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 itq_f16x8 __attribute__((ext_vector_type(8)));
typedef u32 itq_u32x4 __attribute__((ext_vector_type(4)));

static constexpr int ITQF_UNIT_BYTES = 32 * 256;                // 32 rows x 64 floats
static constexpr int ITQF_WAVES_LDSB = 4;  // waves when the R fragments come from LDS
static constexpr int ITQF_WAVES_BREG = 8;  // waves when they live in registers
static constexpr int ITQF_MAX_CT = 4;                           // up to 128 padded hash bits

// Ballot words into the lanes of their rows.  d0[lane_a] = a0, d0[lane_b] = b0, d1[lane_a] = a1, d1[lane_b] = b1 (the
// values are wave-uniform SGPRs -- the two halves of two ballots -- the other lanes keep what they hold): four
// v_writelane_b32 with immediate lane numbers.  gfx950 needs two wait states between a VALU instruction that writes
// an SGPR (the v_cmp behind a ballot) and a VALU instruction that reads it; hipcc pads that for its own code but not
// inside an asm string, so the block starts with its own s_nop 1.  (Without it the first v_writelane read the
// PREVIOUS ballot: wrong codes on the GPU, caught by test_itq_10m_x_128_codes_equal_float64_torch.)
__device__ __forceinline__ void write_lanes_2x2(u32& d0, u32& d1, u32 a0, u32 b0, u32 a1, u32 b1, int lane_a, int lane_b) {
    asm volatile(
        "s_nop 1\n\t"
        "v_writelane_b32 %0, %2, %6\n\t"
        "v_writelane_b32 %0, %3, %7\n\t"
        "v_writelane_b32 %1, %4, %6\n\t"
        "v_writelane_b32 %1, %5, %7"
        : "+v"(d0), "+v"(d1)
        : "s"(a0), "s"(b0), "s"(a1), "s"(b1), "n"(lane_a), "n"(lane_b));
}
__device__ __forceinline__ void write_lanes_1x2(u32& d0, u32 a0, u32 b0, int lane_a, int lane_b) {
    asm volatile(
        "s_nop 1\n\t"
        "v_writelane_b32 %0, %1, %3\n\t"
        "v_writelane_b32 %0, %2, %4"
        : "+v"(d0)
        : "s"(a0), "s"(b0), "n"(lane_a), "n"(lane_b));
}

// (x0, x1) -> packed float16 highs (round toward zero) and packed float16 lows of the exact residuals:
// 4 vector instructions for two elements.  v_fma_mix_f32 reads the packed half directly: lo = hi * -1.0 + x.
#ifndef SQ_ITQ_MIX
#define SQ_ITQ_MIX 1
#endif
#ifndef SQ_ITQ_GRAM
#define SQ_ITQ_GRAM 1
#endif
__device__ __forceinline__ void split_f16_pair(float x0, float x1, u32& hi, u32& lo) {
    hi = __builtin_bit_cast(u32, __builtin_amdgcn_cvt_pkrtz(x0, x1));
    float l0, l1;
#if SQ_ITQ_MIX
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hi), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hi), "v"(x1));
#else
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    const h2_t hv = __builtin_bit_cast(h2_t, hi);
    l0 = x0 - (float)hv[0];
    l1 = x1 - (float)hv[1];
#endif
    lo = __builtin_bit_cast(u32, __builtin_amdgcn_cvt_pkrtz(l0, l1));
}

struct ItqFastArgs {
    const float* x;        // [n][d] float32 rows, d % 64 == 0, 16-byte aligned
    long long n;
    int d;
    const uint4* rimage;   // image of R: [pc columns][2 planes][dp*2 bytes], dp = d rounded up to 128; chunks swizzled by column & 15
    const float* colnorm;  // [pc] eps_rel |R_b|_2 + |R_b - R_hi,b - R_lo,b|_2, rounded up: times |x|; pad columns 0
    const float* cabs;     // [pc] 2^-24 |R_b|_1: the absolute (float16 subnormal) part of the x split
    const float* cb32;     // [pc] mean . R_b in float32 (normalize=2 form)
    const float* cberr;    // [pc] bound of what cb32 and the float32 division leave out
    u64* out;              // [n][words]
    int words, pad, bits;  // pad = words*64 - bits leading zero columns
    u64* seg;              // [waves of the launch][seg_cap] undecided (row | column tile << 30) << 32 | 32-column mask
    u32* seg_cnt;          // [waves of the launch]
    u64* seg_dummy;        // [waves of the launch] sink of the always-issued entry store (see STORES_PER_TILE)
    long long seg_cap;
    long long n_tiles;
    int nrb;               // workgroups
    int nstage;
};

// R (float64 [d][bits]) -> bfloat16 hi/lo planes, transposed to one row of d values per padded
// column and laid out exactly as the filter kernel reads it; column norms; c_b = mean . R_b.
// One workgroup per padded column.
static __global__ __launch_bounds__(256) void itq_fast_prep_kernel(const double* __restrict__ mean,
                                                                    const double* __restrict__ rot, int d, int bits,
                                                                    int pad,
                                                                    unsigned short* __restrict__ rimage,
                                                                    float* __restrict__ colnorm,
                                                                    float* __restrict__ cb32,
                                                                    float* __restrict__ cberr,
                                                                    double* __restrict__ rt64, double eps_rel,
                                                                    float* __restrict__ cabs_out) {
    const int pc = blockIdx.x;
    const int b = pc - pad;
    __shared__ double red[256];
    __shared__ double red2[256];
    __shared__ double red3[256];
    __shared__ double red4[256];
    __shared__ int s_big;
    if (threadIdx.x == 0) s_big = 0;
    __syncthreads();
    double acc = 0.0, cacc = 0.0, cabs = 0.0, macc = 0.0, rres = 0.0, rl1 = 0.0;
    for (int k = threadIdx.x; k < d; k += 256) {
        const double r = b >= 0 ? rot[(long long)k * bits + b] : 0.0;
        rt64[(long long)pc * d + k] = r;  // column-major float64 copy: the per-bit float64 evaluation reads whole columns
        acc += r * r;
        cacc += mean[k] * r;
        cabs += fabs(mean[k] * r);
        macc += mean[k] * mean[k];
        const float rf = (float)r;
        const _Float16 hi = (_Float16)rf;                 // round to nearest
        const _Float16 lo = (_Float16)(rf - (float)hi);
        const double dr = r - (double)(float)hi - (double)(float)lo;   // what the two planes leave of R (exact)
        rres += dr * dr;
        rl1 += fabs(r);
        if (!(fabs(r) < 60000.0)) s_big = 1;              // beyond float16: the column decides nothing
        // element k of column pc: 256-byte segment k/128, chunk 2*((k%128)/16) + ((k%16)/8), swizzled
        const int seg = k >> 7, kk = k & 127;
        const int chunk = ((2 * (kk >> 4) + ((kk >> 3) & 1)) ^ (pc & 15));
        const int dp = (d + 127) / 128 * 128;             // plane stride: whole 256-byte segments (the swizzle permutes
                                                          // chunks inside a segment)
        const long long base = ((long long)pc * 2) * dp;  // in bf16 elements
        const long long at = (long long)seg * 128 + chunk * 8 + (kk & 7);
        rimage[base + at] = __builtin_bit_cast(unsigned short, hi);
        rimage[base + dp + at] = __builtin_bit_cast(unsigned short, lo);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    const double rnorm = sqrt(red[0]);
    __syncthreads();
    red[threadIdx.x] = rres;
    red4[threadIdx.x] = rl1;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            red[threadIdx.x] += red[threadIdx.x + o];
            red4[threadIdx.x] += red4[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        colnorm[pc] = s_big ? __builtin_inff() : (float)((rnorm * eps_rel + sqrt(red[0])) * (1.0 + 1e-6));
        cabs_out[pc] = (float)(red4[0] * 5.9604644775390625e-08 * (1.0 + 1e-6));
    }
    __syncthreads();
    red[threadIdx.x] = cacc;
    red2[threadIdx.x] = cabs;
    red3[threadIdx.x] = macc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            red[threadIdx.x] += red[threadIdx.x + o];
            red2[threadIdx.x] += red2[threadIdx.x + o];
            red3[threadIdx.x] += red3[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // c_b = mean . R_b: the float32 value the filter subtracts, and a bound for its rounding, the
        // float64 summation and the reference's float32 element-wise x/|x| (2^-22 of the terms)
        cb32[pc] = (float)red[0];
        // ... and, when numpy subtracts the mean in float32 (float32 model), the rounding of x_k - mean_k:
        // <= 2^-24 (|x_k| + |mean_k|) per element; the |mean| part is carried here, the |x| part by eps_rel
        cberr[pc] = (float)((fabs(red[0]) * 1.2e-7 + red2[0] * 1e-13 + sqrt(red3[0]) * rnorm * 1.2e-7) * (1.0 + 1e-6));
    }
    __syncthreads();
}

// WAVES per workgroup, NSTAGE ring slots per wave, KU = d/64 units per 32-row tile, CT = padded
// hash bits / 32 column tiles.  The kernel accumulates x . R on the raw rows and finishes with
// z~ = (x . R_b) * s - mean . R_b, s = 1/|x| of the row for normalize=2 (NORMED; |x| from the same
// pass) and 1 otherwise: the rows are read once, there is no per-element mean or scale work, and
// mean . R_b is an exact float64 product rounded once.  BREG: the R fragments live in registers
// (KU * CT <= 4), which leaves the LDS to eight waves' rings; otherwise they are read from an LDS
// image and four waves run.
template <int WAVES, int NSTAGE, int KU, int CT, bool NORMED, bool BREG>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void itq_fast_kernel(ItqFastArgs a) {
    constexpr int D = KU * 64;
    constexpr int DP = (D + 127) / 128 * 128;  // plane stride of the R image
    constexpr int PC = CT * 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS: [R image][rings].  Not BREG: both bfloat16 planes of every column, PC * 2 * DP*2 bytes, as in global
    // memory.  BREG: only the LO planes, PC * DP*2 bytes (16 KB at d = 128 -> 64 bits); the HI fragments live in
    // registers.  (Both planes in registers -- 128 VGPRs at KU * CT = 4 -- left the compiler 9 registers short at
    // two waves per SIMD, and the scratch reload it then placed in the loop waits vmcnt(0): the DMA ring drained
    // once per tile, 38 % of the wave cycles parked on it.)
    constexpr u32 r_bytes = BREG ? (u32)PC * DP * 2u : (u32)PC * 2u * DP * 2u;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const u32 lds_base = (u32)(uintptr_t)smem;
    const u32 ring_base = lds_base + r_bytes + (u32)wave * (NSTAGE * ITQF_UNIT_BYTES);
    const unsigned char* ring_ptr = smem + r_bytes + wave * (NSTAGE * ITQF_UNIT_BYTES);
    const int r31 = lane & 31, h = lane >> 5;

    // byte offset of the fragment of global k-step ks, column pc, inside the R image (plane 0; plane 1 at + DP*2)
    auto b_off = [&](int ks, int pc) {
        return (u32)pc * 2u * (DP * 2) + (u32)(ks >> 3) * 256u + (u32)(((2 * (ks & 7) + h) ^ (pc & 15)) * 16);
    };
    // BREG: offset of the LO fragment inside the LDS copy of the lo planes ([pc][DP*2 bytes])
    auto blo_off = [&](int ks, int pc) {
        return (u32)pc * (DP * 2) + (u32)(ks >> 3) * 256u + (u32)(((2 * (ks & 7) + h) ^ (pc & 15)) * 16);
    };
    itq_f16x8 breg[BREG ? KU * 4 : 1][CT];   // BREG: the HI fragments of every k-step
    if constexpr (BREG) {
        const unsigned char* img = reinterpret_cast<const unsigned char*>(a.rimage);
#pragma unroll
        for (int ks = 0; ks < KU * 4; ++ks)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) breg[ks][ct] = *reinterpret_cast<const itq_f16x8*>(img + b_off(ks, ct * 32 + r31));
        // lo planes -> LDS: 16-byte chunk i of the copy = chunk (i % cpp) of column (i / cpp)'s second plane
        constexpr u32 cpp = DP * 2 / 16;  // chunks per plane
        const uint4* src = a.rimage;
        for (u32 i = threadIdx.x; i < r_bytes / 16; i += WAVES * 64)
            reinterpret_cast<uint4*>(smem)[i] = src[(i / cpp) * (2 * cpp) + cpp + (i % cpp)];
        __syncthreads();
    } else {
        const uint4* src = a.rimage;
        for (u32 i = threadIdx.x; i < r_bytes / 16; i += WAVES * 64) reinterpret_cast<uint4*>(smem)[i] = src[i];
        __syncthreads();
    }

    const long long wave_id = (long long)blockIdx.x * WAVES + wave;
    const long long nwaves = (long long)a.nrb * WAVES;
    const long long my_tiles = wave_id < a.n_tiles ? (a.n_tiles - wave_id + nwaves - 1) / nwaves : 0;
    const long long total_units = my_tiles * KU;
    u64* myseg = a.seg + wave_id * a.seg_cap;

    // lane L < 32 finalises row L of a tile (the sign / undecided words are dropped into it with v_writelane)

    float cnorm[CT], cb[CT], cberr[CT], cabs[CT];
    u64 valid_lanes[CT];  // ballot of the lanes whose column is a real hash bit (not padding)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int pc = ct * 32 + r31;
        cnorm[ct] = a.colnorm[pc];  // relative part of the bound: times |x| (prep kernel)
        cb[ct] = a.cb32[pc];
        cberr[ct] = a.cberr[pc];
        cabs[ct] = a.cabs[pc];
        // the accumulators start from -c_b without normalisation: c_b takes part in the float32 accumulation,
        // and the absolute part of the split's error is not scaled by a row's 1/|x|
        if constexpr (!NORMED) cberr[ct] += fabsf(cb[ct]) * ((3.f * D + 2.f) * 5.9604644775390625e-08f * 1.0001f) + cabs[ct];
        valid_lanes[ct] = __ballot(pc >= a.pad);
        // complete before the DMA ring starts (see sq_dense_scan.hpp)
        asm volatile("" : "+v"(cnorm[ct]), "+v"(cb[ct]), "+v"(cberr[ct]), "+v"(cabs[ct]));
    }
    if constexpr (BREG) {
#pragma unroll
        for (int ks = 0; ks < KU * 4; ++ks)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) asm volatile("" : "+v"(breg[ks][ct]));
    }

    u32 voff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = 4 * j + (lane >> 4);
        voff[j] = (u32)(r * (D * 4) + (((lane & 15) ^ (r & 15)) * 16));
    }

    // The wave's own stores (codes, undecided-bit entries) sit in the same in-order vmcnt queue as its DMA loads.
    // A wait that only counted the younger UNITS (vmcnt(8 * units)) also waited for the first loads of the next
    // unit whenever stores had been issued after it -- i.e. after every tile.  Every tile therefore issues a FIXED
    // number of store instructions (STORES_PER_TILE, from inline asm: a compiler-visible store could be split or
    // merged) and the waits allow them: vmcnt(8 * younger units + STORES_PER_TILE * younger tile epilogues).
    constexpr int STORES_PER_TILE = CT / 2 + CT;
    int epi_total = 0;          // tile epilogues so far
    int epi_at[NSTAGE];         // epi_total when the unit now in ring slot s was issued
    long long iss_tile = wave_id;
    int iss_kc = 0, iss_slot = 0;
    long long issued = 0;
    auto issue_next = [&]() {
        if (issued >= total_units) return;
#pragma unroll
        for (int sl2 = 0; sl2 < NSTAGE; ++sl2)
            if (iss_slot == sl2) epi_at[sl2] = epi_total;
        long long row0 = iss_tile * 32;
        // the last tile may reach past the matrix: pull it back (rows are re-done, the results of
        // rows >= n are never stored) so that no DMA leaves the allocation
        if (row0 + 32 > a.n) row0 = a.n - 32;
        const u32 dst = ring_base + (u32)iss_slot * ITQF_UNIT_BYTES;
        const unsigned char* base = reinterpret_cast<const unsigned char*>(a.x) + row0 * (D * 4) + iss_kc * 256;
#pragma unroll
        for (int j = 0; j < 8; ++j) glds16<true>(base, voff[j], dst + (u32)j * 1024);   // non-temporal: the rows are read once
        ++issued;
        if (++iss_kc == KU) {
            iss_kc = 0;
            iss_tile += nwaves;
        }
        if (++iss_slot == NSTAGE) iss_slot = 0;
    };
    for (int p = 0; p < NSTAGE; ++p) issue_next();

    auto read_b = [&](int ks, itq_f16x8 (&bf)[CT][2]) {  // fragments of k-step ks from LDS (BREG: the lo plane only)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            if constexpr (BREG) {
                bf[ct][1] = *reinterpret_cast<const itq_f16x8*>(smem + blo_off(ks, ct * 32 + r31));
            } else {
                const unsigned char* col = smem + b_off(ks, ct * 32 + r31);
                bf[ct][0] = *reinterpret_cast<const itq_f16x8*>(col);
                bf[ct][1] = *reinterpret_cast<const itq_f16x8*>(col + DP * 2);
            }
        }
    };

    u32 wcount = 0;
    long long consumed = 0;
    int rd_slot = 0;
    for (long long tile = wave_id; tile < a.n_tiles; tile += nwaves) {
        long long row0 = tile * 32;
        const long long shift = row0 + 32 > a.n ? row0 + 32 - a.n : 0;  // rows the DMA window moved back
        row0 -= shift;
        // without normalisation the accumulators start from -c_b, so that they END as z~ = x . R_b - c_b and the
        // epilogue has no arithmetic left (the float32 accumulation error on c_b is part of cberr_eff)
        itq_f32x16 acc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[ct][i] = NORMED ? 0.f : -cb[ct];
        // |x|^2 per row: normalize=2 needs every row's own (float32 sum of squares, as round 1); without
        // normalisation only the LARGEST |x| of the tile enters the bound, and the matrix cores give it for free:
        // G = X_hi X_hi^T (one more MFMA per k-step, A and B are the same registers) has the squared norms on its
        // diagonal and nothing larger anywhere (Cauchy-Schwarz), so max |G| over the accumulator IS max |x_hi|^2 --
        // 8 v_max3 + a wave reduction per tile instead of a multiply-add per element.
        float sumsq = 0.f;
        itq_f32x16 gram;
        if constexpr (!NORMED && SQ_ITQ_GRAM) {
#pragma unroll
            for (int i = 0; i < 16; ++i) gram[i] = 0.f;
        }
#pragma unroll
        for (int kc = 0; kc < KU; ++kc) {
            // unit `consumed` must have landed; younger units and the stores of younger epilogues may stay in flight
            {
                int epi_young = 0;
#pragma unroll
                for (int sl2 = 0; sl2 < NSTAGE; ++sl2)
                    if (rd_slot == sl2) epi_young = epi_total - epi_at[sl2];
                wait_ops_in_flight<NSTAGE, 8, STORES_PER_TILE>((int)(issued - consumed - 1), epi_young);
            }
            const unsigned char* sl = ring_ptr + rd_slot * ITQF_UNIT_BYTES;
            // The 32 rows x 64 floats of the unit reach the registers in two halves (k-steps 0-1, then 2-3): all four
            // k-steps at once need 32 registers next to the 128 of the R fragments and the 32 accumulators, and the
            // compiler then spilled -- a scratch reload inside the loop makes it wait vmcnt(0), which drains the DMA
            // ring once per tile (38 % of the wave cycles parked, profiles/r02_itq_pmc_before.json).  The slot is
            // handed back to the DMA after the second half has been read.
            itq_f32x4 xa[2][2];
            itq_f16x8 bcur[CT][2], bnxt[CT][2];
            read_b(kc * 4, bnxt);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        xa[s2][e] = *reinterpret_cast<const itq_f32x4*>(
                            sl + r31 * 256 + (((4 * (2 * half + s2) + 2 * h + e) ^ (r31 & 15)) * 16));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (half == 1) {  // the slot is free once its values are in registers
                    ++consumed;
                    if (++rd_slot == NSTAGE) rd_slot = 0;
                    issue_next();
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int s = 2 * half + s2;
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        if constexpr (!BREG) bcur[ct][0] = bnxt[ct][0];
                        bcur[ct][1] = bnxt[ct][1];
                    }
                    if (s < 3) read_b(kc * 4 + s + 1, bnxt);  // the next k-step's fragments under this one's arithmetic
                    itq_u32x4 uhw, ulw;
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        const float u0 = j < 4 ? xa[s2][0][j] : xa[s2][1][j - 4];
                        const float u1 = j < 4 ? xa[s2][0][j + 1] : xa[s2][1][j - 3];
                        if constexpr (NORMED || !SQ_ITQ_GRAM) {
                            sumsq = __fmaf_rn(u0, u0, sumsq);
                            sumsq = __fmaf_rn(u1, u1, sumsq);
                        }
                        u32 hw, lw;
                        split_f16_pair(u0, u1, hw, lw);
                        uhw[j >> 1] = hw;
                        ulw[j >> 1] = lw;
                    }
                    const itq_f16x8 uh = __builtin_bit_cast(itq_f16x8, uhw), ul = __builtin_bit_cast(itq_f16x8, ulw);
                    if constexpr (!NORMED && SQ_ITQ_GRAM) gram = __builtin_amdgcn_mfma_f32_32x32x16_f16(uh, uh, gram, 0, 0, 0);
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        const itq_f16x8 bh = BREG ? breg[BREG ? kc * 4 + s : 0][ct] : bcur[ct][0];
                        const itq_f16x8 bl = bcur[ct][1];
                        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ul, bh, acc[ct], 0, 0, 0);
                        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(uh, bl, acc[ct], 0, 0, 0);
                        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(uh, bh, acc[ct], 0, 0, 0);
                    }
                }
            }
        }
        // ---- tile complete: x . R (- c_b) for 32 rows x PC columns (lane = column, register i = row (i&3)+8(i>>2)+4h)
        // The epilogue is the expensive part of a tile in instructions (the kernel is issue bound, PMC: 43 % of the
        // wave cycles issuing, 36 % stalled on issue), so it does the least it can: per (register, column tile) ONE
        // compare whose ballot is the sign word of two rows, dropped into the lanes of those rows with two
        // v_writelane, and one compare + scalar OR for "some |z~| of this column is within eps"; the per-row words
        // of undecided bits are only built for a column tile in which that happened (or in a saturating tile).
        float rowscale = 1.f, U = 1.f, big;
        if constexpr (NORMED) {
            sumsq += __shfl_xor(sumsq, 32);  // both halves of row r31
            rowscale = sumsq > 0.f ? 1.0f / sqrtf(sumsq) : 0.f;  // zero row: z~ = -mean.R_b
            big = sumsq;
            // the absolute part of the split's error meets the LARGEST 1/|x| of the tile
            float rsmax = rowscale;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) rsmax = fmaxf(rsmax, __shfl_xor(rsmax, o));
            U = rsmax;
        } else if constexpr (!SQ_ITQ_GRAM) {
            sumsq += __shfl_xor(sumsq, 32);
            float g = sumsq;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) g = fmaxf(g, __shfl_xor(g, o));
            big = g;
            U = sqrtf(g) * 1.0001f;
        } else {
            float g = 0.f;
#pragma unroll
            for (int i = 0; i < 16; i += 2) g = __builtin_fmaxf(g, __builtin_fmaxf(fabsf(gram[i]), fabsf(gram[i + 1])));
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) g = fmaxf(g, __shfl_xor(g, o));
            big = g;
            // |x| <= (|x_hi| + 2^-24 sqrt(d)) / (1 - 2^-10), G in float32
            U = (sqrtf(g) + 1e-6f) * 1.002f;
        }
        // a plane saturates silently from |x_k| = 65504 on: such a tile (and one holding an inf) decides nothing here.
        // (A NaN row is not seen by the maxima above; its z~ are NaN and the undecided test below is NaN-safe.)
        const bool bad_rows = __ballot(!(big < 1e9f)) != 0ull;
        u32 half_word[CT], unc_mask[CT];  // this lane's row: sign bits and undecided columns of every column tile
        float eps[CT];
        u64 unc_any[CT];                  // lanes (columns) that saw |z~| <= eps or a NaN in some row of the tile
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            half_word[ct] = unc_mask[ct] = 0u;
            unc_any[ct] = 0ull;
            eps[ct] = NORMED ? cnorm[ct] + cberr[ct] + cabs[ct] * U : cnorm[ct] * U + cberr[ct];
        }
        auto zval = [&](int ct, int i, float rs) {
            if constexpr (NORMED) return __fmaf_rn(acc[ct][i], rs, -cb[ct]);
            return acc[ct][i];
        };
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row_a = (i & 3) + 8 * (i >> 2);  // row of lane half 0; half 1 holds row_a + 4
            float rs = 1.f;
            if constexpr (NORMED) rs = __shfl(rowscale, row_a + 4 * h);
#pragma unroll
            for (int ct = 0; ct < CT; ct += 2) {  // CT is 2 or 4: two column tiles per write block
                const float z0 = zval(ct, i, rs), z1 = zval(ct + 1, i, rs);
                const u64 pos0 = __ballot(z0 >= 0.f), pos1 = __ballot(z1 >= 0.f);
                unc_any[ct] |= __ballot(!(fabsf(z0) > eps[ct]));          // one compare + one scalar OR; true for a NaN
                unc_any[ct + 1] |= __ballot(!(fabsf(z1) > eps[ct + 1]));
                write_lanes_2x2(half_word[ct], half_word[ct + 1], (u32)pos0, (u32)(pos0 >> 32), (u32)pos1, (u32)(pos1 >> 32),
                                row_a, row_a + 4);
            }
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            half_word[ct] = __brev(half_word[ct]);  // column 0 -> most significant
            if (bad_rows || (unc_any[ct] & valid_lanes[ct]) != 0ull) {  // wave-uniform
                const bool cval = (valid_lanes[ct] >> lane) & 1ull;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row_a = (i & 3) + 8 * (i >> 2);
                    float rs = 1.f;
                    if constexpr (NORMED) rs = __shfl(rowscale, row_a + 4 * h);
                    const float z = zval(ct, i, rs);
                    const u64 unc = __ballot(cval && (bad_rows || !(fabsf(z) > eps[ct])));
                    write_lanes_1x2(unc_mask[ct], (u32)unc, (u32)(unc >> 32), row_a, row_a + 4);  // bit c = column c of the tile
                }
            }
        }
        const long long row = row0 + r31;
        const bool mine = lane < 32 && r31 >= (int)shift && row < a.n;  // rows below `shift` belong to the previous tile
        // (at least one lane is `mine` in every tile: exactly CT / 2 store instructions)
        if (mine) {
#pragma unroll
            for (int w = 0; w < CT / 2; ++w) {
                u64 v = ((u64)half_word[2 * w] << 32) | (u64)half_word[2 * w + 1];
                if (w == 0 && a.pad > 0) v &= (~0ull) >> a.pad;
                store_u64_counted(a.out + row * a.words + w, v);
            }
        }
        // undecided bits: one entry per (row, column tile) that has any, for the float64 evaluation of
        // exactly those bits (itq_fix_bits_kernel).  Lane 63 (never `mine`) writes a dummy word so that the
        // instruction is issued in every tile: exactly CT store instructions.
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const bool need = mine && unc_mask[ct] != 0;
            const u64 nb = __ballot(need);
            const u32 p = wcount + __builtin_amdgcn_mbcnt_hi((u32)(nb >> 32), __builtin_amdgcn_mbcnt_lo((u32)nb, 0u));
            if (need || lane == 63)
                store_u64_counted(need ? myseg + p : a.seg_dummy + wave_id,
                                  ((u64)((u32)row | ((u32)ct << 30)) << 32) | (u64)unc_mask[ct]);
            wcount += (u32)__popcll(nb);
        }
        ++epi_total;
    }
    if (lane == 0) a.seg_cnt[wave_id] = wcount;
}

}  // namespace sq
