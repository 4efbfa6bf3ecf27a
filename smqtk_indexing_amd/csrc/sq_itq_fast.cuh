// ITQ hash codes at the HBM rate: a certified bf16x3 filter in front of the
// float64 kernel (sq_itq.hip).
//
// ItqFunctor.get_hash (smqtk_indexing/impls/lsh_functor/itq.py:389-408) only
// keeps the SIGN of z = (v - mean) . R, v = x or x/|x|, but evaluates z in
// float64; on the f64 matrix cores that is 2*n*d*bits flop (10 M x 128 -> 64
// bits: 3.9 ms) under a 0.85 ms HBM floor.  The sign of z is known as soon as
// |z~| exceeds the error bound of a cheaper evaluation z~, so this kernel
// streams the float32 rows once (the dense scan's LDS-DMA ring), splits x and R
// into bfloat16 pairs, evaluates x_hi R_hi + x_hi R_lo + x_lo R_hi on
// v_mfma_f32_32x32x16_bf16 and finishes with z~ = (x . R_b) s - c_b, where
// s = 1/|x| (normalize=2, |x| from the same pass) or 1 and c_b = mean . R_b is a
// float64 product rounded once.  Bits with |z~| > eps(row, column) are final;
// the others (one in ~2500) are listed as (row, column tile, mask) entries and
// evaluated in float64 one bit at a time (itq_fix_bits_kernel, sq_itq.hip), so
// every code is exactly what the float64 evaluation gives.
//
// Error bound.  x = x_hi + x_lo + dx, |dx_k| <= 2^-16 |x_k| and
// the same for R (after its float32 rounding); the dropped x_lo R_lo is
// <= 2^-16 |x_k||R_kb|; the float32 accumulation of 3d terms adds
// <= 3d 2^-24 sum |x_k||R_kb|; Cauchy-Schwarz over k gives
//   |x . R_b - (x . R_b)~| <= (3 * 2^-16 + 3d 2^-24) |x| |R_b|.
// normalize=2: the reference's v_k = fl32(x_k / fl32|x|) is within 2^-21 |x_k|/|x|
// of x_k/|x|, and s is within 2^-18 (float32 sum of squares); scaled to the unit
// row that is (3 * 2^-16 + 3d 2^-24 + 2^-20 + 2^-18) |R_b|.  c_b: its float32
// rounding and float64 summation error, from the prep kernel (cberr).
#pragma once
#include "sq_dma.cuh"

namespace sq {

typedef float itq_f32x4 __attribute__((ext_vector_type(4)));
typedef float itq_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 itq_bf16x8 __attribute__((ext_vector_type(8)));

static constexpr int ITQF_UNIT_BYTES = 32 * 256;                // 32 rows x 64 floats
static constexpr int ITQF_WAVES_LDSB = 4;  // waves when the R fragments come from LDS
static constexpr int ITQF_WAVES_BREG = 8;  // waves when they live in registers
static constexpr int ITQF_MAX_CT = 4;                           // up to 128 padded hash bits

struct ItqFastArgs {
    const float* x;        // [n][d] float32 rows, d % 64 == 0, 16-byte aligned
    long long n;
    int d;
    const uint4* rimage;   // image of R: [pc columns][2 planes][dp*2 bytes], dp = d rounded up to 128; chunks swizzled by column & 15
    const float* colnorm;  // [pc] |R_b|_2 rounded up; pad columns 0
    const float* cb32;     // [pc] mean . R_b in float32 (normalize=2 form)
    const float* cberr;    // [pc] bound of what cb32 and the float32 division leave out
    float eps_rel;         // 3 * 2^-16 + 3d * 2^-24 + 2^-20 (+ 2^-18 for normalize=2): times |R_b| |x|
    u64* out;              // [n][words]
    int words, pad, bits;  // pad = words*64 - bits leading zero columns
    u64* seg;              // [waves of the launch][seg_cap] undecided (row | column tile << 30) << 32 | 32-column mask
    u32* seg_cnt;          // [waves of the launch]
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
                                                                    double* __restrict__ rt64) {
    const int pc = blockIdx.x;
    const int b = pc - pad;
    __shared__ double red[256];
    __shared__ double red2[256];
    __shared__ double red3[256];
    double acc = 0.0, cacc = 0.0, cabs = 0.0, macc = 0.0;
    for (int k = threadIdx.x; k < d; k += 256) {
        const double r = b >= 0 ? rot[(long long)k * bits + b] : 0.0;
        rt64[(long long)pc * d + k] = r;  // column-major float64 copy: the per-bit float64 evaluation reads whole columns
        acc += r * r;
        cacc += mean[k] * r;
        cabs += fabs(mean[k] * r);
        macc += mean[k] * mean[k];
        const float rf = (float)r;
        const __bf16 hi = (__bf16)rf;
        const __bf16 lo = (__bf16)(rf - (float)hi);
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
    if (threadIdx.x == 0) colnorm[pc] = (float)(sqrt(red[0]) * (1.0 + 1e-6));
    __syncthreads();
    const double rnorm = sqrt(red[0]);
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
    // LDS: [R image: PC * 2 planes * DP*2 bytes (not BREG)][rings]
    constexpr u32 r_bytes = BREG ? 0u : (u32)PC * 2u * DP * 2u;
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
    itq_bf16x8 breg[BREG ? KU * 4 : 1][CT][2];
    if constexpr (BREG) {
        const unsigned char* img = reinterpret_cast<const unsigned char*>(a.rimage);
#pragma unroll
        for (int ks = 0; ks < KU * 4; ++ks)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const u32 o = b_off(ks, ct * 32 + r31);
                breg[ks][ct][0] = *reinterpret_cast<const itq_bf16x8*>(img + o);
                breg[ks][ct][1] = *reinterpret_cast<const itq_bf16x8*>(img + o + DP * 2);
            }
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

    // lane L < 32 finalises row L of a tile: accumulator register my_i of lane half my_h holds it
    const int my_i = (r31 & 3) | ((r31 >> 3) << 2), my_h = (r31 >> 2) & 1;

    float cnorm[CT], cb[CT], cberr[CT];
    bool cvalid[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int pc = ct * 32 + r31;
        cnorm[ct] = a.colnorm[pc] * a.eps_rel;
        cb[ct] = a.cb32[pc];
        cberr[ct] = a.cberr[pc];
        cvalid[ct] = pc >= a.pad;
        // complete before the DMA ring starts (see sq_dense_scan.cuh)
        asm volatile("" : "+v"(cnorm[ct]), "+v"(cb[ct]), "+v"(cberr[ct]));
    }
    if constexpr (BREG) {
#pragma unroll
        for (int ks = 0; ks < KU * 4; ++ks)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) asm volatile("" : "+v"(breg[ks][ct][0]), "+v"(breg[ks][ct][1]));
    }

    u32 voff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = 4 * j + (lane >> 4);
        voff[j] = (u32)(r * (D * 4) + (((lane & 15) ^ (r & 15)) * 16));
    }

    long long iss_tile = wave_id;
    int iss_kc = 0, iss_slot = 0;
    long long issued = 0;
    auto issue_next = [&]() {
        if (issued >= total_units) return;
        long long row0 = iss_tile * 32;
        // the last tile may reach past the matrix: pull it back (rows are re-done, the results of
        // rows >= n are never stored) so that no DMA leaves the allocation
        if (row0 + 32 > a.n) row0 = a.n - 32;
        const u32 dst = ring_base + (u32)iss_slot * ITQF_UNIT_BYTES;
        const unsigned char* base = reinterpret_cast<const unsigned char*>(a.x) + row0 * (D * 4) + iss_kc * 256;
#pragma unroll
        for (int j = 0; j < 8; ++j) glds16(base, voff[j], dst + (u32)j * 1024);
        ++issued;
        if (++iss_kc == KU) {
            iss_kc = 0;
            iss_tile += nwaves;
        }
        if (++iss_slot == NSTAGE) iss_slot = 0;
    };
    for (int p = 0; p < NSTAGE; ++p) issue_next();

    auto read_b = [&](int ks, itq_bf16x8 (&bf)[CT][2]) {  // not BREG
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const unsigned char* col = smem + b_off(ks, ct * 32 + r31);
            bf[ct][0] = *reinterpret_cast<const itq_bf16x8*>(col);
            bf[ct][1] = *reinterpret_cast<const itq_bf16x8*>(col + DP * 2);
        }
    };

    u32 wcount = 0;
    long long consumed = 0;
    int rd_slot = 0;
    for (long long tile = wave_id; tile < a.n_tiles; tile += nwaves) {
        long long row0 = tile * 32;
        const long long shift = row0 + 32 > a.n ? row0 + 32 - a.n : 0;  // rows the DMA window moved back
        row0 -= shift;
        itq_f32x16 acc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[ct][i] = 0.f;
        float sumsq = 0.f;
#pragma unroll
        for (int kc = 0; kc < KU; ++kc) {
            // unit `consumed` must have landed; younger units may stay in flight
            wait_units_in_flight<NSTAGE, 8>((int)(issued - consumed - 1));
            const unsigned char* sl = ring_ptr + rd_slot * ITQF_UNIT_BYTES;
            itq_f32x4 xa[4][2];
            itq_bf16x8 bcur[CT][2], bnxt[CT][2];
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    xa[s][e] = *reinterpret_cast<const itq_f32x4*>(sl + r31 * 256 + (((4 * s + 2 * h + e) ^ (r31 & 15)) * 16));
            if constexpr (!BREG) read_b(kc * 4, bnxt);
            // the slot is free once its values are in registers
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            ++consumed;
            if (++rd_slot == NSTAGE) rd_slot = 0;
            issue_next();
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if constexpr (!BREG) {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        bcur[ct][0] = bnxt[ct][0];
                        bcur[ct][1] = bnxt[ct][1];
                    }
                    if (s < 3) read_b(kc * 4 + s + 1, bnxt);  // the next k-step's fragments under this one's arithmetic
                }
                itq_bf16x8 uh, ul;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float u = j < 4 ? xa[s][0][j] : xa[s][1][j - 4];
                    sumsq = __fmaf_rn(u, u, sumsq);
                    const __bf16 hi = (__bf16)u;
                    uh[j] = hi;
                    ul[j] = (__bf16)(u - (float)hi);
                }
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const itq_bf16x8 bh = BREG ? breg[BREG ? kc * 4 + s : 0][ct][0] : bcur[ct][0];
                    const itq_bf16x8 bl = BREG ? breg[BREG ? kc * 4 + s : 0][ct][1] : bcur[ct][1];
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ul, bh, acc[ct], 0, 0, 0);
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uh, bl, acc[ct], 0, 0, 0);
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uh, bh, acc[ct], 0, 0, 0);
                }
            }
        }
        // ---- tile complete: x . R for 32 rows x PC columns (lane = column, register i = row (i&3)+8(i>>2)+4h)
        sumsq += __shfl_xor(sumsq, 32);  // both halves of row r31
        float rowscale = 1.f, U = 1.f;
        if constexpr (NORMED) {
            rowscale = sumsq > 0.f ? 1.0f / sqrtf(sumsq) : 0.f;  // zero row: z~ = -mean.R_b
        } else {
            float umax = sumsq;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) umax = fmaxf(umax, __shfl_xor(umax, o));
            U = sqrtf(umax) * 1.0001f;  // the largest |x| of the tile
        }
        u32 half_word[CT], unc_mask[CT];  // this lane's row: sign bits and undecided columns of every column tile
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float rs = 1.f;
            if constexpr (NORMED) rs = __shfl(rowscale, (i & 3) + 8 * (i >> 2) + 4 * h);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                if (i == 0) half_word[ct] = unc_mask[ct] = 0;
                const float eps = cnorm[ct] * U + cberr[ct];  // cnorm carries eps_rel
                const float z = acc[ct][i] * rs - cb[ct];
                const u64 pos = __ballot(z >= 0.f);
                const u64 unc = __ballot(cvalid[ct] && !(fabsf(z) > eps));
                if (i == my_i) {  // this lane's row sits in register i of lane half my_h
                    half_word[ct] = __brev(my_h ? (u32)(pos >> 32) : (u32)pos);  // column 0 -> most significant
                    unc_mask[ct] = my_h ? (u32)(unc >> 32) : (u32)unc;                  // bit c = column c of the tile
                }
            }
        }
        const long long row = row0 + r31;
        const bool mine = lane < 32 && r31 >= (int)shift && row < a.n;  // rows below `shift` belong to the previous tile
        if (mine) {
#pragma unroll
            for (int w = 0; w < CT / 2; ++w) {
                u64 v = ((u64)half_word[2 * w] << 32) | (u64)half_word[2 * w + 1];
                if (w == 0 && a.pad > 0) v &= (~0ull) >> a.pad;
                a.out[row * a.words + w] = v;
            }
        }
        // undecided bits: one entry per (row, column tile) that has any, for the float64 evaluation of
        // exactly those bits (itq_fix_bits_kernel)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const bool need = mine && unc_mask[ct] != 0;
            const u64 nb = __ballot(need);
            if (need) {
                const u32 p = wcount + __builtin_amdgcn_mbcnt_hi((u32)(nb >> 32), __builtin_amdgcn_mbcnt_lo((u32)nb, 0u));
                myseg[p] = ((u64)((u32)row | ((u32)ct << 30)) << 32) | (u64)unc_mask[ct];
            }
            wcount += (u32)__popcll(nb);
        }
    }
    if (lane == 0) a.seg_cnt[wave_id] = wcount;
}

}  // namespace sq
