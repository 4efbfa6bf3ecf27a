// The certified float16 filter of ItqFunctor.get_hash for the shapes sq_itq_fast.hpp does not cover: descriptors up
// to 512 wide, codes up to 256 bits, float32 OR float64 rows (SMQTK's default descriptor dtype) --
// smqtk_indexing/impls/lsh_functor/itq.py:389-408 handles any of them, and BASELINE config 4's descriptors are
// 512-d.  Those shapes ran the float64 MFMA kernel: 2 M x 512 -> 256 bits in 16.3 ms, 0.03 of the HBM rate.
//
// What is different from the narrow kernel.  The rotation R (512 x 256 as two float16 planes: 512 KB) fits neither the
// LDS nor a wave's registers, so the roles swap: a wave keeps its 32-ROW tile resident -- both float16 planes of all d
// elements as MFMA A fragments, d/2 registers (256 at d = 512: one wave per SIMD, the unified 512-register file) --
// and R streams past it, chunk by chunk (32 columns x 256 k x 2 planes = 32 KB) through a double buffer in LDS that
// the four waves of the workgroup share: 128 rows use every byte of R fetched from L2.  Each wave DMAs a quarter of
// the next chunk while the current one feeds 48 MFMAs per wave; one s_barrier per chunk.  Rows arrive through a small
// per-wave LDS-DMA ring (units of 32 rows x 256 bytes), are split in registers exactly as in the narrow kernel
// (float32: v_cvt_pkrtz + v_fma_mix; float64: a float64 subtraction in between), and the tile's squared norms come
// from the same pass.
// The MFMAs are issued from inline asm with the resident row fragments as the B operand in AGPRs (left to hipcc the
// 256 registers of fragments live in AGPRs anyway and every MFMA is preceded by four v_accvgpr_read copies: as many
// vector-issue cycles as the MFMA itself takes) and R's fragment, fresh from LDS, as A: the product comes out
// transposed -- lane = row, register = column -- so a lane owns the bits of its own row and the epilogue needs no
// ballots: per register one multiply-add, two compares, two bit inserts.
// Accumulators per column tile: one per 256-k block for x_hi R_hi (the float32 accumulation bound then scales with
// 256, not with d) and one for the corrections x_lo R_hi + x_hi R_lo; z~ = (sum) s - c_b as in the narrow kernel, the
// same per-column error coefficients from itq_fast_prep_kernel, the same undecided-bit entries -- (row, column tile,
// mask) with the tile in 3 bits -- for itq_fix_bits_wide_kernel, which evaluates exactly those bits in float64 in
// the reference's arithmetic.  Every code is therefore what the float64 kernel gives.
#pragma once
#include "sq_itq_fast.hpp"

namespace sq {

static constexpr int ITQW_WAVES = 4;
static constexpr int ITQW_NSTAGE = 2;
static constexpr int ITQW_CHUNK_BYTES = 32 * 1024;   // 32 columns x (2 planes x 512 B)
static constexpr int ITQW_MAX_KS = 32;               // k-steps of 16: d <= 512
static constexpr int ITQW_MAX_CT = 8;                // column tiles of 32: <= 256 bits

struct ItqWideArgs {
    const void* x;         // [n][d] rows of T, 16-byte aligned
    long long n;
    int d;                 // d % 64 == 0, d <= 512
    const uint4* rimage;   // itq_fast_prep_kernel's image: [pc][2 planes][dp*2 bytes]
    const float* colnorm;  // [pc] per-column error coefficients (sq_itq_fast.hpp)
    const float* cabs;
    const float* cb32;
    const float* cberr;
    u64* out;              // [n][words]
    int words, pad, bits, ct;
    u64* seg;              // [waves of the launch][seg_cap] undecided entries: (row | column tile << 29) << 32 | 32-column mask
    u32* seg_cnt;
    long long seg_cap;
    long long n_tiles;
    int nrb;
    u64* stamps;           // measurement only: [workgroups][64] wall-clock stamps (100 MHz) of wave 0's phases, or nullptr
    int debug;             // measurement only (wrong codes): 1 = no R stream (no chunk DMA, no barriers), 2 = rows loaded for the first round only, 4 = no epilogue
};

// D = A B + D with A in VGPRs and B in AGPRs (the matrix core reads either file; hipcc's builtin would copy B back)
__device__ __forceinline__ void mfma_f16_agpr_b(itq_f32x16& acc, const itq_f16x8& a_v, const itq_f16x8& b_a) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a_v), "a"(b_a));
}

// The resident fragments must LIVE in AGPRs (the MFMA's B operand): a value hipcc computed in VGPRs and merely handed
// to an "a" constraint is copied there before every use (four v_accvgpr_write per MFMA, with no wait states in
// front of the MFMA that reads them).  An asm result with an "=a" output is at home in AGPRs -- and the one
// instruction that writes a 128-bit AGPR tuple is a load: the fragment takes a bounce through a 2 KB scratch of the
// wave in LDS (in-order per wave: the read sees the write).
__device__ __forceinline__ void frag_pair_to_agpr(unsigned char* scr, int lane, const itq_u32x4& hw, const itq_u32x4& lw,
                                                  itq_f16x8& xh, itq_f16x8& xl) {
    *reinterpret_cast<itq_u32x4*>(scr + lane * 16) = hw;
    *reinterpret_cast<itq_u32x4*>(scr + 1024 + lane * 16) = lw;
    const u32 addr = (u32)(uintptr_t)scr + (u32)lane * 16u;
    asm volatile(
        "ds_read_b128 %0, %2\n\t"
        "ds_read_b128 %1, %2 offset:1024\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=a"(xh), "=a"(xl)
        : "v"(addr)
        : "memory");
}

// ... a whole row unit at a time: the unit's own ring slot (8 KiB, its raw rows are in registers by then) takes the
// converted fragments, one asm block reads them back into AGPRs and waits ONCE (a bounce per k-step exposed the LDS
// round trip 32 times per tile: ~4 us of a 14 us row phase).
template <int KSU>
__device__ __forceinline__ void unit_frags_to_agpr(unsigned char* slot, int lane, const itq_u32x4 (&hw)[KSU], const itq_u32x4 (&lw)[KSU],
                                                   itq_f16x8* xh, itq_f16x8* xl) {
#pragma unroll
    for (int s = 0; s < KSU; ++s) {
        *reinterpret_cast<itq_u32x4*>(slot + (2 * s) * 1024 + lane * 16) = hw[s];
        *reinterpret_cast<itq_u32x4*>(slot + (2 * s + 1) * 1024 + lane * 16) = lw[s];
    }
    const u32 addr = (u32)(uintptr_t)slot + (u32)lane * 16u;
    if constexpr (KSU == 4) {
        asm volatile(
            "ds_read_b128 %0, %8\n\t"
            "ds_read_b128 %1, %8 offset:1024\n\t"
            "ds_read_b128 %2, %8 offset:2048\n\t"
            "ds_read_b128 %3, %8 offset:3072\n\t"
            "ds_read_b128 %4, %8 offset:4096\n\t"
            "ds_read_b128 %5, %8 offset:5120\n\t"
            "ds_read_b128 %6, %8 offset:6144\n\t"
            "ds_read_b128 %7, %8 offset:7168\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=a"(xh[0]), "=a"(xl[0]), "=a"(xh[1]), "=a"(xl[1]), "=a"(xh[2]), "=a"(xl[2]), "=a"(xh[3]), "=a"(xl[3])
            : "v"(addr)
            : "memory");
    } else {
        static_assert(KSU == 2, "k-steps per row unit");
        asm volatile(
            "ds_read_b128 %0, %4\n\t"
            "ds_read_b128 %1, %4 offset:1024\n\t"
            "ds_read_b128 %2, %4 offset:2048\n\t"
            "ds_read_b128 %3, %4 offset:3072\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=a"(xh[0]), "=a"(xl[0]), "=a"(xh[1]), "=a"(xl[1])
            : "v"(addr)
            : "memory");
    }
}

// NKB: 256-k blocks of a row (1: d <= 256, 2: d <= 512).  A block always runs its 16 k-steps: fragments beyond d are
// zero (and the R image is zero-filled there), so there is no per-k-step branch in the MFMA stream.
template <class T, bool NORMED, int NKB>
__global__ __launch_bounds__(ITQW_WAVES * 64, 1) void itq_wide_kernel(ItqWideArgs a) {
    constexpr int EPU = 256 / (int)sizeof(T);   // elements of a row in one 256-byte unit: 64 (float32) / 32 (float64)
    constexpr int KSU = EPU / 16;               // k-steps per unit: 4 / 2
    constexpr int KS = NKB * 16;                // k-steps held per tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS: [R chunk buffers 2 x 32 KB][column constants 4 x 256 floats][rings 4 waves x 2 x 8 KB]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r31 = lane & 31, h = lane >> 5;
    const int D = a.d, DP = (D + 127) / 128 * 128, NU = D / EPU;
    const u32 lds_base = (u32)(uintptr_t)smem;
    constexpr u32 rbuf_bytes = 2u * ITQW_CHUNK_BYTES, const_bytes = 4u * 256u * 4u;   // (three arrays used, four reserved)
    float* lconst = reinterpret_cast<float*>(smem + rbuf_bytes);   // [3][256]: c_b, epsA, epsB: eps(column) = epsA U + epsB
    const u32 ring_base = lds_base + rbuf_bytes + const_bytes + (u32)wave * (ITQW_NSTAGE * ITQF_UNIT_BYTES);
    const unsigned char* ring_ptr = smem + rbuf_bytes + const_bytes + wave * (ITQW_NSTAGE * ITQF_UNIT_BYTES);
    unsigned char* scr = smem + rbuf_bytes + const_bytes + ITQW_WAVES * (ITQW_NSTAGE * ITQF_UNIT_BYTES) + wave * 2048;   // AGPR bounce
    const int PC = a.ct * 32;
    for (int i = threadIdx.x; i < 256; i += ITQW_WAVES * 64) {
        const float cn = i < PC ? a.colnorm[i] : 0.f, cb = i < PC ? a.cb32[i] : 0.f, ce = i < PC ? a.cberr[i] : 0.f;
        const float ca = i < PC ? a.cabs[i] : 0.f;
        lconst[i] = cb;
        if constexpr (NORMED) {
            // z~ = (x . R_b) / |x| - c_b: eps = colnorm + cberr + cabs U, U = the largest 1/|x| of the tile
            lconst[256 + i] = ca;
            lconst[512 + i] = cn + ce;
        } else {
            // z~ = x . R_b - c_b: eps = colnorm U + cberr', U = the largest |x| of the tile; c_b rides through the correction
            // accumulator's 2 d products and the final additions, and the split's absolute error is not scaled
            lconst[256 + i] = cn;
            lconst[512 + i] = ce + fabsf(cb) * ((2.f * D + 8.f) * 5.9604644775390625e-08f * 1.0001f) + ca;
        }
    }
    __syncthreads();

    const long long wave_id = (long long)blockIdx.x * ITQW_WAVES + wave;
    const long long nwaves = (long long)a.nrb * ITQW_WAVES;
    const long long rounds = (a.n_tiles + nwaves - 1) / nwaves;   // every wave of the launch runs the same rounds (barriers)
    u64* myseg = a.seg + wave_id * a.seg_cap;
    u32 wcount = 0;

    // per-lane byte offsets of the 8 DMA pieces of a row unit (row 4j + lane/16, swizzled 16-byte chunk)
    u32 voff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = 4 * j + (lane >> 4);
        voff[j] = (u32)(r * (D * (int)sizeof(T)) + (((lane & 15) ^ (r & 15)) * 16));
    }
    // R chunk DMA: this wave's 8 columns of the chunk; lane l: plane l / 32, 16-byte chunk l % 32 of the 512-byte block
    const unsigned char* img = reinterpret_cast<const unsigned char*>(a.rimage);
    const u32 rlane_off = (u32)((lane >> 5) * (DP * 2) + (lane & 31) * 16);
    const int n_chunks = a.ct * NKB;   // per round: chunk c = column tile c / NKB, k-block c % NKB
    auto issue_chunk = [&](int c, int buf) __attribute__((always_inline)) {
        const int ct = c / NKB, kb = c - ct * NKB;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int col = wave * 8 + j;
            const unsigned char* src = img + (size_t)(ct * 32 + col) * (2 * DP * 2) + kb * 512;
            glds16<false>(src, rlane_off, lds_base + (u32)buf * ITQW_CHUNK_BYTES + (u32)col * 1024);
        }
    };

    long long x_issued = 0, x_consumed = 0;   // row units DMA'd / taken into registers by this wave, all rounds
    for (long long round = 0; round < rounds; ++round) {
        long long tile = wave_id + round * nwaves;
        const bool active = tile < a.n_tiles;
        if (!active) tile = a.n_tiles - 1;          // keeps the barriers and the R stream; nothing is stored
        long long row0 = tile * 32;
        const long long shift = row0 + 32 > a.n ? row0 + 32 - a.n : 0;   // the last tile: the window moves back
        row0 -= shift;
        if (a.stamps && wave == 0 && lane == 0 && round < 20) a.stamps[(size_t)blockIdx.x * 64 + 3 * round] = wall_clock64();
        // ---- the tile's rows: DMA ring -> registers, split into float16 planes (A fragments of every k-step).
        // Four slots of 8 KiB per wave: the wave's own two, and -- the R buffers are idle while the rows load -- a 16 KiB
        // quarter of the R area (between the barrier that ends a round's MFMA phase and the one before chunk 0).  Units
        // 0 and 1 of the NEXT round's tile go into the own slots as soon as this tile's last units have left them: they
        // land under the MFMA phase.  (Two slots and no prefetch: the row phase ran at 64 KiB in flight per CU, 0.85 ms
        // of a 2.1 ms kernel during which the matrix cores idle.)
        itq_f16x8 xh[KS], xl[KS];
        float sumsq = 0.f;
        if (!(a.debug & 2) || round == 0) {
            const unsigned char* xbase = reinterpret_cast<const unsigned char*>(a.x) + row0 * (long long)D * (long long)sizeof(T);
            // the next round's window (same clamping as above)
            long long ntile = wave_id + (round + 1) * nwaves;
            if (ntile >= a.n_tiles) ntile = a.n_tiles - 1;
            long long nrow0 = ntile * 32;
            if (nrow0 + 32 > a.n) nrow0 = a.n - 32;
            const unsigned char* nbase = reinterpret_cast<const unsigned char*>(a.x) + nrow0 * (long long)D * (long long)sizeof(T);
            const bool has_next = round + 1 < rounds;
            auto slot_lds = [&](int sl4) __attribute__((always_inline)) -> u32 {
                return sl4 < 2 ? ring_base + (u32)sl4 * ITQF_UNIT_BYTES : lds_base + (u32)wave * 16384u + (u32)(sl4 - 2) * ITQF_UNIT_BYTES;
            };
            auto slot_ptr = [&](int sl4) __attribute__((always_inline)) -> const unsigned char* {
                return sl4 < 2 ? ring_ptr + sl4 * ITQF_UNIT_BYTES : smem + wave * 16384 + (sl4 - 2) * ITQF_UNIT_BYTES;
            };
            auto issue_unit = [&](const unsigned char* base, int unit, int sl4) __attribute__((always_inline)) {
                const u32 dst = slot_lds(sl4);
#pragma unroll
                for (int j = 0; j < 8; ++j) glds16<true>(base + unit * 256, voff[j], dst + (u32)j * 1024);
                ++x_issued;
            };
            if (round == 0) {   // (later rounds: units 0 and 1 were prefetched by the round before)
                issue_unit(xbase, 0, 0);
                if (NU > 1) issue_unit(xbase, 1, 1);
            }
            if (NU > 2) issue_unit(xbase, 2, 2);
            if (NU > 3) issue_unit(xbase, 3, 3);
#pragma unroll
            for (int u = 0; u < KS / KSU; ++u) {
                if (u >= NU) {   // beyond d: zero fragments (wave-uniform)
                    const itq_u32x4 zero = itq_u32x4{0u, 0u, 0u, 0u};
#pragma unroll
                    for (int s2 = 0; s2 < KSU; ++s2) frag_pair_to_agpr(scr, lane, zero, zero, xh[u * KSU + s2], xl[u * KSU + s2]);
                } else {
                    wait_units_in_flight<4, 8>((int)(x_issued - x_consumed - 1));
                    ++x_consumed;
                    const unsigned char* sl = slot_ptr(u & 3) + r31 * 256;
                    // the slot is free once the unit is in registers: this tile's unit u + 4, or -- own slots only -- the
                    // next tile's unit u & 3
                    auto refill = [&]() __attribute__((always_inline)) {
                        if (u + 4 < NU)
                            issue_unit(xbase, u + 4, u & 3);
                        else if ((u & 3) < 2 && has_next && (u & 3) < NU)
                            issue_unit(nbase, u & 3, u & 3);
                    };
                    if constexpr (sizeof(T) == 4) {
                        itq_f32x4 xa[KSU][2];
#pragma unroll
                        for (int s = 0; s < KSU; ++s)
#pragma unroll
                            for (int e = 0; e < 2; ++e)
                                xa[s][e] = *reinterpret_cast<const itq_f32x4*>(sl + (((4 * s + 2 * h + e) ^ (r31 & 15)) * 16));
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        itq_u32x4 hwv[KSU], lwv[KSU];
#pragma unroll
                        for (int s = 0; s < KSU; ++s) {
#pragma unroll
                            for (int j = 0; j < 8; j += 2) {
                                const float u0 = j < 4 ? xa[s][0][j] : xa[s][1][j - 4];
                                const float u1 = j < 4 ? xa[s][0][j + 1] : xa[s][1][j - 3];
                                sumsq = __fmaf_rn(u0, u0, sumsq);
                                sumsq = __fmaf_rn(u1, u1, sumsq);
                                u32 hh, ll;
                                split_f16_pair(u0, u1, hh, ll);
                                hwv[s][j >> 1] = hh;
                                lwv[s][j >> 1] = ll;
                            }
                        }
                        unit_frags_to_agpr<KSU>(const_cast<unsigned char*>(slot_ptr(u & 3)), lane, hwv, lwv, &xh[u * KSU], &xl[u * KSU]);
                        refill();   // (the slot served as the bounce buffer: free only now)
                    } else {
                        typedef double f64x2 __attribute__((ext_vector_type(2)));
                        f64x2 xa[KSU][4];
#pragma unroll
                        for (int s = 0; s < KSU; ++s)
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                xa[s][e] = *reinterpret_cast<const f64x2*>(sl + (((8 * s + 4 * h + e) ^ (r31 & 15)) * 16));
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        itq_u32x4 hwv[KSU], lwv[KSU];
#pragma unroll
                        for (int s = 0; s < KSU; ++s) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const double d0 = xa[s][e][0], d1 = xa[s][e][1];
                                const float f0 = (float)d0, f1 = (float)d1;
                                sumsq = __fmaf_rn(f0, f0, sumsq);
                                sumsq = __fmaf_rn(f1, f1, sumsq);
                                const auto hv = __builtin_amdgcn_cvt_pkrtz(f0, f1);   // two float16, round toward zero
                                // the residual is formed in float64 (x - hi is exact there), rounded to float32, truncated to float16
                                const float l0 = (float)(d0 - (double)(float)hv[0]), l1 = (float)(d1 - (double)(float)hv[1]);
                                hwv[s][e] = __builtin_bit_cast(u32, hv);
                                lwv[s][e] = __builtin_bit_cast(u32, __builtin_amdgcn_cvt_pkrtz(l0, l1));
                            }
                        }
                        unit_frags_to_agpr<KSU>(const_cast<unsigned char*>(slot_ptr(u & 3)), lane, hwv, lwv, &xh[u * KSU], &xl[u * KSU]);
                        refill();
                    }
                }
            }
        }
        // row norms: lane L < 32 (and its twin L + 32) end with |x|^2 of row r31
        sumsq += __shfl_xor(sumsq, 32);
        float rowscale = 1.f, U, big;
        {
            float g = sumsq;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) g = fmaxf(g, __shfl_xor(g, o));
            big = g;
            if constexpr (NORMED) {
                rowscale = sumsq > 0.f ? 1.0f / sqrtf(sumsq) : 0.f;   // zero row: z~ = -mean . R_b
                float rsmax = rowscale;
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) rsmax = fmaxf(rsmax, __shfl_xor(rsmax, o));
                U = rsmax;   // the absolute part of the split's error meets the largest 1/|x| of the tile
            } else {
                U = sqrtf(g) * 1.0001f;
            }
        }
        const bool bad_rows = __ballot(!(big < 1e9f)) != 0ull;   // a float16 plane saturates from |x_k| = 65504 on

        if (a.stamps && wave == 0 && lane == 0 && round < 20) a.stamps[(size_t)blockIdx.x * 64 + 3 * round + 1] = wall_clock64();
        // ---- R streams past the resident tile (every wave has taken its rows out of the R area: barrier)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue_chunk(0, 0);
        u32 word_hi = 0;   // sign bits of the even column tile of the current output word (this lane's row)
        for (int ct = 0; ct < a.ct; ++ct) {
            itq_f32x16 accM0, accM1, accC, accD;   // accC: x_lo R_hi (and -c_b), accD: x_hi R_lo: no two MFMAs in a row share an accumulator
            {
                // -c_b of this lane's 16 columns: register i <-> column (i & 3) + 8 (i >> 2) + 4 h
                itq_f32x4 cbv[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) cbv[g] = *reinterpret_cast<const itq_f32x4*>(lconst + ct * 32 + 8 * g + 4 * h);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    accM0[i] = 0.f;
                    accM1[i] = 0.f;
                    accD[i] = 0.f;
                    accC[i] = NORMED ? 0.f : -cbv[i >> 2][i & 3];   // without normalisation the sum ends as z~ = x . R_b - c_b
                }
            }
            asm volatile("s_nop 1" : "+v"(accM0), "+v"(accM1), "+v"(accC), "+v"(accD));   // VALU write of C -> MFMA read
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                {
                    const int c = ct * NKB + kb;
                    // chunk c: every wave's pieces have landed (own pieces: vmcnt; the others': the barrier), and every wave
                    // has finished reading the buffer chunk c + 1 is about to overwrite
                    if (!(a.debug & 1)) {   // (measurement: the R stream and its barriers left out)
                        wait_vmcnt<0>();
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                        if (c + 1 < n_chunks) issue_chunk(c + 1, (c + 1) & 1);
                    }
                    const unsigned char* cb_ptr = smem + (size_t)(c & 1) * ITQW_CHUNK_BYTES + r31 * 1024;
                    // R's fragments of k-step s + 1 are requested before the MFMAs of k-step s are issued: read after
                    // them, their ~130-cycle LDS latency followed every 96 cycles of matrix work and the pipe sat idle
                    // half the time (PMC: 31 % busy; the same loop without this: 43 % of the MFMA rate)
                    auto frag_ptr = [&](int s2) __attribute__((always_inline)) {
                        return cb_ptr + (s2 >> 3) * 256 + (((2 * (s2 & 7) + h) ^ (r31 & 15)) * 16);
                    };
                    itq_f16x8 bh_n = *reinterpret_cast<const itq_f16x8*>(frag_ptr(0));
                    itq_f16x8 bl_n = *reinterpret_cast<const itq_f16x8*>(frag_ptr(0) + 512);
#pragma unroll
                    for (int s = 0; s < 16; ++s) {
                        const int ks = kb * 16 + s;
                        const itq_f16x8 bh = bh_n, bl = bl_n;
                        if (s + 1 < 16) {
                            bh_n = *reinterpret_cast<const itq_f16x8*>(frag_ptr(s + 1));
                            bl_n = *reinterpret_cast<const itq_f16x8*>(frag_ptr(s + 1) + 512);
                        }
                        // A = R's fragment (VGPR), B = the resident row fragment (AGPR): D[column][row]
                        mfma_f16_agpr_b(accC, bh, xl[ks]);
                        if (kb == 0)
                            mfma_f16_agpr_b(accM0, bh, xh[ks]);
                        else
                            mfma_f16_agpr_b(accM1, bh, xh[ks]);
                        mfma_f16_agpr_b(accD, bl, xh[ks]);
                    }
                }
            }
            asm volatile("s_nop 15" : "+v"(accM0), "+v"(accM1), "+v"(accC), "+v"(accD));   // MFMA result -> VALU read
            // ---- column tile complete: lane = row r31, register i = column (i & 3) + 8 (i >> 2) + 4 h of the tile
            u32 bits = 0, unc = 0;
            if (!(a.debug & 4)) {
                itq_f32x4 cbv[4], eav[4], ebv[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    cbv[g] = *reinterpret_cast<const itq_f32x4*>(lconst + ct * 32 + 8 * g + 4 * h);
                    eav[g] = *reinterpret_cast<const itq_f32x4*>(lconst + 256 + ct * 32 + 8 * g + 4 * h);
                    ebv[g] = *reinterpret_cast<const itq_f32x4*>(lconst + 512 + ct * 32 + 8 * g + 4 * h);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int col = (i & 3) + 8 * (i >> 2) + 4 * h;
                    const float raw = (accM0[i] + accM1[i]) + (accC[i] + accD[i]);
                    float z;
                    if constexpr (NORMED)
                        z = __fmaf_rn(raw, rowscale, -cbv[i >> 2][i & 3]);
                    else
                        z = raw;
                    const float eps = __fmaf_rn(eav[i >> 2][i & 3], U, ebv[i >> 2][i & 3]);
                    bits |= (z >= 0.f ? 1u : 0u) << (31 - col);       // column 0 -> most significant
                    unc |= (!(fabsf(z) > eps) ? 1u : 0u) << col;      // true for a NaN; bit c = column c of the tile
                }
            }
            bits |= __shfl_xor(bits, 32);   // the other half of the lanes holds the other 16 columns of the same row
            unc |= __shfl_xor(unc, 32);
            {
                const int lo = a.pad - ct * 32;   // columns below `pad` are padding: never undecided
                const u32 valid = lo <= 0 ? ~0u : (lo >= 32 ? 0u : (~0u << lo));
                unc = bad_rows ? valid : (unc & valid);
            }
            const long long row = row0 + r31;
            const bool mine = active && lane < 32 && r31 >= (int)shift && row < a.n;
            if ((ct & 1) == 0) {
                word_hi = bits;
            } else if (mine) {
                u64 v = ((u64)word_hi << 32) | (u64)bits;
                if ((ct >> 1) == 0 && a.pad > 0) v &= (~0ull) >> a.pad;
                a.out[row * a.words + (ct >> 1)] = v;
            }
            {
                const bool need = mine && unc != 0;
                const u64 nb = __ballot(need);
                const u32 p = wcount + __builtin_amdgcn_mbcnt_hi((u32)(nb >> 32), __builtin_amdgcn_mbcnt_lo((u32)nb, 0u));
                if (need && (long long)p < a.seg_cap) myseg[p] = ((u64)((u32)row | ((u32)ct << 29)) << 32) | (u64)unc;
                wcount += (u32)__popcll(nb);
            }
        }
        if (a.stamps && wave == 0 && lane == 0 && round < 20) a.stamps[(size_t)blockIdx.x * 64 + 3 * round + 2] = wall_clock64();
        // every wave is done with the R buffers: the next round's rows may land there
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    if (lane == 0) a.seg_cnt[wave_id] = wcount;
}

// The wide filter's undecided bits, one float64 evaluation each, in the reference's arithmetic (v = x / |x| in x's
// dtype with numpy's norm, minus the mean in the promoted dtype; z_b = sum_k v_k R[k][b]): itq_fix_bits_kernel for rows
// of T up to 512 elements and 8 column tiles.  32 lanes per entry; lane l holds elements 4l .. 4l+3 of every
// 128-element stretch.
template <class T>
static __global__ __launch_bounds__(256) void itq_fix_bits_wide_kernel(ItqArgs a, const u64* __restrict__ seg,
                                                                        const u32* __restrict__ seg_cnt, long long seg_cap,
                                                                        const double* __restrict__ rt64) {
    const long long w = blockIdx.x;
    const long long cnt_raw = seg_cnt[w];
    const u32 cnt = (u32)(cnt_raw < seg_cap ? cnt_raw : seg_cap);
    const int l32 = threadIdx.x & 31, slot = (threadIdx.x >> 5) + 8 * blockIdx.y;
    const T* X = reinterpret_cast<const T*>(a.x);
    const u32 STEP = 8 * gridDim.y;
    for (u32 e = slot; e < cnt; e += STEP) {   // uniform within a 32-lane half wave
        const u64 ent = seg[w * seg_cap + e];
        const long long row = (long long)((u32)(ent >> 32) & 0x1fffffffu);
        const int ct = (int)((ent >> 61) & 7u);
        u32 mask = (u32)ent;
        const T* xr = X + row * a.d;
        T nrm = (T)1;
        if (a.norm == SQ_NORM_L2) {
            // numpy's pairwise order: eight cooperating lanes (np_pairwise_sum), every aligned group of 8 gets the value
            auto term = [xr](int i) { return mul_rn(xr[i], xr[i]); };
            nrm = sqrt_rn(np_pairwise_sum<T>(term, a.d, threadIdx.x & 7));
            if (nrm == (T)0) nrm = (T)1;
        }
        double v[4][4];   // this lane's elements: 4 l32 + 128 t + 0..3
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = 128 * t + 4 * l32;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                double val = 0.0;
                if (k < a.d) {
                    T xe = xr[k + j];
                    if (a.norm == SQ_NORM_L2) xe = div_rn(xe, nrm);
                    if constexpr (sizeof(T) == 4)
                        val = a.sub32 ? (double)__fsub_rn(xe, (float)a.mean[k + j]) : __dsub_rn((double)xe, a.mean[k + j]);
                    else
                        val = __dsub_rn(xe, a.mean[k + j]);
                }
                v[t][j] = val;
            }
        }
        while (mask) {
            const int pc = ct * 32 + __ffs((int)mask) - 1;   // padded column; the filter only flags pc >= pad
            mask &= mask - 1;
            const double* rcol = rt64 + (long long)pc * a.d;   // column pc of R, contiguous (itq_fast_prep_kernel)
            double z = 0.0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int k = 128 * t + 4 * l32;
                if (k < a.d) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) z = __fma_rn(v[t][j], rcol[k + j], z);
                }
            }
            z += __shfl_xor(z, 16);
            z += __shfl_xor(z, 8);
            z += __shfl_xor(z, 4);
            z += __shfl_xor(z, 2);
            z += __shfl_xor(z, 1);
            if (l32 == 0) {
                unsigned long long* word = reinterpret_cast<unsigned long long*>(a.out + row * a.words + (pc >> 6));
                const unsigned long long bit = 1ull << (63 - (pc & 63));
                if (z >= 0.0)
                    atomicOr(word, bit);
                else
                    atomicAnd(word, ~bit);
            }
        }
    }
}

}  // namespace sq
