// The middle tier of the dense L2 search: a second, tighter filter for the queries the bfloat16 scan could not
// certify, in front of the exact all-rows path (dense_exact_group_kernel).
//
// The first filter scores the rows from a bfloat16 copy (row error 2^-8 |x|): its slack is what it is, and a query
// whose k-th distance sits inside it -- a query far longer than the rows, a point set of very low dimension, the
// distances the reference computes in metrics.euclidean_distance (smqtk_indexing/utils/metrics.py:73-86) packed
// within the bound -- took the exact path: every row's distance in numpy's own arithmetic for 8 queries per pass,
// 15x the cost of a filtered batch at 10 M x 128.  Here the ORIGINAL float32 rows are streamed once (LDS-DMA ring
// of 32-row x 64-float units, the shape of the ITQ filter's ring), centred, split in registers into two bfloat16
// planes x' = x_hi + x_lo + r (|r| < 2^-15 |x'|) and scored against both planes of the query,
//     s~ = n' c1 + x_hi q_hi + x_hi q_lo + x_lo q_hi        (v_mfma_f32_32x32x16_bf16, q' = -2 (q - c)),
// an error of 2^-13 |x'||q - c| (+ the float32 accumulation) instead of 2^-7: the slack shrinks 64-fold.  Survivors
// leave exactly as the first filter's do (per-wave segments of (first row, mask, query) entries), go through the same
// exact re-rank and select, and DenseFinalizeL2 certifies them against the tighter bound; what still fails goes on to
// the exact path, so results never depend on either filter being right.
// The threshold of a query does not come from the first filter's bound (its slack is what failed): a small pre-pass
// evaluates the TRUE scores of every 64th row in float64 -- their k-th smallest bounds the k-th smallest of all rows
// with no filter error in it -- and, when the first filter's candidate list was complete, the k-th exact distance
// found there is used if it is smaller.
// Shapes: d <= 512, rows 16-byte aligned with a stride of whole 16-byte chunks (everything else keeps the exact path).
//
// Cosine (round 4; metrics.cosine_similarity, smqtk_indexing/utils/metrics.py:89-137): the same stream with the rows
// scaled to unit length in registers (x^ = x / |x|, float32) and the score taken about the column means c of the rows,
//     -x^.q/|q| = -(x^.c)/|q| - x^.(q - c)/|q|  ->  s~ = u_row w_q + [x^_hi q'_hi + x^_hi q'_lo + x^_lo q'_hi],
// u_row = x^.c (float64 dot, stored float32, computed once per index), w_q = -1/|q|, q' = -(q - c)/|q|: only the
// second term goes through bfloat16, so the error is 2^-14 |q - c|/|q| instead of the first filter's 2^-8 -- for
// descriptors that share a large offset (|q - c| << |q|: every cosine similarity within 1e-3 of 1, all rows inside the
// first filter's slack) that is what certifies them; data about the origin has c ~ 0 and keeps 2^-14.  Each query
// carries its own slack eps_q (dense_mid_cos_queries_kernel), the threshold comes from sampled true similarities.
#pragma once
#include "sq_dense_scan.hpp"

namespace sq {

static constexpr int MID_UNIT_BYTES = 32 * 256;            // 32 rows x 64 floats
static constexpr int MID_SLOT_BYTES = MID_UNIT_BYTES + NORM_BYTES;  // + the tile's 32 stored norms (64 lanes x 4 B land)
static constexpr int MID_NSTAGE = 2;
static constexpr int MID_MAX_Q = 32;                        // queries per pass (one MFMA tile)

// error coefficient of the three-product bf16 score (header comment): products 2^-13 |x'||q - c|
static constexpr double kEpsAMid = 1.220703125e-04;

struct DenseMidArgs {
    const float* x;        // [n][ld] float32 rows
    long long n, ld;
    int d;                 // <= 512 (the last 64-float unit of a row may be partial)
    const float* center;   // [d_pad] or nullptr
    const float* norms;    // [n_pad] the first filter's stored norms n' = RD(|x'|^2 (1 - alpha1))
    float norm_scale;      // c1 = (1 - alpha_mid) / (1 - alpha1), rounded down
    const uint4* qs;       // [32][d_pad/4] prepared planes of the pass's queries (dense_prep_queries_kernel)
    int d_pad;
    const float* thr;      // [32]
    uint2* wave_out;
    u32* wave_cnt;
    u32 wave_cap;
    long long n_tiles;
    int nrb;
    // cosine (COS = true): rowstat = [2][rowstat_ld] float32, 1/|x| and u = x^.c per row; qw[32] = -1/|q|
    const float* rowstat;
    long long rowstat_ld;
    const float* qw;
};

// hi = x rounded half-up to bfloat16 (as float32), lo = x - hi (exact); packs of two: hi words / truncated lo words
__device__ __forceinline__ void mid_split_pair(float x0, float x1, u32& hi_pack, u32& lo_pack) {
    const u32 h0 = (__float_as_uint(x0) + 0x8000u) & 0xffff0000u, h1 = (__float_as_uint(x1) + 0x8000u) & 0xffff0000u;
    const float l0 = __fsub_rn(x0, __uint_as_float(h0)), l1 = __fsub_rn(x1, __uint_as_float(h1));
    hi_pack = (h0 >> 16) | h1;
    lo_pack = (__float_as_uint(l0) >> 16) | (__float_as_uint(l1) & 0xffff0000u);
}

template <int WAVES, bool COS = false>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void dense_mid_scan_kernel(DenseMidArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r31 = lane & 31, h = lane >> 5;
    const int D = a.d, DPAD = a.d_pad, KU = (D + 63) / 64;
    // rows that do not fill their last 64-float unit: the lanes whose 16-byte source chunk lies beyond the row fetch the
    // unit's first chunk instead (in bounds; the query planes are zero there, so a finite value contributes nothing and a
    // non-finite one belongs to a row whose score is NaN anyway)
    const int tail_chunks = (D & 63) ? ((D & 63) + 3) / 4 : 16;
    // LDS: [query planes 32 x DPAD*4][centre DPAD*4][rings]
    const u32 q_bytes = (u32)TILE_ROWS * DPAD * 4, c_bytes = (u32)DPAD * 4;
    const u32 lds_base = (u32)(uintptr_t)smem;
    const u32 ring_base = lds_base + q_bytes + c_bytes + (u32)wave * (MID_NSTAGE * MID_SLOT_BYTES);
    const unsigned char* ring_ptr = smem + q_bytes + c_bytes + wave * (MID_NSTAGE * MID_SLOT_BYTES);
    const float* cen = reinterpret_cast<const float*>(smem + q_bytes);
    {
        const int cpr = DPAD / 4;
        for (int c = threadIdx.x; c < TILE_ROWS * cpr; c += WAVES * 64) {
            const int r = c / cpr, ch = c - r * cpr;
            const int sw = (ch & ~15) | ((ch & 15) ^ (r & 15));
            *reinterpret_cast<uint4*>(smem + (u32)r * DPAD * 4 + sw * 16) = a.qs[c];
        }
        for (int i = threadIdx.x; i < DPAD; i += WAVES * 64)
            reinterpret_cast<float*>(smem + q_bytes)[i] = (!COS && a.center && i < D) ? a.center[i] : 0.f;
    }
    __syncthreads();
    const long long wave_id = (long long)blockIdx.x * WAVES + wave;
    const long long nwaves = (long long)a.nrb * WAVES;
    const long long my_tiles = wave_id < a.n_tiles ? (a.n_tiles - wave_id + nwaves - 1) / nwaves : 0;
    const long long total_units = my_tiles * KU;
    uint2* wout = a.wave_out + wave_id * a.wave_cap;
    float thr_l = a.thr[r31];
    float qw_l = COS ? a.qw[r31] : 0.f;
    asm volatile("" : "+v"(thr_l), "+v"(qw_l));   // complete before the ring starts (a compiler-visible load inside the loop drains it)

    u32 voff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = 4 * j + (lane >> 4);
        voff[j] = (u32)((long long)r * a.ld * 4 + (((lane & 15) ^ (r & 15)) * 16));
    }
    // cosine: lanes 0-31 fetch 1/|x| of the tile's rows, lanes 32-63 their u (one DMA piece, as the L2 norms are)
    const u32 voff_norm = COS ? (u32)(((long long)h * a.rowstat_ld + r31) * 4) : (u32)((lane & 31) * 4);
    long long iss_tile = wave_id;
    int iss_kc = 0, iss_slot = 0;
    long long issued = 0;
    auto issue_next = [&]() __attribute__((always_inline)) {
        if (issued >= total_units) return;
        long long row0 = iss_tile * 32;
        if (row0 + 32 > a.n) row0 = a.n - 32;   // the last tile: the window moves back, rows below `shift` are masked out
        const u32 dst = ring_base + (u32)iss_slot * MID_SLOT_BYTES;
        const unsigned char* base = reinterpret_cast<const unsigned char*>(a.x) + row0 * a.ld * 4 + iss_kc * 256;
        if (tail_chunks < 16 && iss_kc == KU - 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int sc = (lane & 15) ^ ((4 * j + (lane >> 4)) & 15);
                glds16<true>(base, sc < tail_chunks ? voff[j] : voff[j] - (u32)sc * 16u, dst + (u32)j * 1024);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) glds16<true>(base, voff[j], dst + (u32)j * 1024);
        }
        if (iss_kc == 0) glds4((COS ? a.rowstat : a.norms) + row0, voff_norm, dst + MID_UNIT_BYTES);
        ++issued;
        if (++iss_kc == KU) {
            iss_kc = 0;
            iss_tile += nwaves;
        }
        if (++iss_slot == MID_NSTAGE) iss_slot = 0;
    };
    for (int p = 0; p < MID_NSTAGE; ++p) issue_next();

    u32 wcount = 0;
    long long consumed = 0;
    int rd_slot = 0;
    for (long long tile = wave_id; tile < a.n_tiles; tile += nwaves) {
        long long row0 = tile * 32;
        const long long shift = row0 + 32 > a.n ? row0 + 32 - a.n : 0;
        row0 -= shift;
        f32x16 acc;
        f32x4 ur[4];       // cosine: u of the rows this lane's accumulators belong to
        float inv_l = 0.f; // cosine: 1/|x| of this lane's row (A operand: row = lane & 31)
        for (int kc = 0; kc < KU; ++kc) {
            // a unit has 8 DMA pieces, a tile's first unit 9 (its norms): allowing 8 per younger unit is exact for the
            // others and merely conservative for a first unit (allowing MORE than a younger unit holds would let the
            // awaited unit's own pieces count as "younger")
            if (KU == 1)
                wait_units_in_flight<MID_NSTAGE, 9>((int)(issued - consumed - 1));
            else
                wait_units_in_flight<MID_NSTAGE, 8>((int)(issued - consumed - 1));
            const unsigned char* sl = ring_ptr + rd_slot * MID_SLOT_BYTES;
            f32x4 xa[4][2];
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    xa[s2][e] = *reinterpret_cast<const f32x4*>(sl + r31 * 256 + (((4 * s2 + 2 * h + e) ^ (r31 & 15)) * 16));
            if (kc == 0) {
                if constexpr (COS) {
                    inv_l = *reinterpret_cast<const float*>(sl + MID_UNIT_BYTES + r31 * 4);
#pragma unroll
                    for (int c = 0; c < 4; ++c) ur[c] = *reinterpret_cast<const f32x4*>(sl + MID_UNIT_BYTES + 128 + (8 * c + 4 * h) * 4);
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
                } else {
                    f32x4 nr[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) nr[c] = *reinterpret_cast<const f32x4*>(sl + MID_UNIT_BYTES + (8 * c + 4 * h) * 4);
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[i] = nr[i >> 2][i & 3] * a.norm_scale;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the unit is in registers: its slot is free
            ++consumed;
            if (++rd_slot == MID_NSTAGE) rd_slot = 0;
            issue_next();
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                const int ks = kc * 4 + s2;                 // global k-step of 16
                // centre of this lane's 8 elements (two LDS reads per k-step; lanes of one half read the same words)
                u32 hw[4], lw[4];
                if constexpr (COS) {
#pragma unroll
                    for (int j = 0; j < 4; j += 2) {
                        mid_split_pair(__fmul_rn(xa[s2][0][j], inv_l), __fmul_rn(xa[s2][0][j + 1], inv_l), hw[j >> 1], lw[j >> 1]);
                        mid_split_pair(__fmul_rn(xa[s2][1][j], inv_l), __fmul_rn(xa[s2][1][j + 1], inv_l), hw[2 + (j >> 1)], lw[2 + (j >> 1)]);
                    }
                } else {
                    const f32x4 c0 = *reinterpret_cast<const f32x4*>(cen + ks * 16 + 8 * h);
                    const f32x4 c1 = *reinterpret_cast<const f32x4*>(cen + ks * 16 + 8 * h + 4);
#pragma unroll
                    for (int j = 0; j < 4; j += 2) {
                        mid_split_pair(__fsub_rn(xa[s2][0][j], c0[j]), __fsub_rn(xa[s2][0][j + 1], c0[j + 1]), hw[j >> 1], lw[j >> 1]);
                        mid_split_pair(__fsub_rn(xa[s2][1][j], c1[j]), __fsub_rn(xa[s2][1][j + 1], c1[j + 1]), hw[2 + (j >> 1)], lw[2 + (j >> 1)]);
                    }
                }
                const bf16x8 ah = __builtin_bit_cast(bf16x8, f32x4{__uint_as_float(hw[0]), __uint_as_float(hw[1]), __uint_as_float(hw[2]), __uint_as_float(hw[3])});
                const bf16x8 al = __builtin_bit_cast(bf16x8, f32x4{__uint_as_float(lw[0]), __uint_as_float(lw[1]), __uint_as_float(lw[2]), __uint_as_float(lw[3])});
                // query planes of k-step ks: k-unit ks >> 3 (512 bytes per row: 256 hi, 256 lo), chunk 2 (ks & 7) + h
                const unsigned char* brow = smem + (u32)r31 * DPAD * 4 + (ks >> 3) * 512 + ((2 * (ks & 7) + h) ^ (r31 & 15)) * 16;
                const bf16x8 bh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(brow));
                const bf16x8 bl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(brow + 256));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
            }
        }
        // ---- tile complete: scores of 32 rows x 32 queries (lane = query, register i = row (i&3)+8(i>>2)+4h)
        if constexpr (COS) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __fmaf_rn(ur[i >> 2][i & 3], qw_l, acc[i]);
        }
        float m = acc[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) m = fminf(m, acc[i]);
        const u64 hit = __ballot(m <= thr_l);
        if (hit != 0) {
            u32 mask = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rr = (i & 3) + 8 * (i >> 2) + 4 * h;   // row of the tile window
                if (acc[i] <= thr_l && rr >= (int)shift) mask |= 1u << i;
            }
            const u64 bal = __ballot(mask != 0);
            if (mask) {
                const u32 pos = wcount + __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u));
                if (pos < a.wave_cap) wout[pos] = make_uint2((u32)(row0 + 4 * h), (mask << 16) | (u32)r31);
            }
            wcount += (u32)__popcll(bal);
        }
    }
    if (lane == 0) {
        a.wave_cnt[2 * wave_id] = wcount;
        a.wave_cnt[2 * wave_id + 1] = 0u;   // first query tile of the group
    }
}

// The pass's queries: rows idx[0 .. count) of the call's query matrix, gathered into [32][d] (zero padded)
struct MidSelection {
    int idx[MID_MAX_Q];
    int count;
};
static __global__ void dense_mid_gather_kernel(const float* __restrict__ q, int d, MidSelection sel, float* __restrict__ out,
                                               int* __restrict__ qmap) {
    const int j = blockIdx.x;
    for (int i = threadIdx.x; i < d; i += blockDim.x) out[(long long)j * d + i] = j < sel.count ? q[(long long)sel.idx[j] * d + i] : 0.f;
    if (threadIdx.x == 0 && j < sel.count) qmap[j] = sel.idx[j];
}

// Upper bound of the true k-th score for the pass's queries, independent of the first filter's slack (which is what
// failed): the TRUE scores s = |x - q|^2 - |q - c|^2 of every `stride`-th row, in float64, rounded up to float32;
// the k-th smallest of them bounds the k-th smallest over all rows (kth_threshold_f32_kernel picks it, DenseMidThrPost
// turns it into the tier's threshold).  One half-wave per sampled row, lane = query; the queries sit in LDS.
static __global__ __launch_bounds__(256) void dense_mid_sample_kernel(const float* __restrict__ db, long long ld, int d, long long n,
                                                                       long long stride, long long ns,
                                                                       const float* __restrict__ q_mid,   // [32][d]
                                                                       const double* __restrict__ qn2_mid,
                                                                       float* __restrict__ sample) {      // [32][ns]
    extern __shared__ float lq[];   // [d][33]: element k of query j at k * 33 + j (conflict-free for lane = query)
    for (int i = threadIdx.x; i < MID_MAX_Q * d; i += 256) {
        const int j = i / d, k = i - j * d;
        lq[k * 33 + j] = q_mid[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, j = lane & 31;
    const long long i = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + (lane >> 5);   // sampled row index
    if (i >= ns) return;
    const long long row = i * stride < n ? i * stride : n - 1;
    const float* x = db + row * ld;
    double acc = 0.0;
    int k = 0;
    for (; k + 4 <= d; k += 4) {   // (rows are 16-byte aligned with a stride of whole chunks: dense_mid_shape_ok)
        const float4 xv = *reinterpret_cast<const float4*>(x + k);
        const double t0 = (double)xv.x - (double)lq[k * 33 + j], t1 = (double)xv.y - (double)lq[(k + 1) * 33 + j];
        const double t2 = (double)xv.z - (double)lq[(k + 2) * 33 + j], t3 = (double)xv.w - (double)lq[(k + 3) * 33 + j];
        acc = fma(t0, t0, acc);
        acc = fma(t1, t1, acc);
        acc = fma(t2, t2, acc);
        acc = fma(t3, t3, acc);
    }
    for (; k < d; ++k) {
        const double t = (double)x[k] - (double)lq[k * 33 + j];
        acc = fma(t, t, acc);
    }
    const double sc = acc * (1.0 + 1e-12) - qn2_mid[j] * (1.0 - 1e-12);
    float r = (float)sc;
    if ((double)r < sc) r = __uint_as_float(__float_as_uint(r) + (r >= 0.f ? 1 : -1));   // round up
    sample[(long long)j * ns + i] = r;
}

// Threshold of the middle tier for query j of the pass (original index qmap[j]): U = the k-th smallest sampled TRUE
// score (an upper bound of the true k-th score: no filter error in it), or -- when the first filter's candidate list
// was complete (status bits 1 and 4 clear) -- the k-th exact distance found there, whichever is smaller.  The tier's
// threshold is T' = U + beta_mid |q''|^2 (+ rounding), as DenseThrPost forms it: every row of the true top-k has a
// tier score s~ <= T' (FilterBound: s~ <= s + beta |q|^2).
struct DenseMidThrPost {
    const int* qmap;
    const float* out_dist;
    const u32* status1;
    int k, kk;
    const double* qn2_mid;
    double beta_mid;
    __device__ __forceinline__ void prologue(int, double*) const {}
    __device__ __forceinline__ float operator()(int j, float t) const {
        const int q = qmap[j];
        const double Q = qn2_mid[j];
        double U = (double)t;
        if ((status1[q] & 5u) == 0u) {
            const double dk = (double)out_dist[(long long)q * k + (kk - 1)];
            if (dk == dk && dk < (double)__builtin_inff()) {
                const double u2 = dk * dk * (1.0 + 1e-6) - Q;   // numpy's float32 distance: a few ulp
                if (u2 < U) U = u2;
            }
        }
        if (!(U < (double)__builtin_inff())) return __builtin_inff();
        const double tt = U + beta_mid * Q + 4e-6 * fabs(U + Q);
        float r = (float)tt;
        if ((double)r < tt) r = __uint_as_float(__float_as_uint(r) + (r >= 0.f ? 1 : -1));
        return r;
    }
};

// ------------------------------------------------------------------------------------------------ cosine
// Per-row statistics of the cosine tier, computed once per index (again after an append): 1/|x| (float32 of the float64
// value; 0 for a zero row, whose true distance is NaN and ranks last) and u = x.c/|x| with the dot product in float64.
// Eight lanes per row.
static __global__ __launch_bounds__(256) void dense_mid_cos_rows_kernel(const float* __restrict__ db, long long n, long long ld, int d,
                                                                         const float* __restrict__ center,
                                                                         const double* __restrict__ nx64,
                                                                         float* __restrict__ rowstat, long long rowstat_ld) {
    const int j8 = threadIdx.x & 7;
    const long long row = (long long)blockIdx.x * 32 + (threadIdx.x >> 3);
    const long long rc = row < n ? row : n - 1;
    const float* x = db + rc * ld;
    double dot = 0.0;
    for (int i = j8; i < d; i += 8) dot = fma((double)x[i], (double)center[i], dot);
    dot += __shfl_xor(dot, 1);
    dot += __shfl_xor(dot, 2);
    dot += __shfl_xor(dot, 4);
    if (j8 == 0 && row < n) {
        const double nx = nx64[row];
        const double inv = nx > 0.0 ? 1.0 / sqrt(nx) : 0.0;
        rowstat[row] = (float)inv;
        const double u = dot * inv;
        rowstat[rowstat_ld + row] = (u == u && fabs(u) < 3.0e38) ? (float)u : 0.f;   // (non-finite rows: their scores are NaN through x^ anyway)
    }
}

// The pass's queries for the cosine tier (one workgroup per query slot, the layout of dense_prep_queries_kernel):
// planes of q' = -(q_s - c)/|q_s|, w = -1/|q_s| (q_s: the query at the length closest to c), the query's slack
//     eps = (2^-14 + eps_b) |q_s - c|/|q_s| (1 + 1e-5) + 1e-6 (1 + |c|/|q_s|)
// (bf16 three-product error and float32 accumulation of the q' term; the float32 roundings of x^, q', u, w and of the
// final fma), the float32 copy the re-rank reads, and the per-call state (threshold, counters, overflow flag).
static __global__ __launch_bounds__(256) void dense_mid_cos_queries_kernel(const float* __restrict__ q, int nq, int d, int d_pad,
                                                                            const float* __restrict__ center, double eps_b,
                                                                            uint4* __restrict__ qs, double* __restrict__ qn2,
                                                                            float* __restrict__ thr, u32* __restrict__ cnt,
                                                                            u32* __restrict__ oflag, float* __restrict__ q_al, int ldq,
                                                                            float* __restrict__ qw, float2* __restrict__ lin) {
    const int qi = blockIdx.x;
    __shared__ double red[3][4];
    if (threadIdx.x == 0) {
        thr[qi] = -__builtin_inff();
        cnt[qi] = 0u;
        if (qi == 0) *oflag = 0u;
    }
    if (qi < nq)
        for (int i = threadIdx.x; i < ldq; i += 256) q_al[(long long)qi * ldq + i] = i < d ? q[(long long)qi * d + i] : 0.f;
    // cosine does not see the query's length: the query is taken at the length that makes |q_s - c|/|q_s| smallest,
    // q_s = s q with s = |c|^2/(q.c) (q_s - c perpendicular to c: the ratio is the sine of the angle between q and c, and
    // |c|/|q_s| its cosine); a query at more than 90 degrees from c is taken so long that c does not matter (ratio -> 1)
    double a_q = 0.0, a_qc = 0.0, a_c = 0.0;
    if (qi < nq)
        for (int i = threadIdx.x; i < d; i += 256) {
            const double v = (double)q[(long long)qi * d + i], c = (double)center[i];
            a_q += v * v;
            a_qc += v * c;
            a_c += c * c;
        }
    for (int o = 32; o > 0; o >>= 1) {
        a_q += __shfl_xor(a_q, o);
        a_qc += __shfl_xor(a_qc, o);
        a_c += __shfl_xor(a_c, o);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = a_q;
        red[1][threadIdx.x >> 6] = a_qc;
        red[2][threadIdx.x >> 6] = a_c;
    }
    __syncthreads();
    const double tq = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    const double tqc = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const double tc = red[2][0] + red[2][1] + red[2][2] + red[2][3];
    const bool ok = qi < nq && tq > 0.0 && tq < 1e300;
    double sq = 1.0;
    if (ok && tqc > 0.0 && tc > 0.0) sq = tc / tqc;
    else if (ok && tc > 0.0) sq = 1e3 * sqrt(tc / tq);
    if (!(sq > 1e-30 && sq < 1e30)) sq = 1.0;
    const double len = ok ? sq * sqrt(tq) : 0.0;       // |q_s|
    const double scale = ok ? -1.0 / len : 0.0;
    __syncthreads();
    double a_d = 0.0;                                  // |q_s - c|^2, summed directly (the expanded form cancels)
    if (ok)
        for (int i = threadIdx.x; i < d; i += 256) {
            const double t = (double)q[(long long)qi * d + i] * sq - (double)center[i];
            a_d += t * t;
        }
    for (int o = 32; o > 0; o >>= 1) a_d += __shfl_xor(a_d, o);
    if ((threadIdx.x & 63) == 0) red[1][threadIdx.x >> 6] = a_d;
    __syncthreads();
    const double td = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    if (threadIdx.x == 0) {
        qn2[qi] = qi < nq ? tq : 0.0;                  // (the TRUE |q|^2: the sampled scores divide by it)
        qw[qi] = (float)scale;
        const double rd = ok ? sqrt(td) / len : 0.0, rc = ok ? sqrt(tc) / len : 0.0;
        const double eps = (6.103515625e-05 + eps_b) * rd * (1.0 + 1e-5) + 1e-6 * (1.0 + rc);
        lin[qi] = make_float2(0.f, ok ? (float)(eps * (1.0 + 1e-6)) : __builtin_inff());   // (a zero / non-finite query: nothing certifies)
    }
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
                if (ok && k < d) x = (float)(((double)q[(long long)qi * d + k] * sq - (double)center[k]) * scale);
                u32 hi, lo;
                bf16_split(x, hi, lo);
                half[e] = p ? lo : hi;
            }
            w[j >> 1] = half[0] | (half[1] << 16);
        }
        qs[(long long)qi * cpr + cc] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// TRUE scores -x.q/(|x||q|) of every `stride`-th row in float64, rounded up to float32 (NaN -- a zero or non-finite row
// or query -- becomes +inf: such a row bounds nothing).  The layout of dense_mid_sample_kernel.
static __global__ __launch_bounds__(256) void dense_mid_cos_sample_kernel(const float* __restrict__ db, long long ld, int d, long long n,
                                                                           long long stride, long long ns,
                                                                           const float* __restrict__ q_mid,   // [32][d]
                                                                           const double* __restrict__ qn2_mid,
                                                                           const double* __restrict__ nx64,
                                                                           float* __restrict__ sample) {      // [32][ns]
    extern __shared__ float lq[];
    for (int i = threadIdx.x; i < MID_MAX_Q * d; i += 256) {
        const int j = i / d, k = i - j * d;
        lq[k * 33 + j] = q_mid[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, j = lane & 31;
    const long long i = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + (lane >> 5);
    if (i >= ns) return;
    const long long row = i * stride < n ? i * stride : n - 1;
    const float* x = db + row * ld;
    double acc = 0.0;
    int k = 0;
    for (; k + 4 <= d; k += 4) {
        const float4 xv = *reinterpret_cast<const float4*>(x + k);
        acc = fma((double)xv.x, (double)lq[k * 33 + j], acc);
        acc = fma((double)xv.y, (double)lq[(k + 1) * 33 + j], acc);
        acc = fma((double)xv.z, (double)lq[(k + 2) * 33 + j], acc);
        acc = fma((double)xv.w, (double)lq[(k + 3) * 33 + j], acc);
    }
    for (; k < d; ++k) acc = fma((double)x[k], (double)lq[k * 33 + j], acc);
    const double sc = -acc / sqrt(nx64[row] * qn2_mid[j]) + 1e-12;
    float r = (float)sc;
    if ((double)r < sc) r = __uint_as_float(__float_as_uint(r) + (r >= 0.f ? 1 : -1));
    if (!(sc == sc) || !(fabs(sc) <= 2.0)) r = __builtin_inff();
    sample[(long long)j * ns + i] = r;
}

// Threshold of the cosine tier for query j of the pass: U = the k-th smallest sampled TRUE score (-similarity), or the
// score of the k-th exact distance the first filter found when its list was complete, whichever is smaller; every row of
// the true top-k has a tier score <= U + eps_j.  With fewer than k usable samples nothing passes (-inf: the query goes
// on to the exact path with status "fewer than k candidates").
struct DenseMidCosThrPost {
    const int* qmap;
    const double* out_dist;
    const u32* status1;
    int k, kk;
    const float2* lin;
    __device__ __forceinline__ void prologue(int, double*) const {}
    __device__ __forceinline__ float operator()(int j, float t) const {
        const int q = qmap[j];
        double U = (double)t;
        if ((status1[q] & 5u) == 0u) {
            const double dk = out_dist[(long long)q * k + (kk - 1)];
            if (dk == dk && dk >= 0.0 && dk <= 2.0) {
                const double u2 = -cos(dk * 1.5707963267948966) + 1e-9;
                if (u2 < U) U = u2;
            }
        }
        const double eps = (double)lin[j].y;
        if (!(U < (double)__builtin_inff()) || !(eps < 1.0)) return -__builtin_inff();
        const double tt = U + eps + 1e-9;
        float r = (float)tt;
        if ((double)r < tt) r = __uint_as_float(__float_as_uint(r) + (r >= 0.f ? 1 : -1));
        return r;
    }
};

}  // namespace sq
