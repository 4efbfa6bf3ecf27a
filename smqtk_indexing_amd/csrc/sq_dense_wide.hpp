// The first-stage filter for rows wider than 512 dimensions (gfx950).        (included by sq_dense.hip)
//
// The reference's arithmetic takes any d (smqtk_indexing/utils/metrics.py:73-86) and its own examples index 2048- and
// 4096-dimensional descriptors (docs/examples/caffe_build_index.rst:35); up to round 3 such rows had no filter at all: every
// query took the exact path, a pass over the float32 matrix per 8 queries.  dense_scan_kernel keeps the query tile's
// fragments in registers or in LDS, which ends at d_pad = 512 (32 queries x d_pad x 2 planes x 2 bytes: 512 KB at 4096).
// Here a wave owns a 32-row tile and walks its k-units (128 columns = 256 bytes of the bfloat16 scan copy per row) with
// the accumulators of ONE query tile in registers; the row fragments AND the query fragments of a k-unit are plain global
// loads straight into the MFMA operand registers, the next unit's requested before this unit's sixteen
// v_mfma_f32_32x32x16_bf16.  The scan copy is stored in fragment order, so a lane's 16 bytes are a lane's operand;
// the query planes (512 KB for 32 queries at 4096 dimensions) stay in L2.  Same inputs, outputs and error bound as
// dense_scan_kernel's one-tile configuration (DenseScanArgs; scores = n' + x_hi (q_hi + q_lo)): the
// sample pass, the threshold, the exact re-rank, the select and the certification around it are unchanged.
// No LDS, no ring: the compiler's own counted waits order the loads (a DMA ring as in dense_scan_kernel would need the
// query unit staged per wave: 16 KB per stage beside the rows' 8).
#pragma once
#include "sq_dense_scan.hpp"

namespace sq {

static constexpr int WIDE_WAVES = 4;

template <int QP, bool SAMPLE>
__global__ __launch_bounds__(WIDE_WAVES * 64, 2) void dense_wide_scan_kernel(DenseScanArgs a, int ku) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r31 = lane & 31, h = lane >> 5;
    const long long wave_id = (long long)blockIdx.x * WIDE_WAVES + wave;   // unique per wave of the launch
    uint2* wout = a.wave_out + wave_id * a.wave_cap;
    // block -> (row block, query tile): dense_scan_kernel's mapping (blocks that share an XCD walk the tiles of the same rows)
    const int L = blockIdx.x;
    int qt, rb;
    if (a.nqt > 1) {
        const int xcd = L & 7, j = L >> 3;
        qt = j % a.nqt;
        rb = (j / a.nqt) * 8 + xcd;
    } else {
        qt = 0;
        rb = L;
    }
    const long long gw = (long long)rb * WIDE_WAVES + wave;
    const long long nwaves = (long long)a.nrb * WIDE_WAVES;
    const size_t dpad = (size_t)ku * KT;
    const int qglob = qt * TILE_ROWS + r31;
    const float thr_l = SAMPLE ? 0.f : ((a.debug & 4) ? -__builtin_inff() : a.thr[qglob]);
    // this lane's query: per k-unit 256 bytes of q_hi then 256 of q_lo, chunk 2 s + h of each = k-step s
    const unsigned char* qrow = reinterpret_cast<const unsigned char*>(a.qs) + (size_t)qglob * dpad * 4 + (size_t)h * 16;
    u32 tail_mask = 0;   // rows of the last, partial tile that exist (bit i <-> accumulator register i of this lane)
    {
        const int nvalid = (int)(a.n - (a.n_tiles - 1) * TILE_ROWS);
#pragma unroll
        for (int i = 0; i < 16; ++i) tail_mask |= ((i & 3) + 8 * (i >> 2) + 4 * h < nvalid ? 1u : 0u) << i;
    }
    u32 wcount = 0;
    for (long long sel = gw; sel < a.n_sel; sel += nwaves) {
        const long long row0 = sel * a.tile_step * TILE_ROWS;
        const unsigned char* arow = reinterpret_cast<const unsigned char*>(a.scan) + (size_t)(row0 + r31) * dpad * 2 + (size_t)h * 16;
        f32x4 av[8], bh[8], bl[QP == 2 ? 8 : 1];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            av[g] = *reinterpret_cast<const f32x4*>(arow + g * 32);
            bh[g] = *reinterpret_cast<const f32x4*>(qrow + g * 32);
            if constexpr (QP == 2) bl[g] = *reinterpret_cast<const f32x4*>(qrow + 256 + g * 32);
        }
        // the accumulator starts from the rows' stored norms n' (0 for cosine): score = n' + x . q'
        f32x16 acc;
        if (a.norms) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4 nv = *reinterpret_cast<const f32x4*>(a.norms + row0 + 8 * c + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[4 * c + j] = nv[j];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        }
        for (int kc = 0; kc < ku; ++kc) {
            f32x4 an[8], bhn[8], bln[QP == 2 ? 8 : 1];
            const int kn = kc + 1 < ku ? kc + 1 : kc;   // (the last unit re-requests itself: no branch around the loads)
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                an[g] = *reinterpret_cast<const f32x4*>(arow + (size_t)kn * 256 + g * 32);
                bhn[g] = *reinterpret_cast<const f32x4*>(qrow + (size_t)kn * 512 + g * 32);
                if constexpr (QP == 2) bln[g] = *reinterpret_cast<const f32x4*>(qrow + (size_t)kn * 512 + 256 + g * 32);
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const bf16x8 ah = __builtin_bit_cast(bf16x8, av[s]);
                if constexpr (QP == 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, __builtin_bit_cast(bf16x8, bl[s]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, __builtin_bit_cast(bf16x8, bh[s]), acc, 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                av[g] = an[g];
                bh[g] = bhn[g];
                if constexpr (QP == 2) bl[g] = bln[g];
            }
        }
        // ---- tile complete: scores for 32 rows x 32 queries (lane = query, register i = row (i & 3) + 8 (i >> 2) + 4 h)
        const bool is_tail = row0 + TILE_ROWS > a.n;   // wave-uniform: the last, partial tile
        if constexpr (SAMPLE) {
            float ml = __builtin_inff();
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (!is_tail || ((tail_mask >> i) & 1u)) ml = fminf(ml, acc[i]);
            a.sample_out[(long long)qglob * a.ns + sel * 2 + h] = ml;
        } else {
            float m = acc[0];
#pragma unroll
            for (int i = 1; i < 16; ++i) m = fminf(m, acc[i]);
            const u64 hit = __ballot(m <= thr_l);
            if (hit != 0) {
                u32 mask = le_mask16(acc, thr_l);
                if (is_tail) mask &= tail_mask;
                const u64 bal = __ballot(mask != 0);
                if (mask) {
                    const u32 pos = wcount + __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u));
                    if (pos < a.wave_cap) wout[pos] = make_uint2((u32)(row0 + 4 * h), (mask << 16) | (u32)r31);
                }
                wcount += (u32)__popcll(bal);
            }
        }
    }
    if constexpr (!SAMPLE) {
        if (lane == 0) {
            a.wave_cnt[2 * wave_id] = wcount;        // entries written (beyond wave_cap: overflow)
            a.wave_cnt[2 * wave_id + 1] = (u32)qt;   // this wave's query tile
        }
    }
}

}  // namespace sq
