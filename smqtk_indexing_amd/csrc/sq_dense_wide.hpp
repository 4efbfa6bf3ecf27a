// The first-stage filter for rows wider than 512 dimensions (gfx950).        (included by sq_dense.hip)
//
// The reference's arithmetic takes any d (smqtk_indexing/utils/metrics.py:73-86) and its own examples index 2048- and
// 4096-dimensional descriptors (docs/examples/caffe_build_index.rst:35); up to round 3 such rows had no filter at all: every
// query took the exact path, a pass over the float32 matrix per 8 queries.  dense_scan_kernel keeps the query tile's
// fragments in registers or in LDS, which ends at d_pad = 512 (32 queries x d_pad x 2 planes x 2 bytes: 512 KB at 4096).
// Here a wave owns a 32-row tile and walks its k-units (128 columns = 256 bytes of the bfloat16 scan copy per row) with
// the accumulators of ONE query tile in registers; the row fragments of a k-unit are plain global loads straight into
// the MFMA operand registers (the scan copy is stored in fragment order: a lane's 16 bytes are a lane's operand), the
// next unit's requested before this unit's sixteen v_mfma_f32_32x32x16_bf16; the query tile's fragments of the unit --
// the same for all eight waves of a workgroup -- pass through a double-buffered LDS copy, one barrier per unit; the query
// planes themselves (512 KB for 32 queries at 4096 dimensions) stay in L2.  Same inputs, outputs and error bound as
// dense_scan_kernel's one-tile configuration (DenseScanArgs; scores = n' + x_hi (q_hi + q_lo)): the
// sample pass, the threshold, the exact re-rank, the select and the certification around it are unchanged.
// No DMA ring: the compiler's own counted waits order the loads.  (Non-temporal row loads were tried: 2.63 instead of 1.57-1.68 ms
// per pass -- a wave's load touches 32 bytes of 32 different rows and the other three quarters of each line are wanted by its
// next three loads: without the caches every line is fetched four times.)
#pragma once
#include "sq_dense_scan.hpp"

namespace sq {

static constexpr int WIDE_WAVES = 8;

// LDS: two buffers of the query tiles' fragments of ONE k-unit: [tile][plane][query 0..31][16 chunks of 16 bytes, chunk c at
// position c ^ (query & 15): the fragment reads of 16 consecutive queries then hit 16 different bank groups].
// QT query tiles per wave (1; 2 / 4 for batches beyond 32 / 64 queries: a row tile's fragments meet QT sets of query
// fragments, so a pass over the copy serves 32 QT queries -- a half / a quarter of the passes of a large batch; 16 more
// accumulators per lane and tile, 16 KB more LDS per tile and buffer).
template <int QP, int QT, bool SAMPLE>
__global__ __launch_bounds__(WIDE_WAVES * 64, QT == 1 ? 2 : 1) void dense_wide_scan_kernel(DenseScanArgs a, int ku) {
    constexpr int TILE_BYTES = QP * TILE_ROWS * 256;   // one query tile's fragments of a k-unit
    __shared__ __attribute__((aligned(16))) unsigned char qbuf[2][QT * TILE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r31 = lane & 31, h = lane >> 5;
    const long long wave_id = (long long)blockIdx.x * WIDE_WAVES + wave;   // unique per wave of the launch
    uint2* wout = a.wave_out + wave_id * a.wave_cap;
    // block -> (row block, group of QT query tiles): dense_scan_kernel's mapping (blocks that share an XCD walk the groups of
    // the same rows)
    const int L = blockIdx.x;
    int qt, rb;   // qt: first query tile of this block's group
    if (a.nqt > 1) {
        const int xcd = L & 7, j = L >> 3;
        qt = (j % a.nqt) * QT;
        rb = (j / a.nqt) * 8 + xcd;
    } else {
        qt = 0;
        rb = L;
    }
    const long long gw = (long long)rb * WIDE_WAVES + wave;
    const long long nwaves = (long long)a.nrb * WIDE_WAVES;
    const size_t dpad = (size_t)ku * KT;
    const int qglob0 = qt * TILE_ROWS + r31;   // this lane's query in tile t of the group: qglob0 + 32 t
    float thr_l[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) thr_l[t] = SAMPLE ? 0.f : ((a.debug & 4) ? -__builtin_inff() : a.thr[qglob0 + t * TILE_ROWS]);
    // The query tiles' fragments of a k-unit are the same for the eight waves of the workgroup (and for every row tile):
    // thread t brings chunk (t & 15) of query (t >> 4) of each tile and plane -- 16 threads read 256 contiguous bytes --
    // and all waves read their fragments from the LDS copy.  (Every wave loading its own fragments from L2 -- the first
    // version -- moved 2 x the rows' bytes through the L2 fabric and ran the pass at 0.34 of the HBM peak.)
    const int lq = tid >> 4, lc = tid & 15;
    const unsigned char* qsrc = reinterpret_cast<const unsigned char*>(a.qs) + (size_t)(qt * TILE_ROWS + lq) * dpad * 4 + (size_t)lc * 16;
    const size_t qtile_stride = (size_t)TILE_ROWS * dpad * 4;   // the next query tile's rows in a.qs
    const u32 qdst = (u32)(lq * 256 + ((lc ^ (lq & 15)) * 16));
    u32 tail_mask = 0;   // rows of the last, partial tile that exist (bit i <-> accumulator register i of this lane)
    {
        const int nvalid = (int)(a.n - (a.n_tiles - 1) * TILE_ROWS);
#pragma unroll
        for (int i = 0; i < 16; ++i) tail_mask |= ((i & 3) + 8 * (i >> 2) + 4 * h < nvalid ? 1u : 0u) << i;
    }
    // rounds of the WORKGROUP (its first wave has the most tiles): every wave walks them all -- the barriers are the
    // workgroup's -- and skips the arithmetic of a round it has no tile in
    const long long wg_first = (long long)rb * WIDE_WAVES;
    const long long rounds = wg_first < a.n_sel ? (a.n_sel - wg_first + nwaves - 1) / nwaves : 0;
    if (rounds > 0) {   // unit 0 of the query tiles
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            *reinterpret_cast<f32x4*>(qbuf[0] + t * TILE_BYTES + qdst) = *reinterpret_cast<const f32x4*>(qsrc + t * qtile_stride);
            if constexpr (QP == 2)
                *reinterpret_cast<f32x4*>(qbuf[0] + t * TILE_BYTES + TILE_ROWS * 256 + qdst) = *reinterpret_cast<const f32x4*>(qsrc + t * qtile_stride + 256);
        }
    }
    u32 wcount = 0;
    int step = 0;   // (round, k-unit) steps so far: buffer step & 1 holds this step's query fragments
    for (long long it = 0; it < rounds; ++it) {
        const long long sel = gw + it * nwaves;
        const bool active = sel < a.n_sel;
        const long long row0 = (active ? sel : 0) * a.tile_step * TILE_ROWS;
        const unsigned char* arow = reinterpret_cast<const unsigned char*>(a.scan) + (size_t)(row0 + r31) * dpad * 2 + (size_t)h * 16;
        f32x4 av[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) av[g] = *reinterpret_cast<const f32x4*>(arow + g * 32);
        // the accumulators start from the rows' stored norms n' (0 for cosine): score = n' + x . q'
        f32x16 acc[QT];
        if (a.norms) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                // (norm_step 0: the cosine call of a multi-tile batch hands over 32 zeros for every tile, as to the ring kernels)
                const f32x4 nv = *reinterpret_cast<const f32x4*>(a.norms + row0 * a.norm_step + 8 * c + 4 * h);
#pragma unroll
                for (int t = 0; t < QT; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[t][4 * c + j] = nv[j];
            }
        } else {
#pragma unroll
            for (int t = 0; t < QT; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        }
        for (int kc = 0; kc < ku; ++kc, ++step) {
            // requests of the NEXT step: the row fragments of this tile's next unit (the last unit re-requests itself: no
            // branch around the loads) and the workgroup's share of the next query unit (the next round starts at unit 0)
            f32x4 an[8], nq0[QT], nq1[QT];
            const int kn = kc + 1 < ku ? kc + 1 : kc;
            const int kq = kc + 1 < ku ? kc + 1 : 0;
#pragma unroll
            for (int g = 0; g < 8; ++g) an[g] = *reinterpret_cast<const f32x4*>(arow + (size_t)kn * 256 + g * 32);
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                nq0[t] = *reinterpret_cast<const f32x4*>(qsrc + t * qtile_stride + (size_t)kq * 512);
                if constexpr (QP == 2) nq1[t] = *reinterpret_cast<const f32x4*>(qsrc + t * qtile_stride + (size_t)kq * 512 + 256);
            }
            __syncthreads();   // this step's buffer is complete; nobody still reads the other one
            const unsigned char* qb = qbuf[step & 1];
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                f32x4 bh[8], bl[QP == 2 ? 8 : 1];
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const u32 at = (u32)(t * TILE_BYTES + r31 * 256 + (((2 * g + h) ^ (r31 & 15)) * 16));
                    bh[g] = *reinterpret_cast<const f32x4*>(qb + at);
                    if constexpr (QP == 2) bl[g] = *reinterpret_cast<const f32x4*>(qb + TILE_ROWS * 256 + at);
                }
                if (active) {
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        const bf16x8 ah = __builtin_bit_cast(bf16x8, av[s]);
                        if constexpr (QP == 2) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, __builtin_bit_cast(bf16x8, bl[s]), acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, __builtin_bit_cast(bf16x8, bh[s]), acc[t], 0, 0, 0);
                    }
                }
            }
            unsigned char* qn = qbuf[(step + 1) & 1];
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                *reinterpret_cast<f32x4*>(qn + t * TILE_BYTES + qdst) = nq0[t];
                if constexpr (QP == 2) *reinterpret_cast<f32x4*>(qn + t * TILE_BYTES + TILE_ROWS * 256 + qdst) = nq1[t];
            }
#pragma unroll
            for (int g = 0; g < 8; ++g) av[g] = an[g];
        }
        if (!active) continue;
        // ---- tile complete: scores for 32 rows x QT x 32 queries (lane = query, register i = row (i & 3) + 8 (i >> 2) + 4 h)
        const bool is_tail = row0 + TILE_ROWS > a.n;   // wave-uniform: the last, partial tile
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float m = __builtin_inff();   // (the rows that exist: a padding row's score is no row's)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (!is_tail || ((tail_mask >> i) & 1u)) m = fminf(m, acc[t][i]);
            if constexpr (SAMPLE) {
                a.sample_out[(long long)(qglob0 + t * TILE_ROWS) * a.ns + sel * 2 + h] = m;
            } else {
                const u64 hit = __ballot(m <= thr_l[t]);
                if (hit != 0) {
                    u32 mask = le_mask16(acc[t], thr_l[t]);
                    if (is_tail) mask &= tail_mask;
                    const u64 bal = __ballot(mask != 0);
                    if (mask) {
                        const u32 pos = wcount + __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u));
                        if (pos < a.wave_cap) {
                            wout[pos] = make_uint2((u32)(row0 + 4 * h), (mask << 16) | (u32)(t * TILE_ROWS + r31));
                            if (a.wave_score) a.wave_score[wave_id * a.wave_cap + pos] = m;   // (second-level threshold: sq_dense_tighten.hpp)
                        }
                    }
                    wcount += (u32)__popcll(bal);
                }
            }
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
