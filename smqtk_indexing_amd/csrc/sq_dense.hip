// Exact brute-force L2 / cosine top-k over a float32 descriptor matrix (gfx950).
//
// What it replaces: the per-candidate python distance calls + stable sort of
// LSHNearestNeighborIndex._nn (smqtk_indexing/impls/nn_index/lsh.py:505-519)
// and the faiss "IDMap,Flat" search + recompute of
// FaissNearestNeighborsIndex._nn (impls/nn_index/faiss.py:751-831), both of
// which evaluate metrics.euclidean_distance / cosine_distance
// (utils/metrics.py:73-86, 89-137) for every row and keep the n smallest.
//
// Structure (DESIGN.md "dense path"):
//   1. dense_scan_kernel      streams the matrix once per 32-query tile.  Each
//      wave owns 32-row x 64-float units that it pulls HBM -> LDS with
//      global_load_lds_dwordx4 into a private ring (no workgroup barrier in
//      the loop, counted s_waitcnt vmcnt), feeds v_mfma_f32_32x32x2_f32 with
//      A = rows (ds_read_b128, XOR-swizzled, conflict free) and B = the
//      pre-scaled query tile held in LDS, and compares the 32x32 scores
//      s = |x|^2 - 2 x.q (cosine: -x^.q^) with a per-query threshold.
//      Survivors (row ids) go to per-query candidate lists.
//      mode SAMPLE writes the raw scores of every S-th tile instead; the
//      threshold is the ~k-th smallest sample score (kth_threshold_f32_kernel).
//   2. dense_exact_*_kernel   recomputes the distance of every candidate in
//      the REFERENCE arithmetic (float32 subtract, square, numpy pairwise
//      summation order, correctly rounded sqrt; cosine in float64) and forms
//      (distance, row) keys.
//   3. select_topk_kernel     sorts the keys; dense_finalize_kernel converts
//      and CERTIFIES each query: the k-th exact distance must lie below the
//      smallest distance any non-candidate can have given the threshold and
//      the float32 error bound of the MFMA score.  Queries that fail (or
//      overflow their list) are redone on the exact full-keys path.
#include "sq_pairwise.cuh"
#include "sq_select.cuh"

namespace sq {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static constexpr int KT = 64;                 // floats per k-unit (one 256-byte LDS bank row per matrix row)
static constexpr int TILE_ROWS = 32;          // rows per MFMA tile / per wave unit
static constexpr int UNIT_BYTES = TILE_ROWS * KT * 4;  // 8 KiB
static constexpr int EBUF_BYTES = 8192;       // emission buffers of a workgroup: (row, query) pairs, split over its waves
static constexpr int MAX_DPAD = 512;

struct DenseHandle : HandleBase {
    const float* db = nullptr;  // device [n][ld]
    DevBuf owned;
    DevBuf normalized;          // cosine: rows scaled to unit length (filter operand)
    long long n = 0;
    int d = 0;
    int d_pad = 0;
    long long ld = 0;
    int metric = SQ_METRIC_L2;
    long long id_base = 0;
    double xn2_max = 0.0;       // max squared row norm (error bound of the L2 filter)
    // workspace
    DevBuf q_dev, q_scaled, qn2, thr, cand, cnt, keys, sample, out_keys, status, out_dist_dev, out_idx_dev, big_keys,
        scratch;
    HostPinned status_host;
    ~DenseHandle() override {
        for (DevBuf* b : {&owned, &normalized, &q_dev, &q_scaled, &qn2, &thr, &cand, &cnt, &keys, &sample, &out_keys,
                          &status, &out_dist_dev, &out_idx_dev, &big_keys, &scratch})
            b->release();
        status_host.release();
    }
};

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
    sim = fmax(fmin(sim, 1.0), -1.0);
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

// ------------------------------------------------------------- small kernels
// Per-row squared norms -> max (as ordered float bits) and, for cosine, the
// unit-length copy of the matrix used by the filter.
__global__ __launch_bounds__(256) void dense_rowstats_kernel(const float* __restrict__ db, long long n, long long ld,
                                                              int d, int d_pad, u32* __restrict__ max_bits,
                                                              float* __restrict__ normalized) {
    const int lane8 = threadIdx.x & 7;
    const long long row = (long long)blockIdx.x * 32 + (threadIdx.x >> 3);
    const long long r = row < n ? row : n - 1;
    const float* x = db + r * ld;
    double acc = 0.0;
    for (int i = lane8; i < d; i += 8) acc += (double)x[i] * (double)x[i];
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    acc += __shfl_xor(acc, 4);
    if (row < n) {
        if (lane8 == 0) atomicMax(max_bits, __float_as_uint((float)(acc * (1.0 + 1e-6))));
        if (normalized) {
            const double inv = acc > 0.0 ? 1.0 / sqrt(acc) : 0.0;
            float* o = normalized + row * (long long)d_pad;
            for (int i = lane8; i < d_pad; i += 8) o[i] = i < d ? (float)((double)x[i] * inv) : 0.f;
        }
    }
}

// Pad/copy host-layout rows [n][d] into [n][d_pad] (zero padded).
__global__ void dense_pad_rows_kernel(const float* __restrict__ src, long long n, int d, int d_pad,
                                      float* __restrict__ dst) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * (long long)d_pad) return;
    const long long r = i / d_pad;
    const int c = (int)(i - r * d_pad);
    dst[i] = c < d ? src[r * d + c] : 0.f;
}

// Query prep: scaled/padded filter operand [nq_pad][d_pad] and |q|^2 (f64).
//   L2: -2 q        cosine: -q / |q|
__global__ void dense_prep_queries_kernel(const float* __restrict__ q, int nq, int d, int d_pad, int nq_pad,
                                          int metric, float* __restrict__ qs, double* __restrict__ qn2) {
    const int qi = blockIdx.x;
    __shared__ double red[4];
    double acc = 0.0;
    if (qi < nq)
        for (int i = threadIdx.x; i < d; i += blockDim.x) acc += (double)q[(long long)qi * d + i] * (double)q[(long long)qi * d + i];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    double tot = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += red[w];
    if (threadIdx.x == 0) qn2[qi] = qi < nq ? tot : 0.0;
    double scale = -2.0;
    if (metric == SQ_METRIC_COSINE) scale = tot > 0.0 ? -1.0 / sqrt(tot) : 0.0;
    for (int i = threadIdx.x; i < d_pad; i += blockDim.x) {
        float v = 0.f;
        if (qi < nq && i < d) v = (float)((double)q[(long long)qi * d + i] * scale);
        qs[(long long)qi * d_pad + i] = v;
    }
}

// ------------------------------------------------------------- the scan kernel
struct DenseScanArgs {
    const float* db;        // filter operand (the matrix; cosine: its normalized copy)
    long long n;
    long long ld;           // row stride in floats (>= d_pad readable columns)
    int d_pad;              // multiple of KT
    const float* qs;        // [nqt*32][d_pad] pre-scaled queries
    const float* thr;       // [nqt*32] score thresholds (mode EMIT)
    u32* cand;              // [nqt*32][cap] candidate row ids
    u32* cnt;               // [nqt*32]
    u32 cap;
    float* sample_out;      // [nqt*32][ns] (mode SAMPLE)
    long long ns;
    long long tile_first, tile_step, n_tiles;  // tile i covers rows (tile_first + i*tile_step)*32 ...
    int nqt;                // query tiles
    int nrb;                // row blocks (multiple of 8 when nqt > 1)
    int mode;               // 0 EMIT, 1 SAMPLE
    int add_norm;           // 1: L2 (add |x|^2 through one extra MFMA), 0: cosine
    int waves;              // waves per workgroup of the launch (set by scan_launch)
    int debug;              // measurement only: 1 = skip LDS reads + MFMA, 2 = skip the LDS-DMA (results invalid)
};

// LDS-DMA: 64 lanes x 16 bytes land at lds_dst + lane*16 (wave-uniform base in
// M0); the global source is a wave-uniform 64-bit base (SGPR pair) plus a
// per-lane 32-bit byte offset.  Issued from inline asm so that hipcc does not
// fence every later ds_read with vmcnt(0); completion is tracked by the
// counted waits below (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ void glds16(const float* gbase_uniform, u32 voff, u32 lds_dst) {
    u32 keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(gbase_uniform), "s"(lds_dst)
        : "memory");
}
// same with a full per-lane 64-bit address (partial last tile: clamped rows)
__device__ __forceinline__ void glds16_addr(const float* gsrc, u32 lds_dst) {
    u32 keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int NSTAGE>
__device__ __forceinline__ void wait_units_in_flight(int units) {
    // allow `units` younger 8-instruction units to stay outstanding
    if constexpr (NSTAGE >= 4) {
        if (units >= 3) {
            wait_vmcnt<24>();
            return;
        }
    }
    if constexpr (NSTAGE >= 3) {
        if (units == 2) {
            wait_vmcnt<16>();
            return;
        }
    }
    if (units == 1)
        wait_vmcnt<8>();
    else
        wait_vmcnt<0>();
}

typedef __attribute__((address_space(3))) u32 lds_u32;

// WAVES: waves per workgroup (one workgroup per CU: 4 = one wave per SIMD, 8 = two, which lets one
// wave's DMA issue / epilogue run under the other's MFMAs); NSTAGE: ring depth per wave;
// KU = d_pad / 64 k-units per row tile.
template <int WAVES, int NSTAGE, int KU>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void dense_scan_kernel(DenseScanArgs a) {
    constexpr bool QREG = KU <= 2;  // query fragments live in registers for d_pad <= 128
    constexpr int SCAN_WAVES = WAVES;
    constexpr int EBUF_ENTRIES = EBUF_BYTES / 8 / WAVES;
    constexpr int EBUF_FLUSH = EBUF_ENTRIES / 2;
    constexpr int DPAD = KU * KT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the query tile keeps its own LDS region only when it is re-read per unit; when it lives in
    // registers it is staged through the (not yet used) ring area
    constexpr u32 q_bytes = QREG ? 0u : (u32)TILE_ROWS * DPAD * 4;
    // LDS map: [query tile][ring wave0..3][emission buffers][emission counters]
    const u32 lds_base = (u32)(uintptr_t)smem;  // low 32 bits of a flat LDS address = LDS offset
    const u32 ring_base = lds_base + q_bytes + (u32)wave * (NSTAGE * UNIT_BYTES);
    unsigned char* ring_ptr = smem + q_bytes + wave * (NSTAGE * UNIT_BYTES);
    unsigned char* etop = smem + q_bytes + SCAN_WAVES * NSTAGE * UNIT_BYTES;
    uint2* ebuf = reinterpret_cast<uint2*>(etop) + wave * EBUF_ENTRIES;
    lds_u32* ecnt_ptr = (lds_u32*)(etop + SCAN_WAVES * EBUF_ENTRIES * 8 + wave * 16);

    // block -> (row block, query tile); blocks that share an XCD (same id mod 8)
    // walk the query tiles of the same rows so the matrix is re-read from L2.
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

    // stage the query tile: [32][d_pad] floats, 16-byte chunks XOR-swizzled
    // inside each 256-byte group by (row & 15)
    {
        const float* qsrc = a.qs + (long long)qt * TILE_ROWS * DPAD;
        constexpr int chunks_per_row = DPAD / 4;
        for (int c = threadIdx.x; c < TILE_ROWS * chunks_per_row; c += SCAN_WAVES * 64) {
            const int r = c / chunks_per_row, ch = c - r * chunks_per_row;
            f32x4 v = *reinterpret_cast<const f32x4*>(qsrc + (long long)r * DPAD + ch * 4);
            const int sw = (ch & ~15) | ((ch & 15) ^ (r & 15));
            *reinterpret_cast<f32x4*>(smem + (u32)r * DPAD * 4 + sw * 16) = v;
        }
        if (lane == 0) *ecnt_ptr = 0u;
    }
    __syncthreads();

    // Tiles are dealt round-robin over all waves of the launch (wave gw takes tiles
    // gw, gw + nwaves, ...): at any moment the grid reads one compact window of the
    // matrix, which keeps HBM pages open across waves (a private contiguous range
    // per wave measured ~20 % slower).
    const long long gw = (long long)rb * SCAN_WAVES + wave;
    const long long nwaves = (long long)a.nrb * SCAN_WAVES;
    const long long my_tiles = gw < a.n_tiles ? (a.n_tiles - gw + nwaves - 1) / nwaves : 0;
    const long long total_units = my_tiles * KU;

    const int r31 = lane & 31, h = lane >> 5;
    const int qglob = qt * TILE_ROWS + r31;
    float thr_l = a.mode == 0 ? a.thr[qglob] : 0.f;
    // Force hipcc's wait for this load HERE.  Left to its first use inside the
    // loop the compiler emits s_waitcnt vmcnt(0) there (it cannot see the asm
    // LDS-DMAs), draining the whole ring once per tile.
    asm volatile("" : "+v"(thr_l));

    f32x4 bq[QREG ? KU : 1][8];
    if constexpr (QREG) {
#pragma unroll
        for (int kc = 0; kc < KU; ++kc)
#pragma unroll
            for (int g = 0; g < 8; ++g)
                bq[kc][g] = *reinterpret_cast<const f32x4*>(smem + (u32)r31 * DPAD * 4 + kc * 256 +
                                                             ((2 * g + h) ^ (r31 & 15)) * 16);
        __syncthreads();  // every wave holds its fragments before the DMA ring overwrites the staging area
    }

    // per-lane byte offsets of the 8 DMA instructions of a unit (row 4j + lane/16, swizzled 16-byte chunk)
    u32 voff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = 4 * j + (lane >> 4);
        voff[j] = (u32)(((long long)r * a.ld + (((lane & 15) ^ (r & 15)) * 4)) * 4);
    }

    auto emit_global = [&](u32 row, u32 q) {
        u32 pos = atomicAdd(&a.cnt[q], 1u);
        if (pos < a.cap) a.cand[(long long)q * a.cap + pos] = row;
    };
    auto flush = [&](u32 c) {
        const u32 n = c < (u32)EBUF_ENTRIES ? c : (u32)EBUF_ENTRIES;
        for (u32 e = lane; e < n; e += 64) {
            uint2 ent = ebuf[e];
            emit_global(ent.x, ent.y);
        }
        if (lane == 0) *ecnt_ptr = 0u;
    };

    // issue cursor
    long long iss_tile = gw;
    int iss_kc = 0, iss_slot = 0;
    long long issued = 0;
    auto issue_unit = [&]() {
        const long long row0 = (a.tile_first + iss_tile * a.tile_step) * TILE_ROWS;
        const u32 dst = ring_base + (u32)iss_slot * UNIT_BYTES;
        if (row0 + TILE_ROWS <= a.n) {
            const float* base = a.db + row0 * a.ld + (long long)iss_kc * KT;  // wave-uniform
#pragma unroll
            for (int j = 0; j < 8; ++j) glds16(base, voff[j], dst + (u32)j * 1024);
        } else {
            const float* colbase = a.db + (long long)iss_kc * KT;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = 4 * j + (lane >> 4);
                long long row = row0 + r;
                row = row < a.n ? row : a.n - 1;
                glds16_addr(colbase + row * a.ld + ((lane & 15) ^ (r & 15)) * 4, dst + (u32)j * 1024);
            }
        }
        ++issued;
        if (++iss_kc == KU) {
            iss_kc = 0;
            iss_tile += nwaves;
        }
        if (++iss_slot == NSTAGE) iss_slot = 0;
    };

    // Software pipeline over units (u = 0, 1, ...; slot of unit u = u % NSTAGE):
    //   registers hold the A fragments of unit u (av_cur) while its 32 MFMAs run;
    //   the fragments of unit u+1 are read from LDS (av_nxt) under those MFMAs;
    //   the slot of unit u is refilled by the DMA of unit u+NSTAGE as soon as
    //   av_cur is complete.  So NSTAGE-1 units stay in flight behind the one
    //   being waited for, and no LDS latency sits between MFMA groups.
    const bool do_dma = !(a.debug & 2), do_math = !(a.debug & 1);
    auto issue_next = [&]() {
        if (issued < total_units) {
            if (do_dma || issued < NSTAGE)  // ablation: the ring is filled once, then reused
                issue_unit();
            else
                ++issued;
        }
    };
    auto read_frags = [&](int slot_idx, f32x4 (&dst)[8]) {
        const unsigned char* arow = ring_ptr + slot_idx * UNIT_BYTES + r31 * 256;
#pragma unroll
        for (int g = 0; g < 8; ++g) dst[g] = *reinterpret_cast<const f32x4*>(arow + ((2 * g + h) ^ (r31 & 15)) * 16);
    };
    for (int p = 0; p < NSTAGE; ++p) issue_next();

    f32x4 av_cur[8], av_nxt[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) av_nxt[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    long long loaded = 0;  // units whose fragments have been requested from LDS
    int rd_slot = 0;
    if (total_units > 0) {
        wait_units_in_flight<NSTAGE>((int)(issued - 1));
        if (do_math) read_frags(0, av_nxt);
        loaded = 1;
        rd_slot = NSTAGE > 1 ? 1 : 0;
    }
    for (long long tile = gw; tile < a.n_tiles; tile += nwaves) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        float nrm = 0.f;
#pragma unroll
        for (int kc = 0; kc < KU; ++kc) {
            // fragments of this unit are complete once copied (hipcc waits lgkmcnt here); its slot is free
#pragma unroll
            for (int g = 0; g < 8; ++g) av_cur[g] = av_nxt[g];
            asm volatile("" ::: "memory");
            issue_next();
            if (loaded < total_units) {
                wait_units_in_flight<NSTAGE>((int)(issued - loaded - 1));  // younger units may stay in flight
                if (do_math) read_frags(rd_slot, av_nxt);
                ++loaded;
                if (++rd_slot == NSTAGE) rd_slot = 0;
            }
            if (do_math) {
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    f32x4 bvg;
                    if constexpr (QREG)
                        bvg = bq[kc][g];
                    else
                        bvg = *reinterpret_cast<const f32x4*>(smem + (u32)r31 * DPAD * 4 + kc * 256 +
                                                              ((2 * g + h) ^ (r31 & 15)) * 16);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av_cur[g][j], bvg[j], acc, 0, 0, 0);
                        nrm = __builtin_fmaf(av_cur[g][j], av_cur[g][j], nrm);
                    }
                }
            }
        }
        if (!do_math) continue;
        // ---- tile complete: scores for 32 rows x 32 queries (lane = query, regs = rows)
        if (a.add_norm) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(nrm, 1.0f, acc, 0, 0, 0);
        const long long row0 = (a.tile_first + tile * a.tile_step) * TILE_ROWS;
        if (a.mode == 0) {
            float m = acc[0];
#pragma unroll
            for (int i = 1; i < 16; ++i) m = fminf(m, acc[i]);
            if (__any(m <= thr_l)) {
                if (m <= thr_l) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const long long row = row0 + (i & 3) + 8 * (i >> 2) + 4 * h;
                        if (acc[i] <= thr_l && row < a.n) {
                            const u32 pos = __hip_atomic_fetch_add(ecnt_ptr, 1u, __ATOMIC_RELAXED,
                                                                   __HIP_MEMORY_SCOPE_WORKGROUP);
                            if (pos < (u32)EBUF_ENTRIES)
                                ebuf[pos] = make_uint2((u32)row, (u32)qglob);
                            else
                                emit_global((u32)row, (u32)qglob);  // buffer full (degenerate thresholds only)
                        }
                    }
                }
                const u32 c = __builtin_amdgcn_readfirstlane(*ecnt_ptr);
                if (c >= (u32)EBUF_FLUSH) flush(c);
            }
        } else {
            // sample mode: the minimum score of this lane's 16 rows (one row's score: a valid
            // upper bound sample for the k-th smallest, see kth_threshold_f32_kernel)
            float ml = __builtin_inff();
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ro = (i & 3) + 8 * (i >> 2) + 4 * h;
                if (row0 + ro < a.n) ml = fminf(ml, acc[i]);
            }
            a.sample_out[(long long)qglob * a.ns + tile * 2 + h] = ml;
        }
    }
    if (a.mode == 0) {
        const u32 c = __builtin_amdgcn_readfirstlane(*ecnt_ptr);
        if (c > 0) flush(c);
    }
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
    // numpy pairwise recursion (split at n/2 rounded down to a multiple of 8), explicit stack
    __device__ float sum(int d) const {
        if (d <= 128) return leaf(0, d);
        int s_off[24], s_n[24], s_state[24];
        float s_left[24];
        int sp = 1;
        s_off[0] = 0;
        s_n[0] = d;
        s_state[0] = 0;
        float ret = 0.f;
        while (sp > 0) {
            const int top = sp - 1;
            const int off = s_off[top], m = s_n[top];
            if (m <= 128) {
                ret = leaf(off, m);
                --sp;
                continue;
            }
            int m2 = m / 2;
            m2 -= m2 % 8;
            if (s_state[top] == 0) {
                s_state[top] = 1;
                s_off[sp] = off;
                s_n[sp] = m2;
                s_state[sp] = 0;
                ++sp;
            } else if (s_state[top] == 1) {
                s_left[top] = ret;
                s_state[top] = 2;
                s_off[sp] = off + m2;
                s_n[sp] = m - m2;
                s_state[sp] = 0;
                ++sp;
            } else {
                ret = __fadd_rn(s_left[top], ret);
                --sp;
            }
        }
        return ret;
    }
};

// Candidate j of query q (row = cand[q][j], or row_offset + j when cand == nullptr)
// -> key (ordered float32 euclidean distance, row).  Dynamic LDS: round_up(d,4)*4 bytes.
__global__ __launch_bounds__(256) void dense_exact_l2_kernel(const float* __restrict__ db, long long ld, int d,
                                                              const float* __restrict__ q_orig,
                                                              const u32* __restrict__ cand, const u32* __restrict__ cnt,
                                                              u32 cap, long long implicit_n, long long row_offset,
                                                              u64* __restrict__ keys, long long key_stride) {
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
    }
}

__global__ __launch_bounds__(256) void dense_exact_cos_kernel(const float* __restrict__ db, long long ld, int d,
                                                               const float* __restrict__ q_orig,
                                                               const u32* __restrict__ cand, const u32* __restrict__ cnt,
                                                               u32 cap, long long implicit_n, long long row_offset,
                                                               K128* __restrict__ keys, long long key_stride) {
    const int q = blockIdx.y;
    const long long M = cand ? (long long)(cnt[q] < cap ? cnt[q] : cap) : implicit_n;
    const float* qv = q_orig + (long long)q * d;
    for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < M; j += (long long)gridDim.x * 256) {
        const long long row = cand ? (long long)cand[(long long)q * cap + j] : row_offset + j;
        const double dist = cosine_row_f64(db + row * ld, qv, d);
        keys[(long long)q * key_stride + j] = K128{ordered_f64(dist), (u64)(u32)row};
    }
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

// The sampled threshold T is the score of an actual row.  The scan emits with
// T + slack so that a query whose k-th neighbour IS that row still certifies:
// slack covers twice the filter's error bound plus the relative rounding of
// the exact distance (DESIGN.md "certification").
__global__ void dense_inflate_thr_kernel(float* __restrict__ thr, const double* __restrict__ qn2, int nq, int cosine,
                                         double xn2_max, double eps_coef) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const float t = thr[q];
    if (!(t < __builtin_inff())) return;
    double slack;
    if (cosine) {
        slack = 2.0 * eps_coef + 1e-8;
    } else {
        const double eps = eps_coef * (xn2_max + 2.0 * sqrt(xn2_max * qn2[q]));
        slack = 2.0 * eps + 4e-6 * fabs((double)t + qn2[q]);
    }
    // round up so the float threshold is never below T + slack
    float r = (float)((double)t + slack);
    if ((double)r < (double)t + slack) r = __uint_as_float(__float_as_uint(r) + (r >= 0.f ? 1 : -1));
    thr[q] = r;
}

// --------------------------------------------------------------- finalize
// status bits: 1 candidate overflow, 2 certification failed, 4 fewer than kk candidates
__global__ void dense_finalize_l2_kernel(const u64* __restrict__ sorted, const u32* __restrict__ cnt, u32 cap, int k,
                                         int kk, long long id_base, const float* __restrict__ thr,
                                         const double* __restrict__ qn2, double xn2_max, double eps_coef,
                                         int certify, float* __restrict__ out_dist, long long* __restrict__ out_idx,
                                         u32* __restrict__ status) {
    const int q = blockIdx.x;
    for (int j = threadIdx.x; j < k; j += blockDim.x) {
        const u64 key = sorted[(long long)q * k + j];
        const bool pad = key == ~0ull;
        out_dist[(long long)q * k + j] = pad ? __builtin_inff() : unordered_f32((u32)(key >> 32));
        out_idx[(long long)q * k + j] = pad ? -1ll : id_base + (long long)(key & 0xffffffffull);
    }
    if (threadIdx.x == 0) {
        u32 st = 0;
        if (certify) {
            const u32 c = cnt[q];
            if (c > cap) st |= 1u;
            if (c < (u32)kk) st |= 4u;
            if (st == 0) {
                const u64 key = sorted[(long long)q * k + (kk - 1)];
                const double dk = (double)unordered_f32((u32)(key >> 32));
                const double t = (double)thr[q];
                const double eps = eps_coef * (xn2_max + 2.0 * sqrt(xn2_max * qn2[q]));
                const double lo2 = t + qn2[q] - eps;  // smallest squared distance a non-candidate can have
                const double bound = lo2 > 0.0 ? sqrt(lo2) * (1.0 - 1e-6) : 0.0;
                if (!(t == (double)__builtin_inff()) && !(dk < bound)) st |= 2u;
            }
        }
        status[q] = st;
    }
}

__global__ void dense_finalize_cos_kernel(const K128* __restrict__ sorted, const u32* __restrict__ cnt, u32 cap,
                                          int k, int kk, long long id_base, const float* __restrict__ thr,
                                          double eps, int certify, double* __restrict__ out_dist,
                                          long long* __restrict__ out_idx, u32* __restrict__ status) {
    const int q = blockIdx.x;
    for (int j = threadIdx.x; j < k; j += blockDim.x) {
        const K128 key = sorted[(long long)q * k + j];
        const bool pad = key.hi == ~0ull && key.lo == ~0ull;
        out_dist[(long long)q * k + j] = pad ? (double)__builtin_inff() : unordered_f64(key.hi);
        out_idx[(long long)q * k + j] = pad ? -1ll : id_base + (long long)(key.lo & 0xffffffffull);
    }
    if (threadIdx.x == 0) {
        u32 st = 0;
        if (certify) {
            const u32 c = cnt[q];
            if (c > cap) st |= 1u;
            if (c < (u32)kk) st |= 4u;
            if (st == 0) {
                const double dk = unordered_f64(sorted[(long long)q * k + (kk - 1)].hi);
                const double t = (double)thr[q];  // threshold on -sim~
                // non-candidates: -sim~ > t  =>  sim < -t + eps  =>  dist > 2 acos(min(1,-t+eps))/pi
                double smax = -t + eps;
                smax = smax > 1.0 ? 1.0 : (smax < -1.0 ? -1.0 : smax);
                const double bound = 2.0 * acos(smax) / 3.141592653589793 - 1e-9;
                if (!(t == (double)__builtin_inff()) && !(dk < bound)) st |= 2u;
            }
        }
        status[q] = st;
    }
}

// -------------------------------------------------------------- host driver
static constexpr int kSelectLdsKeys64 = 16384;
static constexpr int kSelectLdsKeys128 = 7168;

template <class K>
static int select_launch_t(const K* keys, const u32* cnt, u32 cap, long long stride, int k, int nq, K* out,
                           hipStream_t st) {
    static bool attr_set = false;
    const int lds_keys = sizeof(K) == 8 ? kSelectLdsKeys64 : kSelectLdsKeys128;
    const size_t lds = (size_t)(lds_keys + SELECT_SORT_MAX) * sizeof(K);
    if (!attr_set) {
        SQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&select_topk_kernel<K>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((select_topk_kernel<K>), dim3(nq), dim3(1024), lds, st, keys, cnt, cap, stride, k, lds_keys,
                       out);
    return SQ_OK;
}

template <int WAVES, int NSTAGE, int KU>
static int scan_launch_t(const DenseScanArgs& a, size_t lds, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        SQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_scan_kernel<WAVES, NSTAGE, KU>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL((dense_scan_kernel<WAVES, NSTAGE, KU>), dim3((unsigned)(a.nrb * a.nqt)), dim3(WAVES * 64),
                       lds, st, a);
    return SQ_OK;
}

static constexpr int SCAN_LDS_TAIL = EBUF_BYTES + 8 * 16;  // emission buffers + per-wave counters

// Launch geometry of the scan for a padded dimension: waves per workgroup and ring depth.
struct ScanGeom {
    int waves, stages;
    size_t lds;
};
static ScanGeom scan_geometry(int d_pad) {
    const int ku = d_pad / KT;
    const bool qreg = ku <= 2;
    const int qb = qreg ? 0 : TILE_ROWS * d_pad * 4;
    ScanGeom g{};
    g.waves = qreg ? 8 : 4;
    if (g_opt.dense_waves == 4 || g_opt.dense_waves == 8) g.waves = g_opt.dense_waves;
    if (!qreg) g.waves = 4;
    int ns = (160 * 1024 - qb - SCAN_LDS_TAIL) / (g.waves * UNIT_BYTES);
    const int ns_max = g.waves == 8 ? 2 : 4;
    if (ns > ns_max) ns = ns_max;
    if (g_opt.dense_stages >= 2 && g_opt.dense_stages <= ns) ns = g_opt.dense_stages;
    g.stages = ns;
    g.lds = (size_t)qb + (size_t)g.waves * ns * UNIT_BYTES + SCAN_LDS_TAIL;
    // staging the query tile through the ring needs the ring to be at least as large
    if (qreg && (size_t)TILE_ROWS * d_pad * 4 > (size_t)g.waves * ns * UNIT_BYTES) g.stages = 0;
    return g;
}

template <int KU>
static int scan_launch_ku(const DenseScanArgs& a, const ScanGeom& g, hipStream_t st) {
    if constexpr (KU <= 2) {
        if (g.waves == 8) return scan_launch_t<8, 2, KU>(a, g.lds, st);
    }
    switch (g.stages) {
        case 4: return scan_launch_t<4, 4, KU>(a, g.lds, st);
        case 3: return scan_launch_t<4, 3, KU>(a, g.lds, st);
        default: return scan_launch_t<4, 2, KU>(a, g.lds, st);
    }
}

static int scan_launch(DenseScanArgs a, hipStream_t st) {
    const ScanGeom g = scan_geometry(a.d_pad);
    if (g.stages < 2) return fail(SQ_ERR_UNSUPPORTED, "dense scan: d_pad=%d leaves no room for the LDS ring", a.d_pad);
    a.waves = g.waves;
    switch (a.d_pad / KT) {
        case 1: return scan_launch_ku<1>(a, g, st);
        case 2: return scan_launch_ku<2>(a, g, st);
        case 3: return scan_launch_ku<3>(a, g, st);
        case 4: return scan_launch_ku<4>(a, g, st);
        case 5: return scan_launch_ku<5>(a, g, st);
        case 6: return scan_launch_ku<6>(a, g, st);
        case 7: return scan_launch_ku<7>(a, g, st);
        case 8: return scan_launch_ku<8>(a, g, st);
        default: return fail(SQ_ERR_UNSUPPORTED, "dense scan: d_pad=%d", a.d_pad);
    }
}

static int dense_search_device(DenseHandle* h, const float* q, int nq, int k, void* out_dist, long long* out_idx,
                               hipStream_t st) {
    const long long n = h->n;
    const int d = h->d, d_pad = h->d_pad;
    const int kk = (int)(k < n ? k : n);
    const bool cosine = h->metric == SQ_METRIC_COSINE;
    const bool prof = g_opt.profile != 0;
    const size_t key_bytes = cosine ? sizeof(K128) : sizeof(u64);
    u32 cap = g_opt.candidate_cap > 0 ? (u32)g_opt.candidate_cap : 65536u;
    if (cap < (u32)(4 * kk)) cap = (u32)(4 * kk);
    const bool force_fb = g_opt.force_fallback != 0;
    const bool small = n <= (long long)cap;
    const bool scan_ok = d_pad <= MAX_DPAD && !small;
    const int nqt = (nq + TILE_ROWS - 1) / TILE_ROWS;
    const int nq_pad = nqt * TILE_ROWS;
    h->stats = sq_stats_t{};
    if (prof) {
        for (auto& e : h->ev)
            if (!e) SQ_HIP(hipEventCreate(&e));
        SQ_HIP(hipEventRecord(h->ev[0], st));
        SQ_HIP(hipEventRecord(h->ev[1], st));
        SQ_HIP(hipEventRecord(h->ev[2], st));
    }
    SQ_TRY(h->cnt.reserve((size_t)nq_pad * 4));
    SQ_TRY(h->thr.reserve((size_t)nq_pad * 4));
    SQ_TRY(h->qn2.reserve((size_t)nq_pad * 8));
    SQ_TRY(h->q_scaled.reserve((size_t)nq_pad * d_pad * 4));
    SQ_TRY(h->out_keys.reserve((size_t)nq * k * key_bytes));
    SQ_TRY(h->status.reserve((size_t)nq * 4));
    SQ_TRY(h->status_host.reserve((size_t)nq * 8));
    u32* cnt = h->cnt.as<u32>();
    float* thr = h->thr.as<float>();
    double* qn2 = h->qn2.as<double>();
    float* qs = h->q_scaled.as<float>();
    u32* status = h->status.as<u32>();
    u32* hs = reinterpret_cast<u32*>(h->status_host.p);
    const double eps_coef = 4.0 * (double)(d_pad + 8) * 5.9604644775390625e-08;  // 4 (d+8) 2^-24
    const size_t l2_lds = (size_t)((d + 3) / 4 * 4) * 4;

    hipLaunchKernelGGL(dense_prep_queries_kernel, dim3(nq_pad), dim3(256), 0, st, q, nq, d, d_pad, nq_pad, h->metric,
                       qs, qn2);

    const long long key_stride = small ? n : (long long)cap;
    bool all_fallback = false;
    if (small) {
        // every row is a candidate: exact keys for all rows, no scan
        SQ_TRY(h->keys.reserve((size_t)nq * key_stride * key_bytes));
        hipLaunchKernelGGL(fill_u32_kernel, dim3((nq_pad + 255) / 256), dim3(256), 0, st, cnt, (long long)nq_pad, (u32)n);
        unsigned gx = (unsigned)((n + 31) / 32);
        if (gx > 4096) gx = 4096;
        if (prof) SQ_HIP(hipEventRecord(h->ev[1], st));
        if (cosine)
            hipLaunchKernelGGL(dense_exact_cos_kernel, dim3(gx, nq), dim3(256), 0, st, h->db, h->ld, d, q, nullptr, cnt,
                               (u32)n, n, 0ll, h->keys.as<K128>(), key_stride);
        else
            hipLaunchKernelGGL(dense_exact_l2_kernel, dim3(gx, nq), dim3(256), l2_lds, st, h->db, h->ld, d, q, nullptr, cnt,
                               (u32)n, n, 0ll, h->keys.as<u64>(), key_stride);
        if (prof) SQ_HIP(hipEventRecord(h->ev[2], st));
        h->stats.scan_launches = 1;
        h->stats.bytes_scanned = n * (long long)d * 4;
        if (cosine) {
            SQ_TRY(select_launch_t<K128>(h->keys.as<K128>(), cnt, (u32)n, key_stride, k, nq, h->out_keys.as<K128>(), st));
            hipLaunchKernelGGL(dense_finalize_cos_kernel, dim3(nq), dim3(256), 0, st, h->out_keys.as<K128>(), cnt,
                               (u32)n, k, kk, h->id_base, thr, 0.0, 0, (double*)out_dist, out_idx, status);
        } else {
            SQ_TRY(select_launch_t<u64>(h->keys.as<u64>(), cnt, (u32)n, key_stride, k, nq, h->out_keys.as<u64>(), st));
            hipLaunchKernelGGL(dense_finalize_l2_kernel, dim3(nq), dim3(256), 0, st, h->out_keys.as<u64>(), cnt, (u32)n,
                               k, kk, h->id_base, thr, qn2, 0.0, 0.0, 0, (float*)out_dist, out_idx, status);
        }
    } else if (scan_ok) {
        const float* fdb = cosine ? h->normalized.as<float>() : h->db;
        const long long fld = cosine ? (long long)d_pad : h->ld;
        const long long n_tiles = (n + TILE_ROWS - 1) / TILE_ROWS;
        long long stride = g_opt.sample_stride > 0 ? g_opt.sample_stride : (long long)cap / (8ll * kk);
        if (stride > 64) stride = 64;
        if (stride < 1) stride = 1;
        while (stride > 1 && (n_tiles / stride) * 2 < 8ll * kk) stride >>= 1;
        const long long ns_tiles = (n_tiles + stride - 1) / stride;
        const long long ns = ns_tiles * 2;  // one sample (a 16-row group minimum) per lane half per tile
        SQ_TRY(h->sample.reserve((size_t)nq_pad * ns * 4));
        SQ_TRY(h->cand.reserve((size_t)nq_pad * cap * 4));
        SQ_TRY(h->keys.reserve((size_t)nq * key_stride * key_bytes));
        u32* cand = h->cand.as<u32>();
        const int cus = cu_count(h->device);
        int nrb = g_opt.dense_blocks > 0 ? g_opt.dense_blocks : cus;
        nrb = (nrb + 7) / 8 * 8;
        DenseScanArgs a{};
        a.db = fdb;
        a.n = n;
        a.ld = fld;
        a.d_pad = d_pad;
        a.qs = qs;
        a.thr = thr;
        a.cand = cand;
        a.cnt = cnt;
        a.cap = cap;
        a.sample_out = h->sample.as<float>();
        a.ns = ns;
        a.nqt = nqt;
        a.add_norm = cosine ? 0 : 1;
        a.debug = g_opt.dense_debug;
        // sample pass
        a.mode = 1;
        a.tile_first = 0;
        a.tile_step = stride;
        a.n_tiles = ns_tiles;
        a.nrb = nrb;
        {
            const int wv = scan_geometry(d_pad).waves;
            if (ns_tiles < (long long)nrb * wv) a.nrb = (int)(((ns_tiles + wv - 1) / wv + 7) / 8 * 8);
        }
        SQ_TRY(scan_launch(a, st));
        hipLaunchKernelGGL(fill_f32_kernel, dim3((nq_pad + 255) / 256), dim3(256), 0, st, thr, (long long)nq_pad,
                           -__builtin_inff());
        hipLaunchKernelGGL(kth_threshold_f32_kernel, dim3(nq), dim3(1024), 0, st, a.sample_out, ns, kk, thr);
        hipLaunchKernelGGL(dense_inflate_thr_kernel, dim3((nq + 63) / 64), dim3(64), 0, st, thr, qn2, nq, cosine ? 1 : 0,
                           h->xn2_max, eps_coef);
        SQ_HIP(hipMemsetAsync(cnt, 0, (size_t)nq_pad * 4, st));
        // full pass
        a.mode = 0;
        a.tile_step = 1;
        a.n_tiles = n_tiles;
        a.nrb = nrb;
        if (prof) SQ_HIP(hipEventRecord(h->ev[1], st));
        SQ_TRY(scan_launch(a, st));
        if (prof) SQ_HIP(hipEventRecord(h->ev[2], st));
        h->stats.scan_launches = 2;
        h->stats.bytes_scanned = n * (long long)d * 4;
        // exact re-rank of the candidates, select, certify
        const unsigned gx = 64;
        if (cosine) {
            hipLaunchKernelGGL(dense_exact_cos_kernel, dim3(gx, nq), dim3(256), 0, st, h->db, h->ld, d, q, cand, cnt, cap,
                               0ll, 0ll, h->keys.as<K128>(), key_stride);
            SQ_TRY(select_launch_t<K128>(h->keys.as<K128>(), cnt, cap, key_stride, k, nq, h->out_keys.as<K128>(), st));
            hipLaunchKernelGGL(dense_finalize_cos_kernel, dim3(nq), dim3(256), 0, st, h->out_keys.as<K128>(), cnt, cap, k,
                               kk, h->id_base, thr, eps_coef, 1, (double*)out_dist, out_idx, status);
        } else {
            hipLaunchKernelGGL(dense_exact_l2_kernel, dim3(gx, nq), dim3(256), l2_lds, st, h->db, h->ld, d, q, cand, cnt, cap,
                               0ll, 0ll, h->keys.as<u64>(), key_stride);
            SQ_TRY(select_launch_t<u64>(h->keys.as<u64>(), cnt, cap, key_stride, k, nq, h->out_keys.as<u64>(), st));
            hipLaunchKernelGGL(dense_finalize_l2_kernel, dim3(nq), dim3(256), 0, st, h->out_keys.as<u64>(), cnt, cap, k,
                               kk, h->id_base, thr, qn2, h->xn2_max, eps_coef, 1, (float*)out_dist, out_idx, status);
        }
    } else {
        all_fallback = true;  // dimension outside the MFMA scan's LDS budget: exact path for every query
    }
    if (prof) SQ_HIP(hipEventRecord(h->ev[3], st));
    if (!all_fallback) {
        SQ_HIP(hipMemcpyAsync(hs, status, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
        SQ_HIP(hipMemcpyAsync(hs + nq, cnt, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
        SQ_HIP(hipStreamSynchronize(st));
        SQ_HIP(hipGetLastError());
        if (prof) {
            float t1 = 0, t2 = 0;
            SQ_HIP(hipEventElapsedTime(&t1, h->ev[1], h->ev[2]));
            SQ_HIP(hipEventElapsedTime(&t2, h->ev[0], h->ev[3]));
            h->stats.scan_ms = t1;
            h->stats.total_ms = t2;
        }
        for (int qi = 0; qi < nq; ++qi) h->stats.candidates += hs[nq + qi];
    }
    // exact full-keys path, one query at a time: keys for all n rows -> radix select
    for (int qi = 0; qi < nq; ++qi) {
        const bool need = all_fallback || (!small && (hs[qi] != 0 || force_fb));
        if (!need) continue;
        h->stats.fallback_queries++;
        SQ_TRY(h->big_keys.reserve((size_t)n * key_bytes));
        hipLaunchKernelGGL(fill_u32_kernel, dim3(1), dim3(64), 0, st, cnt + qi, 1ll, (u32)n);
        unsigned gx = (unsigned)((n + 31) / 32);
        if (gx > 8192) gx = 8192;
        if (cosine) {
            hipLaunchKernelGGL(dense_exact_cos_kernel, dim3(gx, 1), dim3(256), 0, st, h->db, h->ld, d,
                               q + (long long)qi * d, nullptr, cnt + qi, (u32)n, n, 0ll, h->big_keys.as<K128>(), n);
            SQ_TRY(select_launch_t<K128>(h->big_keys.as<K128>(), cnt + qi, (u32)n, n, k, 1,
                                         h->out_keys.as<K128>() + (long long)qi * k, st));
            hipLaunchKernelGGL(dense_finalize_cos_kernel, dim3(1), dim3(256), 0, st,
                               h->out_keys.as<K128>() + (long long)qi * k, cnt + qi, (u32)n, k, kk, h->id_base, thr, 0.0,
                               0, (double*)out_dist + (long long)qi * k, out_idx + (long long)qi * k, status + qi);
        } else {
            hipLaunchKernelGGL(dense_exact_l2_kernel, dim3(gx, 1), dim3(256), l2_lds, st, h->db, h->ld, d,
                               q + (long long)qi * d, nullptr, cnt + qi, (u32)n, n, 0ll, h->big_keys.as<u64>(), n);
            SQ_TRY(select_launch_t<u64>(h->big_keys.as<u64>(), cnt + qi, (u32)n, n, k, 1,
                                        h->out_keys.as<u64>() + (long long)qi * k, st));
            hipLaunchKernelGGL(dense_finalize_l2_kernel, dim3(1), dim3(256), 0, st,
                               h->out_keys.as<u64>() + (long long)qi * k, cnt + qi, (u32)n, k, kk, h->id_base, thr, qn2,
                               0.0, 0.0, 0, (float*)out_dist + (long long)qi * k, out_idx + (long long)qi * k,
                               status + qi);
        }
        h->stats.scan_launches++;
    }
    if (h->stats.fallback_queries) {
        SQ_HIP(hipStreamSynchronize(st));
        SQ_HIP(hipGetLastError());
    }
    return SQ_OK;
}

}  // namespace sq

using namespace sq;

extern "C" int sq_dense_create(const float* db, int64_t n, int d, int metric, int mem, int64_t id_base,
                               sq_handle_t* out) {
    if (!db || !out || n <= 0 || d <= 0) return fail(SQ_ERR_INVALID, "sq_dense_create: bad argument");
    if (metric != SQ_METRIC_L2 && metric != SQ_METRIC_COSINE)
        return fail(SQ_ERR_INVALID, "sq_dense_create: unknown metric %d", metric);
    if (n >= (1ll << 32)) return fail(SQ_ERR_UNSUPPORTED, "sq_dense_create: more than 2^32-1 rows per shard");
    const int d_pad = (d + KT - 1) / KT * KT;
    if (mem == SQ_MEM_DEVICE && d != d_pad)
        return fail(SQ_ERR_UNSUPPORTED, "sq_dense_create: borrowing a device matrix needs d %% 64 == 0 (d=%d)", d);
    if (mem == SQ_MEM_DEVICE && (reinterpret_cast<uintptr_t>(db) & 15u) != 0)
        return fail(SQ_ERR_UNSUPPORTED, "sq_dense_create: device matrix must be 16-byte aligned");
    auto* h = new DenseHandle();
    h->kind = H_DENSE;
    h->n = n;
    h->d = d;
    h->d_pad = d_pad;
    h->ld = d_pad;
    h->metric = metric;
    h->id_base = id_base;
    if (hipGetDevice(&h->device) != hipSuccess) {
        delete h;
        return fail(SQ_ERR_HIP, "sq_dense_create: no HIP device");
    }
    auto bail = [&](int rc) {
        delete h;
        return rc;
    };
    if (mem == SQ_MEM_DEVICE) {
        h->db = db;
    } else {
        const size_t bytes = (size_t)n * d_pad * 4;
        int rc = h->owned.reserve(bytes);
        if (rc != SQ_OK) return bail(rc);
        hipError_t e;
        if (d == d_pad) {
            e = hipMemcpy(h->owned.p, db, bytes, hipMemcpyHostToDevice);
        } else {
            e = hipMemcpy2D(h->owned.p, (size_t)d_pad * 4, db, (size_t)d * 4, (size_t)d * 4, (size_t)n,
                            hipMemcpyHostToDevice);
            if (e == hipSuccess) {
                // zero the padding columns
                e = hipMemset2D(reinterpret_cast<char*>(h->owned.p) + (size_t)d * 4, (size_t)d_pad * 4, 0,
                                (size_t)(d_pad - d) * 4, (size_t)n);
            }
        }
        if (e != hipSuccess) return bail(fail(SQ_ERR_HIP, "sq_dense_create: H2D copy failed: %s", hipGetErrorString(e)));
        h->db = h->owned.as<float>();
    }
    // row statistics (+ unit-length copy for cosine)
    {
        int rc = h->scratch.reserve(256);
        if (rc != SQ_OK) return bail(rc);
        if (hipMemset(h->scratch.p, 0, 256) != hipSuccess) return bail(fail(SQ_ERR_HIP, "memset failed"));
        float* norm = nullptr;
        if (metric == SQ_METRIC_COSINE) {
            rc = h->normalized.reserve((size_t)n * d_pad * 4);
            if (rc != SQ_OK) return bail(rc);
            norm = h->normalized.as<float>();
        }
        hipLaunchKernelGGL(dense_rowstats_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, 0, h->db, n, h->ld, d,
                           d_pad, h->scratch.as<u32>(), norm);
        u32 bits = 0;
        hipError_t e = hipMemcpy(&bits, h->scratch.p, 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return bail(fail(SQ_ERR_HIP, "sq_dense_create: row statistics failed: %s", hipGetErrorString(e)));
        float f;
        memcpy(&f, &bits, 4);
        h->xn2_max = (double)f;
    }
    *out = register_handle(h);
    return SQ_OK;
}

extern "C" int sq_dense_search(sq_handle_t hid, const float* queries, int nq, int k, void* out_dist, int64_t* out_idx,
                               int mem, void* stream) {
    auto* h = static_cast<DenseHandle*>(lookup_handle(hid, H_DENSE));
    if (!h) return fail(SQ_ERR_INVALID, "sq_dense_search: unknown handle");
    if (!queries || !out_dist || !out_idx || nq <= 0 || k <= 0) return fail(SQ_ERR_INVALID, "sq_dense_search: bad argument");
    if (k > SQ_MAX_K) return fail(SQ_ERR_UNSUPPORTED, "sq_dense_search: k=%d exceeds SQ_MAX_K=%d", k, SQ_MAX_K);
    if (h->metric == SQ_METRIC_COSINE && k > kSelectLdsKeys128)
        return fail(SQ_ERR_UNSUPPORTED, "sq_dense_search: cosine k=%d exceeds %d", k, kSelectLdsKeys128);
    std::lock_guard<std::mutex> lock(h->mu);
    SQ_HIP(hipSetDevice(h->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t dsz = h->metric == SQ_METRIC_COSINE ? 8 : 4;
    if (mem == SQ_MEM_DEVICE)
        return dense_search_device(h, queries, nq, k, out_dist, reinterpret_cast<long long*>(out_idx), st);
    const size_t qb = (size_t)nq * h->d * 4;
    SQ_TRY(h->q_dev.reserve(qb));
    SQ_TRY(h->out_dist_dev.reserve((size_t)nq * k * dsz));
    SQ_TRY(h->out_idx_dev.reserve((size_t)nq * k * 8));
    SQ_HIP(hipMemcpyAsync(h->q_dev.p, queries, qb, hipMemcpyHostToDevice, st));
    SQ_TRY(dense_search_device(h, h->q_dev.as<float>(), nq, k, h->out_dist_dev.p, h->out_idx_dev.as<long long>(), st));
    SQ_HIP(hipMemcpyAsync(out_dist, h->out_dist_dev.p, (size_t)nq * k * dsz, hipMemcpyDeviceToHost, st));
    SQ_HIP(hipMemcpyAsync(out_idx, h->out_idx_dev.p, (size_t)nq * k * 8, hipMemcpyDeviceToHost, st));
    SQ_HIP(hipStreamSynchronize(st));
    return SQ_OK;
}

extern "C" int sq_dense_destroy(sq_handle_t hid) {
    auto* h = remove_handle(hid, H_DENSE);
    if (!h) return fail(SQ_ERR_INVALID, "sq_dense_destroy: unknown handle");
    (void)hipSetDevice(h->device);
    delete h;
    return SQ_OK;
}

template <class T>
static void distances_launch(const void* rows, long long n, int d, const void* q, int metric, void* out,
                             hipStream_t st) {
    const unsigned gx = (unsigned)((n + 31) / 32);
    hipLaunchKernelGGL((dense_distances_kernel<T>), dim3(gx), dim3(256), 0, st, (const T*)rows, n, d, (const T*)q,
                       metric, (T*)out, (double*)out);
}

extern "C" int sq_dense_distances(const void* query, const void* rows, int dtype, int64_t n, int d, int metric,
                                  void* out, int mem, void* stream) {
    if (!query || !rows || !out || n <= 0 || d <= 0) return fail(SQ_ERR_INVALID, "sq_dense_distances: bad argument");
    if (metric != SQ_METRIC_L2 && metric != SQ_METRIC_COSINE) return fail(SQ_ERR_INVALID, "sq_dense_distances: unknown metric");
    if (dtype != SQ_DTYPE_F32 && dtype != SQ_DTYPE_F64) return fail(SQ_ERR_INVALID, "sq_dense_distances: unknown dtype");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t esz = dtype == SQ_DTYPE_F32 ? 4 : 8;
    const size_t osz = (metric == SQ_METRIC_COSINE || dtype == SQ_DTYPE_F64) ? 8 : 4;
    if (mem == SQ_MEM_DEVICE) {
        if (dtype == SQ_DTYPE_F32)
            distances_launch<float>(rows, n, d, query, metric, out, st);
        else
            distances_launch<double>(rows, n, d, query, metric, out, st);
        SQ_HIP(hipGetLastError());
        return SQ_OK;
    }
    DevBuf dq, dr, dout;
    int rc = SQ_OK;
    auto done = [&](int code) {
        dq.release();
        dr.release();
        dout.release();
        return code;
    };
    if ((rc = dq.reserve((size_t)d * esz)) != SQ_OK) return done(rc);
    if ((rc = dr.reserve((size_t)n * d * esz)) != SQ_OK) return done(rc);
    if ((rc = dout.reserve((size_t)n * osz)) != SQ_OK) return done(rc);
    if (hipMemcpyAsync(dq.p, query, (size_t)d * esz, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(dr.p, rows, (size_t)n * d * esz, hipMemcpyHostToDevice, st) != hipSuccess)
        return done(fail(SQ_ERR_HIP, "sq_dense_distances: H2D copy failed"));
    if (dtype == SQ_DTYPE_F32)
        distances_launch<float>(dr.p, n, d, dq.p, metric, dout.p, st);
    else
        distances_launch<double>(dr.p, n, d, dq.p, metric, dout.p, st);
    if (hipMemcpyAsync(out, dout.p, (size_t)n * osz, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return done(fail(SQ_ERR_HIP, "sq_dense_distances: kernel or D2H copy failed: %s",
                         hipGetErrorString(hipGetLastError())));
    return done(SQ_OK);
}
