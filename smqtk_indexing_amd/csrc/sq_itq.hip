// ITQ hash-code generation (gfx950): rotation + sign + MSB-first packing.
//
// Replaces, for n descriptors at once, ItqFunctor.get_hash
// (smqtk_indexing/impls/lsh_functor/itq.py:389-408), its _norm_vector
// (itq.py:172-191) and bit_vector_to_int_large (utils/bits.py:4-20), i.e. the
// body of the hashing loop of LSHNearestNeighborIndex._build_index
// (impls/nn_index/lsh.py:316-321):
//     v = x / ||x||_2 (optional, in x's dtype, zero norm -> 1)
//     z = (v - mean) . R        float64 (mean, R are float64)
//     bit_j = z_j >= 0          (exact zero and -0.0 map to 1)
// The contraction runs on v_mfma_f64_16x16x4_f64 (A = 16 rows x 4 k of v,
// B = 4 k x 16 hash bits of R from LDS).  Sign bits leave the accumulators
// through wave ballots and are packed so that hash bit 0 is the most
// significant bit of the right-aligned uint64[W] code.
#include <algorithm>

#include "sq_common.hpp"
#include "sq_pairwise.hpp"
#include "sq_itq_fast.hpp"

namespace sq {

typedef double f64x4 __attribute__((ext_vector_type(4)));

struct ItqArgs {
    const void* x;
    long long n;
    int d;
    const double* mean;
    const double* rot;  // [d][bits]
    int bits;
    int words;          // W = ceil(bits/64)
    int pad;            // W*64 - bits leading zero columns
    int norm;           // SQ_NORM_NONE / SQ_NORM_L2 / _L1 / _L0 / _INF / _NEG_INF
    u64* out;           // [n][W]
    int dk;             // k rows of R staged per chunk (multiple of 16)
    int nchunks;
    int d16;            // d rounded up to 16
    const void* nrm;    // [n] row L2 norms in x's dtype (normalize=2), from itq_norms_kernel
    int vec4;           // rows are 4-element aligned (d % 4 == 0, base aligned): vector loads of x
    int sub32;              // x - mean in float32 (float32 rows and a float32 model mean: numpy's promotion)
    const u32* list;        // optional: only these rows (the filter's uncertain rows, sq_itq_fast.hpp)
    const u32* list_total;  // device count of `list`
    int exact;              // option "itq_exact" of the call (the model handle's override, or the process-wide value)
    int debug;              // option "dense_debug" of the call (ablation bits)
};

template <class T>
struct Vec4;
template <>
struct Vec4<float> {
    typedef float type __attribute__((ext_vector_type(4)));
};
template <>
struct Vec4<double> {
    typedef double type __attribute__((ext_vector_type(4)));
};

__device__ __forceinline__ float div_rn(float a, float b) { return __fdiv_rn(a, b); }
__device__ __forceinline__ double div_rn(double a, double b) { return __ddiv_rn(a, b); }
__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ double mul_rn(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ float sqrt_rn(float a) { return (float)sqrt((double)a); }
__device__ __forceinline__ double sqrt_rn(double a) { return sqrt(a); }

// Row norms in numpy's arithmetic (itq.py:185: np.linalg.norm(v, ord, axis, keepdims), numpy/linalg/linalg.py):
//   ord 2:    sqrt(add.reduce(x * x))      pairwise float sum in x's dtype, correctly rounded sqrt
//   ord 1:    add.reduce(abs(x))           the same pairwise sum of |x|
//   ord 0:    (x != 0).astype(dtype).sum() (exact: small integers)
//   ord inf:  abs(x).max()     ord -inf: abs(x).min()      (a NaN wins, as in numpy's maximum / minimum)
// and 0 -> 1 (itq.py:187).  8 lanes per row.  Kept out of the MFMA kernel so that one stays within 256
// VGPRs (two workgroups per CU) without spilling.
__device__ __forceinline__ float abs_t(float v) { return fabsf(v); }
__device__ __forceinline__ double abs_t(double v) { return fabs(v); }
template <class T>
__global__ __launch_bounds__(256) void itq_norms_kernel(const T* __restrict__ X, long long n_all, int d, T* __restrict__ nrm,
                                                        const u32* __restrict__ list, const u32* __restrict__ list_total,
                                                        int ord) {
    const int j8 = threadIdx.x & 7;
    const long long stride = (long long)gridDim.x * 32;
    const long long n = list ? (long long)*list_total : n_all;  // listed rows only (sq_itq_fast.hpp), or all
    for (long long row0 = (long long)blockIdx.x * 32; row0 < n; row0 += stride) {
        long long row = row0 + (threadIdx.x >> 3);
        const bool live = row < n;
        row = live ? row : n - 1;
        if (list) row = (long long)list[row];
        const T* xr = X + row * d;
        T nv;
        if (ord == SQ_NORM_INF || ord == SQ_NORM_NEG_INF) {
            const bool mx = ord == SQ_NORM_INF;
            T m = abs_t(xr[j8 < d ? j8 : 0]);
            bool nan = m != m;
            for (int i = j8 + 8; i < d; i += 8) {
                const T v = abs_t(xr[i]);
                nan |= v != v;
                m = mx ? (v > m ? v : m) : (v < m ? v : m);
            }
            for (int o = 1; o < 8; o <<= 1) {
                const T v = __shfl_xor(m, o);
                nan |= (bool)__shfl_xor((int)nan, o);
                m = mx ? (v > m ? v : m) : (v < m ? v : m);
            }
            nv = nan ? (T)__builtin_nanf("") : m;
        } else if (ord == SQ_NORM_L1) {
            auto term = [xr](int i) { return abs_t(xr[i]); };
            nv = np_pairwise_sum<T>(term, d, j8);
        } else if (ord == SQ_NORM_L0) {
            auto term = [xr](int i) { return xr[i] != (T)0 ? (T)1 : (T)0; };
            nv = np_pairwise_sum<T>(term, d, j8);
        } else {
            auto term = [xr](int i) { return mul_rn(xr[i], xr[i]); };
            nv = sqrt_rn(np_pairwise_sum<T>(term, d, j8));
        }
        if (nv == (T)0) nv = (T)1;
        if (live && j8 == 0) nrm[row] = nv;
    }
}

// CT column tiles of 16 hash bits per pass (CT*16 padded columns), RT = 16/CT
// row tiles of 16 rows per wave.  grid.y walks groups of CT*16 columns.
template <class T, int CT>
__global__ __launch_bounds__(256, 2) void itq_hash_kernel(ItqArgs a) {
    constexpr int RT = 16 / CT;
    constexpr int NCOL = CT * 16;
    constexpr int RSTRIDE = NCOL + 4;        // f64 per staged R row (+32 B: lanes l and l+16 hit different bank halves)
    constexpr int ROWS_PER_WAVE = RT * 16;
    constexpr int ROWS_PER_BLOCK = 4 * ROWS_PER_WAVE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* s_mean = reinterpret_cast<double*>(smem);                 // [d16]
    double* s_rot = s_mean + a.d16;                                    // [dk][RSTRIDE]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const T* X = reinterpret_cast<const T*>(a.x);
    const int col0 = blockIdx.y * NCOL;      // first padded column of this group

    for (int i = threadIdx.x; i < a.d16; i += 256) s_mean[i] = i < a.d ? a.mean[i] : 0.0;

    auto stage_rot = [&](int chunk) {
        const int k0 = chunk * a.dk;
        for (int e = threadIdx.x; e < a.dk * NCOL; e += 256) {
            const int kr = e / NCOL, pc = e - kr * NCOL;
            const int k = k0 + kr;
            const int b = col0 + pc - a.pad;
            double v = 0.0;
            if (k < a.d && b >= 0 && b < a.bits) v = a.rot[(long long)k * a.bits + b];
            s_rot[kr * RSTRIDE + pc] = v;
        }
    };
    if (a.nchunks == 1) stage_rot(0);
    __syncthreads();

    // rows to do: all n, or the listed ones (virtual row v -> list[v])
    const long long nrows = a.list ? (long long)*a.list_total : a.n;
    const long long nblocks = (nrows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    for (long long blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const long long wrow0 = blk * ROWS_PER_BLOCK + (long long)wave * ROWS_PER_WAVE;
        f64x4 acc[RT][CT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = f64x4{0.0, 0.0, 0.0, 0.0};
        T nrm_l[RT];
        const T* xrow[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            long long row = wrow0 + rt * 16 + l15;
            row = row < nrows ? row : nrows - 1;
            if (a.list) row = (long long)a.list[row];
            xrow[rt] = X + row * a.d;
            nrm_l[rt] = a.nrm ? reinterpret_cast<const T*>(a.nrm)[row] : (T)1;
        }
        for (int chunk = 0; chunk < a.nchunks; ++chunk) {
            if (a.nchunks > 1) {
                __syncthreads();
                stage_rot(chunk);
                __syncthreads();
            }
            const int k0 = chunk * a.dk;
            for (int c = 0; c < a.dk; c += 16) {
                const int kb = k0 + c + 4 * g;  // this lane's 4 consecutive k
                if (k0 + c >= a.d16) break;
                double av[RT][4];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    T xq[4];
                    if (a.vec4 && kb < a.d) {  // d % 4 == 0 and 16-byte aligned rows: one vector load
                        const typename Vec4<T>::type v4 = *reinterpret_cast<const typename Vec4<T>::type*>(xrow[rt] + kb);
#pragma unroll
                        for (int j = 0; j < 4; ++j) xq[j] = v4[j];
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) xq[j] = (kb + j < a.d) ? xrow[rt][kb + j] : (T)0;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int k = kb + j;
                        double v = 0.0;
                        if (k < a.d) {
                            T xv = xq[j];
                            if (a.nrm) xv = div_rn(xv, nrm_l[rt]);
                            if constexpr (sizeof(T) == 4) {
                                if (a.sub32)
                                    v = (double)__fsub_rn(xv, (float)s_mean[k]);  // s_mean[k] is a float32 value
                                else
                                    v = __dsub_rn((double)xv, s_mean[k]);
                            } else {
                                v = __dsub_rn((double)xv, s_mean[k]);
                            }
                        }
                        av[rt][j] = v;
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const double* brow = s_rot + (size_t)(c + 4 * g + j) * RSTRIDE + l15;
                    double bv[CT];
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) bv[ct] = brow[ct * 16];
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct)
                            acc[rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[rt][j], bv[ct], acc[rt][ct], 0, 0, 0);
                }
            }
        }
        // ---- sign bits -> packed words.  D layout: col = lane&15, row = (lane>>4) + 4*reg
        constexpr int WPG = (CT + 3) / 4;  // words per column group
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            u64 cw[4][WPG];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int w = 0; w < WPG; ++w) cw[r][w] = 0ull;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const u64 m = __ballot(acc[rt][ct][r] >= 0.0);
                    const u32 m16 = (u32)(m >> (16 * g)) & 0xffffu;
                    const u64 rev = (u64)(__brev(m16) >> 16);  // column 0 -> most significant of the 16
                    cw[r][ct / 4] |= rev << (48 - 16 * (ct % 4));
                }
            }
            if (l15 == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    long long row = wrow0 + rt * 16 + g + 4 * r;
                    if (row < nrows) {
                        if (a.list) row = (long long)a.list[row];
#pragma unroll
                        for (int w = 0; w < WPG; ++w) {
                            const int gw = blockIdx.y * (NCOL / 64) + w;
                            if (gw < a.words) {
                                u64 v = cw[r][w];
                                if (gw == 0 && a.pad > 0) v &= (~0ull) >> a.pad;
                                a.out[row * a.words + gw] = v;
                            }
                        }
                    }
                }
            }
        }
    }
}

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef double f64x2_t __attribute__((ext_vector_type(2)));

// The filter's undecided bits, one float64 evaluation each: z_b = sum_k v_k R[k][b], v as the float64
// kernel above forms it (x / |x| in float32 with numpy's norm, minus the mean in the promoted dtype).
// Workgroups (w, 0..3) share segment w of the filter's output; 32 lanes per entry (row, column tile,
// mask of undecided columns): lane l holds elements 4l..4l+3 of every 128-element stretch, so the row
// (a random 512-byte read) and the column of the column-major float64 copy of R arrive as whole
// cache lines.  The filter already stored the sign of its own estimate; the bit is set to the float64
// sign in place.  (A whole-row float64 MFMA recompute of the ~4 % of rows owning such a bit cost
// 0.31 ms at 10 M x 128 -> 64 bits -- 64x the flops needed; one or eight lanes per entry cost as much:
// every 16-byte piece of a row then pulls its own cache line through L2.)
#ifndef SQ_ITQ_FIX_PARTS
#define SQ_ITQ_FIX_PARTS 4
#endif
static constexpr int ITQ_FIX_PARTS = SQ_ITQ_FIX_PARTS;
static __global__ __launch_bounds__(256) void itq_fix_bits_kernel(ItqArgs a, const u64* __restrict__ seg,
                                                                  const u32* __restrict__ seg_cnt, long long seg_cap,
                                                                  const double* __restrict__ rt64) {
    const long long w = blockIdx.x;
    const u32 cnt = seg_cnt[w];
    const int l32 = threadIdx.x & 31, slot = (threadIdx.x >> 5) + 8 * blockIdx.y;
    const float* X = reinterpret_cast<const float*>(a.x);
    constexpr u32 STEP = 8 * ITQ_FIX_PARTS;
    u64 ent_next = slot < (int)cnt ? seg[w * seg_cap + slot] : 0ull;
    for (u32 e = slot; e < cnt; e += STEP) {  // uniform within a 32-lane half wave
        // per entry the dependent chain is entry -> (row | R column) -> sum: the next entry is requested a turn
        // early and the result goes out as a fire-and-forget atomic
        const u64 ent = ent_next;
        if (e + STEP < cnt) ent_next = seg[w * seg_cap + e + STEP];
        const long long row = (long long)((u32)(ent >> 32) & 0x3fffffffu);
        const int ct = (int)(ent >> 62);
        u32 mask = (u32)ent;
        const float* xr = X + row * a.d;
        // the first (usually only) undecided column's slice of R: requested together with the row
        f64x2_t rfirst[2][2];
        {
            const double* rcol = rt64 + (long long)(ct * 32 + __ffs((int)mask) - 1) * a.d;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int k = 128 * t + 4 * l32;
                rfirst[t][0] = rfirst[t][1] = f64x2_t{0.0, 0.0};
                if (k < a.d) {
                    rfirst[t][0] = *reinterpret_cast<const f64x2_t*>(rcol + k);
                    rfirst[t][1] = *reinterpret_cast<const f64x2_t*>(rcol + k + 2);
                }
            }
        }
        // this lane's elements: 4 l32 + 128 t + 0..3, t < d/128 rounded up; v = x/|x| - mean formed once
        double v[2][4];  // d <= 256
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int k = 128 * t + 4 * l32;
            const bool in = k < a.d;
            f32x4_t xv = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (in) xv = *reinterpret_cast<const f32x4_t*>(xr + k);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[t][j] = (double)xv[j];  // raw value for now
        }
        float nrm = 1.f;
        if (a.norm == SQ_NORM_L2) {
            // numpy's pairwise order needs the row's own layout: eight cooperating lanes (np_pairwise_sum), every
            // aligned group of 8 computes the same value
            auto term = [xr](int i) { return mul_rn(xr[i], xr[i]); };
            nrm = sqrt_rn(np_pairwise_sum<float>(term, a.d, threadIdx.x & 7));
            if (nrm == 0.f) nrm = 1.f;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int k = 128 * t + 4 * l32;
            if (k < a.d) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float xe = (float)v[t][j];
                    if (a.norm == SQ_NORM_L2) xe = div_rn(xe, nrm);
                    v[t][j] = a.sub32 ? (double)__fsub_rn(xe, (float)a.mean[k + j]) : __dsub_rn((double)xe, a.mean[k + j]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[t][j] = 0.0;
            }
        }
        bool first = true;
        while (mask) {  // nearly always one bit
            const int pc = ct * 32 + __ffs((int)mask) - 1;   // padded column; the filter only flags pc >= pad
            mask &= mask - 1;
            const double* rcol = rt64 + (long long)pc * a.d;   // column pc of R, contiguous (itq_fast_prep_kernel)
            double z = 0.0;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int k = 128 * t + 4 * l32;
                if (k < a.d) {
                    f64x2_t r0 = rfirst[t][0], r1 = rfirst[t][1];
                    if (!first) {
                        r0 = *reinterpret_cast<const f64x2_t*>(rcol + k);
                        r1 = *reinterpret_cast<const f64x2_t*>(rcol + k + 2);
                    }
                    z = __fma_rn(v[t][0], r0[0], z);
                    z = __fma_rn(v[t][1], r0[1], z);
                    z = __fma_rn(v[t][2], r1[0], z);
                    z = __fma_rn(v[t][3], r1[1], z);
                }
            }
            z += __shfl_xor(z, 16);
            z += __shfl_xor(z, 8);
            z += __shfl_xor(z, 4);
            z += __shfl_xor(z, 2);
            z += __shfl_xor(z, 1);
            first = false;
            if (l32 == 0) {  // set the bit to the float64 sign (no read of the word: nothing to wait for)
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

template <class T, int CT>
static int itq_launch_t(const ItqArgs& a0, hipStream_t st, int device) {
    ItqArgs a = a0;
    constexpr int NCOL = CT * 16, RSTRIDE = NCOL + 4, RT = 16 / CT;
    const size_t fixed = (size_t)a.d16 * 8;
    const size_t budget = 76 * 1024;  // two workgroups per CU: one hides the other's load + conversion phase
    if (fixed + (size_t)16 * RSTRIDE * 8 > budget)
        return fail(SQ_ERR_UNSUPPORTED, "sq_itq_hash: d=%d too large for the LDS mean vector", a.d);
    int dk = (int)((budget - fixed) / ((size_t)RSTRIDE * 8));
    dk = dk / 16 * 16;
    if (dk > a.d16) dk = a.d16;
    a.dk = dk;
    a.nchunks = (a.d16 + dk - 1) / dk;
    a.vec4 = (a.d % 4 == 0) && (reinterpret_cast<uintptr_t>(a.x) % (4 * sizeof(T)) == 0);
    const size_t lds = fixed + (size_t)dk * RSTRIDE * 8;
    static bool attr_set = false;
    if (!attr_set) {
        SQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&itq_hash_kernel<T, CT>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const long long rows_per_block = 4ll * RT * 16;
    const long long nblocks = (a.n + rows_per_block - 1) / rows_per_block;
    long long gx = 2ll * cu_count(device);
    if (gx > nblocks) gx = nblocks;
    const int groups = (a.words * 64 + NCOL - 1) / NCOL;
    hipLaunchKernelGGL((itq_hash_kernel<T, CT>), dim3((unsigned)gx, (unsigned)groups), dim3(256), lds, st, a);
    SQ_HIP(hipGetLastError());
    return SQ_OK;
}

// ---- the certified bf16x3 filter in front of the float64 kernel (sq_itq_fast.hpp)
template <int WAVES, int NSTAGE, int KU, int CT, bool NORMED, bool BREG>
static int itq_fast_launch_t(const ItqFastArgs& fa, size_t lds, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        SQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&itq_fast_kernel<WAVES, NSTAGE, KU, CT, NORMED, BREG>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL((itq_fast_kernel<WAVES, NSTAGE, KU, CT, NORMED, BREG>), dim3((unsigned)fa.nrb), dim3(WAVES * 64), lds,
                       st, fa);
    SQ_HIP(hipGetLastError());
    return SQ_OK;
}

// Geometry of the filter for (d, words); stages == 0: the filter does not apply.
struct ItqFastGeom {
    int ku, ct, stages, waves;
    bool breg;
    size_t lds;
};
static ItqFastGeom itq_fast_geometry(int d, int words) {
    ItqFastGeom g{};
    if (d % 64 != 0 || d > 256 || words > 2) return g;
    g.ku = d / 64;
    g.ct = words * 2;
    // R's hi fragments in registers (lo planes in LDS), eight waves -- unless four column tiles of accumulators are
    // live as well (64-d -> 128 bits): that spilled, and a scratch reload in the loop drains the DMA ring
    g.breg = g.ku * g.ct <= 4 && g.ct <= 2;
    g.waves = g.breg ? ITQF_WAVES_BREG : ITQF_WAVES_LDSB;
    const int dp = (d + 127) / 128 * 128;
    // LDS copy of R: both bfloat16 planes, or only the lo planes when the hi fragments live in registers
    const size_t fixed = (size_t)g.ct * 32 * dp * (g.breg ? 2 : 4);
    for (int ns = g.breg ? 2 : 4; ns >= 2; --ns) {
        const size_t lds = fixed + (size_t)g.waves * ns * ITQF_UNIT_BYTES;
        if (lds <= 160 * 1024) {
            g.stages = ns;
            g.lds = lds;
            break;
        }
    }
    return g;
}

template <bool NORMED>
static int itq_fast_dispatch(const ItqFastArgs& fa, const ItqFastGeom& g, hipStream_t st) {
    if (g.breg) {
        switch (g.ku * 10 + g.ct) {
            case 12: return itq_fast_launch_t<8, 2, 1, 2, NORMED, true>(fa, g.lds, st);
            default: return itq_fast_launch_t<8, 2, 2, 2, NORMED, true>(fa, g.lds, st);  // 22
        }
    }
#define SQ_ITQF_CASE(KUv, CTv)                                                                                 \
    case KUv * 10 + CTv:                                                                                       \
        if (g.stages == 4) return itq_fast_launch_t<4, 4, KUv, CTv, NORMED, false>(fa, g.lds, st);             \
        if (g.stages == 3) return itq_fast_launch_t<4, 3, KUv, CTv, NORMED, false>(fa, g.lds, st);             \
        return itq_fast_launch_t<4, 2, KUv, CTv, NORMED, false>(fa, g.lds, st);
    switch (g.ku * 10 + g.ct) {
        SQ_ITQF_CASE(1, 4)
        SQ_ITQF_CASE(2, 4)
        SQ_ITQF_CASE(3, 2)
        SQ_ITQF_CASE(3, 4)
        SQ_ITQF_CASE(4, 2)
        SQ_ITQF_CASE(4, 4)
        default: return fail(SQ_ERR_UNSUPPORTED, "itq filter: no kernel for d=%d words=%d", g.ku * 64, g.ct / 2);
    }
#undef SQ_ITQF_CASE
}

// Stream-ordered scratch from a pool of the library's OWN (one per device).  The pool keeps up to kPoolKeepBytes
// across synchronisations, so small latency-bound calls (one query vector: ~100 KB) never pay for a fresh
// allocation, while the scratch of a bulk call (gigabytes for a 10 M-row hash from host memory) goes back to the
// driver at the next synchronisation instead of staying reserved for the life of the process.  (Round 1 raised the
// release threshold of the process-wide DEFAULT pool to "never": a global setting other users of that pool saw,
// and memory that hipMalloc-based buffers and torch's allocator could no longer get.)
static constexpr uint64_t kPoolKeepBytes = 256ull << 20;
static hipMemPool_t scratch_pool(int device) {
    static std::mutex mu;
    static hipMemPool_t pools[64] = {};
    static bool tried[64] = {};
    if (device < 0 || device >= 64) return nullptr;
    std::lock_guard<std::mutex> l(mu);
    if (!tried[device]) {
        tried[device] = true;
        hipMemPoolProps props{};
        props.allocType = hipMemAllocationTypePinned;
        props.handleTypes = hipMemHandleTypeNone;
        props.location.type = hipMemLocationTypeDevice;
        props.location.id = device;
        hipMemPool_t pool = nullptr;
        if (hipMemPoolCreate(&pool, &props) == hipSuccess) {
            uint64_t keep = kPoolKeepBytes;
            (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
            pools[device] = pool;
        } else {
            (void)hipGetLastError();
        }
    }
    return pools[device];
}
static hipError_t scratch_alloc(void** p, size_t bytes, hipStream_t st, int device) {
    hipMemPool_t pool = scratch_pool(device);
    if (pool) return hipMallocFromPoolAsync(p, bytes, pool, st);
    return hipMallocAsync(p, bytes, st);  // (no private pool: the default one, with its default threshold)
}

static size_t align256(size_t v) { return (v + 255) / 256 * 256; }

// float32 rows through the filter; the rows it cannot decide through the float64 kernel.
static int itq_fast_path(const ItqArgs& a, const ItqFastGeom& g, hipStream_t st, int device) {
    const int pc = a.words * 64;
    const bool l2 = a.norm == SQ_NORM_L2;
    const long long n_tiles = (a.n + 31) / 32;
    const int nrb = cu_count(device);
    const long long nwaves = (long long)nrb * g.waves;
    const long long seg_cap = ((n_tiles + nwaves - 1) / nwaves) * 32 * g.ct;  // every (row, column tile) of a wave's tiles
    // one stream-ordered scratch block: colnorm | c_b | c_b error | R image | segments | counts | R^T float64
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off += align256(bytes);
        return at;
    };
    const size_t o_cn = take((size_t)pc * 4), o_cb = take((size_t)pc * 4), o_cbe = take((size_t)pc * 4);
    const size_t o_cabs = take((size_t)pc * 4);
    const size_t o_img = take((size_t)pc * ((a.d + 127) / 128 * 128) * 4);
    const size_t o_seg = take((size_t)nwaves * seg_cap * 8), o_cnt = take((size_t)nwaves * 4);
    const size_t o_dummy = take((size_t)nwaves * 8);
    const size_t o_rt = take((size_t)pc * a.d * 8);
    unsigned char* base = nullptr;
    SQ_HIP(scratch_alloc(reinterpret_cast<void**>(&base), off, st, device));
    auto done = [&](int rc) {
        (void)hipFreeAsync(base, st);
        return rc;
    };
    // relative error of x . R_b per unit |x||R_b| (sq_itq_fast.hpp): 2^-20 (x: two round-toward-zero float16 planes)
    // + 2^-21 (the dropped x_lo R_lo) + 3d * 2^-24 (float32 accumulation of 3d products) + 2^-20 (the float32 scale /
    // subtract, the reference's float32 x/|x|) [+ 2^-18: float32 |x|^2, normalize=2].  R's own residual is measured
    // by the prep kernel and added to the column's coefficient.
    const double eps_rel = ((9.5367431640625e-07 + 4.76837158203125e-07) * 1.001 + 3.0 * a.d * 5.9604644775390625e-08 +
                            9.5367431640625e-07 + (l2 ? 3.814697265625e-06 : 0.0)) * 1.001;
    hipLaunchKernelGGL(itq_fast_prep_kernel, dim3((unsigned)pc), dim3(256), 0, st, a.mean, a.rot, a.d, a.bits, a.pad,
                       reinterpret_cast<unsigned short*>(base + o_img), reinterpret_cast<float*>(base + o_cn),
                       reinterpret_cast<float*>(base + o_cb), reinterpret_cast<float*>(base + o_cbe),
                       reinterpret_cast<double*>(base + o_rt), eps_rel, reinterpret_cast<float*>(base + o_cabs));
    ItqFastArgs fa{};
    fa.x = reinterpret_cast<const float*>(a.x);
    fa.n = a.n;
    fa.d = a.d;
    fa.rimage = reinterpret_cast<const uint4*>(base + o_img);
    fa.colnorm = reinterpret_cast<const float*>(base + o_cn);
    fa.cb32 = reinterpret_cast<const float*>(base + o_cb);
    fa.cberr = reinterpret_cast<const float*>(base + o_cbe);
    fa.cabs = reinterpret_cast<const float*>(base + o_cabs);
    fa.out = a.out;
    fa.words = a.words;
    fa.pad = a.pad;
    fa.bits = a.bits;
    fa.seg = reinterpret_cast<u64*>(base + o_seg);
    fa.seg_cnt = reinterpret_cast<u32*>(base + o_cnt);
    fa.seg_dummy = reinterpret_cast<u64*>(base + o_dummy);
    fa.seg_cap = seg_cap;
    fa.n_tiles = n_tiles;
    fa.nrb = nrb;
    fa.nstage = g.stages;
    int rc = l2 ? itq_fast_dispatch<true>(fa, g, st) : itq_fast_dispatch<false>(fa, g, st);
    if (rc != SQ_OK) return done(rc);
    // the undecided bits, one float64 dot product each, straight from the per-wave segments
    hipLaunchKernelGGL(itq_fix_bits_kernel, dim3((unsigned)nwaves, ITQ_FIX_PARTS), dim3(256), 0, st, a, fa.seg, fa.seg_cnt, seg_cap,
                       reinterpret_cast<const double*>(base + o_rt));
    return done(rc);
}

}  // namespace sq
#include "sq_itq_wide.hpp"   // (needs ItqArgs and the numpy-order helpers above)
namespace sq {

// The wide filter (sq_itq_wide.hpp): float32 rows beyond the narrow kernel's shapes, float64 rows of every shape it takes.
template <class T>
static bool itq_wide_applies(const ItqArgs& a) {
    return a.d % 64 == 0 && a.d <= 512 && a.words <= 4 && a.n >= 32 && a.n < (1ll << 29) &&
           (reinterpret_cast<uintptr_t>(a.x) & 15u) == 0 && !a.exact && (a.norm == SQ_NORM_NONE || a.norm == SQ_NORM_L2);
}

template <class T>
static int itq_wide_path(const ItqArgs& a, hipStream_t st, int device) {
    const int pc = a.words * 64, ct = a.words * 2;
    const bool l2 = a.norm == SQ_NORM_L2;
    const long long n_tiles = (a.n + 31) / 32;
    int nrb = cu_count(device);
    if ((long long)nrb * ITQW_WAVES > n_tiles) nrb = (int)((n_tiles + ITQW_WAVES - 1) / ITQW_WAVES);
    const long long nwaves = (long long)nrb * ITQW_WAVES;
    const long long rounds = (n_tiles + nwaves - 1) / nwaves;
    const long long seg_cap = rounds * 32 * ct;   // every (row, column tile) of a wave's tiles
    const int dp = (a.d + 127) / 128 * 128;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off += align256(bytes);
        return at;
    };
    const size_t o_cn = take((size_t)pc * 4), o_cb = take((size_t)pc * 4), o_cbe = take((size_t)pc * 4);
    const size_t o_cabs = take((size_t)pc * 4);
    const size_t o_img = take((size_t)pc * dp * 4 + 1024);   // (+ slack: the last k-block of a 384-wide plane is DMA'd whole)
    const size_t o_seg = take((size_t)nwaves * seg_cap * 8), o_cnt = take((size_t)nwaves * 4);
    const size_t o_rt = take((size_t)pc * a.d * 8);
    unsigned char* base = nullptr;
    SQ_HIP(scratch_alloc(reinterpret_cast<void**>(&base), off, st, device));
    auto done = [&](int rc) {
        (void)hipFreeAsync(base, st);
        return rc;
    };
    // relative error of x . R_b per unit |x||R_b|: 2^-20 (x: two truncated float16 planes) + 2^-21 (the dropped
    // x_lo R_lo) + the float32 accumulation: x_hi R_hi is summed per 256-k block (m = min(d, 256) products each, the
    // blocks' bounds add up under Cauchy-Schwarz), the 2 d correction products are 2^-10 of that, three final
    // additions; + 2^-20 (the float32 scale / subtract, the reference's x/|x| rounding) [+ 2^-18: float32 |x|^2].
    // (k beyond d inside a 256-k block runs with zero row fragments: the image must hold finite numbers there)
    SQ_HIP(hipMemsetAsync(base + o_img, 0, (size_t)pc * dp * 4 + 1024, st));
    const int m = a.d < 256 ? a.d : 256;
    const double eps_rel = ((9.5367431640625e-07 + 4.76837158203125e-07) * 1.001 + 1.5 * (m + 8.0) * 5.9604644775390625e-08 +
                            2.0 * a.d * 5.9604644775390625e-08 * 9.765625e-04 + 9.5367431640625e-07 + (l2 ? 3.814697265625e-06 : 0.0)) * 1.001;
    hipLaunchKernelGGL(itq_fast_prep_kernel, dim3((unsigned)pc), dim3(256), 0, st, a.mean, a.rot, a.d, a.bits, a.pad,
                       reinterpret_cast<unsigned short*>(base + o_img), reinterpret_cast<float*>(base + o_cn),
                       reinterpret_cast<float*>(base + o_cb), reinterpret_cast<float*>(base + o_cbe),
                       reinterpret_cast<double*>(base + o_rt), eps_rel, reinterpret_cast<float*>(base + o_cabs));
    ItqWideArgs wa{};
    wa.x = a.x;
    wa.n = a.n;
    wa.d = a.d;
    wa.rimage = reinterpret_cast<const uint4*>(base + o_img);
    wa.colnorm = reinterpret_cast<const float*>(base + o_cn);
    wa.cb32 = reinterpret_cast<const float*>(base + o_cb);
    wa.cberr = reinterpret_cast<const float*>(base + o_cbe);
    wa.cabs = reinterpret_cast<const float*>(base + o_cabs);
    wa.out = a.out;
    wa.words = a.words;
    wa.pad = a.pad;
    wa.bits = a.bits;
    wa.ct = ct;
    wa.seg = reinterpret_cast<u64*>(base + o_seg);
    wa.seg_cnt = reinterpret_cast<u32*>(base + o_cnt);
    wa.seg_cap = seg_cap;
    wa.n_tiles = n_tiles;
    wa.nrb = nrb;
    wa.debug = a.debug & 15;
    static DevBuf stamp_buf;   // (measurement: option dense_debug bit 16 -> phase stamps of every workgroup, printed by the host)
    wa.stamps = nullptr;
    if (a.debug & 16) {
        SQ_TRY(stamp_buf.reserve((size_t)nrb * 64 * 8));
        SQ_HIP(hipMemsetAsync(stamp_buf.p, 0, (size_t)nrb * 64 * 8, st));
        wa.stamps = stamp_buf.as<u64>();
    }   // (measurement: the ablation bits of sq_itq_wide.hpp ride on option dense_debug)
    const size_t lds = 2 * (size_t)ITQW_CHUNK_BYTES + 4 * 256 * 4 + (size_t)ITQW_WAVES * ITQW_NSTAGE * ITQF_UNIT_BYTES + ITQW_WAVES * 2048;
    auto launch = [&](auto kern) -> int {
        SQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL(kern, dim3((unsigned)nrb), dim3(ITQW_WAVES * 64), lds, st, wa);
        return SQ_OK;
    };
    int rc;
    if (a.d <= 256)
        rc = l2 ? launch(&itq_wide_kernel<T, true, 1>) : launch(&itq_wide_kernel<T, false, 1>);
    else
        rc = l2 ? launch(&itq_wide_kernel<T, true, 2>) : launch(&itq_wide_kernel<T, false, 2>);
    if (rc != SQ_OK) return done(rc);
    SQ_HIP(hipGetLastError());
    if (wa.stamps) {
        std::vector<unsigned long long> hst((size_t)nrb * 64);
        SQ_HIP(hipMemcpy(hst.data(), wa.stamps, hst.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull;
        for (int b = 0; b < nrb; ++b) if (hst[(size_t)b * 64] && hst[(size_t)b * 64] < t0) t0 = hst[(size_t)b * 64];
        for (int b : {0, 1, 7, 100, 255}) {
            if (b >= nrb) continue;
            fprintf(stderr, "wg %3d:", b);
            for (int r = 0; r < 8; ++r)
                fprintf(stderr, " [r%d x %.1f mfma %.1f | start %.1f]", r, (hst[(size_t)b * 64 + 3 * r + 1] - hst[(size_t)b * 64 + 3 * r]) / 100.0,
                        (hst[(size_t)b * 64 + 3 * r + 2] - hst[(size_t)b * 64 + 3 * r + 1]) / 100.0, (hst[(size_t)b * 64 + 3 * r] - t0) / 100.0);
            fprintf(stderr, "\n");
        }
    }
    hipLaunchKernelGGL((itq_fix_bits_wide_kernel<T>), dim3((unsigned)nwaves, ITQ_FIX_PARTS), dim3(256), 0, st, a, wa.seg, wa.seg_cnt, seg_cap,
                       reinterpret_cast<const double*>(base + o_rt));
    return done(SQ_OK);
}

template <class T>
static int itq_launch(const ItqArgs& a0, hipStream_t st, int device) {
    ItqArgs a = a0;
    if constexpr (sizeof(T) == 4) {
        const ItqFastGeom g = itq_fast_geometry(a.d, a.words);
        if (g.stages >= 2 && a.n >= 32 && a.n < (1ll << 30) && (reinterpret_cast<uintptr_t>(a.x) & 15u) == 0 &&
            !a.exact && (a.norm == SQ_NORM_NONE || a.norm == SQ_NORM_L2))  // (the other orders: float64 kernel)
            return itq_fast_path(a, g, st, device);
    }
    if (itq_wide_applies<T>(a)) return itq_wide_path<T>(a, st, device);
    void* nrm = nullptr;
    if (a.norm != SQ_NORM_NONE) {  // stream-ordered scratch: [n] norms in x's dtype
        SQ_HIP(scratch_alloc(&nrm, (size_t)a.n * sizeof(T), st, device));
        long long gx = std::min<long long>((a.n + 31) / 32, 16ll * cu_count(device));
        hipLaunchKernelGGL((itq_norms_kernel<T>), dim3((unsigned)gx), dim3(256), 0, st, reinterpret_cast<const T*>(a.x),
                           a.n, a.d, reinterpret_cast<T*>(nrm), (const u32*)nullptr, (const u32*)nullptr, a.norm);
        a.nrm = nrm;
    }
    int rc;
    if (a.words == 1) rc = itq_launch_t<T, 4>(a, st, device);
    else if (a.words == 2) rc = itq_launch_t<T, 8>(a, st, device);
    else rc = itq_launch_t<T, 16>(a, st, device);
    if (nrm) (void)hipFreeAsync(nrm, st);
    return rc;
}

static bool itq_norm_supported(int ord) {
    return ord == SQ_NORM_NONE || ord == SQ_NORM_L2 || ord == SQ_NORM_L1 || ord == SQ_NORM_L0 || ord == SQ_NORM_INF ||
           ord == SQ_NORM_NEG_INF;
}

}  // namespace sq

using namespace sq;

extern "C" int sq_itq_hash(const void* x, int x_dtype, int64_t n, int d, const double* mean, int mean_dtype,
                           const double* rotation, int bits, int norm_ord, uint64_t* out_codes, int mem, void* stream) {
    if (!x || !mean || !rotation || !out_codes || n <= 0 || d <= 0 || bits <= 0)
        return fail(SQ_ERR_INVALID, "sq_itq_hash: bad argument");
    if (x_dtype != SQ_DTYPE_F32 && x_dtype != SQ_DTYPE_F64) return fail(SQ_ERR_INVALID, "sq_itq_hash: unknown dtype %d", x_dtype);
    if (mean_dtype != SQ_DTYPE_F32 && mean_dtype != SQ_DTYPE_F64)
        return fail(SQ_ERR_INVALID, "sq_itq_hash: unknown mean dtype %d", mean_dtype);
    if (!itq_norm_supported(norm_ord))
        return fail(SQ_ERR_UNSUPPORTED, "sq_itq_hash: normalize code %d not supported on the device", norm_ord);
    int device = 0;
    SQ_HIP(hipGetDevice(&device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int words = (bits + 63) / 64;
    const size_t esz = x_dtype == SQ_DTYPE_F32 ? 4 : 8;
    ItqArgs a{};
    a.n = n;
    a.d = d;
    a.bits = bits;
    a.words = words;
    a.pad = words * 64 - bits;
    a.norm = norm_ord;
    a.sub32 = (x_dtype == SQ_DTYPE_F32 && mean_dtype == SQ_DTYPE_F32) ? 1 : 0;
    a.exact = g_opt.itq_exact;
    a.debug = g_opt.dense_debug;
    a.d16 = (d + 15) / 16 * 16;
    if (mem == SQ_MEM_DEVICE) {
        a.x = x;
        a.mean = mean;
        a.rot = rotation;
        a.out = reinterpret_cast<u64*>(out_codes);
        return x_dtype == SQ_DTYPE_F32 ? itq_launch<float>(a, st, device) : itq_launch<double>(a, st, device);
    }
    // Host buffers: ONE stream-ordered allocation for rows | mean | rotation | codes (the library's pool keeps up to
    // 256 MB across calls).  Four hipMalloc / hipFree pairs per call made hashing one query vector -- what every
    // LSHNearestNeighborIndex.nn does first -- cost 94 us.
    const size_t o_x = 0, o_m = align256(o_x + (size_t)n * d * esz), o_r = align256(o_m + (size_t)d * 8);
    const size_t o_out = align256(o_r + (size_t)d * bits * 8), total = o_out + (size_t)n * words * 8;
    unsigned char* base = nullptr;
    SQ_HIP(scratch_alloc(reinterpret_cast<void**>(&base), total, st, device));
    auto done = [&](int code) {
        (void)hipFreeAsync(base, st);
        return code;
    };
    if (hipMemcpyAsync(base + o_x, x, (size_t)n * d * esz, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(base + o_m, mean, (size_t)d * 8, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(base + o_r, rotation, (size_t)d * bits * 8, hipMemcpyHostToDevice, st) != hipSuccess)
        return done(fail(SQ_ERR_HIP, "sq_itq_hash: H2D copy failed"));
    a.x = base + o_x;
    a.mean = reinterpret_cast<const double*>(base + o_m);
    a.rot = reinterpret_cast<const double*>(base + o_r);
    a.out = reinterpret_cast<u64*>(base + o_out);
    const int rc = x_dtype == SQ_DTYPE_F32 ? itq_launch<float>(a, st, device) : itq_launch<double>(a, st, device);
    if (rc != SQ_OK) return done(rc);
    if (hipMemcpyAsync(out_codes, base + o_out, (size_t)n * words * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
        stream_wait(st) != hipSuccess)
        return done(fail(SQ_ERR_HIP, "sq_itq_hash: kernel or D2H copy failed: %s", hipGetErrorString(hipGetLastError())));
    return done(SQ_OK);
}

// ------------------------------------------------------------------ resident model
// ItqFunctor's model (mean, rotation) kept on the device, with pinned staging for small batches: hashing ONE query
// vector -- the first thing every LSHNearestNeighborIndex.nn does (lsh.py:473) -- otherwise uploads the 64 KB
// rotation and the mean from pageable memory on every call (95 us per query, most of it copies).
namespace sq {
struct ItqModelHandle : HandleBase {
    DevBuf mean, rot, x_dev, out_dev;
    HostPinned stage;   // [rows | codes] of one small batch
    int d = 0, bits = 0, norm = SQ_NORM_NONE, mean_dtype = SQ_DTYPE_F64;
    ~ItqModelHandle() override {
        for (DevBuf* b : {&mean, &rot, &x_dev, &out_dev}) b->release();
        stage.release();
    }
};
}  // namespace sq

extern "C" int sq_itq_model_create(const double* mean, int mean_dtype, const double* rotation, int d, int bits,
                                   int norm_ord, sq_handle_t* out) {
    if (!mean || !rotation || !out || d <= 0 || bits <= 0) return fail(SQ_ERR_INVALID, "sq_itq_model_create: bad argument");
    if (mean_dtype != SQ_DTYPE_F32 && mean_dtype != SQ_DTYPE_F64)
        return fail(SQ_ERR_INVALID, "sq_itq_model_create: unknown mean dtype %d", mean_dtype);
    if (!itq_norm_supported(norm_ord))
        return fail(SQ_ERR_UNSUPPORTED, "sq_itq_model_create: normalize code %d not supported on the device", norm_ord);
    auto* h = new ItqModelHandle();
    h->kind = H_ITQ;
    h->d = d;
    h->bits = bits;
    h->norm = norm_ord;
    h->mean_dtype = mean_dtype;
    auto bail = [&](int rc) {
        delete h;
        return rc;
    };
    if (hipGetDevice(&h->device) != hipSuccess) return bail(fail(SQ_ERR_HIP, "sq_itq_model_create: no HIP device"));
    int rc = h->mean.reserve((size_t)d * 8);
    if (rc == SQ_OK) rc = h->rot.reserve((size_t)d * bits * 8);
    if (rc != SQ_OK) return bail(rc);
    if (hipMemcpy(h->mean.p, mean, (size_t)d * 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(h->rot.p, rotation, (size_t)d * bits * 8, hipMemcpyHostToDevice) != hipSuccess)
        return bail(fail(SQ_ERR_HIP, "sq_itq_model_create: H2D copy failed"));
    *out = register_handle(h);
    return SQ_OK;
}

extern "C" int sq_itq_model_hash(sq_handle_t hid, const void* x, int x_dtype, int64_t n, uint64_t* out_codes, int mem,
                                 void* stream) {
    auto* h = static_cast<ItqModelHandle*>(lookup_handle(hid, H_ITQ));
    if (!h) return fail(SQ_ERR_INVALID, "sq_itq_model_hash: unknown handle");
    if (!x || !out_codes || n <= 0) return fail(SQ_ERR_INVALID, "sq_itq_model_hash: bad argument");
    if (x_dtype != SQ_DTYPE_F32 && x_dtype != SQ_DTYPE_F64) return fail(SQ_ERR_INVALID, "sq_itq_model_hash: unknown dtype %d", x_dtype);
    std::lock_guard<std::mutex> lock(h->mu);
    h->refresh_options();   // (a per-handle "itq_exact" / "dense_debug" wins over the process-wide value)
    SQ_HIP(hipSetDevice(h->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int words = (h->bits + 63) / 64;
    const size_t esz = x_dtype == SQ_DTYPE_F32 ? 4 : 8;
    ItqArgs a{};
    a.n = n;
    a.d = h->d;
    a.bits = h->bits;
    a.words = words;
    a.pad = words * 64 - h->bits;
    a.norm = h->norm;
    a.sub32 = (x_dtype == SQ_DTYPE_F32 && h->mean_dtype == SQ_DTYPE_F32) ? 1 : 0;
    a.exact = h->opt.itq_exact;
    a.debug = h->opt.dense_debug;
    a.d16 = (h->d + 15) / 16 * 16;
    a.mean = h->mean.as<double>();
    a.rot = h->rot.as<double>();
    if (mem == SQ_MEM_DEVICE) {
        a.x = x;
        a.out = reinterpret_cast<u64*>(out_codes);
        return x_dtype == SQ_DTYPE_F32 ? itq_launch<float>(a, st, h->device) : itq_launch<double>(a, st, h->device);
    }
    const size_t xb = (size_t)n * h->d * esz, ob = (size_t)n * words * 8;
    SQ_TRY(h->x_dev.reserve(xb));
    SQ_TRY(h->out_dev.reserve(ob));
    // small batches go through pinned staging (an asynchronous copy from pageable memory is a blocking staged copy)
    const bool staged = xb + ob <= (1u << 20);
    const void* src = x;
    void* dst = out_codes;
    if (staged) {
        SQ_TRY(h->stage.reserve(xb + ob));
        memcpy(h->stage.p, x, xb);
        src = h->stage.p;
        dst = static_cast<char*>(h->stage.p) + xb;
    }
    SQ_HIP(hipMemcpyAsync(h->x_dev.p, src, xb, hipMemcpyHostToDevice, st));
    a.x = h->x_dev.p;
    a.out = h->out_dev.as<u64>();
    SQ_TRY(x_dtype == SQ_DTYPE_F32 ? itq_launch<float>(a, st, h->device) : itq_launch<double>(a, st, h->device));
    SQ_HIP(hipMemcpyAsync(dst, h->out_dev.p, ob, hipMemcpyDeviceToHost, st));
    SQ_HIP(stream_wait(st));
    if (staged) memcpy(out_codes, dst, ob);
    return SQ_OK;
}

extern "C" int sq_itq_model_destroy(sq_handle_t hid) {
    auto* h = remove_handle(hid, H_ITQ);
    if (!h) return fail(SQ_ERR_INVALID, "sq_itq_model_destroy: unknown handle");
    (void)hipSetDevice(h->device);
    delete h;
    return SQ_OK;
}
