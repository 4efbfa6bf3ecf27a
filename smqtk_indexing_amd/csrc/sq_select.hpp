// Top-k selection over per-query candidate keys, and the sample-threshold
// selector.  Shared by the Hamming and dense paths (gfx950, wave64).
//
// Keys are totally ordered and UNIQUE per query: (ordered distance bits, row
// id).  Sorting keys ascending therefore IS the canonical result order
// (distance ascending, then row id ascending), which
// restates the stable `sorted(...)[:n]` of lsh.py:513-518 and the
// `heapq.nsmallest` of linear.py:235-238 up to their set-iteration tie order.
#pragma once
#include "sq_common.hpp"

namespace sq {

// ---------------------------------------------------------------- key types
// K64: [63:32] ordered distance (f32 bits or integer), [31:0] local row id.
// K128: hi = ordered f64 distance bits, lo = local row id.
struct K128 {
    u64 hi, lo;
};

__device__ __forceinline__ u32 ordered_f32(float f) {
    u32 u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unordered_f32(u32 u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
__device__ __forceinline__ u64 ordered_f64(double f) {
    u64 u = (u64)__double_as_longlong(f);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double unordered_f64(u64 u) {
    return __longlong_as_double((long long)((u >> 63) ? (u & 0x7fffffffffffffffull) : ~u));
}

template <class K>
struct KeyOps;
template <>
struct KeyOps<u64> {
    static constexpr int NBYTES = 8;
    __device__ static __forceinline__ u64 maxv() { return ~0ull; }
    __device__ static __forceinline__ bool less(u64 a, u64 b) { return a < b; }
    __device__ static __forceinline__ bool is_max(u64 a) { return a == ~0ull; }
    // byte i, i = 0 most significant
    __device__ static __forceinline__ u32 byte(u64 a, int i) { return (u32)(a >> (8 * (7 - i))) & 255u; }
    __device__ static __forceinline__ u64 zero() { return 0ull; }
    __device__ static __forceinline__ void set_byte(u64& a, int i, u32 v) { a |= (u64)v << (8 * (7 - i)); }
    // first nb bytes equal
    __device__ static __forceinline__ bool prefix_eq(u64 a, u64 p, int nb) {
        return nb == 0 || (a >> (8 * (8 - nb))) == (p >> (8 * (8 - nb)));
    }
    // first nb bytes (1..8) of a <= those of p
    __device__ static __forceinline__ bool prefix_le(u64 a, u64 p, int nb) {
        return (a >> (8 * (8 - nb))) <= (p >> (8 * (8 - nb)));
    }
    // number of leading bytes shared by all keys in [lo, hi]
    __device__ static __forceinline__ int common_bytes(u64 lo, u64 hi) {
        const u64 x = lo ^ hi;
        return x == 0 ? 8 : (__clzll((long long)x) >> 3);
    }
};
template <>
struct KeyOps<K128> {
    static constexpr int NBYTES = 12;  // 8 bytes of hi + low 4 bytes of lo (row id < 2^32)
    __device__ static __forceinline__ K128 maxv() { return K128{~0ull, ~0ull}; }
    __device__ static __forceinline__ bool less(K128 a, K128 b) {
        return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo);
    }
    __device__ static __forceinline__ bool is_max(K128 a) { return a.hi == ~0ull && a.lo == ~0ull; }
    __device__ static __forceinline__ u32 byte(K128 a, int i) {
        return i < 8 ? (u32)(a.hi >> (8 * (7 - i))) & 255u : (u32)(a.lo >> (8 * (11 - i))) & 255u;
    }
    __device__ static __forceinline__ K128 zero() { return K128{0ull, 0ull}; }
    __device__ static __forceinline__ void set_byte(K128& a, int i, u32 v) {
        if (i < 8)
            a.hi |= (u64)v << (8 * (7 - i));
        else
            a.lo |= (u64)v << (8 * (11 - i));
    }
    __device__ static __forceinline__ bool prefix_eq(K128 a, K128 p, int nb) {
        if (nb == 0) return true;
        if (nb <= 8) return (a.hi >> (8 * (8 - nb))) == (p.hi >> (8 * (8 - nb)));
        if (a.hi != p.hi) return false;
        int r = nb - 8;  // 1..4 bytes of the low word's 32-bit id
        return ((a.lo & 0xffffffffull) >> (8 * (4 - r))) == ((p.lo & 0xffffffffull) >> (8 * (4 - r)));
    }
    __device__ static __forceinline__ bool prefix_le(K128 a, K128 p, int nb) {
        if (nb <= 8) return (a.hi >> (8 * (8 - nb))) <= (p.hi >> (8 * (8 - nb)));
        if (a.hi != p.hi) return a.hi < p.hi;
        int r = nb - 8;
        return ((a.lo & 0xffffffffull) >> (8 * (4 - r))) <= ((p.lo & 0xffffffffull) >> (8 * (4 - r)));
    }
};

// ------------------------------------------------------- LDS bitonic sort
// Sorts sk[0..P) ascending, P a power of two >= 2; all threads of the block call.
template <class K>
__device__ __forceinline__ void bitonic_sort_lds(K* sk, int P) {
    const int T = blockDim.x;
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int i = threadIdx.x; i < (P >> 1); i += T) {
                int lo = 2 * i - (i & (stride - 1));
                int hi = lo + stride;
                bool asc = (lo & size) == 0;
                K a = sk[lo], b = sk[hi];
                if (KeyOps<K>::less(b, a) == asc) {
                    sk[lo] = b;
                    sk[hi] = a;
                }
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ int pow2_ceil(int v) {
    int p = 2;
    while (p < v) p <<= 1;
    return p;
}

// --------------------------------------------------------------- select_topk
// One workgroup per query.  keys: [nq][stride] candidate keys, cnt[q] of them
// valid (clamped to cap).  Writes the min(k, M) smallest keys ascending to
// out[q][0..k) and pads the rest with the max key.
//
//   * the keys are copied to LDS when they fit (M <= lds_keys);
//   * an MSB-first radix select (8-bit digits, LDS histogram) finds the k-th
//     smallest key; leading bytes shared by every key are skipped (u64 keys)
//     and the walk stops early once a digit bin holds exactly the keys still
//     needed;
//   * the selected keys (exactly k: keys are unique) are gathered and
//     bitonic-sorted in LDS.
//   Small inputs (M <= 512) or k beyond the sort buffer are bitonic-sorted whole.
// Requires k <= lds_keys.  Dynamic LDS: (lds_keys + SELECT_SORT_MAX) * sizeof(K).
// `post(q, out_q, k)` runs in the same workgroup once out[q][0..k) is written (all threads call it):
// the dense path converts keys to (distance, id) and certifies the query there instead of in a
// separate launch.
static constexpr int SELECT_SORT_MAX = 2048;

struct SelectNoPost {
    template <class K>
    __device__ __forceinline__ void operator()(int, const K*, int) const {}
};

template <class K, class Post = SelectNoPost>
__global__ __launch_bounds__(1024) void select_topk_kernel(const K* __restrict__ keys,
                                                            const u32* __restrict__ cnt, u32 cap,
                                                            long long stride, int k, int lds_keys,
                                                            K* __restrict__ out, Post post = Post(), int cnt_shift = 0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    K* sk = reinterpret_cast<K*>(smem_raw);
    K* so = sk + lds_keys;
    __shared__ u32 hist[256];
    __shared__ u32 sh_sel, sh_rem, sh_n, sh_stop;
    __shared__ u64 sh_red[32];
    const int q = blockIdx.x;
    const int T = blockDim.x;
    const u32 craw = cnt[(long long)q << cnt_shift];   // (cnt_shift: counters a cache line apart, see I8_CNT_SHIFT)
    const int M = (int)(craw < cap ? craw : cap);
    const K* src = keys + (long long)q * stride;
    K* dst = out + (long long)q * k;
    const int kk = k < M ? k : M;
    const bool in_lds = M <= lds_keys;
    if (in_lds)
        for (int i = threadIdx.x; i < M; i += T) sk[i] = src[i];
    __syncthreads();
    K* sorted = sk;
    int nsort = M;
    const bool whole = in_lds && (M <= 512 || kk > SELECT_SORT_MAX || kk == M);
    if (!whole) {
        const K* rd = in_lds ? sk : src;
        K* gather = in_lds ? so : sk;
        const int gcap = in_lds ? SELECT_SORT_MAX : lds_keys;
        int b0 = 0;
        if constexpr (sizeof(K) == 8) {
            // leading bytes common to every key need no pass
            u64 lo = ~0ull, hi = 0ull;
            for (int i = threadIdx.x; i < M; i += T) {
                const u64 v = rd[i];
                lo = v < lo ? v : lo;
                hi = v > hi ? v : hi;
            }
            for (int o = 32; o > 0; o >>= 1) {
                const u64 l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
                lo = l2 < lo ? l2 : lo;
                hi = h2 > hi ? h2 : hi;
            }
            const int wv = threadIdx.x >> 6, nw = T >> 6;
            if ((threadIdx.x & 63) == 0) {
                sh_red[wv] = lo;
                sh_red[16 + wv] = hi;
            }
            __syncthreads();
            lo = ~0ull;
            hi = 0ull;
            for (int w = 0; w < nw; ++w) {
                lo = sh_red[w] < lo ? sh_red[w] : lo;
                hi = sh_red[16 + w] > hi ? sh_red[16 + w] : hi;
            }
            b0 = KeyOps<K>::common_bytes(lo, hi);
            if (b0 > 7) b0 = 7;
            __syncthreads();
        }
        K prefix = KeyOps<K>::zero();
        if constexpr (sizeof(K) == 8) {
            if (b0 > 0) prefix = rd[0] & ~((~0ull) >> (8 * b0));
        }
        if (threadIdx.x == 0) {
            sh_rem = (u32)kk;
            sh_stop = 0;
        }
        int nb_done = b0;
        for (int b = b0; b < KeyOps<K>::NBYTES; ++b) {
            for (int i = threadIdx.x; i < 256; i += T) hist[i] = 0;
            __syncthreads();
            for (int i = threadIdx.x; i < M; i += T) {
                const K key = rd[i];
                if (KeyOps<K>::prefix_eq(key, prefix, b)) atomicAdd(&hist[KeyOps<K>::byte(key, b)], 1u);
            }
            __syncthreads();
            if (threadIdx.x < 64) {
                // first bin whose inclusive running count reaches `need` (bin 255 if none does): wave 0
                // scans the 256 bins, four per lane -- one thread walking them costs ~10 us per digit
                const int l = threadIdx.x;
                const u32 need = sh_rem;
                const u32 h0 = hist[4 * l], h1 = hist[4 * l + 1], h2 = hist[4 * l + 2], h3 = hist[4 * l + 3];
                const u32 s4 = h0 + h1 + h2 + h3;
                u32 inc = s4;
                for (int o = 1; o < 64; o <<= 1) {
                    const u32 t = __shfl_up(inc, o);
                    if (l >= o) inc += t;
                }
                const u32 exc = inc - s4;
                const u32 total = __shfl(inc, 63);
                const bool none = total < need;                       // cannot happen for a consistent count
                const bool mine = none ? l == 63 : (exc < need && need <= inc);
                if (mine) {
                    u32 c = exc, v = 4u * l, hv = h0;
                    if (c + h0 < need) { c += h0; v++; hv = h1;
                        if (c + h1 < need) { c += h1; v++; hv = h2;
                            if (c + h2 < need) { c += h2; v++; hv = h3; } } }
                    sh_sel = v;
                    sh_rem = need - c;
                    sh_stop = (hv == need - c) ? 1u : 0u;  // the whole bin is wanted: no finer digit needed
                }
            }
            __syncthreads();
            KeyOps<K>::set_byte(prefix, b, sh_sel);
            nb_done = b + 1;
            const bool stop = sh_stop != 0;
            __syncthreads();
            if (stop) break;
        }
        // keys whose first nb_done bytes are <= the prefix's are exactly the kk smallest
        if (threadIdx.x == 0) sh_n = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < M; i += T) {
            const K key = rd[i];
            if (KeyOps<K>::prefix_le(key, prefix, nb_done)) {
                const u32 pos = atomicAdd(&sh_n, 1u);
                if ((int)pos < gcap) gather[pos] = key;
            }
        }
        __syncthreads();
        nsort = (int)sh_n < gcap ? (int)sh_n : gcap;
        sorted = gather;
    }
    if (nsort > 1) {
        const int P = pow2_ceil(nsort);
        for (int i = nsort + threadIdx.x; i < P; i += T) sorted[i] = KeyOps<K>::maxv();
        bitonic_sort_lds<K>(sorted, P);
    } else {
        __syncthreads();
    }
    for (int i = threadIdx.x; i < k; i += T) dst[i] = (i < kk && i < nsort) ? sorted[i] : KeyOps<K>::maxv();
    __syncthreads();  // the block's own global writes are visible to all its threads
    post(q, dst, k);
}

// ------------------------------------------------------- any-k sorted select
// The one-workgroup select above needs k <= its LDS capacity (16384 64-bit keys, 7168 128-bit keys).  The reference
// has no such limit (linear.py:235-238 and lsh.py:513-518 slice whatever n is asked), so larger k takes this path:
// every query's list (cnt[q] keys, clamped to cap, the rest padding) is sorted completely -- an LDS bitonic sort of
// chunks, then log2(P / chunk) passes of merge-by-ranking (every key finds its place in the merged run with one
// binary search in the sibling run: fully parallel, no data-dependent control flow between threads), ping-pong
// between two scratch buffers of nq * P keys -- and the first k keys are the answer.  O(P log P) per query with
// P = cap rounded up to a power of two: a rare path, built to be correct and simple rather than fast.
template <class K>
struct SortLarge {
    static constexpr int CH = sizeof(K) == 8 ? 4096 : 2048;  // keys per LDS chunk (32 KB)
};

template <class K>
__global__ __launch_bounds__(1024) void sortl_chunk_kernel(const K* __restrict__ keys, const u32* __restrict__ cnt, u32 cap,
                                                            long long stride, K* __restrict__ dst, long long P) {
    constexpr int CH = SortLarge<K>::CH;
    __shared__ K sk[CH];
    const int q = blockIdx.y;
    const long long base = (long long)blockIdx.x * CH;
    const u32 craw = cnt[q];
    const long long M = craw < cap ? craw : cap;
    for (int i = threadIdx.x; i < CH; i += blockDim.x)
        sk[i] = base + i < M ? keys[(long long)q * stride + base + i] : KeyOps<K>::maxv();
    bitonic_sort_lds<K>(sk, CH);
    for (int i = threadIdx.x; i < CH; i += blockDim.x) dst[(long long)q * P + base + i] = sk[i];
}

// runs of `run` sorted keys -> runs of 2 * run.  Real keys are unique; padding keys (all equal, the maximum) keep
// distinct places because the left run counts the right keys BELOW and the right run the left keys AT OR BELOW.
template <class K>
__global__ __launch_bounds__(256) void sortl_merge_kernel(const K* __restrict__ src, K* __restrict__ dst, long long P,
                                                           long long run) {
    const int q = blockIdx.y;
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= P) return;
    const long long pair = e / (2 * run), within = e - pair * 2 * run;
    const K* a = src + (long long)q * P + pair * 2 * run;
    const bool left = within < run;
    const long long i = left ? within : within - run;
    const K key = a[within];
    const K* other = left ? a + run : a;
    long long lo = 0, hi = run;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        const bool go_right = left ? KeyOps<K>::less(other[mid], key) : !KeyOps<K>::less(key, other[mid]);
        if (go_right) lo = mid + 1; else hi = mid;
    }
    dst[(long long)q * P + pair * 2 * run + i + lo] = key;
}

template <class K, class Post>
__global__ __launch_bounds__(1024) void sortl_finish_kernel(const K* __restrict__ sorted, long long P, int k,
                                                             K* __restrict__ out, Post post) {
    const int q = blockIdx.x;
    K* dst = out + (long long)q * k;
    for (int i = threadIdx.x; i < k; i += blockDim.x) dst[i] = i < P ? sorted[(long long)q * P + i] : KeyOps<K>::maxv();
    __syncthreads();
    post(q, dst, k);
}

// keys: [nq][stride], cnt[q] valid (clamped to cap); out: [nq][k]; scratch grows to 2 * nq * P keys.
template <class K, class Post>
static int sort_select_large(const K* keys, const u32* cnt, u32 cap, long long stride, int k, int nq, K* out,
                             DevBuf& scratch, const Post& post, hipStream_t st) {
    constexpr int CH = SortLarge<K>::CH;
    long long P = CH;
    while (P < (long long)cap) P <<= 1;
    SQ_TRY(scratch.reserve((size_t)2 * nq * P * sizeof(K)));
    K* a = scratch.as<K>();
    K* b = a + (size_t)nq * P;
    hipLaunchKernelGGL((sortl_chunk_kernel<K>), dim3((unsigned)(P / CH), (unsigned)nq), dim3(1024), 0, st, keys, cnt, cap, stride,
                       a, P);
    for (long long run = CH; run < P; run <<= 1) {
        hipLaunchKernelGGL((sortl_merge_kernel<K>), dim3((unsigned)((P + 255) / 256), (unsigned)nq), dim3(256), 0, st, a, b, P, run);
        K* t = a;
        a = b;
        b = t;
    }
    hipLaunchKernelGGL((sortl_finish_kernel<K, Post>), dim3((unsigned)nq), dim3(1024), 0, st, a, P, k, out, post);
    SQ_HIP(hipGetLastError());
    return SQ_OK;
}

// ------------------------------------------------------------ block helpers
__device__ __forceinline__ float wave_min(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ u32 wave_sum(u32 v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ------------------------------------------------------- kth_threshold_f32
// Per query (one 1024-thread workgroup): an element T of scores[q][0..ns)
// whose rank r among the finite scores satisfies k <= r (and r <= ~2k unless
// the data has that many duplicates).  T upper-bounds the k-th smallest score
// of ANY superset of the sample, which is all the scan needs (DESIGN.md
// "threshold").  +inf entries mark padding rows and are ignored; when fewer
// than k finite scores exist T = +inf (scan emits everything).
// Method: iterated linear bucketing (2048 buckets over [lo,hi]); bucket index
// is a monotone function of the score, so everything in lower buckets is
// strictly smaller than everything in the selected bucket.
// Fast path (k <= 1024): the k-th smallest of the 1024 per-thread minima is an
// upper bound T0 of the k-th smallest sample (each minimum is a sample); the
// samples <= T0 (about k of them) are collected in LDS and sorted, and T is the
// exact k-th smallest sample: two read passes and two small bitonic sorts.
// `post(q, T)` maps the result before it is stored (the dense path adds its
// slack there, sq_dense_exact.hpp: DenseThrPost).
struct KthIdentity {
    __device__ __forceinline__ float operator()(int, float t) const { return t; }
    __device__ __forceinline__ void prologue(int, double*) const {}
};

__device__ __forceinline__ void bitonic_sort_f32_lds(float* sk, int P) {
    const int T = blockDim.x;
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int i = threadIdx.x; i < (P >> 1); i += T) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool asc = (lo & size) == 0;
                const float a = sk[lo], b = sk[hi];
                if ((b < a) == asc) {
                    sk[lo] = b;
                    sk[hi] = a;
                }
            }
        }
    }
    __syncthreads();
}

template <class Post>
static __global__ __launch_bounds__(1024) void kth_threshold_f32_kernel(const float* __restrict__ scores,
                                                                  long long ns, int k,
                                                                  float* __restrict__ thr, Post post) {
    constexpr int NB = 2048;
    constexpr int LCAP = 4096;
    __shared__ u32 hist[NB];
    __shared__ float flist[LCAP];
    __shared__ float red_a[16], red_b[16];
    __shared__ u32 red_c[16];
    __shared__ u32 sh_b, sh_below, sh_cin, sh_cnt;
    __shared__ float sh_t0;
    __shared__ double pre_red[16];
    const int q = blockIdx.x;
    const int T = blockDim.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = T >> 6;
    const float* s = scores + (long long)q * ns;
    const float INF = __builtin_inff();
    post.prologue(q, pre_red);  // (what a caller needs done once per query before the threshold is used: see DenseThrPost)

    if (k <= T && T <= LCAP) {
        constexpr int U = 8;      // loads in flight per thread when streaming
        constexpr int VMAX = 40;  // samples per thread kept in registers (ns <= 40960: one load latency, no re-read)
        const bool in_regs = ns <= (long long)VMAX * T;
        float vr[VMAX];
        float mymin = INF;
        if (in_regs) {
#pragma unroll
            for (int u = 0; u < VMAX; ++u) {
                const long long i = (long long)u * T + threadIdx.x;
                vr[u] = i < ns ? s[i] : INF;
            }
#pragma unroll
            for (int u = 0; u < VMAX; ++u) mymin = fminf(mymin, vr[u]);
        } else {
            for (long long base = threadIdx.x; base < ns; base += (long long)T * U) {
                float v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const long long i = base + (long long)u * T;
                    v[u] = i < ns ? s[i] : INF;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) mymin = fminf(mymin, v[u]);
            }
        }
        // An upper bound t0 of the k-th smallest sample from the T thread minima: every wave sorts
        // its 64 minima in registers (shuffle network, no barrier); with p = ceil(k / waves), each
        // wave has p minima <= its p-th smallest, so at least k samples are <= the largest of the
        // waves' p-th smallest minima.
        {
            float v = mymin;
            for (int kk2 = 2; kk2 <= 64; kk2 <<= 1)
                for (int j = kk2 >> 1; j > 0; j >>= 1) {
                    const float o = __shfl_xor(v, j);
                    const bool up = (lane & kk2) == 0, lower = (lane & j) == 0;
                    v = (lower == up) ? fminf(v, o) : fmaxf(v, o);
                }
            const int p = (k + nw - 1) / nw;  // <= 64 because k <= T = 64 nw
            if (lane == p - 1) red_a[wv] = v;
            if (threadIdx.x == 0) sh_cnt = 0;
            __syncthreads();
            float m = -INF;
            for (int w = 0; w < nw; ++w) m = fmaxf(m, red_a[w]);
            if (threadIdx.x == 0) sh_t0 = m;
            __syncthreads();
        }
        const float t0 = sh_t0;
        __syncthreads();
        if (t0 < INF) {
            if (in_regs) {
#pragma unroll
                for (int u = 0; u < VMAX; ++u) {
                    if (vr[u] <= t0) {
                        const u32 pos = atomicAdd(&sh_cnt, 1u);
                        if (pos < (u32)LCAP) flist[pos] = vr[u];
                    }
                }
            } else {
                for (long long base = threadIdx.x; base < ns; base += (long long)T * U) {
                    float v[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const long long i = base + (long long)u * T;
                        v[u] = i < ns ? s[i] : INF;
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (v[u] <= t0) {
                            const u32 pos = atomicAdd(&sh_cnt, 1u);
                            if (pos < (u32)LCAP) flist[pos] = v[u];
                        }
                    }
                }
            }
            __syncthreads();
            const u32 c = sh_cnt;  // >= k: at least k thread minima are <= t0
            if (c <= (u32)T) {
                // rank by counting (about k elements; the reads are LDS broadcasts)
                if (threadIdx.x < c) {
                    const float v = flist[threadIdx.x];
                    u32 rank = 0;
                    for (u32 j = 0; j < c; ++j) {
                        const float e = flist[j];
                        rank += (e < v || (e == v && j < threadIdx.x)) ? 1u : 0u;
                    }
                    if (rank == (u32)(k - 1)) thr[q] = post(q, v);
                }
                return;
            }
            if (c <= (u32)LCAP) {
                const int P = pow2_ceil((int)c);
                for (int i = (int)c + threadIdx.x; i < P; i += T) flist[i] = INF;
                bitonic_sort_f32_lds(flist, P);
                if (threadIdx.x == 0) thr[q] = post(q, flist[k - 1]);
                return;
            }
            __syncthreads();  // mass duplicates at t0: the general method below
        }
    }

    // pass 0: min / max / count of finite scores
    float vmin = INF, vmax = -INF;
    u32 nfin = 0;
    for (long long i = threadIdx.x; i < ns; i += T) {
        float v = s[i];
        if (v < INF) {
            vmin = fminf(vmin, v);
            vmax = fmaxf(vmax, v);
            ++nfin;
        }
    }
    vmin = wave_min(vmin);
    vmax = wave_max(vmax);
    nfin = wave_sum(nfin);
    if (lane == 0) {
        red_a[wv] = vmin;
        red_b[wv] = vmax;
        red_c[wv] = nfin;
    }
    __syncthreads();
    vmin = INF;
    vmax = -INF;
    nfin = 0;
    for (int w = 0; w < nw; ++w) {
        vmin = fminf(vmin, red_a[w]);
        vmax = fmaxf(vmax, red_b[w]);
        nfin += red_c[w];
    }
    __syncthreads();
    if (nfin < (u32)k) {
        if (threadIdx.x == 0) thr[q] = post(q, INF);
        return;
    }
    float lo = vmin, hi = vmax, result = vmax;
    u32 base = 0;
    for (int it = 0; it < 8; ++it) {
        if (!(lo < hi)) {
            result = hi;
            break;
        }
        float scale = (float)NB / (hi - lo);
        if (!(scale > 0.f) || !(scale < INF)) {
            result = hi;
            break;
        }
        for (int i = threadIdx.x; i < NB; i += T) hist[i] = 0;
        __syncthreads();
        for (long long i = threadIdx.x; i < ns; i += T) {
            float v = s[i];
            if (v >= lo && v <= hi) {
                int b = (int)((v - lo) * scale);
                b = b > NB - 1 ? NB - 1 : b;
                atomicAdd(&hist[b], 1u);
            }
        }
        __syncthreads();
        if (wv == 0) {  // wave 0 finds the bucket holding rank k
            const u32 need = (u32)k - base;
            u32 mysum = 0;
            for (int j = 0; j < NB / 64; ++j) mysum += hist[lane * (NB / 64) + j];
            u32 incl = mysum;
            for (int o = 1; o < 64; o <<= 1) {
                u32 t = __shfl_up(incl, o);
                if (lane >= o) incl += t;
            }
            u64 m = __ballot(incl >= need);
            int L = __ffsll((long long)m) - 1;  // nfin >= k guarantees m != 0
            if (lane == L) {
                u32 c = incl - mysum;
                int b = lane * (NB / 64);
                for (int j = 0; j < NB / 64; ++j, ++b) {
                    if (c + hist[b] >= need) break;
                    c += hist[b];
                }
                sh_b = (u32)b;
                sh_below = base + c;
                sh_cin = hist[b];
            }
        }
        __syncthreads();
        const int bsel = (int)sh_b;
        const u32 below = sh_below, cin = sh_cin;
        // min / max of the selected bucket
        float bmin = INF, bmax = -INF;
        for (long long i = threadIdx.x; i < ns; i += T) {
            float v = s[i];
            if (v >= lo && v <= hi) {
                int b = (int)((v - lo) * scale);
                b = b > NB - 1 ? NB - 1 : b;
                if (b == bsel) {
                    bmin = fminf(bmin, v);
                    bmax = fmaxf(bmax, v);
                }
            }
        }
        bmin = wave_min(bmin);
        bmax = wave_max(bmax);
        if (lane == 0) {
            red_a[wv] = bmin;
            red_b[wv] = bmax;
        }
        __syncthreads();
        bmin = INF;
        bmax = -INF;
        for (int w = 0; w < nw; ++w) {
            bmin = fminf(bmin, red_a[w]);
            bmax = fmaxf(bmax, red_b[w]);
        }
        __syncthreads();
        result = bmax;
        if (below + cin <= 2u * (u32)k || it == 7) break;
        base = below;
        lo = bmin;
        hi = bmax;
    }
    if (threadIdx.x == 0) thr[q] = post(q, result);
}

// ------------------------------------------------------------------ fills
static __global__ void fill_u32_kernel(u32* p, long long n, u32 v) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
}  // namespace sq
