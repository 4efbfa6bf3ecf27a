// numpy's add-reduce order over a contiguous axis, for 8 cooperating lanes.
//
// numpy sums float arrays pairwise (numpy/_core/src/umath/loops_utils.h.src,
// @TYPE@_pairwise_sum; numpy 2.2.6 is the version pinned in this image):
//   n < 8     : serial loop starting from 0
//   n <= 128  : eight interleaved accumulators r[j] += a[i+j], combined as
//               ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7)), then the n%8 tail serially
//   otherwise : split at n/2 rounded down to a multiple of 8, recurse, add.
// metrics.euclidean_distance (smqtk_indexing/utils/metrics.py:86) and
// ItqFunctor._norm_vector (impls/lsh_functor/itq.py:185) both reduce this way,
// so reproducing the order makes float results bit identical.  The scalar
// restatement that pins this is oracle/cpu_ref.py:np_pairwise_sum_f32.
//
// An aligned group of 8 lanes calls these together; lane j8 owns accumulator
// r[j8]; every lane of the group returns the same value.  `term(i)` yields
// element i already rounded to T.  No FMA contraction anywhere.
#pragma once
#include <hip/hip_runtime.h>

namespace sq {

__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ double add_rn(double a, double b) { return __dadd_rn(a, b); }

template <class T, class F>
__device__ __forceinline__ T pw_leaf(F& term, int off, int n, int j8) {
    if (n < 8) {
        T r = (T)0;
        for (int i = 0; i < n; ++i) r = add_rn(r, term(off + i));
        return r;
    }
    T r = term(off + j8);
    const int nfull = n - (n % 8);
    for (int i = 8; i < nfull; i += 8) r = add_rn(r, term(off + i + j8));
    r = add_rn(r, __shfl_xor(r, 1));
    r = add_rn(r, __shfl_xor(r, 2));
    r = add_rn(r, __shfl_xor(r, 4));
    for (int i = nfull; i < n; ++i) r = add_rn(r, term(off + i));
    return r;
}

// numpy's recursion over the leaves without a per-thread stack array.
// `leaf(off, n)` returns the leaf sum of elements [off, off+n), n <= 128.  The
// walk is post-order; the position in the tree is the bit string `path` (bit i
// = went right at depth i) and a node's range is re-derived from the root when
// the walk moves to a right sibling (a few integer operations per 128-element
// leaf).  Pending left sums sit in a 12-deep shift register of scalars with
// static indices, so everything stays in a handful of VGPRs (an indexed stack
// of (offset, size, state, sum) cost ~100 registers or scratch and held the
// kernels that use this at one wave per SIMD).  Depth 12: n <= 128 * 2^12.
static constexpr int PW_MAX_DEPTH = 12;
static constexpr int PW_MAX_N = 128 << PW_MAX_DEPTH;

__device__ __forceinline__ int pw_split(int m) {
    int m2 = m / 2;
    return m2 - (m2 % 8);
}

// numpy's add-reduce walks an array through its iterator's buffer: NPY_BUFSIZE = 8192 elements at a time, each buffer
// summed pairwise as above and the buffers' sums added in order -- np.sum of 8300 float32 values is
// pairwise(a[:8192]) + pairwise(a[8192:]), not a pairwise split at 4144 (found with 8300-dimensional rows in round 4;
// oracle/cpu_ref.py:np_pairwise_sum_f32 restates it, tests/test_oracle_golden.py pins it against np.sum).
static constexpr int PW_BUFSIZE = 8192;

template <class T, class L>
__device__ __forceinline__ T pw_tree_buffer(L&& leaf, const int base, const int n);

template <class T, class L>
__device__ __forceinline__ T pw_tree(L&& leaf, int n) {
    if (n <= PW_BUFSIZE) return pw_tree_buffer<T>(leaf, 0, n);
    T total = pw_tree_buffer<T>(leaf, 0, PW_BUFSIZE);
    for (int s = PW_BUFSIZE; s < n; s += PW_BUFSIZE) total = add_rn(total, pw_tree_buffer<T>(leaf, s, n - s < PW_BUFSIZE ? n - s : PW_BUFSIZE));
    return total;
}

// one buffer: elements [base, base + n), n <= 8192
template <class T, class L>
__device__ __forceinline__ T pw_tree_buffer(L&& leaf0, const int base, const int n) {
    auto leaf = [&](int off, int m) { return leaf0(base + off, m); };
    if (n <= 128) return leaf(0, n);
    T st[PW_MAX_DEPTH];
#pragma unroll
    for (int i = 0; i < PW_MAX_DEPTH; ++i) st[i] = (T)0;
    unsigned path = 0;
    int depth = 0, off = 0, m = n;
    for (;;) {
        while (m > 128) {  // descend to the leftmost leaf of this subtree
            m = pw_split(m);
            path &= ~(1u << depth);
            ++depth;
        }
        T v = leaf(off, m);
        while (depth > 0 && ((path >> (depth - 1)) & 1u)) {  // a right child: combine with the pending left sum
            v = add_rn(st[0], v);
#pragma unroll
            for (int i = 0; i + 1 < PW_MAX_DEPTH; ++i) st[i] = st[i + 1];
            --depth;
        }
        if (depth == 0) return v;
#pragma unroll
        for (int i = PW_MAX_DEPTH - 1; i > 0; --i) st[i] = st[i - 1];
        st[0] = v;
        path |= 1u << (depth - 1);
        // the right sibling: the parent's range from the root, then its right part
        off = 0;
        m = n;
        for (int i = 0; i + 1 < depth; ++i) {
            const int m2 = pw_split(m);
            if ((path >> i) & 1u) {
                off += m2;
                m -= m2;
            } else {
                m = m2;
            }
        }
        const int m2 = pw_split(m);
        off += m2;
        m -= m2;
    }
}

// The same walk for G sums that share their leaves' inputs (G queries against one row): `leaf(off, n, v)`
// fills the G leaf sums, the pending left sums are G shift registers of depth D (n <= 128 * 2^D).
template <class T, int G, int D, class L>
__device__ __forceinline__ void pw_tree_multi(L&& leaf, int n, T (&out)[G]) {
    if (n <= 128) {
        leaf(0, n, out);
        return;
    }
    T st[D][G];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int g = 0; g < G; ++g) st[i][g] = (T)0;
    unsigned path = 0;
    int depth = 0, off = 0, m = n;
    for (;;) {
        while (m > 128) {
            m = pw_split(m);
            path &= ~(1u << depth);
            ++depth;
        }
        leaf(off, m, out);
        while (depth > 0 && ((path >> (depth - 1)) & 1u)) {
#pragma unroll
            for (int g = 0; g < G; ++g) out[g] = add_rn(st[0][g], out[g]);
#pragma unroll
            for (int i = 0; i + 1 < D; ++i)
#pragma unroll
                for (int g = 0; g < G; ++g) st[i][g] = st[i + 1][g];
            --depth;
        }
        if (depth == 0) return;
#pragma unroll
        for (int i = D - 1; i > 0; --i)
#pragma unroll
            for (int g = 0; g < G; ++g) st[i][g] = st[i - 1][g];
#pragma unroll
        for (int g = 0; g < G; ++g) st[0][g] = out[g];
        path |= 1u << (depth - 1);
        off = 0;
        m = n;
        for (int i = 0; i + 1 < depth; ++i) {
            const int m2 = pw_split(m);
            if ((path >> i) & 1u) {
                off += m2;
                m -= m2;
            } else {
                m = m2;
            }
        }
        const int m2 = pw_split(m);
        off += m2;
        m -= m2;
    }
}

template <class T, class F>
__device__ __forceinline__ T np_pairwise_sum(F term, int n, int j8) {
    return pw_tree<T>([&](int off, int m) { return pw_leaf<T>(term, off, m, j8); }, n);
}

}  // namespace sq
