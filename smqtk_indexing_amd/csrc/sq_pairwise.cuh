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

template <class T, class F>
__device__ T np_pairwise_sum(F term, int n, int j8) {
    if (n <= 128) return pw_leaf<T>(term, 0, n, j8);
    // explicit-stack post-order walk of the recursion (depth <= log2(n/64))
    int s_off[24], s_n[24], s_state[24];
    T s_left[24];
    int sp = 1;
    s_off[0] = 0;
    s_n[0] = n;
    s_state[0] = 0;
    T ret = (T)0;
    while (sp > 0) {
        const int top = sp - 1;
        const int off = s_off[top], m = s_n[top];
        if (m <= 128) {
            ret = pw_leaf<T>(term, off, m, j8);
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
            ret = add_rn(s_left[top], ret);
            --sp;
        }
    }
    return ret;
}

}  // namespace sq
