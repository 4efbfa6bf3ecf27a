// Small-batch Hamming top-k in three launches (gfx950): head -> stream -> pick.      (included by sq_hamming.hip)
//
// The reference loop is LinearHashIndex._nn (smqtk_indexing/impls/hash_index/linear.py:235-240): heapq.nsmallest over one
// metrics.hamming_distance call per stored code (utils/metrics.py:140-155).  The general path of sq_hamming.hip answers a
// call with a memset and five kernels (sampled histogram, threshold, stream, compaction, select).  At BASELINE config 3
// (10 M x 64-bit codes, 80 MB) the stream of a one-query call is a third of the call and the rest is launch boundaries and a
// radix select over ~8 k candidates.  Distances are small integers, so the select needs no radix passes:
//
//   sample hamming_hist_kernel<W, C>    block-sampled distance histogram, unchanged (its buffer is zero between calls:
//                                       the pick kernel wipes it -- no memset)
//   stream hamming_body_kernel<W, C>    every workgroup turns the histogram into the per-query thresholds in its prologue,
//                                       under its first chunk's loads (no threshold launch; a last-workgroup-done tail in
//                                       the sample kernel was measured first: +8 us of ticket / atomic-load latency);
//                                       queries in a VGPR's lanes, v_readlane broadcasts; survivors into per-(workgroup,
//                                       query) mini-lists as in hamming_stream_kernel
//   pick   hamming_pick_kernel          one workgroup per query: the mini-lists read once into registers (the compaction,
//                                       without a compacted list), an exact distance histogram of the survivors -> the k-th
//                                       distance T, a gather of the entries with distance <= T (k .. k + ties of them
//                                       instead of thousands), one LDS rank sort, results and status words
//
// Integer exact: the pick kernel selects by (distance, caller row) exactly as the select it replaces.  A query whose
// entries at distance <= T outnumber the LDS sort buffer (huge tie groups) is flagged (status bit 3) and redone by the
// general compaction + select from the same mini-lists when the call is resolved.
#pragma once

namespace sq {

static constexpr int HF_MAX_NQ = 32;      // queries per fused call (LDS histogram of the head: nq x (bits + 1) counters)
static constexpr int HF_SORT_CAP = 4096;  // keys the pick kernel sorts in LDS (k + the tie group at the k-th distance)
static constexpr int HF_MAX_BINS = 1025;  // code widths up to 1024 bits

// wave-level inclusive scan
__device__ __forceinline__ u32 hf_wave_incl(u32 v, int lane) {
    for (int o = 1; o < 64; o <<= 1) {
        const u32 t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}

// The sample histogram for batches of 9 .. 32 queries.  hamming_hist_kernel gives every lane a code and walks the queries:
// the 64 lanes of an LDS atomic then hit the ~16 distance bins of ONE query's row -- 4-8 lanes per address, serialised --
// and at 32 queries the kernel is those atomics (11 us for 156 k sampled codes).  Here a lane owns a QUERY (lane mod 32;
// the two half-waves take two codes at a time, broadcast from an LDS copy of the block's codes): the lanes of an atomic
// hit 32 different rows, at most two lanes per address.  Same sample (every block_step-th block of 256 * C codes), same
// counts, same output as hamming_hist_kernel.  hist: [nq][bits + 1].
template <int W, int C>
__global__ __launch_bounds__(256) void hamming_hist_by_query_kernel(const u64* __restrict__ codes, long long n,
                                                                     const u64* __restrict__ qs, int nq, int bits,
                                                                     u32* __restrict__ hist, int block_step) {
    __shared__ u64 lc[256 * C * W];
    extern __shared__ u32 lh[];   // [32][bits + 1]
    const int nb = bits + 1;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < 32 * nb; i += 256) lh[i] = 0;
    const long long bb = ((long long)blockIdx.x * block_step) * 256 * C;
    const long long left = n - bb;
    const int here = left >= 256ll * C ? 256 * C : (left > 0 ? (int)left : 0);   // codes of this block
    for (int i = tid; i < here * W; i += 256) lc[i] = codes[bb * W + i];
    const int ql = lane & 31, half = lane >> 5;
    u64 qw[W];
#pragma unroll
    for (int w = 0; w < W; ++w) qw[w] = qs[(long long)(ql < nq ? ql : nq - 1) * W + w];
    __syncthreads();
    // wave wv takes codes wv * 2 + half, + 8, ...
    u32* row = lh + ql * nb;
    // (eight codes per round: the LDS reads of a round are in flight together -- one code per round is a chain of
    // read -> popcount -> atomic latencies, 15 us for the same sample)
    for (int i0 = wv * 2 + half; i0 < here; i0 += 64) {
        u64 cw[8][W];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + 8 * u;
#pragma unroll
            for (int w = 0; w < W; ++w) cw[u][w] = i < here ? lc[i * W + w] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            u32 dist = 0;
#pragma unroll
            for (int w = 0; w < W; ++w) dist = popc64_acc(cw[u][w] ^ qw[w], dist);
            if (ql < nq && i0 + 8 * u < here) atomicAdd(&row[dist], 1u);
        }
    }
    __syncthreads();
    for (int i = tid; i < nq * nb; i += 256)
        if (lh[i]) atomicAdd(&hist[i], lh[i]);
}

// Threshold of one query from its sampled histogram (hamming_thr_kernel's rule: the smallest t whose cumulative sample
// count reaches k; `bits` when the sample holds fewer than k codes), computed by ONE WAVE: lane l owns bins
// [l * per, (l + 1) * per), per = ceil((bits + 1) / 64) <= HF_PER.  Every lane returns the threshold.
static constexpr int HF_PER = 5;   // 256-bit codes: 257 bins
__device__ __forceinline__ int hf_wave_threshold(const u32* __restrict__ h, int nb, int bits, int kk, int lane) {
    const int per = (nb + 63) / 64;
    u32 v[HF_PER], sum = 0;
#pragma unroll
    for (int j = 0; j < HF_PER; ++j) {
        const int b = lane * per + j;
        v[j] = (j < per && b < nb) ? h[b] : 0u;
        sum += v[j];
    }
    const u32 inc = hf_wave_incl(sum, lane), exc = inc - sum;
    int t = bits;
    if (exc < (u32)kk && (u32)kk <= inc) {
        u32 cum = exc;
        bool found = false;
#pragma unroll
        for (int j = 0; j < HF_PER; ++j) {
            cum += v[j];
            if (!found && cum >= (u32)kk) {
                t = lane * per + j;
                found = true;
            }
        }
    }
    // exactly one lane found a bin (or none: the sample is short and every lane holds `bits`): the minimum is the answer
    for (int o = 32; o > 0; o >>= 1) {
        const int u = __shfl_xor(t, o);
        t = u < t ? u : t;
    }
    return t;
}

__device__ __forceinline__ u64 hf_readlane64(u64 v, int l) {
    const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)v, l), hi = (u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), l);
    return ((u64)hi << 32) | lo;
}

// A survivor's key to its mini-list slot, issued from inline asm: with a compiler-visible store (or the rank-table load of
// orig_row) inside the query loop hipcc puts `s_waitcnt vmcnt(0)` at the loop's head, which waits for the NEXT chunk's
// loads before the first query of this one is compared -- the prefetch then overlaps nothing.  The store needs no wait of
// its own (its registers are read at issue; the wave's stores drain before it ends) and an uncounted older operation only
// makes the compiler's counted waits stricter.  Keys therefore carry the PHYSICAL row; the pick kernel maps it.
__device__ __forceinline__ void hf_store_key(u64* p, u64 v) {
    asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(p), "v"(v) : "memory");
}

// The stream of a fused call: hamming_stream_kernel's persistent workgroups, chunk prefetch and mini-lists, with
//   * the thresholds computed in the prologue from the sampled histogram (one wave per query, under the first chunk's
//     loads) -- no threshold launch; workgroup 0 also leaves them in thr_out for the pick kernel;
//   * query words and thresholds held across the lanes of a VGPR (lane l: query l) and broadcast per query with
//     v_readlane -- no LDS reads and no scalar-memory waits in the loop;
//   * the popcounts of a code chained through v_bcnt's accumulate operand (popc64_acc).
// Batches of up to 32 queries (HF_MAX_NQ) hand over `hist`, the sample histogram [nq][bits + 1] of hamming_hist_kernel, and
// get their thresholds in the prologue; larger ones (hist == nullptr) read them from `thr_out`, written by hamming_thr_kernel,
// and walk the queries in groups of 64 (a lane per query of the group).  Dynamic LDS: nq * (8 W + 8) bytes.
template <int W, int C>
__global__ __launch_bounds__(256) void hamming_body_kernel(const u64* __restrict__ codes, long long n, RowPerm pmul,
                                                            const u64* __restrict__ qs, int nq, const u32* __restrict__ hist,
                                                            int bits, int rank, int* __restrict__ thr_out,
                                                            u64* __restrict__ seg, u32* __restrict__ bcnt, u32 S) {
    static_assert(W * 64 + 1 <= HF_PER * 64, "one wave per histogram row");
    extern __shared__ __attribute__((aligned(16))) unsigned char hbm[];
    u64* lq = reinterpret_cast<u64*>(hbm);                      // [nq][W]
    int* lthr = reinterpret_cast<int*>(lq + (size_t)nq * W);    // [nq]
    u32* lcnt = reinterpret_cast<u32*>(lthr + nq);              // [nq]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long per_chunk = 256ll * C;
    const long long nchunks = (n + per_chunk - 1) / per_chunk;
    u64 cn[C][W];
    bool validn[C];
    if ((long long)blockIdx.x < nchunks) load_codes<W, C>(codes, n, (long long)blockIdx.x * per_chunk, tid, cn, validn);
    if (hist) {
        for (int q = wv; q < nq; q += 4) {
            const int t = hf_wave_threshold(hist + (long long)q * (bits + 1), bits + 1, bits, rank, lane);
            if (lane == 0) {
                lthr[q] = t;
                if (blockIdx.x == 0) thr_out[q] = t;
            }
        }
    } else {
        for (int i = tid; i < nq; i += 256) lthr[i] = thr_out[i];
    }
    for (int i = tid; i < nq; i += 256) lcnt[i] = 0u;
    // (the queries reach the lanes through LDS: loaded from global memory straight into the registers the loop reads with
    // v_readlane, hipcc put `s_waitcnt vmcnt(0)` at the head of the query loop -- the prefetch of the next chunk then
    // overlapped nothing)
    for (int i = tid; i < nq * W; i += 256) lq[i] = qs[i];
    __syncthreads();
    u64* myseg = seg + (long long)blockIdx.x * nq * S;
    for (long long chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const long long bb = chunk * per_chunk;
        u64 c[C][W];
        bool valid[C];
#pragma unroll
        for (int i = 0; i < C; ++i) {
            valid[i] = validn[i];
#pragma unroll
            for (int w = 0; w < W; ++w) c[i][w] = cn[i][w];
        }
        // the next chunk's codes travel while this one is compared with every query
        if (chunk + gridDim.x < nchunks) load_codes<W, C>(codes, n, (chunk + gridDim.x) * per_chunk, tid, cn, validn);
        for (int g0 = 0; g0 < nq; g0 += 64) {
            const int gn = nq - g0 < 64 ? nq - g0 : 64;
            // lane l of every wave holds query g0 + l (W words) and its threshold
            u64 qv[W];
            const int ql = g0 + (lane < gn ? lane : gn - 1);
#pragma unroll
            for (int w = 0; w < W; ++w) qv[w] = lq[(size_t)ql * W + w];
            const int tv = lane < gn ? lthr[ql] : -1;
            for (int qi = 0; qi < gn; ++qi) {
                const int q = g0 + qi;
                u64 qw[W];
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    // the query word as a VECTOR operand (all lanes equal): v_xor_b32 with a scalar source issues at half the
                    // rate of the all-VGPR form (tools/micro/valu_rate.hip: 4 against 2-2.5 cycles per wave-instruction and SIMD)
                    u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)qv[w], qi), hi = (u32)__builtin_amdgcn_readlane((int)(u32)(qv[w] >> 32), qi);
                    asm volatile("" : "+v"(lo), "+v"(hi));
                    qw[w] = ((u64)hi << 32) | lo;
                }
                const int t = __builtin_amdgcn_readlane(tv, qi);
                u32 dist[C];
                u32 m = 0xffffffffu;
#pragma unroll
                for (int i = 0; i < C; ++i) {
                    u32 dsum = 0;
#pragma unroll
                    for (int w = 0; w < W; ++w) dsum = popc64_acc(c[i][w] ^ qw[w], dsum);
                    dist[i] = dsum;
                    m = dsum < m ? dsum : m;
                }
                if ((int)m <= t) {  // rare: one branch per (query, C codes); padding codes are screened inside
#pragma unroll
                    for (int i = 0; i < C; ++i) {
                        if (valid[i] && (int)dist[i] <= t) {
                            const u32 pos = atomicAdd(&lcnt[q], 1u);
                            if (pos < S) hf_store_key(myseg + (long long)q * S + pos, ((u64)dist[i] << 32) | (u64)(u32)code_row<W, C>(bb, tid, i));
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < nq; i += 256) bcnt[(long long)blockIdx.x * nq + i] = lcnt[i];
}

// One workgroup (1024 threads) per query over the mini-lists the stream left: seg[(g * nq + q) * S + e], fills
// bcnt[g * nq + q], g < G <= 2048; thr[q] = the stream's threshold (every entry's distance is <= it).
// A prefix sum over the fills (the compaction, without a compacted list) numbers the M entries; thread t takes entries
// [8 (1024 b + t), + 8) of batch b -- one binary search in the offsets, then a walk -- so the work is even whatever the
// lists' lengths, all of a batch's loads are in flight together, and with M <= 8192 (one batch: the usual case, ~8 k
// candidates) the entries stay in registers between the two passes.  Pass 1: the exact histogram of the entries'
// distances -> T, the k-th smallest distance.  Pass 2: the entries at distance <= T (k .. k + the tie group at T) are
// gathered and sorted.  On the way out the workgroup wipes its row of the sampled histogram (zero between calls).
// status bit0 = a mini-list overflowed or more candidates than `cap` (exact path), bit2 = fewer candidates than k,
// bit3 = more entries at distance <= T than the sort buffer holds (general select at resolve time), bit4 = the tightened
// threshold admitted fewer than k codes (the call is redone by the general chain).
static constexpr int HF_REG = 8;
__global__ __launch_bounds__(1024) void hamming_pick_kernel(const u64* __restrict__ seg, const u32* __restrict__ bcnt, int G,
                                                             int nq, u32 S, int bits, int k, int kk, const int* __restrict__ thr,
                                                             int map_rows, RowPerm pm, long long n, long long id_base,
                                                             int* __restrict__ out_dist, long long* __restrict__ out_idx,
                                                             u32* __restrict__ status, u32* __restrict__ host_words, int host_nq,
                                                             u32 cap, u32* __restrict__ sample_hist) {
    __shared__ u32 s_off[2049];
    __shared__ u32 s_hist[HF_MAX_BINS];
    __shared__ u64 s_keys[HF_SORT_CAP];
    __shared__ u32 s_wave[16];
    __shared__ u32 s_over, s_n;
    __shared__ int s_T;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nb = bits + 1;
    if (tid == 0) {
        s_over = 0u;
        s_n = 0u;
        s_T = bits;
    }
    for (int i = tid; i < nb; i += 1024) s_hist[i] = 0u;
    const int ts = thr[q];
    // ---- offsets of the mini-lists (thread t owns fills 2t, 2t + 1)
    u32 c0 = 0, c1 = 0;
    bool over = false;
    {
        const int g0 = 2 * tid, g1 = 2 * tid + 1;
        if (g0 < G) c0 = bcnt[(long long)g0 * nq + q];
        if (g1 < G) c1 = bcnt[(long long)g1 * nq + q];
        if (c0 > S) over = true, c0 = S;
        if (c1 > S) over = true, c1 = S;
    }
    __syncthreads();
    if (over) s_over = 1u;
    const u32 sum = c0 + c1;
    const u32 inc = hf_wave_incl(sum, lane);
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    u32 base = inc - sum;
    for (int w = 0; w < wv; ++w) base += s_wave[w];
    if (2 * tid <= G) s_off[2 * tid] = base;
    if (2 * tid + 1 <= G) s_off[2 * tid + 1] = base + c0;
    if (tid == 1023) s_off[2048] = base + sum;   // G == 2048: the total has no owner above
    __syncthreads();
    const u32 M = s_off[G];
    const int nbatch = (int)((M + 1024u * HF_REG - 1) / (1024u * HF_REG));
    // entries [i0, i0 + 8) of the numbering; ~0 where the numbering ends
    auto load_batch = [&](int b, u64 (&ent)[HF_REG]) {
        const u32 i0 = ((u32)b * 1024u + (u32)tid) * HF_REG;
        int g = 0;
        if (i0 < M) {   // the list that holds entry i0: the last g with s_off[g] <= i0 (empty lists share their offset with the next)
            int lo = 0, hi = G;   // invariant: s_off[lo] <= i0 < s_off[hi]
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (s_off[mid] <= i0) lo = mid;
                else hi = mid;
            }
            g = lo;
        }
        u32 end = s_off[g + 1];
#pragma unroll
        for (int j = 0; j < HF_REG; ++j) {
            const u32 i = i0 + (u32)j;
            if (i < M) {
                while (i >= end) end = s_off[++g + 1];
                ent[j] = seg[((long long)g * nq + q) * S + (i - s_off[g])];
            } else {
                ent[j] = ~0ull;
            }
        }
    };
    // ---- pass 1: exact histogram of the entries' distances: the three top values (nearly every entry) by ballot, the rest by
    // LDS atomics (same-address atomics of a whole wave serialise)
    u64 ent[HF_REG];
    u32 n0 = 0, n1 = 0, n2 = 0;
    for (int b = 0; b < nbatch; ++b) {
        load_batch(b, ent);
#pragma unroll
        for (int j = 0; j < HF_REG; ++j) {
            const u32 d = (u32)(ent[j] >> 32);
            const int di = ent[j] == ~0ull ? 0x7fffffff : (int)d;   // (an empty slot matches nothing)
            n0 += (u32)__popcll(__ballot(di == ts));
            n1 += (u32)__popcll(__ballot(di == ts - 1));
            n2 += (u32)__popcll(__ballot(di == ts - 2));
            if (di < ts - 2) atomicAdd(&s_hist[d], 1u);
        }
    }
    if (lane == 0) {
        if (n0) atomicAdd(&s_hist[ts], n0);
        if (n1 && ts >= 1) atomicAdd(&s_hist[ts - 1], n1);
        if (n2 && ts >= 2) atomicAdd(&s_hist[ts - 2], n2);
    }
    __syncthreads();
    // ---- T = the k-th smallest distance (wave 0)
    const u32 need = (u32)kk < M ? (u32)kk : M;
    if (wv == 0 && need > 0) {
        const int per = (nb + 63) / 64;
        u32 ls = 0;
        for (int j = 0; j < per; ++j) {
            const int b = lane * per + j;
            if (b < nb) ls += s_hist[b];
        }
        const u32 li = hf_wave_incl(ls, lane), le = li - ls;
        if (le < need && need <= li) {
            u32 cum = le;
            for (int j = 0; j < per; ++j) {
                const int b = lane * per + j;
                if (b < nb) {
                    cum += s_hist[b];
                    if (cum >= need) {
                        s_T = b;
                        break;
                    }
                }
            }
        }
    }
    __syncthreads();
    const u32 T = (u32)s_T;
    // ---- pass 2: gather the entries with distance <= T (keys carry the caller's row from here on); one LDS atomic per wave and slot
    for (int b = 0; b < nbatch; ++b) {
        if (nbatch > 1) load_batch(b, ent);   // (one batch: still in registers)
#pragma unroll
        for (int j = 0; j < HF_REG; ++j) {
            u64 key = ent[j];
            const bool take = (u32)(key >> 32) <= T;   // (empty slots: 0xffffffff > T)
            const u64 mask = __ballot(take);
            if (mask != 0ull) {
                u32 at = 0;
                if (lane == 0) at = atomicAdd(&s_n, (u32)__popcll(mask));
                at = (u32)__builtin_amdgcn_readfirstlane((int)at);
                if (take) {
                    const u32 pos = at + (u32)__popcll(mask & ((1ull << lane) - 1ull));
                    if (pos < (u32)HF_SORT_CAP) {
                        if (map_rows) key = (key & 0xffffffff00000000ull) | (u64)orig_row((long long)(key & 0xffffffffull), pm, n);
                        s_keys[pos] = key;
                    }
                }
            }
        }
    }
    __syncthreads();
    const u32 n_sel = s_n;
    // (more candidates than the call's candidate capacity count as an overflow, as in the general chain: option candidate_cap)
    // (M < k: the stream's threshold -- tightened below the rank-k rule, see hamming_enqueue -- admitted fewer than k codes:
    // bit4, the call is redone with the safe threshold when it is resolved)
    u32 stw = ((s_over || M > cap) ? 1u : 0u) | (M < (u32)kk ? (4u | 16u) : 0u);
    if (n_sel > (u32)HF_SORT_CAP) {
        stw |= 8u;
    } else {
        if (n_sel > 1 && n_sel <= 1024u) {
            if (tid < 8) s_keys[n_sel + tid] = ~0ull;   // (HF_SORT_CAP > 1024 + 8)
            __syncthreads();
            // rank sort: keys are unique, a key's rank is the number of keys below it (broadcast LDS reads, no barriers)
            const u64 mykey = tid < (int)n_sel ? s_keys[tid] : ~0ull;
            u32 rank = 0;
            if (wv * 64 < (int)n_sel) {
                // (the buffer is padded to a multiple of 8 with maximal keys: eight independent broadcast reads per round)
                const u32 n8 = (n_sel + 7u) & ~7u;
                for (u32 j = 0; j < n8; j += 8) {
                    u64 kx[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) kx[u] = s_keys[j + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) rank += kx[u] < mykey ? 1u : 0u;
                }
            }
            __syncthreads();
            if (tid < (int)n_sel) s_keys[rank] = mykey;
            __syncthreads();
        } else if (n_sel > 1) {
            const int P = pow2_ceil((int)n_sel);
            for (int i = (int)n_sel + tid; i < P; i += 1024) s_keys[i] = ~0ull;
            bitonic_sort_lds<u64>(s_keys, P);
        }
        for (int j = tid; j < k; j += 1024) {
            const bool pad = j >= (int)need;
            const u64 key = pad ? ~0ull : s_keys[j];
            out_dist[(long long)q * k + j] = pad ? 0x7fffffff : (int)(key >> 32);
            out_idx[(long long)q * k + j] = pad ? -1ll : id_base + (long long)(key & 0xffffffffull);
        }
    }
    if (sample_hist)
        for (int i = tid; i < nb; i += 1024) sample_hist[(long long)q * nb + i] = 0u;
    if (tid == 0) {
        status[q] = stw;
        if (host_words) {
            host_words[q] = stw;
            host_words[host_nq + q] = M;
        }
    }
}

}  // namespace sq
