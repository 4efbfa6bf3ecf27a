// Hamming top-k over packed unique hash codes (gfx950).
//
// Replaces the hot loop of LinearHashIndex._nn
// (smqtk_indexing/impls/hash_index/linear.py:235-240): one python call of
// metrics.hamming_distance (utils/metrics.py:140-155) per stored code inside
// heapq.nsmallest.  Here: a persistent grid streams the code array
// (uint64[N][W], coalesced 16-byte loads, stored in a low-discrepancy
// permutation of the caller's sorted order), XOR + v_bcnt popcount per
// (code, query) with the batch's query words in LDS, a per-query integer
// distance threshold from a histogram of a block sample, survivors
// (distance, sorted row) into per-(workgroup, query) lists without global
// atomics, a prefix-sum compaction per query and `select_topk_kernel` over the
// few survivors.  Small arrays and overflowing queries take the all-keys path.
// Integer-exact.  DESIGN.md section 4.3.
#include <algorithm>

#include "sq_dma.hpp"
#include "sq_select.hpp"

namespace sq {

// (row * mul) mod n, or -- once the index has been appended to / removed from -- an explicit table: see orig_row
struct RowPerm {
    u64 mul, inv;
    const u32* rank;  // non-null: physical row p holds the code of sorted rank rank[p]
};

// Per-call workspace.  A synchronous search uses slot 0; asynchronous searches (SQ_MEM_DEVICE_ASYNC) rotate through
// `depth` slots, each with a stream of its own, so that the short kernels around the scan (histogram, threshold,
// compaction, select) of neighbouring calls overlap the scans and the host reads a call's status words one call later
// -- the scheme of the dense search (sq_dense.hip).
struct HammingCall {
    bool pending = false;
    const u64* qs = nullptr;
    int nq = 0, k = 0;
    int* out_dist = nullptr;
    long long* out_idx = nullptr;
    hipStream_t st = nullptr;
    bool small = false, prof = false, use_event = false, force_fb = false;
    bool fused = false;           // answered by hamming_pick_kernel: the mini-lists below are what a flagged query is redone from
    int G = 0, map_rows = 0;
    u32 S = 0;
    long long key_stride = 0;
    u32 cap = 0;
    sq_stats_t stats{};
};
struct HammingSlot {
    DevBuf keys, cnt, hist, thr, out_keys, status, seg, bcnt, sort_tmp;
    DevBuf fhist;                 // fused small-batch calls: histogram + ticket, zero between calls (sq_hamming_fused.hpp)
    void* fhist_zeroed = nullptr; // the allocation of fhist that has been wiped once
    HostPinned status_host;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_in = nullptr, ev_done = nullptr;
    hipStream_t own = nullptr;
    HammingCall call;
    void release() {
        for (DevBuf* b : {&keys, &cnt, &hist, &thr, &out_keys, &status, &seg, &bcnt, &sort_tmp, &fhist}) b->release();
        fhist_zeroed = nullptr;
        status_host.release();
        for (auto& e : ev)
            if (e) (void)hipEventDestroy(e), e = nullptr;
        if (ev_in) (void)hipEventDestroy(ev_in), ev_in = nullptr;
        if (ev_done) (void)hipEventDestroy(ev_done), ev_done = nullptr;
        if (own) (void)hipStreamDestroy(own), own = nullptr;
    }
};

struct HammingHandle : HandleBase {
    const u64* codes = nullptr;  // device, [n][words]
    DevBuf owned;                // backing store when the library owns the copy
    long long n = 0;
    int words = 0;
    long long id_base = 0;
    RowPerm pmul{1, 0, nullptr};  // physical row p holds caller row (p * pmul.mul) mod n (or pmul.rank[p])
    DevBuf rank;                  // u32[n] explicit ranks, after the first append / remove
    DevBuf mut_tmp;               // scratch of the mutation kernels
    static constexpr int kMaxDepth = 4;
    HammingSlot slot[kMaxDepth];
    bool no_fused = false;               // set while a fused call whose tightened threshold fell short is redone by the general chain
    int depth = 2;                       // asynchronous calls in flight (option hamming_async_depth, fixed while any is)
    unsigned long long async_calls = 0;  // asynchronous calls so far (slot = calls % depth)
    // shared by all calls: host-memory staging and the exact path (both synchronous)
    DevBuf q_dev, out_dist_dev, out_idx_dev, big_keys, fb_sort;
    PinnedStage stage;
    ~HammingHandle() override {
        for (DevBuf* b : {&owned, &rank, &mut_tmp, &q_dev, &out_dist_dev, &out_idx_dev, &big_keys, &fb_sort}) b->release();
        for (auto& sl : slot) sl.release();
        stage.release();
    }
};

// ------------------------------------------------------------------ kernels
// The host index keeps its codes SORTED (row id = rank of the code, which is what makes
// (distance, row) the canonical (distance, code value) order).  In sorted order the codes near a
// query sit in a few narrow ranges, so a block sample of the array says little about the distance
// distribution and the survivors pile up in a handful of workgroups' lists (every second query of a
// 10 M-code index overflowed to the exact path).  The device copy is therefore stored in a
// low-discrepancy permutation: physical row p holds sorted row (p * pmul) mod n with pmul ~ 0.618 n
// coprime to n, so any run of consecutive physical rows is spread evenly over the sorted array;
// keys carry the sorted row, recomputed only for survivors.
// (row * mul) mod n without a 64-bit division: Barrett with inv = floor((2^64 - 1) / n); the
// quotient estimate is at most 2 short.
__device__ __forceinline__ u32 orig_row(long long row, RowPerm pm, long long n) {
    if (pm.rank) return pm.rank[row];
    if (pm.mul == 1ull) return (u32)row;
    const u64 x = (u64)row * pm.mul;
    u64 r = x - __umul64hi(x, pm.inv) * (u64)n;
    if (r >= (u64)n) r -= (u64)n;
    if (r >= (u64)n) r -= (u64)n;
    return (u32)r;
}

static __global__ void hamming_permute_kernel(const u64* __restrict__ src, long long n, int W, u64 pmul,
                                              u64* __restrict__ dst) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const long long r = (long long)(((u64)p * pmul) % (u64)n);
    for (int w = 0; w < W; ++w) dst[p * W + w] = src[r * W + w];
}

// Row handled by register slot i of thread t in the block that starts at code bb.
// W <= 2: every 16-byte vector load is fully coalesced across the wave (lane t
// takes vector j*256+t of the block); wider codes stay contiguous per thread.
template <int W, int C>
__device__ __forceinline__ long long code_row(long long bb, int t, int i) {
    if constexpr (W <= 2) {
        constexpr int CPV = 2 / W;  // codes per 16-byte vector
        return bb + ((long long)(i / CPV) * 256 + t) * CPV + (i % CPV);
    } else {
        return bb + (long long)t * C + i;
    }
}

template <int W, int C>
__device__ __forceinline__ void load_codes(const u64* __restrict__ codes, long long n, long long bb, int t,
                                           u64 (&c)[C][W], bool (&valid)[C]) {
    if (bb + 256ll * C <= n) {
        if constexpr (W <= 2) {
            constexpr int CPV = 2 / W;
            constexpr int NV = C / CPV;
            const ulonglong2* p2 = reinterpret_cast<const ulonglong2*>(codes + bb * W);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                ulonglong2 v = p2[j * 256 + t];
                if constexpr (W == 1) {
                    c[2 * j][0] = v.x;
                    c[2 * j + 1][0] = v.y;
                } else {
                    c[j][0] = v.x;
                    c[j][1] = v.y;
                }
            }
        } else if constexpr ((W % 2) == 0) {
            const ulonglong2* p2 = reinterpret_cast<const ulonglong2*>(codes + (bb + (long long)t * C) * W);
#pragma unroll
            for (int i = 0; i < C * W / 2; ++i) {
                ulonglong2 v = p2[i];
                c[(2 * i) / W][(2 * i) % W] = v.x;
                c[(2 * i + 1) / W][(2 * i + 1) % W] = v.y;
            }
        } else {
            const u64* p = codes + (bb + (long long)t * C) * W;
#pragma unroll
            for (int i = 0; i < C * W; ++i) c[i / W][i % W] = p[i];
        }
#pragma unroll
        for (int i = 0; i < C; ++i) valid[i] = true;
    } else {
#pragma unroll
        for (int i = 0; i < C; ++i) {
            const long long r = code_row<W, C>(bb, t, i);
            valid[i] = r < n;
#pragma unroll
            for (int w = 0; w < W; ++w) c[i][w] = valid[i] ? codes[r * W + w] : 0ull;
        }
    }
}

// mode 0: emit keys with dist <= thr[q] through per-query atomic counters.
// mode 1: write the key of EVERY code at its own position (keys[q][row]).
template <int W, int C>
__global__ __launch_bounds__(256) void hamming_scan_kernel(const u64* __restrict__ codes, long long n, RowPerm pmul,
                                                            const u64* __restrict__ qs, int nq,
                                                            const int* __restrict__ thr,
                                                            u64* __restrict__ keys, u32* __restrict__ cnt,
                                                            u32 cap, long long key_stride, int mode) {
    const long long bb = (long long)blockIdx.x * 256 * C;
    const int tid = threadIdx.x;
    u64 c[C][W];
    bool valid[C];
    load_codes<W, C>(codes, n, bb, tid, c, valid);
    for (int q = 0; q < nq; ++q) {
        const u64* qp = qs + (long long)q * W;  // wave-uniform: scalar loads
        u64 qw[W];
#pragma unroll
        for (int w = 0; w < W; ++w) qw[w] = qp[w];
        const int t = mode == 0 ? thr[q] : 0x7fffffff;
#pragma unroll
        for (int i = 0; i < C; ++i) {
            int dist = 0;
#pragma unroll
            for (int w = 0; w < W; ++w) dist += __popcll(c[i][w] ^ qw[w]);
            if (valid[i] && dist <= t) {
                const long long row = code_row<W, C>(bb, tid, i);
                const u64 key = ((u64)(u32)dist << 32) | (u64)orig_row(row, pmul, n);
                if (mode == 0) {
                    u32 pos = atomicAdd(&cnt[q], 1u);
                    if (pos < cap) keys[(long long)q * key_stride + pos] = key;
                } else {
                    keys[(long long)q * key_stride + row] = key;
                }
            }
        }
    }
}

// ---- the streaming scan of large code arrays --------------------------------
// Persistent workgroups (a few per CU) walk the 256*C-code chunks of the array
// round-robin.  Queries and thresholds sit in LDS and are taken QG at a time
// into registers, so the inner loop is pure VALU (xor + v_bcnt + compare, ~5 ops
// per (code word, query)); a lane whose distance is within the threshold takes a
// slot in the block's own (block, query) mini-list with an LDS atomic and stores
// the key there -- no global atomic anywhere in the stream (per-survivor
// returning atomics on 32 hot counters cost 10x at 32 queries).  Mini-lists:
// seg[(block*nq + q)*S + slot]; their fills go to bcnt[block*nq + q].
template <int W, int C>
__global__ __launch_bounds__(256) void hamming_stream_kernel(const u64* __restrict__ codes, long long n, RowPerm pmul,
                                                              const u64* __restrict__ qs, int nq,
                                                              const int* __restrict__ thr, u64* __restrict__ seg,
                                                              u32* __restrict__ bcnt, u32 S) {
    constexpr int QG = 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];
    u64* lq = reinterpret_cast<u64*>(hsm);                       // [nq][W]
    int* lthr = reinterpret_cast<int*>(lq + (size_t)nq * W);     // [nq]
    u32* lcnt = reinterpret_cast<u32*>(lthr + nq);               // [nq]
    const int tid = threadIdx.x;
    for (int i = tid; i < nq * W; i += 256) lq[i] = qs[i];
    for (int i = tid; i < nq; i += 256) {
        lthr[i] = thr[i];
        lcnt[i] = 0u;
    }
    __syncthreads();
    const long long per_chunk = 256ll * C;
    const long long nchunks = (n + per_chunk - 1) / per_chunk;
    u64* myseg = seg + (long long)blockIdx.x * nq * S;
    u64 cn[C][W];
    bool validn[C];
    if ((long long)blockIdx.x < nchunks) load_codes<W, C>(codes, n, (long long)blockIdx.x * per_chunk, tid, cn, validn);
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
        for (int q0 = 0; q0 < nq; q0 += QG) {
            u64 qw[QG][W];
            int tq[QG];
#pragma unroll
            for (int j = 0; j < QG; ++j) {
                const int q = (q0 + j) < nq ? (q0 + j) : (nq - 1);
#pragma unroll
                for (int w = 0; w < W; ++w) qw[j][w] = lq[(size_t)q * W + w];
                tq[j] = (q0 + j) < nq ? lthr[q] : -1;
            }
#pragma unroll
            for (int j = 0; j < QG; ++j) {
                int dist[C];
                int m = 0x7fffffff;
#pragma unroll
                for (int i = 0; i < C; ++i) {
                    int dsum = 0;
#pragma unroll
                    for (int w = 0; w < W; ++w) dsum += __popcll(c[i][w] ^ qw[j][w]);
                    dist[i] = dsum;
                    m = dsum < m ? dsum : m;
                }
                if (m <= tq[j]) {  // one branch per (query, C codes); padding codes are screened inside
#pragma unroll
                    for (int i = 0; i < C; ++i) {
                        if (valid[i] && dist[i] <= tq[j]) {
                            const int q = q0 + j;
                            const u32 pos = atomicAdd(&lcnt[q], 1u);
                            if (pos < S)
                                myseg[(long long)q * S + pos] =
                                    ((u64)(u32)dist[i] << 32) | (u64)orig_row(code_row<W, C>(bb, tid, i), pmul, n);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < nq; i += 256) bcnt[(long long)blockIdx.x * nq + i] = lcnt[i];
}

// ---- the same stream through an LDS-DMA ring (arrays far beyond the MALL, few queries: HBM bound) -------------
// The register-load stream above keeps one chunk per thread in flight (its prefetch) and reaches 5.0 TB/s on the
// 4 GB shard of BASELINE config 5; the dense scan's ring of `global_load_lds_dwordx4` pieces reaches 6.5-6.8 TB/s on
// the same HBM (MI355X_MICROARCH.md, "ldsdma-fill").  Here: one 512-thread workgroup per CU; every wave owns a ring
// of NSTAGE units of 8 KiB in LDS (8 x 1 KiB DMA pieces, lane-linear: the LDS image of a unit is its global image),
// units dealt round-robin over all waves of the launch, no workgroup barrier in the loop -- a wave consumes only what
// it loaded, ordered by counted `s_waitcnt vmcnt`.  A lane reads WHOLE codes out of the image (`CPL` consecutive
// 16-byte chunks at lane stride 16 CPL bytes: 2-way bank conflicts at 256 bits, once per unit, not per query), so
// the inner loop is xor + v_bcnt per 32 bits + one min per code with the query words and thresholds in scalar
// registers.  Survivors leave exactly as in the register kernel: LDS counter per query, per-(workgroup, query)
// mini-lists, hamming_compact_kernel afterwards.  The survivor stores are younger VMEM operations in the same
// in-order queue as the DMA pieces: a counted wait then waits for a little more than it must (never for less).
// Code widths 64 .. 1024 bits in powers of two; the array must be 16-byte aligned.  The last partial unit is read by
// one wave with guarded register loads after its ring has drained.
template <int W>
struct RingShape {
    static constexpr int CPL = W >= 2 ? W / 2 : 1;   // 16-byte chunks a lane reads in a row (one code; W = 1: two codes)
    static constexpr int NG = 8 / CPL;               // such groups per 8 KiB unit and lane
    static constexpr int CODES = W == 1 ? 16 : NG;   // codes per lane and unit
    static constexpr int UNIT_CODES = 1024 / W;      // codes per unit
    static_assert(W == 1 || W == 2 || W == 4 || W == 8 || W == 16, "ring kernel: 64 .. 1024-bit codes, powers of two");
};
static constexpr int HR_WAVES = 8, HR_NSTAGE = 2, HR_UNIT = 8192;

// row of code slot i of lane `lane` in the unit that starts at code `code0`
template <int W>
__device__ __forceinline__ long long ring_row(long long code0, int lane, int i) {
    if constexpr (W == 1)
        return code0 + ((long long)(i >> 1) * 64 + lane) * 2 + (i & 1);
    else
        return code0 + (long long)i * 64 + lane;
}

// popcount(x) + acc as two v_bcnt_u32_b32 with their accumulate operand: one serial chain per code.  (Left to hipcc the
// 2 W popcounts of a code become W pairs plus an adder tree: 75 instead of 67 vector instructions per query and four
// 256-bit codes, in a loop that is VALU bound from ~8 queries per pass on.)
__device__ __forceinline__ u32 popc64_acc(u64 x, u32 acc) {
    u32 r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"((u32)x), "v"(acc));
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(acc) : "v"((u32)(x >> 32)), "v"(r));
    return acc;
}

template <int W, bool NT>
__global__ __launch_bounds__(HR_WAVES * 64, 1) void hamming_ring_kernel(const u64* __restrict__ codes, long long n, RowPerm pmul,
                                                                         const u64* __restrict__ qs, int nq,
                                                                         const int* __restrict__ thr, u64* __restrict__ seg,
                                                                         u32* __restrict__ bcnt, u32 S) {
    using R = RingShape<W>;
    extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u32* lcnt = reinterpret_cast<u32*>(hsm + (size_t)HR_WAVES * HR_NSTAGE * HR_UNIT);   // [nq]
    for (int i = threadIdx.x; i < nq; i += HR_WAVES * 64) lcnt[i] = 0u;
    __syncthreads();
    const u32 lds_base = (u32)(uintptr_t)hsm;  // low 32 bits of a flat LDS address = LDS offset
    const u32 ring_base = lds_base + (u32)wave * (HR_NSTAGE * HR_UNIT);
    const unsigned char* ring_ptr = hsm + (size_t)wave * (HR_NSTAGE * HR_UNIT);
    u64* myseg = seg + (long long)blockIdx.x * nq * S;

    const long long units = (n * (long long)W * 8) / HR_UNIT;            // whole units; the rest: tail below
    const long long nwaves = (long long)gridDim.x * HR_WAVES;
    const long long gw = (long long)blockIdx.x * HR_WAVES + wave;
    const long long mine = units > gw ? (units - gw + nwaves - 1) / nwaves : 0;   // units of this wave
    const unsigned char* gbase = reinterpret_cast<const unsigned char*>(codes);
    const u32 voff = (u32)lane * 16u;

    // compare the codes a lane holds with every query of the batch.  Query words and threshold of query q + 1 are
    // requested (scalar loads) before query q is compared, so their latency sits under ~70 vector instructions.
    // Keys carry the PHYSICAL row: the compaction maps it to the caller's row (orig_row) -- a rank-table load here
    // would be a compiler-visible VMEM load, and hipcc waits for those with vmcnt(0), which drains the ring.
    auto compare = [&](const u64 (&c)[R::CODES][W], const long long code0) __attribute__((always_inline)) {
        u64 qn[W];
#pragma unroll
        for (int w = 0; w < W; ++w) qn[w] = qs[w];
        int tn = thr[0];
        for (int q = 0; q < nq; ++q) {
            u64 qw[W];
#pragma unroll
            for (int w = 0; w < W; ++w) {
                // the query word as a VECTOR operand (one v_mov per half): v_xor_b32 with a scalar source issues at 4 cycles
                // per wave and SIMD, with vector sources at 2.5 (tools/micro/valu_rate.hip, profiles/r04_valu_issue_rates.txt)
                u32 lo = (u32)qn[w], hi = (u32)(qn[w] >> 32);
                asm volatile("" : "+v"(lo), "+v"(hi));
                qw[w] = ((u64)hi << 32) | lo;
            }
            const int t = tn;
            {
                const int q1 = q + 1 < nq ? q + 1 : q;   // wave-uniform: scalar loads
                const u64* qp = qs + (long long)q1 * W;
#pragma unroll
                for (int w = 0; w < W; ++w) qn[w] = qp[w];
                tn = thr[q1];
            }
            int dist[R::CODES];
            int m = 0x7fffffff;
#pragma unroll
            for (int i = 0; i < R::CODES; ++i) {
                u32 dsum = 0;
#pragma unroll
                for (int w = 0; w < W; ++w) dsum = popc64_acc(c[i][w] ^ qw[w], dsum);
                dist[i] = (int)dsum;
                m = (int)dsum < m ? (int)dsum : m;
            }
            if (m <= t) {  // rare: one branch per (query, codes of a lane)
#pragma unroll
                for (int i = 0; i < R::CODES; ++i) {
                    const long long row = ring_row<W>(code0, lane, i);
                    if (dist[i] <= t && row < n) {
                        const u32 pos = atomicAdd(&lcnt[q], 1u);
                        if (pos < S) myseg[(long long)q * S + pos] = ((u64)(u32)dist[i] << 32) | (u64)(u32)row;
                    }
                }
            }
        }
    };

    long long issued = 0;
    auto issue_unit = [&]() __attribute__((always_inline)) {
        const long long u = gw + issued * nwaves;
        const unsigned char* src = gbase + u * HR_UNIT;
        const u32 dst = ring_base + (u32)(issued % HR_NSTAGE) * HR_UNIT;
#pragma unroll
        for (int j = 0; j < 8; ++j) glds16<NT>(src + j * 1024, voff, dst + (u32)j * 1024);
        ++issued;
    };
    for (int p = 0; p < HR_NSTAGE; ++p)
        if (issued < mine) issue_unit();
    for (long long it = 0; it < mine; ++it) {
        wait_units_in_flight<HR_NSTAGE, 8>((int)(issued - it - 1));   // younger units may stay in flight
        const unsigned char* sl = ring_ptr + (it % HR_NSTAGE) * HR_UNIT;
        u64 c[R::CODES][W];
#pragma unroll
        for (int g = 0; g < R::NG; ++g) {
#pragma unroll
            for (int cc = 0; cc < R::CPL; ++cc) {
                const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(sl + ((g * 64 + lane) * R::CPL + cc) * 16);
                if constexpr (W == 1) {
                    c[2 * g][0] = v.x;
                    c[2 * g + 1][0] = v.y;
                } else {
                    c[g][2 * cc] = v.x;
                    c[g][2 * cc + 1] = v.y;
                }
            }
        }
        // the codes are in registers (lgkmcnt(0)): the slot is free for the unit NSTAGE further on
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (issued < mine) issue_unit();
        compare(c, (gw + it * nwaves) * R::UNIT_CODES);
    }
    // the last, partial unit: the wave after the one that took the last whole unit, with guarded register loads
    const long long tail0 = units * R::UNIT_CODES;
    if (tail0 < n && gw == units % nwaves) {
        wait_vmcnt<0>();
        u64 c[R::CODES][W];
#pragma unroll
        for (int i = 0; i < R::CODES; ++i) {
            const long long row = ring_row<W>(tail0, lane, i);
#pragma unroll
            for (int w = 0; w < W; ++w) c[i][w] = row < n ? codes[row * W + w] : 0ull;
        }
        compare(c, tail0);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nq; i += HR_WAVES * 64) bcnt[(long long)blockIdx.x * nq + i] = lcnt[i];
}

// Concatenate the G blocks' mini-lists of a query into keys[q][..] (prefix sum over the fills, no
// atomics) and publish the total in cnt[q]; a mini-list that overflowed its S slots marks the query
// (cnt = cap + 1) so that it is recomputed on the exact path.  Grid (nq, COMPACT_SLICES): every
// workgroup scans all G fills (G <= 2048: eight per thread, wave scan) and copies the lists of its
// slice -- one workgroup per query with a one-thread prefix loop took 55 us for 32 queries.
static constexpr int COMPACT_SLICES = 8;
// map_rows: the mini-lists hold PHYSICAL rows (hamming_ring_kernel); the copy turns them into caller rows.
__global__ __launch_bounds__(256) void hamming_compact_kernel(const u64* __restrict__ seg,
                                                               const u32* __restrict__ bcnt, int G, int nq, u32 S,
                                                               u64* __restrict__ keys, u32* __restrict__ cnt, u32 cap,
                                                               long long key_stride, int map_rows, RowPerm pm, long long n) {
    __shared__ u32 s_off[2049];
    __shared__ u32 s_wave[4];
    __shared__ u32 s_over;
    const int q = blockIdx.x;
    if (threadIdx.x == 0) s_over = 0u;
    __syncthreads();
    // thread t owns fills 8t .. 8t+7
    u32 c[8], sum = 0;
    bool over = false;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int g = 8 * threadIdx.x + i;
        u32 v = g < G ? bcnt[(long long)g * nq + q] : 0u;
        if (v > S) {
            over = true;
            v = S;
        }
        c[i] = v;
        sum += v;
    }
    if (over) s_over = 1u;
    u32 inc = sum;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) {
        const u32 t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    u32 base = inc - sum;
    for (int w = 0; w < wv; ++w) base += s_wave[w];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int g = 8 * threadIdx.x + i;
        if (g <= G) s_off[g] = base;
        base += c[i];
    }
    if (threadIdx.x == 255) s_off[2048] = base;  // G == 2048: the total has no owner above
    __syncthreads();
    const u32 total = s_off[G];
    const int per = (G + COMPACT_SLICES - 1) / COMPACT_SLICES;
    const int g0 = blockIdx.y * per, g1 = min(G, g0 + per);
    for (int g = g0 + (threadIdx.x >> 3); g < g1; g += 32) {  // 8 lanes per mini-list
        const u32 b0 = s_off[g], n_e = s_off[g + 1] - b0;
        const u64* src = seg + ((long long)g * nq + q) * S;
        for (u32 e = threadIdx.x & 7; e < n_e; e += 8)
            if (b0 + e < cap) {
                u64 key = src[e];
                if (map_rows) key = (key & 0xffffffff00000000ull) | (u64)orig_row((long long)(key & 0xffffffffull), pm, n);
                keys[(long long)q * key_stride + b0 + e] = key;
            }
    }
    if (threadIdx.x == 0 && blockIdx.y == 0) cnt[q] = s_over ? cap + 1u : total;
}

// Generic word count (W not specialised): one code per thread.
__global__ __launch_bounds__(256) void hamming_scan_generic_kernel(const u64* __restrict__ codes, long long n, RowPerm pmul,
                                                                    int W, const u64* __restrict__ qs, int nq,
                                                                    const int* __restrict__ thr,
                                                                    u64* __restrict__ keys, u32* __restrict__ cnt,
                                                                    u32 cap, long long key_stride, int mode) {
    const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
    if (row >= n) return;
    const u64* cp = codes + row * W;
    for (int q = 0; q < nq; ++q) {
        const u64* qp = qs + (long long)q * W;
        int dist = 0;
        for (int w = 0; w < W; ++w) dist += __popcll(cp[w] ^ qp[w]);
        const int t = mode == 0 ? thr[q] : 0x7fffffff;
        if (dist <= t) {
            const u64 key = ((u64)(u32)dist << 32) | (u64)orig_row(row, pmul, n);
            if (mode == 0) {
                u32 pos = atomicAdd(&cnt[q], 1u);
                if (pos < cap) keys[(long long)q * key_stride + pos] = key;
            } else {
                keys[(long long)q * key_stride + row] = key;
            }
        }
    }
}

// Histogram of distances over every `block_step`-th 256*C-code block.
// grid.y chunks the queries so the LDS histogram fits.  hist: [nq][bits+1].
template <int W, int C>
__global__ __launch_bounds__(256) void hamming_hist_kernel(const u64* __restrict__ codes, long long n,
                                                            const u64* __restrict__ qs, int nq, int bits,
                                                            u32* __restrict__ hist, int block_step, int qchunk) {
    extern __shared__ u32 lh[];
    const int nb = bits + 1;
    const int q0 = blockIdx.y * qchunk;
    const int qn = (nq - q0) < qchunk ? (nq - q0) : qchunk;
    for (int i = threadIdx.x; i < qn * nb; i += 256) lh[i] = 0;
    __syncthreads();
    const long long bb = ((long long)blockIdx.x * block_step) * 256 * C;
    if (bb < n) {
        u64 c[C][W];
        bool valid[C];
        load_codes<W, C>(codes, n, bb, (int)threadIdx.x, c, valid);
        for (int q = 0; q < qn; ++q) {
            const u64* qp = qs + (long long)(q0 + q) * W;
#pragma unroll
            for (int i = 0; i < C; ++i) {
                int dist = 0;
#pragma unroll
                for (int w = 0; w < W; ++w) dist += __popcll(c[i][w] ^ qp[w]);
                if (valid[i]) atomicAdd(&lh[q * nb + dist], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < qn * nb; i += 256)
        if (lh[i]) atomicAdd(&hist[(long long)q0 * nb + i], lh[i]);
}

__global__ void hamming_hist_generic_kernel(const u64* __restrict__ codes, long long n, int W,
                                            const u64* __restrict__ qs, int nq, int bits,
                                            u32* __restrict__ hist, int block_step) {
    const long long row = ((long long)blockIdx.x * block_step) * 256 + threadIdx.x;
    if (row >= n) return;
    const int nb = bits + 1;
    const u64* cp = codes + row * W;
    for (int q = 0; q < nq; ++q) {
        const u64* qp = qs + (long long)q * W;
        int dist = 0;
        for (int w = 0; w < W; ++w) dist += __popcll(cp[w] ^ qp[w]);
        atomicAdd(&hist[(long long)q * nb + dist], 1u);
    }
}

// thr[q] = smallest t whose cumulative sample count reaches k (an upper bound
// of the k-th smallest distance over the whole array); `bits` when the sample
// holds fewer than k codes.
__global__ void hamming_thr_kernel(const u32* __restrict__ hist, int nq, int bits, int k, int* __restrict__ thr) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const u32* h = hist + (long long)q * (bits + 1);
    u32 c = 0;
    int t = bits;
    for (int b = 0; b <= bits; ++b) {
        c += h[b];
        if (c >= (u32)k) {
            t = b;
            break;
        }
    }
    thr[q] = t;
}

// sorted keys -> (distance, global id); status bit0 = candidate overflow, bit2 = fewer candidates than k.
// Runs as the post-operation of the select (sq_select.hpp: `post(q, sorted keys of query q, k)` runs in the
// selecting workgroup once its k keys are written): one launch less per search.
struct HammingFinalize {
    const u32* cnt;
    u32 cap;
    int kk;
    long long id_base;
    int* out_dist;
    long long* out_idx;
    u32* status;
    u32* host_words;  // pinned [status (host_nq) | counts (host_nq)] or nullptr
    int host_nq;
    __device__ __forceinline__ void operator()(int q, const u64* sorted, int k) const {
        for (int j = threadIdx.x; j < k; j += blockDim.x) {
            const u64 key = sorted[j];
            const bool pad = key == ~0ull;
            out_dist[(long long)q * k + j] = pad ? 0x7fffffff : (int)(key >> 32);
            out_idx[(long long)q * k + j] = pad ? -1ll : id_base + (long long)(key & 0xffffffffull);
        }
        if (threadIdx.x == 0) {
            const u32 c = cnt[q];
            const u32 stw = (c > cap ? 1u : 0u) | (c < (u32)kk ? 4u : 0u);
            status[q] = stw;
            if (host_words) {
                host_words[q] = stw;
                host_words[host_nq + q] = c;
            }
        }
    }
};

}  // namespace sq
#include "sq_hamming_fused.hpp"
namespace sq {

// ------------------------------------------------------------- host driver
template <int W, int C>
static void launch_scan(const HammingHandle* h, const u64* qs, int nq, const int* thr, u64* keys, u32* cnt, u32 cap,
                        long long key_stride, int mode, hipStream_t st) {
    long long per_block = 256ll * C;
    unsigned blocks = (unsigned)((h->n + per_block - 1) / per_block);
    hipLaunchKernelGGL((hamming_scan_kernel<W, C>), dim3(blocks), dim3(256), 0, st, h->codes, h->n, h->pmul, qs, nq, thr, keys,
                       cnt, cap, key_stride, mode);
}

static void scan_dispatch(const HammingHandle* h, const u64* qs, int nq, const int* thr, u64* keys, u32* cnt, u32 cap,
                          long long key_stride, int mode, hipStream_t st) {
    switch (h->words) {
        case 1: launch_scan<1, 4>(h, qs, nq, thr, keys, cnt, cap, key_stride, mode, st); break;
        case 2: launch_scan<2, 2>(h, qs, nq, thr, keys, cnt, cap, key_stride, mode, st); break;
        case 4: launch_scan<4, 1>(h, qs, nq, thr, keys, cnt, cap, key_stride, mode, st); break;
        default: {
            unsigned blocks = (unsigned)((h->n + 255) / 256);
            hipLaunchKernelGGL(hamming_scan_generic_kernel, dim3(blocks), dim3(256), 0, st, h->codes, h->n, h->pmul, h->words,
                               qs, nq, thr, keys, cnt, cap, key_stride, mode);
        }
    }
}

template <int W, int C>
static void launch_hist(const HammingHandle* h, const u64* qs, int nq, int bits, u32* hist, int step, hipStream_t st) {
    long long per_block = 256ll * C;
    long long blocks_all = (h->n + per_block - 1) / per_block;
    unsigned blocks = (unsigned)((blocks_all + step - 1) / step);
    int nb = bits + 1;
    int qchunk = (48 * 1024 / 4) / nb;
    if (qchunk < 1) qchunk = 1;
    if (qchunk > nq) qchunk = nq;
    unsigned gy = (unsigned)((nq + qchunk - 1) / qchunk);
    size_t lds = (size_t)qchunk * nb * 4;
    hipLaunchKernelGGL((hamming_hist_kernel<W, C>), dim3(blocks, gy), dim3(256), lds, st, h->codes, h->n, qs, nq, bits,
                       hist, step, qchunk);
}

static void hist_dispatch(const HammingHandle* h, const u64* qs, int nq, int bits, u32* hist, int step,
                          hipStream_t st) {
    switch (h->words) {
        case 1: launch_hist<1, 4>(h, qs, nq, bits, hist, step, st); break;
        case 2: launch_hist<2, 2>(h, qs, nq, bits, hist, step, st); break;
        case 4: launch_hist<4, 1>(h, qs, nq, bits, hist, step, st); break;
        default: {
            long long blocks_all = (h->n + 255) / 256;
            unsigned blocks = (unsigned)((blocks_all + step - 1) / step);
            hipLaunchKernelGGL(hamming_hist_generic_kernel, dim3(blocks), dim3(256), 0, st, h->codes, h->n, h->words,
                               qs, nq, bits, hist, step);
        }
    }
}

static constexpr int kSelectLdsKeys64 = 16384;  // 128 KiB of LDS for the candidate keys

template <class Post>
static int select_launch(const u64* keys, const u32* cnt, u32 cap, long long stride, int k, int nq, u64* out,
                         hipStream_t st, DevBuf& sort_scratch, Post post) {
    static bool attr_set = false;
    if (k > kSelectLdsKeys64)  // linear.py:235-238 has no limit on n: the any-k sorted select (sq_select.hpp)
        return sort_select_large<u64, Post>(keys, cnt, cap, stride, k, nq, out, sort_scratch, post, st);
    const size_t lds = (size_t)(kSelectLdsKeys64 + SELECT_SORT_MAX) * sizeof(u64);
    if (!attr_set) {
        SQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&select_topk_kernel<u64, Post>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((select_topk_kernel<u64, Post>), dim3(nq), dim3(1024), lds, st, keys, cnt, cap, stride, k,
                       kSelectLdsKeys64, out, post);
    return SQ_OK;
}

// Wait for an event the way stream_wait waits for a stream (poll, then block).
static hipError_t hamming_event_wait(hipEvent_t ev) {
    const long long budget_us = g_opt.spin_wait_us;
    if (budget_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            for (int i = 0; i < 64; ++i) {
                const hipError_t e = hipEventQuery(ev);
                if (e != hipErrorNotReady) return e;
            }
            if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > budget_us)
                break;
        }
    }
    return hipEventSynchronize(ev);
}

template <int W>
static int ring_launch(const HammingHandle* h, bool nt, int G, const u64* qc, int nqc, const int* thr, u64* seg, u32* bcnt, u32 S,
                       hipStream_t st) {
    static bool attr_set[2] = {false, false};
    const size_t lds = (size_t)HR_WAVES * HR_NSTAGE * HR_UNIT + (size_t)nqc * 4;
    if (!attr_set[nt ? 1 : 0]) {
        if (nt)
            SQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hamming_ring_kernel<W, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        else
            SQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hamming_ring_kernel<W, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[nt ? 1 : 0] = true;
    }
    if (nt)
        hipLaunchKernelGGL((hamming_ring_kernel<W, true>), dim3(G), dim3(HR_WAVES * 64), lds, st, h->codes, h->n, h->pmul, qc, nqc, thr,
                           seg, bcnt, S);
    else
        hipLaunchKernelGGL((hamming_ring_kernel<W, false>), dim3(G), dim3(HR_WAVES * 64), lds, st, h->codes, h->n, h->pmul, qc, nqc, thr,
                           seg, bcnt, S);
    return SQ_OK;
}

// Enqueue one search on `st` with the workspace of slot `s`; nothing is waited for.  hamming_resolve() finishes
// the call: it waits for the kernels, reads the status words and sends overflowing queries down the exact path.
static int hamming_enqueue(HammingHandle* h, HammingSlot& s, const u64* qs, int nq, int k, int* out_dist, long long* out_idx,
                           hipStream_t st, bool use_event) {
    const long long n = h->n;
    const int W = h->words, bits = W * 64;
    const int kk = (int)(k < n ? k : n);
    const bool prof = h->opt.profile == 1 || (h->opt.profile > 1 && (!use_event || h->async_calls % (unsigned)h->opt.profile == 0));
    u32 cap = h->opt.candidate_cap > 0 ? (u32)h->opt.candidate_cap : 65536u;
    // room for the tie group at the threshold distance (integer distances: the codes AT the threshold can outnumber
    // those below it several times)
    if (h->opt.candidate_cap <= 0 && cap < (u32)std::min<long long>(16ll * kk, 1ll << 30)) cap = (u32)std::min<long long>(16ll * kk, 1ll << 30);
    if (cap < (u32)(2 * kk)) cap = (u32)(2 * kk);
    const bool small = n <= (long long)cap;
    HammingCall& c = s.call;
    c = HammingCall{};
    c.qs = qs;
    c.nq = nq;
    c.k = k;
    c.out_dist = out_dist;
    c.out_idx = out_idx;
    c.st = st;
    c.small = small;
    c.prof = prof;
    c.use_event = use_event;
    c.cap = cap;
    if (prof) {
        for (auto& e : s.ev)
            if (!e) SQ_HIP(hipEventCreate(&e));
        SQ_HIP(hipEventRecord(s.ev[0], st));
    }
    SQ_TRY(s.cnt.reserve((size_t)nq * 4));
    SQ_TRY(s.thr.reserve((size_t)nq * 4));
    SQ_TRY(s.out_keys.reserve((size_t)nq * k * 8));
    SQ_TRY(s.status.reserve((size_t)nq * 4));
    SQ_TRY(s.status_host.reserve((size_t)nq * 8));
    u32* hs = reinterpret_cast<u32*>(s.status_host.p);  // [status (nq) | counts (nq)]
    u32* hs_dev = nullptr;
    SQ_TRY(s.status_host.device_ptr(reinterpret_cast<void**>(&hs_dev)));
    u32* cnt = s.cnt.as<u32>();
    int* thr = s.thr.as<int>();
    u64* okeys = s.out_keys.as<u64>();
    u32* status = s.status.as<u32>();
    const long long key_stride = small ? n : (long long)cap;
    SQ_TRY(s.keys.reserve((size_t)nq * key_stride * 8));
    u64* keys = s.keys.as<u64>();

    if (small) {
        hipLaunchKernelGGL(fill_u32_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, cnt, (long long)nq, (u32)n);
        if (prof) SQ_HIP(hipEventRecord(s.ev[1], st));
        scan_dispatch(h, qs, nq, thr, keys, cnt, (u32)n, key_stride, /*mode*/ 1, st);
        if (prof) SQ_HIP(hipEventRecord(s.ev[2], st));
        c.stats.scan_launches = 1;
        c.stats.bytes_scanned = n * W * 8;
        SQ_TRY(select_launch(keys, cnt, (u32)n, key_stride, k, nq, okeys, st, s.sort_tmp,
                             HammingFinalize{cnt, (u32)n, kk, h->id_base, out_dist, out_idx, status, hs_dev, nq}));
    } else {
        int step = h->opt.sample_stride > 0 ? h->opt.sample_stride : 64;
        // keep the sample comfortably larger than k
        const long long per_block = 256ll * (W == 1 ? 4 : W == 2 ? 2 : 1);
        const long long blocks_all = (n + per_block - 1) / per_block;
        while (step > 1 && (blocks_all / step) * per_block < 64ll * kk) step >>= 1;
        // the threshold admits ~step * k codes (k of them in the 1/step sample) times the tie expansion: keep that
        // inside the candidate lists.  (At the default step of 64 every query with k >= ~500 overflowed its list
        // and took the exact path: 18 ms per query at 10 M codes, found by bench.py --workload lsh_c3.)
        if (h->opt.sample_stride <= 0)
            while (step > 1 && (long long)step * kk * 8 > (long long)cap) step >>= 1;
        const bool stream_ok = W == 1 || W == 2 || W == 4;
        // LDS-DMA ring (hamming_ring_kernel): HBM bound batches over arrays the MALL cannot hold.  Beyond ~24 queries
        // the inner loop is VALU bound and the register kernel's 32 waves per CU win.
        const size_t code_bytes = (size_t)n * W * 8;
        const bool ring_shape = (W == 1 || W == 2 || W == 4 || W == 8 || W == 16) && (reinterpret_cast<uintptr_t>(h->codes) & 15u) == 0 &&
                                n * (long long)W * 8 >= 2ll * HR_UNIT * HR_WAVES;
        const bool ring = ring_shape && (h->opt.hamming_ring == 1 ||
                                         (h->opt.hamming_ring < 0 && nq <= 24 && code_bytes >= ((size_t)256 << 20)) ||
                                         (h->opt.hamming_ring < 0 && !stream_ok && nq <= 64));
        // Small batches in three launches (sq_hamming_fused.hpp): head (sampled histogram + thresholds by the last
        // workgroup), the stream, pick (prefix sum over the mini-lists, exact k-th distance, gather, sort, results).
        // (calls beyond 32 queries -- up to one stream launch's batch -- keep the threshold launch: hamming_body_kernel then reads
        // the thresholds instead of computing them in its prologue)
        const int fused_max_nq = ring ? HF_MAX_NQ : (W == 4 ? 384 : 1024);
        const bool fused = h->opt.hamming_fused != 0 && !h->no_fused && stream_ok && nq <= fused_max_nq && 2 * kk <= HF_SORT_CAP;
        // The fused call's threshold: the general chain takes the smallest t whose SAMPLE count reaches k -- safe (the sample
        // is a subset) and loose: ~step x k codes pass (7.8 k per query at 10 M x 64 bits, k = 100), and with 0.4 survivors
        // per wave and chunk the emission path, not the popcounts, is half of the stream's time.  The pick kernel counts what
        // the stream admitted, so the threshold may be a bet: the smallest t whose sample count reaches r, where a t that
        // admits fewer than k codes overall would show r in a 1/step sample with probability < 1e-9 (Poisson tail:
        // r = lambda + 7 sqrt(lambda) + 6, lambda = k / step).  A lost bet (M < k) is seen by the pick kernel and the call
        // is redone with the safe rule (hamming_resolve); results are exact either way.
        int thr_rank = kk;
        if (fused && h->opt.hamming_tighten != 0) {
            const double lambda = (double)kk / (double)step;
            const int r = (int)ceil(lambda + 7.0 * sqrt(lambda) + 6.0);
            if (r < thr_rank) thr_rank = r;
            if (h->opt.hamming_tighten >= 2) thr_rank = 1;   // (testing: a bet that is usually lost -- the redo path)
        }
        c.fused = fused;
        // the stream of a fused call: hamming_body_kernel (queries broadcast with v_readlane: two instructions per query word
        // and chunk -- 128- and 256-bit codes beyond 32 queries measure 4-35 % slower with it than with the LDS-broadcast
        // stream kernel, tools/hamming_width_sweep.sh) or, for those, hamming_stream_kernel between the tightened threshold
        // and the pick kernel
        const bool body_kernel = fused && !ring && (W == 1 || nq <= HF_MAX_NQ);
        u32* hist = nullptr;
        if (fused) {
            const size_t words = (size_t)(nq > HF_MAX_NQ ? nq : HF_MAX_NQ) * (bits + 1);
            SQ_TRY(s.fhist.reserve(words * 4));
            if (s.fhist_zeroed != s.fhist.p) {   // a new allocation: wiped once, the pick kernel leaves it clean
                SQ_HIP(hipMemsetAsync(s.fhist.p, 0, s.fhist.cap, st));
                s.fhist_zeroed = s.fhist.p;
            }
            hist = s.fhist.as<u32>();
            if (nq > 8 && nq <= HF_MAX_NQ) {   // a lane per query: no same-address LDS atomics (sq_hamming_fused.hpp)
                const long long blocks_all2 = (n + per_block - 1) / per_block;
                const unsigned blocks = (unsigned)((blocks_all2 + step - 1) / step);
                const size_t lds = (size_t)32 * (bits + 1) * 4;
                if (W == 1)
                    hipLaunchKernelGGL((hamming_hist_by_query_kernel<1, 4>), dim3(blocks), dim3(256), lds, st, h->codes, n, qs, nq, bits, hist, step);
                else if (W == 2)
                    hipLaunchKernelGGL((hamming_hist_by_query_kernel<2, 2>), dim3(blocks), dim3(256), lds, st, h->codes, n, qs, nq, bits, hist, step);
                else
                    hipLaunchKernelGGL((hamming_hist_by_query_kernel<4, 1>), dim3(blocks), dim3(256), lds, st, h->codes, n, qs, nq, bits, hist, step);
            } else {
                hist_dispatch(h, qs, nq, bits, hist, step, st);
            }
            if (ring || nq > HF_MAX_NQ) hipLaunchKernelGGL(hamming_thr_kernel, dim3((nq + 63) / 64), dim3(64), 0, st, hist, nq, bits, thr_rank, thr);
        } else {
            SQ_TRY(s.hist.reserve((size_t)nq * (bits + 1) * 4));
            hist = s.hist.as<u32>();
            SQ_HIP(hipMemsetAsync(hist, 0, (size_t)nq * (bits + 1) * 4, st));
            if (!stream_ok && !ring) SQ_HIP(hipMemsetAsync(cnt, 0, (size_t)nq * 4, st));  // the compaction writes cnt itself
            hist_dispatch(h, qs, nq, bits, hist, step, st);
            hipLaunchKernelGGL(hamming_thr_kernel, dim3((nq + 63) / 64), dim3(64), 0, st, hist, nq, bits, kk, thr);
        }
        if (prof) SQ_HIP(hipEventRecord(s.ev[1], st));
        if (stream_ok || ring) {
            // streaming scan into per-(block, query) mini-lists, then a prefix-sum compaction: no global atomics
            // register kernel: 8 workgroups per CU (32 waves: the VALU-bound inner loop wants full occupancy); the LDS copy
            // of the queries (nq * (8W+8) bytes) must fit 8 times, so wide codes take the queries in smaller batches.
            // ring kernel: one 512-thread workgroup per CU -- on three quarters of the CUs when calls are pipelined and the
            // pass is HBM bound: the neighbouring calls' select needs most of a CU's LDS, and such a stream loses
            // nothing on 192 CUs
            const int qbatch = ring ? 64 : (W == 4 ? 384 : 1024);
            const int cus = cu_count(h->device);
            // (from ~8 queries per pass the ring kernel is VALU bound and wants every CU; the neighbours' short kernels
            // then simply queue behind it)
            int G = ring ? (use_event && nq <= 4 ? cus * 3 / 4 : cus) : 8 * cus;
            if (G > 2048) G = 2048;
            if (ring) {
                const long long units = (n * (long long)W * 8) / HR_UNIT;
                if ((long long)G * HR_WAVES > units) G = (int)std::max<long long>(1, units / HR_WAVES);
            } else {
                const long long per_chunk = 256ll * (W == 1 ? 8 : W == 2 ? 4 : 2);
                const long long nchunks = (n + per_chunk - 1) / per_chunk;
                if (fused && nchunks > (long long)8 * cus && 8 * cus <= 2048) {
                    // every CU holds j workgroups that walk ceil(nchunks / (j CUs)) chunks each: the pass lasts j * that many
                    // chunk times on the fullest CU.  10 M x 64-bit codes = 4883 chunks: 8 workgroups per CU -> 8 x 3 = 24
                    // chunk times where 4883 / 256 = 19.1 would do; 5 per CU -> 5 x 4 = 20
                    // (at most 6 workgroups of the body kernel fit a CU: 80 VGPRs, and nq * (8 W + 8) bytes of LDS each)
                    int jmax = 6;
                    const size_t lds_wg = (size_t)nq * (W * 8 + 8);
                    if (lds_wg * jmax > (size_t)160 * 1024) jmax = (int)((size_t)160 * 1024 / lds_wg);
                    if (jmax < 3) jmax = 3;
                    int best_j = jmax;
                    long long best = -1;
                    for (int j = jmax; j >= 3; --j) {
                        const long long cost = (long long)j * ((nchunks + (long long)j * cus - 1) / ((long long)j * cus));
                        if (best < 0 || cost < best) best = cost, best_j = j;
                    }
                    G = best_j * cus;
                }
                if ((long long)G > nchunks) G = (int)nchunks;  // short arrays: fewer, fuller mini-lists
            }
            long long want = 8ll * 128ll * kk / G;  // ~8x the expected fill of a mini-list
            if (fused && thr_rank < kk) want = 8ll * 128ll * thr_rank / G;   // (the tightened threshold admits ~ rank x step codes)
            u32 S = 32;
            while ((long long)S < want && S < 4096u) S <<= 1;
            SQ_TRY(s.bcnt.reserve((size_t)G * nq * 4));
            u32* bcnt = s.bcnt.as<u32>();
            // non-temporal stream for arrays far beyond the MALL (read once per call)
            const bool nt = code_bytes >= ((size_t)512 << 20);
            for (int q0 = 0; q0 < nq; q0 += qbatch) {
                const int nqc = nq - q0 < qbatch ? nq - q0 : qbatch;
                SQ_TRY(s.seg.reserve((size_t)G * nqc * S * 8));
                u64* seg = s.seg.as<u64>();
                const size_t lds = (size_t)nqc * (W * 8 + 8);
                const u64* qc = qs + (long long)q0 * W;
                if (ring) {
                    switch (W) {
                        case 1: SQ_TRY(ring_launch<1>(h, nt, G, qc, nqc, thr + q0, seg, bcnt, S, st)); break;
                        case 2: SQ_TRY(ring_launch<2>(h, nt, G, qc, nqc, thr + q0, seg, bcnt, S, st)); break;
                        case 4: SQ_TRY(ring_launch<4>(h, nt, G, qc, nqc, thr + q0, seg, bcnt, S, st)); break;
                        case 8: SQ_TRY(ring_launch<8>(h, nt, G, qc, nqc, thr + q0, seg, bcnt, S, st)); break;
                        default: SQ_TRY(ring_launch<16>(h, nt, G, qc, nqc, thr + q0, seg, bcnt, S, st)); break;
                    }
                } else if (body_kernel) {
                    if (W == 1)
                        hipLaunchKernelGGL((hamming_body_kernel<1, 8>), dim3(G), dim3(256), lds, st, h->codes, n, h->pmul, qc, nqc,
                                           nq <= HF_MAX_NQ ? hist : nullptr, bits, thr_rank, thr, seg, bcnt, S);
                    else if (W == 2)
                        hipLaunchKernelGGL((hamming_body_kernel<2, 4>), dim3(G), dim3(256), lds, st, h->codes, n, h->pmul, qc, nqc,
                                           nq <= HF_MAX_NQ ? hist : nullptr, bits, thr_rank, thr, seg, bcnt, S);
                    else
                        hipLaunchKernelGGL((hamming_body_kernel<4, 2>), dim3(G), dim3(256), lds, st, h->codes, n, h->pmul, qc, nqc,
                                           nq <= HF_MAX_NQ ? hist : nullptr, bits, thr_rank, thr, seg, bcnt, S);
                } else if (W == 1)
                    hipLaunchKernelGGL((hamming_stream_kernel<1, 8>), dim3(G), dim3(256), lds, st, h->codes, n, h->pmul, qc, nqc,
                                       thr + q0, seg, bcnt, S);
                else if (W == 2)
                    hipLaunchKernelGGL((hamming_stream_kernel<2, 4>), dim3(G), dim3(256), lds, st, h->codes, n, h->pmul, qc, nqc,
                                       thr + q0, seg, bcnt, S);
                else
                    hipLaunchKernelGGL((hamming_stream_kernel<4, 2>), dim3(G), dim3(256), lds, st, h->codes, n, h->pmul, qc, nqc,
                                       thr + q0, seg, bcnt, S);
                if (fused) {   // (nq <= qbatch: one pass)
                    c.G = G;
                    c.S = S;
                    c.map_rows = (ring || body_kernel) ? 1 : 0;   // (ring and body kernels leave physical rows in the mini-lists)
                    c.key_stride = key_stride;
                    if (prof) SQ_HIP(hipEventRecord(s.ev[2], st));
                    hipLaunchKernelGGL(hamming_pick_kernel, dim3(nq), dim3(1024), 0, st, seg, bcnt, G, nq, S, bits, k, kk, thr, (ring || body_kernel) ? 1 : 0,
                                       h->pmul, n, h->id_base, out_dist, out_idx, status, hs_dev, nq, cap, hist);
                } else
                    hipLaunchKernelGGL(hamming_compact_kernel, dim3(nqc, COMPACT_SLICES), dim3(256), 0, st, seg, bcnt, G, nqc, S,
                                       keys + (long long)q0 * key_stride, cnt + q0, cap, key_stride, ring ? 1 : 0, h->pmul, n);
            }
        } else {
            scan_dispatch(h, qs, nq, thr, keys, cnt, cap, key_stride, /*mode*/ 0, st);
        }
        if (prof && !fused) SQ_HIP(hipEventRecord(s.ev[2], st));
        c.stats.scan_launches = 1;
        c.stats.bytes_scanned = n * W * 8;
        if (!fused)
            SQ_TRY(select_launch(keys, cnt, cap, key_stride, k, nq, okeys, st, s.sort_tmp,
                                 HammingFinalize{cnt, cap, kk, h->id_base, out_dist, out_idx, status, hs_dev, nq}));
    }
    if (prof) SQ_HIP(hipEventRecord(s.ev[3], st));
    if (use_event) {
        if (!s.ev_done) SQ_HIP(hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming));
        SQ_HIP(hipEventRecord(s.ev_done, st));
    }
    c.force_fb = h->opt.force_fallback != 0;
    c.pending = true;
    return SQ_OK;
}

// Finish the call enqueued on slot `s`: wait for its kernels, collect the statistics and redo every query whose
// candidate list overflowed on the exact path (synchronously, on the call's stream).  h->stats = this call's.
static int hamming_resolve(HammingHandle* h, HammingSlot& s) {
    HammingCall& c = s.call;
    if (!c.pending) return SQ_OK;
    c.pending = false;
    const long long n = h->n;
    const int W = h->words;
    const int nq = c.nq, k = c.k;
    const int kk = (int)(k < n ? k : n);
    hipStream_t st = c.st;
    // status words and candidate counts are in pinned host memory once the call has drained (written by the
    // finalisation): the host decides whether any query needs the exact path
    SQ_HIP(c.use_event ? hamming_event_wait(s.ev_done) : stream_wait(st));
    SQ_HIP(hipGetLastError());
    h->stats = c.stats;
    if (c.prof) {
        float a = 0, b = 0;
        SQ_HIP(hipEventElapsedTime(&a, s.ev[1], s.ev[2]));
        SQ_HIP(hipEventElapsedTime(&b, s.ev[0], s.ev[3]));
        h->stats.scan_ms = a;
        h->stats.total_ms = b;
    }
    const u32* hs = reinterpret_cast<const u32*>(s.status_host.p);
    u32* cnt = s.cnt.as<u32>();
    int* thr = s.thr.as<int>();
    u64* okeys = s.out_keys.as<u64>();
    u32* status = s.status.as<u32>();
    if (c.fused) {
        bool short_bet = false;
        for (int q = 0; q < nq; ++q) short_bet = short_bet || (hs[q] & 16u) != 0;
        if (short_bet) {
            // the tightened threshold admitted fewer than k codes for some query: the whole call again through the general
            // chain (safe threshold), synchronously, on the same slot and stream
            const HammingCall again = c;
            h->no_fused = true;
            int rc = hamming_enqueue(h, s, again.qs, nq, k, again.out_dist, again.out_idx, st, false);
            h->no_fused = false;
            SQ_TRY(rc);
            SQ_TRY(hamming_resolve(h, s));
            h->stats.fallback_queries += nq;   // (counted: it cost a second pass)
            return SQ_OK;
        }
        // hamming_pick_kernel flags a query whose entries at distance <= T outnumber its sort buffer (a huge tie group): the
        // general compaction + select answers the call from the same mini-lists
        bool redo = false;
        for (int q = 0; q < nq; ++q) redo = redo || (hs[q] & 8u) != 0;
        if (redo) {
            u32* hs_dev = nullptr;
            SQ_TRY(s.status_host.device_ptr(reinterpret_cast<void**>(&hs_dev)));
            hipLaunchKernelGGL(hamming_compact_kernel, dim3(nq, COMPACT_SLICES), dim3(256), 0, st, s.seg.as<u64>(), s.bcnt.as<u32>(), c.G, nq,
                               c.S, s.keys.as<u64>(), cnt, c.cap, c.key_stride, c.map_rows, h->pmul, n);
            SQ_TRY(select_launch(s.keys.as<u64>(), cnt, c.cap, c.key_stride, k, nq, okeys, st, s.sort_tmp,
                                 HammingFinalize{cnt, c.cap, kk, h->id_base, c.out_dist, c.out_idx, status, hs_dev, nq}));
            SQ_HIP(hipStreamSynchronize(st));
            SQ_HIP(hipGetLastError());
            h->stats.scan_launches++;
        }
    }
    for (int q = 0; q < nq; ++q) h->stats.candidates += hs[nq + q];
    // exact path (candidate overflow): every key of the query, radix-selected from global memory
    const bool force_fb = c.force_fb || h->opt.force_fallback != 0;
    for (int q = 0; q < nq; ++q) {
        if (!c.small && (hs[q] != 0 || force_fb)) {
            h->stats.fallback_queries++;
            SQ_TRY(h->big_keys.reserve((size_t)n * 8));
            u64* bk = h->big_keys.as<u64>();
            hipLaunchKernelGGL(fill_u32_kernel, dim3(1), dim3(64), 0, st, cnt + q, 1ll, (u32)(n > 0xffffffffll ? 0xffffffffu : n));
            scan_dispatch(h, c.qs + (long long)q * W, 1, thr, bk, cnt + q, (u32)n, n, /*mode*/ 1, st);
            SQ_TRY(select_launch(bk, cnt + q, (u32)n, n, k, 1, okeys + (long long)q * k, st, h->fb_sort,
                                 HammingFinalize{cnt + q, (u32)n, kk, h->id_base, c.out_dist + (long long)q * k,
                                                 c.out_idx + (long long)q * k, status + q, nullptr, 0}));
            h->stats.scan_launches++;
            // (one query at a time: big_keys is shared)
            SQ_HIP(hipStreamSynchronize(st));
            SQ_HIP(hipGetLastError());
        }
    }
    return SQ_OK;
}

// Finish every asynchronous call still in flight, oldest first.
static int hamming_sync_all(HammingHandle* h) {
    for (int j = 0; j < h->depth; ++j)
        SQ_TRY(hamming_resolve(h, h->slot[(h->async_calls + (unsigned)j) % (unsigned)h->depth]));
    return SQ_OK;
}

static int hamming_search_device(HammingHandle* h, const u64* qs, int nq, int k, int* out_dist, long long* out_idx,
                                 hipStream_t st) {
    SQ_TRY(hamming_enqueue(h, h->slot[0], qs, nq, k, out_dist, out_idx, st, false));
    return hamming_resolve(h, h->slot[0]);
}

// ------------------------------------------------------------------ mutations
// LinearHashIndex.update_index / remove_from_index are a set union / difference on the host
// (impls/hash_index/linear.py:167-204).  The device copy follows without a re-upload of the whole array: the
// caller -- who owns the sorted order, row id = rank of the code -- says where the new codes rank
// (sq_hamming_append) or which ranks leave (sq_hamming_remove); new codes are appended physically, removed ones are
// filled from the tail, and an explicit rank table (u32 per code, read for survivors only) replaces the
// arithmetic permutation.
static __global__ void hamming_rank_init_kernel(u32* __restrict__ rank, long long n, RowPerm pm) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) rank[p] = orig_row(p, pm, n);
}
// pos[j] ascending = number of OLD codes smaller than new code j: old rank r moves up by #{j : pos[j] <= r};
// new code j (they arrive sorted) gets rank pos[j] + j
static __global__ void hamming_rank_insert_kernel(u32* __restrict__ rank, long long n_old, const long long* __restrict__ pos,
                                                  long long m) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n_old) {
        const long long r = rank[p];
        long long lo = 0, hi = m;  // first j with pos[j] > r
        while (lo < hi) {
            const long long mid = (lo + hi) >> 1;
            if (pos[mid] <= r) lo = mid + 1; else hi = mid;
        }
        rank[p] = (u32)(r + lo);
    } else if (p < n_old + m) {
        const long long j = p - n_old;
        rank[p] = (u32)(pos[j] + j);
    }
}
__device__ __forceinline__ long long lower_bound_ll(const long long* __restrict__ a, long long m, long long v) {
    long long lo = 0, hi = m;  // first j with a[j] >= v
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// rr ascending = ranks that leave.  Dead rows in the head [0, n - m) become holes, live rows in the tail
// [n - m, n) movers: there are as many of one as of the other.  cnt[0] holes, cnt[1] movers.
static __global__ void hamming_remove_mark_kernel(const u32* __restrict__ rank, long long n, const long long* __restrict__ rr,
                                                  long long m, u32* __restrict__ holes, u32* __restrict__ movers,
                                                  u32* __restrict__ cnt) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const long long r = rank[p];
    const long long j = lower_bound_ll(rr, m, r);
    const bool dead = j < m && rr[j] == r;
    if (p < n - m) {
        if (dead) holes[atomicAdd(&cnt[0], 1u)] = (u32)p;
    } else if (!dead) {
        movers[atomicAdd(&cnt[1], 1u)] = (u32)p;
    }
}
static __global__ void hamming_remove_move_kernel(u64* __restrict__ codes, u32* __restrict__ rank, int W,
                                                  const u32* __restrict__ holes, const u32* __restrict__ movers,
                                                  const u32* __restrict__ cnt) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt[0] || i >= cnt[1]) return;
    const long long dst = holes[i], src = movers[i];
    for (int w = 0; w < W; ++w) codes[dst * W + w] = codes[src * W + w];
    rank[dst] = rank[src];
}
static __global__ void hamming_rank_remove_kernel(u32* __restrict__ rank, long long n_new, const long long* __restrict__ rr,
                                                  long long m) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_new) return;
    const long long r = rank[p];
    rank[p] = (u32)(r - lower_bound_ll(rr, m, r));  // ranks below r that left
}

// room for `bytes` in b with its first `used` bytes kept (grown by half again at least)
static int hamming_grow_keep(DevBuf& b, size_t used, size_t need) {
    if (need <= b.cap) return SQ_OK;
    DevBuf nb;
    SQ_TRY(nb.reserve(std::max(need, used + used / 2)));
    if (used && b.p && hipMemcpy(nb.p, b.p, used, hipMemcpyDeviceToDevice) != hipSuccess) {
        nb.release();
        return fail(SQ_ERR_HIP, "device copy failed while growing the code array");
    }
    b.release();
    b = nb;
    return SQ_OK;
}

static int hamming_materialise_rank(HammingHandle* h, long long rows_needed) {
    if (!h->rank.p) {
        SQ_TRY(h->rank.reserve((size_t)std::max(rows_needed, h->n) * 4));
        hipLaunchKernelGGL(hamming_rank_init_kernel, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, 0, h->rank.as<u32>(),
                           h->n, h->pmul);
        SQ_HIP(hipDeviceSynchronize());
    } else {
        SQ_TRY(hamming_grow_keep(h->rank, (size_t)h->n * 4, (size_t)rows_needed * 4));
    }
    h->pmul.rank = h->rank.as<u32>();
    return SQ_OK;
}

}  // namespace sq

using namespace sq;

extern "C" int sq_hamming_create(const uint64_t* codes, int64_t n, int words, int mem, int64_t id_base,
                                 sq_handle_t* out) {
    if (!codes || !out || n <= 0 || words <= 0) return fail(SQ_ERR_INVALID, "sq_hamming_create: bad argument");
    if (n >= (1ll << 32)) return fail(SQ_ERR_UNSUPPORTED, "sq_hamming_create: more than 2^32-1 codes per shard");
    if (words > 64) return fail(SQ_ERR_UNSUPPORTED, "sq_hamming_create: codes wider than 4096 bits");
    auto* h = new HammingHandle();
    h->kind = H_HAMMING;
    h->n = n;
    h->words = words;
    h->id_base = id_base;
    if (hipGetDevice(&h->device) != hipSuccess) {
        delete h;
        return fail(SQ_ERR_HIP, "sq_hamming_create: no HIP device");
    }
    {
        // stride ~ n / golden ratio, coprime to n (permutation of Z_n); tiny arrays stay in caller order
        u64 mul = 1;
        if (n >= 4096 && !g_opt.hamming_no_permute) {
            mul = (u64)((double)n * 0.6180339887498949);
            auto gcd = [](u64 a, u64 b) {
                while (b) {
                    const u64 t = a % b;
                    a = b;
                    b = t;
                }
                return a;
            };
            while (mul < 2 || gcd(mul, (u64)n) != 1) ++mul;
        }
        h->pmul = RowPerm{mul, ~0ull / (u64)n, nullptr};
        const size_t bytes = (size_t)n * words * 8;
        const u64* src = reinterpret_cast<const u64*>(codes);
        DevBuf staged;
        auto bail = [&](int rc) {
            staged.release();
            delete h;
            return rc;
        };
        if (mem != SQ_MEM_DEVICE) {
            // host codes: straight into the owned buffer when no permutation is needed, else through a staging copy
            DevBuf& first = mul == 1 ? h->owned : staged;
            int rc = first.reserve(bytes);
            if (rc != SQ_OK) return bail(rc);
            hipError_t e = hipMemcpy(first.p, codes, bytes, hipMemcpyHostToDevice);
            if (e != hipSuccess) return bail(fail(SQ_ERR_HIP, "sq_hamming_create: H2D copy failed: %s", hipGetErrorString(e)));
            src = first.as<u64>();
        }
        if (mul == 1) {
            h->codes = src;  // caller order: the borrowed device array or the owned upload
        } else {
            int rc = h->owned.reserve(bytes);
            if (rc != SQ_OK) return bail(rc);
            hipLaunchKernelGGL(hamming_permute_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, src, (long long)n,
                               words, mul, h->owned.as<u64>());
            hipError_t e = hipDeviceSynchronize();
            if (e != hipSuccess) return bail(fail(SQ_ERR_HIP, "sq_hamming_create: permutation failed: %s", hipGetErrorString(e)));
            h->codes = h->owned.as<u64>();
        }
        staged.release();
    }
    *out = register_handle(h);
    return SQ_OK;
}

extern "C" int sq_hamming_search(sq_handle_t hid, const uint64_t* queries, int nq, int k, int32_t* out_dist,
                                 int64_t* out_idx, int mem, void* stream) {
    auto* h = static_cast<HammingHandle*>(lookup_handle(hid, H_HAMMING));
    if (!h) return fail(SQ_ERR_INVALID, "sq_hamming_search: unknown handle");
    if (!queries || !out_dist || !out_idx || nq <= 0 || k <= 0)
        return fail(SQ_ERR_INVALID, "sq_hamming_search: bad argument");
    std::lock_guard<std::mutex> lock(h->mu);
    h->refresh_options();
    SQ_HIP(hipSetDevice(h->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (mem == SQ_MEM_DEVICE_ASYNC) {
        // The pipelined form (see sq_dense_search): call i is enqueued on its slot's own stream before call i - 1 is
        // waited for; a call's status words are read -- and its overflowing queries redone -- `depth - 1` calls later.
        int want = h->opt.hamming_async_depth;
        want = want < 2 ? 2 : want > HammingHandle::kMaxDepth ? HammingHandle::kMaxDepth : want;
        if (want != h->depth) {  // a new depth starts from an empty pipeline
            SQ_TRY(hamming_sync_all(h));
            h->depth = want;
            h->async_calls = 0;
        }
        HammingSlot& s = h->slot[h->async_calls % (unsigned)h->depth];
        SQ_TRY(hamming_resolve(h, s));  // (the call `depth` back; normally resolved during an earlier call)
        if (!s.own) SQ_HIP(hipStreamCreateWithFlags(&s.own, hipStreamNonBlocking));
        if (h->opt.hamming_async_order) {
            if (!s.ev_in) SQ_HIP(hipEventCreateWithFlags(&s.ev_in, hipEventDisableTiming));
            SQ_HIP(hipEventRecord(s.ev_in, st));  // the caller's earlier work on `stream` (the queries) comes first
            SQ_HIP(hipStreamWaitEvent(s.own, s.ev_in, 0));
        }
        SQ_TRY(hamming_enqueue(h, s, reinterpret_cast<const u64*>(queries), nq, k, out_dist, reinterpret_cast<long long*>(out_idx),
                               s.own, true));
        h->async_calls++;
        if (!h->opt.hamming_async_wait) return SQ_OK;
        return hamming_resolve(h, h->slot[h->async_calls % (unsigned)h->depth]);
    }
    SQ_TRY(hamming_sync_all(h));
    if (mem == SQ_MEM_DEVICE) {
        return hamming_search_device(h, reinterpret_cast<const u64*>(queries), nq, k, out_dist,
                                     reinterpret_cast<long long*>(out_idx), st);
    }
    size_t qb = (size_t)nq * h->words * 8;
    SQ_TRY(h->q_dev.reserve(qb));
    SQ_TRY(h->out_dist_dev.reserve((size_t)nq * k * 4));
    SQ_TRY(h->out_idx_dev.reserve((size_t)nq * k * 8));
    SQ_TRY(h->stage.begin(qb + (size_t)nq * k * 12));
    SQ_HIP(h->stage.in(h->q_dev.p, queries, qb, st));
    SQ_TRY(hamming_search_device(h, h->q_dev.as<u64>(), nq, k, h->out_dist_dev.as<int>(),
                                 h->out_idx_dev.as<long long>(), st));
    SQ_HIP(h->stage.out(out_dist, h->out_dist_dev.p, (size_t)nq * k * 4, st));
    SQ_HIP(h->stage.out(out_idx, h->out_idx_dev.p, (size_t)nq * k * 8, st));
    SQ_HIP(stream_wait(st));
    h->stage.finish();
    return SQ_OK;
}

extern "C" int sq_hamming_sync(sq_handle_t hid) {
    auto* h = static_cast<HammingHandle*>(lookup_handle(hid, H_HAMMING));
    if (!h) return fail(SQ_ERR_INVALID, "sq_hamming_sync: unknown handle");
    std::lock_guard<std::mutex> lock(h->mu);
    h->refresh_options();
    SQ_HIP(hipSetDevice(h->device));
    return hamming_sync_all(h);
}

extern "C" int sq_hamming_append(sq_handle_t hid, const uint64_t* new_codes, int64_t m, const int64_t* insert_pos) {
    auto* h = static_cast<HammingHandle*>(lookup_handle(hid, H_HAMMING));
    if (!h) return fail(SQ_ERR_INVALID, "sq_hamming_append: unknown handle");
    if (!new_codes || !insert_pos || m <= 0) return fail(SQ_ERR_INVALID, "sq_hamming_append: bad argument");
    std::lock_guard<std::mutex> lock(h->mu);
    h->refresh_options();
    SQ_TRY(hamming_sync_all(h));
    if (!h->owned.p) return fail(SQ_ERR_UNSUPPORTED, "sq_hamming_append: the index borrows the caller's device array");
    const long long n_old = h->n, n_new = n_old + m;
    if (n_new >= (1ll << 32)) return fail(SQ_ERR_UNSUPPORTED, "sq_hamming_append: more than 2^32-1 codes per shard");
    for (int64_t j = 0; j < m; ++j)
        if (insert_pos[j] < 0 || insert_pos[j] > n_old || (j && insert_pos[j] < insert_pos[j - 1]))
            return fail(SQ_ERR_INVALID, "sq_hamming_append: insert positions must be ascending and within [0, n]");
    SQ_HIP(hipSetDevice(h->device));
    const int W = h->words;
    SQ_TRY(hamming_grow_keep(h->owned, (size_t)n_old * W * 8, (size_t)n_new * W * 8));
    h->codes = h->owned.as<u64>();
    SQ_TRY(hamming_materialise_rank(h, n_new));
    SQ_TRY(h->mut_tmp.reserve((size_t)m * 8));
    SQ_HIP(hipMemcpy(h->owned.as<u64>() + n_old * W, new_codes, (size_t)m * W * 8, hipMemcpyHostToDevice));
    SQ_HIP(hipMemcpy(h->mut_tmp.p, insert_pos, (size_t)m * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(hamming_rank_insert_kernel, dim3((unsigned)((n_new + 255) / 256)), dim3(256), 0, 0, h->rank.as<u32>(),
                       n_old, h->mut_tmp.as<long long>(), (long long)m);
    SQ_HIP(hipDeviceSynchronize());
    h->n = n_new;
    return SQ_OK;
}

extern "C" int sq_hamming_remove(sq_handle_t hid, const int64_t* ranks, int64_t m) {
    auto* h = static_cast<HammingHandle*>(lookup_handle(hid, H_HAMMING));
    if (!h) return fail(SQ_ERR_INVALID, "sq_hamming_remove: unknown handle");
    if (!ranks || m <= 0) return fail(SQ_ERR_INVALID, "sq_hamming_remove: bad argument");
    std::lock_guard<std::mutex> lock(h->mu);
    h->refresh_options();
    SQ_TRY(hamming_sync_all(h));
    if (!h->owned.p) return fail(SQ_ERR_UNSUPPORTED, "sq_hamming_remove: the index borrows the caller's device array");
    const long long n = h->n;
    if (m >= n) return fail(SQ_ERR_INVALID, "sq_hamming_remove: cannot remove every code (destroy the index instead)");
    for (int64_t j = 0; j < m; ++j)
        if (ranks[j] < 0 || ranks[j] >= n || (j && ranks[j] <= ranks[j - 1]))
            return fail(SQ_ERR_INVALID, "sq_hamming_remove: ranks must be strictly ascending and below n");
    SQ_HIP(hipSetDevice(h->device));
    SQ_TRY(hamming_materialise_rank(h, n));
    // scratch: [ranks i64 m][holes u32 m][movers u32 m][cnt u32 2]
    SQ_TRY(h->mut_tmp.reserve((size_t)m * 16 + 64));
    long long* rr = h->mut_tmp.as<long long>();
    u32* holes = reinterpret_cast<u32*>(rr + m);
    u32* movers = holes + m;
    u32* cnt = movers + m;
    SQ_HIP(hipMemcpy(rr, ranks, (size_t)m * 8, hipMemcpyHostToDevice));
    SQ_HIP(hipMemset(cnt, 0, 8));
    hipLaunchKernelGGL(hamming_remove_mark_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, h->rank.as<u32>(), n, rr,
                       (long long)m, holes, movers, cnt);
    hipLaunchKernelGGL(hamming_remove_move_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, 0, h->owned.as<u64>(),
                       h->rank.as<u32>(), h->words, holes, movers, cnt);
    hipLaunchKernelGGL(hamming_rank_remove_kernel, dim3((unsigned)((n - m + 255) / 256)), dim3(256), 0, 0, h->rank.as<u32>(),
                       n - m, rr, (long long)m);
    SQ_HIP(hipDeviceSynchronize());
    h->n = n - m;
    return SQ_OK;
}

extern "C" int sq_hamming_info(sq_handle_t hid, int64_t* out_n, int* out_words) {
    auto* h = static_cast<HammingHandle*>(lookup_handle(hid, H_HAMMING));
    if (!h) return fail(SQ_ERR_INVALID, "sq_hamming_info: unknown handle");
    std::lock_guard<std::mutex> lock(h->mu);
    if (out_n) *out_n = h->n;
    if (out_words) *out_words = h->words;
    return SQ_OK;
}

extern "C" int sq_hamming_destroy(sq_handle_t hid) {
    auto* hb = remove_handle(hid, H_HAMMING);
    if (!hb) return fail(SQ_ERR_INVALID, "sq_hamming_destroy: unknown handle");
    auto* h = static_cast<HammingHandle*>(hb);
    (void)hipSetDevice(h->device);
    {
        std::lock_guard<std::mutex> lock(h->mu);
        h->refresh_options();
        (void)hamming_sync_all(h);  // nothing of this handle is left on the device when its buffers go
    }
    delete h;
    return SQ_OK;
}
