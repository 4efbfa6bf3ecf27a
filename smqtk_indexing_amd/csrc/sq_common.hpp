// Internal helpers shared by the libsmqtk_hip translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/smqtk_hip.h"

typedef unsigned long long u64;
typedef unsigned int u32;

namespace sq {

// ------------------------------------------------------------------ errors
extern thread_local char g_err[512];
inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define SQ_HIP(expr)                                                                  \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess)                                                         \
            return sq::fail(e_ == hipErrorOutOfMemory ? SQ_ERR_NOMEM : SQ_ERR_HIP,    \
                            "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
                            __FILE__, __LINE__);                                      \
    } while (0)

#define SQ_TRY(expr)              \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != SQ_OK) return rc_; \
    } while (0)

// ----------------------------------------------------------------- options
struct Options {
    int profile = 0;         // 1: hipEvent timing of the search stages into sq_stats_t; N > 1: only every N-th asynchronous dense search is timed (the others report scan_ms = 0)
    int sample_stride = 0;   // 0 = auto
    int candidate_cap = 0;   // 0 = auto
    int force_fallback = 0;
    int dense_stages = 0;    // 0 = auto (LDS ring depth of the dense scan)
    int dense_blocks = 0;    // 0 = auto (row blocks of the dense scan grid)
    int dense_sample_blocks = 0;  // pipelined one-group searches: workgroups of the SAMPLE pass; 0 = auto (the CUs the full pass leaves free), -1 = as many as the full pass, N = N
    int dense_debug = 0;     // measurement only (see DenseScanArgs::debug)
    int dense_waves = 0;     // 0 = auto, 4 or 8 waves per scan workgroup
    int dense_qt = 0;        // 0 = auto, 1 / 2 / 4 query tiles per scan wave
    int itq_exact = 0;       // 1 = every row through the float64 ITQ kernel (no bf16 filter)
    int dense_qplanes = 0;       // 0 = auto, 2 = keep q_hi + q_lo also in the multi-tile scan
    int dense_no_center = 0;     // 1 = the dense L2 filter scores the rows as given (no column-mean origin; measurement)
    int dense_rerank_segments = 0;  // survivor segments per re-rank workgroup (0 = all waves of a scan workgroup; measurement)
    int merge_threads = 0;       // host threads of the shard merge (0 = by size)
    int spin_wait_us = 2000;     // searches poll the stream this long before blocking (0 = block at once)
    int dense_async_streams = 2; // asynchronous dense searches: 2 = the two call slots run on streams of their own (tail of call i overlaps the head of call i + 1), 1 = everything on the caller's stream
    int dense_async_depth = 2;   // asynchronous dense searches in flight (2..4): the results of a call are final when the (depth - 1)-th call after it returns
    int dense_async_wait = 1;    // 1: an asynchronous dense search returns once the oldest call in flight is final; 0: it returns right after enqueueing (the wait moves to the start of the next call: one more call of lag, host work between calls overlaps the device)
    int dense_async_order = 1;   // 1: an asynchronous dense search on internal streams starts behind the work already on the caller's stream (an event per call); 0: the caller guarantees its queries are complete -- no event, the call starts as soon as the device has room
    int dense_nt = -1;           // non-temporal LDS-DMA of the dense scan's row stream (launches of one query group): -1 = automatic, 0 = never, 1 = every byte
    int dense_nt_keep_mb = 0;    // one-tile dense scan, automatic mode: MB at the head of the scan copy that keep the default cache policy (0 = 192)
    int hamming_no_permute = 0;  // 1 = keep the Hamming code array in caller order on the device (measurement)
    int dense_fused_prep = 1;    // 1 = L2 searches of one query tile build the query planes inside the scan kernels (no prep launch); 0 = dense_prep_queries_kernel
    int dense_int8_batch = 64;   // largest batch the int8 filter takes (33 .. 64: two query tiles per wave, 128-byte rows; up to 256: four tiles, which only ties with the bf16 kernels; 32 = one tile only)
    int dense_fused = 1;         // int8 calls of one query tile as three launches (head: query prep + sample pass + threshold by the last workgroup; full pass with the re-rank as its tail; select) instead of six; 0 = the six-launch chain
    int dense_tighten = 1;       // fused int8 calls: the full pass histograms its entries' scores and its tail re-ranks only those under the tightened threshold (sq_dense_i8.hpp); rows beyond 512 dimensions: the second-level threshold of sq_dense_tighten.hpp between the pass and the re-rank; 0 = every entry is re-ranked (measurement)
    int dense_graph = 1;         // pipelined int8 calls: the call's kernels as one captured graph launch (0 = eager launches)
    int dense_int8 = -1;         // int8 first-stage filter (L2, d <= 128, one query tile): -1 = automatic, 0 = never (bf16 filter), 1 = whenever the copy exists
    int dense_mid_tier = 1;      // 1 = queries the first filter could not certify get a second, tighter filter pass (L2 and cosine: bf16 planes of the float32 rows built on the fly) before the exact all-rows path, and calls start there once the first filter's lists overflow call after call; 0 = straight to the exact path
    int hamming_async_depth = 2; // asynchronous Hamming searches in flight (2..4), as dense_async_depth
    int hamming_async_wait = 1;  // as dense_async_wait
    int hamming_async_order = 1; // as dense_async_order
    int hamming_ring = -1;       // Hamming stream kernel: -1 = automatic (LDS-DMA ring for arrays beyond the MALL), 0 = register loads, 1 = ring
    int hamming_tighten = 1;     // fused Hamming searches: stream threshold from a lower sample rank than k (far fewer candidates); a bet the pick kernel checks -- a lost one redoes the call with the safe rule; 0 = the safe rank-k rule
    int hamming_fused = 1;       // Hamming searches of up to 32 queries (64 .. 256-bit codes, k <= 2048) as three launches (head: sampled histogram + thresholds by the last workgroup; stream; pick: exact k-th distance + gather + sort per query) instead of a memset and five; 0 = the general chain
};
extern Options g_opt;   // process-wide defaults (sq_set_option); a handle may override any of them (sq_handle_set_option)
// name -> member, shared by sq_set_option and sq_handle_set_option (sq_core.hip)
int Options::*option_member(const char* name);

// Wait for a stream the way a latency-bound caller wants to: poll hipStreamQuery for a while (a search is
// a fraction of a millisecond; the blocking wait's wake-up costs tens of microseconds of it), then block.
// Option "spin_wait_us" (default 2000; 0 = always block).
inline hipError_t stream_wait(hipStream_t st) {
    const long long budget_us = g_opt.spin_wait_us;
    if (budget_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            for (int i = 0; i < 64; ++i) {
                const hipError_t e = hipStreamQuery(st);
                if (e != hipErrorNotReady) return e;
            }
            if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > budget_us)
                break;
        }
    }
    return hipStreamSynchronize(st);
}

// ------------------------------------------------------- device buffer (RAII)
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return SQ_OK;
        if (p) {
            (void)hipFree(p);
            p = nullptr;
            cap = 0;
        }
        size_t want = bytes + (bytes >> 3) + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(SQ_ERR_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        }
        cap = want;
        return SQ_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

struct HostPinned {
    void* p = nullptr;
    size_t cap = 0;
    void* dev = nullptr;   // the block's device address (looked up once per allocation: device_ptr)
    int reserve(size_t bytes) {
        if (bytes <= cap) return SQ_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        dev = nullptr;
        cap = 0;
        hipError_t e = hipHostMalloc(&p, bytes + 256, hipHostMallocDefault);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(SQ_ERR_NOMEM, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        }
        cap = bytes + 256;
        return SQ_OK;
    }
    // device address of the block (a runtime call of a few microseconds: cached per allocation)
    int device_ptr(void** out) {
        if (!dev) {
            hipError_t e = hipHostGetDevicePointer(&dev, p, 0);
            if (e != hipSuccess) {
                dev = nullptr;
                return fail(SQ_ERR_HIP, "hipHostGetDevicePointer failed: %s", hipGetErrorString(e));
            }
        }
        *out = dev;
        return SQ_OK;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        dev = nullptr;
        cap = 0;
    }
};

// Small host buffers travel through pinned staging: an "asynchronous" copy from or to pageable memory is a
// blocking staged copy inside the runtime (~10-15 us each; a one-query search makes three to five of them).
// in(): memcpy into the pinned block, asynchronous H2D from there.  out(): asynchronous D2H into the block;
// finish() -- after the stream has drained -- copies the pieces to the caller's buffers.  Calls above
// kStageMax bytes in total copy directly.
struct PinnedStage {
    static constexpr size_t kStageMax = 1u << 20;
    HostPinned pin;
    size_t used = 0;
    bool on = false;
    struct Out {
        void* dst;
        size_t off, bytes;
    };
    Out outs[4];
    int n_out = 0;
    int begin(size_t total_bytes) {
        used = 0;
        n_out = 0;
        on = total_bytes <= kStageMax;
        return on ? pin.reserve(total_bytes + 64 * 8) : SQ_OK;
    }
    hipError_t in(void* dev, const void* host, size_t bytes, hipStream_t st) {
        if (!on) return hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, st);
        char* at = static_cast<char*>(pin.p) + used;
        memcpy(at, host, bytes);
        used += (bytes + 63) / 64 * 64;
        return hipMemcpyAsync(dev, at, bytes, hipMemcpyHostToDevice, st);
    }
    hipError_t out(void* host, const void* dev, size_t bytes, hipStream_t st) {
        if (!on || n_out >= 4) return hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st);
        outs[n_out++] = Out{host, used, bytes};
        char* at = static_cast<char*>(pin.p) + used;
        used += (bytes + 63) / 64 * 64;
        return hipMemcpyAsync(at, dev, bytes, hipMemcpyDeviceToHost, st);
    }
    void finish() {  // the stream has been waited for
        for (int i = 0; i < n_out; ++i) memcpy(outs[i].dst, static_cast<char*>(pin.p) + outs[i].off, outs[i].bytes);
        n_out = 0;
    }
    void release() { pin.release(); }
};

// ------------------------------------------------------------------ handles
enum HandleKind { H_HAMMING = 1, H_DENSE = 2, H_ROWS = 3, H_FIT = 4, H_ITQ = 5 };

struct HandleBase {
    int kind = 0;
    int device = 0;
    std::mutex mu;
    sq_stats_t stats{};
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    // Options of THIS handle: the process-wide values with the handle's own overrides on top, re-read under the
    // handle's lock at every API entry (refresh_options).  Two indexes searched from two threads therefore keep their
    // own pipeline depth, candidate lists, profiling ... (the reference's contract: "implementations should be
    // thread safe", interfaces/nearest_neighbor_index.py:22-23).
    std::vector<std::pair<int Options::*, int>> overrides;
    Options opt;
    void refresh_options() {
        opt = g_opt;
        for (const auto& o : overrides) opt.*(o.first) = o.second;
    }
    virtual ~HandleBase() {
        for (auto& e : ev)
            if (e) (void)hipEventDestroy(e);
    }
};

sq_handle_t register_handle(HandleBase* h);
HandleBase* lookup_handle(sq_handle_t id, int kind);
HandleBase* remove_handle(sq_handle_t id, int kind);

inline int cu_count(int device) {
    static int cached[64] = {0};
    if (device >= 0 && device < 64 && cached[device]) return cached[device];
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) return 256;
    if (device >= 0 && device < 64) cached[device] = p.multiProcessorCount;
    return p.multiProcessorCount;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device's copy of a kernel: once per (kernel,
// device), the mask being the caller's static state for that kernel (a second GPU used from the same process otherwise
// never gets the attribute and its launches with more than 64 KiB of LDS fail).
inline int ensure_dyn_lds(const void* fn, int bytes, std::atomic<unsigned long long>& done_devices) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const unsigned long long bit = dev >= 0 && dev < 64 ? 1ull << dev : 0ull;
    if (bit && (done_devices.load(std::memory_order_relaxed) & bit)) return SQ_OK;
    SQ_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    if (bit) done_devices.fetch_or(bit, std::memory_order_relaxed);
    return SQ_OK;
}

inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace sq
