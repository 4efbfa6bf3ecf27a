// LDS-DMA helpers shared by the streaming kernels (dense scan, ITQ filter):
// per-wave HBM -> LDS copies that bypass registers, and the counted waits that
// order a wave's own ring without a workgroup barrier.
#pragma once
#include "sq_common.hpp"

namespace sq {

// (readfirstlane: a no-op where hipcc has the wave-uniform value in scalar registers anyway; where it has moved a
// uniform computation to the vector unit, the "s" constraint alone would be handed a VGPR)
__device__ __forceinline__ const void* uniform_ptr(const void* p) {
    const u64 v = (u64)(uintptr_t)p;
    const u32 lo = __builtin_amdgcn_readfirstlane((u32)v), hi = __builtin_amdgcn_readfirstlane((u32)(v >> 32));
    return (const void*)(uintptr_t)(((u64)hi << 32) | lo);
}
// LDS-DMA: 64 lanes x 16 bytes land at lds_dst + lane*16 (wave-uniform base in
// M0); the global source is a wave-uniform 64-bit base (SGPR pair) plus a
// per-lane 32-bit byte offset.  Issued from inline asm so that hipcc does not
// fence every later ds_read with vmcnt(0); completion is tracked by the counted
// waits below (cdna_hip_programming.md section 5.7).
// NT: non-temporal cache policy for bytes that are read once (a whole-matrix stream): they do not displace what
// other kernels keep in L2 / MALL and land sooner (MI355X_MICROARCH.md, "nt-weights").  Never for data that several
// workgroups re-read from L2.
template <bool NT = false>
__device__ __forceinline__ void glds16(const void* gbase_uniform, u32 voff, u32 lds_dst) {
    u32 keep;
    if constexpr (NT)
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %3\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %2 nt\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff), "s"(uniform_ptr(gbase_uniform)), "s"(__builtin_amdgcn_readfirstlane(lds_dst))
            : "memory");
    else
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %3\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %2\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff), "s"(uniform_ptr(gbase_uniform)), "s"(__builtin_amdgcn_readfirstlane(lds_dst))
            : "memory");
}
__device__ __forceinline__ void glds4(const void* gbase_uniform, u32 voff, u32 lds_dst) {
    u32 keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dword %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(uniform_ptr(gbase_uniform)), "s"(__builtin_amdgcn_readfirstlane(lds_dst))
        : "memory");
}

// The same copies with M0 written once for several pieces: the instruction offset OFF (-4096 .. 4095) moves the LDS
// destination AND the global address, so a caller that wants the global address unshifted takes OFF off its per-lane
// offset.  Used where one wave per SIMD makes scalar issue slots count (five scalar instructions per piece in the
// self-contained form above).  Nothing else in those kernels touches M0 (gfx9 LDS instructions do not need it).
__device__ __forceinline__ void glds_set_m0(u32 lds_dst) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");
}
template <int OFF, bool NT = false>
__device__ __forceinline__ void glds16_m0(const void* gbase_uniform, u32 voff) {
    static_assert(OFF >= -4096 && OFF <= 4095, "13-bit signed instruction offset");
    if constexpr (NT)
        asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2 nt" ::"v"(voff), "s"(uniform_ptr(gbase_uniform)), "n"(OFF) : "memory");
    else
        asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(voff), "s"(uniform_ptr(gbase_uniform)), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void glds4_m0(const void* gbase_uniform, u32 voff) {
    static_assert(OFF >= -4096 && OFF <= 4095, "13-bit signed instruction offset");
    asm volatile("global_load_lds_dword %0, %1 offset:%2" ::"v"(voff), "s"(uniform_ptr(gbase_uniform)), "n"(OFF) : "memory");
}
// call f(integral_constant<int, j>) for the one j in [0, N) that equals the (compile-time foldable) argument
template <int N, int J = 0, class F>
__device__ __forceinline__ void static_for_one(int j, F&& f) {
    if constexpr (J < N) {
        if (j == J)
            f(std::integral_constant<int, J>{});
        else
            static_for_one<N, J + 1>(j, f);
    }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Allow `units` younger units to stay outstanding.  PER = DMA instructions per
// unit: 8, or 9 when every unit also carries its tile's norms (KU == 1, L2).
// With KU > 1 only a tile's first unit has the norm load: counting 8 is then
// merely conservative.
template <int NSTAGE, int PER>
__device__ __forceinline__ void wait_units_in_flight(int units) {
    if constexpr (NSTAGE >= 4) {
        if (units >= 3) {
            wait_vmcnt<3 * PER>();
            return;
        }
    }
    if constexpr (NSTAGE >= 3) {
        if (units == 2) {
            wait_vmcnt<2 * PER>();
            return;
        }
    }
    if (units == 1)
        wait_vmcnt<PER>();
    else
        wait_vmcnt<0>();
}

// A store the counted waits know about: exactly one VMEM instruction under the current exec mask.
__device__ __forceinline__ void store_u64_counted(u64* p, u64 v) {
    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}

// Allow `units` younger DMA units (PER instructions each) and the stores of `epis` younger tile epilogues
// (S store instructions each) to stay outstanding: vmcnt is one in-order queue for a wave's loads and stores.
template <int PER, int S, int EMAX, int U, int E>
__device__ __forceinline__ void wait_ops_rec(int units, int epis) {
    if (units == U && epis == E) {
        wait_vmcnt<(U * PER + E * S < 63 ? U * PER + E * S : 63)>();
        return;
    }
    if constexpr (E > 0)
        wait_ops_rec<PER, S, EMAX, U, E - 1>(units, epis);
    else if constexpr (U > 0)
        wait_ops_rec<PER, S, EMAX, U - 1, EMAX>(units, epis);
    else
        wait_vmcnt<0>();  // anything unexpected: drain
}
template <int NSTAGE, int PER, int S>
__device__ __forceinline__ void wait_ops_in_flight(int units, int epis) {
    wait_ops_rec<PER, S, NSTAGE, NSTAGE - 1, NSTAGE>(units, epis);
}

}  // namespace sq
