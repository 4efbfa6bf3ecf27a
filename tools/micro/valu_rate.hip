// Measurement helper (not part of the product): issue cost of the vector instructions of the Hamming inner loop
// (v_xor_b32, v_bcnt_u32_b32, v_min3_u32, v_readlane_b32) in cycles per wave-instruction per SIMD, with 1, 2, 4 and 8
// waves per SIMD on every CU.   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o tools/micro/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// MODE 0: v_xor_b32, 1: v_bcnt_u32_b32 (independent), 2: v_bcnt chained through the accumulate operand, 3: v_min3_u32,
// 4: the loop's mix (2 xor + 2 bcnt chained + 1/2 min3), 5: v_add_u32, 6: the mix with the queries from v_readlane
template <int MODE>
__global__ void k(unsigned* out, int iters, unsigned seed, long long* cycles) {
    unsigned r[16];
    for (int i = 0; i < 16; ++i) r[i] = seed * (i + 1) + threadIdx.x;
    unsigned q0 = seed ^ 0x12345u, q1 = seed * 3u;
    const long long t0 = wall_clock64();
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(r[i]) : "s"(q0));
            } else if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 1) & 15]));
            } else if (MODE == 2) {
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(r[i]) : "v"(r[i + 1]));
                    asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(r[i]) : "v"(r[i + 1]));
                }
            } else if (MODE == 3) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(r[(i + 1) & 15]), "v"(r[(i + 2) & 15]));
            } else if (MODE == 4 || MODE == 6) {
                unsigned a0 = q0, a1 = q1;
                if (MODE == 6) {
                    asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(a0) : "v"(r[15]), "s"(u));
                    asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(a1) : "v"(r[14]), "s"(u));
                }
                unsigned d[4], t;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(t) : "s"(a0), "v"(r[2 * i]));
                    asm volatile("v_bcnt_u32_b32 %0, %1, 0" : "=v"(d[i]) : "v"(t));
                    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(t) : "s"(a1), "v"(r[2 * i + 1]));
                    asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(d[i]) : "v"(t));
                }
                asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(r[12]) : "v"(d[0]), "v"(d[1]));
                asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(r[12]) : "v"(d[2]), "v"(d[3]));
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(r[i]) : "v"(r[(i + 1) & 15]));
            }
        }
    }
    const long long c1 = clock64();
    const long long t1 = wall_clock64();
    unsigned s = 0;
    for (int i = 0; i < 16; ++i) s += r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        cycles[0] = c1 - c0;
        cycles[1] = t1 - t0;
    }
}

template <int MODE>
static void run(const char* name, int per_iter, unsigned* out, long long* cyc) {
    for (int waves_per_simd : {1, 2, 4, 8}) {
        const int threads = 256, blocks = 256 * waves_per_simd;   // 4 waves per workgroup, one workgroup per SIMD-wave slot
        const int iters = 2000;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        k<MODE><<<blocks, threads>>>(out, 10, 1u, cyc);
        hipEventRecord(e0);
        k<MODE><<<blocks, threads>>>(out, iters, 1u, cyc);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        long long h[2];
        hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
        const double n_inst = (double)iters * 4 * per_iter;
        printf("%-28s %d waves/SIMD: %7.3f ms, %6.2f shader cycles per wave-instruction of one wave, %6.2f per SIMD (clock64 %lld, wall_clock64 %lld)\n",
               name, waves_per_simd, ms, (double)h[0] / n_inst, (double)h[0] / n_inst / waves_per_simd, h[0], h[1]);
    }
}

int main() {
    unsigned* out;
    long long* cyc;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    hipMalloc(&cyc, 16);
    run<0>("v_xor_b32 (sgpr operand)", 16, out, cyc);
    run<5>("v_add_u32", 16, out, cyc);
    run<1>("v_bcnt_u32_b32 independent", 16, out, cyc);
    run<2>("v_bcnt_u32_b32 chained", 16, out, cyc);
    run<3>("v_min3_u32", 16, out, cyc);
    run<4>("hamming mix (18 instr)", 18, out, cyc);
    run<6>("hamming mix + 2 readlane", 20, out, cyc);
    return 0;
}
