// Measurement helper (not part of the product): sustained rate of v_mfma_f32_32x32x16_{f16,bf16} with the B operand in
// VGPRs or in AGPRs, one wave per SIMD on every CU.   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, float seed) {
    f32x16 acc0, acc1, acc2, acc3;
    for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = acc2[i] = acc3[i] = 0.f;
    f32x4 a = {seed + threadIdx.x, seed * 2, seed * 3, seed * 5};
    f32x4 b0 = {seed * 7, seed + 1, seed + 2, seed + 3}, b1 = {seed, seed * 11, seed - 1, seed - 2};
    f32x4 ba0, ba1;
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(ba0[0]) : "v"(b0[0]));
    // (the four components one by one: there is no 128-bit move into AGPRs)
    ba0 = b0; ba1 = b1;
    asm volatile("" : "+a"(ba0), "+a"(ba1));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) {   // builtin, f16, everything in VGPRs
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b0), acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b1), acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b0), acc2, 0, 0, 0);
            } else if (MODE == 1) {   // asm, f16, B in AGPRs
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "a"(ba0));
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "a"(ba1));
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc2) : "v"(a), "a"(ba0));
            } else if (MODE == 2) {   // asm, bf16, B in AGPRs
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "a"(ba0));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "a"(ba1));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc2) : "v"(a), "a"(ba0));
            } else {   // asm, f16, B in VGPRs
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b0));
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b1));
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc2) : "v"(a), "v"(b0));
            }
        }
    }
    asm volatile("s_nop 15" : "+v"(acc0), "+v"(acc1), "+v"(acc2));
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + acc2[i] + acc3[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// MODE 4: the wide ITQ kernel's register pattern -- 64 resident B fragments in AGPRs (random float16 bits), A fragments
// alternating between two VGPR tuples, four accumulators; MODE 5: the same plus two ds_read_b128 per three MFMAs.
template <int MODE>
__global__ __launch_bounds__(256, 1) void k2(float* out, int iters, const f32x4* __restrict__ rnd) {
    __shared__ f32x4 lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = rnd[i];
    __syncthreads();
    f32x16 acc0, acc1, acc2, acc3;
    for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = acc2[i] = acc3[i] = 0.f;
    f32x4 ba[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        const unsigned addr = (unsigned)(size_t)&lds[(i * 64 + threadIdx.x) & 2047];
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=a"(ba[i]) : "v"(addr) : "memory");
    }
    f32x4 a0 = rnd[threadIdx.x], a1 = rnd[threadIdx.x + 256];
    const f32x4* lp = &lds[threadIdx.x & 63];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            f32x4 x0 = a0, x1 = a1;
            if (MODE == 5) {
                x0 = lp[(u * 2 * 64) & 2047];
                x1 = lp[((u * 2 + 1) * 64) & 2047];
            }
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc0) : "v"(x0), "a"(ba[2 * u]));
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc1) : "v"(x0), "a"(ba[2 * u + 1]));
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc2) : "v"(x1), "a"(ba[2 * u + 1]));
        }
    }
    asm volatile("s_nop 15" : "+v"(acc0), "+v"(acc1), "+v"(acc2));
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + acc2[i] + acc3[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static void run2(const char* name, float* out, const f32x4* rnd) {
    const int iters = 1000;   // 96 MFMAs per iteration per wave
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k2<MODE>), dim3(256), dim3(256), 0, 0, out, 10, rnd);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k2<MODE>), dim3(256), dim3(256), 0, 0, out, iters, rnd);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)iters * 96;
    printf("%-34s %8.3f ms  %6.1f ns per MFMA per wave; %6.1f TFLOP/s\n", name, ms, ms * 1e6 / mfmas, mfmas * 1024 * 32768.0 / (ms * 1e-3) / 1e12);
}

template <int MODE>
static void run(const char* name, float* out, float seed) {
    const int iters = 4000;   // 24 MFMAs per iteration per wave
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, out, 100, seed);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, out, iters, seed);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)iters * 24;
    printf("%-34s %8.3f ms  %6.1f ns per MFMA per wave = %5.1f cycles at 2.4 GHz; %6.1f TFLOP/s (seed %.3g)\n", name, ms, ms * 1e6 / mfmas,
           ms * 1e6 / mfmas * 2.4, mfmas * 1024 * 32768.0 / (ms * 1e-3) / 1e12, seed);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 256 * 4);
    for (float seed : {0.0f, 1.37f}) {   // all-zero operands vs toggling bits (power)
        run<0>("builtin f16, VGPR operands", out, seed);
        run<3>("asm f16, B in VGPRs", out, seed);
        run<1>("asm f16, B in AGPRs", out, seed);
        run<2>("asm bf16, B in AGPRs", out, seed);
    }
    // random float16 bit patterns of moderate magnitude (exponent field 0x3c00 +- a few): the power the real data draws
    unsigned* h = new unsigned[2048 * 4];
    unsigned st = 12345u;
    for (int i = 0; i < 2048 * 4; ++i) {
        st = st * 1664525u + 1013904223u;
        const unsigned lo = 0x3800u + ((st >> 8) & 0x7ffu) + (((st >> 20) & 1u) << 15), hi = 0x3800u + ((st >> 3) & 0x7ffu) + (((st >> 21) & 1u) << 15);
        h[i] = lo | (hi << 16);
    }
    f32x4* rnd;
    hipMalloc(&rnd, 2048 * 16);
    hipMemcpy(rnd, h, 2048 * 16, hipMemcpyHostToDevice);
    run2<4>("64 AGPR fragments, random data", out, rnd);
    run2<5>("  + 2 ds_read_b128 per 3 MFMAs", out, rnd);
    return 0;
}
