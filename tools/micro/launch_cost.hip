// Measurement helper: host cost of enqueueing a chain of short dependent kernels on one stream -- seven eager launches
// against one hipGraphLaunch of the same chain (captured once), three streams round-robin like the search pipeline.
// build: hipcc -O2 --offload-arch=gfx950 tools/micro/launch_cost.hip -o tools/micro/launch_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct Params { float* p; int n; int spin; };
__global__ void k_small(Params a, const Params* __restrict__ ind) {
    const Params b = ind ? *ind : a;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = b.p[i % b.n];
    for (int s = 0; s < b.spin; ++s) v = v * 1.0001f + 0.5f;
    b.p[i % b.n] = v;
}
int main(int argc, char** argv) {
    const int chain = argc > 1 ? atoi(argv[1]) : 7, iters = 3000, nstream = 3;
    float* buf; CK(hipMalloc(&buf, 1 << 22));
    Params* ind; CK(hipHostMalloc(&ind, sizeof(Params) * 8));
    std::vector<hipStream_t> st(nstream);
    for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    Params a{buf, 1 << 20, 50};
    for (int i = 0; i < 8; ++i) ind[i] = a;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto d) { return std::chrono::duration<double, std::micro>(d).count(); };
    for (int grid : {32, 256}) {
        // eager
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipDeviceSynchronize());
            auto t0 = now();
            for (int it = 0; it < iters; ++it)
                for (int c = 0; c < chain; ++c) hipLaunchKernelGGL(k_small, dim3(grid), dim3(256), 0, st[it % nstream], a, (const Params*)nullptr);
            auto t1 = now();
            CK(hipDeviceSynchronize());
            auto t2 = now();
            if (rep) printf("grid %3d eager : host %.2f us per chain of %d (%.2f per launch), wall %.2f us per chain\n", grid, us(t1 - t0) / iters, chain,
                            us(t1 - t0) / iters / chain, us(t2 - t0) / iters);
        }
        // graph per stream
        std::vector<hipGraphExec_t> ex(nstream);
        for (int s = 0; s < nstream; ++s) {
            hipGraph_t g;
            CK(hipStreamBeginCapture(st[s], hipStreamCaptureModeThreadLocal));
            for (int c = 0; c < chain; ++c) hipLaunchKernelGGL(k_small, dim3(grid), dim3(256), 0, st[s], a, (const Params*)(ind + s));
            CK(hipStreamEndCapture(st[s], &g));
            CK(hipGraphInstantiate(&ex[s], g, nullptr, nullptr, 0));
            CK(hipGraphDestroy(g));
        }
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipDeviceSynchronize());
            auto t0 = now();
            for (int it = 0; it < iters; ++it) CK(hipGraphLaunch(ex[it % nstream], st[it % nstream]));
            auto t1 = now();
            CK(hipDeviceSynchronize());
            auto t2 = now();
            if (rep) printf("grid %3d graph : host %.2f us per chain of %d, wall %.2f us per chain\n", grid, us(t1 - t0) / iters, chain, us(t2 - t0) / iters);
        }
        for (auto e : ex) CK(hipGraphExecDestroy(e));
    }
    return 0;
}
