// ITQ model training on the device (gfx950): the three O(n) products of
// ItqFunctor.fit (smqtk_indexing/impls/lsh_functor/itq.py:291-387) and
// _find_itq_rotation (itq.py:239-289).
//
//   mean   = column means of norm(x)                      (itq.py:330-331)
//   cov    = np.cov((norm(x) - mean).T)            d x d   (itq.py:337)
//   v      = (norm(x) - mean) . pc_top             n x b   (itq.py:362)
//   per iteration:  ux = sign(v . r),  c = ux^T . v  b x b (itq.py:271-275)
//
// The d x d eigen-decomposition and the b x b SVD of every iteration stay on
// the host (numpy / LAPACK, as in the reference): they are tiny.  The descriptor
// matrix and v stay on the device for the whole fit.  float64 throughout, on
// the vector unit: MI355X's vector and matrix FP64 peaks are the same 78.6
// TFLOP/s, and the products are reductions over n with small outputs, which a
// register-tiled outer-product kernel fed from LDS handles without the MFMA
// operand shuffles.  The result is not bit-reproducible against numpy (neither
// is numpy across LAPACK builds): parity is the
// agreement of covariance / projection / per-iteration c to rounding and of the
// final codes (tests/test_hip_plugins.py::test_itq_fit_on_device_matches_host).
#include <algorithm>
#include <vector>

#include "sq_common.hpp"
#include "sq_pairwise.hpp"

namespace sq {

struct FitHandle : HandleBase {
    const void* x = nullptr;  // device [n][d] of dtype
    DevBuf owned;
    int dtype = SQ_DTYPE_F32;
    long long n = 0;
    int d = 0, norm = SQ_NORM_NONE;
    int bits = 0;
    DevBuf nrm;     // double [n]: |x| (normalize=2; 1 for zero rows)
    DevBuf mean;    // double [d]
    DevBuf acc;     // double scratch for the reductions (max(d*d, b*b))
    DevBuf v;       // double [n][bits]
    DevBuf small;   // pc / r on the device
    DevBuf codes;   // sign bits of v . r of the current iteration
    DevBuf zero;    // a zero mean for the hash kernel
    ~FitHandle() override {
        for (DevBuf* b : {&owned, &nrm, &mean, &acc, &v, &small, &codes, &zero}) b->release();
    }
};

// element (row, k) of norm(x) as float64
template <class T>
__device__ __forceinline__ double fit_elem(const T* __restrict__ x, long long row, int d, int k, const double* __restrict__ nrm) {
    const double v = (double)x[row * d + k];
    return nrm ? v / nrm[row] : v;
}

template <class T>
__global__ __launch_bounds__(256) void fit_rownorm_kernel(const T* __restrict__ x, long long n, int d, double* __restrict__ nrm) {
    const int j8 = threadIdx.x & 7;
    const long long row = (long long)blockIdx.x * 32 + (threadIdx.x >> 3);
    const long long r = row < n ? row : n - 1;
    const T* xr = x + r * d;
    auto term = [xr](int i) { return (double)xr[i] * (double)xr[i]; };
    const double s = np_pairwise_sum<double>(term, d, j8);
    if (row < n && j8 == 0) {
        const double v = sqrt(s);
        nrm[row] = v == 0.0 ? 1.0 : v;
    }
}

// column sums of norm(x) over a slab of rows per workgroup
template <class T>
__global__ __launch_bounds__(256) void fit_colsum_kernel(const T* __restrict__ x, long long n, int d, long long rows_per_block,
                                                          const double* __restrict__ nrm, double* __restrict__ colsum) {
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = std::min(r0 + rows_per_block, n);
    for (int k = threadIdx.x; k < d; k += 256) {
        double acc = 0.0;
        for (long long r = r0; r < r1; ++r) acc += fit_elem(x, r, d, k, nrm);
        atomicAdd(&colsum[k], acc);
    }
}

// out[P][Q] += sum over the workgroup's rows of a_row (x) b_row.  The output is cut into tiles of GT x GT
// (grid.y x grid.z), one workgroup per (row slab, tile): the slices of the two operands' rows that a tile needs are
// staged in LDS (RB rows at a time); a thread owns a TP x TQ register tile of the output tile (256 threads x 8 x 8 =
// 128 x 128).  Round 1 had one tile, i.e. P, Q <= 128; tiling covers 512-wide descriptors (BASELINE config 4, the
// reference's own 256-d test) and 256-bit codes.
//   MODE 0: a = b = norm(x) - mean                (covariance; P = Q = d)
//   MODE 1: a = sign bits of the row's code as +-1, b = v   (ITQ iteration; P = Q = bits)
static constexpr int FIT_GT = 128;
template <class T, int MODE, int TP, int TQ>
__global__ __launch_bounds__(256) void fit_gram_kernel(const T* __restrict__ x, const double* __restrict__ nrm,
                                                        const double* __restrict__ mean, const double* __restrict__ v,
                                                        const u64* __restrict__ codes, int words, long long n, int P, int Q,
                                                        long long rows_per_block, double* __restrict__ out) {
    constexpr int RB = 16;
    const int pb = blockIdx.y * FIT_GT, qb = blockIdx.z * FIT_GT;                 // this workgroup's output tile
    const int PW = P - pb < FIT_GT ? P - pb : FIT_GT, QW = Q - qb < FIT_GT ? Q - qb : FIT_GT;   // its extent
    extern __shared__ double s_ab[];  // [RB][PW] then [RB][QW]
    double* s_a = s_ab;
    double* s_b = s_ab + RB * FIT_GT;
    const int tq = threadIdx.x % ((QW + TQ - 1) / TQ), tp = threadIdx.x / ((QW + TQ - 1) / TQ);
    const int p0 = tp * TP, q0 = tq * TQ;
    const bool live = p0 < PW && q0 < QW;
    double acc[TP][TQ];
#pragma unroll
    for (int i = 0; i < TP; ++i)
#pragma unroll
        for (int j = 0; j < TQ; ++j) acc[i][j] = 0.0;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = std::min(r0 + rows_per_block, n);
    for (long long rb = r0; rb < r1; rb += RB) {
        const int nr = (int)std::min<long long>(RB, r1 - rb);
        __syncthreads();
        if (MODE == 0) {
            for (int e = threadIdx.x; e < RB * PW; e += 256) {
                const int r = e / PW, k = e - r * PW;
                s_a[e] = r < nr ? fit_elem(x, rb + r, P, pb + k, nrm) - mean[pb + k] : 0.0;
            }
            for (int e = threadIdx.x; e < RB * QW; e += 256) {
                const int r = e / QW, k = e - r * QW;
                s_b[e] = r < nr ? fit_elem(x, rb + r, Q, qb + k, nrm) - mean[qb + k] : 0.0;
            }
        } else {
            for (int e = threadIdx.x; e < RB * PW; e += 256) {
                const int r = e / PW, k = e - r * PW;
                // bit pb + k of the row's code, MSB first in right-aligned words (sq_itq_hash layout)
                const int pos = words * 64 - P + pb + k;
                double val = 0.0;
                if (r < nr) val = ((codes[(rb + r) * words + (pos >> 6)] >> (63 - (pos & 63))) & 1ull) ? 1.0 : -1.0;
                s_a[e] = val;
            }
            for (int e = threadIdx.x; e < RB * QW; e += 256) {
                const int r = e / QW, k = e - r * QW;
                s_b[e] = r < nr ? v[(rb + r) * Q + qb + k] : 0.0;
            }
        }
        __syncthreads();
        if (live) {
            for (int r = 0; r < RB; ++r) {
                double av[TP], bv[TQ];
#pragma unroll
                for (int i = 0; i < TP; ++i) av[i] = p0 + i < PW ? s_a[r * PW + p0 + i] : 0.0;
#pragma unroll
                for (int j = 0; j < TQ; ++j) bv[j] = q0 + j < QW ? s_b[r * QW + q0 + j] : 0.0;
#pragma unroll
                for (int i = 0; i < TP; ++i)
#pragma unroll
                    for (int j = 0; j < TQ; ++j) acc[i][j] = fma(av[i], bv[j], acc[i][j]);
            }
        }
    }
    if (live) {
#pragma unroll
        for (int i = 0; i < TP; ++i)
#pragma unroll
            for (int j = 0; j < TQ; ++j)
                if (p0 + i < PW && q0 + j < QW) atomicAdd(&out[(long long)(pb + p0 + i) * Q + qb + q0 + j], acc[i][j]);
    }
}

// v[row][c0..c0+bc) = (norm(x_row) - mean) . pc[:, c0..c0+bc): the columns of pc go through LDS a chunk at a time
// (grid.y = chunk: [d][bc] doubles), the rows 32 at a time in k-slices of FIT_KC elements ([32][FIT_KC] doubles); a thread
// owns one row and up to 16 of the chunk's columns (bc <= 128), accumulated in registers across the k-slices.
static constexpr int FIT_KC = 128;
template <class T>
__global__ __launch_bounds__(256) void fit_project_kernel(const T* __restrict__ x, const double* __restrict__ nrm,
                                                           const double* __restrict__ mean, const double* __restrict__ pc,
                                                           long long n, int d, int b, int bc, double* __restrict__ v) {
    extern __shared__ double s_pc[];  // [d][bc] then [32][FIT_KC]
    double* s_x = s_pc + (size_t)d * bc;
    const int c0 = blockIdx.y * bc;
    const int cw = b - c0 < bc ? b - c0 : bc;
    for (int e = threadIdx.x; e < d * cw; e += 256) {
        const int k = e / cw, j = e - k * cw;
        s_pc[k * bc + j] = pc[(long long)k * b + c0 + j];
    }
    const int r = threadIdx.x >> 3, c8 = threadIdx.x & 7;
    for (long long rb = (long long)blockIdx.x * 32; rb < n; rb += (long long)gridDim.x * 32) {
        double acc[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = 0.0;
        for (int k0 = 0; k0 < d; k0 += FIT_KC) {
            const int kw = d - k0 < FIT_KC ? d - k0 : FIT_KC;
            __syncthreads();
            for (int e = threadIdx.x; e < 32 * kw; e += 256) {
                const int rr = e / kw, k = e - rr * kw;
                s_x[rr * FIT_KC + k] = rb + rr < n ? fit_elem(x, rb + rr, d, k0 + k, nrm) - mean[k0 + k] : 0.0;
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int j = c8 + 8 * t;
                if (j < cw) {
                    double a = acc[t];
                    for (int k = 0; k < kw; ++k) a = fma(s_x[r * FIT_KC + k], s_pc[(k0 + k) * bc + j], a);
                    acc[t] = a;
                }
            }
        }
        if (rb + r < n) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int j = c8 + 8 * t;
                if (j < cw) v[(rb + r) * b + c0 + j] = acc[t];
            }
        }
    }
}

static constexpr int FIT_MAX_D = 512, FIT_MAX_BITS = 256;

template <class T>
static int fit_gram_cov(FitHandle* h, double* out_dev) {
    const int d = h->d;
    const long long rpb = 2048;
    const unsigned grid = (unsigned)((h->n + rpb - 1) / rpb);
    const unsigned tiles = (unsigned)((d + FIT_GT - 1) / FIT_GT);
    const size_t lds = (size_t)2 * 16 * FIT_GT * 8;
    const double* nrm = h->norm == SQ_NORM_L2 ? h->nrm.as<double>() : nullptr;
    if (d > FIT_MAX_D) return fail(SQ_ERR_UNSUPPORTED, "sq_itqfit: d=%d above %d", d, FIT_MAX_D);
    // thread tiles: 256 threads x (TP x TQ) cover an output tile (4 x 4 when the whole output is <= 64 x 64)
    if (d <= 64)
        hipLaunchKernelGGL((fit_gram_kernel<T, 0, 4, 4>), dim3(grid, tiles, tiles), dim3(256), lds, 0, (const T*)h->x, nrm,
                           h->mean.as<double>(), nullptr, nullptr, 0, h->n, d, d, rpb, out_dev);
    else
        hipLaunchKernelGGL((fit_gram_kernel<T, 0, 8, 8>), dim3(grid, tiles, tiles), dim3(256), lds, 0, (const T*)h->x, nrm,
                           h->mean.as<double>(), nullptr, nullptr, 0, h->n, d, d, rpb, out_dev);
    SQ_HIP(hipGetLastError());
    return SQ_OK;
}

}  // namespace sq

using namespace sq;

extern "C" int sq_itq_hash(const void* x, int x_dtype, int64_t n, int d, const double* mean, int mean_dtype,
                           const double* rotation, int bits, int norm_ord, uint64_t* out_codes, int mem, void* stream);

extern "C" int sq_itqfit_create(const void* x, int dtype, int64_t n, int d, int norm_ord, int mem, double* out_mean,
                                sq_handle_t* out) {
    if (!x || !out || !out_mean || n <= 1 || d <= 0) return fail(SQ_ERR_INVALID, "sq_itqfit_create: bad argument");
    if (dtype != SQ_DTYPE_F32 && dtype != SQ_DTYPE_F64) return fail(SQ_ERR_INVALID, "sq_itqfit_create: unknown dtype %d", dtype);
    if (norm_ord != SQ_NORM_NONE && norm_ord != SQ_NORM_L2)
        return fail(SQ_ERR_UNSUPPORTED, "sq_itqfit_create: normalize=%d not supported on the device (None or 2)", norm_ord);
    if (d > FIT_MAX_D) return fail(SQ_ERR_UNSUPPORTED, "sq_itqfit_create: d=%d above %d", d, FIT_MAX_D);
    auto* h = new FitHandle();
    h->kind = H_FIT;
    h->dtype = dtype;
    h->n = n;
    h->d = d;
    h->norm = norm_ord;
    auto bail = [&](int rc) {
        delete h;
        return rc;
    };
    if (hipGetDevice(&h->device) != hipSuccess) return bail(fail(SQ_ERR_HIP, "sq_itqfit_create: no HIP device"));
    const size_t esz = dtype == SQ_DTYPE_F32 ? 4 : 8;
    if (mem == SQ_MEM_DEVICE) {
        h->x = x;
    } else {
        int rc = h->owned.reserve((size_t)n * d * esz);
        if (rc != SQ_OK) return bail(rc);
        if (hipMemcpy(h->owned.p, x, (size_t)n * d * esz, hipMemcpyHostToDevice) != hipSuccess)
            return bail(fail(SQ_ERR_HIP, "sq_itqfit_create: H2D copy failed"));
        h->x = h->owned.p;
    }
    int rc;
    if ((rc = h->mean.reserve((size_t)d * 8)) != SQ_OK) return bail(rc);
    if ((rc = h->acc.reserve((size_t)std::max(d * d, FIT_MAX_BITS * FIT_MAX_BITS) * 8)) != SQ_OK) return bail(rc);
    if (norm_ord == SQ_NORM_L2) {
        if ((rc = h->nrm.reserve((size_t)n * 8)) != SQ_OK) return bail(rc);
        const unsigned g = (unsigned)((n + 31) / 32);
        if (dtype == SQ_DTYPE_F32)
            hipLaunchKernelGGL((fit_rownorm_kernel<float>), dim3(g), dim3(256), 0, 0, (const float*)h->x, (long long)n, d, h->nrm.as<double>());
        else
            hipLaunchKernelGGL((fit_rownorm_kernel<double>), dim3(g), dim3(256), 0, 0, (const double*)h->x, (long long)n, d, h->nrm.as<double>());
    }
    if (hipMemset(h->acc.p, 0, (size_t)d * 8) != hipSuccess) return bail(fail(SQ_ERR_HIP, "memset failed"));
    const long long rpb = 1024;
    const unsigned g = (unsigned)((n + rpb - 1) / rpb);
    const double* nrm = norm_ord == SQ_NORM_L2 ? h->nrm.as<double>() : nullptr;
    if (dtype == SQ_DTYPE_F32)
        hipLaunchKernelGGL((fit_colsum_kernel<float>), dim3(g), dim3(256), 0, 0, (const float*)h->x, (long long)n, d, rpb, nrm, h->acc.as<double>());
    else
        hipLaunchKernelGGL((fit_colsum_kernel<double>), dim3(g), dim3(256), 0, 0, (const double*)h->x, (long long)n, d, rpb, nrm, h->acc.as<double>());
    std::vector<double> sums((size_t)d);
    if (hipMemcpy(sums.data(), h->acc.p, (size_t)d * 8, hipMemcpyDeviceToHost) != hipSuccess)
        return bail(fail(SQ_ERR_HIP, "sq_itqfit_create: column means failed: %s", hipGetErrorString(hipGetLastError())));
    for (int k = 0; k < d; ++k) out_mean[k] = sums[(size_t)k] / (double)n;
    if (hipMemcpy(h->mean.p, out_mean, (size_t)d * 8, hipMemcpyHostToDevice) != hipSuccess)
        return bail(fail(SQ_ERR_HIP, "sq_itqfit_create: H2D copy failed"));
    *out = register_handle(h);
    return SQ_OK;
}

// The model's mean may be stored in the descriptors' dtype (numpy's np.mean of float32 data is float32):
// the caller passes back the values it will keep so that covariance and projection use exactly those.
extern "C" int sq_itqfit_set_mean(sq_handle_t hid, const double* mean) {
    auto* h = static_cast<FitHandle*>(lookup_handle(hid, H_FIT));
    if (!h || !mean) return fail(SQ_ERR_INVALID, "sq_itqfit_set_mean: bad argument");
    std::lock_guard<std::mutex> lock(h->mu);
    SQ_HIP(hipSetDevice(h->device));
    SQ_HIP(hipMemcpy(h->mean.p, mean, (size_t)h->d * 8, hipMemcpyHostToDevice));
    return SQ_OK;
}

extern "C" int sq_itqfit_cov(sq_handle_t hid, double* out_cov) {
    auto* h = static_cast<FitHandle*>(lookup_handle(hid, H_FIT));
    if (!h || !out_cov) return fail(SQ_ERR_INVALID, "sq_itqfit_cov: bad argument");
    std::lock_guard<std::mutex> lock(h->mu);
    SQ_HIP(hipSetDevice(h->device));
    const int d = h->d;
    SQ_HIP(hipMemset(h->acc.p, 0, (size_t)d * d * 8));
    SQ_TRY(h->dtype == SQ_DTYPE_F32 ? fit_gram_cov<float>(h, h->acc.as<double>()) : fit_gram_cov<double>(h, h->acc.as<double>()));
    SQ_HIP(hipMemcpy(out_cov, h->acc.p, (size_t)d * d * 8, hipMemcpyDeviceToHost));
    const double inv = 1.0 / (double)(h->n - 1);  // np.cov: ddof = 1
    for (long long i = 0; i < (long long)d * d; ++i) out_cov[i] *= inv;
    return SQ_OK;
}

extern "C" int sq_itqfit_project(sq_handle_t hid, const double* pc, int bits) {
    auto* h = static_cast<FitHandle*>(lookup_handle(hid, H_FIT));
    if (!h || !pc || bits <= 0 || bits > FIT_MAX_BITS)
        return fail(SQ_ERR_INVALID, "sq_itqfit_project: bad argument (1 <= bits <= %d)", FIT_MAX_BITS);
    std::lock_guard<std::mutex> lock(h->mu);
    SQ_HIP(hipSetDevice(h->device));
    const int d = h->d;
    h->bits = bits;
    SQ_TRY(h->v.reserve((size_t)h->n * bits * 8));
    SQ_TRY(h->small.reserve((size_t)std::max(d, bits) * bits * 8));
    SQ_HIP(hipMemcpy(h->small.p, pc, (size_t)d * bits * 8, hipMemcpyHostToDevice));
    // columns of pc per pass: as many as fit the LDS beside the 32 staged rows (multiples of 8)
    int bc = (int)((150 * 1024 / 8 - 32 * FIT_KC) / d) / 8 * 8;
    if (bc > 128) bc = 128;
    if (bc > (bits + 7) / 8 * 8) bc = (bits + 7) / 8 * 8;
    if (bc < 8) return fail(SQ_ERR_UNSUPPORTED, "sq_itqfit_project: d=%d exceeds the LDS", d);
    const size_t lds = ((size_t)d * bc + (size_t)32 * FIT_KC) * 8;
    const unsigned chunks = (unsigned)((bits + bc - 1) / bc);
    const double* nrm = h->norm == SQ_NORM_L2 ? h->nrm.as<double>() : nullptr;
    const unsigned g = (unsigned)std::min<long long>((h->n + 31) / 32, 4ll * cu_count(h->device));
    if (h->dtype == SQ_DTYPE_F32) {
        SQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&fit_project_kernel<float>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL((fit_project_kernel<float>), dim3(g, chunks), dim3(256), lds, 0, (const float*)h->x, nrm,
                           h->mean.as<double>(), h->small.as<double>(), h->n, d, bits, bc, h->v.as<double>());
    } else {
        SQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&fit_project_kernel<double>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL((fit_project_kernel<double>), dim3(g, chunks), dim3(256), lds, 0, (const double*)h->x, nrm,
                           h->mean.as<double>(), h->small.as<double>(), h->n, d, bits, bc, h->v.as<double>());
    }
    SQ_HIP(hipDeviceSynchronize());
    return SQ_OK;
}

// One ITQ iteration's O(n) work: ux = sign(v . r) (the hash kernel on v with a zero mean), c = ux^T . v.
extern "C" int sq_itqfit_iterate(sq_handle_t hid, const double* r, double* out_c) {
    auto* h = static_cast<FitHandle*>(lookup_handle(hid, H_FIT));
    if (!h || !r || !out_c) return fail(SQ_ERR_INVALID, "sq_itqfit_iterate: bad argument");
    if (h->bits <= 0) return fail(SQ_ERR_INVALID, "sq_itqfit_iterate: call sq_itqfit_project first");
    std::lock_guard<std::mutex> lock(h->mu);
    SQ_HIP(hipSetDevice(h->device));
    const int b = h->bits, words = (b + 63) / 64;
    DevBuf& codes = h->codes;
    DevBuf& zero = h->zero;
    auto done = [&](int rc) { return rc; };
    int rc;
    if ((rc = codes.reserve((size_t)h->n * words * 8)) != SQ_OK) return done(rc);
    if ((rc = zero.reserve((size_t)b * 8)) != SQ_OK) return done(rc);
    if (hipMemset(zero.p, 0, (size_t)b * 8) != hipSuccess || hipMemset(h->acc.p, 0, (size_t)b * b * 8) != hipSuccess ||
        hipMemcpy(h->small.p, r, (size_t)b * b * 8, hipMemcpyHostToDevice) != hipSuccess)
        return done(fail(SQ_ERR_HIP, "sq_itqfit_iterate: staging failed"));
    rc = sq_itq_hash(h->v.p, SQ_DTYPE_F64, h->n, b, zero.as<double>(), SQ_DTYPE_F64, h->small.as<double>(), b, SQ_NORM_NONE,
                     reinterpret_cast<uint64_t*>(codes.p), SQ_MEM_DEVICE, nullptr);
    if (rc != SQ_OK) return done(rc);
    const long long rpb = 2048;
    const unsigned grid = (unsigned)((h->n + rpb - 1) / rpb);
    const size_t lds = (size_t)2 * 16 * FIT_GT * 8;
    const unsigned tiles = (unsigned)((b + FIT_GT - 1) / FIT_GT);
    if (b <= 64)
        hipLaunchKernelGGL((fit_gram_kernel<double, 1, 4, 4>), dim3(grid, tiles, tiles), dim3(256), lds, 0, nullptr, nullptr,
                           nullptr, h->v.as<double>(), codes.as<u64>(), words, h->n, b, b, rpb, h->acc.as<double>());
    else
        hipLaunchKernelGGL((fit_gram_kernel<double, 1, 8, 8>), dim3(grid, tiles, tiles), dim3(256), lds, 0, nullptr, nullptr,
                           nullptr, h->v.as<double>(), codes.as<u64>(), words, h->n, b, b, rpb, h->acc.as<double>());
    if (hipMemcpy(out_c, h->acc.p, (size_t)b * b * 8, hipMemcpyDeviceToHost) != hipSuccess)
        return done(fail(SQ_ERR_HIP, "sq_itqfit_iterate: failed: %s", hipGetErrorString(hipGetLastError())));
    return done(SQ_OK);
}

extern "C" int sq_itqfit_destroy(sq_handle_t hid) {
    auto* h = remove_handle(hid, H_FIT);
    if (!h) return fail(SQ_ERR_INVALID, "sq_itqfit_destroy: unknown handle");
    (void)hipSetDevice(h->device);
    delete h;
    return SQ_OK;
}
