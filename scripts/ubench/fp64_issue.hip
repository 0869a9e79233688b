// Micro-benchmark (round 5): how fast does ONE wave per SIMD issue v_fma_f64 -- dependent chains of length 1, 2, 4, 8 interleaved -- and what
// does a wave-uniform s_load + s_waitcnt, a v_rcp_f64 + Newton chain, an LDS round trip cost in the same stream?  Cycles from s_memtime,
// the effective clock from the wall time of a long run on every SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>

template <int CH> __global__ __launch_bounds__(64, 1) void k_fma(double* out, const double* in, int iters, long long* cyc) {
    double a[CH], x = in[threadIdx.x & 7], y = in[8];
#pragma unroll
    for (int c = 0; c < CH; ++c) a[c] = in[c] + threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 64 / CH; ++r)
#pragma unroll
            for (int c = 0; c < CH; ++c) a[c] = __builtin_fma(a[c], x, y);
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += a[c];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// 64 FMAs (8 chains) + one reciprocal chain per iteration
__global__ __launch_bounds__(64, 1) void k_rcp(double* out, const double* in, int iters, long long* cyc) {
    double a[8], x = in[threadIdx.x & 7], y = in[8], s = in[9] + threadIdx.x;
#pragma unroll
    for (int c = 0; c < 8; ++c) a[c] = in[c] + threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < 8; ++c) a[c] = __builtin_fma(a[c], x, y);
        double r0 = __builtin_amdgcn_rcp(s + a[0]);
        double e = __builtin_fma(-(s + a[0]), r0, 1.0);
        r0 = __builtin_fma(r0, e, r0);
        e = __builtin_fma(-(s + a[0]), r0, 1.0);
        r0 = __builtin_fma(r0, e, r0);
#pragma unroll
        for (int c = 0; c < 8; ++c) a[c] *= r0;      // everything waits for the reciprocal
    }
    long long t1 = __builtin_readcyclecounter();
    double sum = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) sum += a[c];
    out[blockIdx.x * 64 + threadIdx.x] = sum;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// 64 FMAs + a scalar load of 16 dwords whose values are used right away
__global__ __launch_bounds__(64, 1) void k_sload(double* out, const double* in, const double* __restrict__ cst, int iters, long long* cyc) {
    double a[8], y = in[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) a[c] = in[c] + threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        const double* q = cst + (i & 7) * 8;      // (uniform address, changes per iteration: the loads cannot be hoisted)
        double k[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) k[c] = q[c];
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < 8; ++c) a[c] = __builtin_fma(a[c], k[c], y);
    }
    long long t1 = __builtin_readcyclecounter();
    double sum = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) sum += a[c];
    out[blockIdx.x * 64 + threadIdx.x] = sum;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const int nb = 1024, iters = 20000;
    double *out, *in, *cst;
    long long* cyc;
    CHK(hipMalloc(&out, nb * 64 * 8));
    CHK(hipMalloc(&in, 16 * 8));
    CHK(hipMalloc(&cst, 64 * 8));
    CHK(hipMalloc(&cyc, nb * 8));
    std::vector<double> h(64, 0.0);
    for (int i = 0; i < 64; ++i) h[i] = 0.999 + 1e-4 * i;
    h[8] = 1e-3;
    CHK(hipMemcpy(in, h.data(), 16 * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(cst, h.data(), 64 * 8, hipMemcpyHostToDevice));
    std::vector<long long> hc(nb);
    auto run = [&](const char* name, auto launch, double fma_per_iter, int blocks) {
        launch(blocks);
        (void)hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        launch(blocks);
        (void)hipDeviceSynchronize();
        auto t1 = std::chrono::steady_clock::now();
        (void)hipMemcpy(hc.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        double mc = 0;
        for (int i = 0; i < blocks; ++i) mc += (double)hc[i];
        mc /= blocks;
        const double wall = std::chrono::duration<double>(t1 - t0).count();
        printf("%-28s blocks %4d: %.2f counter ticks per FMA, wall %.3f ms -> %.3f ns per FMA per wave; ticks/s %.3e\n", name, blocks, mc / (iters * fma_per_iter), wall * 1e3,
               wall * 1e9 / (iters * fma_per_iter), mc / wall);
    };
    for (int blocks : {1, 1024}) {
        run("fma 1 chain", [&](int b) { hipLaunchKernelGGL(k_fma<1>, dim3(b), dim3(64), 0, 0, out, in, iters, cyc); }, 64, blocks);
        run("fma 2 chains", [&](int b) { hipLaunchKernelGGL(k_fma<2>, dim3(b), dim3(64), 0, 0, out, in, iters, cyc); }, 64, blocks);
        run("fma 4 chains", [&](int b) { hipLaunchKernelGGL(k_fma<4>, dim3(b), dim3(64), 0, 0, out, in, iters, cyc); }, 64, blocks);
        run("fma 8 chains", [&](int b) { hipLaunchKernelGGL(k_fma<8>, dim3(b), dim3(64), 0, 0, out, in, iters, cyc); }, 64, blocks);
        run("64 fma + rcp chain", [&](int b) { hipLaunchKernelGGL(k_rcp, dim3(b), dim3(64), 0, 0, out, in, iters, cyc); }, 64 + 5 + 8, blocks);
        run("64 fma + s_load x16", [&](int b) { hipLaunchKernelGGL(k_sload, dim3(b), dim3(64), 0, 0, out, in, cst, iters, cyc); }, 64, blocks);
    }
    // two waves per SIMD: 2048 blocks
    run("fma 1 chain, 2 waves/SIMD", [&](int b) { hipLaunchKernelGGL(k_fma<1>, dim3(b), dim3(64), 0, 0, out, in, iters, cyc); }, 64, 1024);
    {
        const int b2 = 2048;
        long long* cyc2;
        double* out2;
        CHK(hipMalloc(&cyc2, b2 * 8));
        CHK(hipMalloc(&out2, b2 * 64 * 8));
        for (int rep = 0; rep < 2; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(k_fma<1>, dim3(b2), dim3(64), 0, 0, out2, in, iters, cyc2);
            (void)hipDeviceSynchronize();
            auto t1 = std::chrono::steady_clock::now();
            if (rep) printf("fma 1 chain, 2048 blocks: wall %.3f ms\n", std::chrono::duration<double>(t1 - t0).count() * 1e3);
        }
        for (int rep = 0; rep < 2; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(k_fma<8>, dim3(b2), dim3(64), 0, 0, out2, in, iters, cyc2);
            (void)hipDeviceSynchronize();
            auto t1 = std::chrono::steady_clock::now();
            if (rep) printf("fma 8 chains, 2048 blocks: wall %.3f ms\n", std::chrono::duration<double>(t1 - t0).count() * 1e3);
        }
    }
    return 0;
}
