// Development microbenchmark (gfx950): the latency of dependent fp64 operations in ONE wave -- what bounds the sequential head of the one-launch
// kernel.  hipcc --offload-arch=gfx950 -O3 chain.hip -o chain && ./chain
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ double readlane_d(double x, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), l), hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
    return __hiloint2double(hi, lo);
}

template <int MODE>
__global__ void k(double* out, long long* cyc, int n, double a, double b) {
    double z = threadIdx.x * 1e-3, w = 0.5, v = 0.25;
    const long long t0 = clock64();
    for (int s = 0; s < n; ++s) {
        if (MODE == 0) {          // one dependent fma per iteration
            z = fma(a, z, b);
        } else if (MODE == 1) {   // four dependent fmas
            z = fma(a, z, b);
            z = fma(a, z, b);
            z = fma(a, z, b);
            z = fma(a, z, b);
        } else if (MODE == 2) {   // readlane (uniform index) then fma
            const int su = __builtin_amdgcn_readfirstlane(s & 63);
            z = fma(readlane_d(z, su), a, b);
        } else if (MODE == 3) {   // three independent chains
            z = fma(a, z, b);
            w = fma(a, w, b);
            v = fma(a, v, b);
        } else if (MODE == 4) {   // dependent add
            z = z + a;
        } else if (MODE == 5) {   // dependent fp32 fma
            float f = (float)z;
            f = fmaf((float)a, f, (float)b);
            z = f;
        }
    }
    const long long t1 = clock64();
    out[threadIdx.x] = z + w + v;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    double* out;
    long long *cyc, h;
    hipMalloc(&out, 64 * 8);
    hipMalloc(&cyc, 8);
    const int n = 4096;
    const char* names[] = {"1 dependent fma", "4 dependent fmas", "readlane + fma", "3 independent fmas", "dependent add", "cvt + fp32 fma + cvt"};
#define RUN(M)                                                        \
    for (int rep = 0; rep < 2; ++rep) {                               \
        hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), 0, 0, out, cyc, n, 0.999, 0.001); \
        hipDeviceSynchronize();                                       \
    }                                                                 \
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);                     \
    printf("%-24s %.1f cycles per iteration\n", names[M], (double)h / n);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
    return 0;
}
