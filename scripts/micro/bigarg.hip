// Development: does a kernel-argument segment beyond 4 KB launch on gfx950?  (a by-value struct of N doubles, read in place; hipcc --offload-arch=gfx950 -O2 -o bigarg bigarg.hip: 3.2 ... 12 KB all launch and sum correctly)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N> struct Big { double v[N]; double* out; };
template <int N> __global__ void k(const Big<N> by_value) {
    (void)by_value;
    const Big<N>& a = *(const Big<N>*)__builtin_amdgcn_kernarg_segment_ptr();
    double s = 0.0;
    for (int i = threadIdx.x; i < N; i += 64) s += a.v[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if (threadIdx.x == 0) a.out[0] = s;
}
template <int N> void run() {
    Big<N> b;
    for (int i = 0; i < N; ++i) b.v[i] = 1.0 + i;
    double* d;
    hipMalloc(&d, 8);
    b.out = d;
    hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, b);
    hipError_t e = hipGetLastError();
    double h = -1.0;
    hipError_t e2 = hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("N=%d bytes=%zu launch=%s copy=%s sum=%.1f expect=%.1f\n", N, sizeof(b), hipGetErrorString(e), hipGetErrorString(e2), h, N * (N + 1) / 2.0);
}
int main() { run<400>(); run<511>(); run<600>(); run<830>(); run<1500>(); return 0; }
