// fp64 MFMA issue rate / dependent latency on gfx950: v_mfma_f64_16x16x4_f64 (run on the GPU box).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC> __global__ void k(double* out, int iters, double a, double b) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double x = a + threadIdx.x * 1e-9, y = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> void run(int waves_per_simd) {
    double* out;
    const int blocks = 256 * 4, threads = 64 * 4 * waves_per_simd / 4;   // 4 blocks per CU
    hipMalloc(&out, (size_t)blocks * threads * 8);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<NACC><<<blocks, threads>>>(out, 100, 1.0, 2.0);
    hipEventRecord(e0);
    k<NACC><<<blocks, threads>>>(out, iters, 1.0, 2.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double nm = (double)blocks * (threads / 64) * iters * NACC;
    const double tf = nm * 2048.0 / (ms * 1e-3) / 1e12;
    // cycles per MFMA per SIMD at 2.4 GHz: waves per SIMD = blocks*threads/64 / 1024
    const double per_simd = nm / 1024.0;
    printf("NACC=%d waves/SIMD=%d: %.3f ms, %.1f TF/s, %.1f cycles/MFMA/SIMD @2.4GHz\n", NACC, waves_per_simd, ms, tf, ms * 1e-3 * 2.4e9 / per_simd);
    hipFree(out);
}
int main() {
    run<1>(1); run<2>(1); run<4>(1); run<8>(1); run<4>(2); run<1>(4); run<9>(2);
    return 0;
}
