// Read-only streaming microbenchmark (round 6): what ONE launch that reads 80 MB once (the logpdf call's y) can reach on gfx950, by access pattern.
//   grid     : many small workgroups, thread t of block b reads consecutive 16-byte pieces (the usual copy-kernel pattern), U loads in flight per lane
//   runs     : 2048 persistent waves, each streaming its own contiguous run in tiles of 16 KB (tgp_lml.hip's first pattern)
//   sweep    : 2048 persistent waves, tile k of the series goes to wave k mod 2048 (all waves read one moving window of 32 MB)
//   wgsweep  : 256 workgroups, the workgroup's 8 waves read 8 consecutive tiles, then the workgroup moves on by 256 x 8 tiles
// Each variant: time of one launch between events on an otherwise idle stream (min / median of 30), and back-to-back throughput.
// Build: hipcc -O3 --offload-arch=gfx950 scripts/ubench/read_bw.hip -o scripts/ubench/read_bw
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double v2d __attribute__((ext_vector_type(2)));

template <int U>
__global__ __launch_bounds__(256) void k_grid(const v2d* __restrict__ p, long long n16, double* out) {
    // block b reads the pieces [b * 256 * U, (b + 1) * 256 * U)
    const long long base = (long long)blockIdx.x * 256 * U + threadIdx.x;
    v2d v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
        const long long i = base + (long long)k * 256;
        v[k] = i < n16 ? __builtin_nontemporal_load(p + i) : v2d{0.0, 0.0};
    }
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < U; ++k) s += v[k].x + v[k].y;
    if (s == 1.2345e300) out[blockIdx.x] = s;
}

// MODE 0: runs, 1: sweep, 2: wgsweep
template <int MODE, int PPL>
__global__ __launch_bounds__(512, 2) void k_persist(const v2d* __restrict__ p, long long ntiles, double* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long W = (long long)gridDim.x * 8, w = (long long)blockIdx.x * 8 + wave;
    constexpr int TP = 64 * PPL;      // pieces per tile
    double s = 0.0;
    v2d st[PPL];
    auto tile_of = [&](long long k) -> long long {      // the k-th tile of this wave
        if (MODE == 0) return w * ntiles / W + k;
        if (MODE == 1) return w + k * W;
        return (long long)blockIdx.x * 8 + wave + k * W;
    };
    const long long mine = MODE == 0 ? (w + 1) * ntiles / W - w * ntiles / W : (ntiles - w + W - 1) / W;
    if (mine <= 0) return;
    auto issue = [&](long long t) {
        const v2d* src = p + t * TP;
#pragma unroll
        for (int k = 0; k < PPL; ++k) st[k] = __builtin_nontemporal_load(src + k * 64 + lane);
    };
    issue(tile_of(0));
    for (long long k = 0; k < mine; ++k) {
        double a = 0.0;
#pragma unroll
        for (int j = 0; j < PPL; ++j) a += st[j].x + st[j].y;
        if (k + 1 < mine) issue(tile_of(k + 1));
        s += a;
    }
    if (s == 1.2345e300) out[w] = s;
}

int main(int argc, char** argv) {
    const long long n = argc > 1 ? atoll(argv[1]) : 10000000;      // doubles
    const long long n16 = n / 2;
    v2d* d;
    double* out;
    hipMalloc(&d, n16 * 16 + (1 << 20));
    hipMalloc(&out, 1 << 20);
    hipMemset(d, 0, n16 * 16 + (1 << 20));
    hipStream_t st;
    hipStreamCreate(&st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipStreamSynchronize(st);
        std::vector<float> ms;
        for (int r = 0; r < 30; ++r) {
            hipEventRecord(e0, st);
            launch();
            hipEventRecord(e1, st);
            hipStreamSynchronize(st);
            float t;
            hipEventElapsedTime(&t, e0, e1);
            ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        hipEventRecord(e0, st);
        for (int r = 0; r < 50; ++r) launch();
        hipEventRecord(e1, st);
        hipStreamSynchronize(st);
        float tb;
        hipEventElapsedTime(&tb, e0, e1);
        printf("%-34s one launch min %6.2f median %6.2f us (%5.2f TB/s at the median); back to back %6.2f us (%5.2f TB/s)\n", name, ms[0] * 1e3, ms[15] * 1e3,
               n * 8.0 / (ms[15] * 1e-3) / 1e12, tb / 50 * 1e3, n * 8.0 / (tb / 50 * 1e-3) / 1e12);
    };
    hipLaunchKernelGGL(k_grid<1>, dim3(1), dim3(256), 0, st, d, 0, out);
    run("empty kernel (1 block)", [&] { hipLaunchKernelGGL(k_grid<1>, dim3(1), dim3(256), 0, st, d, 0LL, out); });
    run("grid U=4", [&] { hipLaunchKernelGGL(k_grid<4>, dim3((unsigned)((n16 + 1023) / 1024)), dim3(256), 0, st, d, n16, out); });
    run("grid U=8", [&] { hipLaunchKernelGGL(k_grid<8>, dim3((unsigned)((n16 + 2047) / 2048)), dim3(256), 0, st, d, n16, out); });
    run("grid U=16", [&] { hipLaunchKernelGGL(k_grid<16>, dim3((unsigned)((n16 + 4095) / 4096)), dim3(256), 0, st, d, n16, out); });
    const long long nt16 = n16 / (64 * 16), nt8 = n16 / (64 * 8);
    run("runs, 16 KB tiles, 256 wg", [&] { hipLaunchKernelGGL((k_persist<0, 16>), dim3(256), dim3(512), 0, st, d, nt16, out); });
    run("sweep, 16 KB tiles, 256 wg", [&] { hipLaunchKernelGGL((k_persist<1, 16>), dim3(256), dim3(512), 0, st, d, nt16, out); });
    run("runs, 8 KB tiles, 256 wg", [&] { hipLaunchKernelGGL((k_persist<0, 8>), dim3(256), dim3(512), 0, st, d, nt8, out); });
    run("sweep, 8 KB tiles, 256 wg", [&] { hipLaunchKernelGGL((k_persist<1, 8>), dim3(256), dim3(512), 0, st, d, nt8, out); });
    run("runs, 8 KB tiles, 512 wg", [&] { hipLaunchKernelGGL((k_persist<0, 8>), dim3(512), dim3(512), 0, st, d, nt8, out); });
    run("sweep, 8 KB tiles, 512 wg", [&] { hipLaunchKernelGGL((k_persist<1, 8>), dim3(512), dim3(512), 0, st, d, nt8, out); });
    run("runs, 16 KB tiles, 128 wg", [&] { hipLaunchKernelGGL((k_persist<0, 16>), dim3(128), dim3(512), 0, st, d, nt16, out); });
    run("sweep, 16 KB tiles, 128 wg", [&] { hipLaunchKernelGGL((k_persist<1, 16>), dim3(128), dim3(512), 0, st, d, nt16, out); });
    return 0;
}
