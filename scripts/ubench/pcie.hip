// PCIe microbenchmark for the host-staging design (tgp_stage.hpp): pinned vs pageable vs registered-in-place, 80 MB in / 160 MB out.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t n = 80u << 20;
    void *dev, *pin;
    hipMalloc(&dev, 2 * n);
    hipHostMalloc(&pin, 2 * n, hipHostMallocDefault);
    char* pg = (char*)aligned_alloc(4096, 2 * n);
    memset(pg, 1, 2 * n);
    memset(pin, 1, 2 * n);
    hipStream_t st;
    hipStreamCreate(&st);
    auto t = [&](const char* name, size_t bytes, auto fn) {
        fn();
        hipStreamSynchronize(st);
        double best = 1e9;
        for (int r = 0; r < 5; ++r) {
            double t0 = now();
            fn();
            hipStreamSynchronize(st);
            double dt = now() - t0;
            if (dt < best) best = dt;
        }
        printf("%-44s %7.2f ms  %6.1f GB/s\n", name, best * 1e3, bytes / best / 1e9);
    };
    t("H2D pinned 80 MB", n, [&] { hipMemcpyAsync(dev, pin, n, hipMemcpyHostToDevice, st); });
    t("D2H pinned 160 MB", 2 * n, [&] { hipMemcpyAsync(pin, dev, 2 * n, hipMemcpyDeviceToHost, st); });
    t("H2D pageable 80 MB", n, [&] { hipMemcpyAsync(dev, pg, n, hipMemcpyHostToDevice, st); });
    t("D2H pageable 160 MB", 2 * n, [&] { hipMemcpyAsync(pg, dev, 2 * n, hipMemcpyDeviceToHost, st); });
    t("hipHostRegister + unregister 80 MB", n, [&] { hipHostRegister(pg, n, hipHostRegisterDefault); hipHostUnregister(pg); });
    t("register + H2D + unregister 80 MB", n, [&] { hipHostRegister(pg, n, hipHostRegisterDefault); hipMemcpyAsync(dev, pg, n, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); hipHostUnregister(pg); });
    t("register + D2H + unregister 160 MB", 2 * n, [&] { hipHostRegister(pg, 2 * n, hipHostRegisterDefault); hipMemcpyAsync(pg, dev, 2 * n, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); hipHostUnregister(pg); });
    for (int nt : {1, 2, 4, 8, 16, 32}) {
        char name[64];
        snprintf(name, sizeof name, "host memcpy pageable -> pinned, %d threads", nt);
        t(name, n, [&] {
            std::vector<std::thread> th;
            const size_t per = n / nt;
            for (int i = 0; i < nt; ++i) th.emplace_back([&, i] { memcpy((char*)pin + i * per, pg + i * per, per); });
            for (auto& x : th) x.join();
        });
        snprintf(name, sizeof name, "host memcpy pinned -> pageable, %d threads", nt);
        t(name, n, [&] {
            std::vector<std::thread> th;
            const size_t per = n / nt;
            for (int i = 0; i < nt; ++i) th.emplace_back([&, i] { memcpy(pg + i * per, (char*)pin + i * per, per); });
            for (auto& x : th) x.join();
        });
    }
    // zero-copy: a kernel reading pinned host memory directly
    return 0;
}
