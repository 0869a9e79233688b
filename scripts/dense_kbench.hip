// Kernel-level benchmark / trace harness for the dense path (development tool, not part of the product library):
// includes the product's kernels verbatim and times them in isolation on synthetic data.
#define DK_TRACE 1
#include "../temporalgps.jl_amd/csrc/tgp_dense.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace tgp_dense;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int n = 256, D = 768;
    std::vector<double> S((size_t)n * n);
    srand(1);
    std::vector<double> X((size_t)n * n);
    for (auto& v : X) v = (rand() / (double)RAND_MAX - 0.5);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0;
            for (int k = 0; k < n; ++k) s += X[i * n + k] * X[j * n + k];
            S[i + (size_t)j * n] = s / n + (i == j ? 0.5 : 0.0);
        }
    double *dS, *dL, *dDinv, *dscal, *dV, *dB;
    long long* dtr;
    CK(hipMalloc(&dS, S.size() * 8)); CK(hipMalloc(&dL, S.size() * 8)); CK(hipMalloc(&dDinv, 16 * 256 * 8)); CK(hipMalloc(&dscal, 64));
    CK(hipMalloc(&dtr, 16 * 4 * 2 * 8));
    CK(hipMalloc(&dV, (size_t)n * (D + 16) * 8 + 512)); CK(hipMalloc(&dB, (size_t)n * (D + 16) * 8 + 512));
    CK(hipMemcpy(dS, S.data(), S.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(dL, 0, S.size() * 8));
    CK(hipMemset(dV, 0, (size_t)n * (D + 16) * 8));
    int* dslots;
    {
        std::vector<int> tab;
        chol_slot_table(n / 16, tab);
        CK(hipMalloc(&dslots, tab.size() * 4));
        CK(hipMemcpy(dslots, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(dk_chol, dim3(1), dim3(1024), 0, 0, dS, (int64_t)n, n, 0.0, dslots, dL, (int64_t)n, dDinv, dscal, dtr);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("dk_chol: %.2f us per launch\n", ms * 100.0);
    }
    std::vector<long long> tr(16 * 4 * 2);
    CK(hipMemcpy(tr.data(), dtr, tr.size() * 8, hipMemcpyDeviceToHost));
    printf("panel:  F (chain wave) | gap to next F || worker wave 1: trsm phase + barrier | update (d2) + wait   [cycles]\n");
    auto T = [&](int kb, int slot) { return tr[(kb * 4 + slot) * 2 + (slot >= 2 ? 1 : 0)]; };
    for (int kb = 0; kb < 16; ++kb)
        printf(" kb=%2d: %6lld %6lld || %6lld %6lld\n", kb, T(kb, 1) - T(kb, 0), kb < 15 ? T(kb + 1, 0) - T(kb, 1) : 0LL, T(kb, 3) - T(kb, 2),
               kb < 15 ? T(kb + 1, 2) - T(kb, 3) : 0LL);
    std::vector<double> Lh(S.size());
    CK(hipMemcpy(Lh.data(), dL, S.size() * 8, hipMemcpyDeviceToHost));
    double err = 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = 0;
            for (int k = 0; k <= j; ++k) s += Lh[i + (size_t)k * n] * Lh[j + (size_t)k * n];
            err = fmax(err, fabs(s - S[i + (size_t)j * n]));
        }
    printf("max |L L' - S| = %.3e\n", err);
    // trsm
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(dk_trsm, dim3((D + 16) / 16), dim3(256), 0, 0, dL, (int64_t)n, dDinv, dV, (int64_t)n, n, dB, (int64_t)(D + 16));
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("dk_trsm: %.2f us per launch\n", ms * 100.0);
    }
    // gemm 768^3, warm
    double *dA, *dP, *dC;
    CK(hipMalloc(&dA, (size_t)D * D * 8 + 512)); CK(hipMalloc(&dP, (size_t)D * D * 8 + 512)); CK(hipMalloc(&dC, (size_t)D * D * 8 + 512));
    CK(hipMemset(dA, 0, (size_t)D * D * 8)); CK(hipMemset(dP, 0, (size_t)D * D * 8));
    Engine* e = create(0);
    set_attrs(e);
    for (int cfg = 0; cfg < 4; ++cfg) {
        GemmArgs g;
        g.A = dA; g.lda = D; g.B = dP; g.ldb = D; g.C = dC; g.ldc = D;
        if (cfg == 0) { g.M = g.N = g.K = D; }
        if (cfg == 1) { g.M = 256; g.N = D; g.K = D; g.lda = 256; g.ldc = 256; }
        if (cfg == 2) { g.M = 256; g.N = 256; g.K = D; g.lda = 256; g.ldb = 256; g.ldc = 256; }
        if (cfg == 3) { g.M = D; g.N = D; g.K = 256; g.E = dC; g.lde = D; g.sign = -1.0; }      // the shape of P = Pp - B'B (without its rider)
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 20; ++i) launch_gemm(g, 0);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double fl = 2.0 * g.M * g.N * g.K;
            printf("gemm %dx%dx%d: %.2f us per launch, %.1f TF/s\n", g.M, g.N, g.K, ms * 50.0, fl / (ms * 50e-6) / 1e12);
        }
    }
    return 0;
}
