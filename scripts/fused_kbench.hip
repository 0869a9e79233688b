// Timing harness for the persistent mid-d filter kernel (development tool): times dk_fused_filter<DP> on a synthetic LTI model,
// optionally with phases compiled out (-DFUSED_SKIP=1 predict GEMMs, 2 update, 4 log/div) to see where a step's cycles go.
#include "../temporalgps.jl_amd/csrc/tgp_dense.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace tgp_dense;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int DP> void run(int64_t T) {
    const int d = DP, p = 1, Pq = 16;
    std::vector<double> A((size_t)DP * DP, 0.0), Q((size_t)DP * DP, 0.0), H((size_t)Pq * DP, 0.0), a(DP, 0.01), h(Pq, 0.0), R(Pq, 0.2), x0((size_t)DP * DP + DP, 0.0), y(T);
    srand(3);
    for (int i = 0; i < DP; ++i) {
        for (int j = 0; j < DP; ++j) A[i + (size_t)j * DP] = (i == j ? 0.9 : 0.0) + 0.01 * (rand() / (double)RAND_MAX - 0.5);
        Q[i + (size_t)i * DP] = 0.1; x0[i + (size_t)i * DP] = 1.0; H[0 + (size_t)i * Pq] = rand() / (double)RAND_MAX - 0.5;
    }
    for (auto& v : y) v = rand() / (double)RAND_MAX - 0.5;
    double *dA, *dQ, *dH, *da, *dh, *dR, *dx0, *dy, *dres;
    CK(hipMalloc(&dA, A.size() * 8)); CK(hipMalloc(&dQ, Q.size() * 8)); CK(hipMalloc(&dH, H.size() * 8)); CK(hipMalloc(&da, a.size() * 8));
    CK(hipMalloc(&dh, h.size() * 8)); CK(hipMalloc(&dR, R.size() * 8)); CK(hipMalloc(&dx0, x0.size() * 8)); CK(hipMalloc(&dy, y.size() * 8)); CK(hipMalloc(&dres, 64));
    CK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dQ, Q.data(), Q.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dH, H.data(), H.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(da, a.data(), a.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dh, h.data(), h.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dR, R.data(), R.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dx0, x0.data(), x0.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dy, y.data(), y.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(dres, 0, 64));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dk_fused_filter<DP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FusedCfg<DP>::LDS_BYTES));
    FusedArgs g;
    g.T = T; g.step0 = 0; g.step1 = T; g.d = d; g.p = p; g.Pq = Pq; g.ordering = 0;
    g.A = dA; g.Q = dQ; g.H = dH; g.a = da; g.h = dh; g.R = dR; g.x0 = dx0; g.y = dy; g.result8 = dres;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(dk_fused_filter<DP>, dim3(1), dim3(256), FusedCfg<DP>::LDS_BYTES, 0, g);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        double res[8];
        CK(hipMemcpy(res, dres, 64, hipMemcpyDeviceToHost));
        printf("DP=%d T=%lld: %.3f us per step (lml %.6f)\n", DP, (long long)T, ms * 1e3 / T, res[0]);
    }
}
int main(int argc, char** argv) {
    const int64_t T = argc > 1 ? atoll(argv[1]) : 20000;
    run<32>(T);
    run<64>(T);
    return 0;
}
