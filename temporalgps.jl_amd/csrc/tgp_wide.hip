// Stationary-gain engine for wide states (16 < d <= 63) -- see tgp_wide.hpp.  gfx950 only (wave64).
#include "tgp_wide.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "tgp_alloc.hpp"

namespace tgp_wide {

namespace {

typedef double v2d __attribute__((ext_vector_type(2)));

struct ZArg {
    double z[64];      // the filtered mean behind the head, one component per lane (zero beyond d)
};

__device__ __forceinline__ void lds_sync() {      // one wave talking to itself through LDS (DS operations of a wave execute in order)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double readlane_d(double x, int l) {      // l wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), l), hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
    return __hiloint2double(hi, lo);
}

// tab: [DP + 2][64] -- column j of the lanes' rows (lane i < d: row i of Phi; lane d: -g; zero beyond), then the lanes' input gains (K_i; 1 for the
// observer) and constants (c_i; -g0 for the observer).  One wave per chunk [s0, s1) of the steps behind the head; its warm-up starts `halo` steps
// early from zero, or at the head's end from the head's own end state where that is nearer.
template <int DP>
__global__ __launch_bounds__(64) void k_wide_lml(const double* __restrict__ tab, const double* __restrict__ y, double hh, long long T, long long t_head,
                                                  long long chunk_len, long long halo, int obs_lane, ZArg z0, double* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) double zb[64];
    const int lane = threadIdx.x;
    const long long chunk = blockIdx.x;
    const long long s0 = t_head + chunk * chunk_len;
    long long s1 = s0 + chunk_len;
    if (s1 > T) s1 = T;
    const bool from_head = s0 - halo <= t_head;
    const long long w0 = from_head ? t_head : s0 - halo;
    double phi[DP];
#pragma unroll
    for (int j = 0; j < DP; ++j) phi[j] = tab[(size_t)j * 64 + lane];
    const double kin = tab[(size_t)DP * 64 + lane], cin = tab[(size_t)(DP + 1) * 64 + lane];
    zb[lane] = from_head ? z0.z[lane] : 0.0;
    lds_sync();
    double ssq = 0.0;
    double yn = (w0 + lane < s1) ? y[w0 + lane] : 0.0;
    for (long long tb = w0; tb < s1; tb += 64) {
        const double yv = yn;
        yn = (tb + 64 + lane < s1) ? y[tb + 64 + lane] : 0.0;      // (the next block's observations: on their way while this block runs)
        const int nb = (int)((s1 - tb < 64) ? (s1 - tb) : 64);
        for (int l = 0; l < nb; ++l) {
            const double u = readlane_d(yv, l) - hh;
            double a0 = fma(kin, u, cin), a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
            for (int j = 0; j < DP; j += 8) {
                const v2d q0 = *reinterpret_cast<const v2d*>(&zb[j]), q1 = *reinterpret_cast<const v2d*>(&zb[j + 2]);
                const v2d q2 = *reinterpret_cast<const v2d*>(&zb[j + 4]), q3 = *reinterpret_cast<const v2d*>(&zb[j + 6]);
                a0 = fma(phi[j], q0.x, a0);
                a1 = fma(phi[j + 1], q0.y, a1);
                a2 = fma(phi[j + 2], q1.x, a2);
                a3 = fma(phi[j + 3], q1.y, a3);
                a0 = fma(phi[j + 4], q2.x, a0);
                a1 = fma(phi[j + 5], q2.y, a1);
                a2 = fma(phi[j + 6], q3.x, a2);
                a3 = fma(phi[j + 7], q3.y, a3);
            }
            const double acc = (a0 + a1) + (a2 + a3);
            lds_sync();      // (every lane has read the old state)
            zb[lane] = acc;
            lds_sync();
            if (tb + l >= s0) ssq = fma(acc, acc, ssq);      // (the observer's acc is the step's innovation)
        }
    }
    const double s = readlane_d(ssq, obs_lane);
    if (lane == 0) part[chunk] = s;
}

// ---- host: small dense linear algebra, row-major ---------------------------------------------------------------------------------------
void matmul(int d, const double* X, const double* Y, double* Z) {      // Z = X Y
    for (int i = 0; i < d; ++i) {
        double* zi = Z + (size_t)i * d;
        for (int j = 0; j < d; ++j) zi[j] = 0.0;
        for (int k = 0; k < d; ++k) {
            const double x = X[(size_t)i * d + k];
            const double* yk = Y + (size_t)k * d;
            for (int j = 0; j < d; ++j) zi[j] += x * yk[j];
        }
    }
}
double norm_inf(int d, const double* X) {
    double n = 0.0;
    for (int i = 0; i < d; ++i) {
        double s = 0.0;
        for (int j = 0; j < d; ++j) s += std::fabs(X[(size_t)i * d + j]);
        n = std::max(n, s);
    }
    return n;
}

}  // namespace

struct Engine {
    Info info{};
    bool have = false;
    std::vector<double> key;
    long long key_T = -1;
    int d = 0, dp = 0;
    std::vector<double> A, avec, hvec;      // row-major A, a, h
    double hh = 0.0, g0 = 0.0, Sss = 0.0, sum_logS_head = 0.0;
    std::vector<double> x0m;
    std::vector<double> Kt, St;            // the head's gains [n0][d] and innovation variances [n0]
    std::vector<double> tab_host;          // the kernel's table (see k_wide_lml)
    double* tab_dev = nullptr;
    size_t tab_cap = 0;
    bool tab_current = false;
    double* pinned = nullptr;              // [kHeadMax] head observations | [kMaxChunks] the chunks' sums
    const char* kname = "k_wide_lml<32>";
};

Engine* create() { return new Engine(); }
void destroy(Engine* e) {
    if (!e) return;
    if (e->tab_dev) (void)tgp_alloc::dev_free(e->tab_dev);
    if (e->pinned) (void)tgp_alloc::host_free(e->pinned);
    delete e;
}
const Info& last_plan(const Engine* e) { return e->info; }
const char* kernel_name(const Engine* e) { return e->kname; }

namespace {
template <class F>
void model_words(const ModelHost& m, F&& f) {
    const size_t d = (size_t)m.d;
    f(m.A, d * d); f(m.a, d); f(m.Q, d * d); f(m.H, d); f(&m.hh, 1); f(&m.R, 1); f(m.x0m, d); f(m.x0P, d * d);
}
bool same_model(const Engine* e, const ModelHost& m, long long T) {
    if (!e->have || e->key_T != T || e->d != m.d) return false;
    const double* k = e->key.data();
    bool same = true;
    model_words(m, [&](const double* p, size_t n) {
        same = same && std::memcmp(k, p, n * sizeof(double)) == 0;
        k += n;
    });
    return same;
}
}  // namespace

bool plan(Engine* e, const ModelHost& m, long long T) {
    if (same_model(e, m, T)) {
        e->info.plan_ms = 0.0;
        return e->info.why == kOk;
    }
    static const bool cpu_ok = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");      // (this object's host code is built with both)
    if (!cpu_ok) {
        e->info = Info{};
        e->info.why = kAlloc;
        return false;
    }
    const auto t_begin = std::chrono::steady_clock::now();
    e->have = false;
    e->tab_current = false;
    e->info = Info{};
    const int d = m.d;
    const size_t dd = (size_t)d * d;
    e->d = d;
    e->dp = d <= 31 ? 32 : 64;
    e->kname = e->dp == 32 ? "k_wide_lml<32>" : "k_wide_lml<64>";
    e->A.assign(dd, 0.0);
    std::vector<double> Q(dd), P(dd), AP(dd), Pp(dd), Pf(dd), v(d);
    for (int i = 0; i < d; ++i)
        for (int k = 0; k < d; ++k) {
            e->A[(size_t)i * d + k] = m.A[i + (size_t)k * d];
            Q[(size_t)i * d + k] = 0.5 * (m.Q[i + (size_t)k * d] + m.Q[k + (size_t)i * d]);
            P[(size_t)i * d + k] = 0.5 * (m.x0P[i + (size_t)k * d] + m.x0P[k + (size_t)i * d]);
        }
    e->avec.assign(m.a, m.a + d);
    e->hvec.assign(m.H, m.H + d);
    e->x0m.assign(m.x0m, m.x0m + d);
    e->hh = m.hh;
    const double* A = e->A.data();
    const double* h = e->hvec.data();
    auto done = [&](int why) {
        e->info.why = why;
        e->info.plan_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
        // (a plan that declined is kept as well: the next call of the same model asks no more than the memcmp)
        e->key.clear();
        model_words(m, [&](const double* p, size_t n) { e->key.insert(e->key.end(), p, p + n); });
        e->key_T = T;
        e->have = true;
        return why == kOk;
    };
    // ---- the covariance half of lgssm.jl:99-165 to its fixed point: P <- A P A' + Q; S = h' P h + R; K = P h / S; P <- P - K S K'
    e->Kt.clear();
    e->St.clear();
    e->sum_logS_head = 0.0;
    int n0 = -1;
    double prev_chg = 1e300;
    for (int t = 0; t < kHeadMax; ++t) {
        matmul(d, A, P.data(), AP.data());
        double scale = 0.0;
        for (int i = 0; i < d; ++i)
            for (int j = 0; j <= i; ++j) {
                const double *x = AP.data() + (size_t)i * d, *yv = A + (size_t)j * d;
                double s = Q[(size_t)i * d + j];
                for (int k = 0; k < d; ++k) s += x[k] * yv[k];
                Pp[(size_t)i * d + j] = Pp[(size_t)j * d + i] = s;
            }
        double S = m.R;
        for (int i = 0; i < d; ++i) {
            double s = 0.0;
            for (int k = 0; k < d; ++k) s += Pp[(size_t)i * d + k] * h[k];
            v[i] = s;
            S += h[i] * s;
        }
        if (!(S > 0.0) || !std::isfinite(S)) return done(kNotPD);
        const double iS = 1.0 / S;
        double chg = 0.0;
        for (int i = 0; i < d; ++i)
            for (int j = 0; j <= i; ++j) {
                const double pf = Pp[(size_t)i * d + j] - v[i] * v[j] * iS;
                chg = std::max(chg, std::fabs(pf - P[(size_t)i * d + j]));
                scale = std::max(scale, std::fabs(pf));
                Pf[(size_t)i * d + j] = Pf[(size_t)j * d + i] = pf;
            }
        for (int i = 0; i < d; ++i) e->Kt.push_back(v[i] * iS);
        e->St.push_back(S);
        e->sum_logS_head += std::log(S);
        P.swap(Pf);
        // settled: the step changes nothing beyond rounding -- a few ulps of the largest entry, or no longer shrinking at the rounding floor
        if (chg <= 4.0 * 2.220446049250313e-16 * scale || (t >= 16 && chg >= prev_chg && chg <= 1e-13 * scale)) {
            n0 = t + 1;
            break;
        }
        prev_chg = chg;
    }
    if (n0 < 0) return done(kNotSettled);
    e->info.n0 = n0;
    e->info.nhs = n0;
    e->Sss = e->St.back();
    const double* K = e->Kt.data() + (size_t)(n0 - 1) * d;
    // ---- Phi = (I - K h') A = A - K g', g = A' h; c = a - K g0, g0 = h . a
    std::vector<double> g(d, 0.0), Phi(dd);
    for (int k = 0; k < d; ++k)
        for (int j = 0; j < d; ++j) g[j] += h[k] * A[(size_t)k * d + j];
    double g0 = 0.0;
    for (int k = 0; k < d; ++k) g0 += h[k] * e->avec[k];
    e->g0 = g0;
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) Phi[(size_t)i * d + j] = A[(size_t)i * d + j] - K[i] * g[j];
    // ---- halo: the smallest tested k with |Phi^k|_inf <= 2^-60 (squarings, then the lower bits of the exponent)
    {
        const double thr = std::ldexp(1.0, -60);
        std::vector<std::vector<double>> pw;
        pw.push_back(Phi);
        int j = 0;
        while (norm_inf(d, pw.back().data()) > thr) {
            if (j >= 20) return done(kSlowMixing);
            std::vector<double> sq(dd);
            matmul(d, pw.back().data(), pw.back().data(), sq.data());
            pw.push_back(std::move(sq));
            ++j;
        }
        long long halo = 1LL << j;
        if (j >= 2) {
            std::vector<double> cur = pw[j - 1], cand(dd);
            long long ex = 1LL << (j - 1);
            int b = j - 2;
            const int bmin = std::max(0, j - 5);
            for (; b >= bmin; --b) {
                matmul(d, cur.data(), pw[b].data(), cand.data());
                if (norm_inf(d, cand.data()) > thr) {
                    cur = cand;
                    ex += 1LL << b;
                }
            }
            halo = ex + (1LL << bmin);
        }
        e->info.halo = (int)halo;
    }
    const long long Tb = T - n0;      // steps behind the head
    if (Tb < 64) return done(kTooShort);
    // chunks: as many waves as the chip holds several times over, none shorter than 64 steps
    long long chunks = std::min<long long>(kMaxChunks, Tb / 64);
    long long len = (Tb + chunks - 1) / chunks;
    chunks = (Tb + len - 1) / len;
    e->info.chunks = chunks;
    e->info.chunk_len = len;
    // ---- the kernel's table
    const int DP = e->dp;
    e->tab_host.assign((size_t)(DP + 2) * 64, 0.0);
    for (int i = 0; i < d; ++i) {
        for (int j = 0; j < d; ++j) e->tab_host[(size_t)j * 64 + i] = Phi[(size_t)i * d + j];
        e->tab_host[(size_t)DP * 64 + i] = K[i];
        e->tab_host[(size_t)(DP + 1) * 64 + i] = e->avec[i] - K[i] * g0;
    }
    for (int j = 0; j < d; ++j) e->tab_host[(size_t)j * 64 + d] = -g[j];      // the observer: r = u - g . z - g0
    e->tab_host[(size_t)DP * 64 + d] = 1.0;
    e->tab_host[(size_t)(DP + 1) * 64 + d] = -g0;
    return done(kOk);
}

int logpdf(Engine* e, hipStream_t stream, const double* y, long long T, double* lml_out, bool* not_pd, std::string* err) {
    *not_pd = false;
    auto fail = [&](hipError_t rc, const char* what) {
        if (err) *err = std::string("tgp_wide: ") + what + ": " + hipGetErrorString(rc);
        return (int)rc;
    };
    if (!e->have || e->info.why != kOk) return fail(hipErrorInvalidValue, "no plan");
    const int d = e->d, DP = e->dp, n0 = e->info.n0;
    hipError_t rc;
    if (!e->pinned) {
        rc = tgp_alloc::host_malloc(reinterpret_cast<void**>(&e->pinned), (size_t)(kHeadMax + kMaxChunks) * sizeof(double), hipHostMallocDefault);
        if (rc != hipSuccess) return fail(rc, "pinned buffer");
    }
    const size_t tab_bytes = e->tab_host.size() * sizeof(double);
    if (tab_bytes > e->tab_cap) {
        if (e->tab_dev) (void)tgp_alloc::dev_free(e->tab_dev);
        e->tab_dev = nullptr;
        e->tab_cap = 0;
        rc = tgp_alloc::dev_malloc(reinterpret_cast<void**>(&e->tab_dev), tab_bytes);
        if (rc != hipSuccess) return fail(rc, "table");
        e->tab_cap = tab_bytes;
        e->tab_current = false;
    }
    if (!e->tab_current) {
        rc = hipMemcpyAsync(e->tab_dev, e->tab_host.data(), tab_bytes, hipMemcpyHostToDevice, stream);
        if (rc != hipSuccess) return fail(rc, "table upload");
        e->tab_current = true;
    }
    double* yh = e->pinned;
    double* part = e->pinned + kHeadMax;
    rc = hipMemcpyAsync(yh, y, (size_t)n0 * sizeof(double), hipMemcpyDeviceToHost, stream);
    if (rc != hipSuccess) return fail(rc, "head observations");
    rc = hipStreamSynchronize(stream);
    if (rc != hipSuccess) return fail(rc, "head observations");
    // ---- the head: lgssm.jl:147-165 with the plan's gains
    ZArg z0;
    for (int i = 0; i < 64; ++i) z0.z[i] = 0.0;
    double quad = 0.0;
    {
        std::vector<double> mcur(e->x0m), mp(d);
        const double *A = e->A.data(), *h = e->hvec.data();
        for (int t = 0; t < n0; ++t) {
            double pred = e->hh;
            for (int i = 0; i < d; ++i) {
                double s = e->avec[i];
                const double* ai = A + (size_t)i * d;
                for (int k = 0; k < d; ++k) s += ai[k] * mcur[k];
                mp[i] = s;
                pred += h[i] * s;
            }
            const double r = yh[t] - pred;
            quad += r * r / e->St[t];
            const double* K = e->Kt.data() + (size_t)t * d;
            for (int i = 0; i < d; ++i) mcur[i] = mp[i] + K[i] * r;
        }
        for (int i = 0; i < d; ++i) z0.z[i] = mcur[i];
    }
    const long long chunks = e->info.chunks;
    if (DP == 32)
        hipLaunchKernelGGL(k_wide_lml<32>, dim3((unsigned)chunks), dim3(64), 0, stream, e->tab_dev, y, e->hh, T, (long long)n0, e->info.chunk_len, (long long)e->info.halo, d, z0,
                           part);
    else
        hipLaunchKernelGGL(k_wide_lml<64>, dim3((unsigned)chunks), dim3(64), 0, stream, e->tab_dev, y, e->hh, T, (long long)n0, e->info.chunk_len, (long long)e->info.halo, d, z0,
                           part);
    rc = hipGetLastError();
    if (rc != hipSuccess) return fail(rc, "launch");
    rc = hipStreamSynchronize(stream);
    if (rc != hipSuccess) return fail(rc, "kernel");
    double ssq = 0.0;
    for (long long c = 0; c < chunks; ++c) ssq += part[c];
    const double kLog2Pi = 1.8378770664093454835606594728112;
    *lml_out = -0.5 * ((double)T * kLog2Pi + e->sum_logS_head + (double)(T - n0) * std::log(e->Sss) + quad + ssq / e->Sss);
    if (!std::isfinite(*lml_out) && std::isfinite(ssq)) *not_pd = true;
    return 0;
}

}  // namespace tgp_wide
